ulimit -c 0
bash tools/gpu_debug.sh 2>&1 | tail -2
MI_ASYNC=0 timeout 120 python /tmp/dbg.py 2>&1 | tail -1
for a in 1 0; do
echo "async $a"
MI_ASYNC=$a timeout 200 python bench.py --steps 30 --warmup 245 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['ms_per_step'], {k:round(v,3) for k,v in d['stage_ms'].items()})"
done
