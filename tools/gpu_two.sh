ulimit -c 0
bash tools/gpu_debug.sh 2>&1 | tail -1
timeout 200 python bench.py --steps 20 --warmup 245 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['ms_per_step'], {k:round(v,3) for k,v in d['stage_ms'].items()})"
