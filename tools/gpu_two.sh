ulimit -c 0
for lib in libmi_physics.so libmi_physics_w2.so; do
  for l in 0 54000; do
  echo "lib $lib lds $l"
  MI_FLOW_LDS=$l MI_PHYSICS_LIB=d3d12renderer_amd/$lib timeout 200 python bench.py --steps 20 --warmup 245 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['stage_ms']['solve'],3))"
  done
done
