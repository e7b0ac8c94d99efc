#!/bin/bash
# round 5: the whole GPU suite at HEAD, the driver-flag bench (default / -DMI_NO_DIET), the visit stamps of the persistent solver
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
cd oracle && make >/dev/null 2>&1; cd ..
T0=$(date +%s)
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/r5d_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5d_pytest.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/r5d_pytest.log | tail -12
echo "suite took $(( $(date +%s) - T0 )) s"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest"
for v in default nodiet default; do
  if [ $v = nodiet ]; then export MI_PHYSICS_LIB=$PWD/build_exp/libmi_physics_nodiet.so; else unset MI_PHYSICS_LIB; fi
  timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), 'steps/s solver', round(d['roofline']['avg_launch_us'],1), {k:round(v,3) for k,v in d['stage_ms'].items()})"
done
unset MI_PHYSICS_LIB
bash tools/gpu_timeline2.sh > gpurun_out/r5d_hop_stamps.txt 2>&1; tail -12 gpurun_out/r5d_hop_stamps.txt
echo "all done at $(( $(date +%s) - T0 )) s"
