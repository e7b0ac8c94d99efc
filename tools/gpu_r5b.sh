#!/bin/bash
# round 5, second GPU call: the whole GPU suite (every failure, not just the first), A/B of the switches on the driver-flag bench, kernel timeline, the host-boundary rates
# (with the one-frame-behind view), the other configurations
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
cd oracle && make >/dev/null 2>&1; cd ..
T0=$(date +%s)
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/r5b_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5b_pytest.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/r5b_pytest.log | tail -12
echo "suite took $(( $(date +%s) - T0 )) s"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest"
run() {  # name, env...
  local name=$1; shift
  ( for kv in "$@"; do export "$kv"; done; timeout 300 $B 2>gpurun_out/r5b_$name.err | tail -1 > gpurun_out/r5b_$name.json )
  python - "$name" <<'PY'
import sys, json
n = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r5b_{n}.json"))
    print(f"{n:16s} {d['value']:8.1f} steps/s  {d['ms_per_step']:.4f} ms  dev {d['device_ms_per_step']:.4f}  solver {d['roofline']['avg_launch_us']:.1f} us  contacts {d['config']['contacts']}  reruns {d['step_modes_timed']['synchronous_reruns']}  " + " ".join(f"{k[:6]}={v:.3f}" for k, v in d['stage_ms'].items()))
except Exception as e:
    print(n, "FAILED", e)
PY
}
run default
run nodiet MI_PHYSICS_LIB=$PWD/build_exp/libmi_physics_nodiet.so
run noreset MI_FUSE_RESET=0
run default2
for v in build_exp/libmi_physics_poll*.so; do [ -f "$v" ] && run $(basename $v .so | sed 's/libmi_physics_//') MI_PHYSICS_LIB=$PWD/$v; done
run default3
echo "benches done at $(( $(date +%s) - T0 )) s"
bash tools/gpu_timeline.sh 2>&1 | tail -1; cp gpurun_out/timeline.txt gpurun_out/r5b_step_timeline.txt
timeout 400 python tools/gpu_pcie_rate.py > gpurun_out/r5b_host_boundary_rate.json 2> gpurun_out/r5b_host_boundary_rate.err; python -c "
import json; d=json.load(open('gpurun_out/r5b_host_boundary_rate.json')); print({k:(v['steps_per_s'] if isinstance(v,dict) and 'steps_per_s' in v else None) for k,v in d.items()})" 2>&1 | cut -c1-900
bash tools/gpu_cfgs.sh 2>&1 | tail -9 | cut -c1-260; cp gpurun_out/cfgs.json gpurun_out/r5b_other_configs.json
echo "all done at $(( $(date +%s) - T0 )) s"
