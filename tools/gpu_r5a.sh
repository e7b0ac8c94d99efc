#!/bin/bash
# round 5, first GPU call: the full GPU suite on the new step (reset in the publish kernel, colouring round 0 in k_emit_manifolds, both pair passes in one launch, force
# integration as guests of k_narrow_clip, the slimmed post-arrival path of the persistent solver), then A/B of every switch on the driver-flag bench, a kernel timeline
# and the per-visit stamps of the persistent solver (hop budget)
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
cd oracle && make >/dev/null 2>&1; cd ..
T0=$(date +%s)
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r5a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5a_pytest.log
grep -E "passed|failed|rc=|Error|error" gpurun_out/r5a_pytest.log | tail -6
echo "suite took $(( $(date +%s) - T0 )) s"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest"
run() {  # name, env...
  local name=$1; shift
  ( for kv in "$@"; do export "$kv"; done; timeout 300 $B 2>gpurun_out/r5a_$name.err | tail -1 > gpurun_out/r5a_$name.json )
  python - "$name" <<'PY'
import sys, json
n = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r5a_{n}.json"))
    print(f"{n:16s} {d['value']:8.1f} steps/s  {d['ms_per_step']:.4f} ms  dev {d['device_ms_per_step']:.4f}  solver {d['roofline']['avg_launch_us']:.1f} us  contacts {d['config']['contacts']}  reruns {d['step_modes_timed']['synchronous_reruns']}  " + " ".join(f"{k[:6]}={v:.3f}" for k, v in d['stage_ms'].items()))
except Exception as e:
    print(n, "FAILED", e)
PY
}
run default
run nodiet MI_PHYSICS_LIB=$PWD/build_exp/libmi_physics_nodiet.so
run noreset MI_FUSE_RESET=0
run noround0 MI_ROUND0_EMIT=0
run nolarge MI_FUSE_LARGE=0
run noguest MI_FORCES_GUEST=0
run alloff MI_FUSE_RESET=0 MI_ROUND0_EMIT=0 MI_FUSE_LARGE=0 MI_FORCES_GUEST=0 MI_PHYSICS_LIB=$PWD/build_exp/libmi_physics_nodiet.so
run default2
echo "benches done at $(( $(date +%s) - T0 )) s"
bash tools/gpu_timeline.sh 2>&1 | tail -1; cp gpurun_out/timeline.txt gpurun_out/r5a_step_timeline.txt
bash tools/gpu_timeline2.sh > gpurun_out/r5a_hop_stamps.txt 2>&1; tail -12 gpurun_out/r5a_hop_stamps.txt
echo "all done at $(( $(date +%s) - T0 )) s"
