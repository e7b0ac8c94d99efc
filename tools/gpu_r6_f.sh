#!/bin/bash
# round 6, last session: the large-window terrain instance as a flag-scanning launch — the GPU suite + smoke, the terrain configuration against the tree of two commits
# before (build_exp/head_src) on the same box, terrain worlds of the fuzzer, rocprofv3 kernel stats of the terrain scene, the bench at the driver's flags
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
cd oracle && make >/dev/null 2>&1; cd ..
T0=$(date +%s)
R=$PWD
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/f_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/f_pytest.log
grep -E "^FAILED|passed|failed|rc=" gpurun_out/f_pytest.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
echo "suite + smoke at $(( $(date +%s) - T0 )) s"
for i in 1 2 3; do
  for t in . build_exp/head_src; do
    (cd $t && CFGS=terrain bash $R/tools/gpu_cfgs.sh 2>&1 | tail -1 | python -c "import sys,json; l=sys.stdin.read(); d=json.loads(l[l.index('{'):]); print('$t', round(d['ms_per_step'],4), d['stage_ms'])")
  done
done > gpurun_out/f_terrain_ab.txt 2>&1
cat gpurun_out/f_terrain_ab.txt | cut -c1-250
CFGS=terrain bash tools/gpu_cfgs.sh > /dev/null 2>&1; cp gpurun_out/cfgs.json gpurun_out/f_terrain_cfg.json
timeout 150 python tools/gpu_fuzz.py --seeds 3600:4200 --only-terrain --scale 20 --steps 25 --budget 90 --out gpurun_out/f_fuzz_terrain_large.json 2>&1 | tail -1 | cut -c1-300
timeout 100 python tools/gpu_fuzz.py --seeds 34000:36000 --only-terrain --budget 60 --out gpurun_out/f_fuzz_terrain_small.json 2>&1 | tail -1 | cut -c1-300
SCENE=terrain_big STEPS=60 WARM=300 bash tools/gpu_prof_scene.sh 2>&1 | head -16; cp gpurun_out/scene_kernels.txt gpurun_out/f_terrain_kernels.txt
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/f_bench_driver_flags.json; cut -c1-200 gpurun_out/f_bench_driver_flags.json
echo "all done at $(( $(date +%s) - T0 )) s"
