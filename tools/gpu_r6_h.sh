#!/bin/bash
# round 6, last session: the lowest-point test as a kernel of its own (k_hm_lowest) — terrain scene against the build before it (build_exp/libmi_physics_head.so), same box;
# the GPU suite + smoke; terrain worlds of the fuzzer; the terrain scene's kernel stats; the bench at the driver's flags
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
cd oracle && make >/dev/null 2>&1; cd ..
T0=$(date +%s)
for i in 1 2 3; do
  for v in "" build_exp/libmi_physics_head.so; do
    MI_PHYSICS_LIB=$v CFGS=terrain bash tools/gpu_cfgs.sh 2>&1 | tail -1 | python -c "import sys,json; l=sys.stdin.read(); d=json.loads(l[l.index('{'):]); print('${v:-tree}', round(d['ms_per_step'],4), d['stage_ms'])"
  done
done > gpurun_out/h_terrain_ab.txt 2>&1
cat gpurun_out/h_terrain_ab.txt | cut -c1-200
CFGS=terrain bash tools/gpu_cfgs.sh > /dev/null 2>&1; cp gpurun_out/cfgs.json gpurun_out/h_terrain_cfg.json
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/h_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/h_pytest.log
grep -E "^FAILED|passed|failed|rc=" gpurun_out/h_pytest.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
echo "suite + smoke at $(( $(date +%s) - T0 )) s"
timeout 100 python tools/gpu_fuzz.py --seeds 36000:38000 --only-terrain --budget 50 --out gpurun_out/h_fuzz_terrain_small.json 2>&1 | tail -1 | cut -c1-300
timeout 100 python tools/gpu_fuzz.py --seeds 4200:4800 --only-terrain --scale 20 --steps 25 --budget 50 --out gpurun_out/h_fuzz_terrain_large.json 2>&1 | tail -1 | cut -c1-300
SCENE=terrain_big STEPS=60 WARM=300 bash tools/gpu_prof_scene.sh 2>&1 | head -8; cp gpurun_out/scene_kernels.txt gpurun_out/h_terrain_kernels.txt
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/h_bench_driver_flags.json; cut -c1-200 gpurun_out/h_bench_driver_flags.json
echo "all done at $(( $(date +%s) - T0 )) s"
