#!/bin/bash
# round 6, last session: same-box A/B of the terrain configuration (tree against the commit before the two-instance terrain kernel, built under build_exp/head_src), then the
# differential fuzzer at HEAD: worlds on terrain only (small and 20 x), then fresh seed ranges of everything
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
cd oracle && make >/dev/null 2>&1; cd ..
T0=$(date +%s)
R=$PWD
for i in 1 2 3; do
  for t in . build_exp/head_src; do
    (cd $t && CFGS=terrain bash $R/tools/gpu_cfgs.sh 2>&1 | tail -1 | python -c "import sys,json; l=sys.stdin.read(); d=json.loads(l[l.index('{'):]); print('$t', round(d['ms_per_step'],4), d['stage_ms'])")
  done
done > gpurun_out/e_terrain_ab.txt 2>&1
cat gpurun_out/e_terrain_ab.txt | cut -c1-250
echo "A/B at $(( $(date +%s) - T0 )) s"
timeout 200 python tools/gpu_fuzz.py --seeds 30000:34000 --only-terrain --budget 150 --out gpurun_out/e_fuzz_terrain_small.json 2>&1 | tail -3 | cut -c1-400
timeout 260 python tools/gpu_fuzz.py --seeds 3000:3600 --only-terrain --scale 20 --steps 25 --budget 200 --out gpurun_out/e_fuzz_terrain_large.json 2>&1 | tail -3 | cut -c1-400
timeout 200 python tools/gpu_fuzz.py --seeds 40000:44000 --budget 150 --out gpurun_out/e_fuzz_small.json 2>&1 | tail -3 | cut -c1-400
timeout 160 python tools/gpu_fuzz_sharded.py --seeds 9000:9400 --budget 110 --out gpurun_out/e_fuzz_sharded.json 2>&1 | tail -3 | cut -c1-400
echo "all done at $(( $(date +%s) - T0 )) s"
