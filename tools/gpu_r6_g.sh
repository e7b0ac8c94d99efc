#!/bin/bash
# round 6, last session: the terrain count pass at 4 / 5 waves per SIMD (variant builds, spilling) against the tree, same box; then the GPU suite under the variant switches at HEAD
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
cd oracle && make >/dev/null 2>&1; cd ..
T0=$(date +%s)
for i in 1 2 3; do
  for v in "" build_exp/libmi_physics_hm4.so build_exp/libmi_physics_hm5.so; do
    MI_PHYSICS_LIB=$v CFGS=terrain bash tools/gpu_cfgs.sh 2>&1 | tail -1 | python -c "import sys,json; l=sys.stdin.read(); d=json.loads(l[l.index('{'):]); print('${v:-tree}', round(d['ms_per_step'],4), d['stage_ms'])"
  done
done > gpurun_out/g_terrain_wpe.txt 2>&1
cat gpurun_out/g_terrain_wpe.txt | cut -c1-200
echo "A/B at $(( $(date +%s) - T0 )) s"
bash tools/gpu_suite_variants.sh > gpurun_out/g_variants.txt 2>&1; cat gpurun_out/g_variants.txt
echo "all done at $(( $(date +%s) - T0 )) s"
