# dev helper: lane-per-pair (MI_GJK_WAVE=0) against wave-per-pair (1) GJK / EPA on all-shape zoos of three sizes
for v in 0 1; do for a in "16,6,16" "40,10,40" "64,12,64"; do echo "== MI_GJK_WAVE=$v shape_zoo $a"; MI_GJK_WAVE=$v SCENE=shape_zoo SCENE_ARGS=$a WARM=200 STEPS=60 TAIL=1 bash tools/gpu_graph_dbg.sh 2>&1 | sed "s/.*ms\/step/ms\/step/" | cut -c1-18; done; done
