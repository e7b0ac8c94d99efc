#!/bin/bash
# timelines of the other configs (where does a step of cfg4 / cfg5 / terrain / cfg2 go?)
for sc in ragdolls vehicles terrain_big mixed_stack; do
  SCENE=$sc WARM=240 bash tools/gpu_tl_scene.sh > gpurun_out/tl_$sc.log 2>&1; cp gpurun_out/timeline_scene.txt gpurun_out/timeline_$sc.txt; tail -1 gpurun_out/tl_$sc.log
done
