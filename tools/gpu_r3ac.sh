#!/bin/bash
SCENE=terrain_big WARM=240 bash tools/gpu_tl_scene.sh > gpurun_out/tl_terrain.log 2>&1; tail -1 gpurun_out/tl_terrain.log; grep -n "hm_\|narrow\|emit" gpurun_out/timeline_scene.txt | cut -c1-110
