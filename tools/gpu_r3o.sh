#!/bin/bash
# round 3, call O: kernel timelines of the middle tile of 1 and 8 x-slabs (where does the per-rank cost of the replicated scene go?)
bash tools/gpu_tl_weak.sh 1
bash tools/gpu_tl_weak.sh 8
