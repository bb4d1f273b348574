#!/bin/bash
# round 4 visit 10: per-wave filler shift (variants 5, 6 = placements 1, 2 with wave w's fillers w gaps later) against 1 (= 0) and 2
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python tools/sweep_w4_variants.py variants=0,2,5,6 rounds=3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_v10_stagger.txt
