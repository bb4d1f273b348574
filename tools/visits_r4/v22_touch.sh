#!/bin/bash
# round 4 visit 22: L2 touches two pairs ahead of the operand pieces (MODE 5: activation rows, 6: + weight rows) against the shipped placement 2
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
( timeout 400 python tools/sweep_w4_variants.py variants=0,5,6 rounds=3 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_v22_touch_sweep.txt
