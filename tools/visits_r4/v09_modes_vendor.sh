#!/bin/bash
# round 4 visit 9: placements of the LDS-DMA pieces in the 16x16x32 loop (bias epilogue, variant 0 = shipped mode 1) + the vendor library on the same box
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python tools/sweep_w4_variants.py variants=0,2,3,4 rounds=3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_v09_modes.txt
timeout 300 python tools/vendor_gemm_compare.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_v09_vendor.txt
