#!/bin/bash
# round 4 visit 1: the LDS-DMA operand path of the four-wave GEMM, first contact: bit-equality with the register-staged loop + speed
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python tools/sweep_w4_variants.py variants=0,1,2,3 rounds=3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_v01_sweep.txt
MD_W4_VARIANT=1 timeout 500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -6 | tee gpurun_out/r04_v01_tests.txt
