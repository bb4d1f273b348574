#!/bin/bash
# round 4 visit 33: 256 x 192 tiles for the N = 1152 layers: GEMM kernel tests (auto selection), then bitwise + timing against 256 x 256
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -6 | tee gpurun_out/r04_v33_tests.txt
timeout 600 python tools/sweep_w4_nj.py rounds=5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_v33_nj.txt
