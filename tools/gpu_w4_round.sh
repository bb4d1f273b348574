#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -6
timeout 400 python tools/sweep_gemm.py tiles=20,11 rounds=3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sweep_w4.txt
