#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v10
export PYTHONUNBUFFERED=1
timeout -k 5 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm" > gpurun_out/v10/tests_gemm.log 2>&1; echo "gemm kernel tests rc=$?"
grep -E "passed|failed" gpurun_out/v10/tests_gemm.log | tail -2; grep -E "^FAILED|^ERROR|^E  " gpurun_out/v10/tests_gemm.log | cut -c1-260 | head -20
timeout -k 5 300 python tools/sweep_gemm.py tiles=20,15 rounds=3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/v10/sweep_gemm.txt
