#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v7
export PYTHONUNBUFFERED=1
timeout -k 5 200 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -s -k "f8" > gpurun_out/v7/tests_f8.log 2>&1; echo "f8 kernel tests rc=$?"
grep -E "passed|failed" gpurun_out/v7/tests_f8.log | tail -2; grep -E "^FAILED|^ERROR|^E  |rel-rms" gpurun_out/v7/tests_f8.log | cut -c1-260 | head -40
timeout -k 5 240 python tools/sweep_gemm_f8.py rounds=2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/v7/sweep_f8.txt
