#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "fp8" 2>&1 | tail -8
timeout 300 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "fp8_decode_mode_tiny" -s 2>&1 | grep -E "fp8 decode|passed|failed|Error|error|assert" | head -20
timeout 300 python tools/sweep_fp8.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sweep_fp8.txt
