#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=line -p no:cacheprovider > gpurun_out/tests_k.log 2>&1
grep -E "passed|failed" gpurun_out/tests_k.log | tail -2
timeout 300 python tools/sweep_decode.py 2>&1 | grep -v amdgpu.ids > gpurun_out/sweep_decode.log
cat gpurun_out/sweep_decode.log
timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep -v amdgpu.ids > gpurun_out/kernel_bench.log
tail -18 gpurun_out/kernel_bench.log
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=line -p no:cacheprovider > gpurun_out/tests_m.log 2>&1
grep -E "passed|failed" gpurun_out/tests_m.log | tail -2
timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log | cut -c1-1600
