#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -m gpu -q --tb=line -p no:cacheprovider > gpurun_out/tests.log 2>&1
grep -E "passed|failed|^/.*Error|^E " gpurun_out/tests.log | tail -8
timeout 300 python tools/kernel_bench.py gemm attn 2>&1 | grep -v amdgpu.ids > gpurun_out/kernel_bench.log
grep -E "m= 2|m= 4|attn" gpurun_out/kernel_bench.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log | cut -c1-1500
