#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests.log 2>&1
echo "tests exit $?" >> gpurun_out/tests.log
grep -E "passed|failed|FAILED|Error" gpurun_out/tests.log | tail -12
timeout 600 python tools/kernel_bench.py gemm decode > gpurun_out/kernel_bench.log 2>&1
tail -22 gpurun_out/kernel_bench.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
tail -2 gpurun_out/bench.log
