#!/bin/bash
# Full validation visit: all GPU tests, the bench (JSON line kept), rocprof kernel stats of the same command.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests_all.log 2>&1
echo "gpu tests rc=$?"; grep -E "passed|failed" gpurun_out/tests_all.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/tests_all.log | head -20
timeout 900 python bench.py --steps 6 --warmup 2 > gpurun_out/bench_full.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/bench_full.log | cut -c1-2500
