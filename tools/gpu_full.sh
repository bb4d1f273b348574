#!/bin/bash
# Full validation visit: all GPU tests, then the bench exactly as the driver runs it (defaults; JSON line kept).
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 5 80 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/tests_all.log 2>&1
echo "gpu tests rc=$?"; grep -E "passed|failed" gpurun_out/tests_all.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/tests_all.log | head -20
grep -E "fp8 decode|bench64 parity" gpurun_out/tests_all.log | head
t0=$(date +%s)
timeout -k 5 100 python bench.py > gpurun_out/bench_full.log 2>&1
echo "bench rc=$? wall=$(( $(date +%s) - t0 ))s"; tail -1 gpurun_out/bench_full.log | cut -c1-4000
timeout -k 5 40 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
