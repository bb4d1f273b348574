#!/bin/bash
# Second on-GPU pass: full parity suite incl. full-size models, B=64 bench, rocprof, microbench.
mkdir -p gpurun_out/prof
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/tests.log 2>&1
echo "tests exit $?" >> gpurun_out/tests.log
tail -4 gpurun_out/tests.log
timeout 600 python tools/kernel_bench.py > gpurun_out/kernel_bench.log 2>&1
tail -32 gpurun_out/kernel_bench.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
tail -2 gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --latency-runs 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof | head -20
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
find gpurun_out/prof -name "*kernel_trace.csv" -size +30M -delete
