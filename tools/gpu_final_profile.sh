#!/bin/bash
# Short, bounded visit: the model tests, then ONE traced bench run (rocprofv3 --kernel-trace --stats) whose JSON line and
# kernel summary go to gpurun_out/.  A profiler run that faults at start-up is killed at once instead of at its timeout.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/prof4
timeout -k 5 90 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout -k 5 140 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof4 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --latency-runs 0 > $R/gpurun_out/prof4/bench.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do
  if grep -q "Memory access fault" $R/gpurun_out/prof4/bench.log 2>/dev/null; then echo "profiler run faulted at start-up: killed"; pkill -9 -P $pid; kill -9 $pid; break; fi
  sleep 2
done
wait $pid 2>/dev/null; echo "kernel-trace rc=$?"
cd $R
grep '^{' gpurun_out/prof4/bench.log | tail -1 | cut -c1-400
f=$(find gpurun_out/prof4 -name "*kernel_stats.csv" | head -1); echo "stats: $f"; [ -n "$f" ] && cp "$f" gpurun_out/bench_kernel_stats_v4.csv && head -12 "$f" | cut -c1-160
find gpurun_out/prof4 -name "*kernel_trace.csv" -size +8M -delete
