#!/bin/bash
# First GPU visit of a round, bounded (<= ~6 min of box time): all GPU tests, the default bench line, one traced bench run
# with kernel stats, the two PMC traffic passes -- every profiler run is killed at once if it faults at start-up (one box
# in round 2 did: profiles/r02_trace_steps.txt, last note) instead of sitting out its timeout.
#   gpurun --timeout 420 -- 'bash tools/gpu_round_start.sh'
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/rs
export PYTHONUNBUFFERED=1
timeout -k 5 90 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/rs/tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 gpurun_out/rs/tests.log
timeout -k 5 100 python bench.py > gpurun_out/rs/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/rs/bench.log | cut -c1-300
guarded() {  # guarded <seconds> <log> <command...>: run, kill on a start-up fault
  local t=$1 log=$2; shift 2
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 $t "$@" > $log 2>&1 ) &
  local pid=$!
  while kill -0 $pid 2>/dev/null; do
    if grep -q "Memory access fault" $log 2>/dev/null; then echo "  faulted at start-up: killed ($log)"; pkill -9 -P $pid; kill -9 $pid; break; fi
    sleep 2
  done
  wait $pid 2>/dev/null
}
B="python $R/bench.py --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --latency-runs 0"
guarded 120 $R/gpurun_out/rs/trace.log rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rs/trace -o bench -- $B --steps 3 --warmup 1
f=$(find gpurun_out/rs/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/rs/kernel_stats.csv && head -8 "$f" | cut -c1-140
for c in FETCH_SIZE WRITE_SIZE; do
  guarded 150 $R/gpurun_out/rs/$c.log rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/rs/$c -o r1 -- $B --steps 1 --warmup 0 --tokens 1 --batch 64 --no-graphs --no-pipeline --only-timed-steps
done
find gpurun_out/rs -name "*kernel_trace.csv" -size +8M -delete
find gpurun_out/rs -name "*counter_collection.csv" | head -3
