#!/bin/bash
# rocprofv3 evidence for profiles/: per-kernel time of the bench command, then the PMC traffic passes.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --latency-runs 0 > $R/gpurun_out/prof/bench.log 2>&1
echo "kernel-trace rc=$?"; tail -1 $R/gpurun_out/prof/bench.log | cut -c1-300
cd $R
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); echo "stats: $f"; [ -n "$f" ] && cp "$f" gpurun_out/bench_kernel_stats.csv && head -25 "$f" | cut -c1-200
find gpurun_out/prof -name "*kernel_trace.csv" -size +8M -delete
bash tools/gpu_pmc_traffic.sh
