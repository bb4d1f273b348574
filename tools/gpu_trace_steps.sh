#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ts
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ts -o ts -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --latency-runs 0 > $R/gpurun_out/ts/bench.log 2>&1
echo "rc=$?"; tail -1 $R/gpurun_out/ts/bench.log | cut -c1-1500
cd $R
f=$(find gpurun_out/ts -name "*kernel_trace.csv" | head -1)
python tools/trace_steps.py $f | tee gpurun_out/ts/steps.txt
find gpurun_out/ts -name "*kernel_trace.csv" -size +8M -delete
