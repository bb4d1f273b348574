#!/bin/bash
# round 4 visit 17: what stands on either side of the ~26 ms idle gap per pipelined step
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/tp17
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tp17 -o t -- python $R/bench.py --steps 8 --warmup 2 --only-timed-steps > $R/gpurun_out/tp17/run.log 2>&1
cd $R
f=$(find gpurun_out/tp17 -name "*kernel_trace.csv" | head -1)
python tools/trace_timeline.py $f 0.5 0.97 | tee gpurun_out/r04_v17_timeline.txt
find gpurun_out/tp17 -name "*.csv" -size +1M -delete
