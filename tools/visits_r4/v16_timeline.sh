#!/bin/bash
# round 4 visit 16: why did the timed step not follow the tile GEMM's gain?  bench variants (timed steps only) + a timeline reading of a pipelined trace
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/tp16
export PYTHONUNBUFFERED=1
for v in "" "--no-pipeline" "--no-pipeline --no-graphs" "--no-graphs"; do
  echo "== bench.py --steps 6 --warmup 2 --only-timed-steps $v"
  timeout 200 python bench.py --steps 6 --warmup 2 --only-timed-steps $v 2>&1 | grep '^{' | tail -1
done | tee gpurun_out/r04_v16_variants.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tp16 -o t -- python $R/bench.py --steps 8 --warmup 2 --only-timed-steps > $R/gpurun_out/tp16/run.log 2>&1
cd $R
f=$(find gpurun_out/tp16 -name "*kernel_trace.csv" | head -1)
python tools/trace_timeline.py $f 0.5 0.97 | tee gpurun_out/r04_v16_timeline.txt
tail -1 gpurun_out/tp16/run.log | cut -c1-200
find gpurun_out/tp16 -name "*.csv" -size +1M -delete
