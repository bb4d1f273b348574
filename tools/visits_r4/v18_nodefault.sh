#!/bin/bash
# round 4 visit 18: ids to pinned memory on the decode stream, no pageable H2D in the launch path, nothing on the default stream:
# does the ~26 ms idle gap per pipelined step go away?  model tests first, then timed steps + timeline, then the default bench line
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/tp18
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3
for v in "" "--no-pipeline"; do
  echo "== bench.py --steps 6 --warmup 2 --only-timed-steps $v"
  timeout 200 python bench.py --steps 6 --warmup 2 --only-timed-steps $v 2>&1 | grep '^{' | tail -1
done | tee gpurun_out/r04_v18_variants.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tp18 -o t -- python $R/bench.py --steps 8 --warmup 2 --only-timed-steps > $R/gpurun_out/tp18/run.log 2>&1
cd $R
f=$(find gpurun_out/tp18 -name "*kernel_trace.csv" | head -1)
python tools/trace_timeline.py $f 0.5 0.97 | tee gpurun_out/r04_v18_timeline.txt
find gpurun_out/tp18 -name "*.csv" -size +1M -delete
timeout 400 python bench.py > gpurun_out/r04_v18_bench.log 2>&1
echo "bench rc=$?"; grep '^{' gpurun_out/r04_v18_bench.log | tail -1 > gpurun_out/r04_v18_bench.json; cut -c1-300 gpurun_out/r04_v18_bench.json
grep "cpu_baseline" gpurun_out/r04_v18_bench.log | tail -12
