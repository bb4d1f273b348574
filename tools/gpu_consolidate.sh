#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/cons
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --tb=line -p no:cacheprovider > gpurun_out/cons/tests.log 2>&1
tail -2 gpurun_out/cons/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/cons/smoke.log 2>&1; tail -2 gpurun_out/cons/smoke.log
timeout 900 python bench.py > gpurun_out/cons/bench.json 2>gpurun_out/cons/bench.err
tail -1 gpurun_out/cons/bench.json | cut -c1-2600
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cons/prof -o r1 -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --latency-runs 0 > $R/gpurun_out/cons/prof.log 2>&1
cd $R
find gpurun_out/cons -name "*kernel_trace.csv" -delete
head -14 $(find gpurun_out/cons/prof -name "*kernel_stats.csv" | head -1) | cut -c1-150
