#!/bin/bash
# round 4 visit 34 (after the int4 weight stream and ABI 4), two-stream engine with pinned id collection, default-mode batch invariance, per-decision parity): every GPU test, the default bench line as the driver runs it, smoke, then one traced bench run.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/prof34
export PYTHONUNBUFFERED=1
t0=$(date +%s)
timeout -k 5 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/r04_v34_tests.log 2>&1
echo "gpu tests rc=$? wall=$(( $(date +%s) - t0 ))s"; grep -E "passed|failed" gpurun_out/r04_v34_tests.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/r04_v34_tests.log | head -20
grep -iE "second oracle|mutation|bench64 parity|batch_equals" gpurun_out/r04_v34_tests.log | cut -c1-300 | head -20
t0=$(date +%s)
timeout -k 5 400 python bench.py > gpurun_out/r04_v34_bench.log 2>&1
echo "bench rc=$? wall=$(( $(date +%s) - t0 ))s"; grep '^{' gpurun_out/r04_v34_bench.log | tail -1 > gpurun_out/r04_v34_bench.json; cut -c1-1500 gpurun_out/r04_v34_bench.json
timeout -k 5 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof34 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-fp8-full-leg --no-detect13-leg --no-second-oracle --no-strict-leg --latency-runs 0 > $R/gpurun_out/prof34/bench.log 2>&1
echo "kernel-trace rc=$?"
cd $R
grep '^{' gpurun_out/prof34/bench.log | tail -1 | cut -c1-600
f=$(find gpurun_out/prof34 -name "*kernel_stats.csv" | head -1); echo "stats: $f"; [ -n "$f" ] && cp "$f" gpurun_out/r04_v34_kernel_stats.csv && head -14 "$f" | cut -c1-170
find gpurun_out/prof34 -name "*kernel_trace.csv" -size +8M -delete
