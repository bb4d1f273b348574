#!/bin/bash
# Round 6 validation visit: all GPU tests, the drop-in on the reference's own class (scratch checkout, see tools/gpu_r6_dropin.sh),
# the default bench line at 20 steps, and the bench once more WITH the reference present so that cpu_baseline is the reference itself.
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/val6; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout -k 5 400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/tests.log 2>&1; echo "gpu tests rc=$?"; tail -15 $O/tests.log
if [ -d $R/_ref_scratch ]; then
  export MOONDREAM_REFERENCE=$R/_ref_scratch
  timeout -k 5 400 python -m pytest tests/test_dropin_gpu.py -m gpu -q --tb=short -rs -p no:cacheprovider > $O/dropin_tests.log 2>&1; echo "dropin rc=$?"; tail -5 $O/dropin_tests.log
  timeout -k 5 300 python tools/dropin_report.py > $O/dropin_report.json 2> $O/dropin_report.err; tail -c 600 $O/dropin_report.json
  unset MOONDREAM_REFERENCE
fi
timeout -k 5 600 python bench.py --steps 20 --warmup 5 > $O/bench_default_20steps.log 2>&1; echo "bench rc=$?"; grep '^{"metric"' $O/bench_default_20steps.log | tail -1 > $O/bench_default_20steps.json; cut -c1-400 $O/bench_default_20steps.json
if [ -d $R/_ref_scratch ]; then
  MOONDREAM_REFERENCE=$R/_ref_scratch timeout -k 5 400 python bench.py --steps 5 --warmup 2 --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --no-fp8-full-leg --no-strict-leg --latency-runs 0 > $O/bench_with_reference.log 2>&1
  grep '^{"metric"' $O/bench_with_reference.log | tail -1 > $O/bench_with_reference.json; python -c "
import json; r=json.load(open('$O/bench_with_reference.json')); print('cpu_baseline:', json.dumps(r.get('cpu_baseline'))[:700])"
fi
