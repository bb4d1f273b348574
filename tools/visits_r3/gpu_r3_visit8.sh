#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v8
export PYTHONUNBUFFERED=1
timeout -k 5 300 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/v8/tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed" gpurun_out/v8/tests.log | tail -2; grep -E "^FAILED|^ERROR|fp8 full|^E  " gpurun_out/v8/tests.log | cut -c1-400 | head -20
timeout -k 5 300 python bench.py --no-cpu-baseline --no-vqa-leg --no-dedup-leg > gpurun_out/v8/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/v8/bench.log > gpurun_out/v8/bench.json; tail -3 gpurun_out/v8/bench.log | head -2 | cut -c1-300; python - <<'P'
import json
d=json.load(open('gpurun_out/v8/bench.json'))
for k in ('value','ms_per_step','phase_ms','fp8_decode','fp8_full','parity_exact','parity_ok'):
    print(k, json.dumps(d.get(k))[:1200])
print('roofline', d['roofline']['achieved'], d['roofline']['frac'])
P
