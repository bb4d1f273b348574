#!/bin/bash
# round 4 visit 20: default-mode batch invariance (four-wave GEMM pinned per call), one-stream engine, strict leg, pinned cpu_baseline:
# every GPU test, then the default bench line
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
t0=$(date +%s)
timeout -k 5 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/r04_v20_tests.log 2>&1
echo "gpu tests rc=$? wall=$(( $(date +%s) - t0 ))s"; grep -E "passed|failed" gpurun_out/r04_v20_tests.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/r04_v20_tests.log | head -20
grep -iE "second oracle|bench64 parity|batch vs sequential" gpurun_out/r04_v20_tests.log | cut -c1-400 | head -20
t0=$(date +%s)
timeout -k 5 400 python bench.py > gpurun_out/r04_v20_bench.log 2>&1
echo "bench rc=$? wall=$(( $(date +%s) - t0 ))s"; grep '^{' gpurun_out/r04_v20_bench.log | tail -1 > gpurun_out/r04_v20_bench.json
python - <<'PY'
import json
r=json.load(open('gpurun_out/r04_v20_bench.json'))
print(round(r['value'],1), round(r['ms_per_step'],1), r['phase_ms'], round(r['roofline']['achieved']), r['p50_caption_latency_ms'])
print(r.get('batch_equals_sequential_default_mode')); print({k:v for k,v in r.get('strict_batch_invariance',{}).items() if k!='note'})
print(r['cpu_baseline']['value'], r['cpu_baseline']['details'])
print(r['parity_exact'], r['parity_ok'], r.get('parity_second_oracle_exact'), r.get('parity_min_exact'))
PY
grep "cpu_baseline" gpurun_out/r04_v20_bench.log | tail -12
