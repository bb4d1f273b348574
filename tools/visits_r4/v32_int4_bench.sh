#!/bin/bash
# round 4 visit 32: bench.py with the int4-stream leg (other auxiliary legs off), then the int4 / fp8 tests once more
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python bench.py --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-fp8-full-leg --no-dedup-leg --no-detect13-leg --no-second-oracle --no-strict-leg --latency-runs 2 > gpurun_out/r04_v32_bench.log 2>&1
echo "bench rc=$?"; grep '^{' gpurun_out/r04_v32_bench.log | tail -1 > gpurun_out/r04_v32_bench.json
python - <<'PY'
import json
r=json.load(open('gpurun_out/r04_v32_bench.json'))
print(round(r['value'],1), r['phase_ms'])
print(json.dumps(r.get('int4_decode'), indent=1)[:2500])
PY
tail -5 gpurun_out/r04_v32_bench.log | cut -c1-300
