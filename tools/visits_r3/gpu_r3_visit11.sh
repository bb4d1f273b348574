#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v11
export PYTHONUNBUFFERED=1
timeout -k 5 120 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "rope_kv_write_in_gemm" > gpurun_out/v11/t1.log 2>&1; echo "rope fuse test rc=$?"; grep -E "passed|failed|^E  " gpurun_out/v11/t1.log | cut -c1-300 | tail -6
timeout -k 5 360 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/v11/tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed" gpurun_out/v11/tests.log | tail -2; grep -E "^FAILED|^ERROR|fp8 full|^E  " gpurun_out/v11/tests.log | cut -c1-300 | head -20
for f in 1 0; do MD_ROPE_FUSE=$f timeout -k 5 200 python bench.py --no-cpu-baseline --no-vqa-leg --no-dedup-leg --no-fp8-leg --no-fp8-full-leg --no-detect13-leg --latency-runs 0 --steps 4 --warmup 2 > gpurun_out/v11/bench_f$f.log 2>&1; grep '^{"metric"' gpurun_out/v11/bench_f$f.log | tail -1 > gpurun_out/v11/bench_f$f.json; python - <<P 2>&1 | tail -3
import json
d=json.load(open('gpurun_out/v11/bench_f$f.json'))
print('rope_fuse=$f', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],1), d['phase_ms'], 'gemm', round(d['roofline']['achieved'],1), 'parity', d.get('parity_exact'), d.get('parity_ok'))
P
done
