#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v24
export PYTHONUNBUFFERED=1
timeout -k 5 300 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "prefetched or mixed_encoded or batched_equals" > gpurun_out/v24/t.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^E  " gpurun_out/v24/t.log | cut -c1-300 | tail -4
timeout -k 5 400 python bench.py --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-fp8-full-leg --latency-runs 0 --steps 2 --warmup 1 > gpurun_out/v24/bench.log 2>&1
grep '^{"metric"' gpurun_out/v24/bench.log | tail -1 > gpurun_out/v24/bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/v24/bench.json"))
print("value %.1f"%d["value"])
for k in ("detect13","detect13_fp8"):
    x=d.get(k)
    if x: print(k, {kk: x[kk] for kk in ("images_per_sec","ms_per_step","images_per_sec_tiling_prefetched","ms_per_step_tiling_prefetched","phase_ms") if kk in x}, (x.get("parity") or {}).get("ok"))
PY
tail -3 gpurun_out/v24/bench.log | cut -c1-300
