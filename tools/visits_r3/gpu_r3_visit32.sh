#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v32
export PYTHONUNBUFFERED=1
timeout -k 5 300 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "single_sequence" > gpurun_out/v32/t.log 2>&1; echo "b1 tests rc=$?"; grep -E "passed|failed|^E  " gpurun_out/v32/t.log | cut -c1-300 | tail -3
for rep in 1 2 3; do for v in A B; do
  if [ $v = B ]; then export MD_HIP_LIB=$R/moondream_amd/libmoondream_hip_ab.so; else unset MD_HIP_LIB; fi
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --no-fp8-full-leg --latency-runs 9 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$v (A = non-temporal weight rows) rep $rep: p50 caption latency %.2f ms' % d['p50_caption_latency_ms'])"
done; done 2>&1 | tee gpurun_out/v32/ab.txt
