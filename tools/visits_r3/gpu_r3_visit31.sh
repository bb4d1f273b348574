#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v31
export PYTHONUNBUFFERED=1
timeout -k 5 400 python -m pytest tests/test_kernels_gpu.py tests/test_abi.py -m gpu -q --tb=short -p no:cacheprovider -x -k "attention or f8 or e4m3 or abi" > gpurun_out/v31/t.log 2>&1; echo "kernel tests rc=$?"; grep -E "passed|failed|^E  " gpurun_out/v31/t.log | cut -c1-300 | tail -4
timeout -k 5 400 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -s -k "fp8 or rope" > gpurun_out/v31/t2.log 2>&1; echo "model fp8 tests rc=$?"; grep -E "passed|failed|^E  |fp8 full" gpurun_out/v31/t2.log | cut -c1-300 | tail -6
bash tools/ab_lib.sh 1 > /dev/null 2>&1
for rep in 1 2; do
  timeout -k 5 400 python bench.py --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --latency-runs 0 --steps 3 --warmup 1 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); f=d['fp8_full']
print('rep $rep: bf16 %.1f images/s | fp8_full %.1f images/s  ms/step %.1f  phases %s' % (d['value'], f['images_per_sec'], f['ms_per_step'], f['phase_ms']))"
done 2>&1 | tee gpurun_out/v31/fp8.txt
