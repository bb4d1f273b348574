#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v41
export PYTHONUNBUFFERED=1
timeout -k 5 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm" > gpurun_out/v41/t.log 2>&1; echo "gemm tests rc=$?"; grep -E "passed|failed|^E  " gpurun_out/v41/t.log | cut -c1-300 | tail -3
for rep in 1 2 3; do for v in new prev; do
  if [ $v = prev ]; then export MD_HIP_LIB=$R/moondream_amd/libmoondream_hip_ab.so; else unset MD_HIP_LIB; fi
  timeout -k 5 400 python bench.py --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --latency-runs 0 --steps 6 --warmup 2 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); f=d['fp8_full']
print('$v rep $rep: bf16 %.1f images/s  ms/step %.1f  tile GEMM %.0f TF/s  vision %.2f prefill %.2f | fp8_full %.1f images/s' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['phase_ms']['vision'], d['phase_ms']['image_prefill'], f['images_per_sec']))"
done; done 2>&1 | tee gpurun_out/v41/ab.txt
