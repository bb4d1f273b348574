#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v26
export PYTHONUNBUFFERED=1
timeout -k 5 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm" > gpurun_out/v26/t.log 2>&1; echo "gemm tests rc=$?"; grep -E "passed|failed|^E  " gpurun_out/v26/t.log | cut -c1-300 | tail -4
LEGS="--no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --no-fp8-full-leg --latency-runs 0"
for rep in 1 2 3; do for v in 1 0; do
  MD_GEMM_TAIL_SPLIT=$v timeout -k 5 300 python bench.py $LEGS --steps 6 --warmup 2 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('tail_split=$v rep $rep: %.1f images/s  ms/step %.1f | vision %.2f ms  vit frac %.3f | tile GEMM %.0f TF/s (launches %d)' % (d['value'], d['ms_per_step'], d['phase_ms']['vision'], d['vit_encoder']['frac'], d['roofline']['achieved'], d['roofline']['launches']))"
done; done 2>&1 | tee gpurun_out/v26/ab.txt
