#!/bin/bash
# usage (on the GPU box): tools/ab_lib.sh [reps]  -- interleaved bench of the in-tree library (A) and MD_HIP_LIB=libmoondream_hip_ab.so (B)
R=$GRAFT_REPO_ROOT; cd $R
REPS=${1:-3}
for i in $(seq $REPS); do
  for v in A B; do
    if [ $v = B ]; then export MD_HIP_LIB=$R/moondream_amd/libmoondream_hip_ab.so; else unset MD_HIP_LIB; fi
    timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --latency-runs 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('$v rep $i: %.1f images/s  ms/step %.1f  tile-GEMM %.0f TF/s  vision %.1f prefill %.1f decode %.1f' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['phase_ms']['vision'], d['phase_ms']['image_prefill'], d['phase_ms']['decode']))"
  done
done
