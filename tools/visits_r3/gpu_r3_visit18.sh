#!/bin/bash
# decode attention with non-temporal K / V loads: kernel tests, then interleaved A/B in the bench (MD_ATTN_DECODE_NT=0|1)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v19
export PYTHONUNBUFFERED=1

LEGS="--no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --latency-runs 0"
for rep in 1 2; do for nt in 1 0; do
  MD_DECODE_NT=$nt timeout -k 5 300 python bench.py $LEGS --steps 6 --warmup 2 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
f=d.get('fp8_full') or {}
print('decode-GEMM weights nt=$nt rep $rep: %.1f images/s  ms/step %.1f | decode phase %.2f ms  decode_step.frac %.3f | fp8_full %.1f images/s decode %.2f ms' % (d['value'], d['ms_per_step'], d['phase_ms']['decode'], d['decode_step']['frac'], f.get('images_per_sec',0), (f.get('phase_ms') or {}).get('decode',0)))"
done; done 2>&1 | tee gpurun_out/v19/ab.txt
