#!/bin/bash
# usage: tools/ab_bench.sh ENVVAR valA valB [reps]   -- interleaved A/B of the headline bench on one box
V=$1; A=$2; B=$3; R=${4:-2}
mkdir -p gpurun_out
for i in $(seq $R); do
  for x in $A $B; do
    env $V=$x timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --latency-runs 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$V=$x', round(d['value'],1), 'img/s', d['phase_ms'], 'gemm', round(d['roofline']['achieved']), 'decode_gemm', round(d['decode_gemm']['achieved']))
"
  done
done
