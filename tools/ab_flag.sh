#!/bin/bash
# usage: tools/ab_flag.sh "<flags A>" "<flags B>" [reps]  -- interleaved A/B of bench.py command-line variants on one box
A=$1; B=$2; R=${3:-2}
for i in $(seq $R); do
  for x in "$A" "$B"; do
    timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --latency-runs 0 $x 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[$x]', round(d['value'],1), 'img/s', d['phase_ms'], 'gemm', round(d['roofline']['achieved']))
"
  done
done
