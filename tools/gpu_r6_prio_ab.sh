#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
L="--no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --no-fp8-full-leg --no-strict-leg --no-second-oracle --latency-runs 0 --only-timed-steps"
for rep in 1 2; do
  for prio in -1 0; do
    echo "decode stream priority $prio: $(MD_PIPE_DECODE_PRIORITY=$prio python bench.py $L --steps 12 --warmup 4 2>/dev/null | tail -1 | cut -c1-200)"
  done
done
