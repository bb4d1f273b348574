#!/bin/bash
# Round 6: what would lockstep decode over BOTH slot groups of the pipelined engine buy?  Upper bound = the same engine fed batches of 128
# (decode launches of 128 rows, encode launches twice as large) against batches of 64, same number of images, same box, interleaved.
R=$GRAFT_REPO_ROOT; cd $R
L="--no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --no-fp8-full-leg --no-strict-leg --no-second-oracle --latency-runs 0 --only-timed-steps"
for rep in 1 2; do
  for cfg in "64 12" "128 6" "96 8"; do
    set -- $cfg
    echo "batch $1 steps $2: $(python bench.py $L --batch $1 --steps $2 --warmup 2 2>/dev/null | tail -1 | cut -c1-200)"
  done
done
