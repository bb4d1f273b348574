#!/bin/bash
# Round 6: the pipelined engine with the decode of two consecutive batches merged into one 128-row lockstep (MD_PIPE_PAIR=1, default)
# against one decode per batch (MD_PIPE_PAIR=0) -- same box, interleaved, timed region only
R=$GRAFT_REPO_ROOT; cd $R
L="--no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --no-fp8-full-leg --no-strict-leg --no-second-oracle --latency-runs 0 --only-timed-steps"
for rep in 1 2 3; do
  for pair in 0 1; do
    echo "MD_PIPE_PAIR=$pair: $(MD_PIPE_PAIR=$pair python bench.py $L --steps 12 --warmup 4 2>/dev/null | tail -1 | cut -c1-200)"
  done
done
