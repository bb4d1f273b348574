#!/bin/bash
# Round 5: crops per ViT launch group (--vit-chunk) in the timed configuration, same box, interleaved twice.  128 = all crops of a
# B = 64 step in one group (activations 215 MB per layer: beyond the 256 MB Infinity Cache together with the next layer's); smaller
# groups keep producer -> consumer activations cache-resident but quantise the tile rounds of the persistent GEMM harder.
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; LOG=$O/r05_vit_chunk_sweep.txt; : > $LOG
for rep in 1 2; do
  for c in 128 64 32 96; do
    echo "== rep $rep --vit-chunk $c" >> $LOG
    python bench.py --steps 8 --warmup 2 --only-timed-steps --vit-chunk $c 2>/dev/null | tail -1 >> $LOG
  done
done
cat $LOG
