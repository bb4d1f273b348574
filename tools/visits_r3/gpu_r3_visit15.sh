#!/bin/bash
# pipelined timed region with the tile GEMMs on fewer than 256 CUs: does the decode stream co-run on the CUs left free?
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v15
export PYTHONUNBUFFERED=1
for g in 0 208 176 0 192 160 144; do
  timeout -k 5 200 python bench.py --steps 6 --warmup 2 --only-timed-steps --w4-grid $g 2>/dev/null | grep only_timed | sed "s/^/grid $g: /"
done 2>&1 | tee gpurun_out/v15/grid.txt
