#!/bin/bash
# round 4 visit 21: same-box A/B of the pipelined engine: 1 = one stream (host-side pipelining only), 2 = decode on its own priority
# stream (kernels of the two streams overlap), 3 = 2 + rounds 1-3's synchronous id collection on the default stream; twice, interleaved
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for rep in 1 2; do
  for m in 1 2 3; do
    echo -n "rep $rep MD_PIPE_STREAMS=$m: "
    MD_PIPE_STREAMS=$m timeout 200 python bench.py --steps 8 --warmup 2 --only-timed-steps 2>&1 | grep '^{' | tail -1
  done
done | tee gpurun_out/r04_v21_engine_ab.txt
echo -n "no-pipeline: "; timeout 200 python bench.py --steps 6 --warmup 2 --only-timed-steps --no-pipeline 2>&1 | grep '^{' | tail -1 | tee -a gpurun_out/r04_v21_engine_ab.txt
