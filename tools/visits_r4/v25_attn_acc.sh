#!/bin/bash
# round 4 visit 25: prefill attention with the O^T accumulators (default) / S and O^T accumulators (b) in AGPRs against round 3's kernel (d)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for v in d a b; do
  echo "== MD_ATTN_VARIANT=$v kernel tests"; MD_ATTN_VARIANT=$v timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -2
done 2>&1 | tee gpurun_out/r04_v25_attn_tests.txt
for rep in 1 2; do for v in d a b; do
  echo "== MD_ATTN_VARIANT=$v"; MD_ATTN_VARIANT=$v timeout 200 python tools/kernel_bench.py attn 2>&1 | grep "^attn"
done; done | tee gpurun_out/r04_v25_attn_bench.txt
