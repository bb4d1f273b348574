#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -2
for i in 1 2 3; do
  echo "A (in-tree):"; timeout 120 python tools/kernel_bench.py attn 2>&1 | grep -i "attn"
  echo "B (MD_HIP_LIB = previous attention.hip):"; MD_HIP_LIB=$R/moondream_amd/libmoondream_hip_ab.so timeout 120 python tools/kernel_bench.py attn 2>&1 | grep -i "attn"
done
