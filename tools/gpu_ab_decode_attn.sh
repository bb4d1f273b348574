#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2 3; do
  echo "A (in-tree):"; timeout 120 python tools/kernel_bench.py decode 2>&1 | grep decode-attn
  echo "B (MD_HIP_LIB = build with the previous attention.hip):"; MD_HIP_LIB=$R/moondream_amd/libmoondream_hip_ab.so timeout 120 python tools/kernel_bench.py decode 2>&1 | grep decode-attn
done
