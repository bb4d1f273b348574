#!/bin/bash
# A/B of tools/patches/attn_mask_limit_variant.patch (built into libmoondream_hip_ab.so by tools/build_ab_patch.sh):
# the attention kernel tests with the variant selected, then interleaved timing of in-tree / patched default / patched lim.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
AB=$R/moondream_amd/libmoondream_hip_ab.so
{
MD_HIP_LIB=$AB MD_ATTN_VARIANT=lim timeout -k 3 60 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention_no_mask or attention_spiky or attention_prefix_lm" 2>&1 | tail -2
for i in 1 2; do
  echo "A (in-tree):"; timeout -k 3 40 python tools/kernel_bench.py attn 2>&1 | grep "^attn"
  echo "B (patched library, default variant):"; MD_HIP_LIB=$AB timeout -k 3 40 python tools/kernel_bench.py attn 2>&1 | grep "^attn"
  echo "C (patched library, MD_ATTN_VARIANT=lim):"; MD_HIP_LIB=$AB MD_ATTN_VARIANT=lim timeout -k 3 40 python tools/kernel_bench.py attn 2>&1 | grep "^attn"
done
} 2>&1 | tee gpurun_out/ab_attn_lim.txt
