#!/bin/bash
# Round 5: prefill-attention variants on one box (MD_ATTN_VARIANT letters: csrc/attention.hip, md_attention_prefill):
# kernel tests under the asm-owned-accumulator kernels, then tools/kernel_bench.py attn per variant, interleaved twice.
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; mkdir -p $O
LOG=$O/r05_attn_variants.txt; : > $LOG
for v in x y; do
  echo "== tests MD_ATTN_VARIANT=$v" >> $LOG
  MD_ATTN_VARIANT=$v timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or attn" 2>&1 | tail -3 >> $LOG
done
for rep in 1 2; do
  for v in d p a b pb pbs s x y; do
    echo "== rep $rep MD_ATTN_VARIANT=$v" >> $LOG
    MD_ATTN_VARIANT=$v timeout 120 python tools/kernel_bench.py attn 2>&1 | grep "^attn" >> $LOG
  done
done
cat $LOG
