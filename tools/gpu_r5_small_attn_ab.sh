#!/bin/bash
# Round 5: the decode attention in a footprint that CO-RESIDES with the persistent tile-GEMM workgroups (64 registers, 12.4 KiB LDS)
# in the pipelined two-stream engine, against the shipped shape; same box, interleaved.
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; LOG=$O/r05_small_attn_ab.txt; : > $LOG
LEGS="--no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --no-fp8-full-leg --no-strict-leg --no-second-oracle --latency-runs 0"
for rep in 1 2; do
  for v in 0 1; do
    echo "== rep $rep MD_DECODE_ATTN_SMALL=$v" >> $LOG
    MD_DECODE_ATTN_SMALL=$v python bench.py $LEGS --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('images/s %.1f  ms/step %.1f  phase %s  parity_exact %s ok %s' % (d['value'], d['ms_per_step'], d['phase_ms'], d.get('parity_exact'), d.get('parity_ok')))" >> $LOG
  done
done
cat $LOG
