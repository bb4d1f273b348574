#!/bin/bash
# Round 5: decode-regime config 16 (shipped) vs 17 (four compute waves, 128-wide K slices) in the whole bench step, same box, interleaved.
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; LOG=$O/r05_decode_cfg_ab.txt; : > $LOG
LEGS="--no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --no-fp8-full-leg --no-strict-leg --no-second-oracle"
for rep in 1 2; do
  for c in 16 17; do
    echo "== rep $rep MD_DECODE_CFG=$c" >> $LOG
    MD_DECODE_CFG=$c python bench.py $LEGS --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('images/s %.1f  ms/step %.1f  decode_gemm %.3f  phase %s  p50 latency %.2f ms  parity_exact %s ok %s' % (d['value'], d['ms_per_step'], d['decode_gemm']['frac'], d['phase_ms'], d['p50_caption_latency_ms'], d.get('parity_exact'), d.get('parity_ok')))" >> $LOG
  done
done
cat $LOG
