#!/bin/bash
# Round 5 probe: a decode block's fc2 partials on a second stream beside the attention (MD_DECODE_FORK=1: a fork / join per block inside the
# captured step) vs the shipped single-stream chain, whole bench step, same box, interleaved.
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; LOG=$O/r05_decode_fork_ab.txt; : > $LOG
LEGS="--no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --no-fp8-full-leg --no-strict-leg --no-second-oracle"
for rep in 1 2; do
  for c in 0 1; do
    echo "== rep $rep MD_DECODE_FORK=$c" >> $LOG
    MD_DECODE_FORK=$c timeout 300 python bench.py $LEGS --steps 8 --warmup 2 2>$O/r05_decode_fork_err_$c.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('images/s %.1f  ms/step %.1f  decode_gemm %.3f  phase %s  p50 latency %.2f ms  parity_exact %s ok %s' % (d['value'], d['ms_per_step'], d['decode_gemm']['frac'], d['phase_ms'], d['p50_caption_latency_ms'], d.get('parity_exact'), d.get('parity_ok')))" >> $LOG 2>&1
  done
done
cat $LOG; tail -3 $O/r05_decode_fork_err_1.log
