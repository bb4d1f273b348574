#!/bin/bash
# register-staged decode-regime GEMM (decode_cfg 19): bitwise tests, kernel sweep, then the bench A/B (MD_DECODE_CFG=r | h)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v28
export PYTHONUNBUFFERED=1
timeout -k 5 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "decode_regime or block_tail" > gpurun_out/v28/t.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^E  " gpurun_out/v28/t.log | cut -c1-300 | tail -4
timeout -k 5 300 python tools/sweep_decode.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/v28/sweep.txt | cut -c1-260
LEGS="--no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --no-fp8-full-leg --latency-runs 0"
for rep in 1 2; do for v in r h; do
  MD_DECODE_CFG=$v timeout -k 5 300 python bench.py $LEGS --steps 6 --warmup 2 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('decode_cfg=$v rep $rep: %.1f images/s  ms/step %.1f | decode phase %.2f ms  decode_step.frac %.3f decode_gemm.frac %.3f | parity %s/%s ok=%s | p50 %.1f' % (d['value'], d['ms_per_step'], d['phase_ms']['decode'], d['decode_step']['frac'], d['decode_gemm']['frac'], d['parity_exact'], d['parity_checked'], d['parity_ok'], d.get('p50_caption_latency_ms') or 0))"
done; done 2>&1 | tee gpurun_out/v28/ab.txt
