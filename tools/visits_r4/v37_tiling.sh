#!/bin/bash
# round 4 visit 37: identity-resize fast path in the host tiling (crop-sized images): model tests, then the default bench (3 timed steps)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-fp8-full-leg --no-dedup-leg --no-detect13-leg --no-second-oracle --no-strict-leg --latency-runs 3 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],1), r['phase_ms'], round(r['roofline']['achieved']), r['p50_caption_latency_ms'], r['parity_ok'])"
done | tee gpurun_out/r04_v37_bench.txt
