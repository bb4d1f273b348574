#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python tools/sweep_decode.py > gpurun_out/sweep_decode.log 2>&1
cat gpurun_out/sweep_decode.log | grep -v amdgpu.ids
for ch in 32 64 128; do
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --latency-runs 0 --vit-chunk $ch > gpurun_out/bench_chunk$ch.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/bench_chunk$ch.log') if x.startswith('{')]
d=json.loads(l[-1]); print('chunk $ch:', round(d['value'],1),'img/s', round(d['ms_per_step'],1),'ms  gemm', round(d['roofline']['achieved'],1),'TF share',round(d['roofline']['share_of_step'],3))
PY
done
