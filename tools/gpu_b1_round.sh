#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -s 2>&1 | grep -E "passed|failed|rows:|Error|error" | tail -12
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/bench_dedup.log 2>&1; tail -1 gpurun_out/bench_dedup.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['phase_ms'], d['roofline']['frac'], d.get('p50_caption_latency_ms'), d['parity_exact'], d['parity_ok'])
print(d.get('dedup_identical_crops'))
print(d.get('vqa32'))
print({k:v for k,v in d.get('fp8_decode',{}).items() if k!='note'})"
