#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v6
export PYTHONUNBUFFERED=1
timeout -k 5 90 python tools/debug_stale_graph.py 2>&1 | grep -v amdgpu.ids | grep "== ref\|max abs" | tee gpurun_out/v6/stale.txt
timeout -k 5 100 python tools/repeat_pipelined.py 4 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-200 | tee gpurun_out/v6/repeat.txt
timeout -k 5 300 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/v6/tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed" gpurun_out/v6/tests.log | tail -2; grep -E "^FAILED|^ERROR|detect13|^E  " gpurun_out/v6/tests.log | cut -c1-300 | head -20
timeout -k 5 240 python bench.py --no-cpu-baseline > gpurun_out/v6/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/v6/bench.log > gpurun_out/v6/bench.json; python - <<'P'
import json
d=json.load(open('gpurun_out/v6/bench.json'))
for k in ('value','ms_per_step','phase_ms','vit_encoder','p50_caption_latency_ms','detect13','parity_exact','parity_ok'):
    print(k, json.dumps(d.get(k))[:700])
print('roofline', d['roofline']['achieved'], d['roofline']['frac'])
P
timeout -k 5 240 python tools/sweep_w4_stagger.py settings=0,1601,801,1602,1603 rounds=2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/v6/stagger.txt
