#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v2
export PYTHONUNBUFFERED=1
timeout -k 5 120 python tools/diag_w4_variant.py 2>&1 | grep -v amdgpu.ids > gpurun_out/v2/diag.txt; cat gpurun_out/v2/diag.txt | cut -c1-900
timeout -k 5 120 python tools/repeat_pipelined.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/v2/repeat.txt; tail -12 gpurun_out/v2/repeat.txt
timeout -k 5 200 python tools/sweep_w4_variants.py variants=0,1,2,3 rounds=3 2>&1 | grep -v amdgpu.ids > gpurun_out/v2/w4_variants.txt; cat gpurun_out/v2/w4_variants.txt
timeout -k 5 200 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -s -k "full_size or pipelined" > gpurun_out/v2/tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed" gpurun_out/v2/tests.log | tail -2; grep -E "^FAILED|^ERROR|bench64 parity|batch vs sequential|^E  " gpurun_out/v2/tests.log | cut -c1-400 | head -20
