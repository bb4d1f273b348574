#!/bin/bash
# round 4 visit 14: second oracle + mutation sensitivity inside the 2B test, the fp8 KV copy with > 64 sequences, strict mode tests
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu -s -k "full_size_models or fp8_kv_cache_decode or strict or batched_equals" 2>&1 | grep -v "amdgpu.ids" | grep -i "second oracle\|mutation\|bench64 parity\|passed\|failed\|error\|assert\|batch_equals\|strict" | tail -40 | tee gpurun_out/r04_v14_parity.txt
