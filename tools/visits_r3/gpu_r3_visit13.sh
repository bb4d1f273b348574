#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v13
export PYTHONUNBUFFERED=1
timeout -k 5 200 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -s -k "e4m3 or f8" > gpurun_out/v13/t.log 2>&1; echo "f8 kernel tests rc=$?"; grep -E "passed|failed|^E  " gpurun_out/v13/t.log | cut -c1-300 | tail -6
bash tools/gpu_r3_profile.sh
