#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v16
export PYTHONUNBUFFERED=1
timeout -k 5 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -s > gpurun_out/v16/t.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|^E  |2b LoRA|2b reasoning|bench64 parity|detect13" gpurun_out/v16/t.log | cut -c1-300 | tail -14
