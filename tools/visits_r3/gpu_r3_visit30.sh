#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v30
export PYTHONUNBUFFERED=1
timeout -k 5 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/v30/t.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|^E  " gpurun_out/v30/t.log | cut -c1-300 | tail -5
