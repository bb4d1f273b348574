#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "decode or argmax or attention" 2>&1 | tail -3
timeout 200 python tools/latency_profile.py 2>&1 | grep "^run"
