#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "single_sequence" 2>&1 | tail -5
timeout 200 python tools/b1_phase_times.py 2>&1 | grep -v amdgpu.ids | tail -8
timeout 200 python tools/latency_profile.py 2>&1 | grep "^run 2"
