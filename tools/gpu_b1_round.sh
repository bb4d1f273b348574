#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -8
