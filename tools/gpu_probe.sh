#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python tools/w4_probe.py "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/w4_probe.log
