#!/bin/bash
# round 4 visit 2: timing ablations + in-kernel stamps of the LDS-DMA loop (measurement build of the library)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python tools/w4_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_v02_probe.txt
