#!/bin/bash
# round 4 visit 26-27: the graph-replay chain test; tile order (group_m) re-swept on the LDS-DMA kernel, then new rule (0) vs round 3 rule (-1)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "graph_replay or block_tail" 2>&1 | tail -5 | tee gpurun_out/r04_v26_tests.txt
timeout 400 python tools/sweep_w4_group_m.py groups=0,-1 rounds=7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_v27_group_m_confirm.txt
