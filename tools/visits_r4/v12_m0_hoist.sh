#!/bin/bash
# round 4 visit 12: M0 written once per four LDS-DMA pieces (in-tree) against the previous build (ab); kernel tests first
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or rope" 2>&1 | tail -5 | tee gpurun_out/r04_v12_tests.txt
( echo "== in-tree"; timeout 300 python tools/sweep_w4_variants.py variants=0,2,4 rounds=3
  echo "== ab (M0 per piece)"; MD_HIP_LIB=$R/moondream_amd/libmoondream_hip_ab.so timeout 300 python tools/sweep_w4_variants.py variants=0,2 rounds=3 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_v12_sweep.txt
