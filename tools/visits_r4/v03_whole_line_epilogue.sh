#!/bin/bash
# round 4 visit 3: whole-line stores through the pipelined LDS transposition (every epilogue) + per-tile bias slot; A/B against the
# previous build (libmoondream_hip_ab.so = commit "LDS-DMA operand path ...": MD_W4_VARIANT=0 register-staged, =1 LDS-DMA, old epilogues)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -8 | tee gpurun_out/r04_v03_tests.txt
( echo "== in-tree (LDS-DMA + whole-line epilogue)"; timeout 300 python tools/sweep_gemm.py tiles=20 rounds=3
  echo "== previous build, LDS-DMA loop, old epilogues"; MD_HIP_LIB=$R/moondream_amd/libmoondream_hip_ab.so MD_W4_VARIANT=1 timeout 300 python tools/sweep_gemm.py tiles=20 rounds=3
  echo "== previous build, register-staged loop (round 3 kernel)"; MD_HIP_LIB=$R/moondream_amd/libmoondream_hip_ab.so MD_W4_VARIANT=0 timeout 300 python tools/sweep_gemm.py tiles=20 rounds=3
  echo "== in-tree again"; timeout 300 python tools/sweep_gemm.py tiles=20 rounds=3 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_v03_sweep.txt
