#!/bin/bash
# round 4 visit 8: the four-wave kernel on v_mfma_f32_16x16x32_bf16: kernel tests, then the sweep against the 32x32x16 build (ab)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or rope" 2>&1 | tail -25 | tee gpurun_out/r04_v08_tests.txt
( echo "== in-tree (16x16x32)"; timeout 300 python tools/sweep_gemm.py tiles=20 rounds=3
  echo "== ab (32x32x16, LDS-DMA, whole-line epilogue)"; MD_HIP_LIB=$R/moondream_amd/libmoondream_hip_ab.so timeout 300 python tools/sweep_gemm.py tiles=20 rounds=3
  echo "== in-tree again"; timeout 300 python tools/sweep_gemm.py tiles=20 rounds=3 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_v08_sweep.txt
