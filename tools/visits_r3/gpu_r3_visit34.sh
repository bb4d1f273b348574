#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v34
export PYTHONUNBUFFERED=1
timeout -k 5 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "f8" > gpurun_out/v34/t.log 2>&1; echo "f8 tests rc=$?"; grep -E "passed|failed|^E  " gpurun_out/v34/t.log | cut -c1-300 | tail -3
for rep in 1 2; do for v in new prev; do
  if [ $v = prev ]; then export MD_HIP_LIB=$R/moondream_amd/libmoondream_hip_ab.so; else unset MD_HIP_LIB; fi
  echo "== $v (rep $rep)"; timeout -k 5 300 python tools/sweep_gemm_f8.py rounds=2 2>&1 | grep -v amdgpu.ids | awk '{print}' | sed 's/bf16: *[0-9.]* //' | tail -13
done; done 2>&1 | tee gpurun_out/v34/ab.txt
