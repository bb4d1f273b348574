#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v29
export PYTHONUNBUFFERED=1
timeout -k 5 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm or block_tail" > gpurun_out/v29/t.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^E  " gpurun_out/v29/t.log | cut -c1-300 | tail -4
for rep in 1 2; do
  timeout 100 python tools/bench_decode_gemm.py 2>&1 | grep "us per launch"
  MD_HIP_LIB=$R/moondream_amd/libmoondream_hip_ab.so timeout 100 python tools/bench_decode_gemm.py 2>&1 | grep "us per launch"
done | tee gpurun_out/v29/dec.txt
bash tools/ab_lib.sh 2 2>&1 | tee gpurun_out/v29/ab.txt
