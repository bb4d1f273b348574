#!/bin/bash
# residual epilogue of the four-wave GEMM: bitwise tests, then same-box A/B against the previous build (MD_HIP_LIB)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v14
export PYTHONUNBUFFERED=1
timeout -k 5 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm" > gpurun_out/v14/t.log 2>&1; echo "gemm tests rc=$?"; grep -E "passed|failed|^E  " gpurun_out/v14/t.log | cut -c1-300 | tail -6
for rep in 1 2; do
  for v in new prev; do
    if [ $v = prev ]; then export MD_HIP_LIB=$R/moondream_amd/libmoondream_hip_ab.so; else unset MD_HIP_LIB; fi
    echo "== $v (rep $rep)"; timeout -k 5 200 python tools/sweep_gemm.py tiles=20 rounds=3 epi=2 2>&1 | grep -v "^$" | tail -6
  done
done 2>&1 | tee gpurun_out/v14/ab.txt
unset MD_HIP_LIB
echo "== overlap probe: decode-attention stream next to a GEMM stream whose persistent grid leaves CUs free"
timeout -k 5 200 python tools/overlap_probe.py 0,224,208,192,160 2>&1 | tail -16 | tee gpurun_out/v14/overlap.txt
