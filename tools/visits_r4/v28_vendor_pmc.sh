#!/bin/bash
# round 4 visit 28: vendor-library reference on the same box (final round-4 kernel + tile order), then the FETCH_SIZE / WRITE_SIZE passes over one eager B=64 step
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 300 python tools/vendor_gemm_compare.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_v28_vendor.txt
bash tools/gpu_pmc_traffic.sh 2>&1 | tail -3
cp gpurun_out/pmc_traffic.json gpurun_out/r04_pmc_traffic.json
