#!/bin/bash
# round 4 visit 31: the reference's own int4 checkpoint format as the decode regime's weight stream (md_linear_fp8.format 1)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "int4 or fp8" 2>&1 | tail -8 | tee gpurun_out/r04_v31_tests.txt
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -s -k "int4_checkpoint or fp8_decode or fp8" 2>&1 | grep -v amdgpu.ids | tail -12 | tee -a gpurun_out/r04_v31_tests.txt
