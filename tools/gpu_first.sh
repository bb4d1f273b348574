#!/bin/bash
# First on-GPU pass: kernel parity, model parity, a small bench.  Logs -> gpurun_out/
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
rocminfo | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt 2>&1
nproc >> gpurun_out/device.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/kernels.log 2>&1
echo "kernels exit $?" >> gpurun_out/kernels.log
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "not full_size" -s > gpurun_out/model.log 2>&1
echo "model exit $?" >> gpurun_out/model.log
timeout 600 python bench.py --steps 1 --warmup 1 --batch 8 --tokens 8 --no-cpu-baseline > gpurun_out/bench_small.log 2>&1
echo "bench exit $?" >> gpurun_out/bench_small.log
tail -5 gpurun_out/kernels.log; tail -5 gpurun_out/model.log; tail -3 gpurun_out/bench_small.log
