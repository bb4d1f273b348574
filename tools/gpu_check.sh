#!/bin/bash
# One GPU-box visit: kernel parity first (short timeouts: a hand-synchronised kernel that hangs must not
# take the box with it), then the GEMM sweep, the model-level parity tests and a short bench.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
STAGE=${1:-all}
rocminfo 2>/dev/null | grep -m1 "Marketing Name.*MI3" > gpurun_out/device.txt
if [[ $STAGE == all || $STAGE == kernels ]]; then
  timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x --tb=short -p no:cacheprovider -k "w4 or identity or bitwise" > gpurun_out/tests_w4.log 2>&1
  echo "w4 tests rc=$?"; tail -5 gpurun_out/tests_w4.log
  timeout 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider > gpurun_out/tests_k.log 2>&1
  echo "kernel tests rc=$?"; grep -E "passed|failed" gpurun_out/tests_k.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/tests_k.log | head -20
fi
if [[ $STAGE == all || $STAGE == sweep ]]; then
  timeout 400 python tools/sweep_gemm.py tiles=20,11,15 rounds=2 2>&1 | grep -v amdgpu.ids > gpurun_out/sweep_gemm.log
  cat gpurun_out/sweep_gemm.log
fi
if [[ $STAGE == all || $STAGE == model ]]; then
  timeout 900 python -m pytest tests/test_model_gpu.py -q --tb=short -p no:cacheprovider -s > gpurun_out/tests_m.log 2>&1
  echo "model tests rc=$?"; grep -E "passed|failed" gpurun_out/tests_m.log | tail -3; grep -E "^FAILED|^ERROR|bench64 parity|rel-rms vs fp64" gpurun_out/tests_m.log | head -30
fi
if [[ $STAGE == all || $STAGE == bench ]]; then
  timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --latency-runs 3 > gpurun_out/bench.log 2>&1
  tail -1 gpurun_out/bench.log | cut -c1-1800
fi
