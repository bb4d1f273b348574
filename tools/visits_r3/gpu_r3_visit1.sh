#!/bin/bash
# round 3, visit 1: schedule variants of the four-wave GEMM (ring depth x write spacing), all GPU tests (new parity
# instrument, strict batch invariance, b1 support query), one bench line.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v1
export PYTHONUNBUFFERED=1
timeout -k 5 200 python tools/sweep_w4_variants.py variants=0,1,2,3 rounds=3 2>&1 | grep -v amdgpu.ids > gpurun_out/v1/w4_variants.txt; echo "variants rc=$?"; cat gpurun_out/v1/w4_variants.txt
timeout -k 5 240 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/v1/tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed" gpurun_out/v1/tests.log | tail -2; grep -E "^FAILED|^ERROR|bench64 parity|batch vs sequential" gpurun_out/v1/tests.log | head -20
timeout -k 5 150 python bench.py --no-cpu-baseline > gpurun_out/v1/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/v1/bench.log | cut -c1-2500
