#!/usr/bin/env python3
"""Timing ablations of the four-wave 256x256 GEMM (gemm_w4.hip): what each class of filler costs.
Variants are compiled into the library for the bias epilogue only (results are garbage for
ablations); code = SCHED + 16 * ABL, ABL bits: 1 no ds_write, 2 no global loads, 4 no barrier,
8 no fragment reads, 16 no epilogue stores."""
import ctypes as C
import math
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from moondream_amd import _lib
from moondream_amd.weights import PackedLinear
from tools.sweep_gemm import timeit, stream

lib = _lib.load()
BF16 = torch.bfloat16
VARIANTS = [("w4", 20, 0), ("no ds_write", 20, 16), ("no loads", 20, 32), ("no write+loads", 20, 48),
            ("no barrier", 20, 64), ("no frag reads", 20, 128), ("no stores", 20, 256), ("mfma+stores only", 20, 240),
            ("mfma only", 20, 496), ("vmcnt0 after stores", 20, 1024), ("nt stores", 20, 2048), ("8-wave alt", 11, 0)]
SHAPES = [(8192, 8192, 8192), (46720, 2048, 2048), (93312, 1152, 3456)]
if len(sys.argv) > 1:
    VARIANTS = [v for v in VARIANTS if v[0] in sys.argv[1:] or str(v[2]) in sys.argv[1:]] or VARIANTS

for m, k, n in SHAPES:
    a = (torch.randn(m, k, device="cuda") * 0.5).to(BF16)
    w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).to(BF16)
    lin = PackedLinear(w, torch.zeros(n, dtype=BF16), "cuda")
    c = torch.empty(m, lin.n_pad, dtype=BF16, device="cuda")
    args = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), lin.struct(), c.data_ptr(), c.stride(0), None, 0, 0, m, 0, 0, 0, None, 0)
    res = {v[0]: [] for v in VARIANTS}
    for _ in range(2):
        for name, tile, var in VARIANTS:
            lib.md_gemm_set_tuning(b"tile", tile)
            lib.md_gemm_set_tuning(b"w4_variant", var)
            dt = timeit(lambda: _lib.check(lib.md_gemm_bf16(C.byref(args), stream())), iters=5, warm=2)
            res[name].append(2.0 * m * n * k / dt / 1e12)
    lib.md_gemm_set_tuning(b"tile", -1)
    lib.md_gemm_set_tuning(b"w4_variant", 0)
    print(f"m={m} k={k} n={n}: " + " | ".join(f"{nm}: {statistics.median(v):6.0f}" for nm, v in res.items()), flush=True)
