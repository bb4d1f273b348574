#!/usr/bin/env python3
"""Sweep the split-K slice count of the decode-regime GEMM on the GPU."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moondream_amd import _lib
from moondream_amd.weights import PackedLinear
from tools.kernel_bench import timeit, stream
lib = _lib.load()
BF16 = torch.bfloat16
for (m, k, n, epi) in [(64, 2048, 14336, 1), (64, 2048, 6144, 0), (64, 2048, 2048, 2), (64, 2048, 8192, 1), (64, 8192, 2048, 2), (1, 2048, 6144, 0), (1, 8192, 2048, 2)]:
    a = (torch.randn(m, k, device="cuda") * 0.5).to(BF16)
    w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).to(BF16)
    lin = PackedLinear(w, torch.zeros(n, dtype=BF16), "cuda")
    c = torch.empty(m, lin.n_pad, dtype=BF16, device="cuda")
    r = torch.randn(m, lin.n_pad, device="cuda").to(BF16)
    st = lin.struct()
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
    args = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), st, c.data_ptr(), c.stride(0), r.data_ptr(), r.stride(0), 0, m, epi, 0, 0, ws.data_ptr(), ws.numel())
    out = []
    for cfg, nt in (("64", "0"), ("helpers", "0")):
        lib.md_gemm_set_tuning(b"decode_cfg", 10 if cfg == "64" else 16); lib.md_gemm_set_tuning(b"decode_nt", int(nt))
        best = None
        for sl in (1, 2, 4, 8):
            lib.md_gemm_set_tuning(b"decode_slices", sl)
            dt = timeit(lambda: _lib.check(lib.md_gemm_bf16(C.byref(args), stream())), iters=20)
            if best is None or dt < best[0]: best = (dt, sl)
        lib.md_gemm_set_tuning(b"decode_slices", 0)
        dflt = timeit(lambda: _lib.check(lib.md_gemm_bf16(C.byref(args), stream())), iters=20)
        out.append(f"cfg={cfg}{'+nt' if nt=='1' else ''}: default {dflt*1e6:5.1f}us best {best[0]*1e6:5.1f}us@S={best[1]}")
    print(f"m={m} k={k} n={n}: " + " | ".join(out) + f"   (ideal {2*n*k/6e12*1e6:.1f}us @6TB/s)", flush=True)
