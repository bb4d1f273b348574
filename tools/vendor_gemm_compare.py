#!/usr/bin/env python3
"""Measurement reference only: the vendor library (torch.nn.functional.linear -> hipBLASLt / rocBLAS)
next to md_gemm_bf16 on the model's GEMM shapes, same box, same operands.  Not used by the product."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moondream_amd import _lib
from moondream_amd.weights import PackedLinear
from tools.kernel_bench import timeit, stream
lib = _lib.load(); BF16 = torch.bfloat16
shapes = [(93312, 1152, 3456), (93312, 1152, 4304), (93312, 1152, 1152), (93312, 4304, 1152),
          (46720, 2048, 14336), (46720, 2048, 2048), (46720, 8192, 2048), (4096, 4096, 4096), (8192, 8192, 8192)]
if os.environ.get("VENDOR_SHAPES"):  # "m,k,n;m,k,n": a subset for the PMC passes
    shapes = [tuple(int(x) for x in t.split(",")) for t in os.environ["VENDOR_SHAPES"].split(";")]
for (m, k, n) in shapes:
    kp = (k + 63) // 64 * 64
    a = (torch.randn(m, kp, device="cuda") * 0.5).to(BF16)
    if kp > k: a[:, k:] = 0
    w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).to(BF16)
    b = torch.zeros(n, dtype=BF16, device="cuda")
    lin = PackedLinear(w, b, "cuda")
    c = torch.empty(m, lin.n_pad, dtype=BF16, device="cuda")
    args = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), lin.struct(), c.data_ptr(), c.stride(0), None, 0, 0, m, 0, 0, 0, None, 0)
    t_md = timeit(lambda: _lib.check(lib.md_gemm_bf16(C.byref(args), stream())))
    av = a[:, :k].contiguous() if kp > k else a
    wv = w.contiguous()
    with torch.inference_mode():
        t_lib = timeit(lambda: torch.nn.functional.linear(av, wv, b))
    fl = 2.0 * m * n * k
    print(f"m={m} k={k} n={n}: md_gemm_bf16 {fl/t_md/1e12:7.0f} TF/s | vendor library (torch F.linear + bias) {fl/t_lib/1e12:7.0f} TF/s", flush=True)
