#!/usr/bin/env python3
"""Where does a schedule variant of the four-wave GEMM go wrong?  One-hot activations pick ONE k per row, the
weights encode k, so every output element names the K position that reached it (0 = nothing did)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from moondream_amd import _lib
from moondream_amd.weights import PackedLinear
from tools.sweep_gemm import stream

lib = _lib.load()
BF16 = torch.bfloat16
lib.md_gemm_set_tuning(b"tile", 20)
for variant in [int(v) for v in (sys.argv[1:] or ["0", "1", "2", "3"])]:
    lib.md_gemm_set_tuning(b"w4_variant", variant)
    for (m, n, k) in [(256, 256, 128), (256, 256, 256), (512, 512, 256), (256 * 300, 256, 256), (2048, 2048, 1152)]:
        a = torch.zeros(m, k, dtype=BF16, device="cuda")
        rows = torch.arange(m, device="cuda")
        k0 = (rows * 7 + 3) % k
        a[rows, k0] = 1.0
        kk = torch.arange(k, device="cuda")
        w = ((kk % 250) + 1).to(BF16)[None, :].repeat(n, 1) * 1.0  # value names k (mod 250: exact in bf16)
        lin = PackedLinear(w, torch.zeros(n, dtype=BF16), "cuda")
        c = torch.full((m, lin.n_pad), -7.0, dtype=BF16, device="cuda")
        args = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), lin.struct(), c.data_ptr(), c.stride(0), None, 0, 0, m, 0, 0, 0, None, 0)
        _lib.check(lib.md_gemm_bf16(C.byref(args), stream()))
        torch.cuda.synchronize()
        want = ((k0 % 250) + 1).float()[:, None].expand(m, n)
        got = c[:, :n].float()
        bad = got != want
        nb = int(bad.sum())
        line = f"variant {variant} m={m} n={n} k={k}: {nb} wrong of {m * n}"
        if nb:
            idx = bad.nonzero()
            r, cc = idx[:, 0], idx[:, 1]
            line += (f"; rows {int(r.min())}..{int(r.max())} ({len(torch.unique(r))} distinct), cols {int(cc.min())}..{int(cc.max())} "
                     f"({len(torch.unique(cc))} distinct); wrong rows' k0 pair index (k0 // 64) histogram: "
                     f"{torch.bincount(k0[torch.unique(r)] // 64, minlength=k // 64).tolist()}; tile-row histogram {torch.bincount(torch.unique(r) // 256).tolist()[:12]}; "
                     f"sample got/want {[(float(got[i, j]), float(want[i, j])) for i, j in idx[:6].tolist()]}")
        print(line, flush=True)
lib.md_gemm_set_tuning(b"w4_variant", 0)
lib.md_gemm_set_tuning(b"tile", -1)
