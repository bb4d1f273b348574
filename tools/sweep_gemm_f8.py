#!/usr/bin/env python3
"""md_gemm_f8 (fp8 operands, v_mfma_f32_32x32x64_f8f6f4) against the bf16 four-wave kernel on the 2B model's layer
shapes (real epilogues; random operands), interleaved in one process.  TF/s counts 2 m n k for both."""
import ctypes as C
import math
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from moondream_amd import _lib
from moondream_amd.weights import PackedLinear, PackedLinearF8
from tools.sweep_gemm import SHAPES, stream, timeit

lib = _lib.load()
BF16 = torch.bfloat16
F8 = torch.float8_e4m3fn
ROUNDS = int(dict(a.split("=") for a in sys.argv[1:] if "=" in a).get("rounds", "3"))


def main():
    tot = {"bf16": [0.0, 0.0], "fp8": [0.0, 0.0]}
    for m, k, n, epi, label in SHAPES:
        kp = (k + 63) // 64 * 64
        a = (torch.randn(m, kp, device="cuda") * 0.5).to(BF16)
        if kp > k:
            a[:, k:] = 0
        w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).to(BF16)
        lin = PackedLinear(w, torch.randn(n).to(BF16), "cuda")
        lin8 = PackedLinearF8(lin.w, lin.b, n, k)
        a_scale = float(a.float().abs().max()) / 448.0
        a8 = (a.float() / a_scale).clamp(-448, 448).to(F8).view(torch.uint8).contiguous()
        c = torch.empty(m, lin.n_pad, dtype=BF16, device="cuda")
        r = torch.randn(m, lin.n_pad, device="cuda").to(BF16)
        args = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), lin.struct(), c.data_ptr(), c.stride(0), r.data_ptr(), r.stride(0), 0, m,
                               epi, 1 if epi == 1 else 0, 0, None, 0)
        args8 = _lib.MdGemmF8Args(a8.data_ptr(), a8.stride(0), a_scale, lin8.struct(), c.data_ptr(), c.stride(0), None, 0, 1.0, 0,
                                  r.data_ptr(), r.stride(0), 0, m, epi, 1 if epi == 1 else 0, 0)
        runs = {"bf16": lambda: _lib.check(lib.md_gemm_bf16(C.byref(args), stream())),
                "fp8": lambda: _lib.check(lib.md_gemm_f8(C.byref(args8), stream()))}
        res = {kk: [] for kk in runs}
        for _ in range(ROUNDS):
            for kk, fn in runs.items():
                res[kk].append(2.0 * m * n * k / timeit(fn) / 1e12)
        line = f"{label:14s} m={m:6d} k={k:5d} n={n:5d} epi={epi}: "
        for kk in runs:
            med = statistics.median(res[kk])
            line += f" {kk}: {med:7.1f}"
            if label[0] in "vpt":
                mult = 27 if label.startswith("vit") and "patch" not in label else (24 if label.startswith("text") else 1)
                tot[kk][0] += mult * 2.0 * m * n * k
                tot[kk][1] += mult * 2.0 * m * n * k / (med * 1e12)
        print(line + f"   x{statistics.median(res['fp8']) / statistics.median(res['bf16']):.2f}", flush=True)
        del a, w, c, r, lin, lin8, a8
    for kk in tot:
        fl, tm = tot[kk]
        print(f"model-weighted (one B=64 step's tile GEMMs) {kk}: {fl / tm / 1e12:7.1f} TF/s, {tm * 1e3:6.1f} ms", flush=True)


if __name__ == "__main__":
    main()
