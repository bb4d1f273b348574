#!/usr/bin/env python3
"""Interleaved A/B of the four-wave GEMM's schedule variants (md_gemm_set_tuning "w4_variant": 0 = shipped,
1 = two-pair register ring (NP 32), 2 = writes every 2nd gap (WS 2), 3 = both) on the 2B model's layer
shapes -- ALL with the bias epilogue (the variants exist for that kernel only), so the numbers compare
main loops, not epilogues.  Every variant's output is checked bit-identical to variant 0's first.

    python tools/sweep_w4_variants.py [variants=0,1,2,3] [rounds=3]
"""
import ctypes as C
import math
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from moondream_amd import _lib
from moondream_amd.weights import PackedLinear
from tools.sweep_gemm import SHAPES, stream, timeit

lib = _lib.load()
BF16 = torch.bfloat16
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
VARIANTS = [int(t) for t in opts.get("variants", "0,1,2,3").split(",")]
ROUNDS = int(opts.get("rounds", "3"))


def main():
    total = {v: [0.0, 0.0] for v in VARIANTS}
    lib.md_gemm_set_tuning(b"tile", 20)
    for m, k, n, _epi, label in SHAPES:
        kp = (k + 63) // 64 * 64
        a = (torch.randn(m, kp, device="cuda") * 0.5).to(BF16)
        if kp > k:
            a[:, k:] = 0
        w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).to(BF16)
        lin = PackedLinear(w, torch.randn(n).to(BF16), "cuda")
        c = torch.empty(m, lin.n_pad, dtype=BF16, device="cuda")
        args = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), lin.struct(), c.data_ptr(), c.stride(0), None, 0, 0, m, 0, 0, 0, None, 0)
        run = lambda: _lib.check(lib.md_gemm_bf16(C.byref(args), stream()))
        ref = None
        for v in VARIANTS:
            lib.md_gemm_set_tuning(b"w4_variant", v)
            c.zero_()
            run()
            torch.cuda.synchronize()
            if ref is None:
                ref = c.clone()
            elif not torch.equal(c, ref):
                nbad = int((c != ref).sum())
                print(f"!! {label}: variant {v} differs from variant {VARIANTS[0]} in {nbad} of {c.numel()} elements (timed anyway)", flush=True)
        res = {v: [] for v in VARIANTS}
        for _ in range(ROUNDS):
            for v in VARIANTS:
                lib.md_gemm_set_tuning(b"w4_variant", v)
                res[v].append(2.0 * m * n * k / timeit(run) / 1e12)
        lib.md_gemm_set_tuning(b"w4_variant", 0)
        line = f"{label:14s} m={m:6d} k={k:5d} n={n:5d}: "
        for v in VARIANTS:
            med = statistics.median(res[v])
            line += f" v{v}: {med:7.1f} ({min(res[v]):5.0f}..{max(res[v]):5.0f})"
            if label[0] in "vpt":
                mult = 27 if label.startswith("vit") and "patch" not in label else (24 if label.startswith("text") else 1)
                total[v][0] += mult * 2.0 * m * n * k
                total[v][1] += mult * 2.0 * m * n * k / (med * 1e12)
        print(line, flush=True)
        del a, w, c, lin, ref
    lib.md_gemm_set_tuning(b"tile", -1)
    for v in VARIANTS:
        fl, tm = total[v]
        print(f"model-weighted (bias epilogue everywhere) v{v}: {fl / tm / 1e12:7.1f} TF/s, {tm * 1e3:6.1f} ms", flush=True)


if __name__ == "__main__":
    main()
