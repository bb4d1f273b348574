#!/usr/bin/env python3
"""Tile-order sweep of the four-wave GEMM on the 2B model's layer shapes: md_gemm_set_tuning("group_m", g) = row panels per group of
the persistent tile sequence (a group = g row panels x all column panels, walked contiguously by an XCD's 32 workgroups; 0 = the
library's automatic choice).  Interleaved rounds in one process, bias epilogue.

    python tools/sweep_w4_group_m.py [groups=0,1,2,4,8] [rounds=3]
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
GROUPS = [int(t) for t in opts.get("groups", "0,1,2,4,8").split(",")]
ROUNDS = int(opts.get("rounds", "3"))


def main():
    total = {g: [0.0, 0.0] for g in GROUPS}
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
        res = {g: [] for g in GROUPS}
        for _ in range(ROUNDS):
            for g in GROUPS:
                lib.md_gemm_set_tuning(b"group_m", g)
                res[g].append(2.0 * m * n * k / timeit(run) / 1e12)
        lib.md_gemm_set_tuning(b"group_m", 0)
        line = f"{label:14s} m={m:6d} k={k:5d} n={n:5d}: "
        for g in GROUPS:
            med = statistics.median(res[g])
            line += f" g{g}: {med:7.1f}"
            if not label.endswith("^3"):
                reps = 27 if label.startswith("vit") and "patch" not in label else 24 if label.startswith("text") else 1
                total[g][0] += reps * 2.0 * m * n * k
                total[g][1] += reps * 2.0 * m * n * k / (med * 1e12)
        print(line, flush=True)
    for g in GROUPS:
        print(f"model-weighted group_m={g}: {total[g][0] / total[g][1] / 1e12:7.1f} TF/s, {total[g][1] * 1e3:6.1f} ms")
    lib.md_gemm_set_tuning(b"tile", -1)


main()
