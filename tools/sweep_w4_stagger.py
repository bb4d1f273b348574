#!/usr/bin/env python3
"""Interleaved A/B of the four-wave GEMM's start-up stagger (md_gemm_set_tuning "w4_stagger" = mode + 16 * span-%):
mode 1 staggers the 8 XCDs, 2 the 32 CUs of every XCD, 3 all workgroups, over span-% of the estimated tile time, so
that the tiles' epilogues stop hitting the memory system from every CU in the same microsecond.  The model's layer
shapes with their REAL epilogues; outputs are checked bit-identical to the unstaggered launch.

    python tools/sweep_w4_stagger.py [settings=0,1601,801,1602,1603] [rounds=3]
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
SETTINGS = [int(t) for t in opts.get("settings", "0,1601,801,1602,1603").split(",")]
ROUNDS = int(opts.get("rounds", "3"))


def main():
    total = {v: [0.0, 0.0] for v in SETTINGS}
    lib.md_gemm_set_tuning(b"tile", 20)
    for m, k, n, epi, label in SHAPES:
        kp = (k + 63) // 64 * 64
        a = (torch.randn(m, kp, device="cuda") * 0.5).to(BF16)
        if kp > k:
            a[:, k:] = 0
        w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).to(BF16)
        lin = PackedLinear(w, torch.randn(n).to(BF16), "cuda")
        c = torch.empty(m, lin.n_pad, dtype=BF16, device="cuda")
        r = torch.randn(m, lin.n_pad, device="cuda").to(BF16)
        args = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), lin.struct(), c.data_ptr(), c.stride(0), r.data_ptr(), r.stride(0), 0, m,
                               epi, 1 if epi == 1 else 0, 0, None, 0)
        run = lambda: _lib.check(lib.md_gemm_bf16(C.byref(args), stream()))
        ref = None
        for v in SETTINGS:
            lib.md_gemm_set_tuning(b"w4_stagger", v)
            c.zero_()
            run()
            torch.cuda.synchronize()
            if ref is None:
                ref = c.clone()
            elif not torch.equal(c, ref):
                print(f"!! {label}: setting {v} changes the result", flush=True)
        res = {v: [] for v in SETTINGS}
        for _ in range(ROUNDS):
            for v in SETTINGS:
                lib.md_gemm_set_tuning(b"w4_stagger", v)
                res[v].append(2.0 * m * n * k / timeit(run) / 1e12)
        lib.md_gemm_set_tuning(b"w4_stagger", 0)
        line = f"{label:14s} m={m:6d} k={k:5d} n={n:5d} epi={epi}: "
        for v in SETTINGS:
            med = statistics.median(res[v])
            line += f" s{v}: {med:7.1f}"
            if label[0] in "vpt":
                mult = 27 if label.startswith("vit") and "patch" not in label else (24 if label.startswith("text") else 1)
                total[v][0] += mult * 2.0 * m * n * k
                total[v][1] += mult * 2.0 * m * n * k / (med * 1e12)
        print(line, flush=True)
        del a, w, c, r, lin, ref
    lib.md_gemm_set_tuning(b"tile", -1)
    for v in SETTINGS:
        fl, tm = total[v]
        print(f"model-weighted (one B=64 step's tile GEMMs) s{v}: {fl / tm / 1e12:7.1f} TF/s, {tm * 1e3:6.1f} ms", flush=True)


if __name__ == "__main__":
    main()
