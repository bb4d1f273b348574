#!/usr/bin/env python3
"""256 x 192 tiles (md_gemm_set_tuning "w4_nj" = 6) against 256 x 256 (8) on the layers whose width is a multiple of 192 but not of 256
(the ViT's N = 1152), every epilogue: outputs compared bit for bit, then timed interleaved, then the tile order (group_m) of the
192-wide kernel.

    python tools/sweep_w4_nj.py [rounds=5]
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
from tools.sweep_gemm import stream, timeit

lib = _lib.load()
BF16 = torch.bfloat16
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
ROUNDS = int(opts.get("rounds", "5"))
SHAPES = [(93312, 1152, 1152, 2, "vit proj"), (93312, 4304, 1152, 2, "vit fc2"), (93312, 588, 1152, 2, "vit patch_emb"),
          (93312, 1152, 1152, 0, "N=1152 bias"), (93312, 1152, 1152, 1, "N=1152 gelu"), (1000, 1152, 1152, 2, "M=1000 residual"),
          (46720, 2048, 2112, 0, "N=2112 (11 x 192)")]


def main():
    lib.md_gemm_set_tuning(b"tile", 20)
    tot = {8: 0.0, 6: 0.0}
    for m, k, n, epi, label in SHAPES:
        kp = (k + 63) // 64 * 64
        a = (torch.randn(m, kp, device="cuda") * 0.5).to(BF16)
        if kp > k:
            a[:, k:] = 0
        w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).to(BF16)
        lin = PackedLinear(w, torch.randn(n).to(BF16), "cuda")
        r = (torch.randn(m, lin.n_pad, device="cuda") * 0.5).to(BF16) if epi == 2 else None
        c = torch.empty(m, lin.n_pad, dtype=BF16, device="cuda")
        args = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), lin.struct(), c.data_ptr(), c.stride(0), r.data_ptr() if r is not None else None,
                               r.stride(0) if r is not None else 0, 0, m, epi, 0, 0, None, 0)
        run = lambda: _lib.check(lib.md_gemm_bf16(C.byref(args), stream()))
        outs = {}
        for nj in (8, 6):
            lib.md_gemm_set_tuning(b"w4_nj", nj)
            c.fill_(float("nan"))
            run()
            torch.cuda.synchronize()
            outs[nj] = c[:, :n].clone()
        same = torch.equal(outs[8], outs[6])
        ref = (a[:, :k].float() @ w.float().t() + lin.b[:n].float())
        if epi == 1:
            ref = torch.nn.functional.gelu(ref.to(BF16).float(), approximate="tanh")
        if epi == 2:
            ref = ref.to(BF16).float() + r[:, :n].float()
        err = float((outs[6].float() - ref).abs().max() / ref.abs().max())
        res = {8: [], 6: []}
        for _ in range(ROUNDS):
            for nj in (8, 6):
                lib.md_gemm_set_tuning(b"w4_nj", nj)
                res[nj].append(2.0 * m * n * k / timeit(run) / 1e12)
        line = f"{label:18s} m={m:6d} k={k:5d} n={n:5d} epi={epi}: bitwise {'EQUAL' if same else 'DIFFERENT !!'} (max err vs fp32 ref {err:.2e})  " \
               f"256-wide {statistics.median(res[8]):7.1f}  192-wide {statistics.median(res[6]):7.1f} TF/s"
        if label.startswith("vit"):
            lib.md_gemm_set_tuning(b"w4_nj", 6)
            for g in (1, 2, 3, 4, 8):
                lib.md_gemm_set_tuning(b"group_m", g)
                line += f" | g{g} {statistics.median([2.0 * m * n * k / timeit(run) / 1e12 for _ in range(3)]):7.1f}"
            lib.md_gemm_set_tuning(b"group_m", 0)
            reps = 27 if "patch" not in label else 1
            for nj in (8, 6):
                tot[nj] += reps * 2.0 * m * n * k / (statistics.median(res[nj]) * 1e12)
        print(line, flush=True)
    lib.md_gemm_set_tuning(b"w4_nj", 0)
    lib.md_gemm_set_tuning(b"tile", -1)
    print(f"ViT proj + fc2 (x27) + patch_emb per B=64 step: 256-wide {tot[8] * 1e3:.2f} ms, 192-wide {tot[6] * 1e3:.2f} ms")


main()
