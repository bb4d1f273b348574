#!/usr/bin/env python3
"""Interleaved A/B of the big-tile GEMM configs on the 2B model's layer shapes (B = 64 images per
GPU, ViT in groups of 128 crops) and on square problems, random operands.  Every config is timed
`rounds` times in ONE process, round-robin, and the median is printed (run-to-run noise of a single
timing is ~3 %, cdna guide 5.4 rule 24).

    python tools/sweep_gemm.py [tiles=20,11,15] [rounds=3] [zero=0] [epi=2] [group_m=8]
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

lib = _lib.load()
BF16 = torch.bfloat16
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
TILES = [int(t) for t in opts.get("tiles", "20,11,15").split(",")]
ROUNDS = int(opts.get("rounds", "3"))
ZERO = opts.get("zero", "0") == "1"
ONLY_EPI = int(opts["epi"]) if "epi" in opts else None  # epi=2: residual layers only
if "group_m" in opts:  # row panels per tile-order group (the XCD-level blocking of the persistent tile sequence)
    lib.md_gemm_set_tuning(b"group_m", int(opts["group_m"]))

# (m, k, n, epilogue, label): the layers of one B=64 step
SHAPES = [
    (93312, 1152, 3456, 0, "vit qkv"), (93312, 1152, 1152, 2, "vit proj"), (93312, 1152, 4304, 1, "vit fc1"),
    (93312, 4304, 1152, 2, "vit fc2"), (93312, 588, 1152, 2, "vit patch_emb"),
    (46656, 2304, 8192, 1, "proj fc1"), (46656, 8192, 2048, 0, "proj fc2"),
    (46720, 2048, 14336, 1, "text qkv|fc1"), (46720, 2048, 2048, 2, "text proj"), (46720, 8192, 2048, 2, "text fc2"),
    (4096, 4096, 4096, 0, "4096^3"), (8192, 8192, 8192, 0, "8192^3"),
]


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=6, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    total = {t: [0.0, 0.0] for t in TILES}
    for m, k, n, epi, label in SHAPES:
        if ONLY_EPI is not None and epi != ONLY_EPI:
            continue
        kp = (k + 63) // 64 * 64
        a = (torch.randn(m, kp, device="cuda") * 0.5).to(BF16)
        if kp > k:
            a[:, k:] = 0
        w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).to(BF16)
        if ZERO:
            a.zero_(); w.zero_()
        lin = PackedLinear(w, torch.zeros(n, dtype=BF16), "cuda")
        c = torch.empty(m, lin.n_pad, dtype=BF16, device="cuda")
        r = torch.randn(m, lin.n_pad, device="cuda").to(BF16)
        args = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), lin.struct(), c.data_ptr(), c.stride(0), r.data_ptr(), r.stride(0), 0, m,
                               epi, 1 if epi == 1 else 0, 0, None, 0)
        res = {t: [] for t in TILES}
        for _ in range(ROUNDS):
            for t in TILES:
                lib.md_gemm_set_tuning(b"tile", t)
                dt = timeit(lambda: _lib.check(lib.md_gemm_bf16(C.byref(args), stream())))
                res[t].append(2.0 * m * n * k / dt / 1e12)
        lib.md_gemm_set_tuning(b"tile", -1)
        line = f"{label:14s} m={m:6d} k={k:5d} n={n:5d} epi={epi}: "
        for t in TILES:
            med = statistics.median(res[t])
            line += f" tile{t}: {med:7.1f} TF/s ({min(res[t]):6.0f}..{max(res[t]):6.0f})"
            if label[0] in "vpt":
                # per-step weights: ViT layers x27, text x24
                mult = 27 if label.startswith("vit") and "patch" not in label else (24 if label.startswith("text") else 1)
                total[t][0] += mult * 2.0 * m * n * k
                total[t][1] += mult * 2.0 * m * n * k / (med * 1e12)
        print(line, flush=True)
        del a, w, c, r, lin
    for t in TILES:
        fl, tm = total[t]
        print(f"model-weighted (one B=64 step's tile GEMMs) tile{t}: {fl/tm/1e12:7.1f} TF/s, {tm*1e3:6.1f} ms", flush=True)


if __name__ == "__main__":
    main()
