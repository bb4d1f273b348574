#!/usr/bin/env python3
"""Single-image regime (m = 729 / 730 rows): microseconds per layer GEMM for each tile config, interleaved.

    python tools/sweep_gemm_b1.py [tiles=-1,2,1,10,16,3] [rounds=3]
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
TILES = [int(t) for t in opts.get("tiles", "-1,2,1,10,16,3").split(",")]
ROUNDS = int(opts.get("rounds", "3"))
SHAPES = [
    (729, 1152, 3456, 0, "vit qkv"), (729, 1152, 1152, 2, "vit proj"), (729, 1152, 4304, 1, "vit fc1"),
    (729, 4304, 1152, 2, "vit fc2"), (729, 2304, 8192, 1, "proj fc1"), (729, 8192, 2048, 0, "proj fc2"),
    (730, 2048, 14336, 1, "text qkv|fc1"), (730, 2048, 2048, 2, "text proj"), (730, 8192, 2048, 2, "text fc2"),
    (11, 2048, 14336, 1, "prompt qkv|fc1"),
]


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for m, k, n, epi, label in SHAPES:
    kp = (k + 63) // 64 * 64
    a = (torch.randn(m, kp, device="cuda") * 0.5).to(BF16)
    if kp > k:
        a[:, k:] = 0
    w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).to(BF16)
    lin = PackedLinear(w, torch.zeros(n, dtype=BF16), "cuda")
    c = torch.empty(m, lin.n_pad, dtype=BF16, device="cuda")
    r = torch.randn(m, lin.n_pad, device="cuda").to(BF16)
    args = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), lin.struct(), c.data_ptr(), c.stride(0), r.data_ptr(), r.stride(0), 0, m,
                           epi, 1 if epi == 1 else 0, 0, None, 0)
    res = {t: [] for t in TILES}
    outs = {}
    for _ in range(ROUNDS):
        for t in TILES:
            lib.md_gemm_set_tuning(b"tile", t)
            res[t].append(timeit(lambda: _lib.check(lib.md_gemm_bf16(C.byref(args), stream()))))
            outs[t] = c.clone()
    lib.md_gemm_set_tuning(b"tile", -1)
    same = all(torch.equal(outs[TILES[0]], outs[t]) for t in TILES)
    line = f"{label:15s} m={m:4d} k={k:5d} n={n:5d} epi={epi}:"
    for t in TILES:
        line += f"  tile{t}: {statistics.median(res[t]):6.1f}us"
    print(line + ("  [bitwise equal]" if same else "  [DIFFER]"), flush=True)
