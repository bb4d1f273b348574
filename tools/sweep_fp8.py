#!/usr/bin/env python3
"""Decode-regime layer GEMMs: bf16 weight stream vs the FP8 (e4m3, fragment-ordered) stream, interleaved."""
import ctypes as C
import math
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from moondream_amd import _lib
from moondream_amd.weights import PackedLinear, PackedLinearFp8

lib = _lib.load()
BF16 = torch.bfloat16


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=24, warm=3, reps=5):
    """Microseconds per launch with the launches replayed from a hipGraph: a 10 us kernel launched through
    ctypes from Python is otherwise timed at the host's issue rate, not the GPU's."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (iters * reps) * 1e3


# a ring of distinct weight copies larger than the 256 MB Infinity Cache, so every launch streams from HBM
def ring(make, bytes_each):
    n = max(2, int(600e6 // bytes_each) + 1)
    return [make(i) for i in range(n)]


for m in (1, 8, 64):
    for k, n, epi, gf, label in ((2048, 14336, 1, 6144, "qkv|fc1"), (2048, 51200, 0, 0, "lm_head")):
        a = (torch.randn(m, k, device="cuda") * 0.5).to(BF16)
        lins = ring(lambda i: PackedLinear((torch.randn(n, k, device="cuda") / math.sqrt(k)).to(BF16), torch.zeros(n, dtype=BF16), "cuda"), n * k * 2)
        q8 = [PackedLinearFp8(l.w, l.b, n, k) for l in lins]
        c = torch.empty(m, lins[0].n_pad, dtype=BF16, device="cuda")
        ws = torch.zeros(lib.md_gemm_workspace_bytes(C.byref(lins[0].struct()), m, 1) + 16, dtype=torch.uint8, device="cuda")
        state = {"i": 0}

        def run_bf16():
            l = lins[state["i"] % len(lins)]; state["i"] += 1
            args = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), l.struct(), c.data_ptr(), c.stride(0), None, 0, 0, m, epi, 1, gf, ws.data_ptr(), ws.numel())
            _lib.check(lib.md_gemm_bf16(C.byref(args), stream()))

        def run_fp8():
            l = q8[state["i"] % len(q8)]; state["i"] += 1
            st = l.struct()
            _lib.check(lib.md_gemm_fp8w(a.data_ptr(), a.stride(0), C.byref(st), c.data_ptr(), c.stride(0), m, epi, 1, gf, stream()))

        tb, t8 = [], []
        for _ in range(3):
            tb.append(timeit(run_bf16)); t8.append(timeit(run_fp8))
        tb, t8 = statistics.median(tb), statistics.median(t8)
        print(f"m={m:2d} {label:8s} k={k} n={n}: bf16 {tb:6.1f}us ({n * k * 2 / tb / 1e6:5.2f} TB/s)  fp8 {t8:6.1f}us ({n * k / t8 / 1e6:5.2f} TB/s)  x{tb / t8:4.2f}", flush=True)
        del lins, q8
        torch.cuda.empty_cache()


# the block tail's two K-sliced layers in one launch: proj (2048 -> 2048) + fc2 (8192 -> 2048)
for m in (1, 8, 64):
    dim = 2048
    a1 = (torch.randn(m, 2048, device="cuda") * 0.5).to(BF16)
    a2 = (torch.randn(m, 8192, device="cuda") * 0.5).to(BF16)
    nring = 8
    pj = [PackedLinear((torch.randn(dim, 2048, device="cuda") / 45).to(BF16), torch.zeros(dim, dtype=BF16), "cuda") for _ in range(nring)]
    f2 = [PackedLinear((torch.randn(dim, 8192, device="cuda") / 90).to(BF16), torch.zeros(dim, dtype=BF16), "cuda") for _ in range(nring)]
    pj8 = [PackedLinearFp8(l.w, l.b, dim, 2048) for l in pj]
    f28 = [PackedLinearFp8(l.w, l.b, dim, 8192) for l in f2]
    pa = torch.empty(8, m, dim, dtype=torch.float32, device="cuda")
    pb = torch.empty(8, m, dim, dtype=torch.float32, device="cuda")
    state = {"i": 0}

    def run_bf16():
        i = state["i"] % nring; state["i"] += 1
        sa, sb = pj[i].struct(), f2[i].struct()
        _lib.check(lib.md_gemm_partial_f32_pair(a1.data_ptr(), a1.stride(0), C.byref(sa), pa.data_ptr(), a2.data_ptr(), a2.stride(0),
                                                C.byref(sb), pb.data_ptr(), m, dim, m * dim, stream()))

    def run_fp8():
        i = state["i"] % nring; state["i"] += 1
        sa, sb = pj8[i].struct(), f28[i].struct()
        _lib.check(lib.md_gemm_fp8w_partial_f32_pair(a1.data_ptr(), a1.stride(0), C.byref(sa), pa.data_ptr(), a2.data_ptr(), a2.stride(0),
                                                     C.byref(sb), pb.data_ptr(), m, dim, m * dim, stream()))

    tb, t8 = [], []
    for _ in range(3):
        tb.append(timeit(run_bf16)); t8.append(timeit(run_fp8))
    tb, t8 = statistics.median(tb), statistics.median(t8)
    by = dim * (2048 + 8192)
    print(f"m={m:2d} proj+fc2 partial pair: bf16 {tb:6.1f}us ({by * 2 / tb / 1e6:5.2f} TB/s)  fp8 {t8:6.1f}us ({by / t8 / 1e6:5.2f} TB/s)  x{tb / t8:4.2f}", flush=True)
