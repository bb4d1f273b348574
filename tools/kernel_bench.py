#!/usr/bin/env python3
"""On-GPU microbenchmarks of the hot kernels at the shapes the 2B model uses
(B = 64 images/GPU, ViT in chunks of 32 crops).  Prints TFLOP/s per shape and
tile config; used to choose tile configs and to track kernel-level progress."""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from moondream_amd import _lib
from moondream_amd.weights import PackedLinear

BF16 = torch.bfloat16
lib = _lib.load()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=10, warm=3):
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


def bench_gemm(m, k, n, epi=0, tiles=("20", "11", "1", "2")):
    a = (torch.randn(m, (k + 63) // 64 * 64, device="cuda") * 0.5).to(BF16)
    w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).to(BF16)
    lin = PackedLinear(w, torch.zeros(n, dtype=BF16), "cuda")
    if lin.k_pad > k:
        a[:, k:] = 0
    c = torch.empty(m, lin.n_pad, dtype=BF16, device="cuda")
    r = torch.randn(m, lin.n_pad, device="cuda").to(BF16)
    st = lin.struct()
    need = lib.md_gemm_workspace_bytes(C.byref(st), m, 0)
    ws = torch.zeros(max(need, 16), dtype=torch.uint8, device="cuda")
    args = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), st, c.data_ptr(), c.stride(0), r.data_ptr(), r.stride(0), 0, m, epi, 0, 0, ws.data_ptr(), need)
    res = []
    for t in tiles:
        lib.md_gemm_set_tuning(b"tile", int(t))
        dt = timeit(lambda: _lib.check(lib.md_gemm_bf16(C.byref(args), stream())))
        res.append(2.0 * m * n * k / dt / 1e12)
    lib.md_gemm_set_tuning(b"tile", -1)
    dt = timeit(lambda: _lib.check(lib.md_gemm_bf16(C.byref(args), stream())))
    auto = 2.0 * m * n * k / dt / 1e12
    extra = f"  ({2.0*n*k/dt/1e12:5.2f} TB/s weights, {dt*1e6:6.1f} us)" if m <= 64 else ""
    print(f"gemm m={m:6d} k={k:5d} n={n:5d} epi={epi}: " + " ".join(f"tile{t}={x:7.1f}" for t, x in zip(tiles, res)) + f"  auto={auto:7.1f} TF/s" + extra, flush=True)


def bench_attn(b, h, t, hd, prefix=None):
    q = torch.randn(b, t, 3 * h * hd, device="cuda").to(BF16)
    o = torch.empty(b, t, h * hd, dtype=BF16, device="cuda")
    a = _lib.MdAttnArgs()
    d3 = 3 * h * hd
    a.q, a.q_bs, a.q_ts, a.q_hs = q.data_ptr(), t * d3, d3, hd
    a.k, a.k_bs, a.k_ts, a.k_hs = q.data_ptr() + h * hd * 2, t * d3, d3, hd
    a.v, a.v_bs, a.v_ts, a.v_hs = q.data_ptr() + 2 * h * hd * 2, t * d3, d3, hd
    a.o, a.o_bs, a.o_ts, a.o_hs = o.data_ptr(), t * h * hd, h * hd, hd
    a.batch, a.n_heads, a.n_kv_heads, a.head_dim = b, h, h, hd
    a.q_len, a.kv_len_all, a.q_pos0, a.kv_len = t, t, None, None
    a.prefix_len, a.scale = (t if prefix is None else prefix), 1 / math.sqrt(hd)
    dt = timeit(lambda: _lib.check(lib.md_attention_prefill(C.byref(a), stream())))
    fl = 4.0 * b * h * t * t * hd
    print(f"attn b={b} h={h} t={t} hd={hd}: {dt*1e3:8.3f} ms  {fl/dt/1e12:7.1f} TF/s (algorithmic)", flush=True)


def bench_decode_attn(b, h, ctx_used):
    hd, ctx = 64, 2048
    q = torch.randn(b, 3 * h * hd, device="cuda").to(BF16)
    o = torch.empty(b, h * hd, dtype=BF16, device="cuda")
    k = torch.randn(b, h, ctx, hd, device="cuda").to(BF16)
    v = torch.randn(b, h, ctx, hd, device="cuda").to(BF16)
    lens = torch.full((b,), ctx_used, dtype=torch.int32, device="cuda")
    dt = timeit(lambda: _lib.check(lib.md_attention_decode(q.data_ptr(), q.stride(0), o.data_ptr(), h * hd, k.data_ptr(), v.data_ptr(), h * ctx * hd, ctx, lens.data_ptr(), b, h, h, hd, 0.125, stream())))
    by = 2.0 * b * h * ctx_used * hd * 2
    print(f"decode-attn b={b} ctx={ctx_used}: {dt*1e6:8.1f} us  {by/dt/1e12:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "attn", "decode"]
    if "gemm" in which:
        Mv, Mp = 32 * 729, 64 * 730
        for (m, k, n, epi) in [
            (Mv, 588, 1152, 2), (Mv, 1152, 3456, 0), (Mv, 1152, 1152, 2), (Mv, 1152, 4304, 1), (Mv, 4304, 1152, 2),
            (Mp, 2048, 6144, 0), (Mp, 2048, 2048, 2), (Mp, 2048, 8192, 1), (Mp, 8192, 2048, 2),
            (64 * 729, 2304, 8192, 1), (64 * 729, 8192, 2048, 0),
            (8192, 8192, 8192, 0), (4096, 4096, 4096, 0),
            (64, 2048, 6144, 0), (64, 2048, 8192, 1), (64, 8192, 2048, 2), (64, 2048, 51200, 0), (1, 2048, 51200, 0),
        ]:
            bench_gemm(m, k, n, epi)
    if "attn" in which:
        bench_attn(32, 16, 729, 72)
        bench_attn(128, 16, 729, 72)
        bench_attn(64, 32, 730, 64, prefix=730)
    if "decode" in which:
        bench_decode_attn(64, 32, 770)
        bench_decode_attn(1, 32, 770)
