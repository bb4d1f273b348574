#!/usr/bin/env python3
"""Single-image regime (1458 ViT rows = 2 crops, 735 decoder rows): the layer GEMMs on the shipped tile choice (-1) against every
tile config forced in turn: 16 = 64 x 64 decode-regime config (the choice for the few-tile layers), 2 = 128 x 128 two-stage
ring, 1 = 256 x 128, 20 = the four-wave 256 x 256 kernel.  Microseconds per launch, interleaved
rounds, error against fp32, and whether the 32x32x16 configs agree bit for bit.

    python tools/sweep_gemm_b1_configs.py [rounds=3]

(The round-5 experiment this file was written for also had big tiles with the in-launch deterministic split-K, S = 2..8, and five
more ring / wave shapes under experimental tile codes; the record is profiles/r05_b1_tile_config_sweep.txt, the codes are gone.
A back-to-back sweep keeps the layer's weights in the caches: it is optimistic for configs of few workgroups, which in a real caption
pull cold weights through few CUs -- tools/gpu_r5_b1_rule_trace.sh measures the launches inside a caption.)
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
ROUNDS = int(opts.get("rounds", "3"))
SHAPES = [
    (1458, 1152, 1152, 2, "vit proj"), (1458, 4304, 1152, 2, "vit fc2"), (1458, 1152, 3456, 0, "vit qkv"), (1458, 1152, 4304, 1, "vit fc1"),
    (735, 2048, 2048, 2, "text proj"), (735, 8192, 2048, 2, "text fc2"), (735, 2048, 14336, 1, "text qkv|fc1"),
    (729, 2304, 8192, 1, "proj fc1"), (729, 8192, 2048, 0, "proj fc2"),
]
CONFIGS = [(-1, 1)] + [(t, 1) for t in (16, 2, 1, 20)]


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


ws = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda")
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
                           epi, 1 if epi == 1 else 0, 0, ws.data_ptr(), ws.numel())
    ref = a[:, :k].float() @ w.float().t()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref.to(BF16).float(), approximate="tanh")
    if epi == 2:
        ref = ref.to(BF16).float() + r[:, :n].float()
    res = {cfg: [] for cfg in CONFIGS}
    err, outs = {}, {}
    for _ in range(ROUNDS):
        for cfg in CONFIGS:
            t, s = cfg
            lib.md_gemm_set_tuning(b"tile", t)
            try:
                res[cfg].append(timeit(lambda: _lib.check(lib.md_gemm_bf16(C.byref(args), stream()))))
                err[cfg] = float((c[:, :n].float() - ref).norm() / ref.norm())
                outs[cfg] = c.clone()
            except Exception as ex:  # a config that does not apply to this shape
                res[cfg].append(float("nan"))
                err[cfg] = float("nan")
    lib.md_gemm_set_tuning(b"tile", -1)
    same = all(torch.equal(outs[(16, 1)], outs[(t, 1)]) for t in (2, 1))  # (the shipped choice is one of them, or tile 20)
    base = statistics.median(res[(-1, 1)])
    gf = 2.0 * m * n * k / 1e9
    print(f"{label:13s} m={m} k={k} n={n} epi={epi}: shipped {base:6.1f} us = {gf / base:5.3f} PF/s   32x32x16 configs {'agree bit for bit' if same else 'DIFFER'}", flush=True)
    for cfg in CONFIGS[1:]:
        med = statistics.median(res[cfg])
        flag = " <==" if med < 0.9 * base else ""
        print(f"     tile {cfg[0]:3d} S={cfg[1]}: {med:6.1f} us  rel-err {err[cfg]:.2e}{flag}", flush=True)
