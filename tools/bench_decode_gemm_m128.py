#!/usr/bin/env python3
"""Round 6 probe: a 128-row decode-regime tile (config 21: 128 x 64, four compute waves + two DMA helpers) against what the library does
today at 64 and at 128 rows, on the decoder's fused qkv|fc1 layer and lm_head (hipGraph of 24 launches over 24 different weight sets)."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moondream_amd import _lib
from moondream_amd.weights import PackedLinear
lib = _lib.load(); BF16 = torch.bfloat16
D, FF, V = 2048, 8192, 51200
mk = lambda n, k: PackedLinear((torch.randn(n, k, device="cuda") / math.sqrt(k)).to(BF16), torch.zeros(n, dtype=BF16), "cuda")
L = 12
layers = {"fused qkv|fc1 14336x2048": [mk(3 * D + FF, D) for _ in range(L)], "lm_head 51200x2048": [mk(V, D) for _ in range(3)]}
def timed(fn, n):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s): fn(s.cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn(torch.cuda.current_stream().cuda_stream)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(4): g.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (4 * n) * 1e3)
    return best
ref = {}
for rep in range(2):
    for name, ws in layers.items():
        for m, tile in ((64, -1), (128, -1), (128, 21), (96, 21)):
            a = (torch.randn(m, D, device="cuda") * 0.5).to(BF16)
            out = torch.empty(m, ws[0].n, dtype=BF16, device="cuda")
            _lib.check(lib.md_gemm_set_tuning(b"tile", tile))
            def run(st):
                for l in ws:
                    g = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), l.struct(), out.data_ptr(), out.stride(0), None, 0, 0, m, 0, 0, 0, None, 0, 0)
                    _lib.check(lib.md_gemm_bf16(C.byref(g), C.c_void_p(st)))
            t = timed(run, len(ws))
            if rep == 0:
                run(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
                key = (name, m)
                if key in ref: print("   bit-identical to the default path:", torch.equal(ref[key], out))
                else: ref[key] = out.clone()
            print(f"{name:28s} m={m:4d} tile={'default' if tile < 0 else tile}: {t:7.1f} us per launch = {t / m:6.3f} us per row", flush=True)
_lib.check(lib.md_gemm_set_tuning(b"tile", -1))
# ---- proj + fc2 K-slice partial pair (launch-boundary split-K), 64 vs 128 rows
pr, f2 = [mk(D, D) for _ in range(L)], [mk(D, FF) for _ in range(L)]
sa, sb = pr[0].struct(), f2[0].struct()
na, nb = lib.md_gemm_partial_slices(C.byref(sa)), lib.md_gemm_partial_slices(C.byref(sb))
outs = {}
for rep in range(2):
    for m in (64, 128):
        act = (torch.randn(m, 3 * D + FF, device="cuda") * 0.5).to(BF16)
        pa = torch.zeros(na, m, D, dtype=torch.float32, device="cuda"); pb = torch.zeros(nb, m, D, dtype=torch.float32, device="cuda")
        def run(st):
            for p1, p2 in zip(pr, f2):
                s1, s2 = p1.struct(), p2.struct()
                _lib.check(lib.md_gemm_partial_f32_pair(act.data_ptr(), act.stride(0), C.byref(s1), pa.data_ptr(), act.data_ptr() + 3 * D * 2, act.stride(0),
                                                        C.byref(s2), pb.data_ptr(), m, D, m * D, C.c_void_p(st)))
        t = timed(run, L)
        print(f"proj + fc2 partial pair      m={m:4d}: {t:7.1f} us per launch = {t / m:6.3f} us per row", flush=True)
        if rep == 0 and m == 128:   # rows 0..63 of the tall tile == the 64-row config on the same rows (last layer's weights)
            a64 = act[:64].contiguous()
            qa = torch.zeros(na, 64, D, dtype=torch.float32, device="cuda"); qb = torch.zeros(nb, 64, D, dtype=torch.float32, device="cuda")
            s1, s2 = pr[-1].struct(), f2[-1].struct()
            _lib.check(lib.md_gemm_partial_f32_pair(a64.data_ptr(), a64.stride(0), C.byref(s1), qa.data_ptr(), a64.data_ptr() + 3 * D * 2, a64.stride(0),
                                                    C.byref(s2), qb.data_ptr(), 64, D, 64 * D, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.synchronize()
            print("   rows 0..63 of the 128-row launch bit-identical to a 64-row launch:", torch.equal(pa[:, :64], qa), torch.equal(pb[:, :64], qb))
