#!/usr/bin/env python3
"""The decode step's two GEMM launches as the decode graph runs them: a hipGraph of 24 x (fused qkv|fc1 GEMM, proj+fc2 partial pair)
over 24 layers' weights (no Infinity-Cache reuse), 64 rows; us per launch with the graph's launch gaps inside.  MD_HIP_LIB selects the build."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moondream_amd import _lib
from moondream_amd.weights import PackedLinear
lib = _lib.load(); BF16 = torch.bfloat16
m, D, FF = 64, 2048, 8192
mk = lambda n, k: PackedLinear((torch.randn(n, k, device="cuda") / math.sqrt(k)).to(BF16), torch.zeros(n, dtype=BF16), "cuda")
L = 24
qf, pr, f2 = [mk(3 * D + FF, D) for _ in range(L)], [mk(D, D) for _ in range(L)], [mk(D, FF) for _ in range(L)]
a = (torch.randn(m, D, device="cuda") * 0.5).to(BF16)
act = torch.empty(m, 3 * D + FF, dtype=BF16, device="cuda")
sa, sb = pr[0].struct(), f2[0].struct()
na, nb = lib.md_gemm_partial_slices(C.byref(sa)), lib.md_gemm_partial_slices(C.byref(sb))
pa = torch.empty(na, m, D, dtype=torch.float32, device="cuda"); pb = torch.empty(nb, m, D, dtype=torch.float32, device="cuda")
def run_a(st):
    for l in qf:
        g = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), l.struct(), act.data_ptr(), act.stride(0), None, 0, 0, m, 1, 0, 3 * D, None, 0)
        _lib.check(lib.md_gemm_bf16(C.byref(g), C.c_void_p(st)))
def run_b(st):
    for p1, p2 in zip(pr, f2):
        s1, s2 = p1.struct(), p2.struct()
        _lib.check(lib.md_gemm_partial_f32_pair(act.data_ptr(), act.stride(0), C.byref(s1), pa.data_ptr(), act.data_ptr() + 3 * D * 2, act.stride(0),
                                                C.byref(s2), pb.data_ptr(), m, D, m * D, C.c_void_p(st)))
def timed(fn):
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
        best = min(best, e0.elapsed_time(e1) / (4 * L) * 1e3)
    return best
cfgs = [int(c) for c in sys.argv[1:]] or [None]   # decode_cfg values to compare in ONE process (md_gemm_set_tuning), interleaved twice
for rep in range(2 if len(cfgs) > 1 else 1):
    for c in cfgs:
        if c is not None: _lib.check(lib.md_gemm_set_tuning(b"decode_cfg", c))
        for name, fn in (("fused qkv|fc1 (58.7 MB)", run_a), ("proj + fc2 partial pair (41.9 MB)", run_b)):
            print(f"{os.path.basename(os.environ.get('MD_HIP_LIB', 'in-tree'))} decode_cfg={c}: {name:36s} {timed(fn):6.1f} us per launch", flush=True)
