#!/usr/bin/env python3
"""Do kernels from two HIP streams actually run concurrently on this box?
A: big tile GEMMs on stream E; B: decode-attention launches on stream D; wall time alone vs together."""
import ctypes as C, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moondream_amd import _lib
from moondream_amd.weights import PackedLinear
lib = _lib.load(); BF16 = torch.bfloat16
m, k, n = 46720, 2048, 8192
a = (torch.randn(m, k, device="cuda") * 0.5).to(BF16)
lin = PackedLinear((torch.randn(n, k, device="cuda") / math.sqrt(k)).to(BF16), torch.zeros(n, dtype=BF16), "cuda")
c = torch.empty(m, lin.n_pad, dtype=BF16, device="cuda")
ga = _lib.MdGemmArgs(a.data_ptr(), k, lin.struct(), c.data_ptr(), lin.n_pad, None, 0, 0, m, 0, 0, 0, None, 0)
b, h, ctx = 64, 32, 2048
q = torch.randn(b, 3 * h * 64, device="cuda").to(BF16); o = torch.empty(b, h * 64, dtype=BF16, device="cuda")
kk = torch.randn(b, h, ctx, 64, device="cuda").to(BF16); vv = torch.randn(b, h, ctx, 64, device="cuda").to(BF16)
lens = torch.full((b,), 770, dtype=torch.int32, device="cuda")
E, D = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
def run_a(st, reps):
    for _ in range(reps): _lib.check(lib.md_gemm_bf16(C.byref(ga), C.c_void_p(st.cuda_stream)))
def run_b(st, reps):
    for _ in range(reps):
        _lib.check(lib.md_attention_decode(q.data_ptr(), q.stride(0), o.data_ptr(), h * 64, kk.data_ptr(), vv.data_ptr(), h * ctx * 64, ctx, lens.data_ptr(), b, h, h, 64, 0.125, C.c_void_p(st.cuda_stream)))
def wall(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
RA, RB = 60, 1200
GRIDS = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"])]
run_a(E, 3); run_b(D, 10)
for grid in GRIDS:
    lib.md_gemm_set_tuning(b"w4_grid", grid)
    print(f"-- four-wave GEMM grid = {grid or 'one per CU'}")
    ta = wall(lambda: run_a(E, RA)); tb = wall(lambda: run_b(D, RB))
    def both():
        run_b(D, RB); run_a(E, RA)
    tab = wall(both)
    print(f"GEMM stream alone {ta:.1f} ms | decode-attn stream alone {tb:.1f} ms | both streams {tab:.1f} ms (serial sum {ta+tb:.1f}, perfect overlap {max(ta,tb):.1f})")
    # same with a hipGraph of the decode-like work (graph launch on D)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(D):
        run_b(D, 5)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        for _ in range(100):
            _lib.check(lib.md_attention_decode(q.data_ptr(), q.stride(0), o.data_ptr(), h * 64, kk.data_ptr(), vv.data_ptr(), h * ctx * 64, ctx, lens.data_ptr(), b, h, h, 64, 0.125, C.c_void_p(cur.cuda_stream)))
    def gboth():
        with torch.cuda.stream(D):
            for _ in range(RB // 100): g.replay()
        run_a(E, RA)
    tg = wall(lambda: [g.replay() for _ in range(RB // 100)])
    tgb = wall(gboth)
    print(f"graph(decode-attn) alone {tg:.1f} ms | graph on D + GEMMs on E {tgb:.1f} ms")
