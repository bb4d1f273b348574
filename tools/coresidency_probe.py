#!/usr/bin/env python3
"""Round 5: can a decode-attention launch run ON THE SAME CUs as a persistent four-wave tile-GEMM workgroup?  One long GEMM
(46720 x 2048 x 8192, ~1.1 ms, 256 persistent workgroups = every CU, 448 registers per lane and 145 KiB of LDS each) on stream E,
then ONE decode-attention launch (64 sequences x 32 heads, 770 keys; ~60 us alone) on a second, high-priority stream D; events on D
around the attention launch tell when it finished relative to the GEMM.  MD_DECODE_ATTN_SMALL=1 selects the small-footprint shape
(64 registers, 12.4 KiB LDS: what a GEMM workgroup leaves free on its CU)."""
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
gemm = lambda: _lib.check(lib.md_gemm_bf16(C.byref(ga), C.c_void_p(E.cuda_stream)))
attn = lambda: _lib.check(lib.md_attention_decode(q.data_ptr(), q.stride(0), o.data_ptr(), h * 64, kk.data_ptr(), vv.data_ptr(), h * ctx * 64, ctx, lens.data_ptr(), b, h, h, 64, 0.125, C.c_void_p(D.cuda_stream)))
for _ in range(3): gemm(); attn()
torch.cuda.synchronize()
ev = lambda: torch.cuda.Event(enable_timing=True)
def once(n_attn):
    g0, g1, a0, a1 = ev(), ev(), ev(), ev()
    torch.cuda.synchronize()
    g0.record(E); gemm(); gemm(); g1.record(E)          # ~2.2 ms of GEMM
    time.sleep(0.0003)                                    # the GEMM is running
    a0.record(D)
    for _ in range(n_attn): attn()
    a1.record(D)
    torch.cuda.synchronize()
    return g0.elapsed_time(g1), g0.elapsed_time(a0), g0.elapsed_time(a1)
a_alone0, a_alone1 = ev(), ev()
a_alone0.record(D)
for _ in range(10): attn()
a_alone1.record(D); torch.cuda.synchronize()
print(f"MD_DECODE_ATTN_SMALL={os.environ.get('MD_DECODE_ATTN_SMALL', '0')}: attention alone {a_alone0.elapsed_time(a_alone1) / 10 * 1e3:.1f} us per launch")
for n_attn in (1, 10):
    for rep in range(3):
        tg, ta0, ta1 = once(n_attn)
        print(f"  2 GEMMs {tg:.3f} ms | {n_attn} attention launch(es) enqueued at +{ta0:.3f} ms, finished at +{ta1:.3f} ms"
              f"  -> {'INSIDE the GEMMs' if ta1 < tg - 0.05 else 'only after a GEMM launch ended'}", flush=True)
