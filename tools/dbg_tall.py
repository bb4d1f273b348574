import ctypes as C, math, os, sys
sys.path.insert(0, "/root/repo")
import torch
from moondream_amd import _lib
from moondream_amd.weights import PackedLinear
lib = _lib.load(); BF16 = torch.bfloat16
def run(a, lin, m, policy, epi, gelu_from, store_pad):
    out = torch.zeros(m, lin.n_pad if store_pad else lin.n, dtype=BF16, device="cuda")
    g = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), lin.struct(), out.data_ptr(), out.stride(0), None, 0, 0, m, epi, store_pad, gelu_from, None, 0, policy)
    _lib.check(lib.md_gemm_bf16(C.byref(g), C.c_void_p(torch.cuda.current_stream().cuda_stream))); torch.cuda.synchronize()
    return out
for (n, k, gf) in ((1472, 256, 768), (14336, 2048, 6144), (1024, 256, 0), (51200, 2048, 0)):
    torch.manual_seed(0)
    lin = PackedLinear((torch.randn(n, k, device="cuda") / math.sqrt(k)).to(BF16), (torch.randn(n, device="cuda") * 0.1).to(BF16), "cuda")
    a = (torch.randn(128, lin.k_pad, device="cuda")).to(BF16)
    epi = 1 if gf else 0
    tall = run(a, lin, 128, 2, epi, gf, 1 if gf else 0)
    lo = run(a[:64].contiguous(), lin, 64, 1, epi, gf, 1 if gf else 0)
    hi = run(a[64:].contiguous(), lin, 64, 1, epi, gf, 1 if gf else 0)
    print((n, k, gf), "rows 0-63 equal:", torch.equal(tall[:64], lo), "rows 64-127 equal:", torch.equal(tall[64:], hi),
          "max diff", float((tall[:64].float() - lo.float()).abs().max()), float((tall[64:].float() - hi.float()).abs().max()))
