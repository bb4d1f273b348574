import ctypes as C, math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from moondream_amd import _lib
from moondream_amd.weights import PackedLinearInt4, dequantize_int4
from util import quantize_int4
lib = _lib.load(); BF16 = torch.bfloat16
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (m, k, n) in [(1, 128, 64), (1, 256, 64), (1, 128, 128), (4, 512, 64)]:
    g = torch.Generator().manual_seed(1)
    w = torch.randn(n, k, generator=g) / math.sqrt(k)
    packed, scale, zero = quantize_int4(w)
    deq = dequantize_int4(packed, scale, zero, n).cuda()
    q = PackedLinearInt4([(packed, scale, zero, n)], None, "cuda")
    for probe in range(min(k, 4)):
        a = torch.zeros(m, k, dtype=BF16, device="cuda"); a[:, probe * 37 % k] = 1.0
        c = torch.full((m, q.n_pad), float("nan"), dtype=BF16, device="cuda")
        s = q.struct()
        _lib.check(lib.md_gemm_fp8w(a.data_ptr(), a.stride(0), C.byref(s), c.data_ptr(), c.stride(0), m, 0, 1, 0, st()))
        torch.cuda.synchronize()
        col = probe * 37 % k
        want = deq[:, col].float()
        got = c[0, :n].float()
        ok = torch.equal(got, want)
        print(f"m{m} k{k} n{n} unit feature {col}: equal={ok}", flush=True)
        if not ok:
            # which column of deq does the output match?
            d = (deq.float().t()[:, :] - got[None, :]).abs().sum(1)
            print("   best matching feature:", int(d.argmin()), float(d.min()), " got[:6]", got[:6].tolist(), " want[:6]", want[:6].tolist())
            # maybe a scaled/affine relation
            qm = torch.empty(2 * packed.shape[0], 128, dtype=torch.uint8); qm[:packed.shape[0]] = (packed & 0xF0) >> 4; qm[packed.shape[0]:] = packed & 0x0F
            print("   q[:6, col]", qm.reshape(n, k)[:6, col].tolist(), "scale", scale.reshape(n, -1)[:6, col // 128].tolist(), "zero", zero.reshape(n, -1)[:6, col // 128].tolist())
