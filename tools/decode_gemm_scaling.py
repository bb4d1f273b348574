#!/usr/bin/env python3
"""Round 5: where a decode-regime launch's time goes, by scaling K and N (64 rows, GELU-from-column epilogue like the fused qkv|fc1
layer): a hipGraph of 24 launches over 24 different weight matrices (no Infinity-Cache reuse), us per launch with the graph's
launch gap inside.  time(K) at fixed N: slope = streaming rate, intercept = per-launch fixed cost; time(N) at fixed K: the
workgroup count crosses one per CU at N = 16384."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moondream_amd import _lib
from moondream_amd.weights import PackedLinear
lib = _lib.load(); BF16 = torch.bfloat16
m, L = 64, 24
def timed(n, k, epi=1):
    ws = [PackedLinear((torch.randn(n, k, device="cuda") / math.sqrt(k)).to(BF16), torch.zeros(n, dtype=BF16), "cuda") for _ in range(L)]
    a = (torch.randn(m, k, device="cuda") * 0.5).to(BF16)
    out = torch.empty(m, n, dtype=BF16, device="cuda")
    st0 = ws[0].struct()
    need = lib.md_gemm_workspace_bytes(C.byref(st0), m, 0)
    wsb = torch.zeros(max(need, 16), dtype=torch.uint8, device="cuda")
    def run(st):
        for l in ws:
            g = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), l.struct(), out.data_ptr(), out.stride(0), None, 0, 0, m, epi, 0, 0, wsb.data_ptr(), need)
            _lib.check(lib.md_gemm_bf16(C.byref(g), C.c_void_p(st)))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s): run(s.cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): run(torch.cuda.current_stream().cuda_stream)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(4): g.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (4 * L) * 1e3)
    return best
cfgs = [int(c) for c in sys.argv[1:]] or [16]
for c in cfgs:
    _lib.check(lib.md_gemm_set_tuning(b"decode_cfg", c))
    _lib.check(lib.md_gemm_set_tuning(b"decode_slices", 1))   # no in-launch split-K: one workgroup per 64-column tile, whole K
    print(f"decode_cfg {c}: N = 14336 (224 workgroups), K sweep")
    for k in (256, 512, 1024, 2048, 4096, 8192):
        t = timed(14336, k)
        print(f"  K={k:5d}  {2*14336*k/1e6:6.1f} MB  {t:6.1f} us  {2*14336*k/t/1e6:5.2f} TB/s", flush=True)
    print(f"decode_cfg {c}: K = 2048, N sweep (workgroups = N / 64)")
    for n in (1792, 3584, 7168, 14336, 16384, 28672, 32768, 57344):
        t = timed(n, 2048)
        print(f"  N={n:6d} ({n//64:4d} wg)  {2*n*2048/1e6:6.1f} MB  {t:6.1f} us  {2*n*2048/t/1e6:5.2f} TB/s", flush=True)
_lib.check(lib.md_gemm_set_tuning(b"decode_slices", 0))
