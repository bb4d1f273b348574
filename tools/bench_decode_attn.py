#!/usr/bin/env python3
"""Decode-step attention (RoPE + cache update fused, B = 64 x 32 heads, contexts 736..767) as the decode graph runs it:
a hipGraph of 48 launches over alternating caches, replayed; us per launch and TB/s of K / V bytes.
MD_ATTN_DECODE_VAR / MD_ATTN_DECODE_NT select kernel variants (one per process)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moondream_amd import _lib
lib = _lib.load(); BF16 = torch.bfloat16
b, h, ctx, hd = 64, 32, 2048, 64
NC = 3
caches = [(torch.randn(b, h, ctx, hd, device="cuda").to(BF16), torch.randn(b, h, ctx, hd, device="cuda").to(BF16)) for _ in range(NC)]
qkv = torch.randn(b, 3 * h * hd, device="cuda").to(BF16)
o = torch.empty(b, h * hd, dtype=BF16, device="cuda")
freqs = torch.randn(ctx, 16, 2, device="cuda")
lens = (torch.arange(b, device="cuda", dtype=torch.int32) % 32) + 737
def launch(i, st):
    k, v = caches[i % NC]
    _lib.check(lib.md_attention_decode_rope(qkv.data_ptr(), qkv.stride(0), o.data_ptr(), h * hd, freqs.data_ptr(), k.data_ptr(), v.data_ptr(),
                                            h * ctx * hd, ctx, lens.data_ptr(), b, h, hd, 32, 0.125, C.c_void_p(st)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for i in range(6): launch(i, s.cuda_stream)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream().cuda_stream
    for i in range(48): launch(i, cur)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(5):
    e0.record()
    for _ in range(4): g.replay()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / (4 * 48) * 1e3)
nbytes = float(lens.sum().item()) * h * hd * 2 * 2
print(f"var={os.environ.get('MD_ATTN_DECODE_VAR','0')} nt={os.environ.get('MD_ATTN_DECODE_NT','1')}: {best:6.1f} us per launch (graph replay, launch gaps included)  {nbytes / best / 1e6:5.2f} TB/s")
