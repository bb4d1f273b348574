#!/usr/bin/env python3
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moondream_amd import _lib
from moondream_amd.weights import PackedLinear
from tools.kernel_bench import timeit, stream
lib = _lib.load(); BF16 = torch.bfloat16
ZERO = False
def run(m,k,n,epi,env):
    a = (torch.randn(m, (k+63)//64*64, device="cuda")*0.5).to(BF16); 
    if ZERO: a.zero_()
    if a.shape[1]>k: a[:,k:]=0
    w = (torch.randn(n,k,device="cuda")/math.sqrt(k)).to(BF16)
    if ZERO: w.zero_()
    lin = PackedLinear(w, torch.zeros(n,dtype=BF16), "cuda")
    c = torch.empty(m, lin.n_pad, dtype=BF16, device="cuda"); r = torch.randn(m, lin.n_pad, device="cuda").to(BF16)
    args = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), lin.struct(), c.data_ptr(), c.stride(0), r.data_ptr(), r.stride(0), 0, m, epi, 0, 0, None, 0)
    for kk,v in env.items(): os.environ[kk]=v
    dt = timeit(lambda: _lib.check(lib.md_gemm_bf16(C.byref(args), stream())))
    for kk in env: os.environ.pop(kk)
    return 2.0*m*n*k/dt/1e12
shapes=[(23328,1152,4304,1),(23328,1152,3456,0),(23328,1152,1152,2),(23328,4304,1152,2),(46720,2048,14336,1),(46720,2048,2048,2),(46720,8192,2048,2),(4096,4096,4096,0),(8192,8192,8192,0)]
def run2(m,k,n,epi):
    r={}
    for rep in range(2):
        for key,t in (('alt','11'),('persist','15')):
            r.setdefault(key,[]).append(run(m,k,n,epi,{'MD_GEMM_TILE':t}))
    return f"m={m} k={k} n={n} epi={epi}: alternating {r['alt'][0]:6.0f} {r['alt'][1]:6.0f} | + persistent tile loop {r['persist'][0]:6.0f} {r['persist'][1]:6.0f}"
for sh in shapes:
    print(run2(*sh), flush=True)
