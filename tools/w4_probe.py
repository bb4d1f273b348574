#!/usr/bin/env python3
"""Timing ablations and in-kernel stamps of the four-wave 256x256 GEMM (gemm_w4.hip), bias epilogue.  Needs the measurement build
(MD_W4_ABLATIONS=1 python -c "import __graft_entry__ as g; g.build()").  md_gemm_set_tuning "w4_variant" = MODE + 16 * ABL;
MODE 0 = register-staged operands, 1-3 = LDS-DMA schedules; ABL bits: 2 no operand loads, 4 no barrier, 8 no fragment reads,
16 no epilogue stores, 32 no epilogue, 256 every operand load from one L2-resident MiB, 64 = shader-clock stamps around the waits of gap 48 and the epilogue (results are garbage
with bits 2-32 set)."""
import ctypes as C
import math
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from moondream_amd import _lib
from moondream_amd.weights import PackedLinear
from tools.sweep_gemm import timeit, stream

lib = _lib.load()
BF16 = torch.bfloat16
V = lambda mode, abl: mode + 16 * abl
VARIANTS = [("full", V(0, 0)), ("no-dma", V(0, 2)), ("no-barrier", V(0, 4)), ("no-reads", V(0, 8)), ("mfma+epi", V(0, 14)),
            ("no-stores", V(0, 16)), ("no-epi", V(0, 32)), ("mfma only", V(0, 46)), ("loads from 1 MiB", V(0, 256)), ("loads from 1 MiB, no stores", V(0, 272))]
SHAPES = [(8192, 8192, 8192), (46720, 2048, 2048), (93312, 1152, 3456)]
if os.environ.get("W4_PROBE_SHAPES"):  # "m,k,n;m,k,n"
    SHAPES = [tuple(int(x) for x in t.split(",")) for t in os.environ["W4_PROBE_SHAPES"].split(";")]


def problem(m, k, n):
    a = (torch.randn(m, k, device="cuda") * 0.5).to(BF16)
    w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).to(BF16)
    lin = PackedLinear(w, torch.zeros(n, dtype=BF16), "cuda")
    c = torch.empty(m, lin.n_pad, dtype=BF16, device="cuda")
    args = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), lin.struct(), c.data_ptr(), c.stride(0), None, 0, 0, m, 0, 0, 0, None, 0)
    return a, w, lin, c, args


lib.md_gemm_set_tuning(b"tile", 20)
if "stamps" not in sys.argv[1:]:
    for m, k, n in SHAPES:
        keep = problem(m, k, n)
        args = keep[-1]
        res = {v[0]: [] for v in VARIANTS}
        for _ in range(2):
            for name, var in VARIANTS:
                lib.md_gemm_set_tuning(b"w4_variant", var)
                dt = timeit(lambda: _lib.check(lib.md_gemm_bf16(C.byref(args), stream())), iters=5, warm=2)
                res[name].append(2.0 * m * n * k / dt / 1e12)
        lib.md_gemm_set_tuning(b"w4_variant", 0)
        print(f"m={m} k={k} n={n}: " + " | ".join(f"{nm}: {statistics.median(v):6.0f}" for nm, v in res.items()), flush=True)
        del keep

# stamps: per wave [pairs, cycles first..last stamp, lgkm wait, vmcnt wait, barrier wait, epilogue cycles]
dbg = torch.zeros(256 * 4 * 8, dtype=torch.float32, device="cuda")
lib.md_gemm_set_tuning(b"w4_dbg_lo", C.c_int32(dbg.data_ptr() & 0xffffffff).value)
lib.md_gemm_set_tuning(b"w4_dbg_hi", C.c_int32(dbg.data_ptr() >> 32).value)
for abl, what in ((64, "stamps"), (80, "stamps, no epilogue stores"), (320, "stamps, loads from 1 MiB")):
    for m, k, n in SHAPES:
        keep = problem(m, k, n)
        args = keep[-1]
        lib.md_gemm_set_tuning(b"w4_variant", V(0, abl))
        for _ in range(3):
            _lib.check(lib.md_gemm_bf16(C.byref(args), stream()))
        torch.cuda.synchronize()
        d = dbg.view(256, 4, 8).cpu()
        npair = d[:, :, 0]
        tiles = npair / (k // 64)
        per = d[:, :, 1] / (npair - 1).clamp(min=1)
        print(f"{what}: m={m} k={k} n={n}: pairs/wave {npair.mean():.0f}  cycles per pair (epilogues inside) {per.mean():.0f} "
              f"| per pair: lgkm wait {(d[:, :, 2] / npair).mean():.0f}  vmcnt wait {(d[:, :, 3] / npair).mean():.0f} "
              f"(max wave {(d[:, :, 3] / npair).max():.0f})  barrier wait {(d[:, :, 4] / npair).mean():.0f} (max {(d[:, :, 4] / npair).max():.0f}) "
              f"| per tile: epilogue + refill {(d[:, :, 5] / tiles).mean():.0f} cycles, vmcnt wait of the first pair {(d[:, :, 6] / tiles).mean():.0f}", flush=True)
        del keep
lib.md_gemm_set_tuning(b"w4_variant", 0)
lib.md_gemm_set_tuning(b"tile", -1)
