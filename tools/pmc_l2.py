#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import kernel_bench as kb
for xcd, g in (("1", "8"), ("1", "1"), ("1", "32"), ("0", "8"), ("0", "1")):
    os.environ["MD_GEMM_XCD"] = xcd; os.environ["MD_GEMM_GROUP_M"] = g
    print("XCD", xcd, "GROUP_M", g, flush=True)
    kb.bench_gemm(8192, 8192, 8192, 0, tiles=("0",))
    kb.bench_gemm(46720, 2048, 6144, 0, tiles=("0",))
