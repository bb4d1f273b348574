#!/usr/bin/env python3
"""Run the tile GEMM and the two prefill-attention shapes a few times (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import kernel_bench as kb
TILE = os.environ.get("PMC_TILE", "20")  # 20 = four-wave 256x256, 11 = eight-wave baseline
kb.bench_gemm(8192, 8192, 8192, 0, tiles=(TILE,))
kb.bench_gemm(46720, 2048, 2048, 2, tiles=(TILE,))
kb.bench_gemm(93312, 1152, 3456, 0, tiles=(TILE,))
kb.bench_attn(32, 16, 729, 72)
kb.bench_attn(64, 32, 730, 64, prefix=730)
