#!/usr/bin/env python3
"""Prefill attention time vs sequence length at constant FLOPs: separates the per-tile cost
from the per-workgroup (prologue / epilogue / launch) cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import kernel_bench as kb
for hd, h in ((72, 16), (64, 32)):
    for b, t in ((512, 183), (128, 365), (32, 729), (8, 1458), (2, 2916)):
        kb.bench_attn(b * (16 // h if h > 16 else 1) if False else b, h, t, hd, prefix=(t if hd == 64 else None))
