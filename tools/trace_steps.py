#!/usr/bin/env python3
"""Per-step time of the tile-GEMM launches in a rocprofv3 kernel trace of bench.py: the B=64 steps are the groups of
255 big-tile launches in time order; prints each step's summed duration and the implied PF/s (193.0 TFLOP per step
at 2B / B=64), so the pipelined (timed) steps can be compared with the eager profile step that bench.py brackets with
HIP events.

    python tools/trace_steps.py <kernel_trace.csv> [tflop_per_step=193.0]
"""
import csv
import sys

path = sys.argv[1]
tf = float(sys.argv[2]) if len(sys.argv) > 2 else 193.0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if "gemm_w4_kernel" in n or "gemm_bf16_kernel<256" in n or "gemm_bf16_kernel<128" in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
# a step's encode phase is one burst of launches; split where the gap between consecutive tile GEMMs exceeds 20 ms
steps, cur = [], []
for s, e, n in rows:
    if cur and s - cur[-1][1] > 20e6:
        steps.append(cur)
        cur = []
    cur.append((s, e, n))
if cur:
    steps.append(cur)
for i, st in enumerate(steps):
    busy = sum(e - s for s, e, _ in st)
    span = st[-1][1] - st[0][0]
    print(f"burst {i}: {len(st):4d} launches, kernel time {busy / 1e6:7.2f} ms, span {span / 1e6:7.2f} ms, "
          f"{tf / (busy / 1e9) / 1e3:5.3f} PF/s if this is one full step")
