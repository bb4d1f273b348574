#!/usr/bin/env python3
"""Timeline reading of a rocprofv3 kernel trace of the PIPELINED bench (two HIP streams): for the steady-state window
(between the first and the last `argmax`-free boundary it finds: simply the middle 60 % of the trace by time) it prints

  * the window, the union of all kernel intervals (GPU busy), the idle remainder and the gaps' size distribution,
  * summed kernel time by kind, and how much of each kind's time another kind's kernel is running too,
  * average durations of the big kernels inside the window (to compare with the eager step's).

    python tools/trace_timeline.py <kernel_trace.csv> [lo_frac=0.35] [hi_frac=0.95]
"""
import bisect
import collections
import csv
import sys

path = sys.argv[1]
lo_f = float(sys.argv[2]) if len(sys.argv) > 2 else 0.35
hi_f = float(sys.argv[3]) if len(sys.argv) > 3 else 0.95


def kind(n):
    if "attn_decode" in n:
        return "dec_attn"
    if "gemm_bf16_kernel<64, 64" in n or "gemm_pair_kernel" in n:
        return "dec_gemm"
    if "reduce_residual_ln" in n:
        return "dec_tail"
    if "gemm_w4" in n or "gemm_bf16_kernel<" in n:
        return "enc_gemm"
    if "attn_prefill" in n:
        return "enc_attn"
    if "at::native" in n or "rocblas" in n.lower() or "Cijk" in n:
        return "torch"
    return "other"


rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
lo, hi = t0 + int((t1 - t0) * lo_f), t0 + int((t1 - t0) * hi_f)
sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
span = (hi - lo) / 1e6


def union(iv):
    iv = sorted(iv)
    out = []
    cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce:
            out.append((cs, ce))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    out.append((cs, ce))
    return out


allu = union([(s, e) for s, e, _ in sel])
busy = sum(e - s for s, e in allu) / 1e6
gaps = [allu[i + 1][0] - allu[i][1] for i in range(len(allu) - 1)]
print(f"window {span:.1f} ms, {len(sel)} kernels; GPU busy (union) {busy:.1f} ms = {busy / span:.3f}; idle {span - busy:.1f} ms in {len(gaps)} gaps")
hist = collections.Counter()
tot = collections.Counter()
for g in gaps:
    b = "<2us" if g < 2e3 else "<5us" if g < 5e3 else "<20us" if g < 20e3 else "<100us" if g < 1e5 else "<1ms" if g < 1e6 else ">=1ms"
    hist[b] += 1
    tot[b] += g / 1e6
print("gaps:", {k: (hist[k], round(tot[k], 2)) for k in ("<2us", "<5us", "<20us", "<100us", "<1ms", ">=1ms") if hist[k]}, "(count, ms)")
# what stands on either side of the long gaps (>= 1 ms): the kernels that end last before / start first after each
sel_by_end = sorted(sel, key=lambda r: r[1])
ends = [r[1] for r in sel_by_end]
sel_starts = [r[0] for r in sel]
short = lambda n: n.replace("void (anonymous namespace)::", "").split("(")[0][:70]
for i in range(len(allu) - 1):
    g = allu[i + 1][0] - allu[i][1]
    if g >= 1e6:
        j = bisect.bisect_right(ends, allu[i][1]) - 1
        k = bisect.bisect_left(sel_starts, allu[i + 1][0])
        before = [short(sel_by_end[x][2]) for x in range(max(0, j - 2), j + 1)]
        after = [short(sel[x][2]) for x in range(k, min(len(sel), k + 3))]
        print(f"  gap {g / 1e6:6.2f} ms at +{(allu[i][1] - lo) / 1e6:8.1f} ms: after {before} -> before {after}")
by = collections.defaultdict(list)
for s, e, n in sel:
    by[kind(n)].append((s, e))
print("summed kernel ms by kind:", {k: round(sum(e - s for s, e in v) / 1e6, 1) for k, v in by.items()}, {k: len(v) for k, v in by.items()})
enc = union(by["enc_gemm"] + by["enc_attn"]) if (by["enc_gemm"] or by["enc_attn"]) else []
starts = [s for s, _ in enc]


def overlap(s, e):
    t, i = 0, max(0, bisect.bisect_right(starts, s) - 1)
    while i < len(enc) and enc[i][0] < e:
        a, b = max(s, enc[i][0]), min(e, enc[i][1])
        if b > a:
            t += b - a
        i += 1
    return t


for kd in ("dec_attn", "dec_gemm", "dec_tail", "other"):
    ks = by.get(kd, [])
    if not ks:
        continue
    t = sum(e - s for s, e in ks)
    ov = sum(overlap(s, e) for s, e in ks)
    print(f"{kd}: n={len(ks)} avg {t / len(ks) / 1e3:.1f} us; fraction of its time with an encode kernel (tile GEMM / prefill attention) also running: {ov / t:.2f}")
names = collections.defaultdict(list)
for s, e, n in sel:
    names[short(n)].append(e - s)
top = sorted(names.items(), key=lambda kv: -sum(kv[1]))[:12]
for n, d in top:
    print(f"  {sum(d) / 1e6:8.1f} ms  n={len(d):5d} avg {sum(d) / len(d) / 1e3:8.1f} us  {n}")
