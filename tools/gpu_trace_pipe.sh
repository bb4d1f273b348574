#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tp
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tp -o t -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --latency-runs 0 > $R/gpurun_out/tp/run.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/tp/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id","?"), r.get("Stream_Id", r.get("Queue_Id","?"))))
rows.sort()
t0 = rows[0][0]
def kind(n):
    if "attn_decode" in n: return "dec_attn"
    if "Li64ELi128" in n or "<64, 128" in n: return "dec_gemm"
    if "gemm_bf16" in n: return "enc_gemm"
    if "attn_prefill" in n: return "enc_attn"
    return "other"
# take the last 40% of the run (steady state, timed region)
tend = rows[-1][1]; lo = t0 + int((tend - t0) * 0.55)
sel = [r for r in rows if r[0] >= lo]
span = (sel[-1][1] - sel[0][0]) / 1e6
busy = collections.Counter(); cnt = collections.Counter()
for s, e, n, q, st in sel:
    busy[kind(n)] += (e - s) / 1e6; cnt[kind(n)] += 1
print(f"window {span:.1f} ms; summed kernel ms by kind:", {k: round(v, 1) for k, v in busy.items()}, dict(cnt))
# concurrency: fraction of decode-kernel time during which an encode kernel is also running
enc = [(s, e) for s, e, n, q, st in sel if kind(n) in ("enc_gemm", "enc_attn")]
enc.sort()
import bisect
starts = [s for s, e in enc]
def overlap(s, e):
    tot = 0; i = max(0, bisect.bisect_left(starts, s) - 50)
    while i < len(enc) and enc[i][0] < e:
        a, b = max(s, enc[i][0]), min(e, enc[i][1])
        if b > a: tot += b - a
        i += 1
    return tot
for kd in ("dec_attn", "dec_gemm"):
    ks = [(s, e) for s, e, n, q, st in sel if kind(n) == kd]
    tot = sum(e - s for s, e in ks); ov = sum(min(overlap(s, e), e - s) for s, e in ks)
    avg = tot / max(1, len(ks)) / 1e3
    print(f"{kd}: n={len(ks)} avg {avg:.1f} us, fraction of its time with an encode kernel also running: {ov / max(1, tot):.2f}")
# union busy time of encode kernels and of decode kernels
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return (tot + ce - cs) / 1e6
dec = [(s, e) for s, e, n, q, st in sel if kind(n).startswith("dec")]
print(f"union busy: encode {union(enc):.1f} ms, decode {union(dec):.1f} ms, all {union([(s,e) for s,e,*_ in sel]):.1f} ms of window {span:.1f} ms")
print("queues:", collections.Counter((kind(n), q) for s, e, n, q, st in sel).most_common(8))
PY
find gpurun_out/tp -name "*.csv" -size +1M -delete
tail -2 gpurun_out/tp/run.log | cut -c1-200
