#!/bin/bash
# Round 5: kernel timeline of ONE eager, non-pipelined B = 64 step (1 decode token): what the vision / prefill phases contain
# besides tile GEMMs, attention and layer norms (kernel time by name, idle gaps, memcpy activity)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/eager1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
LEGS="--no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --no-fp8-full-leg --no-strict-leg --no-second-oracle --latency-runs 0"
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o e1 -- python $R/bench.py $LEGS --steps 1 --warmup 1 --tokens 1 --batch 64 --no-graphs --no-pipeline --only-timed-steps > $O/run.log 2>&1
echo "rc=$?"; tail -2 $O/run.log
cd $R
python - <<'PY' | tee gpurun_out/r05_eager_step_timeline.txt
import csv, glob, collections
f = glob.glob('gpurun_out/eager1/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
mc = glob.glob('gpurun_out/eager1/**/*memory_copy_trace.csv', recursive=True)
copies = []
if mc:
    for r in csv.DictReader(open(mc[0])):
        copies.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Direction', '?'), r.get('Bytes', '?')))
# the last step = from the last patchify launch to the end
pidx = [i for i, r in enumerate(rows) if 'patchify' in r[2]]
start = pidx[-1]
step = rows[start:]
t0 = step[0][0]
# phase boundary: first launch of the decoder's fused layer (gemm_w4_kernel<3) = prefill starts
pf = next(i for i, r in enumerate(step) if 'gemm_w4_kernel<3' in r[2])
# the text LN right before it
vis = step[:pf - 1]
def report(name, seg):
    span = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"== {name}: {len(seg)} kernels, span {span/1e6:.2f} ms, kernel time {busy/1e6:.2f} ms, idle {sum(pos)/1e6:.2f} ms in {len(pos)} gaps (median {sorted(pos)[len(pos)//2]/1e3 if pos else 0:.2f} us, max {max(pos)/1e3 if pos else 0:.1f} us)")
    d = collections.defaultdict(list)
    for s, e, n in seg: d[n[:90]].append(e - s)
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:14]:
        print(f"   {k:90s} n={len(v):4d} avg={sum(v)/len(v)/1e3:8.1f} us tot={sum(v)/1e6:7.2f} ms")
    big = sorted(((seg[i + 1][0] - seg[i][1], seg[i][2][:50], seg[i + 1][2][:50]) for i in range(len(seg) - 1)), reverse=True)[:6]
    for g, a, b in big: print(f"   gap {g/1e3:8.1f} us between {a} -> {b}")
report("vision phase (patchify .. projector)", vis)
report("prefill + 1 decode token", step[pf - 1:])
cs = [c for c in copies if c[0] >= t0 - 5_000_000]
print("memory copies from 5 ms before the step on:", len(cs))
for s, e, d, b in cs[:12]: print(f"   t={(s - t0)/1e3:9.1f} us dur={(e - s)/1e3:8.1f} us {d} {b} B")
PY
find gpurun_out/eager1 -name "*.csv" -size +4M -delete
