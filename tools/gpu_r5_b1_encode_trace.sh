#!/bin/bash
# Round 5: which kernels make up the encode (vision + image / prompt prefill) of a single-image caption (p50 latency's other 8 ms)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/b1e
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/b1e -o b1e -- python $R/tools/b1_encode_trace.py > $R/gpurun_out/b1e/run.log 2>&1
echo "rc=$?"
cd $R
python - <<'PY' | tee gpurun_out/r05_b1_encode_kernels.txt
import csv, glob, collections
f = glob.glob('gpurun_out/b1e/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'decode_b1_kernel' in r['Kernel_Name']]
# captions are runs of consecutive decode_b1 launches: the encode of the last caption lies between the last two runs
runs = []
for i in idx:
    if runs and i - runs[-1][-1] <= 3: runs[-1].append(i)
    else: runs.append([i])
# (a 32-token caption decodes in chunks of 16 launches: merge runs that are separated by fewer than 20 other kernels)
caps = []
for r in runs:
    if caps and r[0] - caps[-1][-1] < 20: caps[-1].extend(r)
    else: caps.append(list(r))
runs = caps
a, b = runs[-2][-1] + 1, runs[-1][0]
enc = rows[a:b]
dur = collections.defaultdict(list)
for r in enc: dur[r['Kernel_Name'][:100]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
span = int(enc[-1]['End_Timestamp']) - int(enc[0]['Start_Timestamp'])
busy = sum(sum(v) for v in dur.values())
gaps = [int(y['Start_Timestamp']) - int(x['End_Timestamp']) for x, y in zip(enc, enc[1:])]
print(f"encode of the last caption: {len(enc)} kernels, span {span/1e6:.2f} ms, kernel time {busy/1e6:.2f} ms, gaps {sum(g for g in gaps if g > 0)/1e6:.2f} ms (median {sorted(gaps)[len(gaps)//2]/1e3:.2f} us)")
dec = rows[runs[-1][0]:runs[-1][-1] + 1]
print(f"decode of the last caption: {len(dec)} kernels, span {(int(dec[-1]['End_Timestamp']) - int(dec[0]['Start_Timestamp']))/1e6:.2f} ms")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:100s} n={len(v):4d} avg={sum(v)/len(v)/1e3:7.2f} us tot={sum(v)/1e6:6.3f} ms")
PY
find gpurun_out/b1e -name "*.csv" -size +4M -delete
