#!/bin/bash
# B=1 caption latency under rocprofv3 --kernel-trace: per-kernel durations and the gaps between consecutive kernels
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/b1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/b1 -o b1 -- python $R/tools/latency_profile.py > $R/gpurun_out/b1/run.log 2>&1
echo "rc=$?"; grep "^run" $R/gpurun_out/b1/run.log
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/b1/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last caption = the last 40% of the trace; take the final 32*100-ish kernels
tail = rows[-3200:]
dur = collections.defaultdict(list); gaps = []
for a, b in zip(tail, tail[1:]):
    dur[a['Kernel_Name'][:70]].append(int(a['End_Timestamp']) - int(a['Start_Timestamp']))
    gaps.append(int(b['Start_Timestamp']) - int(a['End_Timestamp']))
span = int(tail[-1]['End_Timestamp']) - int(tail[0]['Start_Timestamp'])
busy = sum(sum(v) for v in dur.values())
print(f"last {len(tail)} kernels: span {span/1e6:.2f} ms, kernel time {busy/1e6:.2f} ms, gaps {sum(gaps)/1e6:.2f} ms (median gap {sorted(gaps)[len(gaps)//2]/1e3:.2f} us)")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:70s} n={len(v):5d} avg={sum(v)/len(v)/1e3:7.2f} us tot={sum(v)/1e6:7.2f} ms")
PY
find gpurun_out/b1 -name "*kernel_trace.csv" -size +8M -delete
