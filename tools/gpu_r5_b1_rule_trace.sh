#!/bin/bash
# Round 5: in-situ kernel durations of a single-image caption's encode under round 2's and round 5's tile rule (rocprofv3 kernel trace each)
R=$GRAFT_REPO_ROOT
for rule in ${RULES:-0 1}; do
  mkdir -p $R/gpurun_out/b1e; rm -rf $R/gpurun_out/b1e/*
  ( cd /tmp && export TMPDIR=/tmp && MD_SMALL_M_RULE=$rule timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/b1e -o b1e -- python $R/tools/b1_encode_trace.py > $R/gpurun_out/b1e/run.log 2>&1 )
  echo "=== MD_SMALL_M_RULE=$rule"
  cd $R
  python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/b1e/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'decode_b1_kernel' in r['Kernel_Name']]
runs = []
for i in idx:
    if runs and i - runs[-1][-1] < 20: runs[-1].append(i)
    else: runs.append([i])
a, b = runs[-2][-1] + 1, runs[-1][0]
enc = rows[a:b]
# split at the first decoder fused layer: ViT part / text part
pf = next(i for i, r in enumerate(enc) if 'gemm_w4_kernel<3' in r['Kernel_Name'])
for name, seg in (("vision", enc[:pf - 1]), ("prefill", enc[pf - 1:])):
    dur = collections.defaultdict(list)
    for r in seg: dur[r['Kernel_Name'][:96]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    span = int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])
    print(f" {name}: {len(seg)} kernels, span {span/1e6:.3f} ms, kernel time {sum(sum(v) for v in dur.values())/1e6:.3f} ms")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:7]:
        print(f"   {k:96s} n={len(v):3d} avg={sum(v)/len(v)/1e3:7.2f} us tot={sum(v)/1e6:6.3f} ms")
PY
done
rm -rf $R/gpurun_out/b1e
