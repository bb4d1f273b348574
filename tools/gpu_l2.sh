#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/l2
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/l2 -o p -- python $R/tools/pmc_l2.py > $R/gpurun_out/l2/run.log 2>&1
cd $R
grep -E "XCD|gemm m" gpurun_out/l2/run.log | paste - - - | cut -c1-260
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/l2/**/*counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "gemm_bf16" in r["Kernel_Name"]]
# group consecutive dispatches by (grid) in dispatch order; print hit rate per dispatch id bucket
by = collections.OrderedDict()
for r in rows:
    key = (int(r["Dispatch_Id"]), r["Grid_Size"])
    by.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
last = None; acc = []
def flush():
    if acc:
        h = sum(a.get("TCC_HIT_sum", 0) for a in acc); m = sum(a.get("TCC_MISS_sum", 0) for a in acc)
        print(f"grid {last}: dispatches {len(acc)} hit {h/len(acc):.3e} miss {m/len(acc):.3e} hit-rate {h/(h+m+1e-9):.3f}")
for (d, g), c in by.items():
    if g != last: flush(); acc = []; last = g
    acc.append(c)
flush()
PY
