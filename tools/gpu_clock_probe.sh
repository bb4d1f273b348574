#!/bin/bash
# Effective shader clock per kernel = GRBM_GUI_ACTIVE (cycles the GPU was busy) / kernel duration,
# for this library's tile GEMM next to the vendor library's on the same shapes.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/clock
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/clock -o c -- python $R/tools/vendor_gemm_compare.py > $R/gpurun_out/clock/run.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
cc = glob.glob("gpurun_out/clock/**/*counter_collection.csv", recursive=True)
kt = glob.glob("gpurun_out/clock/**/*kernel_trace.csv", recursive=True)
if not cc or not kt:
    print("missing csv", cc, kt); raise SystemExit
dur = {}
for row in csv.DictReader(open(kt[0])):
    dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), row["Kernel_Name"])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for row in csv.DictReader(open(cc[0])):
    if row["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    d = dur.get(row["Dispatch_Id"])
    if not d or d[0] < 200000: continue   # kernels >= 0.2 ms only
    name = d[1]
    fam = "md tile gemm" if "gemm_bf16_kernel<256" in name else ("vendor " + name[:40] if ("Cijk" in name or "gemm" in name.lower()) else None)
    if fam is None: continue
    a = agg[fam]; a[0] += 1; a[1] += float(row["Counter_Value"]); a[2] += d[0]
out = open("gpurun_out/clock/summary.txt", "w")
for fam, (n, cyc, ns) in sorted(agg.items()):
    line = f"{fam}: launches {n}, GRBM_GUI_ACTIVE/duration = {cyc/ns:.3f} cycles/ns (GHz if per-device; compare ratios)"
    print(line); out.write(line + "\n")
PY
find gpurun_out/clock -name "*.csv" -size +4M -delete
