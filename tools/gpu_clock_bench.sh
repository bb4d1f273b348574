#!/bin/bash
# Effective shader clock per kernel family INSIDE the bench (sustained load): GRBM_GUI_ACTIVE summed over the 8 XCDs
# divided by 8 x kernel duration.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/clockb
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/clockb -o c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --latency-runs 0 --only-timed-steps > $R/gpurun_out/clockb/run.log 2>&1
echo "rc=$?"
cd $R
python - <<'PY'
import csv, glob, collections
cc = glob.glob("gpurun_out/clockb/**/*counter_collection.csv", recursive=True)
kt = glob.glob("gpurun_out/clockb/**/*kernel_trace.csv", recursive=True)
dur = {}
for row in csv.DictReader(open(kt[0])):
    dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), row["Kernel_Name"])
fams = (("gemm_w4_kernel<0", "w4 GEMM bias"), ("gemm_w4_kernel<1", "w4 GEMM gelu"), ("gemm_w4_kernel<2", "w4 GEMM residual"),
        ("attn_prefill_dma_kernel<72", "prefill attention hd72"), ("attn_prefill_dma_kernel<64", "prefill attention hd64"),
        ("attn_decode_kernel", "decode attention"), ("gemm_pair_kernel", "decode proj+fc2 pair"), ("layernorm_kernel<3", "layer norm (ViT)"))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for row in csv.DictReader(open(cc[0])):
    if row["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    d = dur.get(row["Dispatch_Id"])
    if not d or d[0] < 20000: continue
    for pat, fam in fams:
        if pat in d[1]:
            a = agg[fam]; a[0] += 1; a[1] += float(row["Counter_Value"]); a[2] += d[0]
out = open("gpurun_out/clockb/summary.txt", "w")
for fam, (n, cyc, ns) in sorted(agg.items()):
    line = f"{fam:28s} launches {n:5d}  mean duration {ns / n / 1e3:8.1f} us  GRBM_GUI_ACTIVE / (8 x duration) = {cyc / ns / 8:.3f} GHz"
    print(line); out.write(line + "\n")
PY
find gpurun_out/clockb -name "*.csv" -size +4M -delete
