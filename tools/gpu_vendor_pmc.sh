#!/bin/bash
# Instruction mix of the vendor library's GEMM kernel next to this library's tile GEMM (same shapes).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/vpmc
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/vpmc -o v -- python $R/tools/vendor_gemm_compare.py > $R/gpurun_out/vpmc/run.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
cc = glob.glob("gpurun_out/vpmc/**/*counter_collection.csv", recursive=True)
if not cc: print("no csv"); print(open("gpurun_out/vpmc/run.log").read()[-800:]); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for row in csv.DictReader(open(cc[0])):
    name = row["Kernel_Name"]
    fam = "md tile gemm" if "gemm_bf16_kernel<256" in name else ("vendor " + name[:48] if "Cijk" in name else None)
    if fam is None: continue
    key = fam + " grid=" + row.get("Grid_Size", "?") + " wg=" + row.get("Workgroup_Size", "?") + " lds=" + row.get("LDS_Block_Size", "?") + " vgpr=" + row.get("VGPR_Count", "?") + " agpr=" + row.get("Accum_VGPR_Count", "?")
    c = agg[key][row["Counter_Name"]]; c[0] += 1; c[1] += float(row["Counter_Value"])
out = open("gpurun_out/vpmc/summary.txt", "w")
for k, cs in sorted(agg.items()):
    line = k + "\n    " + "  ".join(f"{n}={v[1]/v[0]:.3e}" for n, v in sorted(cs.items()))
    print(line); out.write(line + "\n")
PY
find gpurun_out/vpmc -name "*.csv" -size +4M -delete
