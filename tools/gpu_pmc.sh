#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc/a -o p -- python $R/tools/pmc_kernels.py > $R/gpurun_out/pmc/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc/b -o p -- python $R/tools/pmc_kernels.py > $R/gpurun_out/pmc/b.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for run in ("a","b"):
    fs = glob.glob(f"gpurun_out/pmc/{run}/**/*counter_collection.csv", recursive=True)
    if not fs: print(run, "no csv; log tail:", open(f"gpurun_out/pmc/{run}.log").read()[-600:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0,0.0]))
    for row in csv.DictReader(open(fs[0])):
        k = row["Kernel_Name"]
        if "gemm_bf16" not in k and "gemm_w4" not in k and "attn_prefill" not in k: continue
        key = k[:60] + " grid=" + row.get("Grid_Size","?")
        c = agg[key][row["Counter_Name"]]; c[0]+=1; c[1]+=float(row["Counter_Value"])
    out = open(f"gpurun_out/pmc/summary_{run}.txt","w")
    for k, cs in agg.items():
        line = k + " :: " + "  ".join(f"{n}={v[1]/v[0]:.3e}" for n, v in sorted(cs.items()))
        print(line); out.write(line+"\n")
PY
find gpurun_out/pmc -name "*.csv" -size +5M -delete
