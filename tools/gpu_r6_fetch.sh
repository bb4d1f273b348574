#!/bin/bash
# Round 6: the FETCH_SIZE pass alone (it died at start-up in the first visit) + merge with the WRITE_SIZE figures already collected
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/prof6; mkdir -p $O
LEGS="--no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --no-fp8-full-leg --no-strict-leg --no-second-oracle --latency-runs 0"
ONE="python $R/bench.py $LEGS --steps 1 --warmup 0 --tokens 1 --batch 64 --no-graphs --no-pipeline --only-timed-steps"
for c in FETCH_SIZE WRITE_SIZE; do
  for try in 1 2 3; do
    rm -rf $O/$c; ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o r1 -- $ONE > $O/$c.log 2>&1 )
    f=$(find $O/$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && break; echo "$c try $try: no counter csv"
  done
done
python - <<'PY'
import csv, glob, collections, json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof6"
def fam_of(k):
    if "gemm_w4_kernel" in k or "gemm_bf16_kernel<256" in k or "gemm_bf16_kernel<128" in k: return "tile_gemm"
    if "gemm_bf16_kernel<64" in k or "gemm_pair_kernel" in k: return "decode_gemm"
    if "attn_prefill" in k: return "attn_prefill"
    if "attn_decode" in k: return "attn_decode"
    return None
out = {}
for kind in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"{O}/{kind}/**/*counter_collection.csv", recursive=True)
    if not fs: print(kind, "no counter csv"); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(fs[0])):
        fam = fam_of(row.get("Kernel_Name", ""))
        if fam is None: continue
        agg[fam][0] += 1; agg[fam][1] += float(row.get("Counter_Value", 0) or 0)
    for fam, (n, v) in agg.items():
        out.setdefault(fam, {})["launches"] = n
        out[fam][kind + "_kb_sum"] = v
json.dump(out, open(f"{O}/pmc_traffic.json", "w"), indent=1)
print("traffic:", json.dumps(out)[:700])
PY
find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +30M -delete
