#!/bin/bash
# HBM-side traffic of the bench step per kernel family: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE)
# over the same command (EXACTLY one eager B=64 step, one decode token), summarised into gpurun_out/pmc_traffic.json
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/traffic
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/traffic/$c -o r1 -- python $R/bench.py --steps 1 --warmup 0 --tokens 1 --batch 64 --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --latency-runs 0 --no-graphs --no-pipeline --only-timed-steps > $R/gpurun_out/traffic/$c.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json
out = {}
for kind in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"gpurun_out/traffic/{kind}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(kind, "no counter csv"); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(fs[0])):
        k = row.get("Kernel_Name", "")
        fam = "tile_gemm" if ("gemm_w4_kernel" in k or "gemm_bf16_kernel<256" in k or "gemm_bf16_kernel<128" in k) else \
              "decode_gemm" if "gemm_bf16_kernel<64" in k or "gemm_pair_kernel" in k else \
              "attn_prefill" if "attn_prefill" in k else "attn_decode" if "attn_decode" in k else None
        if fam is None: continue
        agg[fam][0] += 1; agg[fam][1] += float(row.get("Counter_Value", 0) or 0)
    for fam, (n, v) in agg.items():
        out.setdefault(fam, {})["launches"] = n
        out[fam][kind + "_kb_sum"] = v
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
find gpurun_out/traffic -name "*.csv" -size +8M -delete
