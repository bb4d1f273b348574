#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/pair; mkdir -p $O
L="--no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --no-fp8-full-leg --no-strict-leg --no-second-oracle --latency-runs 0 --only-timed-steps"
for pair in 0 1; do
  rm -rf $O/t$pair
  ( cd /tmp && export TMPDIR=/tmp && MD_PIPE_PAIR=$pair timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$pair -o b -- python $R/bench.py $L --steps 8 --warmup 2 > $O/t$pair.log 2>&1 )
  f=$(find $O/t$pair -name "*kernel_stats.csv" | head -1); cp $f $O/stats_pair$pair.csv
  echo "== MD_PIPE_PAIR=$pair: $(grep only_timed $O/t$pair.log | tail -1)"
  python - <<PY
import csv
rows=list(csv.DictReader(open("$O/stats_pair$pair.csv")))
fam={}
for r in rows:
    k=r["Name"]; t=float(r["TotalDurationNs"])/1e6; n=int(r["Calls"])
    f=("tile_gemm" if "gemm_w4" in k or "gemm_bf16_kernel<256" in k or "gemm_bf16_kernel<128, 128" in k else
       "decode_gemm" if "gemm_bf16_kernel<64" in k or "gemm_pair" in k or "gemm_bf16_kernel<128, 64" in k else
       "attn_decode" if "attn_decode" in k else "attn_prefill" if "attn_prefill" in k else "tail/ln" if "reduce_residual" in k or "layernorm" in k else "other")
    a=fam.setdefault(f,[0,0.0]); a[0]+=n; a[1]+=t
for f,(n,t) in sorted(fam.items(), key=lambda x:-x[1][1]): print(f"   {f:14s} calls {n:6d}  total {t:9.1f} ms")
PY
done
find $O -name "*kernel_trace.csv" -delete
