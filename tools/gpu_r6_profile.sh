#!/bin/bash
# Round-6 evidence on the FINAL code, one visit, every profiler run guarded against the start-up fault some boxes show:
#   1. rocprofv3 --kernel-trace --stats of the bench command (3 pipelined steps)         -> kernel_stats.csv + bench json
#   2. --pmc FETCH_SIZE / WRITE_SIZE passes over ONE eager B=64 step (1 decode token)     -> pmc_traffic.json
#   3. --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over the same step   -> mfma_busy.txt (per kernel family:
#      MFMA-busy share of the SIMD-cycles at the clock the chip sustained, and that clock)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/prof6; mkdir -p $O
export PYTHONUNBUFFERED=1
guarded() {  # guarded <seconds> <log> <command...>
  local t=$1 log=$2; shift 2
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 $t "$@" > $log 2>&1 ) &
  local pid=$!
  while kill -0 $pid 2>/dev/null; do
    if grep -q "Memory access fault" $log 2>/dev/null; then echo "  faulted at start-up: killed ($log)"; pkill -9 -P $pid; kill -9 $pid; break; fi
    sleep 2
  done
  wait $pid 2>/dev/null
}
LEGS="--no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --no-fp8-full-leg --no-strict-leg --no-second-oracle --latency-runs 0"
guarded 150 $O/trace.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py $LEGS --steps 3 --warmup 1
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv && head -12 "$f" | cut -c1-150
grep '^{"metric"' $O/trace.log | tail -1 > $O/bench_traced.json
ONE="python $R/bench.py $LEGS --steps 1 --warmup 0 --tokens 1 --batch 64 --no-graphs --no-pipeline --only-timed-steps"
for c in FETCH_SIZE WRITE_SIZE; do
  guarded 150 $O/$c.log rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o r1 -- $ONE
done
guarded 150 $O/SQ.log rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/SQ -o r1 -- $ONE
python - <<'PY'
import csv, glob, collections, json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof6"
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
print("traffic:", json.dumps(out)[:600])
cc = glob.glob(f"{O}/SQ/**/*counter_collection.csv", recursive=True)
kt = glob.glob(f"{O}/SQ/**/*kernel_trace.csv", recursive=True)
if cc and kt:
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt[0]))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(cc[0])):
        d = dur.get(row["Dispatch_Id"])
        if not d: continue
        k = d[1]
        fam = ("w4 GEMM bias" if "gemm_w4_kernel<0" in k else "w4 GEMM gelu" if "gemm_w4_kernel<1" in k else "w4 GEMM residual" if "gemm_w4_kernel<2" in k else "w4 GEMM qkv|fc1 + rope + KV write" if "gemm_w4_kernel<3" in k
               else "prefill attention hd72" if "attn_prefill_dma_kernel<72" in k else "prefill attention hd64" if "attn_prefill_dma_kernel<64" in k else None)
        if fam is None or d[0] < 20000: continue
        a = agg[fam]
        a[row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE": a["ns"] += d[0]; a["n"] += 1
    with open(f"{O}/mfma_busy.txt", "w") as fo:
        for fam, a in sorted(agg.items()):
            ns, gui = a["ns"], a["GRBM_GUI_ACTIVE"]
            clk = gui / ns / 8 if ns else 0.0          # GHz: GRBM_GUI_ACTIVE is summed over the 8 XCDs
            # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs: 256 CUs x 4 SIMDs x (kernel cycles) is 100 %
            busy = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024) if gui else 0.0
            line = (f"{fam:34s} launches {int(a['n']):4d}  time {ns / 1e6:8.2f} ms  clock {clk:.3f} GHz  MFMA-busy {100 * busy:5.1f} % of SIMD-cycles "
                    f"-> {busy * clk / 2.4 * 100:5.1f} % of the 2.4 GHz peak")
            print(line); fo.write(line + "\n")
PY
find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +30M -delete; du -sh $O | cut -f1
