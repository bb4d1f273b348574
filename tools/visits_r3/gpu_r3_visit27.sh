#!/bin/bash
# kernel trace of a run that includes the fp8_full leg (6 fp8 B=64 steps: 2 warm-up + 3 timed + 1 eager)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/v27; mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --no-cpu-baseline --no-vqa-leg --no-fp8-leg --no-dedup-leg --no-detect13-leg --latency-runs 0 --steps 3 --warmup 1 > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
grep '^{"metric"' $O/trace.log | tail -1 > $O/bench.json
find $O/trace -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/v27/kernel_stats.csv")))
for r in rows[:40]:
    n = r["Name"]
    if any(t in n for t in ("f8", "fp8", "F8", "quant", "amax")) or int(r["TotalDurationNs"]) > 3e7:
        print("%-100s %6s %9.2f ms avg %8.1f us" % (n[:100], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
