#!/bin/bash
mkdir -p gpurun_out/prof
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "not full_size" > gpurun_out/tests_m.log 2>&1
grep -E "passed|failed|^E  " gpurun_out/tests_m.log | tail -6
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log | cut -c1-1700
timeout 300 python bench.py --steps 2 --warmup 1 --no-graphs --no-cpu-baseline > gpurun_out/bench_nographs.log 2>&1
tail -1 gpurun_out/bench_nographs.log | cut -c1-300
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/trace -o r1 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --latency-runs 0 > $R/gpurun_out/prof/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof/fetch -o r1 -- python $R/bench.py --steps 1 --warmup 0 --tokens 1 --no-cpu-baseline --latency-runs 0 --no-graphs > $R/gpurun_out/prof/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof/write -o r1 -- python $R/bench.py --steps 1 --warmup 0 --tokens 1 --no-cpu-baseline --latency-runs 0 --no-graphs > $R/gpurun_out/prof/write.log 2>&1
cd $R
find gpurun_out/prof -type f | head -30
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
find gpurun_out/prof -name "*.db" -delete
python - <<'PY'
import csv, glob, collections
for kind in ("fetch","write"):
    fs = glob.glob(f"gpurun_out/prof/{kind}/**/*counter_collection.csv", recursive=True)
    if not fs: print(kind, "no counter csv"); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(fs[0]) as f:
        r = csv.DictReader(f)
        for row in r:
            k = row.get("Kernel_Name", "")[:70]
            agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0) or 0)
    top = sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]
    out = open(f"gpurun_out/prof/{kind}_summary.csv", "w")
    out.write("kernel,dispatches,counter_sum,counter_avg\n")
    for k, (n, v) in top:
        line = f'"{k}",{n},{v:.1f},{v/n:.1f}'
        print(kind, line); out.write(line + "\n")
PY
