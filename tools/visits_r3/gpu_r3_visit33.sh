#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/v33
export PYTHONUNBUFFERED=1
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 5 900 python bench.py > gpurun_out/v33/bench_default.log 2>&1; echo "bench rc=$?"
grep '^{"metric"' gpurun_out/v33/bench_default.log | tail -1 > gpurun_out/v33/bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/v33/bench_default.json"))
print("value %.1f images/s  ms/step %.1f  roofline.frac %.3f  vit %.3f  decode_step %.3f  parity %s/%s ok=%s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["vit_encoder"]["frac"], d["decode_step"]["frac"], d["parity_exact"], d["parity_checked"], d["parity_ok"]))
print("phase", d["phase_ms"], "p50", d.get("p50_caption_latency_ms"))
for k in ("fp8_full","fp8_decode","detect13","detect13_fp8","vqa32","dedup","cpu_baseline"):
    x=d.get(k)
    if isinstance(x,dict): print(k, {kk:(round(v,2) if isinstance(v,float) else v) for kk,v in x.items() if kk in ("images_per_sec","ms_per_step","value","unit","cores","host_cores","kind","speedup_vs_bf16_value","images_per_sec_tiling_prefetched","phase_ms")})
PY
