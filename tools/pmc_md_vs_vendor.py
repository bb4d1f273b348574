#!/usr/bin/env python3
"""Parse the rocprofv3 --pmc passes of tools/vendor_gemm_compare.py (tools/visits_r4/v06_vendor_pmc.sh): per shape, this library's
tile GEMM next to the vendor kernel -- counters per launch, duration from the kernel trace, clock = GRBM_GUI_ACTIVE / duration."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
shapes = sys.argv[2].split(";")
for d in sorted(glob.glob(root + "/pass*")):
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not cc or not kt:
        print(d, "no csv")
        continue
    dur = {}
    for row in csv.DictReader(open(kt[0])):
        dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
    disp = collections.OrderedDict()
    for row in csv.DictReader(open(cc[0])):
        name = row["Kernel_Name"]
        fam = "md" if "gemm_w4_kernel" in name else ("vendor" if "Cijk" in name else None)
        if fam is None:
            continue
        e = disp.setdefault(int(row["Dispatch_Id"]), {"fam": fam, "name": name[:40], "grid": row.get("Grid_Size"), "c": {}})
        e["c"][row["Counter_Name"]] = e["c"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    # segments: md launches of shape 0, vendor launches of shape 0, md launches of shape 1, ...
    seg, prev = [], None
    for did in sorted(disp):
        e = disp[did]
        if e["fam"] != prev:
            seg.append((e["fam"], []))
            prev = e["fam"]
        seg[-1][1].append((did, e))
    print("==", d.split("/")[-1])
    si = -1
    for fam, items in seg:
        if fam == "md":
            si += 1
        items = items[len(items) // 3:]  # skip the warm-up launches
        n = len(items)
        t = sum(dur.get(str(did), 0.0) for did, _ in items) / n
        cs = collections.defaultdict(float)
        for _, e in items:
            for k, v in e["c"].items():
                cs[k] += v / n
        m, k, nn = (int(x) for x in shapes[si].split(","))
        line = f"{shapes[si]:18s} {fam:6s} {items[0][1]['name'][:28]:28s} grid {items[0][1]['grid']:>7s} {t*1e6:8.1f} us {2.0*m*k*nn/t/1e12:7.0f} TF/s | "
        line += "  ".join(f"{k}={v:.4g}" for k, v in sorted(cs.items()))
        if "GRBM_GUI_ACTIVE" in cs:
            line += f"  clock={cs['GRBM_GUI_ACTIVE']/t/1e9:.3f} GHz"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
            line += f"  mfma_busy={cs['SQ_VALU_MFMA_BUSY_CYCLES']/(cs['GRBM_GUI_ACTIVE']*1024):.3f} of SIMD-cycles"
        if "FETCH_SIZE" in cs:
            line += f"  fetch={cs['FETCH_SIZE']*1024*2/1e9:.2f} GB (x2 gfx950)"
        print(line)
