#!/usr/bin/env python3
"""Per-kernel instruction counts of a -save-temps .s file (whole kernel and its hottest inner loop)."""
import re
import sys

KEYS = ("v_mfma", "ds_read_b128", "ds_write_b128", "buffer_load_dwordx4", "s_waitcnt vmcnt", "s_waitcnt lgkmcnt", "v_mov_b32",
        "v_add_u32", "s_nop", "s_barrier", "v_accvgpr", "scratch_")


def main(path, pat="kernel"):
    s = open(path).read()
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\S+):\s*(?:;.*)?$", s, flags=re.M) if pat in m.group(1)]
    for (a, name) in starts:
        end = s.index("s_endpgm", a)
        lines = [l.strip() for l in s[a:end].split("\n")]
        code = [l for l in lines if l and not l.startswith((";", ".", "_Z")) or l.startswith(".LBB")]
        # inner loops: label ... backward branch to the label; pick the one with the most MFMAs
        labels = {l[:-1].split(":")[0]: i for i, l in enumerate(code) if l.startswith(".LBB")}
        best = None
        for i, l in enumerate(code):
            m = re.match(r"s_cbranch_\w+ (\.LBB\w+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                body = code[labels[m.group(1)]:i]
                n = sum(1 for x in body if x.startswith("v_mfma"))
                if best is None or n > best[0]:
                    best = (n, body)
        tot = {k: sum(1 for l in code if l.startswith(k)) for k in KEYS}
        print(name[-48:], "| whole:", {k: v for k, v in tot.items() if v})
        if best:
            loop = {k: sum(1 for l in best[1] if l.startswith(k)) for k in KEYS}
            print("   hottest loop (%d instr):" % len([l for l in best[1] if not l.startswith('.')]), {k: v for k, v in loop.items() if v})


if __name__ == "__main__":
    main(*sys.argv[1:])
