#!/usr/bin/env python3
"""CPU study for the fp8 big-tile GEMM ("what comes next" item 2): what does e4m3 quantisation of
BOTH operands of the prefill / vision linears cost in accuracy?  The oracle's `linear` is replaced by a fake-quantised
one (weights: one scale per output channel, max -> 448; activations: one scale per row, max -> 448; products
accumulated in fp32 -- what v_mfma_scale_f32_32x32x64_f8f6f4 computes), for chosen groups of layers, and the result is
compared with the bf16 oracle on the tiny golden images: vision embeddings, prefill K rows, first-token logits, and how
many greedy ids of the reference caption survive.  Decode steps keep bf16 activations (they are HBM-bound: nothing to
gain from fp8 MFMA).  Usage: [FP8_ACT_SCALE=row|tensor] python tools/fp8_numerics_study.py [tiny|0.5b]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from moondream_amd import synth
from moondream_amd.config import get_config
from oracle import moondream_oracle as O

torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
F8 = torch.float8_e4m3fn
QMAX = 448.0


def fq_rows(x):  # one scale per row
    s = x.float().abs().amax(dim=-1, keepdim=True).clamp_min(1e-12) / QMAX
    return (x.float() / s).to(F8).float() * s


def fq_tensor(x):  # one scale for the whole activation tensor (what a static, calibrated scale can do at best)
    s = x.float().abs().amax().clamp_min(1e-12) / QMAX
    return (x.float() / s).to(F8).float() * s


ACT = os.environ.get("FP8_ACT_SCALE", "row")  # row | tensor


_orig_linear = O.linear
_wq_cache = {}
MODE = {"on": False, "count": 0}


def linear_fp8(x, w, b, fast=False):
    if not MODE["on"] or x.shape[0] < 16:  # decode-regime calls (few rows) stay bf16
        return _orig_linear(x, w, b, fast)
    MODE["count"] += 1
    key = id(w)
    if key not in _wq_cache:
        _wq_cache[key] = fq_rows(w)  # per output channel (rows of [n, k])
    y = (fq_rows(x) if ACT == "row" else fq_tensor(x)) @ _wq_cache[key].t()
    if b is not None:
        y = y + b.float()
    return O._r(y)


O.linear = linear_fp8


def relrms(a, b):
    a, b = a.float(), b.float()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    cfg = get_config(name)
    gdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    g = np.load(os.path.join(gdir, {"tiny": "tiny_seed1.npz", "0.5b": "md05b_seed1.npz"}[name]))
    sd = synth.synthetic_state_dict(cfg, seed=int(g["seed"]))
    orc = O.Oracle(cfg, sd)
    n_img = len(g["image_index"])
    print(f"model {name}: {n_img} golden images, fp8 = e4m3, activation scale per {ACT}, per-channel weight scales, fp32 accumulation")
    for scope in ("vision+projector", "decoder prefill", "both"):
        rows = []
        for idx in range(n_img):
            src = int(g["image_index"][idx])
            img = synth.synthetic_image_array(src, int(g["seed"]), tuple(g[f"img{idx}.cap.size"]))
            crops = np.stack([img, img])
            prompt = g[f"img{idx}.cap.prompt"].tolist()
            ref_tokens = g[f"img{idx}.cap.tokens"].tolist()
            margins = g[f"img{idx}.cap.margins"]
            # bf16 baseline
            MODE["on"] = False
            emb0 = O.run_vision(crops, (1, 1), orc.sd, cfg, None, False)
            pos, kv0 = orc.encode_image(crops, (1, 1))
            lg0, _, _ = orc.prefill_prompt(prompt, pos, kv0.clone())
            # fp8 in the chosen scope
            MODE["on"] = scope in ("vision+projector", "both")
            emb1 = O.run_vision(crops, (1, 1), orc.sd, cfg, None, False)
            MODE["on"] = scope in ("decoder prefill", "both")
            x = torch.cat([orc.embed([cfg.tokenizer.bos_id]), emb1 if scope != "decoder prefill" else emb0], dim=0)
            kv1 = O.OracleKV.empty(cfg)
            O.text_decoder(x, orc.sd, cfg, kv1, torch.arange(x.shape[0]), orc.cos, orc.sin, None, False)
            lg1, _, p1 = orc.prefill_prompt(prompt, pos, kv1)  # the 5-row prompt pass: >= 16 rows rule keeps it bf16 ...
            MODE["on"] = False
            # ... and so do the decode steps: greedy ids from the fp8-prefilled cache
            toks, nxt, p = [], int(lg1.float().argmax()), p1
            for _ in range(len(ref_tokens)):
                toks.append(nxt)
                lg, _ = orc.decode_token(orc.embed([nxt]), p, kv1)
                lg = lg.float().clone()
                lg[cfg.tokenizer.answer_id] = -float("inf")
                nxt, p = int(lg.argmax()), p + 1
            same = 0
            while same < len(ref_tokens) and toks[same] == ref_tokens[same]:
                same += 1
            L = cfg.text.n_layers
            rows.append((relrms(emb1, emb0), relrms(kv1.k[L - 1][:, :pos], kv0.k[L - 1][:, :pos]), relrms(lg1, lg0),
                         float((lg1.float() - lg0.float()).abs().max()), same, len(ref_tokens), float(margins.min())))
        print(f"\nfp8 operands in: {scope}")
        for idx, r in enumerate(rows):
            print(f"  img{idx}: vision emb rel-RMS {r[0]:.4f}  last-layer K rel-RMS {r[1]:.4f}  first logits rel-RMS {r[2]:.4f} (max abs {r[3]:.3f})"
                  f"  greedy ids equal to the reference for {r[4]}/{r[5]} tokens (smallest reference margin {r[6]:.2f})")


if __name__ == "__main__":
    main()
