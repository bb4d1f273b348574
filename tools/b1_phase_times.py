#!/usr/bin/env python3
"""Single-image caption (B = 1), 2B: wall time of one call against the GPU-side phase times of the same call
(host tiling, vision, image / prompt prefill, decode) -- what p50_caption_latency_ms is made of."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moondream_amd import synth
from moondream_amd.config import get_config
from moondream_amd.moondream import MoondreamModel, IdTokenizer

cfg = get_config("2b")
dev = torch.device("cuda", 0)
sd = synth.synthetic_state_dict(cfg, seed=1, device=dev)
model = MoondreamModel(cfg, sd, device=dev, tokenizer=IdTokenizer(), max_batch=1)
model.compile()
img = [synth.synthetic_image(0, 1)]
prompt = [cfg.tokenizer.templates["caption"]["normal"]]
from moondream_amd import _lib


def measure(timing):
    model.collect_timing = timing
    lat, ph = [], []
    for i in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.batch_generate_ids(img, prompt, max_tokens=32, ignore_eos=True)
        torch.cuda.synchronize()
        if i >= 2:
            lat.append((time.perf_counter() - t0) * 1e3)
            if timing:
                ph.append(dict(model.last_phase_ms))
    print(f"collect_timing={timing}: wall p50 {np.median(lat):.2f} ms  min {min(lat):.2f}  max {max(lat):.2f}")
    if ph:
        med = {k: float(np.median([p[k] for p in ph])) for k in ph[0]}
        print("  GPU phases (median, ms):", {k: round(v, 3) for k, v in med.items()}, " sum", round(sum(med.values()), 2))


ab = [a for a in sys.argv[1:] if a.startswith("ab")]
if ab:  # round 5's tile rule of the single-image regime against round 2's (or "ab=1,1024": other values of the knob), interleaved on one box
    pair = [int(v) for v in ab[0].split("=")[1].split(",")] if "=" in ab[0] else [1, 0]
    for rule in pair * 2:
        _lib.check(model.lib.md_gemm_set_tuning(b"small_m_rule", rule))
        print(f"== small_m_rule = {rule} ({'round 5' if rule == 1 else 'round 2' if rule == 0 else 'threshold ' + str(rule)})")
        measure(True)
    _lib.check(model.lib.md_gemm_set_tuning(b"small_m_rule", 1))
else:
    measure(False)
    measure(True)
