#!/usr/bin/env python3
"""Phase times (vision / prefill / decode, ms) of one eager batch_generate_ids call at several batch sizes, 2B, 32 tokens."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moondream_amd import synth
from moondream_amd.config import get_config
from moondream_amd.moondream import MoondreamModel, IdTokenizer
cfg = get_config("2b")
sd = synth.synthetic_state_dict(cfg, seed=1, device="cuda")
model = MoondreamModel(cfg, sd, device="cuda", tokenizer=IdTokenizer(), max_batch=128)
pr = cfg.tokenizer.templates["caption"]["normal"]
model.collect_timing = True
for graphs in (False, True):
    model.use_graphs = graphs
    for b in [int(x) for x in (sys.argv[1:] or ["64", "128", "96"])]:
        imgs = [synth.synthetic_image(i, 1) for i in range(b)]
        for _ in range(3):
            model.batch_generate_ids(imgs, [pr] * b, max_tokens=32, ignore_eos=True)
        ph = model.last_phase_ms
        print(json.dumps({"graphs": graphs, "batch": b, **{k: round(v, 2) for k, v in ph.items()}, "decode_ms_per_token_per_64": round(ph["decode"] / 32 * 64 / b, 3)}), flush=True)
