#!/usr/bin/env python3
"""Round 6: does the planted image code survive the opt-in fp8 mode?  2B, 64 bench images: ids of the bf16 mode vs the fp8-full mode
(e4m3 operands for ViT / projector / prefill GEMMs + e4m3 decode weights), and detect objects of 8 13-crop images in both modes."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moondream_amd import synth
from moondream_amd.config import get_config
from moondream_amd.moondream import MoondreamModel, IdTokenizer

cfg = get_config("2b")
sd = synth.synthetic_state_dict(cfg, seed=1, device="cuda")
model = MoondreamModel(cfg, sd, device="cuda", tokenizer=IdTokenizer(), max_batch=64)
imgs = [synth.synthetic_image(i, 1) for i in range(64)]
pr = cfg.tokenizer.templates["caption"]["normal"]
a = model.batch_generate_ids(imgs, [pr] * 64, max_tokens=32, ignore_eos=True)
big = [synth.synthetic_image(i, 1, (768, 1024)) for i in range(8)]
da = model.batch_detect(big, ["7 8"] * 8, settings={"max_objects": 4})
want = [[synth.region_anchor(synth.image_code_bits(i), w) for w in ("x_first", "y", "w", "h")] for i in range(8)]
model.enable_fp8(imgs[:8], pr)
b = model.batch_generate_ids(imgs, [pr] * 64, max_tokens=32, ignore_eos=True)
db = model.batch_detect(big, ["7 8"] * 8, settings={"max_objects": 4})
same = sum(x == y for x, y in zip(a, b))
first = [next((t for t in range(32) if x[t] != y[t]), 32) for x, y in zip(a, b)]
print(json.dumps({"ids_fp8_equal_bf16": same, "of": 64, "first_divergence_hist": {str(k): first.count(k) for k in sorted(set(first))},
                  "detect_objects_equal": sum(x["objects"] == y["objects"] for x, y in zip(da, db)), "detect_of": 8,
                  "bf16_first_object_centre_bins": [[round((o["objects"][0]["x_min"] + o["objects"][0]["x_max"]) * 512), round((o["objects"][0]["y_min"] + o["objects"][0]["y_max"]) * 512)] for o in da],
                  "expected_anchor_bins": [w[:2] for w in want]}))
