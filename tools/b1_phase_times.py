#!/usr/bin/env python3
"""Phase times inside the persistent single-sequence decode kernel (md_decode_step_b1): workgroup 0's real-time stamps
at every phase boundary (measurement hook: word 768 of the sync state), averaged over the layers of one token."""
import os, sys
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
img = [synth.synthetic_image(0, 1)]
prompt = [cfg.tokenizer.templates["caption"]["normal"]]
model.batch_generate_ids(img, prompt, max_tokens=8, ignore_eos=True)
with torch.inference_mode():
    model._b1_sync[64 * 12] = 1
model.batch_generate_ids(img, prompt, max_tokens=8, ignore_eos=True)
torch.cuda.synchronize()
w = model._b1_sync.cpu().numpy().astype(np.uint32)
L = cfg.text.n_layers
n = 1 + 6 * L
t = np.array([int(w[2048 + 2 * i]) | (int(w[2049 + 2 * i]) << 32) for i in range(n)], dtype=np.int64) * 10  # ns (100 MHz)
d = np.diff(t)
names = ["phase A (ln + qkv|fc1 rows)", "barrier 1 (+ K/V rows requested)", "phase B (attention partials)", "barrier 2 (+ fc2 rows)", "phase C (combine + proj rows)", "barrier 3 (+ next rows requested)"]
print(f"one token, {L} layers: total {(t[-1] - t[0]) / 1e3:.1f} us (the lm_head phase is not stamped)")
for k in range(6):
    print(f"  {names[k]:36s} mean {d[k::6].mean() / 1e3:6.2f} us   (layer 1: {d[6 + k] / 1e3:6.2f})")
