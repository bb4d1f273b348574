#!/usr/bin/env python3
"""Round 5 (advisor, round 4): what the PINNED tile policy costs small batches.  Since round 4 every call with >= 2 sequences
runs its > 64-row launches on the 256x256 four-wave kernel whatever the row count (MD_TILE_PINNED: batch == sequential bit for
bit); a batch of 2..8 images makes few such tiles (B = 2: 1460 prefill rows -> 6 x 8 tiles for N = 2048 on 256 CUs), where the
shape-based choice (MD_TILE_BY_SHAPE) would take 64x64 / 128x128 tiles.  Times batch_generate_ids(B images, caption prompt, 32
tokens) at B = 1, 2, 4, 8, 16 under both policies (same process, interleaved) and reports whether the ids agree."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from moondream_amd import _lib, synth
from moondream_amd.config import get_config
from moondream_amd.moondream import IdTokenizer, MoondreamModel

cfg = get_config("2b")
sd = synth.synthetic_state_dict(cfg, seed=1, device="cuda")
model = MoondreamModel(cfg, sd, device="cuda", tokenizer=IdTokenizer(), max_batch=16)
model.single_sequence_kernel = False  # B = 1 on the batched kernels too: the comparison is about tile choice only
prompt = cfg.tokenizer.templates["caption"]["normal"]
T = 32
pinned_select = MoondreamModel._select_kernels


def by_shape_select(self, n):
    self._tile_policy = _lib.MD_TILE_BY_SHAPE
    self.w.vit.tile_policy = self.w.text.tile_policy = _lib.MD_TILE_BY_SHAPE


def run(b, policy, reps=5):
    MoondreamModel._select_kernels = pinned_select if policy == "pinned" else by_shape_select
    imgs = [synth.synthetic_image(i, 1) for i in range(b)]
    ids = model.batch_generate_ids(imgs, [prompt] * b, max_tokens=T, ignore_eos=True)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.batch_generate_ids(imgs, [prompt] * b, max_tokens=T, ignore_eos=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, ids


print("B   pinned ms   by-shape ms   pinned/by-shape   ids equal   (2B, caption prompt, 32 greedy tokens, median of 5, interleaved twice)")
for b in (1, 2, 4, 8, 16):
    res = {}
    for rep in range(2):
        for pol in ("pinned", "by_shape"):
            ms, ids = run(b, pol)
            res.setdefault(pol, []).append(ms)
            res[pol + "_ids"] = ids
    p, s = min(res["pinned"]), min(res["by_shape"])
    print(f"{b:<3d} {p:9.2f}   {s:11.2f}   {p / s:15.3f}   {res['pinned_ids'] == res['by_shape_ids']}", flush=True)
MoondreamModel._select_kernels = pinned_select
