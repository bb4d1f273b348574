#!/usr/bin/env python3
"""Is test_pipelined_batches_equal_sequential deterministic?  Runs its body N times, eager and graph-replayed, and
reports every (repeat, mode, batch, sequence, first differing position)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from PIL import Image

from moondream_amd import synth
from moondream_amd.config import get_config
from moondream_amd.moondream import MoondreamModel, IdTokenizer

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tiny_seed1.npz"))
cfg = get_config("tiny")
sd = synth.synthetic_state_dict(cfg, seed=int(g["seed"]), device="cuda")
model = MoondreamModel(cfg, sd, device="cuda", tokenizer=IdTokenizer(), max_batch=4)


def img(i):
    return Image.fromarray(synth.synthetic_image_array(int(g["image_index"][i]), int(g["seed"]), tuple(g[f"img{i}.cap.size"])), "RGB")


MODE = os.environ.get("MD_REPEAT_MODE", "")
if MODE == "one_stream":
    st = torch.cuda.Stream()
    model._pipe_streams = (st, st)
elif MODE == "no_priority":
    model._pipe_streams = (torch.cuda.Stream(), torch.cuda.Stream())
elif MODE == "sync_before_decode":
    _orig = model._decode_greedy

    def _synced(*a, **k):
        torch.cuda.synchronize()
        return _orig(*a, **k)

    model._decode_greedy = _synced
elif MODE == "sync_after_decode":
    _orig = model._decode_greedy

    def _synced(*a, **k):
        r = _orig(*a, **k)
        torch.cuda.synchronize()
        return r

    model._decode_greedy = _synced
print("mode:", MODE or "default", flush=True)
images = [img(i) for i in range(3)]
prompts = [g[f"img{i}.cap.prompt"].tolist() for i in range(3)]
n = len(g["img0.cap.tokens"])
ref = [g[f"img{i}.cap.tokens"].tolist() for i in range(3)]
batches = [(images, prompts), (images[::-1], prompts[::-1]), (images[:2] + images[:1], prompts[:2] + prompts[:1]), (images, prompts)]
want = [ref, ref[::-1], ref[:2] + ref[:1], ref]
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    for use_graphs in (False, True):
        model.use_graphs = use_graphs
        if MODE == "fresh_graphs":
            torch.cuda.synchronize()
            model._graphs.clear()
        outs = list(model.batch_generate_ids_pipelined(batches, max_tokens=n))
        model.use_graphs = False
        for bi, (o, w) in enumerate(zip(outs, want)):
            for si, (a, b) in enumerate(zip(o, w)):
                if a != b:
                    j = next(t for t in range(len(b)) if t >= len(a) or a[t] != b[t])
                    print(f"rep {rep} graphs={use_graphs} batch {bi} seq {si}: first difference at {j}: got {a[j:j+3]} want {b[j:j+3]}", flush=True)
                    bad += 1
    # the same four batches, one after the other on one stream
    for bi, ((im, pr), w) in enumerate(zip(batches, want)):
        o = model.batch_generate_ids(im, pr, max_tokens=n)
        for si, (a, b) in enumerate(zip(o, w)):
            if a != b:
                print(f"rep {rep} SEQUENTIAL batch {bi} seq {si} differs", flush=True)
                bad += 1
print(f"repeat_pipelined: {bad} mismatching sequences")
