#!/usr/bin/env python3
"""Round 6: the drop-in executed and timed.  The UNMODIFIED reference MoondreamModel (MOONDREAM_REFERENCE checkout, 2B shapes,
synthetic checkpoint) on cuda:0, captioning the same image three ways:
  (a) as it is            -- its own seam: ATen / torch-ROCm eager (moondream.py:168-192)
  (b) bind_reference(...) -- the same object, the four seam attributes rebound to libmoondream_hip.so
  (c) moondream_amd.MoondreamModel.caption -- this package's mirror of the class (device-resident loop)
and prints one JSON line: ids of (a), (b), (c), ms per caption and per decoded token, and what the reference's own
Python loop costs per token on top of the library (b's per-token time minus the library's GPU time for one seam call)."""
import json
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))


def main():
    import make_golden as mg
    from moondream_amd import synth
    from moondream_amd.config import get_config
    from moondream_amd.integration import bind_reference
    from moondream_amd.moondream import MoondreamModel

    cfg_name = os.environ.get("DROPIN_CFG", "2b")
    n_tok = int(os.environ.get("DROPIN_TOKENS", "32"))
    g = np.load(os.path.join(REPO, "tests", "golden", "md2b_seed1.npz" if cfg_name == "2b" else "tiny_seed1.npz"))
    seed = int(g["seed"])
    cfg = get_config(cfg_name)
    sd = synth.synthetic_state_dict(cfg, seed=seed)
    model, ref_md = mg.load_reference(cfg, sd)
    model = model.to("cuda:0")
    image = Image.fromarray(synth.synthetic_image_array(int(g["image_index"][0]), seed, (378, 378)), "RGB")
    want = g["img0.cap.tokens"].tolist()[:n_tok]
    settings = {"temperature": 0, "max_tokens": n_tok, "variant": None}

    def ids(text):
        return [int(t) for t in text.split()]

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return out, float(np.median(ts))

    def ref_caption():
        return ids(model.caption(image, settings=settings)["caption"])

    def ref_encode():
        return model.encode_image(image, settings)

    out = {"config": cfg_name, "tokens": n_tok, "reference_ids": want}
    a_ids, a_ms = timed(ref_caption, 3)
    _, a_enc = timed(ref_encode, 3)
    out["a_reference_own_seam_aten_rocm"] = {"ids_equal_golden": a_ids == want, "caption_ms": a_ms, "encode_ms": a_enc,
                                            "ms_per_token": (a_ms - a_enc) / n_tok}
    b = bind_reference(model)
    b_ids, b_ms = timed(ref_caption, 5)
    _, b_enc = timed(ref_encode, 5)
    # GPU time of one bound decode step (seam call alone, no reference loop around it)
    enc = model.encode_image(image, settings)
    model.load_encoded_image(enc)
    x = torch.zeros(1, 1, cfg.text.dim, dtype=torch.bfloat16, device="cuda:0")
    mask = torch.zeros(1, 1, cfg.text.max_context, dtype=torch.bool, device="cuda:0")
    mask[:, :, : enc.pos + 6] = 1
    pos_ids = torch.tensor([enc.pos + 5], device="cuda:0")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.inference_mode():
        for _ in range(3):
            model._decode_one_tok(x, mask, pos_ids, None)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            model._decode_one_tok(x, mask, pos_ids, None)
        e1.record()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            model._decode_one_tok(x, mask, pos_ids, None)
        torch.cuda.synchronize()
        seam_wall = (time.perf_counter() - t0) * 1e3 / 20
    seam_gpu = e0.elapsed_time(e1) / 20
    per_tok = (b_ms - b_enc) / n_tok
    out["b_reference_bound_to_libmoondream_hip"] = {
        "ids_equal_golden": b_ids == want, "ids_equal_a": b_ids == a_ids, "caption_ms": b_ms, "encode_ms": b_enc,
        "ms_per_token": per_tok, "seam_decode_call_ms_back_to_back": seam_gpu, "seam_decode_call_wall_ms": seam_wall,
        "reference_loop_host_cost_ms_per_token": per_tok - seam_wall, "seam_calls": dict(b.calls),
    }
    b.unbind()
    del b
    mirror = MoondreamModel(cfg, sd, device="cuda:0")
    def mirror_caption():
        return ids(mirror.caption(image, settings={"temperature": 0, "max_tokens": n_tok})["caption"])
    c_ids, c_ms = timed(mirror_caption, 5)
    out["c_mirror_class"] = {"ids_equal_golden": c_ids == want, "caption_ms": c_ms}
    out["speedup_b_over_a"] = a_ms / b_ms
    print(json.dumps(out))


if __name__ == "__main__":
    main()
