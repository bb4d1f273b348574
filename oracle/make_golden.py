#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation.

TEST INFRASTRUCTURE.  Runs only in the build container, where the read-only
reference checkout exists (``/root/reference``); the GPU box never runs this.
It imports ``moondream.torch`` from that checkout unmodified, with the two
shims SURVEY.md section 8c describes:

  1. ``Tokenizer.from_pretrained`` is replaced by an id-echo stub (the real
     vocabulary needs the network); the hot path is driven with token IDs.
  2. all parameters come from ``moondream_amd.synth.synthetic_state_dict``
     (no checkpoint is available offline).

and records, for fixed seeded inputs, the per-stage activations, KV rows,
logits, greedy token ids and top-1/top-2 margins that the oracle
(``oracle/moondream_oracle.py``) and the HIP path are compared against.

Usage:  python oracle/make_golden.py [tiny] [layers] [multicrop] [crops] [textonly] [detect] [sampling] [reasoning] [lora] [0.5b] [2b] [bench64] [vqa64] [detect13] [reftime]
"""
from __future__ import annotations

import os
import sys
import time
import zlib

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("MOONDREAM_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REFERENCE)

from moondream_amd.config import get_config  # noqa: E402
from moondream_amd import synth  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


class EchoTokenizer:
    """decode() prints ids separated by spaces so generated ids can be parsed
    back out of the reference's streamed text."""

    class _Enc:
        def __init__(self, ids):
            self.ids = ids

    def encode(self, s):
        return self._Enc([int(t) for t in s.split()])

    def decode(self, ids):
        return "".join(f"{int(i)} " for i in ids)


def load_reference(cfg, sd):
    import moondream.torch.moondream as ref_md
    from moondream.torch.config import MoondreamConfig as RefConfig

    ref_md.Tokenizer.from_pretrained = staticmethod(lambda *_a, **_k: EchoTokenizer())
    model = ref_md.MoondreamModel(RefConfig.from_dict(cfg.to_dict()), dtype=torch.bfloat16)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "kv_cache" not in m]
    assert not missing and not unexpected, (missing, unexpected)
    return model, ref_md


def bf16_bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().contiguous().view(torch.int16).numpy().copy()


def run_reference_caption(model, ref_md, image, prompt_ids, max_tokens):
    """Greedy generation through the reference's own _generate_answer with
    taps on the seam methods.  Returns dict of recorded arrays."""
    from PIL import Image

    rec = {"decode_logits": [], "prefill_hidden": []}
    orig_decode, orig_prefill, orig_lm_head = model._decode_one_tok, model._prefill, ref_md.lm_head

    def decode_tap(x, mask, pos_ids, lora):
        logits, hidden = orig_decode(x, mask, pos_ids, lora)
        rec["decode_logits"].append(logits[0].clone())
        return logits, hidden

    def prefill_tap(x, mask, pos_ids, lora):
        h = orig_prefill(x, mask, pos_ids, lora)
        rec["prefill_hidden"].append(h[0].clone())
        return h

    def lm_head_tap(h, w):
        out = orig_lm_head(h, w)
        rec.setdefault("prompt_logits", out[0].clone())  # first call = the prompt prefill
        return out

    model._decode_one_tok, model._prefill, ref_md.lm_head = decode_tap, prefill_tap, lm_head_tap
    try:
        t0 = time.perf_counter()
        enc = model.encode_image(Image.fromarray(image, "RGB"))
        t_enc = time.perf_counter() - t0
        model.load_encoded_image(enc)
        toks = torch.tensor([prompt_ids], device=model.device)
        t0 = time.perf_counter()
        text = "".join(
            model._generate_answer(toks, enc.pos, {"temperature": 0, "max_tokens": max_tokens})
        )
        t_gen = time.perf_counter() - t0
    finally:
        model._decode_one_tok, model._prefill, ref_md.lm_head = orig_decode, orig_prefill, orig_lm_head
    tokens = [int(t) for t in text.split()]
    answer_id = model.config.tokenizer.answer_id
    steps = [rec["prompt_logits"]]
    for lg in rec["decode_logits"]:
        lg = lg.clone()
        lg[answer_id] = float("-inf")
        steps.append(lg)
    margins, argmaxes = [], []
    for lg in steps:
        top = torch.topk(lg.float(), 2).values
        margins.append(float(top[0] - top[1]))
        argmaxes.append(int(torch.argmax(lg.float())))
    return dict(
        enc=enc,
        tokens=tokens,
        margins=margins,
        argmaxes=argmaxes,
        steps=steps,
        image_prefill_hidden=rec["prefill_hidden"][0],
        prompt_hidden=rec["prefill_hidden"][1],
        t_enc=t_enc,
        t_gen=t_gen,
    )


def reference_vision_taps(model, ref_md, image):
    """Stage-by-stage ViT activations using the reference's own functions."""
    from PIL import Image
    from moondream.torch.vision import prepare_crops, create_patches
    from moondream.torch.layers import attn, layer_norm, mlp
    from moondream.torch.image_crops import reconstruct_from_crops

    cfg = model.config.vision
    w = model.vision
    taps = {}
    with torch.inference_mode():
        crops, tiling = prepare_crops(Image.fromarray(image, "RGB"), cfg, device="cpu")
        taps["crops_norm"] = crops
        x = w.patch_emb(create_patches(crops, cfg.enc_patch_size)) + w.pos_emb
        taps["vit.embed"] = x
        for i, block in enumerate(w.blocks):
            x = x + attn(layer_norm(x, block.ln1), block.attn, n_heads=cfg.enc_n_heads)
            x = x + mlp(layer_norm(x, block.ln2), block.mlp)
            if i in (0, cfg.enc_n_layers - 1):
                taps[f"vit.block{i}"] = x
        x = layer_norm(x, w.post_ln)
        taps["vit.out"] = x
        full = model._vis_enc(crops)
        assert torch.equal(full, x), "step-by-step ViT differs from vision_encoder()"
        local = x[1:].view(-1, cfg.enc_n_layers, cfg.enc_n_layers, cfg.enc_dim)
        rec = reconstruct_from_crops(local, tiling, patch_size=1, overlap_margin=cfg.overlap_margin)
        taps["vis.proj"] = model._vis_proj(x[0], rec)
    return taps, tiling


def model_pixel_lut(model):
    """What the reference's prepare_crops makes of each of the 256 byte values."""
    from PIL import Image
    from moondream.torch.vision import prepare_crops

    ramp = np.zeros((378, 378, 3), dtype=np.uint8)
    ramp[0, :256, 0] = np.arange(256)
    crops, _ = prepare_crops(Image.fromarray(ramp, "RGB"), model.config.vision, device="cpu")
    return crops[0, 0, 0, :256].clone()


def gen_model_case(name, cfg_name, seed, image_sizes, max_tokens, full_tensors, n_images=1, min_margin=1.0):
    cfg = get_config(cfg_name)
    t0 = time.perf_counter()
    sd = synth.synthetic_state_dict(cfg, seed=seed)
    print(f"[{name}] synthetic weights in {time.perf_counter()-t0:.1f}s", flush=True)
    model, ref_md = load_reference(cfg, sd)
    caption_ids = cfg.tokenizer.templates["caption"]["normal"]
    out = {"seed": np.int64(seed), "cfg": np.array(cfg_name)}
    # Keep only images on which EVERY greedy decision of the reference has a
    # top-1/top-2 margin >= min_margin (several bf16 ulps at these logit
    # magnitudes): on those, token ids are a well-posed integer output that two
    # correct bf16 implementations must agree on bit-for-bit.
    idx, src = 0, -1
    image_index = []
    while idx < n_images:
        src += 1
        assert src < 400, "could not find enough wide-margin images"
        size = image_sizes[idx % len(image_sizes)]
        image = synth.synthetic_image_array(src, seed, size)
        first = run_reference_caption(model, ref_md, image, caption_ids, max_tokens)
        if min(first["margins"]) < min_margin:
            print(f"[{name}] skip image {src}: min margin {min(first['margins']):.3f}", flush=True)
            continue
        image_index.append(src)
        for kind, prompt in (("cap", caption_ids), ("vqa", synth.synthetic_vqa_prompt(cfg, src, seed))):
            if kind == "vqa" and idx > 0:
                continue
            r = first if kind == "cap" else run_reference_caption(model, ref_md, image, prompt, max_tokens)
            p = f"img{idx}.{kind}."
            out[p + "size"] = np.array(size)
            out[p + "prompt"] = np.array(prompt)
            out[p + "tokens"] = np.array(r["tokens"])
            out[p + "margins"] = np.array(r["margins"], dtype=np.float32)
            out[p + "argmaxes"] = np.array(r["argmaxes"])
            out[p + "pos"] = np.int64(r["enc"].pos)
            if full_tensors:
                out[p + "step_logits"] = bf16_bits(torch.stack(r["steps"]))
            else:
                top = torch.topk(torch.stack(r["steps"]).float(), 8, dim=-1)
                out[p + "top8_val"] = top.values.numpy()
                out[p + "top8_idx"] = top.indices.numpy()
            print(
                f"[{name}] img{idx} {kind}: {len(r['tokens'])} tokens, min margin "
                f"{min(r['margins']):.4f}, encode {r['t_enc']:.2f}s gen {r['t_gen']:.2f}s",
                flush=True,
            )
            if kind == "cap":
                # sampled KV rows written by the image prefill (layer 0 and last)
                L = cfg.text.n_layers
                for li in (0, L - 1):
                    k, v = r["enc"].caches[li]
                    out[p + f"k{li}"] = bf16_bits(k[0, :, ::37])
                    out[p + f"v{li}"] = bf16_bits(v[0, :, ::37])
                rs = 1 if full_tensors else 23
                out[p + "image_prefill_hidden"] = bf16_bits(r["image_prefill_hidden"][::rs])
                out[p + "prompt_hidden"] = bf16_bits(r["prompt_hidden"])
                out["kv_row_stride"] = np.int64(37)
                out["hidden_row_stride"] = np.int64(rs)
        if idx == 0:
            taps, tiling = reference_vision_taps(model, ref_md, image)
            out["img0.tiling"] = np.array(tiling)
            out["img0.crops_norm_lut"] = bf16_bits(
                model_pixel_lut(model)
            )
            for k, t in taps.items():
                if k == "crops_norm":
                    out["img0.crops_norm_crc"] = np.int64(zlib.crc32(bf16_bits(t).tobytes()))
                    continue
                if full_tensors and k in ("vit.out", "vis.proj"):
                    out["img0." + k] = bf16_bits(t)
                elif t.dim() == 3:
                    out["img0." + k] = bf16_bits(t[:, ::9] if full_tensors else t[:, ::31, ::5])
                else:
                    out["img0." + k] = bf16_bits(t[::9] if full_tensors else t[::31, ::5])
            out["vit_token_stride"] = np.int64(9 if full_tensors else 31)
            out["vit_feat_stride"] = np.int64(1 if full_tensors else 5)
        idx += 1
    out["image_index"] = np.array(image_index)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)", flush=True)


def gen_multicrop():
    """Tiny model on non-square images: exercises tiling, stitch and the
    non-trivial adaptive pool."""
    cfg = get_config("tiny")
    sd = synth.synthetic_state_dict(cfg, seed=3)
    model, ref_md = load_reference(cfg, sd)
    out = {}
    for i, size in enumerate([(500, 700), (420, 1000), (900, 640)]):
        image = synth.synthetic_image_array(i, 3, size)
        taps, tiling = reference_vision_taps(model, ref_md, image)
        out[f"case{i}.size"] = np.array(size)
        out[f"case{i}.tiling"] = np.array(tiling)
        out[f"case{i}.vis.proj"] = bf16_bits(taps["vis.proj"])
        out[f"case{i}.vit.out"] = bf16_bits(taps["vit.out"][:, ::27])
        print(f"[multicrop] {size} -> tiling {tiling}", flush=True)
    path = os.path.join(GOLD, "tiny_multicrop.npz")
    np.savez_compressed(path, **out)
    print(f"[multicrop] wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)")


def gen_crops():
    """Host-side integer work: select_tiling table and crop checksums from the
    reference's PIL-LANCZOS branch (pyvips is not installed here).
    reference: image_crops.py:17-167."""
    from moondream.torch.image_crops import select_tiling, overlap_crop_image, HAS_VIPS

    assert not HAS_VIPS
    sizes = []
    for h in list(range(1, 1400, 37)) + [266, 267, 378, 379, 532, 533, 644, 645, 2048, 4000]:
        for w in list(range(1, 1400, 41)) + [266, 267, 378, 379, 532, 533, 644, 645, 2048, 4000]:
            sizes.append((h, w))
    table = np.array(
        [[h, w, *select_tiling(h, w, 266, 12)] for (h, w) in sizes], dtype=np.int32
    )
    out = {"tiling_table": table}
    img_sizes = [(378, 378), (300, 200), (800, 600), (480, 640), (768, 1024), (1000, 333), (97, 1300), (1500, 1500)]
    crcs = []
    for i, (h, w) in enumerate(img_sizes):
        img = synth.synthetic_image_array(i, 11, (h, w))
        r = overlap_crop_image(img, overlap_margin=4, max_crops=12)
        crcs.append([h, w, r["tiling"][0], r["tiling"][1], len(r["crops"]), zlib.crc32(r["crops"].tobytes())])
        print(f"[crops] {(h, w)} -> tiling {r['tiling']} n={len(r['crops'])}")
    out["crop_cases"] = np.array(crcs, dtype=np.int64)
    path = os.path.join(GOLD, "image_crops.npz")
    np.savez_compressed(path, **out)
    print(f"[crops] wrote {path}")


def gen_textonly(name="tiny_textonly", cfg_name="tiny", seed=1, n_cases=3, max_tokens=12, min_margin=1.0):
    """Text-only query through the reference's own public `query(image=None, ...)`
    (moondream.py:564-575: BOS + query prefix, pos 0, plain causal mask)."""
    cfg = get_config(cfg_name)
    sd = synth.synthetic_state_dict(cfg, seed=seed)
    model, ref_md = load_reference(cfg, sd)
    rng = np.random.default_rng(4242 + seed)
    out = {"seed": np.int64(seed), "cfg": np.array(cfg_name)}
    kept, tries = 0, 0
    while kept < n_cases:
        tries += 1
        assert tries < 200, "could not find enough wide-margin questions"
        q = rng.integers(10, min(50000, cfg.text.vocab_size), int(rng.integers(3, 9))).tolist()
        rec = {"decode_logits": []}
        orig_decode, orig_lm_head = model._decode_one_tok, ref_md.lm_head

        def decode_tap(x, mask, pos_ids, lora):
            logits, hidden = orig_decode(x, mask, pos_ids, lora)
            rec["decode_logits"].append(logits[0].clone())
            return logits, hidden

        def lm_head_tap(h, w):
            o = orig_lm_head(h, w)
            rec.setdefault("prompt_logits", o[0].clone())
            return o

        model._decode_one_tok, ref_md.lm_head = decode_tap, lm_head_tap
        try:
            ans = model.query(None, " ".join(str(t) for t in q), settings={"temperature": 0, "max_tokens": max_tokens})["answer"]
        finally:
            model._decode_one_tok, ref_md.lm_head = orig_decode, orig_lm_head
        tokens = [int(t) for t in ans.split()]
        steps = [rec["prompt_logits"]]
        for lg in rec["decode_logits"]:
            lg = lg.clone()
            lg[cfg.tokenizer.answer_id] = float("-inf")
            steps.append(lg)
        margins = []
        for lg in steps:
            top = torch.topk(lg.float(), 2).values
            margins.append(float(top[0] - top[1]))
        if min(margins) < min_margin:
            print(f"[{name}] skip question {q}: min margin {min(margins):.3f}", flush=True)
            continue
        pfx = f"q{kept}."
        out[pfx + "question"] = np.array(q)
        out[pfx + "tokens"] = np.array(tokens)
        out[pfx + "margins"] = np.array(margins, dtype=np.float32)
        out[pfx + "step_logits"] = bf16_bits(torch.stack(steps))
        print(f"[{name}] q{kept}: question {q} -> {tokens} (min margin {min(margins):.3f})", flush=True)
        kept += 1
    out["n_cases"] = np.int64(kept)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)", flush=True)


def gen_bench64(name="md2b_bench64", cfg_name="2b", seed=1, n_images=64, max_tokens=32, prompt_kind="caption"):
    """The TIMED configuration (BASELINE.json configs[2]): the reference itself on the exact 64
    seed-1 378x378 images and caption prompt that bench.py times, Moondream-2B, greedy, 32 tokens.
    No survivorship filter: every image is kept, with the reference's top-1/top-2 margin of every
    decision, so bench.py / the GPU tests can compare the batched HIP ids margin-aware
    (reference: moondream.py:434-539 behind caption(), moondream.py:625-651)."""
    cfg = get_config(cfg_name)
    sd = synth.synthetic_state_dict(cfg, seed=seed)
    model, ref_md = load_reference(cfg, sd)
    caption_ids = cfg.tokenizer.templates["caption"]["normal"]
    # prompt_kind "vqa32" (round 5, BASELINE configs[1] at bench scale): the 32-id question prompts bench.py's vqa32 leg
    # times (synth.synthetic_vqa_prompt: query prefix + 27 seeded ids + suffix, reference moondream.py:564,586-591,604),
    # one per image, through the same _generate_answer the reference's query() drives (moondream.py:606-618)
    prompts = [caption_ids if prompt_kind == "caption" else synth.synthetic_vqa_prompt(cfg, i, seed) for i in range(n_images)]
    toks, margins, top_v, top_i, t_enc, t_gen = [], [], [], [], [], []
    for i in range(n_images):
        image = synth.synthetic_image_array(i, seed, (378, 378))
        r = run_reference_caption(model, ref_md, image, prompts[i], max_tokens)
        assert len(r["tokens"]) == max_tokens, (i, len(r["tokens"]))  # no EOS inside the window
        toks.append(r["tokens"])
        margins.append(r["margins"])
        top = torch.topk(torch.stack(r["steps"]).float(), 8, dim=-1)
        top_v.append(top.values.numpy())
        top_i.append(top.indices.numpy().astype(np.int32))
        t_enc.append(r["t_enc"])
        t_gen.append(r["t_gen"])
        print(f"[{name}] image {i}: min margin {min(r['margins']):.4f} encode {r['t_enc']:.2f}s gen {r['t_gen']:.2f}s", flush=True)
    out = {
        "seed": np.int64(seed), "cfg": np.array(cfg_name),
        "prompt": np.array(caption_ids if prompt_kind == "caption" else prompts),   # caption: [5]; vqa32: [n_images, 32]
        "tokens": np.array(toks, dtype=np.int32), "margins": np.array(margins, dtype=np.float32),
        "top8_val": np.array(top_v, dtype=np.float32), "top8_idx": np.array(top_i, dtype=np.int32),
    }
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)", flush=True)


def gen_detect(name="tiny_detect", cfg_name="tiny", seed=1, n_cases=2, max_objects=3, min_margin=4.0):
    """Region head + points loop through the reference's own public ``detect`` / ``point``
    (moondream.py:735-829 -> _generate_points moondream.py:653-733 -> region.py:12-93), with taps on
    the four region functions: every call's input, output and (for the decoders) the top-1/top-2
    margin of each argmax IN BF16 ULPS of the top logit.  With i.i.d. synthetic weights the 1024-bin
    heads separate their best two bins by a few ulps only, so: a case is kept when every decision of
    its FIRST object (x, y, w, h bins and the next token) has margin >= min_margin ulps, all margins
    are recorded, and consumers compare objects up to the first narrow decision.  Also one ``query`` with spatial_refs (region.py:96-136 +
    moondream.py:293-301: coordinate / size embeddings injected into the prompt)."""
    import moondream.torch.moondream as ref_md_mod
    from PIL import Image

    cfg = get_config(cfg_name)
    sd = synth.synthetic_state_dict(cfg, seed=seed)
    model, ref_md = load_reference(cfg, sd)
    out = {"seed": np.int64(seed), "cfg": np.array(cfg_name), "max_objects": np.int64(max_objects), "object_ids": np.array([7, 8])}
    names = ("decode_coordinate", "encode_coordinate", "decode_size", "encode_size")
    orig = {n: getattr(ref_md, n) for n in names}
    orig_decode_tok = model._decode_one_tok

    def run(kind, image):
        rec = {n: [] for n in names}
        rec["next_logits"] = []

        def tap(n):
            def f(x, w):
                y = orig[n](x, w)
                rec[n].append((x.detach().clone(), y.detach().clone()))
                return y
            return f

        def decode_tok_tap(x, mask, pos_ids, lora):
            logits, hidden = orig_decode_tok(x, mask, pos_ids, lora)
            rec["next_logits"].append(logits[0].detach().clone())
            return logits, hidden

        for n in names:
            setattr(ref_md, n, tap(n))
        model._decode_one_tok = decode_tok_tap
        try:
            fn = model.detect if kind == "detect" else model.point
            # "variant": None -- the reference's encode_image indexes settings["variant"] (moondream.py:241-243)
            res = fn(Image.fromarray(image, "RGB"), "7 8", settings={"max_objects": max_objects, "variant": None})
        finally:
            for n in names:
                setattr(ref_md, n, orig[n])
            model._decode_one_tok = orig_decode_tok
        return res, rec

    def margin(lg):
        """top-1/top-2 gap in units of the bf16 spacing at the top logit (8 significand bits)."""
        top = torch.topk(lg.float().reshape(-1), 2).values
        ulp = 2.0 ** (np.floor(np.log2(max(abs(float(top[0])), 2.0 ** -120))) - 7)
        return float(top[0] - top[1]) / ulp

    def eos_margin(lg):
        """The loop only asks whether the next token is EOS (moondream.py:668-671): the margin of THAT decision is the gap
        between the best non-EOS logit and the EOS logit (in bf16 ulps of the larger), not the gap between the best two
        tokens -- which token it is has no effect on the objects."""
        lg = lg.float().reshape(-1)
        eos = float(lg[cfg.tokenizer.eos_id])
        rest = lg.clone()
        rest[cfg.tokenizer.eos_id] = float("-inf")
        best = float(rest.max())
        ulp = 2.0 ** (np.floor(np.log2(max(abs(best), abs(eos), 2.0 ** -120))) - 7)
        return abs(best - eos) / ulp

    for kind in ("detect", "point"):
        kept, src = 0, -1
        while kept < n_cases:
            src += 1
            assert src < 400, "could not find enough wide-margin detect cases"
            image = synth.synthetic_image_array(src, seed, (378, 378))
            res, rec = run(kind, image)
            objs = res["objects" if kind == "detect" else "points"]
            # decisions in loop order: per object x, y, (w, h,) next-token
            per_obj = 3 if kind == "detect" else 2  # _decode_one_tok calls per object; the last decides the next token
            nxt = rec["next_logits"][per_obj - 1 :: per_obj]
            margins = []
            for k in range(len(objs)):
                margins += [margin(rec["decode_coordinate"][2 * k][1]), margin(rec["decode_coordinate"][2 * k + 1][1])]
                if kind == "detect":
                    sz = rec["decode_size"][k][1]
                    margins += [margin(sz[0]), margin(sz[1])]
                margins.append(eos_margin(nxt[k]))
            per_dec = len(margins) // max(1, len(objs))
            # with i.i.d. synthetic weights the 1024-bin heads have top-1/top-2 gaps of a few bf16 ulps on
            # most decisions; keep images whose FIRST object (all its decisions) is wide-margin and record
            # every margin, so consumers compare objects up to the first narrow decision
            if len(objs) == 0 or min(margins[:per_dec]) < min_margin:
                print(f"[{name}] {kind}: skip image {src}: {len(objs)} objects, first-object min margin {min(margins[:per_dec]) if margins else 0:.3f}", flush=True)
                continue
            pfx = f"{kind}{kept}."
            out[pfx + "image_index"] = np.int64(src)
            keys = ("x_min", "y_min", "x_max", "y_max") if kind == "detect" else ("x", "y")
            out[pfx + "objects"] = np.array([[o[k] for k in keys] for o in objs], dtype=np.float64)
            out[pfx + "margins"] = np.array(margins, dtype=np.float32).reshape(len(objs), per_dec)
            out[pfx + "next_tokens"] = np.array([int(torch.argmax(l.float())) for l in nxt])
            for n in names:
                if rec[n]:
                    out[pfx + n + ".in"] = bf16_bits(torch.stack([x.reshape(-1) for x, _ in rec[n]]))
                    out[pfx + n + ".out"] = bf16_bits(torch.stack([y.reshape(-1) for _, y in rec[n]]))
            print(f"[{name}] {kind}{kept}: image {src} -> {len(objs)} objects, min margin {min(margins):.3f}: {objs[0]}", flush=True)
            kept += 1

    # query with spatial refs: a point and a box
    refs = [(0.25, 0.5), (0.125, 0.25, 0.625, 0.75)]
    kept, src = 0, -1
    while kept < 1:
        src += 1
        assert src < 200
        image = synth.synthetic_image_array(src, seed, (378, 378))
        rec = {"decode_logits": []}
        orig_lm_head = ref_md.lm_head

        def decode_tap(x, mask, pos_ids, lora):
            logits, hidden = orig_decode_tok(x, mask, pos_ids, lora)
            rec["decode_logits"].append(logits[0].clone())
            return logits, hidden

        def lm_head_tap(h, w):
            o = orig_lm_head(h, w)
            rec.setdefault("prompt_logits", o[0].clone())
            return o

        model._decode_one_tok, ref_md.lm_head = decode_tap, lm_head_tap
        try:
            ans = model.query(Image.fromarray(image, "RGB"), "11 12 13", spatial_refs=refs,
                              settings={"temperature": 0, "max_tokens": 10, "variant": None})["answer"]
        finally:
            model._decode_one_tok, ref_md.lm_head = orig_decode_tok, orig_lm_head
        steps = [rec["prompt_logits"]]
        for lg in rec["decode_logits"]:
            lg = lg.clone()
            lg[cfg.tokenizer.answer_id] = float("-inf")
            steps.append(lg)
        margins = [margin(l) for l in steps]
        if min(margins) < min_margin:
            print(f"[{name}] spatial query: skip image {src}: min margin {min(margins):.3f}", flush=True)
            continue
        out["spatial.image_index"] = np.int64(src)
        out["spatial.refs_point"] = np.array(refs[0])
        out["spatial.refs_box"] = np.array(refs[1])
        out["spatial.question"] = np.array([11, 12, 13])
        out["spatial.tokens"] = np.array([int(t) for t in ans.split()])
        out["spatial.margins"] = np.array(margins, dtype=np.float32)
        print(f"[{name}] spatial query: image {src} -> {ans} (min margin {min(margins):.3f})", flush=True)
        kept += 1
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)", flush=True)


def gen_detect13(name="md2b_detect13", cfg_name="2b", seed=1, n_images=8, max_objects=4, size=(768, 1024)):
    """BASELINE.json configs[4]'s workload shape at full size: the reference's own ``detect`` on seeded
    768 x 1024 images (tiling (3, 4) -> 13 crops, image_crops.py:58-167), Moondream-2B, fixed ``max_objects``.
    UNFILTERED: every image is kept with the bf16-ulp margin of every decision (x, y, w, h bins, next token), so
    consumers compare objects up to the first narrow decision; the projected multi-crop embeddings (vision.py:77-89 on
    the stitched 3 x 4 grid, moondream.py:206-228) of the first two images are recorded (sampled) as well.
    reference: moondream.py:735-781 -> 653-733, region.py:12-93."""
    from PIL import Image

    cfg = get_config(cfg_name)
    sd = synth.synthetic_state_dict(cfg, seed=seed)
    model, ref_md = load_reference(cfg, sd)
    out = {"seed": np.int64(seed), "cfg": np.array(cfg_name), "max_objects": np.int64(max_objects), "object_ids": np.array([7, 8]),
           "size": np.array(size), "n_images": np.int64(n_images), "proj_row_stride": np.int64(9), "proj_col_stride": np.int64(8)}
    names = ("decode_coordinate", "decode_size")
    orig = {n: getattr(ref_md, n) for n in names}
    orig_decode_tok, orig_vis_proj = model._decode_one_tok, model._vis_proj

    def ulps(lg):
        top = torch.topk(lg.float().reshape(-1), 2).values
        return float(top[0] - top[1]) / 2.0 ** (np.floor(np.log2(max(abs(float(top[0])), 2.0 ** -120))) - 7)

    def eos_ulps(lg):
        """margin of the loop's EOS test (moondream.py:668-671): best non-EOS logit vs the EOS logit, in bf16 ulps of the larger"""
        lg = lg.float().reshape(-1)
        eos = float(lg[cfg.tokenizer.eos_id])
        rest = lg.clone()
        rest[cfg.tokenizer.eos_id] = float("-inf")
        best = float(rest.max())
        return abs(best - eos) / 2.0 ** (np.floor(np.log2(max(abs(best), abs(eos), 2.0 ** -120))) - 7)

    for i in range(n_images):
        image = synth.synthetic_image_array(i, seed, size)
        rec = {"decode_coordinate": [], "decode_size": [], "next": [], "proj": []}

        def tap(n):
            def f(x, w):
                y = orig[n](x, w)
                rec[n].append(y.detach().clone())
                return y
            return f

        def decode_tok_tap(x, mask, pos_ids, lora):
            logits, hidden = orig_decode_tok(x, mask, pos_ids, lora)
            rec["next"].append(logits[0].detach().clone())
            return logits, hidden

        def proj_tap(g, r):
            y = orig_vis_proj(g, r)
            rec["proj"].append((tuple(r.shape[:2]), y.detach().clone()))
            return y

        for n in names:
            setattr(ref_md, n, tap(n))
        model._decode_one_tok, model._vis_proj = decode_tok_tap, proj_tap
        t0 = time.perf_counter()
        try:
            res = model.detect(Image.fromarray(image, "RGB"), "7 8", settings={"max_objects": max_objects, "variant": None})
        finally:
            for n in names:
                setattr(ref_md, n, orig[n])
            model._decode_one_tok, model._vis_proj = orig_decode_tok, orig_vis_proj
        dt = time.perf_counter() - t0
        objs = res["objects"]
        nxt = rec["next"][2::3]  # three decoder steps per object; the last one decides the next token
        margins = []
        for k in range(len(objs)):
            sz = rec["decode_size"][k]
            margins.append([ulps(rec["decode_coordinate"][2 * k]), ulps(rec["decode_coordinate"][2 * k + 1]), ulps(sz[0]), ulps(sz[1]), eos_ulps(nxt[k])])
        pfx = f"img{i}."
        out[pfx + "objects"] = np.array([[o[k] for k in ("x_min", "y_min", "x_max", "y_max")] for o in objs], dtype=np.float64).reshape(len(objs), 4)
        out[pfx + "margins"] = np.array(margins, dtype=np.float32).reshape(len(objs), 5)
        out[pfx + "next_tokens"] = np.array([int(torch.argmax(l.float())) for l in nxt], dtype=np.int64)
        out[pfx + "seconds"] = np.float64(dt)
        (gh, gw), proj = rec["proj"][0]
        out[pfx + "grid"] = np.array([gh, gw])
        if i < 2:
            out[pfx + "vis.proj"] = bf16_bits(proj[::9, ::8])
        print(f"[{name}] image {i}: grid {gh}x{gw}, {len(objs)} objects in {dt:.1f}s, margins(ulps) "
              f"{[round(min(m), 2) for m in margins]}", flush=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)", flush=True)


def gen_layers(name="tiny_layers", cfg_name="tiny", seed=1):
    """PER-LAYER drift profile (round 6): the reference's activation after EVERY ViT block (as post_ln(x_k): what
    vision_encoder, vision.py:64-74, returns with its block list cut after block k) and after EVERY decoder block of the image
    prefill (text_decoder, text.py:128-160, cut after block k), for image 0 of tiny_seed1 -- plus, per layer, the relative RMS
    distance of the ORACLE from the reference there.  tests/test_model_gpu.py holds the HIP path to 1.3 x that distance layer by
    layer: a single bad layer is caught where it happens instead of being forgiven by an end-of-stack tolerance."""
    from types import SimpleNamespace
    from PIL import Image
    from moondream.torch.vision import prepare_crops, create_patches
    from moondream.torch.layers import attn, layer_norm, mlp
    from moondream.torch.text import text_decoder, text_encoder
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import moondream_oracle as O

    cfg = get_config(cfg_name)
    sd = synth.synthetic_state_dict(cfg, seed=seed)
    model, ref_md = load_reference(cfg, sd)
    g0 = np.load(os.path.join(GOLD, "tiny_seed1.npz"))
    index = int(g0["image_index"][0])
    image = synth.synthetic_image_array(index, seed, (378, 378))
    vc, w = model.config.vision, model.vision
    ts = 9
    out = {"seed": np.int64(seed), "cfg": np.array(cfg_name), "image_index": np.int64(index), "vit_token_stride": np.int64(ts)}

    def rel(a, b):
        a, b = a.float(), b.float()
        return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())

    orc = O.Oracle(cfg, sd)
    tap = {"__all_blocks__": True}
    crops_u8 = np.stack([image, image])
    orc.encode_image(crops_u8, (1, 1), tap)
    drift_v, drift_t = [], []
    with torch.inference_mode():
        crops, tiling = prepare_crops(Image.fromarray(image, "RGB"), vc, device="cpu")
        assert tuple(tiling) == (1, 1)
        x = w.patch_emb(create_patches(crops, vc.enc_patch_size)) + w.pos_emb
        for i, block in enumerate(w.blocks):
            x = x + attn(layer_norm(x, block.ln1), block.attn, n_heads=vc.enc_n_heads)
            x = x + mlp(layer_norm(x, block.ln2), block.mlp)
            y = layer_norm(x, w.post_ln)
            yo = O.layer_norm(tap[f"vit.block{i}"], sd["vision.post_ln.weight"], sd["vision.post_ln.bias"])
            out[f"vit.ln_block{i}"] = bf16_bits(y[:, ::ts])
            drift_v.append(rel(yo, y))
        img_emb = model._run_vision_encoder(Image.fromarray(image, "RGB"))
        bos = text_encoder(torch.tensor([[cfg.tokenizer.bos_id]]), model.text)
        x0 = torch.cat([bos, img_emb[None]], dim=1)
        mask = model.attn_mask[:, :, 0 : x0.size(1), :]
        pos_ids = torch.arange(x0.size(1), dtype=torch.long)
        tap_t = {"__all_blocks__": True}   # the oracle's decoder from the REFERENCE's embeddings: the decoder's drift alone
        O.text_decoder(x0[0], sd, cfg, O.OracleKV.empty(cfg), pos_ids, orc.cos, orc.sin, tap_t, orc.fast)
        for k in range(1, cfg.text.n_layers + 1):
            cut = SimpleNamespace(blocks=model.text.blocks[:k], freqs_cis=model.text.freqs_cis)
            h = text_decoder(x0, cut, mask, pos_ids, model.config.text, None)[0]
            out[f"text.block{k - 1}"] = bf16_bits(h)
            drift_t.append(rel(tap_t[f"text.block{k - 1}"], h))
        out["text.input"] = bf16_bits(x0[0])   # the decoder is driven from the REFERENCE's embeddings: its drift alone
    out["oracle_drift_vit"] = np.array(drift_v, dtype=np.float64)
    out["oracle_drift_text"] = np.array(drift_t, dtype=np.float64)
    print(f"[{name}] oracle-vs-reference drift: ViT {['%.2e' % d for d in drift_v]}; decoder {['%.2e' % d for d in drift_t]}", flush=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)", flush=True)


def gen_sampling(name="sampling_top_p", seed=7):
    """The reference's sampling filter on fixed logits: softmax(logits / T) -> MoondreamModel._apply_top_p
    (moondream.py:270-278, called at :316-317 and :526-527), bf16 like the decode path's logits.
    Rows: peaked, flat-ish and tied distributions; vocab 8192 and the real 51200."""
    cfg = get_config("tiny")
    sd = synth.synthetic_state_dict(cfg, seed=1)
    model, _ = load_reference(cfg, sd)
    g = torch.Generator().manual_seed(seed)
    out = {}
    cases = [(0.5, 0.3, 8192, 3.0), (1.0, 0.9, 8192, 2.0), (0.7, 0.5, 8192, 0.25), (0.5, 0.3, 51200, 2.5), (1.5, 0.95, 8192, 4.0)]
    for i, (temp, top_p, vocab, scale) in enumerate(cases):
        logits = (torch.randn(2, vocab, generator=g) * scale).to(torch.bfloat16)
        logits[1, :64] = logits[1, 0]  # a block of exact ties at a random level
        logits[1, 100] = logits[1].float().max() + 1.0
        probs = torch.softmax(logits / temp, dim=-1)
        kept = model._apply_top_p(probs.clone(), top_p)
        out[f"case{i}.temperature"] = np.float32(temp)
        out[f"case{i}.top_p"] = np.float32(top_p)
        out[f"case{i}.logits"] = bf16_bits(logits)
        out[f"case{i}.probs"] = bf16_bits(probs)
        out[f"case{i}.kept"] = bf16_bits(kept)
        print(f"[{name}] case{i}: T={temp} top_p={top_p} V={vocab}: kept {(kept > 0).sum(dim=-1).tolist()} tokens", flush=True)
    out["n_cases"] = np.int64(len(cases))
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)", flush=True)


def gen_reftime(cfg_name="2b", seed=1, n_images=5, max_tokens=32):
    """The REFERENCE's own CPU wall clock on the bench workload, measured in the build container with
    nothing else running (method of sample.py:159-207: warm-up, then timed runs; time.perf_counter
    around encode_image and around the answer generator).  Committed under profiles/ as the
    cross-check of bench.py's cpu_baseline (which has to use the oracle: /root/reference does not
    exist on the GPU box)."""
    import json

    cfg = get_config(cfg_name)
    sd = synth.synthetic_state_dict(cfg, seed=seed)
    model, ref_md = load_reference(cfg, sd)
    caption_ids = cfg.tokenizer.templates["caption"]["normal"]
    t_enc, t_gen = [], []
    for i in range(n_images):
        r = run_reference_caption(model, ref_md, synth.synthetic_image_array(i, seed, (378, 378)), caption_ids, max_tokens)
        t_enc.append(r["t_enc"])
        t_gen.append(r["t_gen"])
        print(f"[reftime] image {i}: encode {r['t_enc']:.2f}s generate {r['t_gen']:.2f}s ({len(r['tokens'])} tokens)", flush=True)
    timing = {
        "what": "unmodified /root/reference moondream/torch (tokenizer stub + seeded synthetic weights), Moondream-2B bf16, "
                "B=1 sequential, 378x378 synthetic images, caption prompt, greedy, 32 tokens; wall clock (time.perf_counter) "
                "around encode_image and around the _generate_answer generator",
        "host": {"cores": os.cpu_count(), "torch_threads": torch.get_num_threads(), "torch": torch.__version__},
        "images_timed": n_images - 1,
        "encode_s_p50": float(np.median(t_enc[1:])), "generate_s_p50": float(np.median(t_gen[1:])),
        "seconds_per_token": float(np.median(t_gen[1:]) / max_tokens),
        "images_per_sec": float(1.0 / (np.median(t_enc[1:]) + np.median(t_gen[1:]))),
        "encode_s": [round(x, 3) for x in t_enc], "generate_s": [round(x, 3) for x in t_gen],
        "note": "image 0 is the warm-up and is excluded from the medians",
    }
    # Round 5: the PORT (bench.py's cpu_baseline: the oracle in fast mode = the reference's own ATen calls in the reference's
    # order) timed side by side, same process, same container, same weights -- the ratio bounds the drift between what the GPU
    # box can time (the port: /root/reference does not exist there) and what SURVEY 8(d) asks for (the unmodified reference).
    del model
    import bench

    est, cores, note, details = bench.cpu_baseline(cfg, sd, seed, max_tokens, budget_s=60.0)
    timing["port"] = {"images_per_sec": est, "threads": cores, "sample": note, "details": details}
    timing["port_vs_reference"] = est / timing["images_per_sec"]
    timing["port_vs_reference_note"] = ("port images/s / reference images/s, both measured in this container back to back; > 1: the port is "
                                        "faster (it skips Python-side glue: mask slicing, per-token .item(), the streaming detokeniser; it "
                                        "reads K/V over all 2048 slots like the reference)")
    prof = os.path.join(REPO, "profiles", "r05_reference_vs_port_cpu_timing_build_container.json")
    with open(prof, "w") as f:
        json.dump(timing, f, indent=1)
    print(f"[reftime] reference {timing['images_per_sec']:.3f} images/s, port {est:.3f} images/s on {os.cpu_count()} cores "
          f"(port / reference = {timing['port_vs_reference']:.3f}) -> {prof}", flush=True)


def gen_reasoning(name="tiny_reasoning", cfg_name="tiny", seed=1, n_cases=2, max_tokens=8, min_margin=0.75):
    """query(image, question, reasoning=True) through the reference's public API (moondream.py:541-618 ->
    _generate_reasoning moondream.py:323-432 -> _generate_answer): reasoning text ids, grounding, answer ids and
    the top-1/top-2 margin of every decision.  Cases are kept when every decision has margin >= min_margin."""
    from PIL import Image

    cfg = get_config(cfg_name)
    sd = synth.synthetic_state_dict(cfg, seed=seed)
    model, ref_md = load_reference(cfg, sd)
    tk = cfg.tokenizer
    out = {"seed": np.int64(seed), "cfg": np.array(cfg_name), "max_tokens": np.int64(max_tokens)}
    orig_decode, orig_lm_head = model._decode_one_tok, ref_md.lm_head
    kept, src = 0, -1
    while kept < n_cases:
        src += 1
        assert src < 200, "could not find enough wide-margin reasoning cases"
        image = synth.synthetic_image_array(src, seed, (378, 378))
        rec = {"logits": []}

        def decode_tap(x, mask, pos_ids, lora):
            logits, hidden = orig_decode(x, mask, pos_ids, lora)
            rec["logits"].append(("decode", logits[0].clone()))
            return logits, hidden

        def lm_head_tap(h, w):
            o = orig_lm_head(h, w)
            if not rec["logits"] or rec["logits"][-1][0] != "pending":
                rec["logits"].append(("prefill", o[0].clone()))
            return o

        model._decode_one_tok, ref_md.lm_head = decode_tap, lm_head_tap
        try:
            res = model.query(Image.fromarray(image, "RGB"), "11 12 13", reasoning=True,
                              settings={"temperature": 0, "max_tokens": max_tokens, "variant": None})
        finally:
            model._decode_one_tok, ref_md.lm_head = orig_decode, orig_lm_head
        # lm_head is also called inside _decode_one_tok (its tap fires first): drop those twins, keep the prompt prefills
        seq = []
        for kind, lg in rec["logits"]:
            if kind == "decode" and seq and seq[-1][0] == "prefill" and torch.equal(seq[-1][1], lg):
                seq[-1] = (kind, lg)
            else:
                seq.append((kind, lg))
        r_ids = [int(t) for t in res["reasoning"]["text"].split()]
        a_ids = [int(t) for t in res["answer"].split()]
        # decisions in order: reasoning prompt prefill, one decode per reasoning token (eos / size suppressed,
        # moondream.py:397-398), answer prompt prefill, one decode per answer token (answer_id suppressed, :517)
        assert len(seq) == 2 + len(r_ids) + len(a_ids), (len(seq), len(r_ids), len(a_ids))
        margins = []
        for i, (kind, lg) in enumerate(seq):
            lg = lg.clone().float()
            if 1 <= i <= len(r_ids):
                lg[tk.eos_id] = lg[tk.size_id] = float("-inf")
            elif i > len(r_ids) + 1:
                lg[tk.answer_id] = float("-inf")
            top = torch.topk(lg, 2).values
            margins.append(float(top[0] - top[1]))
        if min(margins) < min_margin:
            print(f"[{name}] skip image {src}: min margin {min(margins):.3f}", flush=True)
            continue
        pfx = f"case{kept}."
        out[pfx + "image_index"] = np.int64(src)
        out[pfx + "reasoning_tokens"] = np.array(r_ids)
        out[pfx + "answer_tokens"] = np.array(a_ids)
        out[pfx + "n_grounding"] = np.int64(len(res["reasoning"]["grounding"]))
        out[pfx + "margins"] = np.array(margins, dtype=np.float32)
        print(f"[{name}] case{kept}: image {src}: reasoning {r_ids} answer {a_ids} grounding {res['reasoning']['grounding']} "
              f"(min margin {min(margins):.3f})", flush=True)
        kept += 1
    out["n_cases"] = np.int64(kept)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path}", flush=True)


def gen_lora(name="tiny_lora", cfg_name="tiny", seed=1, rank=8, n_cases=2, max_tokens=12, min_margin=0.75):
    """LoRA variant side path: the reference's caption() with settings["variant"] set, the download replaced by a
    seeded synthetic variant (lora.py:54-79 -> text.py:31-32,55-56, layers.py:129-146; the image prefill uses the
    variant too, moondream.py:241-257).  Records ids / margins with and without the variant."""
    cfg = get_config(cfg_name)
    sd = synth.synthetic_state_dict(cfg, seed=seed)
    model, ref_md = load_reference(cfg, sd)
    lora = synth.synthetic_lora(cfg, seed=seed, rank=rank)
    ref_md.variant_state_dict = lambda variant_id, device=None: (lora if variant_id == "synthetic" else None)
    caption_ids = cfg.tokenizer.templates["caption"]["normal"]
    out = {"seed": np.int64(seed), "cfg": np.array(cfg_name), "rank": np.int64(rank), "max_tokens": np.int64(max_tokens)}

    def run(image, variant):
        from PIL import Image

        rec = {"decode_logits": []}
        orig_decode, orig_lm_head = model._decode_one_tok, ref_md.lm_head

        def decode_tap(x, mask, pos_ids, lr):
            assert (lr is not None) == (variant is not None)
            logits, hidden = orig_decode(x, mask, pos_ids, lr)
            rec["decode_logits"].append(logits[0].clone())
            return logits, hidden

        def lm_head_tap(h, w):
            o = orig_lm_head(h, w)
            rec.setdefault("prompt_logits", o[0].clone())
            return o

        model._decode_one_tok, ref_md.lm_head = decode_tap, lm_head_tap
        try:
            text = model.caption(Image.fromarray(image, "RGB"), settings={"temperature": 0, "max_tokens": max_tokens, "variant": variant})["caption"]
        finally:
            model._decode_one_tok, ref_md.lm_head = orig_decode, orig_lm_head
        steps = [rec["prompt_logits"]]
        for lg in rec["decode_logits"]:
            lg = lg.clone()
            lg[cfg.tokenizer.answer_id] = float("-inf")
            steps.append(lg)
        margins = [float((lambda t: t[0] - t[1])(torch.topk(lg.float(), 2).values)) for lg in steps]
        return [int(t) for t in text.split()], margins

    kept, src = 0, -1
    while kept < n_cases:
        src += 1
        assert src < 200
        image = synth.synthetic_image_array(src, seed, (378, 378))
        toks, margins = run(image, "synthetic")
        base, _ = run(image, None)
        if min(margins) < min_margin or toks == base:
            print(f"[{name}] skip image {src}: min margin {min(margins):.3f}, differs from base: {toks != base}", flush=True)
            continue
        out[f"case{kept}.image_index"] = np.int64(src)
        out[f"case{kept}.tokens"] = np.array(toks)
        out[f"case{kept}.base_tokens"] = np.array(base)
        out[f"case{kept}.margins"] = np.array(margins, dtype=np.float32)
        print(f"[{name}] case{kept}: image {src}: variant {toks} | base {base} (min margin {min(margins):.3f})", flush=True)
        kept += 1
    out["n_cases"] = np.int64(kept)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path}", flush=True)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["tiny", "multicrop", "crops"]
    if "crops" in which:
        gen_crops()
    if "tiny" in which:
        gen_model_case("tiny_seed1", "tiny", 1, [(378, 378)], 24, True, n_images=3)
    if "layers" in which:
        gen_layers()
    if "multicrop" in which:
        gen_multicrop()
    if "textonly" in which:
        gen_textonly()
    if "0.5b" in which:
        gen_model_case("md05b_seed1", "0.5b", 1, [(378, 378)], 32, False, n_images=2, min_margin=0.5)
    if "lora" in which:
        gen_lora()
    if "reasoning" in which:
        gen_reasoning()
    if "sampling" in which:
        gen_sampling()
    if "detect" in which:
        gen_detect()
    if "reftime" in which:
        gen_reftime()
    if "bench64" in which:
        gen_bench64()
    if "vqa64" in which:       # the vqa32 bench leg's fixture (64 images x 32-id question prompts)
        gen_bench64("md2b_vqa64", prompt_kind="vqa32")
    if "detect13" in which:
        gen_detect13()
    if "lora2b" in which:      # the side paths at the 2B shapes (tests/test_model_gpu.py::test_2b_lora_and_reasoning_vs_reference)
        gen_lora("md2b_lora", "2b", n_cases=2, max_tokens=8, min_margin=0.7)
    if "reasoning2b" in which:
        gen_reasoning("md2b_reasoning", "2b", n_cases=2, max_tokens=6, min_margin=0.7)
    if "2b" in which:
        gen_model_case("md2b_seed1", "2b", 1, [(378, 378)], 32, False, n_images=3, min_margin=0.5)


if __name__ == "__main__":
    main()
