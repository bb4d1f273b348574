"""Seeded synthetic checkpoints and inputs.

There are no real weights offline (shim 2), so parity and
the benchmark run on a *deterministic synthetic checkpoint*.  Values come from
a counter-based integer hash of (tensor name, element index, seed), evaluated
with integer tensor ops only, so the very same bits are produced on the CPU
(oracle / reference runs) and on the GPU (fast 2B generation) by any torch
build: no dependence on a library RNG stream.

Tensor names and shapes are the reference module tree's ``state_dict`` names
(reference: vision.py:92-147, text.py:175-221, moondream.py:94-136), i.e. the
same dict loads into the reference ``MoondreamModel`` with ``load_state_dict``
and into this package's weight packer.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict

import numpy as np
import torch

from .config import MoondreamConfig

_M32 = 0xFFFFFFFF
TEXT_PROJ_GAIN = 1.0
TEXT_FC2_GAIN = 0.5
# beta is given for dim 256 and scaled by 16/sqrt(dim) (the planted logit grows like
# beta*sqrt(dim)); see _plant_lm_head
PLANT = dict(beta=1.6, c=8.0, s0=0.1)
WTE_STD = 2.5


def hash_uniform(n: int, key: int, device="cpu") -> torch.Tensor:
    """n floats uniform in [-1, 1), a pure function of (index, key).

    lowbias32-style avalanche on a 32-bit counter; the top 24 bits become the
    mantissa so the float conversion is exact (bit-reproducible everywhere).
    """
    out = torch.empty(n, dtype=torch.float32, device=device)
    step = 1 << 24  # bounds the int64 temporaries to ~128 MiB each
    for s in range(0, n, step):
        e = min(n, s + step)
        x = torch.arange(s, e, dtype=torch.int64, device=device)
        x = (x * 0x9E3779B1 + (key & _M32)) & _M32
        x = x ^ (x >> 16)
        x = (x * 0x7FEB352D) & _M32
        x = x ^ (x >> 15)
        x = (x * 0x846CA68B) & _M32
        x = x ^ (x >> 16)
        out[s:e] = (x >> 8).to(torch.float32) * (2.0 / (1 << 24)) - 1.0
    return out


def _key(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) * 2654435761 + seed * 40503) & _M32


def _tensor(name, shape, std, seed, device, dtype, mean=0.0):
    n = int(np.prod(shape))
    # uniform on [-a, a) has std a/sqrt(3)
    t = hash_uniform(n, _key(name, seed), device) * (std * math.sqrt(3.0))
    if mean != 0.0:
        t = t + mean
    return t.reshape(shape).to(dtype)


def synthetic_state_dict(
    config: MoondreamConfig,
    seed: int = 0,
    device="cpu",
    dtype=torch.bfloat16,
    include_region: bool = True,
    planted: bool = True,
) -> Dict[str, torch.Tensor]:
    """A full checkpoint keyed like ``MoondreamModel.state_dict()``.

    Scales are chosen so that activations stay O(1) through all layers, softmax
    rows are far from uniform (q.k scores have std ~1.5) and LayerNorm
    weights/biases are non-trivial, so that a wrong kernel cannot hide behind a
    degenerate operating point.
    """
    v, t, r = config.vision, config.text, config.region
    sd: Dict[str, torch.Tensor] = {}

    def lin(prefix, out_f, in_f, gain=1.0, bias_std=0.05):
        sd[prefix + ".weight"] = _tensor(
            prefix + ".weight", (out_f, in_f), gain / math.sqrt(in_f), seed, device, dtype
        )
        sd[prefix + ".bias"] = _tensor(prefix + ".bias", (out_f,), bias_std, seed, device, dtype)

    def ln(prefix, d):
        sd[prefix + ".weight"] = _tensor(prefix + ".weight", (d,), 0.1, seed, device, dtype, mean=1.0)
        sd[prefix + ".bias"] = _tensor(prefix + ".bias", (d,), 0.1, seed, device, dtype)

    # ---- vision (reference: vision.py:92-147)
    lin("vision.patch_emb", v.enc_dim, v.patch_dim, gain=1.0)
    for i in range(v.enc_n_layers):
        p = f"vision.blocks.{i}"
        ln(p + ".ln1", v.enc_dim)
        lin(p + ".attn.qkv", 3 * v.enc_dim, v.enc_dim, gain=1.25)
        lin(p + ".attn.proj", v.enc_dim, v.enc_dim, gain=1.0)
        ln(p + ".ln2", v.enc_dim)
        lin(p + ".mlp.fc1", v.enc_ff_dim, v.enc_dim, gain=1.0)
        lin(p + ".mlp.fc2", v.enc_dim, v.enc_ff_dim, gain=1.0)
    ln("vision.post_ln", v.enc_dim)
    lin("vision.proj_mlp.fc1", v.proj_inner_dim, 2 * v.enc_dim)
    lin("vision.proj_mlp.fc2", v.proj_out_dim, v.proj_inner_dim)
    sd["vision.pos_emb"] = _tensor("vision.pos_emb", (1, v.n_patches, v.enc_dim), 0.5, seed, device, dtype)

    # ---- text (reference: text.py:175-221)
    for i in range(t.n_layers):
        p = f"text.blocks.{i}"
        ln(p + ".ln", t.dim)
        lin(p + ".attn.qkv", t.qkv_dim, t.dim, gain=1.0)
        # sharper text attention (q,k rows x1.6 -> score std ~2.5) with a strong
        # output projection, so the residual stream really depends on WHICH
        # keys (image tokens) were attended to
        qk_rows = (t.n_heads + t.n_kv_heads) * t.head_dim
        w = sd[p + ".attn.qkv.weight"].float()
        w[:qk_rows] *= 1.6
        sd[p + ".attn.qkv.weight"] = w.to(dtype)
        lin(p + ".attn.proj", t.dim, t.dim, gain=TEXT_PROJ_GAIN)
        lin(p + ".mlp.fc1", t.ff_dim, t.dim, gain=1.0)
        lin(p + ".mlp.fc2", t.dim, t.ff_dim, gain=TEXT_FC2_GAIN)
    ln("text.post_ln", t.dim)
    sd["text.wte"] = _tensor("text.wte", (t.vocab_size, t.dim), WTE_STD, seed, device, dtype)
    if planted:
        _plant_lm_head(sd, config, seed, device, dtype, **PLANT)
    else:
        lin("text.lm_head", t.vocab_size, t.dim, gain=1.0)

    # ---- region (reference: moondream.py:94-136)
    if include_region:
        lin("region.coord_encoder", r.dim, r.coord_feat_dim)
        lin("region.coord_decoder.fc1", r.inner_dim, r.dim)
        lin("region.coord_decoder.fc2", r.coord_out_dim, r.inner_dim)
        lin("region.size_encoder", r.dim, r.size_feat_dim)
        lin("region.size_decoder.fc1", r.inner_dim, r.dim)
        lin("region.size_decoder.fc2", r.size_out_dim, r.inner_dim)
        sd["region.coord_features"] = _tensor(
            "region.coord_features", (1, r.coord_feat_dim // 2), 2.0, seed, device, dtype
        )
        sd["region.size_features"] = _tensor(
            "region.size_features", (2, r.size_feat_dim // 2), 2.0, seed, device, dtype
        )
    return sd


def _plant_lm_head(sd, config, seed, device, dtype, beta=1.0, c=4.0, s0=0.1):
    """lm_head with a planted "bigram + context bit" structure.

    With i.i.d. random weights the top-1/top-2 logit gap is 0-3 bf16 ulps
    ("token-ID bit-exactness"), so greedy ids are decided
    by ties and cannot be compared between two correct implementations.  Here
    every token t owns a PAIR of candidate next tokens {2m, 2m+1},
    m = perm(t) >> 1: both rows carry beta * unit(wte[t]) (so the pair wins
    against the other V-2 rows by a wide margin once LN(h) has a component
    along wte[t], which the residual stream guarantees), and the two rows
    differ by +-c * r for one fixed random direction r, so WHICH of the two
    wins is the sign of <LN(h), r> -- a feature of the whole context (image
    prefix and all previous tokens through attention).  The result is a
    diverse token stream whose every id depends on the full computation while
    the typical margin is tens of bf16 ulps.  Row 0 (eos) gets a large
    negative bias so generation length is fixed by max_tokens.
    """
    t = config.text
    V, D = t.vocab_size, t.dim
    beta = beta * 16.0 / math.sqrt(D)
    a = int(V * 0.6180339887) | 1  # odd multiplier near V/phi, made coprime with V
    while math.gcd(a, V) != 1:
        a += 2
    a_inv = pow(a, -1, V)
    b = 17
    rows = torch.arange(V, device=device, dtype=torch.int64)
    # perm(t) = (a*t + b) mod V  ->  perm^-1(v) = a_inv * (v - b) mod V
    t_even = (a_inv * (((rows & ~1) - b) % V)) % V
    t_odd = (a_inv * (((rows | 1) - b) % V)) % V
    # Reductions (norms, dot products) run in float64 and are rounded to fp32 once, so
    # that CPU and GPU generation give the same bits despite different summation
    # orders; everything else is elementwise IEEE arithmetic.
    wte = sd["text.wte"].float()
    unit = wte / wte.double().norm(dim=-1, keepdim=True).float()
    r = hash_uniform(D, _key("text.lm_head.context_dir", seed), device) * math.sqrt(3.0)
    r = r / r.double().norm().float()  # <LN(h), r> ~ N(0,1) for |LN(h)| ~ sqrt(D)
    sign = torch.where((rows & 1) == 0, 1.0, -1.0).to(torch.float32).unsqueeze(1)
    noise = hash_uniform(V * D, _key("text.lm_head.weight", seed), device).reshape(V, D)
    ue, uo = unit[t_even], unit[t_odd]
    # context direction with the pair's own token directions projected out, so the
    # winning bit is not a function of the current token's embedding alone
    dot_e = (ue.double() @ r.double()).float().unsqueeze(1)
    dot_o = (uo.double() @ r.double()).float().unsqueeze(1)
    r_pair = r.unsqueeze(0) - dot_e * ue - dot_o * uo
    w = beta * (ue + uo) + c * sign * r_pair + noise * (s0 * math.sqrt(3.0 / D))
    sd["text.lm_head.weight"] = w.to(dtype)
    bias = _tensor("text.lm_head.bias", (V,), 0.05, seed, device, torch.float32)
    bias[config.tokenizer.eos_id] = -60.0
    sd["text.lm_head.bias"] = bias.to(dtype)


# --------------------------------------------------------------------------
# Synthetic inputs (BASELINE.md section 3 / SURVEY.md section 8d)
# --------------------------------------------------------------------------
def synthetic_image_array(index: int, seed: int = 0, size=(378, 378)) -> np.ndarray:
    """HWC uint8 image ``np.random.default_rng(seed+index)`` (PCG64: stable)."""
    rng = np.random.default_rng(seed + index)
    return rng.integers(0, 256, (size[0], size[1], 3), dtype=np.uint8)


def synthetic_image(index: int, seed: int = 0, size=(378, 378)):
    from PIL import Image

    return Image.fromarray(synthetic_image_array(index, seed, size), "RGB")


def synthetic_vqa_prompt(config: MoondreamConfig, index: int, seed: int = 0, n_question: int = 27):
    """query prefix + n seeded question ids + suffix + suffix (32 ids with the
    default templates; mirrors reference moondream.py:564,586-591,604)."""
    tpl = config.tokenizer.templates["query"]
    rng = np.random.default_rng(1000003 * (seed + 1) + index)
    hi = min(50000, config.text.vocab_size)
    q = rng.integers(10, hi, n_question).tolist()
    return list(tpl["prefix"]) + q + list(tpl["suffix"]) + list(tpl["suffix"])


def synthetic_lora(config: MoondreamConfig, seed: int = 0, rank: int = 8, device="cpu", dtype=torch.bfloat16) -> dict:
    """A seeded LoRA "variant" in the nested layout the reference's ``variant_state_dict`` returns
    (lora.py:54-79): per decoder block A [rank, in] / B [out, rank] for attn.qkv, attn.proj, mlp.fc1, mlp.fc2.
    Scales are large enough to move the logits (a wrong or missing side path changes the generated ids)."""
    t = config.text
    blocks = {}
    for i in range(t.n_layers):
        def pair(name, out_f, in_f, gain):
            a = _tensor(f"lora.{i}.{name}.A", (rank, in_f), 1.0 / math.sqrt(in_f), seed, device, dtype)
            b = _tensor(f"lora.{i}.{name}.B", (out_f, rank), gain / math.sqrt(rank), seed, device, dtype)
            return {"A": a, "B": b}

        blocks[str(i)] = {
            "attn": {"qkv": pair("qkv", t.qkv_dim, t.dim, 0.5), "proj": pair("proj", t.dim, t.dim, 0.5)},
            "mlp": {"fc1": pair("fc1", t.ff_dim, t.dim, 0.5), "fc2": pair("fc2", t.dim, t.ff_dim, 0.25)},
        }
    return {"text": {"blocks": blocks}}
