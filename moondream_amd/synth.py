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
PLANT = dict(beta=1.6, c_code=12.0, c_deep=0.8, c_tok=1.0, s0=0.1)
WTE_STD = 2.5

# ---- round 6: the planted "image code" path (see _plant_code_path) -------------------------------------------------
# Every synthetic image carries a colour cast: per channel one of four levels of the normalised pixel mean
# (synthetic_image_array).  2 bits x 3 channels = 64 image classes = a 6-bit code that a few planted weights carry -- through
# the kernels under test, not around them -- to the last hidden state, where the lm_head pair rows and the region decoders read it.
CODE_BITS = 6
SPECIAL_PAIRS = 5                         # candidate pairs {0,1} .. {8,9}: special token ids, never a planted winner
TOK_BITS = 32                             # protected decoder coordinates that carry the codeword of the current token's pair
CAST_LEVELS = (-0.6, -0.2, 0.2, 0.6)      # normalised ((x / 255 - 0.5) / 0.5) channel means of the four levels
CAST_NOISE = 0.35                         # amplitude of the uniform pixel noise around the level (no clipping: 0.6 + 0.35 < 1)
CODE_GAMMA = 8.0                          # ViT protected coordinates: GAMMA x channel mean, and the constant reference GAMMA
CODE_STEEP = 6.0                          # projector units: v = K x GAMMA x level / sigma = +-1.3 (inner levels) / +-4.0 (outer levels)
CODE_SIGMA = 7.3                          # nominal standard deviation of the ViT's last residual stream (tiny 7.2, 2B 7.3): only the
                                          # read-out constants of the "lo" bit depend on it (its midpoint tracks sigma through LN[ref])
CODE_RHO = 1.0                            # decoder: V copy gain of the code coordinates
DEFAULT_CODE = (1, -1, -1, 1, -1, 1)      # the code of "no image": carried by every token embedding
IMAGE_CODE_AMPLITUDE = 0.5                # an image embedding carries its code as +-0.5 ...
TEXT_CODE_AMPLITUDE = 0.25                 # ... a token embedding the default code as +-0.25


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
        _plant_code_path(sd, config, dtype)
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
        if planted:
            _plant_region_heads(sd, config, dtype)
    return sd


def protected_coords(config: MoondreamConfig) -> dict:
    """Feature indices the planted code path owns.  ViT stream (enc_dim): r, g, b channel means, a constant reference, an
    always-zero coordinate.  Decoder stream (dim): ``zero`` (always 0: LayerNorm's mean shift is read off it), ``p1`` (the
    6-bit code as +-1, present at image-token positions only, written by the vision projection and never by a decoder layer),
    ``p2`` (what the attention of every decoder layer copies of p1 from the image keys, accumulated; read by lm_head and the
    region decoders), ``flag`` (1.0 where the input embedding of a position came from encode_coordinate / encode_size)."""
    e, d = config.vision.enc_dim, config.text.dim
    return {
        "vit_rgb": [e - 5, e - 4, e - 3], "vit_ref": e - 2, "vit_zero": e - 1,
        "tok": list(range(d - 16 - TOK_BITS, d - 16)),
        "flag": [d - 16, d - 15], "zero": d - 14, "p1": list(range(d - 12, d - 6)), "p2": list(range(d - 6, d)),
        "all_text": list(range(d - 16 - TOK_BITS, d)), "all_vit": list(range(e - 5, e)),
    }


def pair_codewords(n_pairs: int, device="cpu") -> torch.Tensor:
    """+-1 codeword [n_pairs, 32] of every token pair: the second-order Reed-Muller code RM(2, 5) -- the pair index as the 16
    coefficients of a Boolean polynomial of degree <= 2 in 5 variables, evaluated at the 32 points of {0, 1}^5.  2^16 codewords,
    any two differ in >= 8 of 32 positions (integer arithmetic only: the same bits everywhere)."""
    assert n_pairs <= 1 << 16
    m = torch.arange(n_pairs, dtype=torch.int64, device=device).unsqueeze(1)          # [n, 1]
    x = torch.arange(32, dtype=torch.int64, device=device).unsqueeze(0)               # [1, 32]: the evaluation points
    xs = [(x >> i) & 1 for i in range(5)]
    acc = (m >> 0) & 1                                                                  # constant term
    bit = 1
    for i in range(5):
        acc = acc ^ (((m >> bit) & 1) & xs[i])
        bit += 1
    for i in range(5):
        for j in range(i + 1, 5):
            acc = acc ^ (((m >> bit) & 1) & xs[i] & xs[j])
            bit += 1
    return (1 - 2 * acc).to(torch.float32)


def token_code_amplitude(config: MoondreamConfig) -> float:
    """Amplitude of the pair codeword in a token embedding: roughly 0.15 x the standard deviation of the last hidden state,
    which grows with depth (2.6 after 3 decoder layers, 6.3 after 24 with this checkpoint's gains)."""
    return 0.4 * (1.0 + 0.25 * math.sqrt(config.text.n_layers))


def image_code_bits(index: int) -> list:
    """The 6 code bits (+1 / -1) of synthetic image ``index``: class = index mod 64, two bits per colour channel (level = 2 hi + lo
    hi = the sign of the level (levels 2, 3 -> +1), lo = inner level (levels 1, 2 -> +1): 0 -> (-1, -1), 1 -> (-1, +1), 2 -> (+1, +1),
    3 -> (+1, -1))."""
    levels = image_cast_levels(index)
    bits = []
    for lv in levels:
        bits += [1 if lv >= 2 else -1, 1 if lv in (1, 2) else -1]
    return bits


def image_cast_levels(index: int) -> tuple:
    """Level (0..3) of the R, G, B colour cast of synthetic image ``index``: the base-4 digits of a fixed permutation of index mod 64."""
    cls = (int(index) * 37 + 11) % 64
    return (cls & 3, (cls >> 2) & 3, (cls >> 4) & 3)


def _plant_code_path(sd, config, dtype):
    """Make greedy token ids a WELL-POSED integer output for every synthetic image (round 6).

    Why: a decision taken by the sign of a deep feature <LN(h), r> is ill-posed whenever |feature| is inside the bf16
    implementation noise of h (relative ~1e-2 after 24 layers: two correct implementations are independent rounding-noise
    realisations), i.e. for 2-3 % of all decisions whatever the gain -- amplification scales signal and noise alike.  Rounds 1-5
    therefore compared ids under a measured licence.  Here the member of a token pair is chosen by a DISCRETE property of the
    image that the network carries with a wide gap, while the deep feature keeps contributing to the logits (c_deep) where the
    teacher-forced logit comparison sees it:

      * patch embedding (vision.py:67): three rows average the normalised pixels of one colour channel (GAMMA x mean), one row is
        the constant GAMMA (bias), one is 0.  No ViT block writes to these five coordinates (zero rows in proj / fc2), every block
        reads them like any other feature; post_ln passes them with weight 1 / bias 0.
      * vision projection (vision.py:77-89): per channel four hidden units of v = K (LN[c] - LN[zero]) -- gelu(v + 1) - gelu(v - 1)
        (~0 / ~2: the sign of the level) and gelu(v) + gelu(-v) (~|v|: inner or outer level, against a midpoint that tracks
        LayerNorm's scale through the reference coordinate); LayerNorm's mean shift cancels against the zero coordinate.  Every
        unit output stays below 5 (e4m3-safe).  fc2 turns them into 6 coordinates of +-0.5 (p1) in every image embedding.
      * every decoder block: one attention head (l mod n_kv_heads; its query rows are zero, i.e. it attends uniformly) copies
        RHO (LN(x)[p1] - LN(x)[zero]) into six of its value dims; proj writes them to p2.  What accumulates in p2 at a text
        position is RHO x sum over layers of the key-average of p1 / row sigma: 729 image keys at amplitude 0.5 against a few text
        keys at 0.25 (the default code) -> sign = the image's code bit, through the softmax normalisation, the V rows of the KV
        cache, proj and the residual adds of every layer.  p1 / zero / flag are never written by a block; fc2 does not write p2.
      * text.post_ln passes p2 and zero with weight 1 / bias 0; _plant_lm_head reads p2[j] - zero.
    """
    v, t = config.vision, config.text
    pc = protected_coords(config)
    f32 = torch.float32

    def edit(name, fn):
        w = sd[name].to(f32)
        fn(w)
        sd[name] = w.to(dtype)

    # ---- vision tower
    pd = v.enc_patch_size * v.enc_patch_size  # create_patches: "b c (h p1) (w p2) -> b (h w) (c p1 p2)" -- channel-major
    def patch_w(w):
        w[pc["all_vit"]] = 0.0
        for c, row in enumerate(pc["vit_rgb"]):
            w[row, c * pd : (c + 1) * pd] = CODE_GAMMA / pd
    def patch_b(b):
        b[pc["all_vit"]] = 0.0
        b[pc["vit_ref"]] = CODE_GAMMA
    edit("vision.patch_emb.weight", patch_w)
    edit("vision.patch_emb.bias", patch_b)
    edit("vision.pos_emb", lambda w: w.__setitem__((Ellipsis, pc["all_vit"]), 0.0))
    zero_rows = lambda rows: (lambda w: w.__setitem__(rows, 0.0))   # (w[list].zero_() would zero a COPY)
    for i in range(v.enc_n_layers):
        for name in ("attn.proj", "mlp.fc2"):
            edit(f"vision.blocks.{i}.{name}.weight", zero_rows(pc["all_vit"]))
            edit(f"vision.blocks.{i}.{name}.bias", zero_rows(pc["all_vit"]))
    edit("vision.post_ln.weight", lambda w: w.__setitem__(pc["all_vit"], 1.0))
    edit("vision.post_ln.bias", zero_rows(pc["all_vit"]))
    # projector: the LAST 14 hidden units decode the levels; they read the GLOBAL crop's features (columns [0, enc_dim)).
    # With v = K (LN[c] - LN[zero]) = K x GAMMA x level / sigma (nominally +-1.3 for the inner levels, +-4.0 for the outer ones):
    #   d = gelu(v + 1) - gelu(v - 1)  ~ 0 for a negative level, ~ 2 for a positive one     -> the "hi" bit (sign of the level)
    #   G = gelu(v) + gelu(-v) ~ |v|                                                         -> the "lo" bit (inner vs outer level),
    #       against the midpoint m x r, r = LN[ref] - LN[zero] = GAMMA / sigma carried by a pass-through pair gelu(r) - gelu(-r) = r
    # EVERY unit output stays below 5: the first version (three threshold pairs per channel, outputs up to 27) needed the
    # difference of two LARGE GELU outputs to be exact, which holds in bf16 and not after the fp8 mode's e4m3 quantisation of
    # the projector's hidden activations (the code bits of a quarter of the images flipped there).
    K = CODE_STEEP
    n_units = 3 * 4 + 2
    u0 = v.proj_inner_dim - n_units
    U = lambda c, k: u0 + 4 * c + k          # channel c: gelu(v + 1), gelu(v - 1), gelu(v), gelu(-v)
    R_POS, R_NEG = u0 + 12, u0 + 13          # gelu(r), gelu(-r)
    def fc1_w(w):
        w[u0:] = 0.0
        for c, col in enumerate(pc["vit_rgb"]):
            for k, sg in enumerate((K, K, K, -K)):
                w[U(c, k), col], w[U(c, k), pc["vit_zero"]] = sg, -sg
        w[R_POS, pc["vit_ref"]], w[R_POS, pc["vit_zero"]] = 1.0, -1.0
        w[R_NEG, pc["vit_ref"]], w[R_NEG, pc["vit_zero"]] = -1.0, 1.0
    def fc1_b(b):
        b[u0:] = 0.0
        for c in range(3):
            b[U(c, 0)], b[U(c, 1)] = 1.0, -1.0
    edit("vision.proj_mlp.fc1.weight", fc1_w)
    edit("vision.proj_mlp.fc1.bias", fc1_b)
    # the linear read-out of the units, solved on the four nominal levels in float64 (a few scalars: the same bits everywhere)
    gelu = lambda x: 0.5 * x * (1.0 + math.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))
    r_nom = CODE_GAMMA / CODE_SIGMA
    g_inner, g_outer = (gelu(K * r_nom * lv) + gelu(-K * r_nom * lv) for lv in (CAST_LEVELS[2], CAST_LEVELS[3]))
    lo_mid, lo_half = 0.5 * (g_inner + g_outer) / r_nom, 0.5 * (g_outer - g_inner)
    def fc2_w(w):
        w[:, u0:] = 0.0          # the planted units feed the code coordinates only
        w[pc["all_text"]] = 0.0  # and nothing else writes the protected coordinates of an image embedding
        A = IMAGE_CODE_AMPLITUDE
        for c in range(3):
            hi, lo = pc["p1"][2 * c], pc["p1"][2 * c + 1]
            w[hi, U(c, 0)], w[hi, U(c, 1)] = A, -A                       # hi = A (d - 1)            (bias below)
            w[lo, U(c, 2)], w[lo, U(c, 3)] = -A / lo_half, -A / lo_half  # lo = A (m r - G) / half: +1 inner level, -1 outer
            w[lo, R_POS], w[lo, R_NEG] = A * lo_mid / lo_half, -A * lo_mid / lo_half
    def fc2_b(b):
        b[pc["all_text"]] = 0.0
        for c in range(3):
            b[pc["p1"][2 * c]] = -IMAGE_CODE_AMPLITUDE
    edit("vision.proj_mlp.fc2.weight", fc2_w)
    edit("vision.proj_mlp.fc2.bias", fc2_b)

    # ---- decoder
    def wte(w):
        w[:, pc["all_text"]] = 0.0
        # EVERY token embedding carries a DEFAULT code at half the amplitude of an image embedding's: without an image (the
        # text-only query, moondream.py:564-575) every key holds it, so p2 accumulates it whatever the attention pattern and
        # the pair decisions stay well-posed there too (they follow the default bits); with an image the 729 image keys
        # outweigh the few text keys (a text key counts half, and there are 5-64 of them)
        for j, bit in enumerate(DEFAULT_CODE):
            w[:, pc["p1"][j]] = TEXT_CODE_AMPLITUDE * bit
        # ... and the CODEWORD OF ITS PAIR (the pair of next-token candidates it owns, _plant_lm_head) in the ``tok``
        # coordinates: no decoder layer writes them, so the last hidden state of a position still holds its input token's
        # codeword exactly, and lm_head's pair rows read it -- WHICH pair wins is decided with a gap of >= 8 of 32 positions
        # instead of by the depth-diluted <LN(h), wte[t]> alone (one of 2112 bench decisions had that down to 0.19)
        pairs = _token_pairs(t.vocab_size, w.device)
        w[:, pc["tok"]] = token_code_amplitude(config) * pair_codewords(t.vocab_size // 2, w.device)[pairs]
    edit("text.wte", wte)
    hd = t.dim // t.n_heads
    group = t.n_heads // t.n_kv_heads
    q_dim, kv_dim = t.n_heads * hd, t.n_kv_heads * hd
    for l in range(t.n_layers):
        p = f"text.blocks.{l}"
        kvh = l % t.n_kv_heads
        v_rows = [q_dim + kv_dim + kvh * hd + j for j in range(CODE_BITS)]
        # the query head that reads those value dims attends UNIFORMLY (q = 0): the text keys of this checkpoint carry larger
        # k vectors than the image keys and take most of a random head's softmax mass, which would let the default code of
        # the text tokens (below) outvote the image's; with equal scores the image's 729 keys always outweigh <= 64 text keys
        q_rows = list(range(kvh * group * hd, (kvh * group + 1) * hd))
        def qkv_w(w):
            w[v_rows] = 0.0
            w[q_rows] = 0.0
            for j, row in enumerate(v_rows):
                w[row, pc["p1"][j]] = CODE_RHO
                w[row, pc["zero"]] = -CODE_RHO
        edit(p + ".attn.qkv.weight", qkv_w)
        edit(p + ".attn.qkv.bias", zero_rows(v_rows + q_rows))
        def proj_w(w):
            w[pc["all_text"]] = 0.0
            for j in range(CODE_BITS):
                w[pc["p2"][j], kvh * group * hd + j] = 1.0   # the first query head that shares kv head kvh
        edit(p + ".attn.proj.weight", proj_w)
        edit(p + ".attn.proj.bias", zero_rows(pc["all_text"]))
        edit(p + ".mlp.fc2.weight", zero_rows(pc["all_text"]))
        edit(p + ".mlp.fc2.bias", zero_rows(pc["all_text"]))
        edit(p + ".ln.weight", lambda w: w.__setitem__(pc["all_text"], 1.0))
        edit(p + ".ln.bias", zero_rows(pc["all_text"]))
    edit("text.post_ln.weight", lambda w: w.__setitem__(pc["all_text"], 1.0))
    edit("text.post_ln.bias", zero_rows(pc["all_text"]))


REGION_FLAG = 4.0        # value of the flag coordinate in an encode_coordinate / encode_size embedding
REGION_GATE = 16.0       # gate gain: GATE x FLAG = 64 >> |h[p2]| switches a unit group off
REGION_MISS = -3.0       # contribution of a unit to the anchors whose bit does NOT match (a match contributes +1)
REGION_FC2_GAIN = 0.1    # the random part of the decoders' output layer (the planted anchor must stand clear of 1024 such logits)
COORD_ANCHORS = [32 + 64 * k for k in range(16)]    # coordinate bins the code can select: x = bin / 1024 in (0.03, 0.97)
SIZE_ANCHORS = [520 + 30 * k for k in range(16)]    # size bins: 2^((bin / 1023) 10 - 10) in (0.034, 0.70)


def region_anchor(bits, which: str) -> int:
    """The bin the planted region heads select for an image with code ``bits`` (+-1 x 6): ``which`` in x_first (object 0: bits
    0-3), y (bits 2-5), x_later (objects >= 1: bits 1-4), w (bits 0-3), h (bits 2-5); anchor index = sum (bit > 0) << n."""
    lo = {"x_first": 0, "y": 2, "x_later": 1, "w": 0, "h": 2}[which]
    k = sum((1 << n) for n in range(4) if bits[lo + n] > 0)
    return (SIZE_ANCHORS if which in ("w", "h") else COORD_ANCHORS)[k]


def _plant_region_heads(sd, config, dtype):
    """Region decoders whose 1024-bin argmax is a WELL-POSED output (round 6; reference region.py:32-93, loop moondream.py:653-733).
    With i.i.d. weights the best two of 1024 bins are 0-85 bf16 ulps apart and nothing could be compared at the object level
    (rounds 2-5: "objects up to the first narrow decision").  Planted:

      * encode_coordinate / encode_size write a FLAG (bias only) into a protected coordinate of the embedding they produce; no
        decoder layer writes it, so the last hidden state of that position still carries it: the heads know which step they are at
        (x of the first object: no flag; y: coordinate flag; x of a later object: size flag);
      * fc1: a pair of units gelu(+h[p2[j]] + gate), gelu(-h[p2[j]] + gate) per code bit and step -- the one whose sign matches the
        image's code bit is active with value |h[p2[j]]| (the accumulated code amplitude, several units), the other ~0;
      * fc2: an active unit adds +1 x its value to the 8 of 16 ANCHOR bins whose index has that bit, REGION_MISS x to the other 8:
        the anchor spelled by four code bits gets 4 |h|, every other anchor <= 0, every non-anchor bin only the (scaled down)
        random part.  The winner stands ~100 bf16 ulps clear.
    Objects of an image are therefore a function of its 6 code bits (``region_anchor``), reached through every decoder layer's
    attention exactly like the token decisions."""
    t, r = config.text, config.region
    pc = protected_coords(config)
    f32 = torch.float32
    flag_c, flag_s = pc["flag"]

    def edit(name, fn):
        w = sd[name].to(f32)
        fn(w)
        sd[name] = w.to(dtype)

    for enc, flag in (("region.coord_encoder", flag_c), ("region.size_encoder", flag_s)):
        edit(enc + ".weight", lambda w: w.__setitem__(pc["all_text"], 0.0))
        def bias(b, flag=flag):
            b[pc["all_text"]] = 0.0
            b[flag] = REGION_FLAG
        edit(enc + ".bias", bias)

    GF = REGION_GATE * REGION_FLAG

    def plant(prefix, groups, n_out_blocks):
        """groups: (first code bit, output block, anchors, gate) with gate = {flag coordinate: weight} and a bias."""
        n_units = 8 * len(groups)
        inner = sd[prefix + ".fc1.weight"].shape[0]
        u0 = inner - n_units
        def fc1_w(w):
            w[u0:] = 0.0
            for g, (lo, _blk, _anch, gate, _gb) in enumerate(groups):
                for n in range(4):
                    for s, sg in enumerate((1.0, -1.0)):
                        row = u0 + g * 8 + n * 2 + s
                        w[row, pc["p2"][lo + n]] = sg
                        for col, gw in gate.items():
                            w[row, col] = gw
        def fc1_b(b):
            for g, (_lo, _blk, _anch, _gate, gb) in enumerate(groups):
                b[u0 + g * 8 : u0 + g * 8 + 8] = gb
        def fc2_w(w):
            w *= REGION_FC2_GAIN
            w[:, u0:] = 0.0
            for g, (_lo, blk, anchors, _gate, _gb) in enumerate(groups):
                for n in range(4):
                    for s in range(2):
                        col = u0 + g * 8 + n * 2 + s
                        for k, bin_ in enumerate(anchors):
                            has = (k >> n) & 1
                            w[blk * 1024 + bin_, col] = 1.0 if has == (1 - s) else REGION_MISS
        edit(prefix + ".fc1.weight", fc1_w)
        edit(prefix + ".fc1.bias", fc1_b)
        edit(prefix + ".fc2.weight", fc2_w)

    plant("region.coord_decoder", [
        (0, 0, COORD_ANCHORS, {flag_c: -REGION_GATE, flag_s: -REGION_GATE}, 0.0),   # x of the first object: neither flag
        (2, 0, COORD_ANCHORS, {flag_c: REGION_GATE}, -GF),                            # y: the input was a coordinate embedding
        (1, 0, COORD_ANCHORS, {flag_s: REGION_GATE}, -GF),                            # x of a later object: the input was a size embedding
    ], 1)
    plant("region.size_decoder", [
        (0, 0, SIZE_ANCHORS, {}, 0.0),    # w
        (2, 1, SIZE_ANCHORS, {}, 0.0),    # h
    ], 2)


def _pair_perm(V: int):
    a = int(V * 0.6180339887) | 1  # odd multiplier near V/phi, made coprime with V
    while math.gcd(a, V) != 1:
        a += 2
    return a, 17


def _token_pairs(V: int, device="cpu") -> torch.Tensor:
    """pair index m = perm(t) >> 1 of every token t, perm(t) = (a t + b) mod V (the bijection _plant_lm_head uses)."""
    a, b = _pair_perm(V)
    t = torch.arange(V, dtype=torch.int64, device=device)
    m = ((a * t + b) % V) >> 1
    # the first SPECIAL_PAIRS pairs hold the special ids (eos / bos 0, answer 3 -- SUPPRESSED in every decode step,
    # moondream.py:517 --, thinking 4, coord 5, size 6, grounding 7 / 9): a token that owned one of them would have its
    # winner taken away (round 6: the one narrow decision of the first re-conditioned bench fixture was exactly that: the
    # code picked 3, 3 was -inf, and the next candidates were near-ties).  Their owners share the LAST pairs instead.
    return torch.where(m < SPECIAL_PAIRS, V // 2 - 1 - m, m)


def _plant_lm_head(sd, config, seed, device, dtype, beta=1.0, c_code=12.0, c_deep=0.8, c_tok=1.0, s0=0.1):
    """lm_head with a planted "bigram + image-code bit + deep context" structure.

    With i.i.d. random weights the top-1/top-2 logit gap is 0-3 bf16 ulps
    ("token-ID bit-exactness"), so greedy ids are decided
    by ties and cannot be compared between two correct implementations.  Here
    every token t owns a PAIR of candidate next tokens {2m, 2m+1},
    m = perm(t) >> 1: both rows carry beta * unit(wte[t]) (so the pair wins
    against the other V-2 rows by a wide margin once LN(h) has a component
    along wte[t], which the residual stream guarantees), and the two rows
    differ by +- (c_code * (e[p2[j]] - e[zero]) + c_deep * r):

      * c_code reads coordinate j = j(m) of the image code that the planted path of _plant_code_path accumulated in the last
        hidden state: LN(h)[p2[j]] - LN(h)[zero] = h[p2[j]] / sigma = bit_j x (something positive, O(1)) -- the WINNER of the
        pair is a discrete property of the image (64 classes), decided with a margin of several logit units through the
        attention of every decoder layer;
      * c_deep * <LN(h), r> for one fixed random direction r (pair's own token directions projected out) is the deep,
        continuous context feature rounds 1-5 used ALONE to pick the winner (c = 8).  Its sign is within bf16 implementation noise
        for 2-3 % of all decisions, so it can no longer decide (c_deep x 4.5 sigma stays below the code term) but still moves
        both logits by ~+-1: a wrong kernel shows in the teacher-forced logit comparison, not in a coin flip.

    WHICH pair wins is decided by the tok coordinates (the current token's RM(2,5) pair codeword, carried unchanged from its
    embedding: + c_tok x 32 a / sigma for its own pair, at most half of that for any other) on top of the beta term.  Pairs of
    special ids are nobody's pair (``_token_pairs``).  Row 0 (eos) gets a large negative bias so generation length is fixed by
    max_tokens.  Without an image (text-only query) every key carries the DEFAULT code at half amplitude: the decisions follow
    it with a margin of a few logit units, which the deep term can still overrule (those goldens stay filtered by margin).
    """
    t = config.text
    V, D = t.vocab_size, t.dim
    pc = protected_coords(config)
    beta = beta * 16.0 / math.sqrt(D)
    a, b = _pair_perm(V)
    a_inv = pow(a, -1, V)
    rows = torch.arange(V, device=device, dtype=torch.int64)
    # perm(t) = (a*t + b) mod V  ->  perm^-1(v) = a_inv * (v - b) mod V
    t_even = (a_inv * (((rows & ~1) - b) % V)) % V
    t_odd = (a_inv * (((rows | 1) - b) % V)) % V
    # Reductions (norms, dot products) run in float64 and are rounded to fp32 once, so
    # that CPU and GPU generation give the same bits despite different summation
    # orders; everything else is elementwise IEEE arithmetic.
    wte = sd["text.wte"].float()
    wte[:, pc["all_text"]] = 0.0   # the token directions of the pair rows stay clear of the protected coordinates
    unit = wte / wte.double().norm(dim=-1, keepdim=True).float()
    r = hash_uniform(D, _key("text.lm_head.context_dir", seed), device) * math.sqrt(3.0)
    r[pc["all_text"]] = 0.0
    r = r / r.double().norm().float()  # <LN(h), r> ~ N(0,1) for |LN(h)| ~ sqrt(D)
    sign = torch.where((rows & 1) == 0, 1.0, -1.0).to(torch.float32).unsqueeze(1)
    noise = hash_uniform(V * D, _key("text.lm_head.weight", seed), device).reshape(V, D)
    noise[:, pc["all_text"]] = 0.0
    ue, uo = unit[t_even], unit[t_odd]
    # owners of a pair: the two tokens the permutation maps onto it; none for the special pairs (rows 0 .. 2 SPECIAL_PAIRS - 1),
    # and the owners of special pair k additionally own pair V/2 - 1 - k (``_token_pairs``) -- added in a fixed order below
    special = (rows >> 1) < SPECIAL_PAIRS
    ue = torch.where(special.unsqueeze(1), torch.zeros_like(ue), ue)
    uo = torch.where(special.unsqueeze(1), torch.zeros_like(uo), uo)
    # context direction with the pair's own token directions projected out, so the
    # winning bit is not a function of the current token's embedding alone
    dot_e = (ue.double() @ r.double()).float().unsqueeze(1)
    dot_o = (uo.double() @ r.double()).float().unsqueeze(1)
    r_pair = r.unsqueeze(0) - dot_e * ue - dot_o * uo
    own = ue + uo
    for k in range(SPECIAL_PAIRS):
        for tok in (int((a_inv * ((2 * k - b) % V)) % V), int((a_inv * ((2 * k + 1 - b) % V)) % V)):
            u = unit[tok]
            for row in (V - 2 - 2 * k, V - 1 - 2 * k):       # the two rows of pair V/2 - 1 - k
                own[row] = own[row] + u
                r_pair[row] = r_pair[row] - (u.double() @ r.double()).float() * u
    w = beta * own + c_deep * sign * r_pair + noise * (s0 * math.sqrt(3.0 / D))
    # the code coordinate of pair m: a fixed hash of the pair index
    pair = rows >> 1
    j = ((pair * 2654435761) >> 7) % CODE_BITS
    p2 = torch.tensor(pc["p2"], device=device, dtype=torch.int64)[j]
    w[rows, p2] = c_code * sign[:, 0]
    w[:, pc["zero"]] = -c_code * sign[:, 0]
    # WHICH pair: both rows of pair m read the tok coordinates with m's codeword (minus its sum on the zero coordinate, so
    # that LayerNorm's mean shift cancels): + c_tok x 32 x amplitude / sigma for the pair of the current token, at most half
    # of that for any other pair
    all_cw = pair_codewords(V // 2 + SPECIAL_PAIRS, device)
    cw = all_cw[torch.where(pair < SPECIAL_PAIRS, V // 2 + pair, pair)]   # the special pairs' rows: codewords no token carries
    w[:, pc["tok"]] = c_tok * cw
    w[:, pc["zero"]] += -c_tok * cw.sum(dim=1)
    sd["text.lm_head.weight"] = w.to(dtype)
    bias = _tensor("text.lm_head.bias", (V,), 0.05, seed, device, torch.float32)
    bias[config.tokenizer.eos_id] = -60.0
    sd["text.lm_head.bias"] = bias.to(dtype)


# --------------------------------------------------------------------------
# Synthetic inputs (BASELINE.md section 3 / SURVEY.md section 8d)
# --------------------------------------------------------------------------
def synthetic_image_array(index: int, seed: int = 0, size=(378, 378)) -> np.ndarray:
    """HWC uint8 image: uniform pixel noise (``np.random.default_rng(seed+index)``, PCG64: stable) around a per-channel colour
    cast -- one of four levels per channel (``image_cast_levels``), i.e. one of 64 classes by ``index`` -- so that the image has
    a DISCRETE property (its 6 code bits) besides its noise: normalised value = level + CAST_NOISE * u, u uniform in [-1, 1)."""
    rng = np.random.default_rng(seed + index)
    noise = rng.integers(0, 256, (size[0], size[1], 3), dtype=np.uint8)
    level = np.array([CAST_LEVELS[lv] for lv in image_cast_levels(index)], dtype=np.float64)
    val = 127.5 + 127.5 * (level[None, None, :] + CAST_NOISE * (noise.astype(np.float64) - 127.5) / 127.5)
    return np.clip(np.rint(val), 0, 255).astype(np.uint8)


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
    Scales are large enough to move the logits.  Round 6: the variant also changes the DISCRETE outcome, on purpose and with
    a wide margin -- six rank rows of every block's qkv pair read (LN(x)[p1[j]] - LN(x)[zero]) and add -2 RHO x that to the
    planted value dims (``_plant_code_path``), so the image code arrives NEGATED in p2 and every pair decision flips to the
    other member: a missing or mis-applied side path gives the base model's ids, not a near-tie.  The random part of the
    variant leaves the protected coordinates and the planted head's rows alone (its B rows there are zero)."""
    t = config.text
    pc = protected_coords(config)
    hd = t.dim // t.n_heads
    group = t.n_heads // t.n_kv_heads
    q_dim, kv_dim = t.n_heads * hd, t.n_kv_heads * hd
    assert rank >= CODE_BITS + 1
    blocks = {}
    for i in range(t.n_layers):
        def pair(name, out_f, in_f, gain, zero_out_rows=()):
            a = _tensor(f"lora.{i}.{name}.A", (rank, in_f), 1.0 / math.sqrt(in_f), seed, device, torch.float32)
            b = _tensor(f"lora.{i}.{name}.B", (out_f, rank), gain / math.sqrt(rank), seed, device, torch.float32)
            if len(zero_out_rows):
                b[list(zero_out_rows)] = 0.0
            return a, b

        kvh = i % t.n_kv_heads
        v_rows = [q_dim + kv_dim + kvh * hd + j for j in range(CODE_BITS)]
        q_rows = list(range(kvh * group * hd, (kvh * group + 1) * hd))
        qa, qb = pair("qkv", t.qkv_dim, t.dim, 0.5, v_rows + q_rows)
        qa[:CODE_BITS] = 0.0
        qb[:, :CODE_BITS] = 0.0
        for j in range(CODE_BITS):
            qa[j, pc["p1"][j]], qa[j, pc["zero"]] = 1.0, -1.0
            qb[v_rows[j], j] = -2.0 * CODE_RHO
        pa, pb = pair("proj", t.dim, t.dim, 0.5, pc["all_text"])
        f1a, f1b = pair("fc1", t.ff_dim, t.dim, 0.5)
        f2a, f2b = pair("fc2", t.dim, t.ff_dim, 0.25, pc["all_text"])
        cast = lambda x: x.to(dtype)
        blocks[str(i)] = {
            "attn": {"qkv": {"A": cast(qa), "B": cast(qb)}, "proj": {"A": cast(pa), "B": cast(pb)}},
            "mlp": {"fc1": {"A": cast(f1a), "B": cast(f1b)}, "fc2": {"A": cast(f2a), "B": cast(f2b)}},
        }
    return {"text": {"blocks": blocks}}
