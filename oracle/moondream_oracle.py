"""CPU oracle for the Moondream inference hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain fp32 tensor arithmetic with explicit bf16 rounding
points, the algorithm the reference executes through PyTorch ATen ops
(SURVEY.md section 8a).  It exists to *check* the HIP path.  Nothing under
``moondream_amd/`` may import it; only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg do.

Pinning status: the reference's own tests hold NO golden vectors for this path
("parity unpinned" upstream).  This oracle is therefore
pinned against outputs of the reference code itself, run in the build
container by ``oracle/make_golden.py`` and committed under ``tests/golden/``
(``tests/test_oracle_golden.py`` checks every stage).

Rounding model (what "bf16" means below): every reference op takes bf16
tensors, computes in fp32 and rounds its result to bf16 once.  ``_r`` is that
rounding.  Accumulation order inside a contraction is NOT part of the model
(it differs between BLAS back-ends); comparisons against this oracle are
therefore tolerance-based for activations and exact for integer outputs
(token ids) wherever the top-1/top-2 logit margin exceeds the tolerance.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BF16 = torch.bfloat16


def _r(x: torch.Tensor) -> torch.Tensor:
    """Round an fp32 result to bf16 (one rounding, ties-to-even)."""
    return x.to(BF16)


# --------------------------------------------------------------------------
# primitive ops
# --------------------------------------------------------------------------
_FP32_CACHE: Dict[int, torch.Tensor] = {}


def cache_fp32_weights(sd: Dict[str, torch.Tensor]) -> None:
    """Keep fp32 copies of every 2-D weight so that the timed CPU baseline measures
    the contractions, not bf16 -> fp32 conversion of 2B parameters per call."""
    for v in sd.values():
        if v.dim() == 2 and v.dtype == BF16:
            _FP32_CACHE[id(v)] = v.float()


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], fast: bool = False):
    """y = bf16( fp32(x) . fp32(w)^T + fp32(b) ).  reference: layers.py:34-35
    (F.linear on bf16 operands).  ``fast`` routes the same contraction through
    the CPU bf16 GEMM (fp32 accumulate, one rounding) to avoid materialising
    fp32 copies of 2B parameters; the rounding points are identical."""
    if fast:
        return F.linear(x, w, b)
    w32 = _FP32_CACHE.get(id(w))
    y = x.float() @ (w.float() if w32 is None else w32).t()
    if b is not None:
        y = y + b.float()
    return _r(y)


# Timed-baseline switch (bench.py's cpu_baseline only): when set, the elementwise / normalisation ops below are
# issued as the single ATen calls the reference makes on bf16 tensors (F.layer_norm, F.gelu, bf16 adds) instead
# of their explicit fp32 restatements -- same roundings, a fraction of the memory passes.  Never set by tests.
ATEN_CALLS = False


def add_bf16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """bf16 + bf16 -> bf16 (one rounding): the reference's residual adds (vision.py:70-71, text.py:158)."""
    if ATEN_CALLS:
        return a + b
    return _r(a.float() + b.float())


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5):
    """reference: layers.py:118-119 (F.layer_norm, default eps, affine).
    Statistics in fp32 over the last dim, biased variance, one rounding."""
    if ATEN_CALLS:
        return F.layer_norm(x, (x.shape[-1],), w, b, eps)
    xf = x.float()
    mu = xf.mean(dim=-1, keepdim=True)
    var = ((xf - mu) ** 2).mean(dim=-1, keepdim=True)
    y = (xf - mu) * torch.rsqrt(var + eps) * w.float() + b.float()
    return _r(y)


def gelu_tanh(x: torch.Tensor):
    """reference: layers.py:24-25 (F.gelu(approximate="tanh")), fp32 inside."""
    if ATEN_CALLS:
        return F.gelu(x, approximate="tanh")
    xf = x.float()
    k = math.sqrt(2.0 / math.pi)
    inner = k * (xf + 0.044715 * xf * xf * xf)
    return _r(0.5 * xf * (1.0 + torch.tanh(inner)))


def lora_delta(x, pair, fast=False):
    """(x A^T) B^T as two bf16 linears.  reference: layers.py:132,141, text.py:32,55."""
    return linear(linear(x, pair["A"], None, fast), pair["B"], None, fast)


def mlp(x, sd, prefix, fast=False, lora=None):
    """fc1 -> gelu(tanh) -> fc2; with ``lora`` ({"fc1": {A, B}, "fc2": {A, B}}) each linear's output gets its
    low-rank delta added as a bf16 tensor add BEFORE the next op.  reference: layers.py:129-146."""
    h = linear(x, sd[prefix + ".fc1.weight"], sd[prefix + ".fc1.bias"], fast)
    if lora is not None:
        h = add_bf16(h, lora_delta(x, lora["fc1"], fast))
    h = gelu_tanh(h)
    y = linear(h, sd[prefix + ".fc2.weight"], sd[prefix + ".fc2.bias"], fast)
    if lora is not None:
        y = add_bf16(y, lora_delta(h, lora["fc2"], fast))
    return y


def softmax_attention(q, k, v, allowed: Optional[torch.Tensor], scale: float, fast: bool = False):
    """softmax(q k^T * scale) v per head, fp32 scores, probabilities rounded to
    bf16 for the second contraction, fp32 row-sum normalisation at the end
    (flash-style; reference: F.scaled_dot_product_attention at layers.py:163
    and text.py:48-50).  q [.., Tq, d], k/v [.., Tk, d]; ``allowed`` is a bool
    [Tq, Tk] (True = attend) or None."""
    if fast:
        # the very call the reference makes (bf16 operands, optional bool mask): used by the timed CPU baseline
        if q.dim() == 3:  # [H, T, d]: give SDPA the reference's 4-D shapes (batch 1, mask [1, 1, Tq, Tk]) so that it
            # takes the same fused CPU kernel as in the reference instead of the unfused math path
            m4 = None if allowed is None else allowed[None, None]
            return F.scaled_dot_product_attention(q[None], k[None], v[None], attn_mask=m4, scale=scale)[0]
        return F.scaled_dot_product_attention(q, k, v, attn_mask=allowed, scale=scale)
    s = (q.float() @ k.float().transpose(-1, -2)) * scale
    if allowed is not None:
        s = s.masked_fill(~allowed, float("-inf"))
    m = s.amax(dim=-1, keepdim=True)
    p = torch.exp(s - m)
    l = p.sum(dim=-1, keepdim=True)
    o = (_r(p).float() @ v.float()) / l
    return _r(o)


# --------------------------------------------------------------------------
# vision
# --------------------------------------------------------------------------
def pixel_lut() -> torch.Tensor:
    """The 256 possible results of the reference's pixel normalisation
    ``u8 -> bf16 -> div_(255.0) -> sub_(0.5) -> div_(0.5)`` with a bf16 rounding
    after every in-place op (reference: vision.py:33-40)."""
    x = _r(torch.arange(256, dtype=torch.float32))
    x = _r(x.float() / 255.0)
    x = _r(x.float() - 0.5)
    x = _r(x.float() / 0.5)
    return x


def normalize_crops(crops_u8: np.ndarray) -> torch.Tensor:
    """[N,H,W,3] uint8 -> [N,3,H,W] bf16.  reference: vision.py:32-40."""
    lut = pixel_lut()
    t = torch.from_numpy(np.ascontiguousarray(crops_u8)).long()
    return lut[t].permute(0, 3, 1, 2).contiguous()


def patchify(x: torch.Tensor, p: int) -> torch.Tensor:
    """[N,C,H,W] -> [N,(H/p)(W/p), C*p*p] with feature order (c, py, px).
    reference: vision.py:44-61."""
    n, c, h, w = x.shape
    gh, gw = h // p, w // p
    out = torch.empty(n, gh * gw, c * p * p, dtype=x.dtype)
    for gy in range(gh):
        for gx in range(gw):
            blk = x[:, :, gy * p : (gy + 1) * p, gx * p : (gx + 1) * p]
            out[:, gy * gw + gx, :] = blk.reshape(n, -1)
    return out


def vit_attention(x, sd, prefix, n_heads, fast=False):
    """qkv linear, split into q|k|v thirds, per-head softmax attention without a
    mask, scale 1/sqrt(head_dim), output projection.  reference:
    layers.py:155-166."""
    n, t, d = x.shape
    hd = d // n_heads
    qkv = linear(x, sd[prefix + ".qkv.weight"], sd[prefix + ".qkv.bias"], fast)
    q, k, v = (
        qkv[..., i * d : (i + 1) * d].reshape(n, t, n_heads, hd).permute(0, 2, 1, 3)
        for i in range(3)
    )
    o = softmax_attention(q, k, v, None, 1.0 / math.sqrt(hd), fast)
    o = o.permute(0, 2, 1, 3).reshape(n, t, d)
    return linear(o, sd[prefix + ".proj.weight"], sd[prefix + ".proj.bias"], fast)


def vision_encoder(x_bchw: torch.Tensor, sd, cfg, tap: Optional[dict] = None, fast=False):
    """reference: vision.py:64-74.  Residual adds are bf16 + bf16 -> bf16."""
    v = cfg.vision
    x = patchify(x_bchw, v.enc_patch_size).to(sd["vision.patch_emb.weight"].device)  # (the weights' device: Oracle(device=...))
    x = linear(x, sd["vision.patch_emb.weight"], sd["vision.patch_emb.bias"], fast)
    x = _r(x.float() + sd["vision.pos_emb"].float())
    if tap is not None:
        tap["vit.embed"] = x
    for i in range(v.enc_n_layers):
        p = f"vision.blocks.{i}"
        a = vit_attention(
            layer_norm(x, sd[p + ".ln1.weight"], sd[p + ".ln1.bias"]), sd, p + ".attn", v.enc_n_heads, fast
        )
        x = add_bf16(x, a)
        m = mlp(layer_norm(x, sd[p + ".ln2.weight"], sd[p + ".ln2.bias"]), sd, p + ".mlp", fast)
        x = add_bf16(x, m)
        if tap is not None and (i in (0, v.enc_n_layers - 1) or tap.get("__all_blocks__")):
            tap[f"vit.block{i}"] = x
    x = layer_norm(x, sd["vision.post_ln.weight"], sd["vision.post_ln.bias"])
    if tap is not None:
        tap["vit.out"] = x
    return x


def select_tiling(height: int, width: int, crop: int, max_crops: int) -> Tuple[int, int]:
    """(rows, cols) of local crops.  reference: image_crops.py:17-50."""
    if height <= crop or width <= crop:
        return (1, 1)
    need_h, need_w = math.ceil(height / crop), math.ceil(width / crop)
    if need_h * need_w > max_crops:
        f = math.sqrt(max_crops / (need_h * need_w))
        return (max(1, math.floor(need_h * f)), max(1, math.floor(need_w * f)))
    th = max(math.floor(math.sqrt(max_crops * height / width)), need_h)
    tw = max(math.floor(math.sqrt(max_crops * width / height)), need_w)
    if th * tw > max_crops:
        if tw > th:
            tw = math.floor(max_crops / th)
        else:
            th = math.floor(max_crops / tw)
    return (max(1, th), max(1, tw))


def stitch_local_features(local: torch.Tensor, tiling: Tuple[int, int], margin: int) -> torch.Tensor:
    """[n_local, g, g, D] per-crop feature grids -> one [(g-2m)*th+2m, (g-2m)*tw+2m, D]
    grid keeping each crop's interior plus the outer margins at the image
    border.  reference: image_crops.py:170-231 with patch_size=1
    (moondream.py:221-226)."""
    th, tw = tiling
    g = local.shape[1]
    inner = g - 2 * margin
    out = torch.zeros(inner * th + 2 * margin, inner * tw + 2 * margin, local.shape[-1], dtype=local.dtype)
    for idx in range(local.shape[0]):
        ty, tx = divmod(idx, tw)
        y0 = 0 if ty == 0 else margin
        y1 = g if ty == th - 1 else g - margin
        x0 = 0 if tx == 0 else margin
        x1 = g if tx == tw - 1 else g - margin
        out[ty * inner + y0 : ty * inner + y1, tx * inner + x0 : tx * inner + x1] = local[idx, y0:y1, x0:x1]
    return out


def adaptive_avg_pool_hw(x_hwc: torch.Tensor, out_hw: int) -> torch.Tensor:
    """[H,W,C] -> [out,out,C]; bin i covers [floor(i*H/out), ceil((i+1)*H/out)),
    mean in fp32, one rounding.  reference: vision.py:83-86
    (F.adaptive_avg_pool2d on the permuted tensor)."""
    h, w, c = x_hwc.shape
    out = torch.empty(out_hw, out_hw, c, dtype=x_hwc.dtype)
    xf = x_hwc.float()
    for i in range(out_hw):
        y0, y1 = (i * h) // out_hw, -((-(i + 1) * h) // out_hw)
        for j in range(out_hw):
            x0, x1 = (j * w) // out_hw, -((-(j + 1) * w) // out_hw)
            out[i, j] = _r(xf[y0:y1, x0:x1].mean(dim=(0, 1)))
    return out


def vision_projection(global_feat, stitched, sd, cfg, fast=False):
    """reference: vision.py:77-89."""
    g = cfg.vision.enc_n_layers  # sic: the reference uses enc_n_layers as the grid side
    pooled = adaptive_avg_pool_hw(stitched.cpu(), g).reshape(g * g, -1).to(global_feat.device)
    return mlp(torch.cat([global_feat, pooled], dim=-1), sd, "vision.proj_mlp", fast)


def run_vision(crops_u8: np.ndarray, tiling, sd, cfg, tap=None, fast=False):
    """reference: moondream.py:206-228 (_run_vision_encoder)."""
    v = cfg.vision
    x = normalize_crops(crops_u8)
    feats = vision_encoder(x, sd, cfg, tap, fast)
    g = v.enc_n_layers
    local = feats[1:].reshape(-1, g, g, v.enc_dim).cpu()  # the stitch / pool restatements are host loops
    stitched = stitch_local_features(local, tiling, v.overlap_margin)
    out = vision_projection(feats[0], stitched, sd, cfg, fast)
    if tap is not None:
        tap["vis.proj"] = out
    return out


# --------------------------------------------------------------------------
# text
# --------------------------------------------------------------------------
def rope_table(rot_half: int, n_pos: int, theta: float = 10000.0):
    """fp32 cos/sin [n_pos, rot_half]; freq_j = theta^(-(2j)/(2*rot_half)).
    reference: rope.py:6-17 (dim = 2*rot_half there)."""
    dim = 2 * rot_half
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[:rot_half] / dim))
    ang = torch.arange(n_pos, dtype=torch.float32).unsqueeze(1) * freqs.unsqueeze(0)
    # the reference builds the table as exp(i * angle) in complex64 (rope.py:16-17);
    # its real/imag parts are not bit-identical to cos()/sin() of the fp32 angle
    cis = torch.exp(1j * ang)
    return cis.real.contiguous(), cis.imag.contiguous()


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, pos: torch.Tensor, rot_dim: int):
    """x [H, T, hd].  The first rot_dim features are read half-split
    (re = [0, rot/2), im = [rot/2, rot)), rotated in fp32 and written
    INTERLEAVED (re0, im0, re1, im1, ...); the rest pass through.
    reference: rope.py:20-48 (interleave=False input, stack+flatten output)."""
    half = rot_dim // 2
    re, im = x[..., :half].float(), x[..., half:rot_dim].float()
    c, s = cos[pos].unsqueeze(0), sin[pos].unsqueeze(0)
    out_re = re * c - im * s
    out_im = re * s + im * c
    rot = torch.stack((out_re, out_im), dim=-1).flatten(-2)
    return torch.cat([_r(rot), x[..., rot_dim:]], dim=-1)


def prefix_lm_allowed(q_pos: torch.Tensor, n_kv: int, prefix: int) -> torch.Tensor:
    """bool [Tq, n_kv]: key j visible to query at position i iff j <= i, or both
    lie inside the bidirectional prefix.  reference: moondream.py:138-146."""
    j = torch.arange(n_kv, device=q_pos.device).unsqueeze(0)
    i = q_pos.unsqueeze(1)
    return (j <= i) | ((i < prefix) & (j < prefix))


@dataclass
class OracleKV:
    """Per-layer K/V slabs [H_kv, max_context, hd] (reference: moondream.py:62-78)."""

    k: List[torch.Tensor] = field(default_factory=list)
    v: List[torch.Tensor] = field(default_factory=list)

    @classmethod
    def empty(cls, cfg, device="cpu"):
        t = cfg.text
        shape = (t.n_kv_heads, t.max_context, t.head_dim)
        return cls(
            [torch.zeros(shape, dtype=BF16, device=device) for _ in range(t.n_layers)],
            [torch.zeros(shape, dtype=BF16, device=device) for _ in range(t.n_layers)],
        )

    def clone(self):
        return OracleKV([a.clone() for a in self.k], [a.clone() for a in self.v])


def text_attention(x, sd, prefix, cfg, layer, kv: OracleKV, pos: torch.Tensor, cos, sin, allowed, fast=False, lora=None):
    """reference: text.py:16-60.  x [T, D]."""
    t = cfg.text
    T = x.shape[0]
    hd = t.head_dim
    qkv = linear(x, sd[prefix + ".qkv.weight"], sd[prefix + ".qkv.bias"], fast)
    if lora is not None:
        qkv = add_bf16(qkv, lora_delta(x, lora["qkv"], fast))  # text.py:31-32
    qd, kd = t.n_heads * hd, t.n_kv_heads * hd
    q = qkv[:, :qd].reshape(T, t.n_heads, hd).permute(1, 0, 2)
    k = qkv[:, qd : qd + kd].reshape(T, t.n_kv_heads, hd).permute(1, 0, 2)
    v = qkv[:, qd + kd :].reshape(T, t.n_kv_heads, hd).permute(1, 0, 2)
    q = apply_rope(q, cos, sin, pos, t.rot_dim)
    k = apply_rope(k, cos, sin, pos, t.rot_dim)
    kv.k[layer][:, pos] = k
    kv.v[layer][:, pos] = v
    n_kv = int(pos.max()) + 1  # slots beyond are masked out in the reference
    if fast:
        n_kv = t.max_context  # the reference attends over all cache slots under its bool mask (text.py:48-50)
    kk, vv = kv.k[layer][:, :n_kv], kv.v[layer][:, :n_kv]
    if t.n_kv_heads != t.n_heads:
        rep = t.n_heads // t.n_kv_heads
        kk, vv = kk.repeat_interleave(rep, 0), vv.repeat_interleave(rep, 0)
    o = softmax_attention(q, kk, vv, allowed[:, :n_kv], 1.0 / math.sqrt(hd), fast)
    o = o.permute(1, 0, 2).reshape(T, qd)
    out = linear(o, sd[prefix + ".proj.weight"], sd[prefix + ".proj.bias"], fast)
    if lora is not None:
        out = add_bf16(out, lora_delta(x, lora["proj"], fast))  # text.py:55: the pair is fed the BLOCK INPUT x, not `o`
    return out


def text_decoder(x, sd, cfg, kv: OracleKV, pos: torch.Tensor, cos, sin, tap=None, fast=False, prefix=None, lora=None):
    """Parallel attention + MLP off ONE LayerNorm, then two left-to-right bf16
    adds.  reference: text.py:128-160.  x [T, D] -> [T, D]; writes kv.
    ``prefix`` overrides the bidirectional prefix length (0 = the plain causal
    mask of a text-only query, moondream.py:571-575)."""
    t = cfg.text
    allowed = prefix_lm_allowed(pos, t.max_context, t.prefix_attn if prefix is None else prefix)
    for i in range(t.n_layers):
        p = f"text.blocks.{i}"
        h = layer_norm(x, sd[p + ".ln.weight"], sd[p + ".ln.bias"])
        ll = None if lora is None else lora["text"]["blocks"][str(i)]  # text.py:137-140
        a = text_attention(h, sd, p + ".attn", cfg, i, kv, pos, cos, sin, allowed, fast, None if ll is None else ll["attn"])
        m = mlp(h, sd, p + ".mlp", fast, None if ll is None else ll["mlp"])
        x = add_bf16(add_bf16(x, a), m)
        if tap is not None and (i in (0, t.n_layers - 1) or tap.get("__all_blocks__")):
            tap[f"text.block{i}"] = x
    return x


def lm_head(hidden_last: torch.Tensor, sd, fast=False):
    """[D] or [1, D] last-token hidden -> logits [V].  reference: text.py:163-167."""
    h = layer_norm(hidden_last.reshape(1, -1), sd["text.post_ln.weight"], sd["text.post_ln.bias"])
    return linear(h, sd["text.lm_head.weight"], sd["text.lm_head.bias"], fast)[0]


# --------------------------------------------------------------------------
# sampling
# --------------------------------------------------------------------------
def sampling_probs(logits: torch.Tensor, temperature: float) -> torch.Tensor:
    """reference: moondream.py:316,526 -- softmax(logits / T) on bf16 tensors: the quotient is a
    bf16 tensor, the softmax runs in fp32 and is rounded once."""
    z = _r(logits.float() / temperature)
    return _r(torch.softmax(z.float(), dim=-1))


def apply_top_p(probs: torch.Tensor, top_p: float) -> torch.Tensor:
    """Nucleus filter of one row, restated with its rounding points (reference:
    moondream.py:270-278).  In descending-probability order: the running sum is accumulated in fp32
    and rounded to bf16 per element (torch.cumsum on a bf16 tensor), ``csum - p`` is a bf16
    subtraction and the comparison casts the python scalar top_p to bf16; survivors are divided by
    their bf16 sum (fp32 accumulate) in bf16 and put back at their ids."""
    assert probs.dim() == 1 and probs.dtype == BF16
    order = torch.argsort(probs.float(), descending=True, stable=True)
    ps = probs[order]
    run = _r(torch.cumsum(ps.float(), dim=0))
    before = _r(run.float() - ps.float())
    keep = before.float() <= float(_r(torch.tensor(top_p)))
    kept = torch.where(keep, ps.float(), torch.zeros(()))
    denom = float(_r(kept.sum()))
    out = torch.zeros_like(probs)
    out[order] = _r(kept / denom)
    return out


# --------------------------------------------------------------------------
# end-to-end drivers
# --------------------------------------------------------------------------
@dataclass
class OracleRun:
    tokens: List[int]
    margins: List[float]  # top1 - top2 logit gap at every sampling point
    logits: List[torch.Tensor]
    pos: int


class Oracle:
    """Greedy encode_image + generate, B=1, like the reference's sequential path.

    ``device``: where the tensor arithmetic runs.  "cpu" is the oracle proper.  With ``fast=True`` and a GPU device the SAME
    calls -- the reference's own ATen ops (F.linear, F.scaled_dot_product_attention, F.layer_norm, F.gelu) in the reference's
    order -- run through torch-ROCm's kernels: SURVEY section 8(c)'s "second oracle" for bf16 last-bit behaviour, i.e. what
    happens to the reference's ids when nothing but the BLAS / attention backend changes (bench.py's parity calibration)."""

    def __init__(self, cfg, state_dict: Dict[str, torch.Tensor], fast: bool = False, device="cpu"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.sd = {k: v.detach().to(self.device) for k, v in state_dict.items()}
        self.fast = fast
        self.lora = None  # nested LoRA dict (lora.py:54-79) applied by every decoder call below when set
        cos, sin = rope_table(cfg.text.rot_dim // 2, cfg.text.max_context)
        self.cos, self.sin = cos.to(self.device), sin.to(self.device)

    def embed(self, ids) -> torch.Tensor:
        """reference: text.py:12-13."""
        return self.sd["text.wte"][torch.as_tensor(ids, dtype=torch.long, device=self.device)]

    def encode_image(self, crops_u8: np.ndarray, tiling, tap=None) -> Tuple[int, OracleKV]:
        """reference: moondream.py:230-268.  Returns (pos, kv)."""
        img = run_vision(crops_u8, tiling, self.sd, self.cfg, tap, self.fast)
        x = torch.cat([self.embed([self.cfg.tokenizer.bos_id]), img], dim=0)
        kv = OracleKV.empty(self.cfg, self.device)
        pos = torch.arange(x.shape[0], device=self.device)
        text_decoder(x, self.sd, self.cfg, kv, pos, self.cos, self.sin, tap, self.fast, lora=self.lora)
        return x.shape[0], kv

    def prefill_prompt(self, prompt_ids, pos0: int, kv: OracleKV, tap=None, prompt_emb=None, prefix=None):
        """reference: moondream.py:280-321 (greedy branch)."""
        x = self.embed(prompt_ids) if prompt_emb is None else prompt_emb
        pos = torch.arange(pos0, pos0 + x.shape[0], device=self.device)
        h = text_decoder(x, self.sd, self.cfg, kv, pos, self.cos, self.sin, tap, self.fast, prefix, lora=self.lora)
        logits = lm_head(h[-1], self.sd, self.fast)
        return logits, h, pos0 + x.shape[0]

    def decode_token(self, emb: torch.Tensor, pos: int, kv: OracleKV, prefix=None):
        """reference: moondream.py:183-192.  emb [1, D]."""
        h = text_decoder(emb, self.sd, self.cfg, kv, torch.tensor([pos], device=self.device), self.cos, self.sin, None, self.fast, prefix, lora=self.lora)
        return lm_head(h[-1], self.sd, self.fast), h

    @staticmethod
    def _argmax_margin(logits: torch.Tensor):
        logits = logits.float().cpu()  # (ties: the reference's argmax runs on the CPU)
        top = torch.topk(logits, 2)
        # torch.argmax returns the lowest index among ties on CPU; topk need not
        tok = int(torch.argmax(logits))
        return tok, float(top.values[0] - top.values[1])

    def generate(
        self,
        prompt_ids,
        pos0: int,
        kv: OracleKV,
        max_tokens: int,
        eos_id: Optional[int] = None,
        forced: Optional[List[int]] = None,
        keep_logits: bool = True,
        prefix: Optional[int] = None,
    ) -> OracleRun:
        """Greedy answer generation.  reference: moondream.py:434-539 with
        temperature == 0: prefill the prompt, then per token: stop on eos or
        max_tokens, embed, decode, suppress ``answer_id`` (moondream.py:517),
        argmax.  ``forced`` (teacher forcing) replaces the chosen token at each
        step by a given id while still recording the logits."""
        tk = self.cfg.tokenizer
        eos = tk.eos_id if eos_id is None else eos_id
        logits, _, pos = self.prefill_prompt(prompt_ids, pos0, kv, prefix=prefix)
        out, margins, all_logits = [], [], []
        tok, mg = self._argmax_margin(logits)
        step = 0
        while True:
            margins.append(mg)
            if keep_logits:
                all_logits.append(logits.clone())
            if forced is not None and step < len(forced):
                tok = forced[step]
            if (eos is not None and tok == eos) or step >= max_tokens:
                break
            out.append(tok)
            logits, _ = self.decode_token(self.embed([tok]), pos, kv, prefix)
            logits[tk.answer_id] = float("-inf")
            pos += 1
            step += 1
            tok, mg = self._argmax_margin(logits)
        return OracleRun(out, margins, all_logits, pos)

    # ---- region head -------------------------
    def fourier(self, x: torch.Tensor, w: torch.Tensor):
        """reference: region.py:12-29; bf16 matmul, bf16 scalar multiply, then
        cos/sin in bf16."""
        f = _r(_r(x.float() * (2 * math.pi)).float() @ w.float())
        return torch.cat([_r(torch.cos(f.float())), _r(torch.sin(f.float()))], dim=-1)

    def encode_coordinate(self, c: torch.Tensor):
        """reference: region.py:32-43."""
        return linear(
            self.fourier(c, self.sd["region.coord_features"]),
            self.sd["region.coord_encoder.weight"],
            self.sd["region.coord_encoder.bias"],
            self.fast,
        )

    def decode_coordinate(self, h: torch.Tensor):
        """reference: region.py:46-57."""
        return mlp(h, self.sd, "region.coord_decoder", self.fast)

    def encode_size(self, s: torch.Tensor):
        """reference: region.py:60-71."""
        return linear(
            self.fourier(s, self.sd["region.size_features"]),
            self.sd["region.size_encoder.weight"],
            self.sd["region.size_encoder.bias"],
            self.fast,
        )

    def decode_size(self, h: torch.Tensor):
        """reference: region.py:74-93."""
        return mlp(h, self.sd, "region.size_decoder", self.fast).reshape(2, -1)

    def encode_spatial_refs(self, spatial_refs):
        """reference: region.py:96-136: points -> (x, y); boxes -> (centre x, centre y) + (w, h)."""
        coords, sizes = [], []
        for ref in spatial_refs:
            if len(ref) == 2:
                coords += [ref[0], ref[1]]
            else:
                coords += [(ref[0] + ref[2]) / 2, (ref[1] + ref[3]) / 2]
                sizes.append([ref[2] - ref[0], ref[3] - ref[1]])
        c = self.encode_coordinate(torch.tensor(coords, dtype=BF16).view(-1, 1))
        s = self.encode_size(torch.tensor(sizes, dtype=BF16)) if sizes else None
        return c, s

    def spatial_prompt(self, head_ids, spatial_refs, tail_ids):
        """Prompt ids + embeddings with coordinate / size placeholders replaced by the encoded
        refs (reference: moondream.py:577-604 builds the ids, moondream.py:293-301 injects)."""
        tk = self.cfg.tokenizer
        ids = list(head_ids)
        for ref in spatial_refs:
            ids += [tk.coord_id, tk.coord_id] if len(ref) == 2 else [tk.coord_id, tk.coord_id, tk.size_id]
        ids += list(tail_ids)
        emb = self.embed(ids).clone()
        c, s = self.encode_spatial_refs(spatial_refs)
        idt = torch.tensor(ids)
        emb[idt == tk.coord_id] = c
        if s is not None:
            emb[idt == tk.size_id] = s
        return ids, emb

    def generate_points(self, hidden: torch.Tensor, next_token: int, pos: int, kv: OracleKV, include_size: bool,
                        max_objects: int, trace: Optional[list] = None):
        """reference: moondream.py:653-733.  hidden [1, D] = last prompt position.  Per object: x bin
        from the coordinate head -> embed -> decoder step -> y bin -> (embed -> step -> w, h bins)
        -> embed -> step -> lm_head argmax (eos ends the loop).  ``trace`` receives every decision
        as (kind, logits) in loop order."""
        out = []
        eos = self.cfg.tokenizer.eos_id

        def step(emb):
            nonlocal pos
            logits, h = self.decode_token(emb.reshape(1, -1), pos, kv)
            pos += 1
            return logits, h[-1:].clone()

        def note(kind, lg):
            if trace is not None:
                trace.append((kind, lg.clone()))

        while next_token != eos and len(out) < max_objects:
            x_logits = self.decode_coordinate(hidden)
            note("x", x_logits)
            x_center = torch.argmax(x_logits.float(), dim=-1) / x_logits.size(-1)  # int64 / int -> fp32
            _, hidden = step(self.encode_coordinate(x_center.to(BF16)))
            y_logits = self.decode_coordinate(hidden)
            note("y", y_logits)
            y_center = torch.argmax(y_logits.float(), dim=-1) / y_logits.size(-1)
            emb = self.encode_coordinate(y_center.to(BF16))
            if include_size:
                _, hidden = step(emb)
                size_logits = self.decode_size(hidden)
                note("w", size_logits[0])
                note("h", size_logits[1])
                w_bin = torch.argmax(size_logits[0].float(), dim=-1)
                h_bin = torch.argmax(size_logits[1].float(), dim=-1)
                w = torch.pow(2.0, (w_bin.float() / 1023.0) * 10.0 - 10.0)
                h = torch.pow(2.0, (h_bin.float() / 1023.0) * 10.0 - 10.0)
                emb = self.encode_size(torch.tensor([w, h], dtype=BF16))
                xc, yc, wf, hf = x_center.item(), y_center.item(), w.item(), h.item()
                out.append({"x_min": xc - wf / 2, "y_min": yc - hf / 2, "x_max": xc + wf / 2, "y_max": yc + hf / 2})
            else:
                out.append({"x": x_center.item(), "y": y_center.item()})
            logits, hidden = step(emb)
            note("next", logits)
            next_token = int(torch.argmax(logits.float()))
        return out

    def detect_like(self, kind: str, crops_u8, tiling, object_ids, max_objects: int, trace: Optional[list] = None):
        """reference: detect moondream.py:735-781 / point moondream.py:783-829."""
        tpl = self.cfg.tokenizer.templates[kind]
        pos, kv = self.encode_image(crops_u8, tiling)
        prompt = list(tpl["prefix"]) + list(object_ids) + list(tpl["suffix"])
        logits, h, pos = self.prefill_prompt(prompt, pos, kv)
        nxt = int(torch.argmax(logits.float()))
        return self.generate_points(h[-1:].clone(), nxt, pos, kv, kind == "detect", max_objects, trace)

    def generate_reasoning(self, prompt_ids, pos0: int, kv: OracleKV, max_tokens: int):
        """Greedy reasoning text.  reference: moondream.py:323-432 at temperature 0: prefill the prompt
        (which ends with the thinking token), then per token: stop on ``answer_id`` or max_tokens; a
        ``coord`` token is grounded through the region head (coordinate decoded from the hidden state
        that predicted it, fed back through the coordinate encoder), any other token through the
        embedding table; ``eos`` and ``size`` logits are suppressed (moondream.py:397-398).
        Returns (tokens, grounding coordinates, pos)."""
        tk = self.cfg.tokenizer
        logits, h, pos = self.prefill_prompt(prompt_ids, pos0, kv)
        last = h[-1:].clone()
        tok = int(torch.argmax(logits.float()))
        out, coords = [], []
        while tok != tk.answer_id and len(out) < max_tokens:
            out.append(tok)
            if tok == tk.coord_id:
                cl = self.decode_coordinate(last)
                c = torch.argmax(cl.float(), dim=-1) / cl.size(-1)
                coords.append(c.item())
                emb = self.encode_coordinate(c.to(BF16))
            else:
                emb = self.embed([tok])
            logits, h = self.decode_token(emb.reshape(1, -1), pos, kv)
            last = h[-1:].clone()
            logits[tk.eos_id] = float("-inf")
            logits[tk.size_id] = float("-inf")
            pos += 1
            tok = int(torch.argmax(logits.float()))
        return out, coords, pos
