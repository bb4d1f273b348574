"""Reference-side glue for a maintainer who binds the reference ``MoondreamModel`` to
libmoondream_hip.so.

The only stateful part of the boundary is the KV cache.  The reference owns 2 x n_layers
separately allocated module buffers, ``block.kv_cache.k_cache / v_cache`` of shape
[1, n_kv_heads, max_context, head_dim] (reference: moondream.py:62-72, created per block at
moondream.py:152-162), and every user of them works IN PLACE on the attribute:
``KVCache.update`` (index_put, moondream.py:74-78), the snapshot of ``encode_image``
(moondream.py:259-268) and the copy-back of ``load_encoded_image`` (moondream.py:620-623).
``md_kv_cache`` wants ONE slab per tensor kind with a uniform layer stride.  Rebinding each
block's buffers to views of such a slab satisfies both sides: the reference code keeps working
unchanged (tests/test_integration_cpu.py runs it that way and reproduces the golden token ids),
and the library sees ``[L][B=1][H][ctx][hd]``.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import _lib


def rebind_kv_caches_to_slab(ref_model, batch: int = 1) -> Tuple[torch.Tensor, torch.Tensor, _lib.MdKvCache]:
    """Replace every ``block.kv_cache.{k,v}_cache`` of a reference-style model by a view of one
    ``[L][batch][H][ctx][hd]`` slab (slot 0 of each layer is what the reference code sees; slots
    1.. are there for the batched engine) and return (slab_k, slab_v, md_kv_cache)."""
    blocks = ref_model.text.blocks
    k0 = blocks[0].kv_cache.k_cache
    one, heads, ctx, hd = k0.shape
    assert one == 1
    n_layers = len(blocks)
    slab_k = torch.zeros(n_layers, batch, heads, ctx, hd, dtype=k0.dtype, device=k0.device)
    slab_v = torch.zeros_like(slab_k)
    for l, blk in enumerate(blocks):
        slab_k[l, 0:1].copy_(blk.kv_cache.k_cache)  # keep whatever was cached so far
        slab_v[l, 0:1].copy_(blk.kv_cache.v_cache)
        blk.kv_cache.k_cache = slab_k[l, 0:1]       # registered buffer names: Module.__setattr__ swaps the buffer
        blk.kv_cache.v_cache = slab_v[l, 0:1]
    batch_stride = heads * ctx * hd
    kv = _lib.MdKvCache(slab_k.data_ptr(), slab_v.data_ptr(), batch * batch_stride, batch_stride, ctx)
    return slab_k, slab_v, kv


# ------------------------------------------------------------------ the mask argument of the seam
#
# The reference hands ``_prefill`` / ``_decode_one_tok`` a bool mask instead of a rule.  Three masks reach the seam
# (reference file:line):
#   * ``self.attn_mask[:, :, pos : pos + T, :]`` (moondream.py:304-309; buffer built at moondream.py:138-146): the
#     prefix-LM mask, rows pos..pos+T-1 -- key j visible to the query at position p iff j <= p or (p < P and j < P),
#     P = 1 + (crop_size / patch_size)^2 = 730;
#   * the same slice of a plain ``tril`` (moondream.py:571-575: the text-only query): key j visible iff j <= p;
#   * the decode row ``mask[:, :, :pos + 1] = 1`` of the generator loops (moondream.py:472-474,527,697-699): j <= p.
# The library takes the RULE (``md_text_model.prefix_len``: P, or 0 for the plain causal mask) and reads keys [0, p]
# only.  For rows with p >= P the two rules select the same keys, so such a slice is valid under either; a slice
# that is neither is not something the kernels can honour and is rejected.

MASK_PREFIX_LM = "prefix_lm"   # -> md_text_model.prefix_len = P
MASK_CAUSAL = "causal"         # -> md_text_model.prefix_len = 0
MASK_EITHER = "either"         # every row has p >= P (or the mask is None): both rules select keys [0, p]


def consecutive_positions(pos_ids: torch.Tensor) -> int:
    """``pos_ids`` of the seam (int64 [T]: moondream.py:256,309,473) -> first position.  The library addresses the KV
    slab as pos0 .. pos0 + T - 1; anything else (gaps, repeats, descending) is rejected."""
    p = pos_ids.reshape(-1).to("cpu", torch.int64)
    if p.numel() == 0:
        raise ValueError("pos_ids is empty")
    p0 = int(p[0])
    if not torch.equal(p, torch.arange(p0, p0 + p.numel(), dtype=torch.int64)):
        raise ValueError(f"pos_ids must be consecutive ascending positions (got {p.tolist()[:8]}...): "
                         "the KV slab rows of a pass are pos0 .. pos0 + T - 1")
    if p0 < 0:
        raise ValueError(f"negative position {p0}")
    return p0


def classify_attn_mask(attn_mask, pos_ids: torch.Tensor, prefix_len: int, max_context: int) -> str:
    """Which rule the mask slice the reference passes through the seam encodes.  ``attn_mask``: None, bool
    [1, 1, T, ctx] (prefill) or [1, 1, ctx] (the decode row); ``pos_ids``: int64 [T] consecutive.  Returns
    MASK_PREFIX_LM, MASK_CAUSAL or MASK_EITHER; raises ValueError for any other mask (the kernels evaluate a rule and
    cannot apply an arbitrary mask) and for non-consecutive positions."""
    p0 = consecutive_positions(pos_ids)
    t = int(pos_ids.numel())
    if p0 + t > max_context:
        raise ValueError(f"positions [{p0}, {p0 + t}) do not fit the {max_context}-slot context")
    if attn_mask is None:
        return MASK_EITHER
    m = attn_mask
    if m.dtype != torch.bool:
        raise ValueError(f"attn_mask must be a bool tensor (got {m.dtype})")
    if m.dim() == 3 and t == 1:          # the decode row [1, 1, ctx]
        m = m.reshape(1, -1)
    elif m.dim() == 4:
        m = m.reshape(-1, m.shape[-1])   # [1, 1, T, ctx]
    else:
        raise ValueError(f"attn_mask of shape {tuple(attn_mask.shape)} is neither [1, 1, T, ctx] nor [1, 1, ctx]")
    if m.shape != (t, max_context):
        raise ValueError(f"attn_mask selects {tuple(m.shape)}, expected ({t}, {max_context}) for {t} position(s)")
    dev = m.device
    p = torch.arange(p0, p0 + t, device=dev).unsqueeze(1)
    j = torch.arange(max_context, device=dev).unsqueeze(0)
    causal = j <= p
    prefix = causal | ((p < prefix_len) & (j < prefix_len))
    is_causal, is_prefix = bool(torch.equal(m, causal)), bool(torch.equal(m, prefix))
    if is_causal and is_prefix:
        return MASK_EITHER
    if is_prefix:
        return MASK_PREFIX_LM
    if is_causal:
        return MASK_CAUSAL
    raise ValueError(
        "attn_mask is neither the prefix-LM mask (reference moondream.py:138-146) nor a causal mask "
        "(moondream.py:571-575) for these positions; the kernels evaluate the rule "
        "`j <= p or (p < prefix_len and j < prefix_len)` and cannot apply an arbitrary mask")
