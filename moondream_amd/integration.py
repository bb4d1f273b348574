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

import ctypes as C
from typing import Optional, Tuple

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
    # ordinary (non-inference) tensors even when called from inside torch.inference_mode() (the seam is): the reference's
    # load_encoded_image (moondream.py:620-623) updates them in place from user code outside of it
    with torch.inference_mode(False):
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


def _classify_decode_row(mask: torch.Tensor, pos_ids: torch.Tensor, prefix_len: int, max_context: int) -> str:
    """The per-token case of ``classify_attn_mask`` -- the decode row [1, 1, ctx] of the generator loops
    (moondream.py:472-474,515,697-699) -- with ONE device-to-host transfer of three integers instead of a [1, ctx]
    comparison against both rules (round 5 stalled the host three times per token here).  A set of S visible keys whose
    highest index is S - 1 is exactly {0 .. S - 1}: the causal row of position p is S == p + 1, the prefix-LM row of a
    position p < prefix_len is S == prefix_len."""
    m = mask.reshape(-1)
    idx = torch.arange(1, max_context + 1, device=m.device)
    stats = torch.stack([pos_ids.reshape(-1)[0].to(m.device, torch.int64), m.sum(), (idx * m).max()])
    p, count, top = (int(v) for v in stats.tolist())
    if p < 0 or p >= max_context:
        raise ValueError(f"position {p} does not fit the {max_context}-slot context")
    if count == top:  # keys {0 .. count - 1}
        causal, prefix = count == p + 1, p < prefix_len and count == prefix_len
        if causal and (prefix or p >= prefix_len):
            return MASK_EITHER
        if causal:
            return MASK_CAUSAL
        if prefix:
            return MASK_PREFIX_LM
    raise ValueError(
        f"the decode row at position {p} exposes {count} key(s) up to slot {top - 1}: neither `mask[:, :, :pos + 1] = 1` "
        "(reference moondream.py:472-474,515) nor a row of the prefix-LM buffer (moondream.py:138-146)")


def classify_attn_mask(attn_mask, pos_ids: torch.Tensor, prefix_len: int, max_context: int) -> str:
    """Which rule the mask slice the reference passes through the seam encodes.  ``attn_mask``: None, bool
    [1, 1, T, ctx] (prefill) or [1, 1, ctx] (the decode row); ``pos_ids``: int64 [T] consecutive.  Returns
    MASK_PREFIX_LM, MASK_CAUSAL or MASK_EITHER; raises ValueError for any other mask (the kernels evaluate a rule and
    cannot apply an arbitrary mask) and for non-consecutive positions."""
    t = int(pos_ids.numel())
    if t == 1 and attn_mask is not None and attn_mask.dim() == 3 and attn_mask.dtype == torch.bool \
            and attn_mask.numel() == max_context:
        return _classify_decode_row(attn_mask, pos_ids, prefix_len, max_context)
    p0 = consecutive_positions(pos_ids)
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


# ------------------------------------------------------------------ the drop-in itself
#
# ``bind_reference(model)`` is INTEGRATION.md section 2 as code: it rebinds the four seam attributes of an UNMODIFIED
# reference ``MoondreamModel`` instance (the same four its own ``compile()`` rebinds, moondream.py:194-204) to ctypes
# calls into libmoondream_hip.so.  Everything above the seam -- ``encode_image``, ``caption``, ``query``, ``detect``,
# ``point``, ``_prefill_prompt``, ``_generate_answer``, ``_generate_points``, the tokenizer, the streaming detokeniser --
# stays the reference's own code; what it still computes with ATen on its own parameters is what it computes outside the
# seam (``text_encoder``, the prompt pass's ``lm_head``, the region MLPs, ``reconstruct_from_crops``).
# tests/test_dropin_gpu.py runs the reference's public calls this way and compares with the goldens the same class
# produced on its own.


class ReferenceBinding:
    """What ``bind_reference`` attached to a reference model: the packed weights, the KV slab the reference's
    ``KVCache`` buffers are now views of, the caller-owned workspace, and the original seam methods (``unbind``)."""

    def __init__(self, ref_model, lib, packed, config):
        self.model, self.lib, self.packed, self.config = ref_model, lib, packed, config
        self.slab_k = self.slab_v = self.kv = None
        self.arena: Optional[torch.Tensor] = None
        self.loras: dict = {}
        self.original: dict = {}
        self.calls = {"_vis_enc": 0, "_vis_proj": 0, "_prefill": 0, "_decode_one_tok": 0}

    def workspace(self, nbytes: int) -> torch.Tensor:
        if self.arena is None or self.arena.numel() < nbytes:
            with torch.inference_mode(False):
                self.arena = torch.empty(int(nbytes) + 4096, dtype=torch.uint8, device=self.packed.device)
        return self.arena

    def stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.packed.device).cuda_stream)

    def ensure_slab(self):
        """The reference REPLACES its KVCache modules in ``_setup_caches`` (moondream.py:152-162), which the text-only
        ``query`` calls on every use (moondream.py:568): when the blocks' buffers are no longer views of the slab,
        rebind (the fresh buffers' contents -- zeros -- are copied in, as the reference would see them)."""
        blocks = self.model.text.blocks
        if (self.slab_k is None
                or blocks[0].kv_cache.k_cache.data_ptr() != self.slab_k[0].data_ptr()
                or blocks[-1].kv_cache.v_cache.data_ptr() != self.slab_v[len(blocks) - 1].data_ptr()):
            self.slab_k, self.slab_v, self.kv = rebind_kv_caches_to_slab(self.model, batch=1)

    def lora(self, lora):
        """The reference hands the seam the nested dict of ``variant_state_dict`` (lora.py:54-79); packed once per dict."""
        if lora is None:
            return None
        from .weights import PackedLora

        key = id(lora)
        if key not in self.loras:
            if len(self.loras) >= 5:   # (the reference caches 5 variants, lora.py:54) a loader that hands out fresh dicts must not grow this without bound
                self.loras.pop(next(iter(self.loras)))
            self.loras[key] = (lora, PackedLora(self.config, lora, self.packed.device))  # keeps the dict alive: id() stays unique
        return self.loras[key][1]

    def unbind(self):
        for name, fn in self.original.items():
            if fn is None:
                self.model.__dict__.pop(name, None)   # back to the class's method
            else:
                setattr(self.model, name, fn)
        self.model.__dict__.pop("_mi355x", None)


def bind_reference(ref_model) -> ReferenceBinding:
    """Rebind ``_vis_enc`` / ``_vis_proj`` / ``_prefill`` / ``_decode_one_tok`` (reference moondream.py:168-192) of an
    unmodified reference ``MoondreamModel`` that lives on a GPU to the gfx950 library.  No fallback: raises when the
    library is missing or the model is on the CPU."""
    from .config import MoondreamConfig
    from .weights import PackedModel

    lib = _lib.load()
    dev = ref_model.device
    if dev.type != "cuda":
        raise _lib.MoondreamHipError("bind_reference needs the reference model on a GPU (model.to('cuda')); there is no CPU path")
    cfg = MoondreamConfig.from_dict(ref_model.config.to_dict())
    sd = {k: v for k, v in ref_model.state_dict().items() if ".kv_cache." not in k}
    b = ReferenceBinding(ref_model, lib, PackedModel(cfg, sd, dev), cfg)   # one-time weight packing
    vit, text = b.packed.vit, b.packed.text
    text_causal = type(text).from_buffer_copy(text)                        # the same weights under the plain causal rule
    text_causal.prefix_len = 0
    v, t = cfg.vision, cfg.text
    b.ensure_slab()

    def _vis_enc(x):                                                       # moondream.py:168-169 -> vision.py:64-74
        b.calls["_vis_enc"] += 1
        n = int(x.shape[0])
        if x.dtype != torch.bfloat16 or tuple(x.shape[1:]) != (3, v.crop_size, v.crop_size):
            raise ValueError(f"_vis_enc expects bf16 [N, 3, {v.crop_size}, {v.crop_size}] crops (got {x.dtype} {tuple(x.shape)})")
        x = x.contiguous()
        out = torch.empty(n, v.n_patches, v.enc_dim, dtype=x.dtype, device=dev)
        w = b.workspace(lib.md_vit_workspace_bytes(C.byref(vit), n))
        _lib.check(lib.md_vit_encode(C.byref(vit), x.data_ptr(), _lib.MD_CROPS_BF16_CHW, n, out.data_ptr(),
                                     w.data_ptr(), w.numel(), b.stream()), "md_vit_encode")
        return out

    def _vis_proj(g, r):                                                   # moondream.py:171-172 -> vision.py:77-89
        b.calls["_vis_proj"] += 1
        g, r = g.contiguous(), r.contiguous()
        out = torch.empty(v.n_patches, v.proj_out_dim, dtype=g.dtype, device=dev)
        w = b.workspace(lib.md_vision_project_workspace_bytes(C.byref(vit), 1))
        _lib.check(lib.md_vision_project_grid(C.byref(vit), g.data_ptr(), r.data_ptr(), r.shape[0], r.shape[1],
                                              out.data_ptr(), out.shape[1], w.data_ptr(), w.numel(), b.stream()),
                   "md_vision_project_grid")
        return out

    def text_forward(x, attn_mask, pos_ids, lora):
        if x.dim() != 3 or x.shape[0] != 1 or int(pos_ids.numel()) != x.shape[1]:
            raise ValueError(f"x must be [1, T, D] with T position ids (got {tuple(x.shape)}, {int(pos_ids.numel())})")
        rows = int(x.shape[1])
        kind = classify_attn_mask(attn_mask, pos_ids, t.prefix_attn, t.max_context)   # ValueError for any other mask
        pos0 = pos_ids.reshape(-1)[:1].to(dev, torch.int32)                # consecutive: checked by the classifier
        if rows > 1 or kind == MASK_PREFIX_LM:
            p0 = int(pos0)                                                 # prefill passes only: pos_ids is host data there
            if kind != MASK_CAUSAL and p0 < t.prefix_attn and p0 + rows < t.prefix_attn:
                raise ValueError(f"a prefix-LM pass must reach the end of the {t.prefix_attn}-position bidirectional prefix")
        model = text_causal if kind == MASK_CAUSAL else text
        b.ensure_slab()
        x = x.contiguous()
        hidden = torch.empty_like(x)
        packed_lora = b.lora(lora)
        if packed_lora is not None:                                        # text.py:31-32,55-56; layers.py:131-142
            w = b.workspace(lib.md_text_lora_workspace_bytes(C.byref(model), 1, rows))
            _lib.check(lib.md_text_forward_lora(C.byref(model), packed_lora.ptr(), x.data_ptr(), hidden.data_ptr(), 1, rows,
                                                pos0.data_ptr(), C.byref(b.kv), w.data_ptr(), w.numel(), b.stream()),
                       "md_text_forward_lora")
            return hidden
        w = b.workspace(lib.md_text_workspace_bytes(C.byref(model), 1, rows))
        _lib.check(lib.md_text_forward(C.byref(model), x.data_ptr(), hidden.data_ptr(), 1, rows, pos0.data_ptr(),
                                       C.byref(b.kv), w.data_ptr(), w.numel(), b.stream()), "md_text_forward")
        return hidden

    def _prefill(x, attn_mask, pos_ids, lora):                             # moondream.py:174-181 -> text.py:128-160
        b.calls["_prefill"] += 1
        return text_forward(x, attn_mask, pos_ids, lora)

    def _decode_one_tok(x, attn_mask, pos_ids, lora):                      # moondream.py:183-192
        b.calls["_decode_one_tok"] += 1
        hidden = text_forward(x, attn_mask, pos_ids, lora)
        logits = torch.empty(1, t.vocab_size, dtype=x.dtype, device=dev)
        w = b.workspace(lib.md_lm_head_workspace_bytes(C.byref(text), 1))
        _lib.check(lib.md_lm_head(C.byref(text), hidden.data_ptr(), 1, 1, logits.data_ptr(), logits.shape[1],
                                  w.data_ptr(), w.numel(), b.stream()), "md_lm_head")
        return logits, hidden

    for name, fn in (("_vis_enc", _vis_enc), ("_vis_proj", _vis_proj), ("_prefill", _prefill), ("_decode_one_tok", _decode_one_tok)):
        b.original[name] = ref_model.__dict__.get(name)   # None: the class's method (a compile()d model has instance attributes)
        setattr(ref_model, name, fn)
    ref_model.__dict__["_mi355x"] = b
    return b
