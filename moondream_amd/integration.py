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
