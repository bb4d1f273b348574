"""LoRA "variants" (reference: moondream/torch/lora.py).

A variant is a checkpoint of low-rank pairs for the decoder's four linears per block.  The
reference downloads it from api.moondream.ai into the Hugging Face cache and renames / nests its
keys (lora.py:12-79).  This build has no network path: a variant is either REGISTERED in process
(``MoondreamModel.register_variant(id, nested_or_flat_dict)``) or read from the same cache file the
reference would have written, ``<cache>/md_variants/<id>/final.pt`` -- if it is not there, the call
fails instead of downloading.

Nested layout consumed by the model (what the reference's ``variant_state_dict`` returns):
``lora["text"]["blocks"][str(i)]["attn"]["qkv" | "proj"]["A" | "B"]`` and
``...["mlp"]["fc1" | "fc2"]["A" | "B"]`` with A [r, in] and B [out, r]
(text.py:31-32,55-56; layers.py:129-146).
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Dict, Optional

import torch

RENAME_RULES = [  # reference: lora.py:63-69
    ("text_model.transformer.h", "text.blocks"),
    (".mixer", ".attn"),
    (".out_proj", ".proj"),
    (".Wqkv", ".qkv"),
    (".parametrizations.weight.0", ""),
]


def variant_cache_dir() -> Path:
    """Where variant files live on disk -- an on-disk contract shared with the reference (lora.py:12-21):
    ``$HF_HUB_CACHE/md_variants``, else ``$HF_HOME/hub/md_variants``, else ``~/.cache/huggingface/hub/md_variants``."""
    for env, sub in (("HF_HUB_CACHE", ()), ("HF_HOME", ("hub",))):
        root = os.environ.get(env)
        if root is not None:
            return Path(root, *sub, "md_variants")
    return Path.home() / ".cache" / "huggingface" / "hub" / "md_variants"


def cached_variant_path(variant_id: str) -> Path:
    """reference: lora.py:24-41 without the download: the file must already be in the cache."""
    dest = variant_cache_dir() / variant_id / "final.pt"
    if not dest.exists():
        raise FileNotFoundError(
            f"LoRA variant '{variant_id}' is not registered and not in the cache ({dest}); this build does not download "
            "variants -- place the file there or call MoondreamModel.register_variant()"
        )
    return dest


def nest(flat: Dict[str, torch.Tensor]) -> dict:
    """Dotted keys -> nested dicts (``"a.b.c": t`` -> ``{"a": {"b": {"c": t}}}``), the layout the ``lora`` argument of
    ``text_decoder`` is indexed with (reference: lora.py:43-51, consumed at text.py:31-32,55-56)."""
    tree: dict = {}
    for dotted, tensor in flat.items():
        *path, leaf = dotted.split(".")
        node = tree
        for name in path:
            nxt = node.get(name)
            if nxt is None:
                nxt = node[name] = {}
            elif not isinstance(nxt, dict):
                raise ValueError(f"variant key {dotted!r} nests under a tensor at {name!r}")
            node = nxt
        node[leaf] = tensor
    return tree


def rename_and_nest(state_dict: Dict[str, torch.Tensor]) -> dict:
    """reference: lora.py:61-79."""
    out = {}
    for key, tensor in state_dict.items():
        for old, new in RENAME_RULES:
            if old in key:
                key = key.replace(old, new)
        out[key] = tensor
    return nest(out)


def variant_state_dict(variant_id: Optional[str], device="cpu") -> Optional[dict]:
    """reference: lora.py:54-79."""
    if variant_id is None:
        return None
    sd = torch.load(cached_variant_path(variant_id), map_location=device, weights_only=True)
    return rename_and_nest(sd)
