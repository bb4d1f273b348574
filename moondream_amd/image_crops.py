"""Host-side image tiling (integer / uint8 work in front of the ViT).

Same contract as the reference's ``image_crops`` module (reference:
moondream/torch/image_crops.py:17-231): ``select_tiling``,
``overlap_crop_image`` (global 378x378 crop + overlapping local crops cut from
the image resized to the tiling) and ``reconstruct_from_crops``.

The resize has the reference's TWO branches, chosen as the reference chooses
(image_crops.py:7-14): pyvips when it imports (image_crops.py:124-136:
``Image.new_from_array(img).resize(scale_x, vscale=scale_y)`` with the scales
taken from the image's own size), PIL LANCZOS otherwise (image_crops.py:137-150).
The two give different pixels ("image resize parity"), so a
deployment that has pyvips must use it here as well to match the reference
there.  pyvips is not in this build's image: the PIL branch is the one pinned
by crop CRCs recorded from the reference; the pyvips branch is pinned on its
LOGIC (scales, call order, which image feeds the global crop) by running the
reference's function and this one against the same stand-in module
(tests/test_image_crops.py), not on libvips' pixels.  ``MOONDREAM_RESIZE=pil``
forces the PIL branch where pyvips exists.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple, TypedDict, Union

import os

import numpy as np
import torch
from PIL import Image


def _import_pyvips():
    """The reference's probe (image_crops.py:7-14): pyvips if it imports, anything going wrong = the PIL branch."""
    if os.environ.get("MOONDREAM_RESIZE", "").lower() == "pil":
        return None
    try:
        import pyvips

        return pyvips
    except Exception:
        return None


_pyvips = _import_pyvips()


def resize_backend() -> str:
    """"pyvips" | "pil": the branch ``overlap_crop_image`` takes (decided at import, like the reference's HAS_VIPS)."""
    return "pyvips" if _pyvips is not None else "pil"


class OverlapCropOutput(TypedDict):
    crops: np.ndarray
    tiling: Tuple[int, int]


def select_tiling(height: int, width: int, crop_size: int, max_crops: int) -> Tuple[int, int]:
    """Number of local crops along (height, width).  reference: image_crops.py:17-50."""
    if height <= crop_size or width <= crop_size:
        return (1, 1)
    lo_h = math.ceil(height / crop_size)
    lo_w = math.ceil(width / crop_size)
    if lo_h * lo_w > max_crops:
        # too many even at the minimum: shrink both proportionally
        shrink = math.sqrt(max_crops / (lo_h * lo_w))
        return (max(1, math.floor(lo_h * shrink)), max(1, math.floor(lo_w * shrink)))
    # aspect-ratio-matched split of the crop budget, never below the minimum
    n_h = max(math.floor(math.sqrt(max_crops * height / width)), lo_h)
    n_w = max(math.floor(math.sqrt(max_crops * width / height)), lo_w)
    if n_h * n_w > max_crops:
        if n_w > n_h:
            n_w = math.floor(max_crops / n_h)
        else:
            n_h = math.floor(max_crops / n_w)
    return (max(1, n_h), max(1, n_w))


def _resize(img: np.ndarray, height: int, width: int) -> np.ndarray:
    if img.shape[0] == int(height) and img.shape[1] == int(width):
        # PIL's Image.resize returns self.copy() when the size does not change, whatever the filter: the same pixels.  Skipping the
        # numpy -> PIL -> numpy round trip matters: it is GIL-bound Python / C (frombuffer, copy, tobytes + join: ~0.5 ms per call),
        # so the "thread pool" tiling of a batch of crop-sized images ran serially (64 images: 26 ms on the GPU host, two calls each)
        return img
    pil = Image.fromarray(img)
    return np.asarray(pil.resize((int(width), int(height)), resample=Image.Resampling.LANCZOS))


def overlap_crop_image(
    image: np.ndarray,
    overlap_margin: int,
    max_crops: int,
    base_size: Tuple[int, int] = (378, 378),
    patch_size: int = 14,
    out: "np.ndarray | None" = None,
) -> OverlapCropOutput:
    """crops[0] = whole image resized to base_size; crops[1:] = base_size windows
    at stride (base - 2*margin*patch) over the image resized to the tiling.
    reference: image_crops.py:58-167.  ``out``: a caller-owned uint8 array of the right shape
    (``crop_count`` x base x base x C, e.g. a slice of a pinned staging buffer) to cut the crops into."""
    src_h, src_w = image.shape[:2]
    margin_px = patch_size * overlap_margin
    window = (base_size[0] // patch_size - 2 * overlap_margin) * patch_size  # stride between crops

    tiling = select_tiling(src_h - 2 * margin_px, src_w - 2 * margin_px, window, max_crops)
    th, tw = tiling
    shape = (th * tw + 1, base_size[0], base_size[1], image.shape[2])
    if out is None:
        crops = np.zeros(shape, dtype=np.uint8)
    else:
        assert out.shape == shape and out.dtype == np.uint8, (out.shape, shape)
        crops = out

    if _pyvips is not None:
        # reference: image_crops.py:124-136 -- scale factors from the array's own size, horizontal scale first, the GLOBAL crop
        # resized from the original vips image (not from the tiled resize); whatever size libvips rounds to is taken as it comes
        vimg = _pyvips.Image.new_from_array(image)
        resized = vimg.resize((tw * window + 2 * margin_px) / image.shape[1], vscale=(th * window + 2 * margin_px) / image.shape[0]).numpy()
        crops[0] = vimg.resize(base_size[1] / vimg.width, vscale=base_size[0] / vimg.height).numpy()
    else:
        resized = _resize(image, th * window + 2 * margin_px, tw * window + 2 * margin_px)
        crops[0] = _resize(image, base_size[0], base_size[1])
    for ty in range(th):
        for tx in range(tw):
            y0, x0 = ty * window, tx * window
            piece = resized[y0 : min(y0 + base_size[0], resized.shape[0]), x0 : min(x0 + base_size[1], resized.shape[1])]
            if out is not None and piece.shape[:2] != tuple(base_size):
                crops[1 + ty * tw + tx] = 0  # (cannot happen for the reference's geometry: the resized image is th*window + 2*margin)
            crops[1 + ty * tw + tx, : piece.shape[0], : piece.shape[1]] = piece
    return {"crops": crops, "tiling": tiling}


def crop_count(height: int, width: int, overlap_margin: int, max_crops: int, base_size: Tuple[int, int] = (378, 378),
               patch_size: int = 14) -> Tuple[int, Tuple[int, int]]:
    """(number of crops incl. the global one, tiling) ``overlap_crop_image`` will produce for an image of this size."""
    margin_px = patch_size * overlap_margin
    window = (base_size[0] // patch_size - 2 * overlap_margin) * patch_size
    th, tw = select_tiling(height - 2 * margin_px, width - 2 * margin_px, window, max_crops)
    return th * tw + 1, (th, tw)


def reconstruct_from_crops(
    crops: Union[torch.Tensor, Sequence[torch.Tensor]],
    tiling: Tuple[int, int],
    overlap_margin: int,
    patch_size: int = 14,
) -> torch.Tensor:
    """Stitch per-crop (H, W, C) grids back into one grid: every crop contributes
    its interior, plus its outer margin where it touches the image border.
    reference: image_crops.py:170-231."""
    th, tw = tiling
    ch, cw = crops[0].shape[:2]
    m = overlap_margin * patch_size
    step_h, step_w = ch - 2 * m, cw - 2 * m
    out = torch.zeros(
        (step_h * th + 2 * m, step_w * tw + 2 * m, crops[0].shape[2]), device=crops[0].device, dtype=crops[0].dtype
    )
    for idx, crop in enumerate(crops):
        ty, tx = divmod(idx, tw)
        ys = 0 if ty == 0 else m
        ye = ch if ty == th - 1 else ch - m
        xs = 0 if tx == 0 else m
        xe = cw if tx == tw - 1 else cw - m
        out[ty * step_h + ys : ty * step_h + ye, tx * step_w + xs : tx * step_w + xe] = crop[ys:ye, xs:xe]
    return out
