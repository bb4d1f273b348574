"""Host tiling: (1) the reference's own three tests (reference:
tests/test_image_crops.py:6-57) run against this package's implementation,
(2) bit-exact crops and tilings against goldens recorded from the reference."""
import os
import zlib

import numpy as np
import torch

from moondream_amd import synth
from moondream_amd.image_crops import overlap_crop_image, reconstruct_from_crops, select_tiling


def test_overlap_crop_basic():
    img = np.zeros((800, 600, 3), dtype=np.uint8)
    img[300:500, 200:400] = 255
    r = overlap_crop_image(img, overlap_margin=4, max_crops=12)
    assert r["crops"][0].shape == (378, 378, 3)
    assert len(r["crops"]) > 1
    assert all(c.shape == (378, 378, 3) for c in r["crops"])
    assert len(r["tiling"]) == 2


def test_overlap_crop_small_image():
    r = overlap_crop_image(np.zeros((300, 200, 3), dtype=np.uint8), overlap_margin=4, max_crops=12)
    assert r["crops"][0].shape == (378, 378, 3)
    assert len(r["crops"]) == 2
    assert r["tiling"] == (1, 1)


def test_reconstruction():
    img = np.zeros((800, 600, 3), dtype=np.uint8)
    img[300:500, 200:400] = 255
    r = overlap_crop_image(img, overlap_margin=4, max_crops=12)
    rec = reconstruct_from_crops([torch.from_numpy(c) for c in r["crops"][1:]], r["tiling"], overlap_margin=4).numpy()
    cy, cx = rec.shape[0] // 2, rec.shape[1] // 2
    assert rec[cy - 100 : cy + 100, cx - 100 : cx + 100].mean() > rec[:100, :100].mean() + 100


def test_tiling_table_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, "image_crops.npz"))
    for h, w, th, tw in g["tiling_table"]:
        assert select_tiling(int(h), int(w), 266, 12) == (int(th), int(tw)), (h, w)


def test_crops_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, "image_crops.npz"))
    for i, (h, w, th, tw, n, crc) in enumerate(g["crop_cases"]):
        img = synth.synthetic_image_array(i, 11, (int(h), int(w)))
        r = overlap_crop_image(img, overlap_margin=4, max_crops=12)
        assert tuple(r["tiling"]) == (int(th), int(tw))
        assert len(r["crops"]) == int(n)
        assert zlib.crc32(r["crops"].tobytes()) == int(crc), (h, w)


def test_identical_crops_at_378():
    """At <= 378 px the global and the single local crop are the same pixels
    (SURVEY.md section 7); both are still encoded."""
    r = overlap_crop_image(synth.synthetic_image_array(0, 0), overlap_margin=4, max_crops=12)
    assert r["tiling"] == (1, 1) and np.array_equal(r["crops"][0], r["crops"][1])


def test_random_sizes_against_the_reference_itself():
    """Build container only (needs /root/reference): for random image sizes -- portrait, landscape, tiny, larger than the
    12-tile budget -- this package's select_tiling / overlap_crop_image / reconstruct_from_crops give the SAME tilings,
    the SAME crop bytes and the SAME stitched grid as the reference's functions (image_crops.py:17-231, PIL branch)."""
    import importlib.util
    import pytest

    ref_root = os.environ.get("MOONDREAM_REFERENCE", "/root/reference")
    path = os.path.join(ref_root, "moondream", "torch", "image_crops.py")
    if not os.path.isfile(path):
        pytest.skip("needs the reference checkout (build container)")
    spec = importlib.util.spec_from_file_location("ref_image_crops", path)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    rng = np.random.default_rng(7)
    sizes = [(378, 378), (1, 1), (2, 3000), (3000, 2), (379, 378), (1200, 1600), (4000, 3000), (100, 1000)]
    sizes += [(int(rng.integers(16, 2200)), int(rng.integers(16, 2200))) for _ in range(12)]
    for h, w in sizes:
        assert select_tiling(h, w, 378, 12) == tuple(ref.select_tiling(h, w, 378, 12)), (h, w)
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        a = overlap_crop_image(img, overlap_margin=4, max_crops=12)
        b = ref.overlap_crop_image(img, overlap_margin=4, max_crops=12)
        assert tuple(a["tiling"]) == tuple(b["tiling"]), (h, w)
        assert np.array_equal(np.asarray(a["crops"]), np.asarray(b["crops"])), (h, w)
        th, tw = a["tiling"]
        feats = [torch.from_numpy(rng.standard_normal((27, 27, 8)).astype(np.float32)) for _ in range(th * tw)]
        ra = reconstruct_from_crops(feats, (th, tw), overlap_margin=4, patch_size=1)
        rb = ref.reconstruct_from_crops(feats, (th, tw), overlap_margin=4, patch_size=1)
        assert torch.equal(ra, rb), (h, w)


def test_pyvips_branch_follows_the_reference_with_a_stand_in_module(monkeypatch):
    """Build container only.  pyvips is not installed here, so libvips' pixels cannot be pinned; what CAN be pinned is the
    branch's logic (reference image_crops.py:124-136): which scales are computed from which sizes, horizontal first, and that
    the global crop comes from the ORIGINAL image.  A stand-in ``pyvips`` (new_from_array / resize(scale, vscale=) / numpy /
    width / height, resampling through PIL with the rounding libvips documents) is put in sys.modules; the reference's module
    and this package's are both (re)imported under it and must produce the same tilings and the same crop bytes -- and, the
    stand-in being a different resampler from the PIL branch's direct call, different bytes from the PIL branch on a
    non-identity size (so the test cannot pass by both sides ignoring the module)."""
    import importlib
    import importlib.util
    import sys
    import types
    import pytest
    from PIL import Image as PILImage

    ref_root = os.environ.get("MOONDREAM_REFERENCE", "/root/reference")
    path = os.path.join(ref_root, "moondream", "torch", "image_crops.py")
    if not os.path.isfile(path):
        pytest.skip("needs the reference checkout (build container)")

    calls = []

    class FakeVipsImage:
        def __init__(self, arr):
            self.arr = np.ascontiguousarray(arr)
            self.height, self.width = arr.shape[:2]

        @staticmethod
        def new_from_array(arr):
            return FakeVipsImage(np.asarray(arr))

        def resize(self, scale, vscale=None):
            vs = scale if vscale is None else vscale
            w, h = max(1, int(round(self.width * scale))), max(1, int(round(self.height * vs)))
            calls.append((self.width, self.height, round(scale, 9), round(vs, 9)))
            # BILINEAR on purpose: a different resampler from the PIL branch's LANCZOS
            return FakeVipsImage(np.asarray(PILImage.fromarray(self.arr).resize((w, h), resample=PILImage.Resampling.BILINEAR)))

        def numpy(self):
            return self.arr

    fake = types.ModuleType("pyvips")
    fake.Image = FakeVipsImage
    monkeypatch.setitem(sys.modules, "pyvips", fake)
    monkeypatch.delenv("MOONDREAM_RESIZE", raising=False)
    spec = importlib.util.spec_from_file_location("ref_image_crops_vips", path)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    assert ref.HAS_VIPS
    import moondream_amd.image_crops as ours_mod
    ours = importlib.reload(ours_mod)
    try:
        assert ours.resize_backend() == "pyvips"
        rng = np.random.default_rng(5)
        for h, w in [(378, 378), (768, 1024), (500, 1300), (1200, 640), (97, 2000), (379, 378)]:
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            calls.clear()
            b = ref.overlap_crop_image(img, overlap_margin=4, max_crops=12)
            ref_calls = list(calls)
            calls.clear()
            a = ours.overlap_crop_image(img, overlap_margin=4, max_crops=12)
            assert calls == ref_calls and len(calls) == 2, (h, w, calls, ref_calls)  # same two resize calls, same order, same scales
            assert tuple(a["tiling"]) == tuple(b["tiling"]) and np.array_equal(a["crops"], b["crops"]), (h, w)
        monkeypatch.setenv("MOONDREAM_RESIZE", "pil")
        pil_mod = importlib.reload(ours_mod)
        assert pil_mod.resize_backend() == "pil"
        assert not np.array_equal(pil_mod.overlap_crop_image(img, overlap_margin=4, max_crops=12)["crops"], a["crops"])
    finally:
        monkeypatch.undo()
        importlib.reload(ours_mod)  # back to this environment's real branch for every other test


def test_crop_count_and_out_buffer_match_the_allocating_form():
    """crop_count predicts what overlap_crop_image produces; cutting into a caller-owned buffer (the pinned staging
    path of the engine) gives the same bytes as the allocating form."""
    from moondream_amd.image_crops import crop_count

    rng = np.random.default_rng(11)
    for size in [(378, 378), (200, 300), (500, 700), (768, 1024), (420, 1000), (1500, 900)]:
        img = rng.integers(0, 256, (size[0], size[1], 3), dtype=np.uint8)
        ref = overlap_crop_image(img, overlap_margin=4, max_crops=12)
        n, tiling = crop_count(size[0], size[1], 4, 12)
        assert n == ref["crops"].shape[0] and tuple(tiling) == tuple(ref["tiling"])
        out = np.full((n, 378, 378, 3), 7, dtype=np.uint8)
        got = overlap_crop_image(img, overlap_margin=4, max_crops=12, out=out)
        assert got["crops"] is out and np.array_equal(out, ref["crops"])
