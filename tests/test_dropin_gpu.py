"""THE DROP-IN, EXECUTED: the UNMODIFIED reference ``MoondreamModel`` class (imported from a checkout named by
``MOONDREAM_REFERENCE``) with its four seam attributes rebound by ``moondream_amd.integration.bind_reference`` to
libmoondream_hip.so, driven through the reference's OWN public calls -- ``encode_image`` / ``_generate_answer`` behind
``caption``, ``query(image=None)``, ``query(spatial_refs=...)``, ``detect``, ``point`` -- and compared with the goldens
that same class produced on its own (tests/golden/*.npz, oracle/make_golden.py).

Skipped unless a reference checkout is present: the reference is not part of this repository and is not on the GPU box
the driver uses.  ``tools/gpu_r6_dropin.sh`` is how the run recorded in ``profiles/r06_dropin_reference_class.txt``
was made.  Nothing of the product path depends on this file or on the checkout."""
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

from util import bits_to_bf16, leading_wide_objects

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("MOONDREAM_REFERENCE", "/root/reference")
pytestmark = [
    pytest.mark.gpu,
    pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "moondream", "torch")),
                       reason="needs a reference checkout (MOONDREAM_REFERENCE); the default GPU run has none"),
]


def _bound(cfg_name, seed):
    """(reference model on cuda:0 with the seam bound, reference module, make_golden module, binding)."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import make_golden as mg
    from moondream_amd import synth
    from moondream_amd.config import get_config
    from moondream_amd.integration import bind_reference

    cfg = get_config(cfg_name)
    model, ref_md = mg.load_reference(cfg, synth.synthetic_state_dict(cfg, seed=seed))
    model = model.to("cuda:0")
    # the class under test is the checkout's, not this package's mirror
    assert type(model).__module__ == "moondream.torch.moondream"
    assert os.path.realpath(sys.modules[type(model).__module__].__file__).startswith(os.path.realpath(REFERENCE))
    binding = bind_reference(model)
    return cfg, model, ref_md, mg, binding


@pytest.fixture(scope="module")
def tiny_bound(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_seed1.npz"))
    return (g,) + _bound("tiny", int(g["seed"]))


def _ids_equal_up_to_narrow(got, want, margins, what, thr=0.5):
    """ids equal the reference's; a first difference is only tolerated at a decision whose reference top-1 / top-2
    margin is inside the bf16 licence (the caption cases of the goldens are filtered to have none).  Returns the
    number of leading equal ids."""
    j = next((k for k in range(min(len(got), len(want))) if got[k] != want[k]), None)
    if j is None:
        assert len(got) == len(want), (what, got, want)
        return len(want)
    assert float(margins[j]) <= thr, (what, j, got, want, float(margins[j]))
    return j


def _img(index, seed, size=(378, 378)):
    from moondream_amd import synth

    return synth.synthetic_image_array(int(index), int(seed), size)


def test_reference_caption_and_vqa_ids_on_the_hip_seam(tiny_bound):
    """reference encode_image (moondream.py:230-268) + _generate_answer (434-539) over the bound seam == the ids the
    reference produced alone; the EncodedImage snapshot it takes is the K / V the library wrote."""
    g, cfg, model, ref_md, mg, b = tiny_bound
    before = dict(b.calls)
    for idx in range(len(g["image_index"])):
        for kind in ("cap", "vqa")[: 2 if idx == 0 else 1]:
            pfx = f"img{idx}.{kind}."
            image = _img(g["image_index"][idx], g["seed"], tuple(int(s) for s in g[pfx + "size"]))
            want = g[pfx + "tokens"].tolist()
            r = mg.run_reference_caption(model, ref_md, image, g[pfx + "prompt"].tolist(), len(want))
            n_same = _ids_equal_up_to_narrow(r["tokens"], want, g[pfx + "margins"], (idx, kind))
            ref_logits = bits_to_bf16(g[pfx + "step_logits"]).float()[: n_same + 1]   # same context up to there
            got = torch.stack(r["steps"]).float().cpu()[: n_same + 1]
            ok = torch.isfinite(ref_logits)
            assert float((got[ok] - ref_logits[ok]).abs().max()) <= 0.5
        k0 = bits_to_bf16(g[f"img{idx}.cap.k0"]).float()
        stride = int(g["kv_row_stride"])
        got_k0 = model.encode_image(Image.fromarray(image, "RGB")).caches[0][0][0].float().cpu()[:, ::stride]
        assert got_k0.shape == k0.shape
        assert float((got_k0 - k0).pow(2).mean().sqrt() / k0.pow(2).mean().sqrt()) <= 1.5e-2
    # the seam really is the library: every reference call above went through the bound functions
    assert b.calls["_vis_enc"] > before["_vis_enc"] and b.calls["_decode_one_tok"] > before["_decode_one_tok"]
    assert model._prefill.__module__ == "moondream_amd.integration"


def test_reference_text_only_query_on_the_hip_seam(tiny_bound, golden_dir):
    """reference query(image=None) (moondream.py:541-618): it REPLACES its KVCache modules (_setup_caches, :568) and
    passes a plain tril slice -- the binding re-attaches the slab and takes the causal rule."""
    g0, cfg, model, ref_md, mg, b = tiny_bound
    g = np.load(os.path.join(golden_dir, "tiny_textonly.npz"))
    for i in range(int(g["n_cases"])):
        want = g[f"q{i}.tokens"].tolist()
        q = " ".join(str(t) for t in g[f"q{i}.question"].tolist())
        ans = model.query(None, q, settings={"temperature": 0, "max_tokens": len(want)})["answer"]
        assert [int(t) for t in ans.split()] == want, (i, ans, want)
        assert model.text.blocks[0].kv_cache.k_cache.data_ptr() == b.slab_k[0].data_ptr()


@pytest.mark.parametrize("case", ["detect0", "detect1", "point0", "point1"])
def test_reference_detect_and_point_on_the_hip_seam(tiny_bound, golden_dir, case):
    """reference detect / point (moondream.py:735-829 -> _generate_points 653-733): region heads by the reference's own
    ATen code, every decoder step by the library; objects equal up to the first decision narrower than 4 bf16 ulps."""
    g0, cfg, model, ref_md, mg, b = tiny_bound
    g = np.load(os.path.join(golden_dir, "tiny_detect.npz"))
    kind = "detect" if case.startswith("detect") else "point"
    img = Image.fromarray(_img(g[f"{case}.image_index"], g["seed"]), "RGB")
    obj = " ".join(str(t) for t in g["object_ids"].tolist())
    fn = model.detect if kind == "detect" else model.point
    res = fn(img, obj, settings={"max_objects": int(g["max_objects"]), "variant": None})
    objs = res["objects" if kind == "detect" else "points"]
    ref = g[f"{case}.objects"]
    n_ok = leading_wide_objects(g[f"{case}.margins"], 4.0)
    assert n_ok >= 1 and len(objs) >= n_ok
    keys = ("x_min", "y_min", "x_max", "y_max") if kind == "detect" else ("x", "y")
    for k in range(n_ok):
        # the same BINS (the region heads' argmaxes); the floats come out of the reference's own torch.pow / division, which
        # its GPU build and its CPU build round differently in the last bits (1.5e-8 seen)
        assert [objs[k][f] for f in keys] == pytest.approx(ref[k].tolist(), abs=1e-6), (case, k, objs[k], ref[k])


def test_reference_spatial_query_on_the_hip_seam(tiny_bound, golden_dir):
    g0, cfg, model, ref_md, mg, b = tiny_bound
    g = np.load(os.path.join(golden_dir, "tiny_detect.npz"))
    img = Image.fromarray(_img(g["spatial.image_index"], g["seed"]), "RGB")
    refs = [tuple(g["spatial.refs_point"].tolist()), tuple(g["spatial.refs_box"].tolist())]
    q = " ".join(str(t) for t in g["spatial.question"].tolist())
    want = g["spatial.tokens"].tolist()
    ans = model.query(img, q, spatial_refs=refs, settings={"temperature": 0, "max_tokens": 10, "variant": None})["answer"]
    assert [int(t) for t in ans.split()] == want


def test_binding_refuses_what_the_kernels_cannot_honour(tiny_bound):
    g, cfg, model, ref_md, mg, b = tiny_bound
    t = cfg.text
    x = torch.zeros(1, 4, t.dim, dtype=torch.bfloat16, device="cuda:0")
    bad = torch.tril(torch.ones(1, 1, t.max_context, t.max_context, dtype=torch.bool, device="cuda:0"))[:, :, 0:4, :].clone()
    bad[0, 0, 3, 1] = False
    with pytest.raises(ValueError):
        model._prefill(x, bad, torch.arange(4), None)
    with pytest.raises(ValueError):
        model._prefill(x, None, torch.tensor([0, 1, 3, 4]), None)
    with pytest.raises(ValueError):   # a prefix-LM pass that stops short of the prefix (mask None = the prefix-LM slice)
        model._prefill(x, None, torch.arange(4), None)
    row = torch.zeros(1, 1, t.max_context, dtype=torch.bool, device="cuda:0")
    row[:, :, :800] = 1
    with pytest.raises(ValueError):   # the row of position 800 must expose 801 keys
        model._decode_one_tok(x[:, :1], row, torch.tensor([800], device="cuda:0"), None)


def test_unbind_restores_the_reference_seam(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_seed1.npz"))
    cfg, model, ref_md, mg, b = _bound("tiny", int(g["seed"]))
    assert "_prefill" in model.__dict__
    b.unbind()
    assert "_prefill" not in model.__dict__ and model._prefill.__func__ is type(model)._prefill


def test_reference_2b_caption_ids_on_the_hip_seam(golden_dir):
    """Moondream-2B shapes: the three wide-margin seed-1 images of md2b_seed1.npz through the reference's own
    encode_image + _generate_answer on the bound seam."""
    g = np.load(os.path.join(golden_dir, "md2b_seed1.npz"))
    cfg, model, ref_md, mg, b = _bound("2b", int(g["seed"]))
    for idx in range(len(g["image_index"])):
        for kind in ("cap", "vqa")[: 2 if idx == 0 else 1]:
            pfx = f"img{idx}.{kind}."
            image = _img(g["image_index"][idx], g["seed"], tuple(int(s) for s in g[pfx + "size"]))
            want = g[pfx + "tokens"].tolist()
            r = mg.run_reference_caption(model, ref_md, image, g[pfx + "prompt"].tolist(), len(want))
            n_same = _ids_equal_up_to_narrow(r["tokens"], want, g[pfx + "margins"], (idx, kind))
            top_i = torch.from_numpy(g[pfx + "top8_idx"].astype(np.int64))
            got = torch.stack(r["steps"]).float().cpu().gather(1, top_i)
            err = (got - torch.from_numpy(g[pfx + "top8_val"])).abs()[: n_same + 1]   # same context up to the first divergence
            assert float(err.max()) <= 0.5
    del model, b
    torch.cuda.empty_cache()
