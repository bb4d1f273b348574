"""The boundary's ownership mapping, proven on the REFERENCE itself (build container only): with
every block's KV buffers rebound to views of one [L][B][H][ctx][hd] slab -- the layout md_kv_cache
describes -- the unmodified reference code still reproduces the golden token ids, its EncodedImage
snapshot / restore still works, and the struct handed to the library addresses exactly those bytes."""
import os
import sys

import numpy as np
import pytest
import torch

REFERENCE = os.environ.get("MOONDREAM_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "moondream", "torch")),
                                reason="needs the reference checkout (build container)")


@pytest.fixture(scope="module")
def ref(golden_dir):
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(repo, "oracle"))
    import make_golden as mg
    from moondream_amd import synth
    from moondream_amd.config import get_config

    g = np.load(os.path.join(golden_dir, "tiny_seed1.npz"))
    cfg = get_config("tiny")
    model, ref_md = mg.load_reference(cfg, synth.synthetic_state_dict(cfg, seed=int(g["seed"])))
    return g, cfg, model, ref_md, mg


def test_reference_runs_on_one_slab_and_struct_addresses_it(ref):
    from PIL import Image
    from moondream_amd import synth
    from moondream_amd.integration import rebind_kv_caches_to_slab

    g, cfg, model, ref_md, mg = ref
    t = cfg.text
    slab_k, slab_v, kv = rebind_kv_caches_to_slab(model, batch=3)
    # the struct describes the slab the reference now writes into
    bs = t.n_kv_heads * t.max_context * t.head_dim
    assert (kv.k, kv.v, kv.layer_stride, kv.batch_stride, kv.ctx) == (slab_k.data_ptr(), slab_v.data_ptr(), 3 * bs, bs, t.max_context)
    for l, blk in enumerate(model.text.blocks):
        assert blk.kv_cache.k_cache.data_ptr() == kv.k + l * kv.layer_stride * 2
        assert blk.kv_cache.v_cache.data_ptr() == kv.v + l * kv.layer_stride * 2
        assert blk.kv_cache.k_cache.shape == (1, t.n_kv_heads, t.max_context, t.head_dim)
    # the reference's own generation path, unchanged, on the rebound buffers
    for idx in (0, 1):
        img = synth.synthetic_image_array(int(g["image_index"][idx]), int(g["seed"]), (378, 378))
        r = mg.run_reference_caption(model, ref_md, img, g[f"img{idx}.cap.prompt"].tolist(), len(g[f"img{idx}.cap.tokens"]))
        assert r["tokens"] == g[f"img{idx}.cap.tokens"].tolist()
        # encode_image's snapshot is a clone of the first 730 slots of slot 0; the prefill really landed in the slab
        k_last, _ = r["enc"].caches[t.n_layers - 1]
        assert torch.equal(k_last, slab_k[t.n_layers - 1, 0:1, :, :730])
        assert float(slab_k[:, 0].float().abs().sum()) > 0 and float(slab_k[:, 1:].float().abs().sum()) == 0
    # load_encoded_image copies back through the views
    enc = r["enc"]
    slab_k[:, 0].zero_()
    model.load_encoded_image(enc)
    assert torch.equal(slab_k[0, 0:1, :, :730], enc.caches[0][0])


def test_bind_reference_has_no_cpu_path(ref):
    """The drop-in (moondream_amd.integration.bind_reference) fails LOUDLY on a model that is not on a GPU -- there is no CPU
    fallback behind the seam -- and leaves the reference's own methods in place."""
    from moondream_amd import _lib
    from moondream_amd.integration import bind_reference

    g, cfg, model, ref_md, mg = ref
    with pytest.raises(_lib.MoondreamHipError):
        bind_reference(model)
    assert "_mi355x" not in model.__dict__ and getattr(model._prefill, "__module__", "") != "moondream_amd.integration"
