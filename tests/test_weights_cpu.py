"""Checkpoint files: the key layouts and containers the reference's loader accepts
(reference weights.py:30-171) come back as the module-tree state dict."""
import os

import torch

from moondream_amd import synth
from moondream_amd.config import get_config
from moondream_amd.weights import load_state_dict_file


def _to_legacy(sd):
    """Inverse of the reference's weight_map (weights.py:36-109)."""
    pre = {
        "vision.patch_emb": "vision_encoder.encoder.model.visual.patch_embed.linear",
        "vision.post_ln": "vision_encoder.encoder.model.visual.norm",
        "vision.proj_mlp.fc1": "vision_encoder.projection.mlp.fc1",
        "vision.proj_mlp.fc2": "vision_encoder.projection.mlp.fc2",
        "text.post_ln": "text_model.lm_head.ln",
        "text.lm_head": "text_model.lm_head.linear",
        "region.coord_encoder": "region_model.coordinate_encoder",
        "region.coord_decoder.fc1": "region_model.coordinate_decoder.fc1",
        "region.coord_decoder.fc2": "region_model.coordinate_decoder.fc2",
        "region.size_encoder": "region_model.size_encoder",
        "region.size_decoder.fc1": "region_model.size_decoder.fc1",
        "region.size_decoder.fc2": "region_model.size_decoder.fc2",
    }
    out = {}
    for k, v in sd.items():
        if k == "vision.pos_emb":
            out["vision_encoder.encoder.model.visual.pos_embed"] = v
        elif k == "text.wte":
            out["text_model.transformer.embd.wte.weight"] = v
        elif k == "region.coord_features":
            out["region_model.coordinate_features.weight"] = v.T.contiguous()
        elif k == "region.size_features":
            out["region_model.size_features.weight"] = v.T.contiguous()
        elif k.startswith("vision.blocks."):
            i, rest = k[len("vision.blocks."):].split(".", 1)
            rest = rest.replace("ln1", "norm1").replace("ln2", "norm2")
            out[f"vision_encoder.encoder.model.visual.blocks.{i}.{rest}"] = v
        elif k.startswith("text.blocks."):
            i, rest = k[len("text.blocks."):].split(".", 1)
            rest = rest.replace("attn.qkv", "mixer.Wqkv").replace("attn.proj", "mixer.out_proj")
            out[f"text_model.transformer.h.{i}.{rest}"] = v
        else:
            for new, old in pre.items():
                if k.startswith(new + "."):
                    out[old + k[len(new):]] = v
                    break
            else:
                raise AssertionError(f"unmapped key {k}")
    return out


def _same(a, b):
    assert set(a) == set(b), (sorted(set(a) ^ set(b))[:8])
    for k in a:
        assert a[k].shape == b[k].shape and torch.equal(a[k].cpu(), b[k].cpu()), k


def test_checkpoint_layouts_and_containers(tmp_path):
    from safetensors.torch import save_file

    cfg = get_config("tiny")
    sd = {k: v.contiguous() for k, v in synth.synthetic_state_dict(cfg, seed=3).items()}
    # module-tree layout, "model." prefix, safetensors
    p1 = os.path.join(tmp_path, "a.safetensors")
    save_file({"model." + k: v for k, v in sd.items()}, p1)
    _same(load_state_dict_file(p1), sd)
    # legacy layout with torch.compile's infix, safetensors
    legacy = _to_legacy(sd)
    p2 = os.path.join(tmp_path, "b.safetensors")
    save_file({k.replace("visual.blocks", "visual._orig_mod.blocks"): v for k, v in legacy.items()}, p2)
    _same(load_state_dict_file(p2), sd)
    # legacy layout, torch container
    p3 = os.path.join(tmp_path, "c.pt")
    torch.save(legacy, p3)
    _same(load_state_dict_file(p3), sd)


def _quantize_int4(w, group=128):
    """Test-side packer: the inverse of the checkpoint format dequantize_int4 reads (reference layers.py:38-44)."""
    out_f, in_f = w.shape
    rows = w.float().reshape(-1, group)
    lo, hi = rows.min(1, keepdim=True).values, rows.max(1, keepdim=True).values
    scale = ((hi - lo) / 15).clamp_min(1e-8)
    zero = (-lo / scale)
    q = torch.clamp(torch.round(rows / scale + zero), 0, 15).to(torch.uint8)
    step = q.shape[0] // 2
    packed = (q[:step] << 4) | q[step:]
    return packed, scale, zero


def test_int4_checkpoint_is_dequantised_like_the_reference(tmp_path):
    import pytest
    from safetensors.torch import save_file

    from moondream_amd.weights import dequantize_int4

    cfg = get_config("tiny")
    sd = {k: v.contiguous() for k, v in synth.synthetic_state_dict(cfg, seed=5).items()}
    q_names = [f"text.blocks.{i}.{n}" for i in range(cfg.text.n_layers)
               for n in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2")]
    q_names += ["region.coord_encoder", "region.size_decoder.fc1"]
    qsd, expect = dict(sd), dict(sd)
    for n in q_names:
        w = sd[n + ".weight"]
        if w.numel() % 256:
            continue
        packed, scale, zero = _quantize_int4(w)
        del qsd[n + ".weight"]
        qsd[n + ".weight.packed"], qsd[n + ".weight.scale"], qsd[n + ".weight.zero_point"] = packed, scale, zero
        expect[n + ".weight"] = dequantize_int4(packed, scale, zero, w.shape[0])
        # the 4-bit source travels on beside the bf16 weight (the decode regime's weight stream is built from it)
        expect[n + ".int4.packed"], expect[n + ".int4.scale"], expect[n + ".int4.zero_point"] = packed, scale, zero
        # 4-bit round trip of a smooth weight: within one quantisation step
        assert (expect[n + ".weight"].float() - w.float()).abs().max() <= scale.max() * 1.01 + 1e-2
    assert any(k.endswith(".weight.packed") for k in qsd)
    path = os.path.join(tmp_path, "q.safetensors")
    save_file({k: v.contiguous() for k, v in qsd.items()}, path)
    _same(load_state_dict_file(path), expect)

    # bit-exact against the reference's own dequantize_tensor where the reference tree is present
    if not os.path.isdir("/root/reference/moondream/torch"):
        pytest.skip("reference tree not on this machine")
    # (layers.py imports torchao, absent here: evaluate just that function's source, in place)
    src = open("/root/reference/moondream/torch/layers.py").read()
    start = src.index("def dequantize_tensor")
    end = src.index("\n\n\n", start)
    ns = {"torch": torch}
    exec(compile(src[start:end], "reference:layers.py", "exec"), ns)
    g = torch.Generator().manual_seed(0)
    packed = torch.randint(0, 256, (96, 128), dtype=torch.uint8, generator=g)
    scale = torch.rand(192, 1, generator=g) * 0.02 + 1e-3
    zero = torch.rand(192, 1, generator=g) * 15
    ref = ns["dequantize_tensor"](packed, scale, zero, (48, 512))
    assert torch.equal(ref, dequantize_int4(packed, scale, zero, 48))


def test_int4_weight_stream_holds_the_same_weights_as_the_bf16_copy():
    """PackedLinearInt4 (md_linear_fp8.format = MD_WSTREAM_INT4_G128): the fragment-ordered nibbles + (scale, zero) table,
    decoded with the kernel's arithmetic bf16(bf16(q - zero) * scale), are BIT FOR BIT the weights dequantize_int4 (= the
    reference's dequantize_tensor) produces -- single layers and the fused [qkv | fc1] layer; padding channels decode to 0;
    the byte layout is the one include/moondream_hip.h documents (checked on one hand-computed element)."""
    from moondream_amd.weights import PackedLinearInt4, dequantize_int4

    g = torch.Generator().manual_seed(11)
    wa = torch.randn(96, 256, generator=g) * 0.05          # "qkv": 96 channels, 2 groups per row
    wb = torch.randn(160, 256, generator=g) * 0.05         # "fc1"
    (pa, sa, za), (pb, sb, zb) = _quantize_int4(wa), _quantize_int4(wb)
    za, zb = za + 0.37, zb - 0.21                          # fractional zero points: the first rounding matters
    ref = torch.cat([dequantize_int4(pa, sa, za, 96), dequantize_int4(pb, sb, zb, 160)], 0)
    fused = PackedLinearInt4([(pa, sa, za, 96), (pb, sb, zb, 160)], None, "cpu")
    assert (fused.n, fused.k, fused.n_pad, fused.k_pad) == (256, 256, 256, 256)
    assert torch.equal(fused.dequantized()[:256], ref)
    single = PackedLinearInt4([(pb, sb, zb, 160)], None, "cpu")
    assert single.n_pad == 192 and torch.equal(single.dequantized()[:160], ref[96:])
    assert torch.count_nonzero(single.dequantized()[160:].float()) == 0
    st = single.struct()
    assert st.format == 1 and st.n == 160 and st.k == 256 and st.k_pad == 256
    # the documented address of one weight: channel n = 37, feature k = 128 + 64 + 16 * 2 + 8 * 1 + 5 (step 1, kh 1, t 2, hi 1, j 5)
    n, step, kh, t, hi, j = 37, 1, 1, 2, 1, 5
    k = 128 * step + 64 * kh + 16 * t + 8 * hi + j
    q_all = torch.empty(2 * pb.shape[0], 128, dtype=torch.uint8)
    q_all[: pb.shape[0]], q_all[pb.shape[0] :] = (pb & 0xF0) >> 4, pb & 0x0F
    q_want = int(q_all.reshape(160, 256)[n, k])
    raw = single.w.reshape(-1)
    byte_off = (((n // 32) * 2 + step) * 2 + kh) * 1024 + (hi * 32 + n % 32) * 16 + 4 * t + j // 2
    assert (int(raw[byte_off]) >> (4 * (j % 2))) & 15 == q_want
    assert float(single.qparams[step, n, 0]) == float(sb.reshape(160, 2)[n, step]) and float(single.qparams[step, n, 1]) == float(zb.reshape(160, 2)[n, step])


def test_fp8_fragment_layout_and_scales():
    """PackedLinearFp8: per-channel scale maps max |w| to 448, the fragment-ordered bytes are the row-major e4m3
    bytes permuted as include/moondream_hip.h documents (md_linear_fp8), padding is zero."""
    from moondream_amd.weights import PackedLinear, PackedLinearFp8

    g = torch.Generator().manual_seed(3)
    n, k = 96, 200
    w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16)
    w[5] = 0  # an all-zero channel keeps scale 1
    lin = PackedLinear(w, torch.zeros(n), "cpu")
    q = PackedLinearFp8(lin.w, lin.b, n, k)
    assert (q.n_pad, q.k_pad) == (128, 256) and q.scale.shape == (128,)
    amax = lin.w.float().abs().amax(1)
    assert torch.allclose(q.scale[amax > 0], amax[amax > 0] / 448.0) and float(q.scale[5]) == 1.0
    rows = q.q.view(torch.uint8)
    frag = q.w.reshape(-1)
    for (ch, feat) in [(0, 0), (31, 7), (32, 8), (77, 199), (95, 31), (64, 48), (127, 255), (3, 130)]:
        nb, r, kb, rem = ch // 32, ch % 32, feat // 32, feat % 32
        step, hi, j = rem // 16, (rem % 16) // 8, rem % 8
        lane = hi * 32 + r
        off = ((nb * (q.k_pad // 32) + kb) * 64 + lane) * 16 + step * 8 + j
        assert int(frag[off]) == int(rows[ch, feat]), (ch, feat)
    assert int(rows[n:].max()) == 0 and int(rows[:, k:].max()) == 0
    # quantisation error: half an e4m3 ulp relative to the channel maximum's binade
    err = (q.dequantized()[:n, :k] - w.float()).abs()
    assert float((err / amax[:n, None].clamp_min(1e-30)).max()) <= 2.0 ** -4 + 1e-6


def test_reference_config_jsons_load_field_for_field():
    """Build container only: the reference's shipped JSON configs (moondream/config/config_md2.json, config_md05.json) load
    through this package's MoondreamConfig.from_dict to the same field values as through the reference's own dataclasses
    (config.py:5-94), and the dataclass defaults agree too (a config file may omit any field)."""
    import dataclasses
    import importlib.util
    import json
    import os

    import pytest

    from moondream_amd.config import MoondreamConfig

    ref_root = os.environ.get("MOONDREAM_REFERENCE", "/root/reference")
    path = os.path.join(ref_root, "moondream", "torch", "config.py")
    if not os.path.isfile(path):
        pytest.skip("needs the reference checkout (build container)")
    spec = importlib.util.spec_from_file_location("ref_config", path)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    def same(ours, theirs, where):
        for f in dataclasses.fields(theirs):
            a, b = getattr(ours, f.name), getattr(theirs, f.name)
            if dataclasses.is_dataclass(b):
                same(a, b, f"{where}.{f.name}")
            else:
                assert a == b, (where, f.name, a, b)

    same(MoondreamConfig(), ref.MoondreamConfig(), "defaults")
    for name in ("config_md2.json", "config_md05.json"):
        with open(os.path.join(ref_root, "moondream", "config", name)) as f:
            d = json.load(f)
        same(MoondreamConfig.from_dict(d), ref.MoondreamConfig.from_dict(d), name)
