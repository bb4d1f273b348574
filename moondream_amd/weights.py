"""Checkpoint -> packed device layouts + the C structs the HIP library reads.

Input is a ``state_dict`` keyed like the reference module tree
(``vision.blocks.0.attn.qkv.weight`` ...; reference: vision.py:92-147,
text.py:175-221, moondream.py:94-136).  ``remap_legacy_keys`` additionally
accepts the older on-disk names the reference's loader maps
(reference: weights.py:36-109).

Packing (done once at load time, the GEMM kernel's contract in
include/moondream_hip.h): every nn.Linear weight [n, k] becomes a zero-padded
[n_pad, k_pad] bf16 matrix with n_pad, k_pad rounded up to 64 -- this is how
K = 588 (patch embedding) and FF = 4304 (ViT MLP) reach MFMA-friendly shapes
without touching any activation layout in HBM.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional

import torch

from . import _lib
from .config import MoondreamConfig

BF16 = torch.bfloat16


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def reference_pixel_lut() -> torch.Tensor:
    """bf16 [256]: what the reference's in-place bf16 normalisation makes of each
    byte value (reference: vision.py:33-40: ``.to(bf16).div_(255.0).sub_(0.5).div_(0.5)``)."""
    return torch.arange(256, dtype=torch.uint8).to(BF16).div_(255.0).sub_(0.5).div_(0.5)


def rope_table(rot_dim: int, max_context: int, theta: float = 10000.0) -> torch.Tensor:
    """fp32 [max_context, rot_dim/2, 2] = (cos, sin) of pos * theta^(-2j/rot_dim)
    (reference: rope.py:6-17, called with dim = rot_dim at text.py:214-218)."""
    freqs = 1.0 / (theta ** (torch.arange(0, rot_dim, 2, dtype=torch.float32)[: rot_dim // 2] / rot_dim))
    ang = torch.arange(max_context, dtype=torch.float32).unsqueeze(1) * freqs.unsqueeze(0)
    cis = torch.exp(1j * ang)
    return torch.stack([cis.real, cis.imag], dim=-1).contiguous()


class PackedLinear:
    def __init__(self, w: torch.Tensor, b: Optional[torch.Tensor], device):
        n, k = w.shape
        self.n, self.k = n, k
        self.n_pad, self.k_pad = _round_up(n, 64), _round_up(k, 64)
        wp = torch.zeros(self.n_pad, self.k_pad, dtype=BF16, device=device)
        wp[:n, :k] = w.to(device=device, dtype=BF16)
        bp = torch.zeros(self.n_pad, dtype=BF16, device=device)
        if b is not None:
            bp[:n] = b.to(device=device, dtype=BF16)
        self.w, self.b = wp, bp

    def struct(self) -> _lib.MdLinear:
        return _lib.MdLinear(self.w.data_ptr(), self.b.data_ptr(), self.n, self.k, self.n_pad, self.k_pad)


class FusedLinear:
    """[rows of a | rows of b] over the same input: ONE packed matrix; ``a`` and ``b``
    are exposed as row-range views of it (no second copy of the weights)."""

    def __init__(self, wa, ba, wb, bb, device):
        na, k = wa.shape
        nb = wb.shape[0]
        assert wb.shape[1] == k and na % 64 == 0, "the first layer's rows must end on a 64-row boundary"
        self.k, self.k_pad = k, _round_up(k, 64)
        self.na, self.nb = na, nb
        self.nb_pad = _round_up(nb, 64)
        self.n, self.n_pad = na + nb, na + self.nb_pad
        self.w = torch.zeros(self.n_pad, self.k_pad, dtype=BF16, device=device)
        self.w[:na, :k] = wa.to(device=device, dtype=BF16)
        self.w[na : na + nb, :k] = wb.to(device=device, dtype=BF16)
        self.b = torch.zeros(self.n_pad, dtype=BF16, device=device)
        self.b[:na] = ba.to(device=device, dtype=BF16)
        self.b[na : na + nb] = bb.to(device=device, dtype=BF16)

    def struct(self) -> _lib.MdLinear:
        return _lib.MdLinear(self.w.data_ptr(), self.b.data_ptr(), self.n, self.k, self.n_pad, self.k_pad)

    def struct_a(self) -> _lib.MdLinear:
        return _lib.MdLinear(self.w.data_ptr(), self.b.data_ptr(), self.na, self.k, self.na, self.k_pad)

    def struct_b(self) -> _lib.MdLinear:
        off = self.na
        return _lib.MdLinear(self.w.data_ptr() + off * self.k_pad * 2, self.b.data_ptr() + off * 2, self.nb, self.k,
                             self.nb_pad, self.k_pad)


FP8_MAX = 448.0  # largest finite OCP e4m3fn value


class PackedLinearFp8:
    """An already packed bf16 linear (``w`` [n_pad][k_pad], zero padded; ``b`` bf16 [n_pad] or None) quantised
    to OCP e4m3fn with one fp32 scale per output channel (max |w| -> 448) and laid out in the MFMA-fragment
    order md_linear_fp8 documents (include/moondream_hip.h): an opt-in copy for the decode regime's weight
    stream; the bf16 original stays in place for every other launch."""

    def __init__(self, w: torch.Tensor, b: Optional[torch.Tensor], n: int, k: int):
        assert w.dim() == 2 and w.shape[0] % 64 == 0
        self.n, self.k, self.n_pad = n, k, w.shape[0]
        self.k_pad = _round_up(w.shape[1], 128)
        wf = torch.zeros(self.n_pad, self.k_pad, dtype=torch.float32, device=w.device)
        wf[:, : w.shape[1]] = w.float()
        amax = wf.abs().amax(dim=1)
        self.scale = torch.where(amax > 0, amax / FP8_MAX, torch.ones_like(amax)).contiguous()
        q = (wf / self.scale[:, None]).clamp_(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
        self.q = q  # row-major [n_pad][k_pad]: what the layer computes with is  q.float() * scale[:, None]
        frag = q.view(torch.uint8).view(self.n_pad // 32, 32, self.k_pad // 32, 2, 2, 8)
        self.w = frag.permute(0, 2, 4, 1, 3, 5).contiguous()  # [nb][kb][hi][row][step][j]
        self.b = b

    def dequantized(self) -> torch.Tensor:
        return self.q.float() * self.scale[:, None]

    def struct(self) -> _lib.MdLinearFp8:
        return _lib.MdLinearFp8(self.w.data_ptr(), self.scale.data_ptr(), self.b.data_ptr() if self.b is not None else None,
                                self.n, self.k, self.n_pad, self.k_pad)


class PackedLinearF8:
    """An already packed bf16 linear (``w`` [n_pad][k_pad], zero padded; ``b`` bf16 [n_pad] or None) quantised to OCP
    e4m3fn with one fp32 scale per output channel (max |w| -> 448), ROW-MAJOR with K contiguous: the operand layout
    of md_gemm_f8 (include/moondream_hip.h: md_linear_f8).  An opt-in copy for the MFMA-bound launches of the fp8
    mode; the bf16 original stays in place."""

    def __init__(self, w: torch.Tensor, b: Optional[torch.Tensor], n: int, k: int):
        assert w.dim() == 2 and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0
        self.n, self.k, self.n_pad, self.k_pad = n, k, w.shape[0], w.shape[1]
        wf = w.float()
        amax = wf.abs().amax(dim=1)
        self.scale = torch.where(amax > 0, amax / FP8_MAX, torch.ones_like(amax)).contiguous()
        self.q = (wf / self.scale[:, None]).clamp_(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).contiguous()
        self.b = b

    def dequantized(self) -> torch.Tensor:
        return self.q.float() * self.scale[:, None]

    def struct(self) -> _lib.MdLinearF8:
        return _lib.MdLinearF8(self.q.data_ptr(), self.scale.data_ptr(), self.b.data_ptr() if self.b is not None else None,
                               self.n, self.k, self.n_pad, self.k_pad)


class PackedLayerNorm:
    def __init__(self, w: torch.Tensor, b: torch.Tensor, device):
        self.w = w.to(device=device, dtype=BF16).contiguous()
        self.b = b.to(device=device, dtype=BF16).contiguous()

    def struct(self) -> _lib.MdLayerNorm:
        return _lib.MdLayerNorm(self.w.data_ptr(), self.b.data_ptr())


LEGACY_KEY_PREFIXES = {
    # old checkpoint layout -> module-tree layout (reference: weights.py:36-109)
    "vision_encoder.encoder.model.visual.patch_embed.linear": "vision.patch_emb",
    "vision_encoder.encoder.model.visual.norm": "vision.post_ln",
    "vision_encoder.projection.mlp.fc1": "vision.proj_mlp.fc1",
    "vision_encoder.projection.mlp.fc2": "vision.proj_mlp.fc2",
    "text_model.lm_head.ln": "text.post_ln",
    "text_model.lm_head.linear": "text.lm_head",
    "region_model.coordinate_encoder": "region.coord_encoder",
    "region_model.coordinate_decoder.fc1": "region.coord_decoder.fc1",
    "region_model.coordinate_decoder.fc2": "region.coord_decoder.fc2",
    "region_model.size_encoder": "region.size_encoder",
    "region_model.size_decoder.fc1": "region.size_decoder.fc1",
    "region_model.size_decoder.fc2": "region.size_decoder.fc2",
}


def remap_legacy_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Accept the older safetensors key layout (reference: weights.py:36-109)."""
    if any(k.startswith("vision.") or k.startswith("text.") for k in sd):
        return sd
    out = {}
    for k, v in sd.items():
        nk = k
        if k == "vision_encoder.encoder.model.visual.pos_embed":
            nk = "vision.pos_emb"
        elif k == "text_model.transformer.embd.wte.weight":
            nk = "text.wte"
        elif k == "region_model.coordinate_features.weight":
            nk, v = "region.coord_features", v.T
        elif k == "region_model.size_features.weight":
            nk, v = "region.size_features", v.T
        elif k.startswith("vision_encoder.encoder.model.visual.blocks."):
            rest = k[len("vision_encoder.encoder.model.visual.blocks."):]
            i, rest = rest.split(".", 1)
            rest = rest.replace("norm1", "ln1").replace("norm2", "ln2")
            nk = f"vision.blocks.{i}.{rest}"
        elif k.startswith("text_model.transformer.h."):
            rest = k[len("text_model.transformer.h."):]
            i, rest = rest.split(".", 1)
            rest = rest.replace("mixer.Wqkv", "attn.qkv").replace("mixer.out_proj", "attn.proj")
            nk = f"text.blocks.{i}.{rest}"
        else:
            for old, new in LEGACY_KEY_PREFIXES.items():
                if k.startswith(old + "."):
                    nk = new + k[len(old):]
                    break
        out[nk] = v
    return out


def dequantize_int4(packed: torch.Tensor, scale: torch.Tensor, zero_point: torch.Tensor, out_features: int) -> torch.Tensor:
    """The reference's 4-bit checkpoint format back to a bf16 [out, in] weight (layers.py:38-44): ``packed`` is
    uint8 [out*in/256, 128]; the high nibbles are the first half of the group rows, the low nibbles the second
    half; every 128-wide group row has one ``scale`` and one ``zero_point``; the arithmetic runs in place on a bf16
    tensor (subtract, round, multiply, round) so the result is bit-identical to the reference's dequantize_tensor."""
    step = packed.shape[0]
    w = torch.empty(2 * step, packed.shape[1], dtype=BF16, device=packed.device)
    w[:step] = (packed & 0xF0) >> 4
    w[step:] = packed & 0x0F
    w.sub_(zero_point).mul_(scale)
    if w.numel() % out_features:
        raise ValueError(f"int4 weight of {w.numel()} elements does not divide into {out_features} rows")
    return w.reshape(out_features, -1)


def dequantize_int4_entries(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Replace every ``<p>.weight.packed / .scale / .zero_point`` triple (the reference's QuantizedLinear parameters,
    layers.py:57-74; text blocks' qkv/proj/fc1/fc2 when config.text.group_size is set, text.py:178, the region
    coordinate/size encoders and decoders when config.region.group_size is set, moondream.py:95) by the bf16
    ``<p>.weight`` the kernels consume.  The reference then re-quantises that tensor for torchao's int4 GEMM
    (layers.py:100); this framework runs it as bf16 -- 4x the weight bytes, none of the second rounding."""
    packed = [k for k in sd if k.endswith(".weight.packed")]
    if not packed:
        return sd
    out = dict(sd)
    for k in packed:
        p = k[: -len(".weight.packed")]
        bias = sd.get(p + ".bias")
        if bias is None:
            raise KeyError(f"{p}.bias is needed to recover the shape of the int4 weight {k}")
        out[p + ".weight"] = dequantize_int4(
            sd[k], sd[p + ".weight.scale"], sd[p + ".weight.zero_point"], bias.shape[0])
        # the 4-bit source stays beside the bf16 weight (INT4_KEYS): PackedModel turns the decoder blocks' into the decode
        # regime's weight stream (PackedLinearInt4: the same weights at a quarter of the bytes)
        for suffix, kept in zip((".weight.packed", ".weight.scale", ".weight.zero_point"), INT4_KEYS):
            out[p + kept] = out.pop(p + suffix)
    return out


INT4_KEYS = (".int4.packed", ".int4.scale", ".int4.zero_point")


def int4_source(sd: Dict[str, torch.Tensor], prefix: str):
    """(packed, scale, zero_point) of ``prefix`` if the checkpoint had it as a QuantizedLinear, else None."""
    ks = [prefix + s for s in INT4_KEYS]
    return tuple(sd[k] for k in ks) if all(k in sd for k in ks) else None


class PackedLinearInt4:
    """The reference's 4-bit checkpoint format (layers.py:38-74: groups of 128 input features, one scale and one zero_point
    per group) laid out as the decode regime's weight stream (include/moondream_hip.h: md_linear_fp8 with format
    MD_WSTREAM_INT4_G128).  ``sources``: one (packed, scale, zero_point, out_features) per layer, concatenated along the
    output channels (the fused [qkv | fc1] layer is two).  The kernel rebuilds every weight as
    bf16(bf16(q - zero_point) * scale) -- ``dequantize_int4``'s arithmetic -- so the stream and the bf16 copy hold the same
    weights bit for bit (``dequantized()`` evaluates the kernel's formula from the packed stream; the CPU test compares)."""

    def __init__(self, sources, b: Optional[torch.Tensor], device):
        qs, scs, zps = [], [], []
        for packed, scale, zero, out_features in sources:  # (repacked on `device`: 1.2 G nibbles at 2B)
            packed = packed.to(device)
            step = packed.shape[0]
            q = torch.empty(2 * step, packed.shape[1], dtype=torch.uint8, device=device)
            q[:step] = (packed & 0xF0) >> 4
            q[step:] = packed & 0x0F
            k = q.numel() // out_features
            if q.numel() % out_features or k % 128 or packed.shape[1] != 128:
                raise ValueError("int4 weight stream: groups of 128 input features, in_features % 128 == 0")
            qs.append(q.reshape(out_features, k))
            scs.append(scale.to(device).float().reshape(out_features, k // 128))
            zps.append(zero.to(device).float().reshape(out_features, k // 128))
        q, sc, zp = torch.cat(qs, 0), torch.cat(scs, 0), torch.cat(zps, 0)
        self.n, self.k = q.shape
        self.n_pad, self.k_pad = _round_up(self.n, 64), self.k
        steps = self.k // 128
        qp = torch.zeros(self.n_pad, self.k, dtype=torch.uint8, device=device)
        qp[: self.n] = q
        par = torch.zeros(steps, self.n_pad, 2, dtype=torch.float32, device=device)  # padding channels: scale 0 -> weight 0
        par[:, : self.n, 0] = sc.t()
        par[:, : self.n, 1] = zp.t()
        # [nb][row][step][kh][t][hi][j] -> [nb][step][kh][hi][row][t][j], then two nibbles per byte (nibble j = bits 4 j ..)
        frag = qp.view(self.n_pad // 32, 32, steps, 2, 4, 2, 8).permute(0, 2, 3, 5, 1, 4, 6).contiguous()
        self.w = (frag[..., 0::2] | (frag[..., 1::2] << 4)).contiguous().to(device)  # [...][t][4 bytes]
        self.qparams = par.contiguous().to(device)
        self.b = b

    def dequantized(self) -> torch.Tensor:
        """bf16 [n_pad][k] rebuilt FROM THE STREAM with the kernel's arithmetic (test instrument)."""
        w = self.w.cpu()
        nb, steps = w.shape[0], w.shape[1]
        nib = torch.stack([w & 0x0F, w >> 4], dim=-1).reshape(nb, steps, 2, 2, 32, 4, 8)  # [nb][step][kh][hi][row][t][j]
        q = nib.permute(0, 4, 1, 2, 5, 3, 6).reshape(self.n_pad, self.k).float()
        par = self.qparams.cpu()
        sc = par[:, :, 0].t().repeat_interleave(128, dim=1)
        zp = par[:, :, 1].t().repeat_interleave(128, dim=1)
        return ((q - zp).to(BF16).float() * sc).to(BF16)

    def struct(self) -> _lib.MdLinearFp8:
        return _lib.MdLinearFp8(self.w.data_ptr(), self.qparams.data_ptr(), self.b.data_ptr() if self.b is not None else None,
                                self.n, self.k, self.n_pad, self.k_pad, _lib.MD_WSTREAM_INT4_G128)


def load_state_dict_file(weights_file: str) -> Dict[str, torch.Tensor]:
    """Read a checkpoint the way the reference's ``load_weights_into_model`` does
    (weights.py:112-171): ``.safetensors`` or a torch ``.pt`` state dict, in the
    module-tree key layout (optionally prefixed ``model.``; 4-bit QuantizedLinear
    triples are dequantised to bf16, see dequantize_int4_entries) or the older
    ``vision_encoder.* / text_model.* / region_model.*`` layout (optionally with
    torch.compile's ``._orig_mod`` infix).  Returns module-tree keys; tensors
    stay on the CPU (``PackedModel`` moves and packs them)."""
    if weights_file.endswith(".safetensors"):
        import safetensors

        with safetensors.safe_open(weights_file, framework="pt") as st:
            raw = {k: st.get_tensor(k) for k in st.keys()}
    else:
        raw = torch.load(weights_file, map_location="cpu", weights_only=True)
    norm = {}
    for k, v in raw.items():
        k = k.replace("._orig_mod", "")
        if k.startswith("model."):
            k = k[len("model."):]
        norm[k] = v
    return dequantize_int4_entries(remap_legacy_keys(norm))


class PackedLora:
    """A LoRA variant packed for md_text_forward_lora: per block an md_text_block_lora of bias-free (A, B) pairs,
    A [r, in] and B [out, r] zero-padded to 64-multiples like every other linear (reference layout: lora.py:54-79)."""

    def __init__(self, config: MoondreamConfig, lora: dict, device):
        t = config.text
        self._keep: List[PackedLinear] = []
        self.blocks = (_lib.MdTextBlockLora * t.n_layers)()
        tree = lora.get("text", {}).get("blocks", {})
        for i in range(t.n_layers):
            layer = tree.get(str(i), {})
            for group, names in (("attn", ("qkv", "proj")), ("mlp", ("fc1", "fc2"))):
                for name in names:
                    pair = layer.get(group, {}).get(name)
                    if pair is None:
                        continue
                    a = PackedLinear(pair["A"], None, device)
                    b = PackedLinear(pair["B"], None, device)
                    assert a.n_pad <= 256 and b.k_pad == a.n_pad, "LoRA rank must be <= 256"
                    self._keep += [a, b]
                    setattr(self.blocks[i], name, _lib.MdLoraPair(a.struct(), b.struct()))

    def ptr(self):
        return C.cast(self.blocks, C.c_void_p)


class PackedModel:
    """All weights resident on one device + the md_vit_model / md_text_model
    structs (kept alive here; the library only borrows the pointers)."""

    def __init__(self, config: MoondreamConfig, state_dict: Dict[str, torch.Tensor], device):
        self.config = config
        self.device = torch.device(device)
        sd = dequantize_int4_entries(remap_legacy_keys(state_dict))
        self.sd_keys = set(sd)
        v, t = config.vision, config.text
        dev = self.device
        self._keep: List[object] = []

        def lin(prefix):
            p = PackedLinear(sd[prefix + ".weight"], sd.get(prefix + ".bias"), dev)
            self._keep.append(p)
            return p

        def ln(prefix):
            p = PackedLayerNorm(sd[prefix + ".weight"], sd[prefix + ".bias"], dev)
            self._keep.append(p)
            return p

        # ---------------- vision
        self.patch_emb = lin("vision.patch_emb")
        self.pos_emb = sd["vision.pos_emb"].to(device=dev, dtype=BF16).reshape(v.n_patches, v.enc_dim).contiguous()
        self.pixel_lut = reference_pixel_lut().to(dev)
        self.vit_blocks = (_lib.MdVitBlock * v.enc_n_layers)()
        self._vit_packed: List[dict] = []  # per block: the packed bf16 linears, for the fp8 mode's copies
        for i in range(v.enc_n_layers):
            p = f"vision.blocks.{i}"
            blk = self.vit_blocks[i]
            pk = {"qkv": lin(p + ".attn.qkv"), "proj": lin(p + ".attn.proj"), "fc1": lin(p + ".mlp.fc1"), "fc2": lin(p + ".mlp.fc2")}
            self._vit_packed.append(pk)
            blk.ln1 = ln(p + ".ln1").struct()
            blk.qkv = pk["qkv"].struct()
            blk.proj = pk["proj"].struct()
            blk.ln2 = ln(p + ".ln2").struct()
            blk.fc1 = pk["fc1"].struct()
            blk.fc2 = pk["fc2"].struct()
        self.vit_post_ln = ln("vision.post_ln")
        self.proj_fc1 = lin("vision.proj_mlp.fc1")
        self.proj_fc2 = lin("vision.proj_mlp.fc2")
        self.vit = _lib.MdVitModel(
            v.enc_dim, v.enc_n_heads, v.enc_n_layers, v.enc_ff_dim, v.enc_patch_size, v.crop_size,
            self.patch_emb.struct(), self.pos_emb.data_ptr(),
            C.cast(self.vit_blocks, C.POINTER(_lib.MdVitBlock)), self.vit_post_ln.struct(),
            self.proj_fc1.struct(), self.proj_fc2.struct(), self.pixel_lut.data_ptr(),
        )

        # ---------------- text
        self.text_blocks = (_lib.MdTextBlock * t.n_layers)()
        self._text_packed: List[dict] = []  # per block: the packed bf16 linears, for enable_fp8_decode
        for i in range(t.n_layers):
            p = f"text.blocks.{i}"
            blk = self.text_blocks[i]
            blk.ln = ln(p + ".ln").struct()
            if t.qkv_dim % 64 == 0:
                # qkv and fc1 share their input: pack once, run as one GEMM
                fused = FusedLinear(sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"],
                                    sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"], dev)
                self._keep.append(fused)
                blk.qkv, blk.fc1, blk.qkv_fc1 = fused.struct_a(), fused.struct_b(), fused.struct()
            else:
                fused = None
                blk.qkv = lin(p + ".attn.qkv").struct()
                blk.fc1 = lin(p + ".mlp.fc1").struct()
            proj, fc2 = lin(p + ".attn.proj"), lin(p + ".mlp.fc2")
            blk.proj, blk.fc2 = proj.struct(), fc2.struct()
            self._text_packed.append({"qkv_fc1": fused, "proj": proj, "fc2": fc2,
                                      "int4": {n: int4_source(sd, f"{p}.{n}") for n in ("attn.qkv", "mlp.fc1", "attn.proj", "mlp.fc2")}})
        self.text_post_ln = ln("text.post_ln")
        self.lm_head = lin("text.lm_head")
        self.wte = sd["text.wte"].to(device=dev, dtype=BF16).contiguous()
        self.freqs = rope_table(t.rot_dim, t.max_context).to(dev)
        self.text = _lib.MdTextModel(
            t.dim, t.n_heads, t.n_kv_heads, t.n_layers, t.ff_dim, t.vocab_size, t.max_context,
            t.prefix_attn, t.rot_dim, C.cast(self.text_blocks, C.POINTER(_lib.MdTextBlock)),
            self.text_post_ln.struct(), self.lm_head.struct(), self.wte.data_ptr(), self.freqs.data_ptr(),
        )

        # ---------------- region head (detect / point), optional
        self.region = None
        if "region.coord_encoder.weight" in sd:
            self.region = {
                "coord_encoder": lin("region.coord_encoder"),
                "coord_dec_fc1": lin("region.coord_decoder.fc1"),
                "coord_dec_fc2": lin("region.coord_decoder.fc2"),
                "size_encoder": lin("region.size_encoder"),
                "size_dec_fc1": lin("region.size_decoder.fc1"),
                "size_dec_fc2": lin("region.size_decoder.fc2"),
                "coord_features": sd["region.coord_features"].to(device=dev, dtype=BF16).contiguous(),
                "size_features": sd["region.size_features"].to(device=dev, dtype=BF16).contiguous(),
            }

    def has_int4_source(self) -> bool:
        """Every decoder block's four linears came from the checkpoint as QuantizedLinear triples (text.py:178 with
        config.text.group_size set) and the fused qkv|fc1 packing is in use."""
        t = self.config.text
        return bool(self._text_packed) and t.dim % 128 == 0 and t.ff_dim % 128 == 0 and all(  # a group = a whole 128-wide K step of a row
            pk["qkv_fc1"] is not None and all(v is not None for v in pk["int4"].values()) for pk in self._text_packed)

    def enable_int4_decode(self) -> None:
        """Attach the checkpoint's own 4-bit weights as the decode regime's weight stream (md_text_model.fp8 with
        md_linear_fp8.format = MD_WSTREAM_INT4_G128; lm_head has no quantised form in the reference and stays bf16): launches
        of <= 64 rows then read a QUARTER of the weight bytes and rebuild the very bf16 weights the other launches multiply
        with (bf16(bf16(q - zero) * scale), layers.py:38-44).  Not a numerical mode (results agree with the bf16 stream to
        fp32 accumulation order); opt-in while the in-register dequantisation is VALU-bound (MoondreamModel.enable_int4_decode)."""
        if getattr(self, "_fp8", None) is not None:
            return
        if not self.has_int4_source():
            raise ValueError("no QuantizedLinear source for every decoder block in this checkpoint")
        t = self.config.text
        keep: List[PackedLinearInt4] = []
        dev = self.lm_head.w.device

        def q4(out_features, pk_lin, srcs):
            f = PackedLinearInt4([(p_, s_, z_, n_) for (p_, s_, z_), n_ in zip(srcs, out_features)], pk_lin.b, dev)
            assert (f.n, f.k, f.n_pad) == (pk_lin.n, pk_lin.k, pk_lin.n_pad), (f.n, f.k, f.n_pad, pk_lin.n, pk_lin.k, pk_lin.n_pad)
            keep.append(f)
            return f.struct()

        qkv_n = (t.n_heads + 2 * t.n_kv_heads) * (t.dim // t.n_heads)
        blocks = (_lib.MdTextBlockFp8 * t.n_layers)()
        for i, pk in enumerate(self._text_packed):
            src = pk["int4"]
            blocks[i].qkv_fc1 = q4((qkv_n, t.ff_dim), pk["qkv_fc1"], (src["attn.qkv"], src["mlp.fc1"]))
            blocks[i].proj = q4((pk["proj"].n,), pk["proj"], (src["attn.proj"],))
            blocks[i].fc2 = q4((pk["fc2"].n,), pk["fc2"], (src["mlp.fc2"],))
        head = _lib.MdLinearFp8()  # w == NULL: lm_head stays on the bf16 stream
        self._fp8_blocks = blocks
        self._fp8 = _lib.MdTextFp8(C.cast(blocks, C.POINTER(_lib.MdTextBlockFp8)), head)
        self._fp8_keep = keep
        self._fp8_is_int4 = True
        self.text.fp8 = C.pointer(self._fp8)

    def enable_fp8_decode(self) -> None:
        """Attach FP8 (e4m3fn, per-channel scale) copies of the decoder's weight stream -- fused qkv|fc1, proj,
        fc2 of every block and lm_head -- to md_text_model.fp8: launches of <= 64 rows (decode steps) then read half
        the weight bytes; prefill keeps the bf16 weights.  An opt-in numerical mode (BASELINE configs[4]); the
        default path stays at the reference's precision.  Call before capturing hipGraphs."""
        if getattr(self, "_fp8", None) is not None:
            return
        t = self.config.text
        keep: List[PackedLinearFp8] = []

        def q8(p, n, k):
            f = PackedLinearFp8(p.w, p.b, n, k)
            keep.append(f)
            return f.struct()

        blocks = (_lib.MdTextBlockFp8 * t.n_layers)()
        for i, pk in enumerate(self._text_packed):
            if pk["qkv_fc1"] is None:
                raise ValueError("fp8 decode needs the fused qkv|fc1 packing (qkv_dim % 64 == 0)")
            blocks[i].qkv_fc1 = q8(pk["qkv_fc1"], pk["qkv_fc1"].n, pk["qkv_fc1"].k)
            blocks[i].proj = q8(pk["proj"], pk["proj"].n, pk["proj"].k)
            blocks[i].fc2 = q8(pk["fc2"], pk["fc2"].n, pk["fc2"].k)
        head = q8(self.lm_head, self.lm_head.n, self.lm_head.k)
        self._fp8_blocks = blocks
        self._fp8 = _lib.MdTextFp8(C.cast(blocks, C.POINTER(_lib.MdTextBlockFp8)), head)
        self._fp8_keep = keep
        self.text.fp8 = C.pointer(self._fp8)

    def disable_fp8_decode(self) -> None:
        self.text.fp8 = C.POINTER(_lib.MdTextFp8)()
        self._fp8 = None
        self._fp8_is_int4 = False

    # ---- FP8 mode of the MFMA-bound launches (ViT, projector, decoder prefill): md_gemm_f8 -------------------------------
    def begin_f8_calibration(self) -> None:
        """Attach calibration buffers (md_vit_f8.calib / md_text_f8.calib): the bf16 path keeps running and records the
        running max |x| of every tensor the fp8 mode will quantise."""
        v, t = self.config.vision, self.config.text
        self.disable_f8()
        self._f8_calib_vit = torch.zeros(4 * v.enc_n_layers + 2, dtype=torch.float32, device=self.device)
        self._f8_calib_text = torch.zeros(3 * t.n_layers, dtype=torch.float32, device=self.device)
        self._f8_vit = _lib.MdVitF8(C.POINTER(_lib.MdVitBlockF8)(), _lib.MdLinearF8(), _lib.MdLinearF8(), 0.0, 0.0,
                                   self._f8_calib_vit.data_ptr())
        self._f8_text = _lib.MdTextF8(C.POINTER(_lib.MdTextBlockF8)(), self._f8_calib_text.data_ptr())
        self.vit.f8 = C.pointer(self._f8_vit)
        self.text.f8 = C.pointer(self._f8_text)

    def finish_f8_calibration(self, margin: float = 1.5) -> dict:
        """Turn the recorded ranges into static per-tensor scales (max |x| * margin -> 448: e4m3 is a floating-point
        format, so head-room costs no precision, only range at the small end) and attach e4m3 copies (one fp32 scale per
        output channel) of every linear of the ViT blocks, the projector and the decoder blocks.  Returns the ranges."""
        v, t = self.config.vision, self.config.text
        av = self._f8_calib_vit.cpu().tolist()
        at = self._f8_calib_text.cpu().tolist()
        sc = lambda a: (a * margin / FP8_MAX) if a > 0 else 1.0
        keep: List[PackedLinearF8] = []

        def q8(p):
            f = PackedLinearF8(p.w, p.b, p.n, p.k)
            keep.append(f)
            return f.struct()

        vb = (_lib.MdVitBlockF8 * v.enc_n_layers)()
        for i, pk in enumerate(self._vit_packed):
            vb[i].qkv, vb[i].proj, vb[i].fc1, vb[i].fc2 = q8(pk["qkv"]), q8(pk["proj"]), q8(pk["fc1"]), q8(pk["fc2"])
            vb[i].s_ln1, vb[i].s_att, vb[i].s_ln2, vb[i].s_ff = (sc(av[4 * i + j]) for j in range(4))
        L = v.enc_n_layers
        self._f8_vit_blocks = vb
        self._f8_vit = _lib.MdVitF8(C.cast(vb, C.POINTER(_lib.MdVitBlockF8)), q8(self.proj_fc1), q8(self.proj_fc2),
                                   sc(av[4 * L]), sc(av[4 * L + 1]), None)
        tb = (_lib.MdTextBlockF8 * t.n_layers)()
        for i, pk in enumerate(self._text_packed):
            if pk["qkv_fc1"] is None:
                raise ValueError("the fp8 prefill needs the fused qkv|fc1 packing (qkv_dim % 64 == 0)")
            tb[i].qkv_fc1, tb[i].proj, tb[i].fc2 = q8(pk["qkv_fc1"]), q8(pk["proj"]), q8(pk["fc2"])
            tb[i].s_ln, tb[i].s_att, tb[i].s_ff = (sc(at[3 * i + j]) for j in range(3))
        self._f8_text_blocks = tb
        self._f8_text = _lib.MdTextF8(C.cast(tb, C.POINTER(_lib.MdTextBlockF8)), None)
        self._f8_keep = keep
        self.vit.f8 = C.pointer(self._f8_vit)
        self.text.f8 = C.pointer(self._f8_text)
        return {"vit_amax": av, "text_amax": at, "margin": margin}

    def disable_f8(self) -> None:
        self.vit.f8 = C.POINTER(_lib.MdVitF8)()
        self.text.f8 = C.POINTER(_lib.MdTextF8)()
        self._f8_vit = self._f8_text = self._f8_keep = None

    def f8_enabled(self) -> bool:
        return bool(self.vit.f8) and bool(self.vit.f8.contents.blocks)

    def param_bytes(self) -> int:
        total = 0
        for p in self._keep:
            total += p.w.numel() * 2 + p.b.numel() * 2
        return total + self.wte.numel() * 2 + self.pos_emb.numel() * 2
