"""MoondreamModel: the reference's Python API on top of libmoondream_hip.so.

Public surface mirrors the reference class (reference:
moondream/torch/moondream.py:81-973): ``encode_image``, ``caption``, ``query``,
``detect``, ``point``, ``load_encoded_image``, ``compile`` and the four seam
methods ``_vis_enc / _vis_proj / _prefill / _decode_one_tok``
(moondream.py:168-192) -- here bound to hand-written gfx950 kernels through the
C ABI instead of ATen ops.  New: ``batch_generate`` / ``batch_caption`` /
``batch_query`` run B images in lockstep (the reference has no batching;
hf_moondream.py:99-103 is a sequential loop) and return what the sequential
path returns (bit for bit under ``set_strict_batch_invariance``; see
``batch_generate_ids`` for the default mode's contract).

PyTorch is used for device memory, streams and (optionally) hipGraph capture
only.  There is no eager / CPU fallback: without the built library or a GPU the
constructor raises.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from dataclasses import dataclass
from typing import Any, Dict, Iterable, List, Literal, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from PIL import Image

from . import _lib
from .config import MoondreamConfig
from .image_crops import crop_count, overlap_crop_image, reconstruct_from_crops
from .integration import MASK_CAUSAL, MASK_PREFIX_LM, classify_attn_mask
from .lora import variant_state_dict
from .weights import PackedLora, PackedModel

BF16 = torch.bfloat16

DEFAULT_MAX_TOKENS = 768
DEFAULT_TEMPERATURE = 0.5
DEFAULT_TOP_P = 0.3
DEFAULT_MAX_OBJECTS = 50

SpatialRefs = List[Union[Tuple[float, float], Tuple[float, float, float, float]]]


@dataclass(frozen=True)
class EncodedImage:
    """reference: moondream.py:56-59."""

    pos: int
    caches: List[Tuple[torch.Tensor, torch.Tensor]]


class IdTokenizer:
    """Stand-in used when the real vocabulary (``moondream/starmie-v1``) cannot be
    loaded offline: text is a space-separated list of token ids."""

    class _Enc:
        def __init__(self, ids):
            self.ids = ids

    def encode(self, s: str):
        try:
            return self._Enc([int(t) for t in s.split()])
        except ValueError:
            raise ValueError(
                "IdTokenizer only understands space-separated token ids; load the real vocabulary "
                "(tokenizers.Tokenizer 'moondream/starmie-v1') and pass it as tokenizer= for text prompts"
            ) from None

    def decode(self, ids: Iterable[int]) -> str:
        return "".join(f"{int(i)} " for i in ids)


def _load_tokenizer():
    """The reference loads ``moondream/starmie-v1`` from the Hub (moondream.py:89).  Offline that
    fails; the id-echo stand-in is then used, LOUDLY: with it, prompts and outputs are token ids."""
    try:
        from tokenizers import Tokenizer

        return Tokenizer.from_pretrained("moondream/starmie-v1")
    except Exception as e:  # no network / no cache
        import warnings

        warnings.warn(
            f"could not load the 'moondream/starmie-v1' vocabulary ({type(e).__name__}); falling back to IdTokenizer: "
            "text prompts must be space-separated token ids and answers come back as ids.  Pass tokenizer= to avoid this.",
            RuntimeWarning, stacklevel=3,
        )
        return IdTokenizer()


def _is_cjk_char(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF) or (0x3400 <= cp <= 0x4DBF) or (0x2F800 <= cp <= 0x2FA1F)


class MoondreamModel:
    def __init__(
        self,
        config: MoondreamConfig,
        state_dict: Dict[str, torch.Tensor],
        device: Union[str, torch.device] = "cuda",
        dtype: torch.dtype = BF16,
        setup_caches: bool = True,
        tokenizer: Any = None,
        max_batch: int = 1,
        vit_chunk_crops: int = 128,
    ):
        if dtype != BF16:
            raise ValueError("the Moondream hot path is bf16-only (reference: vision.py:36)")
        self.config = config
        self.lib = _lib.load()  # raises when the HIP library has not been built
        self._device = torch.device(device)
        if self._device.type != "cuda":
            raise _lib.MoondreamHipError("MoondreamModel needs a GPU device; there is no CPU path")
        self.tokenizer = tokenizer if tokenizer is not None else _load_tokenizer()
        self.w = PackedModel(config, state_dict, self._device)
        # the checkpoint's own 4-bit weight stream for the decode regime (enable_int4_decode): exact but OPT-IN -- the in-register
        # dequantisation is VALU-bound today (0.82x the bf16 stream's decode phase at B = 64, profiles/r04_int4_weight_stream.txt)
        self.int4_decode = False
        self.vit_chunk_crops = int(vit_chunk_crops)
        self._arenas: Dict[int, torch.Tensor] = {}
        self._max_batch = 0
        self._kv_k = self._kv_v = None
        self._kv_k8 = self._kv_v8 = None   # fp8 mode: e4m3 copy of the cache for the decode steps
        self._kv8_scales = None            # (ctypes float[L] for K, for V) when that copy is in use
        self._graphs: Dict[Any, Any] = {}
        self.use_graphs = False
        self.collect_timing = False
        self.last_phase_ms: Dict[str, float] = {}
        self._region_tables = None
        self._variants: Dict[str, PackedLora] = {}
        # batch-1 greedy decode on the persistent single-sequence kernel (md_decode_step_b1): one launch per token for
        # all decoder blocks; False = the batched kernels at one row (bit-identical to a row of a batch)
        self.single_sequence_kernel = True
        self.strict_batch_invariance = False  # see set_strict_batch_invariance
        self._tile_policy = _lib.MD_TILE_BY_SHAPE  # of the call in flight: _select_kernels
        self._b1_sync = None  # barrier state of that kernel: zeroed once, then owned by it
        self._b1_used = False
        # batch_generate over raw images: image prefix + prompt in one decoder pass (False: the reference's two passes).
        # Token generation only: detect / point keep the two passes, so that a raw image and its EncodedImage give the
        # same bits there (the region heads' decisions have no planted margins to absorb a changed accumulation order)
        self.fused_prefill = True
        # encode byte-identical crops of an image once (images that fit one crop: global == local); off = what the reference does
        self.dedup_identical_crops = False
        if setup_caches:
            self._setup_caches(max_batch)

    # ------------------------------------------------------------------ infra
    @property
    def device(self) -> torch.device:
        return self._device

    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)

    def _h2d(self, t: torch.Tensor) -> torch.Tensor:
        """Small host tensor -> device WITHOUT stalling the host: a copy from pageable memory blocks the calling thread until the
        stream has reached it (on the pipelined engine's streams: until the whole encode in front of it has run); from pinned
        memory it is queued like a kernel."""
        if t.device.type != "cpu":
            return t.to(self._device)
        with torch.inference_mode(False):
            pinned = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        pinned.copy_(t)
        return pinned.to(self._device, non_blocking=True)

    def _setup_caches(self, max_batch: Optional[int] = None):
        """Zeroed KV slabs [L][B][H_kv][ctx][hd] (reference: moondream.py:62-72,152-162)."""
        t = self.config.text
        b = max(1, int(max_batch or self._max_batch or 1))
        old_k, old_v = self._kv_k, self._kv_v
        if old_k is not None:
            # growing: work queued on the pipelined engine's streams may still use the old slabs,
            # arenas and captured graphs
            torch.cuda.synchronize(self._device)
        # ordinary (non-inference) tensors even when a generate call grows the slabs from inside
        # torch.inference_mode(): load_encoded_image updates them in place from user code
        with torch.inference_mode(False):
            self._kv_k = torch.zeros(t.n_layers, b, t.n_kv_heads, t.max_context, t.head_dim, dtype=BF16, device=self._device)
            self._kv_v = torch.zeros_like(self._kv_k)
            old_k8, old_v8 = self._kv_k8, self._kv_v8
            if self._kv8_scales is not None:  # fp8 mode: the e4m3 copy the decode steps read
                self._kv_k8 = torch.zeros(self._kv_k.shape, dtype=torch.uint8, device=self._device)
                self._kv_v8 = torch.zeros_like(self._kv_k8)
            if old_k is not None:  # slots loaded earlier (load_encoded_image) survive the growth
                keep = min(b, old_k.shape[1])
                self._kv_k[:, :keep] = old_k[:, :keep]
                self._kv_v[:, :keep] = old_v[:, :keep]
                if old_k8 is not None and self._kv_k8 is not None:
                    self._kv_k8[:, :keep] = old_k8[:, :keep]
                    self._kv_v8[:, :keep] = old_v8[:, :keep]
        self._max_batch = b
        self._graphs.clear()

    def _ensure_batch(self, b: int):
        if self._kv_k is None or b > self._max_batch:
            self._setup_caches(b)

    def _kv_struct(self, slot0: int = 0) -> _lib.MdKvCache:
        t = self.config.text
        bs = t.n_kv_heads * t.max_context * t.head_dim
        off = slot0 * bs * 2
        kv = _lib.MdKvCache(
            self._kv_k.data_ptr() + off, self._kv_v.data_ptr() + off, self._max_batch * bs, bs, t.max_context
        )
        if self._kv8_scales is not None and self._kv_k8 is not None:
            off8 = slot0 * bs
            kv.k8, kv.v8 = self._kv_k8.data_ptr() + off8, self._kv_v8.data_ptr() + off8
            kv.k_scale = C.cast(self._kv8_scales[0], C.c_void_p)
            kv.v_scale = C.cast(self._kv8_scales[1], C.c_void_p)
        return kv

    def _workspace(self, nbytes: int, which: int = 0) -> torch.Tensor:
        """Caller-owned arenas for the C ABI.  0: encode-side stages, 1: lm_head of the
        prompt prefill, 2: the decode loop (its own arena so that a decode running on
        the decode stream never shares scratch with an encode on the encode stream)."""
        ws = self._arenas.get(which)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(int(nbytes * 1.05) + 4096, dtype=torch.uint8, device=self._device)
            self._arenas[which] = ws
            self._graphs.clear()
        return ws

    @classmethod
    def from_pretrained(cls, weights_file: str, config: Optional[MoondreamConfig] = None, **kwargs):
        """Build the model from a checkpoint file (``.safetensors`` / ``.pt``; both key layouts the
        reference accepts, weights.py:112-171).  ``config`` defaults to the 2B configuration."""
        from .config import get_config
        from .weights import load_state_dict_file

        return cls(config or get_config("2b"), load_state_dict_file(weights_file), **kwargs)

    def register_variant(self, variant_id: str, lora: dict):
        """Make a LoRA variant available to ``settings={"variant": variant_id}`` without the network: ``lora`` is the
        nested dict the reference's ``variant_state_dict`` returns (lora.py:54-79)."""
        self._variants[variant_id] = PackedLora(self.config, lora, self._device)

    def _lora(self, settings: Optional[dict]) -> Optional[PackedLora]:
        """The packed variant named by ``settings["variant"]`` (reference: moondream.py:241-245,455-459), or None."""
        vid = (settings or {}).get("variant")
        if vid is None:
            return None
        if vid not in self._variants:
            self._variants[vid] = PackedLora(self.config, variant_state_dict(vid, device="cpu"), self._device)
        return self._variants[vid]

    def enable_fp8_decode(self, on: bool = True):
        """Opt-in numerical mode (BASELINE configs[4]): every decoder launch of <= 64 rows streams FP8 e4m3fn copies of
        the decoder weights -- half the bytes of the bandwidth-bound decode step -- with bf16 activations and fp32
        accumulation.  That is every decode step, and also any SHORT prefill (a text-only prompt, the prompt pass of the
        two-pass path at batch 1) and the lm_head of a prompt prefill of <= 64 sequences; the 730-row image prefill, the
        vision path and the KV cache stay bf16.  Outputs are judged by tolerance against
        the bf16 path (tests/test_model_gpu.py), not bit parity; off by default."""
        if on:
            if getattr(self.w, "_fp8_is_int4", False):
                self.w.disable_fp8_decode()  # one weight stream at a time: the e4m3 copy replaces the checkpoint's int4 stream
            self.w.enable_fp8_decode()
        else:
            self.w.disable_fp8_decode()
            if self.int4_decode and self.w.has_int4_source():
                self.w.enable_int4_decode()
        self._graphs.clear()  # captured decode steps baked the other launches in

    def enable_int4_decode(self, on: bool = True):
        """A checkpoint that stores the decoder blocks as the reference's QuantizedLinear triples (layers.py:47-109) carries
        its own 4-bit weight stream: decode launches (<= 64 rows) read the nibbles and rebuild the bf16 weights in registers
        with the reference's dequantisation arithmetic -- the SAME weights as the bf16 copy the prefill multiplies with, at a
        quarter of the bytes of the bandwidth-bound step.  Not a numerical mode (the operands are bit-identical), but OPT-IN:
        the rebuild costs ~4 VALU instructions per weight and the first version of the kernel is VALU-bound (decode phase
        0.82x the bf16 stream's at B = 64; profiles/r04_int4_weight_stream.txt).  The single-sequence persistent kernel
        streams bf16 weights only, so a lone sequence then decodes on the batched kernels."""
        self.int4_decode = bool(on)
        self.w.disable_fp8_decode()
        if on:
            if not self.w.has_int4_source():
                raise ValueError("this checkpoint has no QuantizedLinear (int4) decoder blocks")
            self.w.enable_int4_decode()
        self._graphs.clear()

    def enable_fp8(self, calibration_images: Optional[Sequence[Image.Image]] = None, prompt: Optional[Sequence[int]] = None,
                   on: bool = True, decode_weights: bool = True, margin: float = 1.5, kv_cache: bool = True) -> Optional[dict]:
        """Opt-in FP8 mode of the whole hot path (BASELINE configs[4] "fp8 weights, CDNA4 fp8 MFMA"): every MFMA-bound
        linear of the ViT blocks, the projector and the decoder prefill runs on ``md_gemm_f8`` (e4m3 operands,
        v_mfma_f32_32x32x64_f8f6f4 at twice the bf16 matrix rate, fp32 accumulation) with per-channel weight scales and ONE
        static scale per quantised activation tensor; ``decode_weights`` also streams e4m3 weights in the decode steps
        (``enable_fp8_decode``).  Patch embedding, layer-norm statistics, attention, RoPE, the KV cache, the residual
        stream, lm_head at prefill and the region head stay bf16 / fp32.  ``kv_cache`` (needs ``decode_weights``, MHA,
        head_dim 64) additionally keeps an e4m3 COPY of the KV cache, one static scale per layer for K and for V, that the
        decode steps attend over instead of the bf16 slabs -- half the bytes of the step's dominant stream; prefill
        attention, ``EncodedImage`` snapshots and the reference-compatible views keep using the bf16 slabs.

        The activation scales are CALIBRATED: ``calibration_images`` (a handful is enough) are run through the bf16
        path once with range recording on (vision + image / prompt prefill), then the e4m3 weight copies are built.
        The reference has no fp8 path: outputs are judged by tolerance against the bf16 mode (tests/test_model_gpu.py),
        never by bit parity, and the mode is off by default.  Returns the recorded ranges."""
        self._graphs.clear()
        if not on:
            self.w.disable_f8()
            if decode_weights:
                self.enable_fp8_decode(False)
            torch.cuda.synchronize(self._device)
            self._kv8_scales = None
            self._kv_k8 = self._kv_v8 = None
            return None
        if not calibration_images:
            raise ValueError("enable_fp8 needs a few calibration images (PIL) to size the activation scales")
        # a previous call's e4m3 cache copy does not survive this one (kv_cache=False after kv_cache=True must really turn it off)
        self._kv8_scales = None
        self._kv_k8 = self._kv_v8 = None
        tpl = self.config.tokenizer.templates["caption"]
        prompt = list(prompt) if prompt is not None else list(tpl["normal"] if tpl else [self.config.tokenizer.bos_id])
        self.w.begin_f8_calibration()
        try:
            with torch.inference_mode():
                n = len(calibration_images)
                self._prepare_sequences(list(calibration_images), [prompt] * n, None, None, fuse=True)
                torch.cuda.synchronize(self._device)
                t = self.config.text
                p1 = t.prefix_attn + len(prompt)
                # (the e4m3 decode attention keeps a context's scores in LDS: contexts of at most 2048 positions)
                if kv_cache and decode_weights and t.n_heads == t.n_kv_heads and t.head_dim == 64 and t.max_context <= 2048:
                    # K / V ranges of the calibration batch per layer (rows 0 .. p1 - 1 of its slots: the prefill just written)
                    ka = self._kv_k[:, :n, :, :p1].float().abs().amax(dim=(1, 2, 3, 4)).cpu().tolist()
                    va = self._kv_v[:, :n, :, :p1].float().abs().amax(dim=(1, 2, 3, 4)).cpu().tolist()
                    sc = lambda a: (a * margin / 448.0) if a > 0 else 1.0
                    self._kv8_scales = ((C.c_float * t.n_layers)(*[sc(a) for a in ka]), (C.c_float * t.n_layers)(*[sc(a) for a in va]))
                    self._kv_k8 = torch.zeros(self._kv_k.shape, dtype=torch.uint8, device=self._device)
                    self._kv_v8 = torch.zeros_like(self._kv_k8)
                    # slots that already hold sequences (load_encoded_image before this call): their e4m3 rows are built now
                    kv = self._kv_struct(0)
                    _lib.check(self.lib.md_kv_quantize_f8(C.byref(kv), t.n_layers, int(self._kv_k.shape[1]), t.n_kv_heads, None, 0,
                                                          int(t.max_context), self._stream()), "md_kv_quantize_f8")
            info = self.w.finish_f8_calibration(margin)
            info["kv_cache_fp8"] = self._kv8_scales is not None
        except Exception:
            self.w.disable_f8()
            raise
        if decode_weights:
            self.enable_fp8_decode(True)
        return info

    def set_strict_batch_invariance(self, on: bool = True):
        """``on``: every sequence gets the same bits whatever the batch it travels in -- ``batch_generate_ids([a, b, ..])[i]``
        == ``batch_generate_ids([x])`` == ``caption(x)`` bit for bit -- by giving up the two B=1 / batch-level shortcuts whose
        accumulation order differs from the batched kernels': the persistent single-sequence decode kernel (fp32 matrix-vector
        products, per-slice softmax maxima) and the fused [image | prompt] prefill pass (``caption`` / ``query`` make the
        reference's two passes), and by keeping every launch of a short (<= 64-token) prompt pass at <= 64 rows, so that it
        takes the same split-K kernels whether its sequence travels alone or in a batch.  Off (the default) those two are used; outputs then agree with the strict mode within bf16
        accumulation-order noise, i.e. ids can part only at decisions whose top-1/top-2 margin is inside that noise
        (quantified on the 64 bench images by tests/test_model_gpu.py::test_batch_equals_sequential_unfiltered)."""
        self.single_sequence_kernel = not on
        self.fused_prefill = not on
        self.strict_batch_invariance = bool(on)
        self._select_kernels(1)
        self._graphs.clear()

    def _select_kernels(self, n_sequences: int):
        """The library's tile choice by row count would give a lone image (730 / 1458 rows) the small-shape tile configs and a
        batch the four-wave kernel, whose 16x16x32 MFMAs sum K in another association.  Every call is therefore made with the
        tile config PINNED to the layer shape (``tile_policy = MD_TILE_PINNED`` in THIS model's ``md_vit_model`` /
        ``md_text_model`` structs, read by the library per call: ABI 5; no process-wide state, so two models or threads in
        one process do not affect each other's bits) -- so that ``batch_generate_ids(B)[i] == batch_generate_ids([x_i])``
        holds bit for bit in the DEFAULT mode as long as the lone sequence runs on the batched kernels -- except the
        documented latency path: ONE sequence with ``single_sequence_kernel`` on (small-shape tiles at prefill, the persistent
        kernel at decode).  Called by every public entry point that can launch a GEMM of more than 64 rows."""
        pin = (self.strict_batch_invariance or not (n_sequences == 1 and self.single_sequence_kernel))
        self._tile_policy = _lib.MD_TILE_PINNED if pin else _lib.MD_TILE_BY_SHAPE
        self.w.vit.tile_policy = self._tile_policy
        self.w.text.tile_policy = self._tile_policy

    def compile(self):
        """The reference rebinds the seam to torch.compile'd functions here
        (moondream.py:194-204).  The seam is already native; ``compile`` turns on
        hipGraph replay of the device-resident decode step."""
        self.use_graphs = True

    # ------------------------------------------------------------------ seam
    def _vis_enc(self, x: torch.Tensor) -> torch.Tensor:
        """bf16 [N,3,378,378] -> bf16 [N,729,D_v].  reference: moondream.py:168-169."""
        v = self.config.vision
        assert x.dtype == BF16 and x.dim() == 4 and x.shape[1:] == (3, v.crop_size, v.crop_size)
        x = x.contiguous()
        self._select_kernels(1)  # the seam is the reference's one-image-at-a-time path
        return self._vit_run(x, _lib.MD_CROPS_BF16_CHW)

    def _vit_run(self, crops: torch.Tensor, kind: int) -> torch.Tensor:
        v = self.config.vision
        n = crops.shape[0]
        out = torch.empty(n, v.n_patches, v.enc_dim, dtype=BF16, device=self._device)
        chunk = max(1, self.vit_chunk_crops)
        need = self.lib.md_vit_workspace_bytes(C.byref(self.w.vit), min(n, chunk))
        ws = self._workspace(need)
        for i in range(0, n, chunk):
            m = min(chunk, n - i)
            _lib.check(
                self.lib.md_vit_encode(
                    C.byref(self.w.vit), crops[i : i + m].data_ptr(), kind, m, out[i : i + m].data_ptr(),
                    ws.data_ptr(), ws.numel(), self._stream(),
                ),
                "md_vit_encode",
            )
        return out

    def _vis_proj(self, g: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
        """global [729,D_v] + stitched grid [h,w,D_v] -> [729,D].  reference: moondream.py:171-172."""
        v = self.config.vision
        g, r = g.contiguous(), r.contiguous()
        self._select_kernels(1)
        out = torch.empty(v.n_patches, v.proj_out_dim, dtype=BF16, device=self._device)
        need = self.lib.md_vision_project_workspace_bytes(C.byref(self.w.vit), 1)
        ws = self._workspace(need)
        _lib.check(
            self.lib.md_vision_project_grid(
                C.byref(self.w.vit), g.data_ptr(), r.data_ptr(), r.shape[0], r.shape[1], out.data_ptr(),
                v.proj_out_dim, ws.data_ptr(), ws.numel(), self._stream(),
            ),
            "md_vision_project_grid",
        )
        return out

    def _causal_text_struct(self):
        """The decoder description with an empty bidirectional prefix: the plain causal
        mask the reference builds for a text-only query (moondream.py:571-575)."""
        # rebuilt on every call: a cached byte copy would keep a stale ``fp8`` pointer across
        # enable_fp8_decode / disable_fp8_decode (the library reads the struct on the host, during the call)
        st = type(self.w.text).from_buffer_copy(self.w.text)
        st.prefix_len = 0
        self._text_causal = st
        return st

    def _text_forward(self, x: torch.Tensor, pos0: Union[int, Sequence[int]], slot0: int = 0, causal: bool = False,
                      pos_dev: Optional[torch.Tensor] = None, lora: Optional[PackedLora] = None) -> torch.Tensor:
        """x [B,T,D] embeddings -> hidden [B,T,D]; K,V written at pos0[b]..pos0[b]+T-1.
        ``pos0`` is host data (one int for the whole batch or one per sequence): the slab has
        max_context slots per head and the kernels do not bounds-check, so the check is here
        (the reference fails at this point too: its index_put / mask indexing raises)."""
        b, t, d = x.shape
        if self.strict_batch_invariance and lora is None and pos_dev is None and 1 < t <= 64 and b * t > 64:
            # The library picks the split-K decode-regime kernels by the ROW COUNT of a launch (<= 64 rows), and split-K sums
            # fp32 partials in another association than the sequential-K tile kernels.  A short prompt pass would so take
            # different kernels alone (t rows) and in a batch (b x t rows).  Strict mode keeps every launch of a short pass
            # at <= 64 rows: groups of 64 // t sequences, each through the kernels a lone sequence gets.
            per = max(1, 64 // t)
            outs = []
            for i0 in range(0, b, per):
                p0 = pos0 if isinstance(pos0, int) else list(pos0)[i0 : i0 + per]
                outs.append(self._text_forward(x[i0 : i0 + per], p0, slot0 + i0, causal=causal))
            return torch.cat(outs, dim=0)
        hi = pos0 if isinstance(pos0, int) else max(int(p) for p in pos0)
        lo = pos0 if isinstance(pos0, int) else min(int(p) for p in pos0)
        if lo < 0 or hi + t > self.config.text.max_context:
            raise ValueError(
                f"positions [{lo}, {hi + t}) do not fit the {self.config.text.max_context}-slot context "
                "(image prefix + prompt + generated tokens)"
            )
        if pos_dev is not None:  # the same positions, already on the device (loops that advance them there)
            assert pos_dev.dtype == torch.int32 and pos_dev.numel() == b
            pos0 = pos_dev
        elif isinstance(pos0, int):
            pos0 = torch.full((b,), pos0, dtype=torch.int32, device=self._device)
        else:
            assert len(pos0) == b
            pos0 = self._h2d(torch.tensor([int(p) for p in pos0], dtype=torch.int32))
        text = self._causal_text_struct() if causal else self.w.text
        self._ensure_batch(slot0 + b)
        x = x.contiguous()
        hidden = torch.empty_like(x)
        kv = self._kv_struct(slot0)
        if lora is not None:  # LoRA side path: unfused kernels + low-rank pairs (text.py:31-32,55-56; layers.py:129-146)
            need = self.lib.md_text_lora_workspace_bytes(C.byref(text), b, t)
            ws = self._workspace(need)
            _lib.check(
                self.lib.md_text_forward_lora(
                    C.byref(text), lora.ptr(), x.data_ptr(), hidden.data_ptr(), b, t, pos0.data_ptr(), C.byref(kv),
                    ws.data_ptr(), ws.numel(), self._stream(),
                ),
                "md_text_forward_lora",
            )
            return hidden
        need = self.lib.md_text_workspace_bytes(C.byref(text), b, t)
        ws = self._workspace(need)
        _lib.check(
            self.lib.md_text_forward(
                C.byref(text), x.data_ptr(), hidden.data_ptr(), b, t, pos0.data_ptr(), C.byref(kv),
                ws.data_ptr(), ws.numel(), self._stream(),
            ),
            "md_text_forward",
        )
        return hidden

    def _lm_head(self, hidden: torch.Tensor) -> torch.Tensor:
        """hidden [B,T,D] -> logits of the last token [B,V].  reference: text.py:163-167."""
        b, t, d = hidden.shape
        logits = torch.empty(b, self.config.text.vocab_size, dtype=BF16, device=self._device)
        need = self.lib.md_lm_head_workspace_bytes(C.byref(self.w.text), b)
        ws = self._workspace(need, 1)
        _lib.check(
            self.lib.md_lm_head(
                C.byref(self.w.text), hidden.contiguous().data_ptr(), b, t, logits.data_ptr(), logits.shape[1],
                ws.data_ptr(), ws.numel(), self._stream(),
            ),
            "md_lm_head",
        )
        return logits

    def _seam_rule(self, attn_mask: Optional[torch.Tensor], pos_ids: torch.Tensor, t: int):
        """(first position, causal?) for the mask slice and positions the reference passes through the seam
        (moondream.py:304-309: a slice of the prefix-LM buffer; 571-575: of a plain tril for the text-only query;
        472-474,515: the decode row).  Any other mask, and non-consecutive positions, raise ValueError."""
        if int(pos_ids.numel()) != t:
            raise ValueError(f"pos_ids has {int(pos_ids.numel())} entries for {t} embedding row(s)")
        tc = self.config.text
        kind = classify_attn_mask(attn_mask, pos_ids, tc.prefix_attn, tc.max_context)
        pos0 = int(pos_ids.reshape(-1)[0])
        if attn_mask is None and pos0 < tc.prefix_attn:
            kind = MASK_PREFIX_LM  # None = the prefix-LM buffer's slice (moondream.py:303-304): same check as an explicit one
        if kind == MASK_PREFIX_LM and pos0 + t < tc.prefix_attn:
            # such a slice lets its rows see keys [pos0 + t, prefix) that this pass does not write; the library reads keys
            # [0, pos0 + t) only.  The reference always prefills the whole prefix in one pass (moondream.py:254-258).
            raise ValueError(f"a prefix-LM pass must reach the end of the {tc.prefix_attn}-position bidirectional prefix "
                             f"(got positions [{pos0}, {pos0 + t}))")
        return pos0, kind == MASK_CAUSAL

    def _prefill(self, x: torch.Tensor, attn_mask: Optional[torch.Tensor], pos_ids: torch.Tensor, lora=None):
        """reference: moondream.py:174-181.  x [1,T,D]; attn_mask bool [1,1,T,ctx] (or None = the prefix-LM buffer's
        slice); pos_ids int64 [T] consecutive.  The mask is CLASSIFIED, not applied: the prefix-LM slice and the plain
        causal slice (text-only query) select the library's two rules; anything else is a ValueError."""
        if x.dim() != 3 or x.shape[0] != 1:
            raise ValueError(f"x must be [1, T, D] (got {tuple(x.shape)})")
        pos0, causal = self._seam_rule(attn_mask, pos_ids, int(x.shape[1]))
        self._select_kernels(1)
        return self._text_forward(x.to(self._device), pos0, 0, causal=causal, lora=lora)

    def _decode_one_tok(self, x: torch.Tensor, attn_mask: Optional[torch.Tensor], pos_ids: torch.Tensor, lora=None):
        """reference: moondream.py:183-192.  x [1,1,D], attn_mask bool [1,1,ctx] with ones on [0, pos] ->
        (logits [1,V], hidden [1,1,D])."""
        hidden = self._prefill(x, attn_mask, pos_ids, lora)
        return self._lm_head(hidden), hidden

    # ------------------------------------------------------------ vision path
    def _embed(self, ids: torch.Tensor) -> torch.Tensor:
        """token ids [..] -> embeddings [.., D]  (reference: text.py:12-13)."""
        if ids.device.type == "cpu" and ids.numel():  # user-supplied prompt ids: the gather kernel does not range-check
            lo, hi = int(ids.min()), int(ids.max())
            if lo < 0 or hi >= self.config.text.vocab_size:
                raise ValueError(f"token id out of range [0, {self.config.text.vocab_size}): {lo if lo < 0 else hi}")
        flat = self._h2d(ids.reshape(-1).to(torch.int32).contiguous())
        d = self.config.text.dim
        out = torch.empty(flat.numel(), d, dtype=BF16, device=self._device)
        _lib.check(
            self.lib.md_embed_tokens(flat.data_ptr(), self.w.wte.data_ptr(), d, out.data_ptr(), d, flat.numel(), d, self._stream()),
            "md_embed_tokens",
        )
        return out.reshape(*ids.shape, d)

    def _crop(self, image: Image.Image):
        v = self.config.vision
        arr = np.array(image.convert("RGB"))
        oc = overlap_crop_image(arr, max_crops=v.max_crops, overlap_margin=v.overlap_margin)
        return oc["crops"], tuple(oc["tiling"])

    def _run_vision_encoder_batch(self, images: Sequence[Image.Image], mark=None, staged=None) -> torch.Tensor:
        """images -> [B,729,D] projected embeddings (reference: moondream.py:206-228, per image).

        Host tiling (PIL, reference image_crops.py:58-167) runs on a thread pool and is
        pipelined against the GPU: the ViT of image chunk k is enqueued (async) while
        the crops of chunk k+1 are still being cut.  The workers cut their crops STRAIGHT
        INTO the pinned staging buffer the H2D copy reads (the tiling, hence every image's
        crop count and offset, follows from its size alone): no concatenation pass over the
        crop bytes on the host."""
        v = self.config.vision
        n_img = len(images)
        if staged is None:
            # (a prefetched entry is taken only when it is going to be USED: popped and dropped it would leak its pinned buffers)
            pre = getattr(self, "_prefetched_crops", {}).pop(tuple(id(im) for im in images), None)
            staged = pre[1] if pre is not None else self._stage_crops(images)
            try:
                return self._run_vision_encoder_batch(images, mark, staged)
            except BaseException:
                self._drain_staged(staged)  # this call owns the staging it made / took over: nothing stays marked busy
                raise
        chunks, staged = staged
        cropped: List[Tuple[np.ndarray, Tuple[int, int]]] = []
        feat_parts = []
        # OWNERSHIP of the pinned staging buffers: ``staged`` is consumed IN PLACE -- a chunk leaves the list at the moment its
        # buffer is handed back (uploaded or released), so whoever holds the list (the pipelined engine's drain, on an early
        # exit or an exception in here) releases exactly the chunks that are still in it and no token twice
        for ci in range(len(staged)):
            host, token, futs = staged[0]
            part = [f.result() for f in futs]
            if mark is not None and ci == 0:
                mark("host_tiling")  # phase timing: the GPU has nothing of this batch to run before the first crops exist
            cropped.extend(part)
            if self.dedup_identical_crops:
                # An image that is no larger than one crop has tiling (1, 1) and its single local crop holds the same
                # pixels as the global crop (image_crops.py:124-167: both are the image resized to crop_size).  The
                # encoder maps equal inputs to equal outputs whatever else is in the launch, so such a crop is encoded
                # once and its features are used for both positions.  Opt-in: the reference encodes both.
                uniq, expand = [], []
                for c, tiling in part:
                    if tiling == (1, 1) and c.shape[0] == 2 and np.array_equal(c[0], c[1]):
                        expand += [len(uniq), len(uniq)]
                        uniq.append(c[:1])
                    else:
                        expand += list(range(len(uniq), len(uniq) + c.shape[0]))
                        uniq.extend(c[k : k + 1] for k in range(c.shape[0]))
                # ownership (advisor, round 5): from here this block owns the chunk's buffer (``token``: popped from ``staged``,
                # so the drain no longer releases it) and the second buffer (``token2``) until the upload has taken it over;
                # whatever raises in between, both go back exactly once
                staged.pop(0)
                try:
                    host2, token2 = self._pinned_crops(len(uniq), (v.crop_size, v.crop_size, 3))
                    try:
                        np.concatenate(uniq, axis=0, out=host2)
                        dev_crops = self._upload_pinned(host2, token2)
                    except BaseException:
                        self._release_pinned(token2)
                        raise
                finally:
                    self._release_pinned(token)
                f = self._vit_run(dev_crops, _lib.MD_CROPS_U8_HWC)
                if len(expand) != dev_crops.shape[0]:
                    f = f[torch.tensor(expand, dtype=torch.int64, device=self._device)]
                feat_parts.append(f)
                continue
            dev_crops = self._upload_pinned(host, token)
            staged.pop(0)
            feat_parts.append(self._vit_run(dev_crops, _lib.MD_CROPS_U8_HWC))
        feats = feat_parts[0] if len(feat_parts) == 1 else torch.cat(feat_parts, dim=0)  # [sum crops, 729, Dv]
        out = torch.empty(n_img, v.n_patches, v.proj_out_dim, dtype=BF16, device=self._device)
        # images with the same tiling are projected together
        offsets, off = [], 0
        for c, _ in cropped:
            offsets.append(off)
            off += c.shape[0]
        groups: Dict[Tuple[int, int], List[int]] = {}
        for i, (_, tiling) in enumerate(cropped):
            groups.setdefault(tiling, []).append(i)
        for tiling, idxs in groups.items():
            nc = 1 + tiling[0] * tiling[1]
            contiguous = idxs == list(range(idxs[0], idxs[0] + len(idxs))) and len(groups) == 1
            if contiguous:
                f = feats
            else:
                f = torch.cat([feats[offsets[i] : offsets[i] + nc] for i in idxs], dim=0)
            o = out if contiguous else torch.empty(len(idxs), v.n_patches, v.proj_out_dim, dtype=BF16, device=self._device)
            need = self.lib.md_vision_project_workspace_bytes(C.byref(self.w.vit), len(idxs))
            ws = self._workspace(need)
            _lib.check(
                self.lib.md_vision_project(
                    C.byref(self.w.vit), f.data_ptr(), len(idxs), tiling[0], tiling[1], v.overlap_margin,
                    o.data_ptr(), v.proj_out_dim, ws.data_ptr(), ws.numel(), self._stream(),
                ),
                "md_vision_project",
            )
            if not contiguous:
                out[torch.tensor(idxs, device=self._device)] = o
        return out

    def _stage_crops(self, images: Sequence[Image.Image], pool=None):
        """Queue the host tiling of a batch on the thread pool, chunk by chunk, each image cutting its crops into its
        slice of a pinned staging buffer.  Returns (chunks, staged) for ``_run_vision_encoder_batch``."""
        v = self.config.vision
        n_img = len(images)
        pool = pool or self._crop_pool()
        per_chunk = max(1, self.vit_chunk_crops // 2)  # images per ViT launch group (2 crops each at 378 x 378; more for large images)
        counts = [crop_count(im.size[1], im.size[0], v.overlap_margin, v.max_crops, (v.crop_size, v.crop_size), v.enc_patch_size)
                  for im in images]
        # chunks: consecutive images whose crops total <= vit_chunk_crops (at least one image)
        chunks, i0 = [], 0
        while i0 < n_img:
            i1, tot = i0, 0
            while i1 < n_img and (i1 == i0 or (tot + counts[i1][0] <= max(self.vit_chunk_crops, 1) and i1 - i0 < per_chunk * 8)):
                tot += counts[i1][0]
                i1 += 1
            chunks.append((i0, i1, tot))
            i0 = i1
        staged = []
        for (c0, c1, tot) in chunks:  # all host work is queued up front; the GPU side follows chunk by chunk
            host, token = self._pinned_crops(tot, (v.crop_size, v.crop_size, 3))
            futs, off = [], 0
            for i in range(c0, c1):
                n = counts[i][0]
                futs.append(pool.submit(self._crop_into, images[i], host[off : off + n]))
                off += n
            staged.append((host, token, futs))
        return chunks, staged

    def prefetch_crops(self, images: Sequence[Image.Image]) -> None:
        """Start the host tiling of a batch NOW (thread pool, pinned staging buffers) so that a later call that encodes
        exactly these image objects -- ``batch_detect``, ``batch_generate_ids``, ... -- finds its crops cut: a serving
        loop calls this for batch k+1 before it runs batch k, which hides the PIL resize of large images behind the
        GPU (the pipelined caption engine does the same by construction).  Results are identical."""
        if not hasattr(self, "_prefetched_crops"):
            self._prefetched_crops = {}
        images = list(images)  # (kept alive with the entry: the key is made of object identities)
        key = tuple(id(im) for im in images)
        if key in self._prefetched_crops:
            return
        while len(self._prefetched_crops) >= 2:  # the batch about to run + one ahead
            self.discard_prefetched_crops(next(iter(self._prefetched_crops)))
        # a FEW background workers: the step that is running needs the host for its own launches, and the tiling of one
        # batch (tens of ms of CPU per large image) has a whole step's GPU time to finish
        pool = getattr(self, "_prefetch_pool", None)
        if pool is None:
            from concurrent.futures import ThreadPoolExecutor

            pool = self._prefetch_pool = ThreadPoolExecutor(max_workers=max(1, int(self.prefetch_workers)))
        self._prefetched_crops[key] = (images, self._stage_crops(images, pool))

    def wait_prefetched_crops(self) -> None:
        """Block until every prefetched batch's crops are cut (tests, benchmarks: a steady-state serving loop never waits)."""
        for _, (_, staged) in getattr(self, "_prefetched_crops", {}).values():
            for _, _, futs in staged:
                for f in futs:
                    f.result()

    def discard_prefetched_crops(self, key=None) -> None:
        """Drop prefetched batches that were never encoded (waits for their workers, returns the pinned buffers)."""
        d = getattr(self, "_prefetched_crops", {})
        for k in ([key] if key is not None else list(d)):
            _, staged = d.pop(k)
            self._drain_staged(staged)

    def _drain_staged(self, staged) -> None:
        """Return the pinned buffers of staged chunks that were never uploaded (``_run_vision_encoder_batch`` removes a chunk
        from the list when it hands its buffer back, so what is left here is exactly what is still owned)."""
        lst = staged[1]
        while lst:
            host, token, futs = lst.pop(0)
            for f in futs:
                try:
                    f.result()  # the worker must be done with the buffer before it is handed out again
                except Exception:
                    pass  # (a failed crop was / will be reported by the consumer; here only the buffer matters)
            self._release_pinned(token)

    def _crop_into(self, image: Image.Image, out: np.ndarray):
        v = self.config.vision
        arr = np.asarray(image if image.mode == "RGB" else image.convert("RGB"))  # (convert() of an RGB image is a plain copy)
        oc = overlap_crop_image(arr, max_crops=v.max_crops, overlap_margin=v.overlap_margin, base_size=(v.crop_size, v.crop_size),
                                patch_size=v.enc_patch_size, out=out)
        return oc["crops"], tuple(oc["tiling"])

    # PINNED staging for the uint8 crops (a pageable source makes the runtime bounce the copy through its own small pinned
    # buffers, synchronously: ~2x the time, and the host cannot queue the ViT launches behind it).  A small set of buffers
    # rotates; a buffer is handed out again only after the copy that read it has completed (its event).
    def _pinned_crops(self, n_crops: int, crop_shape):
        shape = (int(n_crops),) + tuple(crop_shape)
        nbytes = int(np.prod(shape))
        ring = getattr(self, "_pinned", None)
        if ring is None:
            ring = self._pinned = {"bufs": [], "events": [], "busy": []}
        pick = None
        for i, (buf, busy) in enumerate(zip(ring["bufs"], ring["busy"])):
            if not busy and buf.numel() >= nbytes:
                pick = i
                break
        if pick is None:
            with torch.inference_mode(False):
                ring["bufs"].append(torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, pin_memory=True))
            ring["events"].append(None)
            ring["busy"].append(False)
            pick = len(ring["bufs"]) - 1
        if ring["events"][pick] is not None:
            ring["events"][pick].synchronize()
            ring["events"][pick] = None
        ring["busy"][pick] = True
        host = ring["bufs"][pick][:nbytes].numpy().reshape(shape)
        return host, pick

    def _upload_pinned(self, host: np.ndarray, token: int) -> torch.Tensor:
        ring = self._pinned
        nbytes = int(host.size)
        dev = ring["bufs"][token][:nbytes].view(host.shape).to(self._device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self._device))
        ring["events"][token] = ev
        ring["busy"][token] = False  # reusable once the event has passed (checked when it is handed out again)
        return dev

    def _release_pinned(self, token: int):
        self._pinned["busy"][token] = False

    prefetch_workers = 4

    def _crop_pool(self):
        pool = getattr(self, "_pool", None)
        if pool is None:
            import os
            from concurrent.futures import ThreadPoolExecutor

            try:
                n = len(os.sched_getaffinity(0))
            except AttributeError:
                n = os.cpu_count() or 1
            # one process per GPU: the ranks of a node share its cores (8 ranks x 32 workers + prefetch workers would
            # oversubscribe a 256-thread host)
            ranks_here = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1))
            total = os.cpu_count() or n
            if ranks_here > 1 and n * ranks_here <= total + ranks_here:
                share = n          # the affinity mask already IS this rank's share (dist.bind_rank_cpus): do not divide twice
            else:
                share = n // ranks_here
            pool = self._pool = ThreadPoolExecutor(max_workers=max(1, min(32, share - 2 if ranks_here > 1 else n)))
        return pool

    def _run_vision_encoder(self, image: Image.Image) -> torch.Tensor:
        self._select_kernels(1)
        return self._run_vision_encoder_batch([image])[0]

    def _prefill_images(self, img_emb: torch.Tensor, slot0: int = 0, lora: Optional[PackedLora] = None) -> int:
        """[B,729,D] -> image prefix in the KV slabs of slots [slot0, slot0+B); returns pos (730)."""
        b = img_emb.shape[0]
        bos = self._embed(torch.full((b, 1), self.config.tokenizer.bos_id, dtype=torch.int32))
        x = torch.cat([bos, img_emb], dim=1)
        self._text_forward(x, 0, slot0, lora=lora)
        return x.shape[1]

    def encode_image(self, image: Union[Image.Image, EncodedImage], settings: Optional[dict] = None) -> EncodedImage:
        """reference: moondream.py:230-268."""
        if isinstance(image, EncodedImage):
            return image
        if not isinstance(image, Image.Image):
            raise ValueError("image must be a PIL Image or EncodedImage")
        lora = self._lora(settings)  # the image prefix depends on the variant (moondream.py:241-257)
        with torch.inference_mode():
            self._ensure_batch(1)
            pos = self._prefill_images(self._run_vision_encoder(image)[None], 0, lora)
            caches = [
                (self._kv_k[l, 0:1, :, :pos, :].clone(), self._kv_v[l, 0:1, :, :pos, :].clone())
                for l in range(self.config.text.n_layers)
            ]
        return EncodedImage(pos=pos, caches=caches)

    def load_encoded_image(self, encoded_image: EncodedImage, slot: int = 0):
        """reference: moondream.py:620-623."""
        self._ensure_batch(slot + 1)
        with torch.inference_mode():
            for l, (k, v) in enumerate(encoded_image.caches):
                self._kv_k[l, slot : slot + 1, :, : k.size(2), :] = k
                self._kv_v[l, slot : slot + 1, :, : v.size(2), :] = v
            if self._kv8_scales is not None:  # fp8 mode: rebuild the e4m3 copy of the rows just loaded
                kv = self._kv_struct(slot)
                t = self.config.text
                _lib.check(self.lib.md_kv_quantize_f8(C.byref(kv), t.n_layers, 1, t.n_kv_heads, None, 0, int(encoded_image.pos), self._stream()),
                           "md_kv_quantize_f8")

    # --------------------------------------------------------------- sampling
    def _pick(self, logits: torch.Tensor, temperature: float, top_p: float, suppress_id: int = -1,
              generator: Optional[torch.Generator] = None, probs_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[B,V] -> int32 [B]   (reference: moondream.py:313-318,521-528).  Greedy: argmax, ties to the
        lowest id.  Otherwise temperature + top-p on the device (md_sample_top_p): the reference's
        softmax / _apply_top_p / multinomial semantics with one uniform per sequence from torch's
        generator (same distribution as torch.multinomial, different random stream)."""
        b, v = logits.shape
        nxt = torch.empty(b, dtype=torch.int32, device=self._device)
        if temperature == 0:
            _lib.check(
                self.lib.md_argmax_bf16(logits.data_ptr(), logits.stride(0), b, v, suppress_id, nxt.data_ptr(), self._stream()),
                "md_argmax_bf16",
            )
            return nxt
        u = torch.rand(b, device=self._device, dtype=torch.float32, generator=generator)
        _lib.check(
            self.lib.md_sample_top_p(
                logits.data_ptr(), logits.stride(0), b, v, suppress_id, float(temperature), float(top_p), u.data_ptr(),
                nxt.data_ptr(), probs_out.data_ptr() if probs_out is not None else None,
                probs_out.stride(0) if probs_out is not None else 0, self._stream(),
            ),
            "md_sample_top_p",
        )
        return nxt

    # ---------------------------------------------------------- batched engine
    def _prefill_prompts(self, prompts: Sequence[Sequence[int]], pos: int, slot0: int = 0, prompt_embs=None,
                         lora: Optional[PackedLora] = None):
        """Prefill B equal-length prompts at position ``pos``; returns (logits [B,V], hidden [B,T,D], pos+T).
        reference: moondream.py:280-321 (per sequence)."""
        b = len(prompts)
        ids = torch.tensor(prompts, dtype=torch.int32)
        x = self._embed(ids) if prompt_embs is None else prompt_embs
        hidden = self._text_forward(x, pos, slot0, lora=lora)
        return self._lm_head(hidden), hidden, pos + ids.shape[1]

    def _decode_greedy(self, first: torch.Tensor, pos: Union[int, Sequence[int]], max_tokens: int, suppress_id: int,
                       slot0: int = 0, eos_id: Optional[int] = None, check_every: int = 16,
                       lora: Optional[PackedLora] = None, allow_b1: bool = True, temperature: float = 0.0, top_p: float = 0.0,
                       generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """``_decode_greedy_impl`` plus the safety net of the persistent single-sequence kernel: its software grid barriers
        need every workgroup resident; if one times out (the GPU was shared with another persistent kernel) the kernel
        raises an error word and finishes with garbage.  That state is fully re-initialised by decoding the same tokens
        again (K / V rows at positions >= ``pos``, the id history, the position buffer), so the call is repeated on the
        batched kernels and the persistent kernel is switched off for this model."""
        hist = self._decode_greedy_impl(first, pos, max_tokens, suppress_id, slot0, eos_id, check_every, lora, allow_b1,
                                        temperature, top_p, generator)
        if self._b1_used:
            torch.cuda.current_stream(self._device).synchronize()
            if int(self._b1_sync[64 * 11]) != 0:
                import warnings

                with torch.inference_mode():
                    self._b1_sync[64 * 11] = 0
                self.single_sequence_kernel = False
                self._graphs.clear()
                warnings.warn("md_decode_step_b1: a grid barrier timed out (GPU shared with another persistent kernel?); "
                              "repeating the decode on the batched kernels and disabling the single-sequence kernel", RuntimeWarning)
                hist = self._decode_greedy_impl(first, pos, max_tokens, suppress_id, slot0, eos_id, check_every, lora, False)
        return hist

    def _decode_greedy_impl(self, first: torch.Tensor, pos: Union[int, Sequence[int]], max_tokens: int, suppress_id: int,
                            slot0: int = 0, eos_id: Optional[int] = None, check_every: int = 16,
                            lora: Optional[PackedLora] = None, allow_b1: bool = True, temperature: float = 0.0,
                            top_p: float = 0.0, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """Device-resident decode loop: returns int32 [steps+1, B] (row 0 = ``first``).  ``temperature`` 0: greedy.  Otherwise every
        step draws each sequence's token with the reference's rule (moondream.py:521-528: softmax(logits / T), _apply_top_p,
        multinomial) from the step's own logits on the device (md_sample_top_p), with one uniform per (step, sequence) taken
        UP FRONT from torch's generator -- no host round trip per token, replayable from a hipGraph.
        reference: the generator of moondream.py:471-530 without its per-token host sync.
        ``pos`` is the position of the next token, one int or one per sequence (sequences whose
        prompts differ in length decode in the same lockstep batch).
        With ``compile()`` the steps are replayed from a captured hipGraph in chunks."""
        b = first.shape[0]
        t = self.config.text
        self._ensure_batch(slot0 + b)
        pos_list = [int(pos)] * b if isinstance(pos, int) else [int(p) for p in pos]
        assert len(pos_list) == b
        max_tokens = max(0, min(max_tokens, t.max_context - 1 - max(pos_list)))
        hist = torch.zeros(max_tokens + 1, b, dtype=torch.int32, device=self._device)
        hist[0] = first
        self._b1_used = False
        if max_tokens == 0:
            return hist
        logits = self._decode_logits(b)
        need = self.lib.md_decode_workspace_bytes(C.byref(self.w.text), b)
        ws = self._workspace(need, 2)
        kv = self._kv_struct(slot0)
        pos_base = self._h2d(torch.tensor(pos_list, dtype=torch.int32))
        sample = temperature != 0
        uniforms = (torch.rand(max_tokens, b, device=self._device, dtype=torch.float32, generator=generator) if sample else None)

        # one sequence, greedy, no side path: the whole step as ONE persistent launch (csrc/decode_b1.hip) when the library
        # says this model / cache / device fits its static limits and its grid can be co-resident; anything else decodes on
        # the batched kernels.  Never from the pipelined engine (allow_b1 = False): a second stream's persistent GEMMs
        # could keep workgroups of the grid off the chip and its software barriers would time out.
        b1 = (b == 1 and allow_b1 and not sample and self.single_sequence_kernel and lora is None and not bool(self.w.text.fp8)
              and bool(self.lib.md_decode_step_b1_supported(C.byref(self.w.text), C.byref(kv))))
        self._b1_used = b1
        if b1:
            if self._b1_sync is None:
                self._b1_sync = torch.zeros(4096, dtype=torch.int32, device=self._device)
            need = max(need, self.lib.md_decode_step_b1_workspace_bytes(C.byref(self.w.text)))
            ws = self._workspace(need, 2)

        def one_step(tok_in, tok_out, pos_buf, u_row=None):
            if b1:
                _lib.check(
                    self.lib.md_decode_step_b1(
                        C.byref(self.w.text), tok_in.data_ptr(), tok_out.data_ptr(), pos_buf.data_ptr(), C.byref(kv),
                        suppress_id, logits.data_ptr(), t.vocab_size, ws.data_ptr(), ws.numel(), self._b1_sync.data_ptr(),
                        self._stream(),
                    ),
                    "md_decode_step_b1",
                )
                return
            _lib.check(
                self.lib.md_decode_step(
                    C.byref(self.w.text), tok_in.data_ptr(), tok_out.data_ptr(), pos_buf.data_ptr(), b, C.byref(kv),
                    suppress_id, logits.data_ptr(), t.vocab_size, ws.data_ptr(), ws.numel(), self._stream(),
                ),
                "md_decode_step",
            )
            if sample:  # the step left its logits [B, V] in ``logits``: draw from them instead of the argmax it wrote
                _lib.check(
                    self.lib.md_sample_top_p(
                        logits.data_ptr(), logits.stride(0), b, t.vocab_size, suppress_id, float(temperature), float(top_p),
                        u_row.data_ptr(), tok_out.data_ptr(), None, 0, self._stream(),
                    ),
                    "md_sample_top_p",
                )

        def all_done(upto):
            return eos_id is not None and bool((hist[: upto + 1] == eos_id).any(dim=0).all())

        steps = 0
        if lora is not None:
            # LoRA variant: the fused device-resident step has no side path; same loop from its pieces
            # (embed -> decoder with the low-rank pairs -> lm_head -> suppress + argmax), still without a host sync per token
            pos_t, pos_h = pos_base.clone(), list(pos_list)
            while steps < max_tokens:
                emb = self._embed(hist[steps].reshape(b, 1))
                h = self._text_forward(emb, pos_h, slot0, pos_dev=pos_t, lora=lora)
                if sample:
                    hist[steps + 1] = self._pick(self._lm_head(h), temperature, top_p, suppress_id, generator)
                else:
                    hist[steps + 1] = self._pick(self._lm_head(h), 0.0, 0.0, suppress_id)
                pos_t.add_(1)
                pos_h = [p + 1 for p in pos_h]
                steps += 1
                if check_every and steps % check_every == 0 and all_done(steps):
                    break
            return hist[: steps + 1]
        if not self.use_graphs:
            pos_t = pos_base.clone()
            while steps < max_tokens:
                one_step(hist[steps], hist[steps + 1], pos_t, uniforms[steps] if sample else None)
                steps += 1
                if check_every and steps % check_every == 0 and all_done(steps):
                    break
            return hist[: steps + 1]

        # ---- hipGraph path: one graph = `chunk` consecutive steps over fixed buffers
        chunk = max(1, min(check_every or 16, max_tokens))
        while steps < max_tokens:
            n = min(chunk, max_tokens - steps)
            # a graph is replayed only on the stream (context) it was captured for: the pipelined engine's
            # decode stream and the default stream each keep their own captures
            key = ("decode", b, slot0, n, suppress_id, ws.data_ptr(), self._kv_k.data_ptr(), logits.data_ptr(),
                   torch.cuda.current_stream(self._device).cuda_stream, b1, float(temperature), float(top_p))
            entry = self._graphs.get(key)
            if entry is None:
                buf = torch.zeros(n + 1, b, dtype=torch.int32, device=self._device)
                pos_buf = torch.zeros(b, dtype=torch.int32, device=self._device)
                u_buf = torch.zeros(n, b, dtype=torch.float32, device=self._device)
                # eager warm-up on scratch state is not possible (KV side effects), so the
                # first chunk of a new shape runs eagerly and the graph is captured afterwards
                buf[0] = hist[steps]
                pos_buf.copy_(pos_base + steps)
                if sample:
                    u_buf.copy_(uniforms[steps : steps + n])
                for i in range(n):
                    one_step(buf[i], buf[i + 1], pos_buf, u_buf[i])
                hist[steps + 1 : steps + n + 1] = buf[1:]
                torch.cuda.synchronize(self._device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for i in range(n):
                        one_step(buf[i], buf[i + 1], pos_buf, u_buf[i])
                # the capture did not execute; nothing to undo
                self._graphs[key] = (g, buf, pos_buf, u_buf)
            else:
                g, buf, pos_buf, u_buf = entry
                buf[0] = hist[steps]
                pos_buf.copy_(pos_base + steps)
                if sample:
                    u_buf.copy_(uniforms[steps : steps + n])
                g.replay()
                hist[steps + 1 : steps + n + 1] = buf[1:]
            steps += n
            if all_done(steps):
                break
        return hist[: steps + 1]

    def _check_b1_barriers(self):
        """Call after the tokens of a single-sequence decode have reached the host: the persistent kernel's grid barriers
        spin with a bound (they need every workgroup resident, which a second such kernel on another stream, or another
        process on the GPU, can prevent); a barrier that gave up raises the error word and the tokens are not valid."""
        if self._b1_sync is not None and int(self._b1_sync[64 * 11]) != 0:
            with torch.inference_mode():  # (the state tensor was created under inference mode)
                self._b1_sync[64 * 11] = 0
            raise _lib.MoondreamHipError(
                "md_decode_step_b1: a grid barrier timed out (the GPU was shared with another persistent kernel?); "
                "set single_sequence_kernel = False to decode on the batched kernels")

    def _decode_logits(self, b: int) -> torch.Tensor:
        buf = getattr(self, "_logits_buf", None)
        if buf is None or buf.shape[0] < b:
            buf = torch.empty(max(b, self._max_batch), self.config.text.vocab_size, dtype=BF16, device=self._device)
            self._logits_buf = buf
            self._graphs.clear()
        return buf

    @staticmethod
    def _truncate(seq: List[int], eos_id: Optional[int], max_tokens: int) -> List[int]:
        out = []
        for tok in seq:
            if (eos_id is not None and tok == eos_id) or len(out) >= max_tokens:
                break
            out.append(tok)
        return out

    def _prepare_sequences(self, images, prompts: Sequence[Sequence[int]], mark=None, lora: Optional[PackedLora] = None,
                           fuse: bool = False, logits_capture: Optional[torch.Tensor] = None, sampler=None):
        """Everything before the first generated token, for B (image, prompt-ids) pairs: sequences are
        placed in KV slots in order of prompt length (stable), so that every group of equal-length
        prompts occupies a contiguous slot range; raw images are encoded together and prefilled
        straight into their slots, EncodedImages are copied into theirs; one prompt prefill per
        distinct length.  Returns (order, first int32 [B], hidden_last [B, D], next_pos list) in slot
        order; ``order[slot]`` is the caller's index.  ``sampler`` = (temperature, top_p, suppress_id, generator) draws the first token
        as the reference's _prefill_prompt does (moondream.py:313-318; suppress_id -1 there); None = argmax.  Must run under
        torch.inference_mode()."""
        mark = mark or (lambda name: None)
        b = len(images)
        self._select_kernels(b)
        order = sorted(range(b), key=lambda i: len(prompts[i]))
        images = [images[i] for i in order]
        prompts = [list(prompts[i]) for i in order]
        if any(len(p) == 0 for p in prompts):
            raise ValueError("empty prompt")
        self._ensure_batch(b)
        raw_idx = [i for i, im in enumerate(images) if not isinstance(im, EncodedImage)]
        for i in raw_idx:
            if not isinstance(images[i], Image.Image):
                raise ValueError("image must be a PIL Image or EncodedImage")
        mark("start")
        pos = None
        if len(raw_idx) == b and fuse and self.fused_prefill:
            # Every image is raw: image prefix and prompt go through the decoder in ONE pass per group of equal-length
            # prompts ([bos | 729 image embeddings | prompt] at position 0).  The attention kernels evaluate the
            # reference's mask rule per element (bidirectional inside the first 730 positions, causal after), so this is
            # the same computation as the reference's two passes (moondream.py:228-262 then 280-321) up to accumulation
            # order -- one pass over the weights saved (fused_prefill = False keeps the two passes).
            img_emb = self._run_vision_encoder_batch(images, mark)
            mark("vision")
            bos = self._embed(torch.full((b, 1), self.config.tokenizer.bos_id, dtype=torch.int32))
            first = torch.empty(b, dtype=torch.int32, device=self._device)
            hidden_last = torch.empty(b, self.config.text.dim, dtype=BF16, device=self._device)
            next_pos = [0] * b
            g0 = 0
            while g0 < b:
                g1 = g0
                while g1 < b and len(prompts[g1]) == len(prompts[g0]):
                    g1 += 1
                pe = self._embed(torch.tensor(prompts[g0:g1], dtype=torch.int32))
                x = torch.cat([bos[g0:g1], img_emb[g0:g1], pe], dim=1)
                hidden = self._text_forward(x, 0, g0, lora=lora)
                lg = self._lm_head(hidden)
                if logits_capture is not None:
                    logits_capture[g0:g1] = lg
                first[g0:g1] = self._pick(lg, 0.0, 0.0) if sampler is None else self._pick(lg, *sampler)
                hidden_last[g0:g1] = hidden[:, -1, :]
                next_pos[g0:g1] = [x.shape[1]] * (g1 - g0)
                g0 = g1
            mark("image_prefill")
            mark("prompt_prefill")
            return order, first, hidden_last, next_pos
        if raw_idx:
            img_emb = self._run_vision_encoder_batch([images[i] for i in raw_idx], mark)
            mark("vision")
            # every run of consecutive raw images is prefilled straight into its own slots
            j = 0
            while j < len(raw_idx):
                k = j
                while k + 1 < len(raw_idx) and raw_idx[k + 1] == raw_idx[k] + 1:
                    k += 1
                pos = self._prefill_images(img_emb[j : k + 1], raw_idx[j], lora)
                j = k + 1
            mark("image_prefill")
        for i, im in enumerate(images):
            if isinstance(im, EncodedImage):
                if pos is not None and im.pos != pos:
                    raise ValueError("EncodedImage with a different prefix length than the rest of the batch")
                self.load_encoded_image(im, i)
                pos = im.pos
        first = torch.empty(b, dtype=torch.int32, device=self._device)
        hidden_last = torch.empty(b, self.config.text.dim, dtype=BF16, device=self._device)
        next_pos = [0] * b
        g0 = 0
        while g0 < b:  # one prefill per distinct prompt length
            g1 = g0
            while g1 < b and len(prompts[g1]) == len(prompts[g0]):
                g1 += 1
            logits, hidden, p1 = self._prefill_prompts(prompts[g0:g1], pos, g0, lora=lora)
            if logits_capture is not None:
                logits_capture[g0:g1] = logits
            first[g0:g1] = self._pick(logits, 0.0, 0.0) if sampler is None else self._pick(logits, *sampler)
            hidden_last[g0:g1] = hidden[:, -1, :]
            next_pos[g0:g1] = [p1] * (g1 - g0)
            g0 = g1
        mark("prompt_prefill")
        return order, first, hidden_last, next_pos

    def batch_generate_ids(
        self,
        images: Sequence[Union[Image.Image, EncodedImage]],
        prompts: Sequence[Sequence[int]],
        max_tokens: int = DEFAULT_MAX_TOKENS,
        eos_id: Optional[int] = None,
        ignore_eos: bool = False,
        variant: Optional[str] = None,
        temperature: float = 0.0,
        top_p: float = DEFAULT_TOP_P,
        generator: Optional[torch.Generator] = None,
    ) -> List[List[int]]:
        """Token ids for B (image, prompt-ids) pairs, decoded in lockstep; greedy by default.

        ``temperature`` > 0 (round 6): every sequence SAMPLES each of its tokens -- the first from the prompt pass's logits,
        the rest inside the lockstep loop -- with the reference's rule (softmax(logits / T), ``_apply_top_p``, multinomial:
        moondream.py:313-318, 521-528) on the device, one uniform per (step, sequence) from ``generator``; this is what the
        reference's loop of ``query`` / ``caption`` calls does at its DEFAULT settings (temperature 0.5, top_p 0.3:
        moondream.py:50-53; hf_moondream.py:99-103).  Sequences are statistically independent (own uniforms), not
        reproducible against torch.multinomial's stream.  The greedy contract below is the ``temperature = 0`` case.

        Defined as: element i is what the sequential path (encode_image ->
        load_encoded_image -> _generate_answer with temperature 0) returns for
        (images[i], prompts[i]).  Every BATCHED kernel's accumulation order is a
        function of the layer shape only, never of the number of sequences in
        the launch.  DEFAULT mode (round 4): ``batch_generate_ids(B)[i] ==
        batch_generate_ids([x_i])`` BIT FOR BIT whenever the lone sequence runs
        on the batched kernels (``single_sequence_kernel = False``): every launch
        of more than 64 rows takes the same four-wave GEMM whatever the batch
        (``_select_kernels``).  The one documented exception is the latency path
        of a lone sequence (``single_sequence_kernel`` on, the default: small
        tiles at prefill, the persistent kernel at decode), which agrees within
        bf16 accumulation-order noise.  ``caption`` / ``query`` on an
        ``EncodedImage`` make the reference's TWO decoder passes where a raw
        image takes one fused [image | prompt] pass here; equality with THAT
        path bit for bit is ``set_strict_batch_invariance(True)`` (two passes
        everywhere, priced in bench.py's ``strict_batch_invariance`` leg).
        Tested on the 64 unfiltered bench images at 2B.
        """
        b = len(images)
        assert b == len(prompts) and b > 0
        tk = self.config.tokenizer
        eos = tk.eos_id if eos_id is None else eos_id
        marks = []

        def mark(name):
            if self.collect_timing:
                e = torch.cuda.Event(enable_timing=True)
                e.record(torch.cuda.current_stream(self._device))
                marks.append((name, e))

        lora = self._lora({"variant": variant})
        with torch.inference_mode():
            sampler = None if temperature == 0 else (float(temperature), float(top_p), -1, generator)
            order, first, _, next_pos = self._prepare_sequences(list(images), prompts, mark, lora, fuse=True, sampler=sampler)
            b = len(order)
            stop = None if ignore_eos else eos
            hist = self._decode_greedy(first, next_pos if len(set(next_pos)) > 1 else next_pos[0], max_tokens,
                                       tk.answer_id, 0, stop, lora=lora, temperature=float(temperature), top_p=float(top_p),
                                       generator=generator)
            mark("decode")
            cols = hist.t().tolist()
            if b == 1:
                self._check_b1_barriers()
            results: List[Optional[List[int]]] = [None] * b
            for slot, src in enumerate(order):
                results[src] = self._truncate(cols[slot], stop, max_tokens)
        if self.collect_timing and len(marks) > 1:
            torch.cuda.synchronize(self._device)
            self.last_phase_ms = {marks[i][0]: marks[i - 1][1].elapsed_time(marks[i][1]) for i in range(1, len(marks))}
        return results  # type: ignore[return-value]

    def teacher_forced_logits(self, images, prompts: Sequence[Sequence[int]], forced_ids, gather_idx) -> torch.Tensor:
        """Parity instrument (tests / bench.py): the logits of every greedy decision when each sequence is FORCED to follow
        ``forced_ids[i]`` (the reference's ids) instead of its own argmax, gathered at ``gather_idx[i][j]`` (the reference's
        top-k ids of decision j).  Decision 0 is the prompt prefill's, decision j the decode step that consumed
        forced_ids[i][j-1] -- the same launches as ``batch_generate_ids`` at B > 1 (fused or two-pass prefill as configured, the
        batched decode step ``md_decode_step``), so the error measured here is the error of the ids' own logits.  At B = 1
        with ``single_sequence_kernel`` on, generation uses the persistent single-sequence kernel and this method still the
        batched one: it certifies batches, not that kernel (which tests/test_model_gpu.py compares separately).  Returns float32 [B, T+1, k] on the CPU (``answer_id`` suppressed from decision 1 on, moondream.py:517)."""
        b = len(images)
        forced = torch.as_tensor(np.asarray(forced_ids), dtype=torch.int32)
        idx = torch.as_tensor(np.asarray(gather_idx), dtype=torch.int64)
        t_steps = forced.shape[1]
        assert forced.shape[0] == b and idx.shape[0] == b and idx.shape[1] >= t_steps + 1
        tk, t = self.config.tokenizer, self.config.text
        out = torch.empty(b, t_steps + 1, idx.shape[2], dtype=torch.float32)
        with torch.inference_mode():
            cap = torch.empty(b, t.vocab_size, dtype=BF16, device=self._device)
            order, _, _, next_pos = self._prepare_sequences(list(images), prompts, None, None, fuse=True, logits_capture=cap)
            src = torch.tensor(order, dtype=torch.int64)
            forced_s, idx_s = forced[src].to(self._device), idx[src].to(self._device)  # slot order
            vals = torch.empty(b, t_steps + 1, idx.shape[2], dtype=torch.float32, device=self._device)
            vals[:, 0] = torch.gather(cap.float(), 1, idx_s[:, 0])
            hist = torch.cat([forced_s.t().contiguous(), torch.zeros(1, b, dtype=torch.int32, device=self._device)], 0)
            scratch = torch.empty(b, dtype=torch.int32, device=self._device)
            logits = self._decode_logits(b)
            ws = self._workspace(self.lib.md_decode_workspace_bytes(C.byref(self.w.text), b), 2)
            kv = self._kv_struct(0)
            pos_t = torch.tensor([int(p) for p in next_pos], dtype=torch.int32, device=self._device)
            for j in range(t_steps):
                _lib.check(
                    self.lib.md_decode_step(
                        C.byref(self.w.text), hist[j].data_ptr(), scratch.data_ptr(), pos_t.data_ptr(), b, C.byref(kv),
                        tk.answer_id, logits.data_ptr(), t.vocab_size, ws.data_ptr(), ws.numel(), self._stream(),
                    ),
                    "md_decode_step",
                )
                vals[:, j + 1] = torch.gather(logits[:b].float(), 1, idx_s[:, j + 1])
            out[src] = vals.cpu()
        return out

    # ------------------------------------------------------ pipelined batches
    def _streams(self):
        st = getattr(self, "_pipe_streams", None)
        if st is None:
            prio = int(os.environ.get("MD_PIPE_DECODE_PRIORITY", "-1"))   # (A/B knob; -1 = high priority, the default)
            st = self._pipe_streams = (torch.cuda.Stream(device=self._device), torch.cuda.Stream(device=self._device, priority=prio))
        return st

    pipeline_streams = int(os.environ.get("MD_PIPE_STREAMS", "2"))  # 2: decode on its own (priority) stream; 1: encode and decode in order on one stream

    def batch_generate_ids_pipelined(self, batches, max_tokens: int = DEFAULT_MAX_TOKENS, ignore_eos: bool = False):
        """Generator over an iterable of (images, prompt-id lists) batches; yields each batch's greedy ids in order, one
        decode group late (a group = two consecutive batches of <= 64 sequences decoded as ONE lockstep of <= 128 -- round 6,
        ``pair_decode`` -- or one batch).  Two HIP streams: the MFMA-bound encode (vision + prefill) of batch k+1 on one, the HBM-bound lockstep
        decode of batch k on a second, higher-priority one, over disjoint KV-slab slot groups and disjoint workspaces.

          * The two streams' kernels DO overlap on the GPU (profiles/r04_pipelined_engine_streams_ab.txt: 50-86 % of a
            decode kernel's time has a tile GEMM or prefill attention running beside it; each side runs slower, the step
            as a whole is 3 % shorter than with every kernel in order on one stream: 260.5 vs 269 ms on the same box --
            ``pipeline_streams = 1`` / MD_PIPE_STREAMS=1 is that one-stream engine).
          * The host tiling of batch k+1 (PIL resize + crop cutting, ~26 ms at B = 64) is started on the thread pool as
            soon as batch k's crops are cut, i.e. under batch k's GPU time.
          * The ids of batch k travel to pinned host memory behind its last decode step and are collected after batch
            k+1 has been queued; nothing touches the default stream and no host array is copied from pageable memory
            (either makes the host wait for the device).

        Per batch the result is identical to ``batch_generate_ids`` (same kernels, same order)."""
        pending = []
        it = iter(batches)
        cur = next(it, None)
        staged = self._stage_crops(list(cur[0])) if cur is not None else None
        try:
            yield from self._pipelined_loop(it, cur, staged, pending, max_tokens, ignore_eos)
        finally:
            st = getattr(self, "_pipe_staged", None)  # a consumer that stops early: the tiling started ahead is drained, its buffers returned
            self._pipe_staged = None
            if st is not None:
                self._drain_staged(st)

    # Round 6: the decode of TWO consecutive batches runs as one lockstep decode (<= 128 sequences) when the library takes that
    # many rows in one pass over the weights (md_decode_step: 65 .. 128 rows on the 128 x 64 weight-streaming tile, same bits as
    # two passes of 64): 2.07 instead of 2.46 ms per token and 64 sequences at 2B.  Each batch is still encoded on its own
    # (launches of 64 images: what BASELINE's batch=64 names), results are still yielded per batch and in order, one PAIR late.
    pair_decode = os.environ.get("MD_PIPE_PAIR", "1") not in ("0", "")

    def _pipelined_loop(self, it, cur, staged, pending, max_tokens, ignore_eos):
        tk = self.config.tokenizer
        eos = tk.eos_id
        run_s, dec_s = self._streams()
        if self.pipeline_streams not in (2, 3):
            dec_s = run_s
        group = 0
        self._pipe_staged = staged
        while cur is not None:
            b = len(cur[0])
            pair = self.pair_decode and 2 * b <= 128 and not bool(self.w.text.fp8) and self._kv8_scales is None
            cap = 2 * b if pair else b            # slots of one group: two groups alternate (encode into one, decode from the other)
            self._ensure_batch(2 * cap)
            self._select_kernels(b)
            base = group * cap
            group ^= 1
            firsts, p1s, sizes = [], [], []
            with torch.inference_mode():
                run_s.wait_stream(torch.cuda.current_stream(self._device))
                while True:
                    images, prompts = cur
                    images = list(images)
                    nb = len(images)
                    assert nb == len(prompts) and nb > 0 and len({len(p) for p in prompts}) == 1
                    slot0 = base + sum(sizes)
                    nxt = next(it, None)
                    with torch.cuda.stream(run_s):
                        img_emb = self._run_vision_encoder_batch(images, staged=staged)
                        # this batch's crops are cut and queued for upload: the pool is free for the next batch's tiling
                        staged = self._pipe_staged = self._stage_crops(list(nxt[0])) if nxt is not None else None
                        if self.fused_prefill:  # [bos | image | prompt] in one decoder pass, as in _prepare_sequences
                            # assembled in a PREALLOCATED arena (one per batch shape; stream-ordered reuse on run_s: the previous
                            # pass has read it before these copies run) instead of a fresh torch.cat per step; the BOS column is
                            # written once
                            n_img, n_pr = img_emb.shape[1], len(prompts[0])
                            x = self._prefill_arena(nb, 1 + n_img + n_pr, tk.bos_id)
                            x[:, 1 : 1 + n_img].copy_(img_emb)
                            x[:, 1 + n_img :].copy_(self._embed(torch.tensor(prompts, dtype=torch.int32)))
                            logits = self._lm_head(self._text_forward(x, 0, slot0))
                            p1 = x.shape[1]
                        else:
                            pos = self._prefill_images(img_emb, slot0)
                            logits, _, p1 = self._prefill_prompts(prompts, pos, slot0)
                        firsts.append(self._pick(logits, 0.0, 0.0))
                    p1s.append(p1)
                    sizes.append(nb)
                    cur = nxt
                    if not pair or len(sizes) == 2 or cur is None or len(cur[0]) != b:
                        break
                with torch.cuda.stream(run_s):
                    first = firsts[0] if len(firsts) == 1 else torch.cat(firsts)
                    if dec_s is not run_s:
                        ev = torch.cuda.Event()
                        ev.record(run_s)
                with torch.cuda.stream(dec_s):
                    if dec_s is not run_s:
                        dec_s.wait_event(ev)
                        first.record_stream(dec_s)
                    pos1 = p1s[0] if len(set(p1s)) == 1 else [p for p, n in zip(p1s, sizes) for _ in range(n)]
                    hist = self._decode_greedy(first, pos1, max_tokens, tk.answer_id, base, None, check_every=16, allow_b1=False)
                    # ids -> PINNED host memory behind the last decode step (a ``.tolist()`` at collection time is a
                    # synchronous copy on the default stream: it waits for everything queued on the device)
                    if self.pipeline_streams == 3:  # measurement only: rounds 1-3's collection (synchronous copy on the default stream)
                        hist_host = hist
                    else:
                        hist_host = self._pinned_ids(tuple(hist.shape))   # recycled by _collect: no cudaHostAlloc per step
                        hist_host.copy_(hist, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(dec_s)
            pending.append((hist_host, done, sizes))
            if len(pending) > 1:
                yield from self._collect(pending.pop(0), None if ignore_eos else eos, max_tokens)
        while pending:
            yield from self._collect(pending.pop(0), None if ignore_eos else eos, max_tokens)

    def _prefill_arena(self, b: int, t: int, bos_id: int) -> torch.Tensor:
        """bf16 [b, t, D] buffer of the pipelined engine's fused prefill input, column 0 = the BOS embedding."""
        key = (b, t)
        arenas = self.__dict__.setdefault("_prefill_arenas", {})
        x = arenas.get(key)
        if x is None:
            if len(arenas) >= 4:   # a handful of batch shapes at most: do not hoard
                arenas.clear()
            x = arenas[key] = torch.empty(b, t, self.config.text.dim, dtype=BF16, device=self._device)
            x[:, :1].copy_(self._embed(torch.full((b, 1), bos_id, dtype=torch.int32)))
        return x

    def _pinned_ids(self, shape) -> torch.Tensor:
        """A pinned int32 host buffer for one step's ids, from a free list that ``_collect`` refills (a pinned allocation per
        step is a driver call and a page-locking of its own; review, round 5)."""
        free = self.__dict__.setdefault("_pinned_id_free", {}).setdefault(shape, [])
        if free:
            return free.pop()
        with torch.inference_mode(False):
            return torch.empty(shape, dtype=torch.int32, pin_memory=True)

    def _collect(self, item, eos, max_tokens):
        """One decoded group -> the id lists of its batches, in order (a generator: one item per batch)."""
        hist, done, sizes = item
        done.synchronize()
        cols = hist.t().tolist()
        if hist.device.type == "cpu" and hist.is_pinned():
            self.__dict__.setdefault("_pinned_id_free", {}).setdefault(tuple(hist.shape), []).append(hist)
        if sum(sizes) == 1:
            self._check_b1_barriers()
        i0 = 0
        for n in sizes:
            yield [self._truncate(cols[i], eos, max_tokens) for i in range(i0, i0 + n)]
            i0 += n

    @staticmethod
    def _sampling_kwargs(settings: Optional[dict]) -> dict:
        """``settings`` of the string API -> keyword arguments of ``batch_generate_ids``, with the REFERENCE'S defaults
        (moondream.py:50-53, read at :447-457): max_tokens 768, temperature 0.5, top_p 0.3.  ``settings["generator"]`` (an
        extension) seeds the draws; ``{"temperature": 0}`` is greedy."""
        st = settings or {}
        return dict(max_tokens=st.get("max_tokens", DEFAULT_MAX_TOKENS), variant=st.get("variant"),
                    temperature=st.get("temperature", DEFAULT_TEMPERATURE), top_p=st.get("top_p", DEFAULT_TOP_P),
                    generator=st.get("generator"))

    def batch_caption(self, images, length: str = "normal", settings: Optional[dict] = None) -> List[str]:
        tpl = self.config.tokenizer.templates["caption"]
        if tpl is None:
            raise NotImplementedError("Model does not support captioning.")
        if length not in tpl:
            raise ValueError(f"Model does not support caption length '{length}'.")
        ids = self.batch_generate_ids(images, [tpl[length]] * len(images), **self._sampling_kwargs(settings))
        return [self.tokenizer.decode(s) for s in ids]

    def batch_query(self, images, questions: Sequence[str], settings: Optional[dict] = None) -> List[str]:
        tpl = self.config.tokenizer.templates["query"]
        if tpl is None:
            raise NotImplementedError("Model does not support querying.")
        prompts = [
            list(tpl["prefix"]) + list(self.tokenizer.encode(q).ids) + list(tpl["suffix"]) + list(tpl["suffix"])
            for q in questions
        ]
        ids = self.batch_generate_ids(images, prompts, **self._sampling_kwargs(settings))
        return [self.tokenizer.decode(s) for s in ids]

    def batch_generate(self, images, prompts: Optional[Sequence[str]] = None, settings: Optional[dict] = None) -> List[str]:
        """BASELINE.json's ``batch_generate``: captions when ``prompts`` is None, else answers, with the reference's sampling
        settings (``settings``: max_tokens / temperature / top_p / variant; defaults 768 / 0.5 / 0.3 as moondream.py:50-53).
        At ``{"temperature": 0}`` element i == caption(images[i]) / query(images[i], prompts[i]): bit for bit under
        ``set_strict_batch_invariance``, within bf16 accumulation-order noise otherwise -- see ``batch_generate_ids``."""
        if prompts is None:
            return self.batch_caption(images, "normal", settings)
        return self.batch_query(images, prompts, settings)

    # ------------------------------------------------------- sequential API
    def _prefill_prompt(self, prompt_tokens: torch.Tensor, pos: int, temperature: float, top_p: float,
                        spatial_refs: Optional[SpatialRefs] = None, attn_mask=None, lora=None, causal: bool = False):
        """reference: moondream.py:280-321.  ``causal`` stands for the reference's
        ``attn_mask`` argument: the only non-default mask it ever passes is the plain
        lower-triangular one of a text-only query (moondream.py:571-575)."""
        self._select_kernels(1)  # a lone sequence's own tile policy, whatever an earlier (batch) call left in the structs
        with torch.inference_mode():
            ids = prompt_tokens.to(torch.int32)
            emb = self._embed(ids)
            if spatial_refs:
                enc = self.encode_spatial_refs(spatial_refs)
                emb[ids.to(self._device) == self.config.tokenizer.coord_id] = enc["coords"]
                if enc["sizes"] is not None:
                    emb[ids.to(self._device) == self.config.tokenizer.size_id] = enc["sizes"]
            hidden = self._text_forward(emb, pos, 0, causal=causal, lora=lora if isinstance(lora, PackedLora) else None)
            logits = self._lm_head(hidden)
            nxt = self._pick(logits, temperature, top_p)
        return logits, hidden, nxt.reshape(1, 1), pos + ids.shape[1]

    def _generate_answer(self, prompt_tokens: torch.Tensor, pos: int, settings: Optional[dict] = None,
                         spatial_refs: Optional[SpatialRefs] = None, eos_id: Optional[int] = None, attn_mask=None,
                         causal: bool = False):
        """Generator of text pieces.  reference: moondream.py:434-539."""
        settings = settings or {}
        max_tokens = settings.get("max_tokens", DEFAULT_MAX_TOKENS)
        temperature = settings.get("temperature", DEFAULT_TEMPERATURE)
        top_p = settings.get("top_p", DEFAULT_TOP_P)
        lora = self._lora(settings)
        eos = eos_id if eos_id is not None else self.config.tokenizer.eos_id
        # decode steps attend to keys [0, pos] only, which both masks allow: the mask matters for the prompt
        _, _, nxt, pos = self._prefill_prompt(prompt_tokens, pos, temperature, top_p, spatial_refs, attn_mask, lora=lora, causal=causal)

        def token_source():
            if temperature == 0 and lora is None:
                done = 0
                first = nxt.reshape(1).to(torch.int32)
                cur_pos = pos
                while done < max_tokens:
                    chunk = min(16, max_tokens - done)
                    hist = self._decode_greedy(first, cur_pos, chunk, self.config.tokenizer.answer_id, 0, None)
                    toks = hist[:, 0].tolist()
                    self._check_b1_barriers()
                    for tok in toks[:-1]:
                        yield tok
                    done += len(toks) - 1
                    cur_pos += len(toks) - 1
                    first = hist[-1]
                    if len(toks) - 1 < chunk:
                        break
                yield int(first[0])
            else:
                tok = nxt.reshape(1).to(torch.int32)
                cur_pos = pos
                while True:
                    yield int(tok[0])
                    if cur_pos >= self.config.text.max_context:  # context full: same bound as the greedy loop
                        return
                    with torch.inference_mode():
                        emb = self._embed(tok.reshape(1, 1))
                        hidden = self._text_forward(emb, cur_pos, 0, lora=lora)
                        logits = self._lm_head(hidden)
                        cur_pos += 1
                        tok = self._pick(logits, temperature, top_p, self.config.tokenizer.answer_id)  # moondream.py:517

        def generator():
            # streaming detokeniser: flush on newline, CJK, or up to the last space
            # (reference: moondream.py:477-537)
            cache: List[int] = []
            print_len = 0
            n = 0
            for tok in token_source():
                if tok == eos or n >= max_tokens:
                    break
                n += 1
                cache.append(tok)
                text = self.tokenizer.decode(cache)
                if text.endswith("\n"):
                    piece, cache, print_len = text[print_len:], [], 0
                    if piece:
                        yield piece
                elif len(text) > 0 and _is_cjk_char(ord(text[-1])):
                    piece = text[print_len:]
                    print_len += len(piece)
                    if piece:
                        yield piece
                else:
                    sp = text.rfind(" ", print_len)
                    if sp >= print_len:
                        piece = text[print_len : sp + 1]
                        print_len += len(piece)
                        if piece:
                            yield piece
            if cache:
                piece = self.tokenizer.decode(cache)[print_len:]
                if piece:
                    yield piece

        return generator()

    def caption(self, image, length: Literal["normal", "short", "long"] = "normal", stream: bool = False,
                settings: Optional[dict] = None):
        """reference: moondream.py:625-651."""
        tpl = self.config.tokenizer.templates["caption"]
        if tpl is None:
            raise NotImplementedError("Model does not support captioning.")
        if length not in tpl:
            raise ValueError(f"Model does not support caption length '{length}'.")
        enc = self.encode_image(image, settings)
        self.load_encoded_image(enc)
        prompt = torch.tensor([tpl[length]])
        gen = self._generate_answer(prompt, enc.pos, settings)
        return {"caption": gen} if stream else {"caption": "".join(list(gen))}

    def query(self, image=None, question: str = None, reasoning: bool = False,
              spatial_refs: Optional[SpatialRefs] = None, stream: bool = False, settings: Optional[dict] = None):
        """reference: moondream.py:541-618, including the ``reasoning`` branch (thinking token -> _generate_reasoning
        -> answer after a second suffix)."""
        tpl = self.config.tokenizer.templates["query"]
        if tpl is None:
            raise NotImplementedError("Model does not support querying.")
        if question is None:
            raise ValueError("question must be provided.")
        if spatial_refs and image is None:
            raise ValueError("spatial_refs can only be used with an image.")
        if image is not None:
            enc = self.encode_image(image, settings)
            self.load_encoded_image(enc)
            pos, head, causal = enc.pos, list(tpl["prefix"]), False
        else:
            # text only: BOS + prefix at position 0 under the plain causal mask (moondream.py:564-575);
            # stale K/V beyond the positions written here are never read (kv_len bounds every kernel)
            self._ensure_batch(1)
            pos, head, causal = 0, [self.config.tokenizer.bos_id] + list(tpl["prefix"]), True
        spatial = []
        if spatial_refs:
            tk = self.config.tokenizer
            for ref in spatial_refs:
                spatial.extend([tk.coord_id, tk.coord_id] if len(ref) == 2 else [tk.coord_id, tk.coord_id, tk.size_id])
        prompt = head + spatial + list(self.tokenizer.encode(question).ids) + list(tpl["suffix"])
        extra = {}
        if reasoning:
            # moondream.py:593-603: [.., suffix, thinking] -> reasoning text until answer_id, then the answer after [suffix]
            prompt = prompt + [self.config.tokenizer.thinking_id]
            pos, text, grounding = self._generate_reasoning(torch.tensor([prompt]), pos, settings, spatial_refs, causal=causal)
            extra = {"reasoning": {"text": text, "grounding": grounding}}
            gen = self._generate_answer(torch.tensor([list(tpl["suffix"])]), pos, settings, None, causal=causal)
        else:
            prompt = prompt + list(tpl["suffix"])
            gen = self._generate_answer(torch.tensor([prompt]), pos, settings, spatial_refs, causal=causal)
        return {**extra, "answer": gen} if stream else {**extra, "answer": "".join(list(gen))}

    def _generate_reasoning(self, prompt_tokens: torch.Tensor, pos: int, settings: Optional[dict] = None,
                            spatial_refs: Optional[SpatialRefs] = None, attn_mask=None, causal: bool = False):
        """reference: moondream.py:323-432.  Generates the reasoning text (stops at ``answer_id``; ``eos`` and
        ``size`` tokens suppressed), grounding every ``coord`` token through the region head: the token's
        coordinate is decoded from the hidden state that predicted it and fed back as the next embedding.
        Returns (pos, text, grounding).  One host decision per token, like the reference."""
        settings = settings or {}
        max_tokens = settings.get("max_tokens", DEFAULT_MAX_TOKENS)
        temperature = settings.get("temperature", DEFAULT_TEMPERATURE)
        top_p = settings.get("top_p", DEFAULT_TOP_P)
        lora = self._lora(settings)
        tk = self.config.tokenizer
        _, hidden, nxt, pos = self._prefill_prompt(prompt_tokens, pos, temperature, top_p, spatial_refs, attn_mask, lora=lora, causal=causal)
        last_hidden = hidden[:, -1:, :].reshape(1, -1)
        text_chunks: List[List[int]] = [[]]
        grounding_chunks: List[List[float]] = [[]]
        generated = 0
        bins = torch.zeros(1, 1, dtype=torch.int32, device=self._device)
        n_bins = self.config.region.coord_out_dim
        suppress = torch.tensor([tk.eos_id, tk.size_id], device=self._device)
        with torch.inference_mode():
            tok = int(nxt.reshape(-1)[0])
            while tok != tk.answer_id and generated < max_tokens and pos < self.config.text.max_context:
                if tok in (tk.start_ground_points_id, tk.end_ground_id):
                    text_chunks.append([])
                    grounding_chunks.append([])
                text_chunks[-1].append(tok)
                if tok == tk.coord_id:
                    # coordinate bin from the hidden state, value fed back through the coordinate encoder (device resident)
                    emb = self._region_pick_encode(last_hidden, "coord", bins)
                    grounding_chunks[-1].append((bins[0, 0].to(torch.int64) / n_bins).item())
                else:
                    emb = self._embed(torch.tensor([[tok]]))
                h = self._text_forward(emb.reshape(1, 1, -1), pos, 0, lora=lora)
                logits = self._lm_head(h)
                logits[:, suppress] = float("-inf")  # moondream.py:397-398
                pos += 1
                last_hidden = h.reshape(1, -1)
                tok = int(self._pick(logits, temperature, top_p)[0])
                generated += 1
        texts = [self.tokenizer.decode(c) for c in text_chunks]
        grounding, start = [], 0
        for t, g in zip(texts, grounding_chunks):
            if len(g) > 1:
                pts = [(g[i], g[i + 1]) for i in range(0, len(g) - (len(g) % 2), 2)]
                grounding.append({"start_idx": start, "end_idx": start + len(t), "points": pts})
            start += len(t)
        return pos, "".join(texts), grounding

    # ------------------------------------------------------------ region head
    # Device-resident and batched: the heads are decode-regime GEMMs over B rows, "argmax the bin,
    # turn it into a value, Fourier-encode it" is one kernel (md_region_pick_encode), the decoder step
    # is md_text_forward -- no host round trip per coordinate (the reference syncs 3-4 times per
    # object: moondream.py:669,718-721).  One D2H per object decides whether every sequence is done.
    def _region(self):
        if self.w.region is None:
            raise NotImplementedError("checkpoint has no region head")
        if self._region_tables is None:
            r, rc = self.w.region, self.config.region
            nb_c, nb_s = rc.coord_out_dim, rc.size_out_dim // 2
            # bin -> value exactly as the reference computes it (moondream.py:673-674: argmax / size(-1),
            # fp32; :699-701: 2^(bin/1023*10-10), fp32), then the dtype of the logits (bf16)
            coord = (torch.arange(nb_c) / nb_c).to(BF16)
            size = torch.pow(2.0, (torch.arange(nb_s).float() / 1023.0) * 10.0 - 10.0).to(BF16)
            need = 0
            for lin in (r["coord_dec_fc1"], r["coord_dec_fc2"], r["size_dec_fc1"], r["size_dec_fc2"], r["coord_encoder"], r["size_encoder"]):
                st = lin.struct()
                need = max(need, self.lib.md_gemm_workspace_bytes(C.byref(st), 64, 1))
            self._region_tables = {
                "coord": coord.to(self._device), "size": size.to(self._device),
                # decode-regime split-K scratch: tickets start zeroed, every launch leaves them zeroed
                "splitk": torch.zeros(max(need, 16), dtype=torch.uint8, device=self._device),
            }
        return self.w.region, self._region_tables

    def _lin(self, a: torch.Tensor, lin, epi: int = _lib.MD_EPI_BIAS) -> torch.Tensor:
        """rows [m, k] (bf16, contiguous) -> [m, n] through md_gemm_bf16."""
        m = a.shape[0]
        if a.shape[1] != lin.k_pad or not a.is_contiguous():
            ap = torch.zeros(m, lin.k_pad, dtype=BF16, device=self._device)
            ap[:, : lin.k] = a
            a = ap
        _, tabs = self._region()
        ws = tabs["splitk"]
        out = torch.empty(m, lin.n_pad, dtype=BF16, device=self._device)
        args = _lib.MdGemmArgs(a.data_ptr(), lin.k_pad, lin.struct(), out.data_ptr(), lin.n_pad, None, 0, 0, m, epi, 1, 0,
                               ws.data_ptr(), ws.numel(), self._tile_policy)
        _lib.check(self.lib.md_gemm_bf16(C.byref(args), self._stream()), "md_gemm_bf16")
        return out[:, : lin.n]

    def _gemm(self, a, lin, epi: int = _lib.MD_EPI_BIAS):
        return self._lin(a.to(self._device, BF16), lin, epi)

    def _mlp(self, x: torch.Tensor, fc1, fc2) -> torch.Tensor:
        return self._lin(self._lin(x, fc1, _lib.MD_EPI_GELU), fc2)

    def _fourier(self, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        """reference: region.py:12-29.  x bf16 [n, in_dim], w bf16 [in_dim, half] -> [n, 2*half]."""
        x = x.to(self._device, BF16).contiguous()
        n, in_dim = x.shape
        half = w.shape[1]
        out = torch.empty(n, 2 * half, dtype=BF16, device=self._device)
        _lib.check(
            self.lib.md_fourier_features(x.data_ptr(), x.stride(0), n, in_dim, w.data_ptr(), half, out.data_ptr(), out.stride(0), self._stream()),
            "md_fourier_features",
        )
        return out

    def encode_coordinate(self, coord: torch.Tensor) -> torch.Tensor:
        """reference: region.py:32-43."""
        r, _ = self._region()
        return self._lin(self._fourier(coord.reshape(-1, 1), r["coord_features"]), r["coord_encoder"])

    def decode_coordinate(self, hidden: torch.Tensor) -> torch.Tensor:
        """reference: region.py:46-57."""
        r, _ = self._region()
        return self._mlp(hidden.reshape(-1, hidden.shape[-1]).to(self._device, BF16), r["coord_dec_fc1"], r["coord_dec_fc2"])

    def encode_size(self, size: torch.Tensor) -> torch.Tensor:
        """reference: region.py:60-71."""
        r, _ = self._region()
        return self._lin(self._fourier(size.reshape(-1, 2), r["size_features"]), r["size_encoder"])

    def decode_size(self, hidden: torch.Tensor) -> torch.Tensor:
        """reference: region.py:74-93.  [1, D] -> [2, bins]; [B, D] -> [B, 2, bins]."""
        r, _ = self._region()
        h = hidden.reshape(-1, hidden.shape[-1]).to(self._device, BF16)
        out = self._mlp(h, r["size_dec_fc1"], r["size_dec_fc2"])
        return out.reshape(2, -1) if h.shape[0] == 1 else out.reshape(h.shape[0], 2, -1)

    def encode_spatial_refs(self, spatial_refs: SpatialRefs):
        """reference: region.py:96-136."""
        coords, sizes = [], []
        for ref in spatial_refs:
            if len(ref) == 2:
                coords += [ref[0], ref[1]]
            else:
                coords += [(ref[0] + ref[2]) / 2, (ref[1] + ref[3]) / 2]
                sizes.append([ref[2] - ref[0], ref[3] - ref[1]])
        c = self.encode_coordinate(torch.tensor(coords, dtype=BF16).view(-1, 1))
        s = self.encode_size(torch.tensor(sizes, dtype=BF16)) if sizes else None
        return {"coords": c, "sizes": s}

    def _region_pick_encode(self, hidden: torch.Tensor, which: str, bins_out: torch.Tensor) -> torch.Tensor:
        """hidden [B, D] -> next embedding [B, D]: head MLP -> per-sequence argmax bin(s) (written to
        ``bins_out`` [B, 1|2], int32) -> value -> Fourier features -> encoder.  All on the device.
        reference: moondream.py:672-677 (x), :682-687 (y), :693-713 (size)."""
        r, tabs = self._region()
        b = hidden.shape[0]
        if which == "coord":
            logits = self._mlp(hidden, r["coord_dec_fc1"], r["coord_dec_fc2"])
            groups, table, fw, enc = 1, tabs["coord"], r["coord_features"], r["coord_encoder"]
        else:
            logits = self._mlp(hidden, r["size_dec_fc1"], r["size_dec_fc2"])
            groups, table, fw, enc = 2, tabs["size"], r["size_features"], r["size_encoder"]
        n_bins = logits.shape[1] // groups
        half = fw.shape[1]
        feats = torch.empty(b, 2 * half, dtype=BF16, device=self._device)
        assert bins_out.dtype == torch.int32 and bins_out.stride(-1) == 1
        _lib.check(
            self.lib.md_region_pick_encode(
                logits.data_ptr(), logits.stride(0), b, groups, n_bins, table.data_ptr(), fw.data_ptr(), half,
                bins_out.data_ptr(), bins_out.stride(0), feats.data_ptr(), feats.stride(0), self._stream(),
            ),
            "md_region_pick_encode",
        )
        return self._lin(feats, enc)

    def _points_loop(self, hidden: torch.Tensor, first: torch.Tensor, pos: Sequence[int], slot0: int, include_size: bool,
                     max_objects: int, lora: Optional[PackedLora] = None, run_all: bool = False) -> List[List[dict]]:
        """The loop of moondream.py:653-733 for B sequences in lockstep.  hidden [B, D] = last prompt
        position, first int32 [B] = the token after the prompt, pos[b] = next position."""
        b = hidden.shape[0]
        eos = self.config.tokenizer.eos_id
        n_bins = self.config.region.coord_out_dim
        steps_per_obj = 3 if include_size else 2
        if max(pos) + steps_per_obj * max_objects > self.config.text.max_context:
            max_objects = max(0, (self.config.text.max_context - max(pos)) // steps_per_obj)
        bins = torch.zeros(max(1, max_objects), b, 4, dtype=torch.int32, device=self._device)
        toks = torch.full((max_objects + 1, b), eos, dtype=torch.int32, device=self._device)
        toks[0] = first
        pos_host = [int(p) for p in pos]
        pos_dev = torch.tensor(pos_host, dtype=torch.int32, device=self._device)
        hidden = hidden.reshape(b, -1).contiguous()

        def step(emb):
            nonlocal pos_host, hidden
            h = self._text_forward(emb.reshape(b, 1, -1), pos_host, slot0, pos_dev=pos_dev, lora=lora)
            pos_host = [p + 1 for p in pos_host]
            pos_dev.add_(1)
            hidden = h.reshape(b, -1)

        n_done = 0
        for k in range(max_objects):
            # one host decision per object: is any sequence still emitting objects?
            # (run_all: benchmarks with a fixed amount of work -- every sequence runs max_objects rounds; what a
            # sequence emits after its eos is still dropped below)
            if not run_all:
                alive = (toks[: k + 1] != eos).all(dim=0)
                if not bool(alive.any()):
                    break
            step(self._region_pick_encode(hidden, "coord", bins[k, :, 0:1]))       # x -> y's hidden state
            emb = self._region_pick_encode(hidden, "coord", bins[k, :, 1:2])      # y
            if include_size:
                step(emb)
                emb = self._region_pick_encode(hidden, "size", bins[k, :, 2:4])    # w, h
            step(emb)
            toks[k + 1] = self._pick(self._lm_head(hidden.reshape(b, 1, -1)), 0.0, 0.0)  # next token: x again, or eos
            n_done = k + 1
        bins_h = bins[:n_done].cpu() if n_done else torch.zeros(0, b, 4, dtype=torch.int32)
        toks_h = toks[: n_done + 1].cpu()
        # host floats exactly as the reference forms them (moondream.py:673,683,699-721)
        xc = (bins_h[..., 0].to(torch.int64) / n_bins)
        yc = (bins_h[..., 1].to(torch.int64) / n_bins)
        wv = torch.pow(2.0, (bins_h[..., 2].float() / 1023.0) * 10.0 - 10.0)
        hv = torch.pow(2.0, (bins_h[..., 3].float() / 1023.0) * 10.0 - 10.0)
        out: List[List[dict]] = []
        for i in range(b):
            objs = []
            for k in range(n_done):
                if int(toks_h[k, i]) == eos:
                    break
                x, y = xc[k, i].item(), yc[k, i].item()
                if include_size:
                    w, h = wv[k, i].item(), hv[k, i].item()
                    objs.append({"x_min": x - w / 2, "y_min": y - h / 2, "x_max": x + w / 2, "y_max": y + h / 2})
                else:
                    objs.append({"x": x, "y": y})
            out.append(objs)
        return out

    def _generate_points(self, hidden: torch.Tensor, next_token: torch.Tensor, pos: int, include_size: bool = True,
                         max_objects: int = DEFAULT_MAX_OBJECTS):
        """reference: moondream.py:653-733 (B = 1 form of the lockstep loop; KV slot 0)."""
        with torch.inference_mode():
            first = next_token.reshape(1).to(device=self._device, dtype=torch.int32)
            return self._points_loop(hidden.reshape(1, -1), first, [pos], 0, include_size, max_objects)[0]

    def _batch_detect_like(self, images, objects: Sequence[str], kind: str, include_size: bool, settings: Optional[dict]):
        tpl = self.config.tokenizer.templates[kind]
        if tpl is None:
            raise NotImplementedError(f"Model does not support {kind}.")
        self._region()
        lora = self._lora(settings)  # moondream.py:757-761
        max_objects = (settings or {}).get("max_objects", DEFAULT_MAX_OBJECTS)
        prompts = [list(tpl["prefix"]) + list(self.tokenizer.encode(" " + o).ids) + list(tpl["suffix"]) for o in objects]
        marks = []

        def mark(name):
            if self.collect_timing:
                e = torch.cuda.Event(enable_timing=True)
                e.record(torch.cuda.current_stream(self._device))
                marks.append((name, e))

        with torch.inference_mode():
            order, first, hidden, next_pos = self._prepare_sequences(list(images), prompts, mark, lora)
            res = self._points_loop(hidden, first, next_pos, 0, include_size, max_objects, lora,
                                    run_all=bool((settings or {}).get("_run_all_objects", False)))
            mark("points_loop")
        if self.collect_timing and len(marks) > 1:
            torch.cuda.synchronize(self._device)
            self.last_phase_ms = {marks[i][0]: marks[i - 1][1].elapsed_time(marks[i][1]) for i in range(1, len(marks))}
        out = [None] * len(order)
        for slot, src in enumerate(order):
            out[src] = res[slot]
        return out

    def batch_detect(self, images, objects: Sequence[str], settings: Optional[dict] = None) -> List[dict]:
        """B (image, object) pairs in lockstep; element i == detect(images[i], objects[i])."""
        return [{"objects": o} for o in self._batch_detect_like(images, objects, "detect", True, settings)]

    def batch_point(self, images, objects: Sequence[str], settings: Optional[dict] = None) -> List[dict]:
        return [{"points": o} for o in self._batch_detect_like(images, objects, "point", False, settings)]

    def batch_detect_pipelined(self, batches, settings: Optional[dict] = None, kind: str = "detect"):
        """Generator over an iterable of (images, objects) batches; yields ``batch_detect`` / ``batch_point`` (``kind``) of
        each, in order.  The detect counterpart of ``batch_generate_ids_pipelined``: while batch k runs on the GPU, the host
        tiling of batch k+1 (PIL LANCZOS resize + crop cutting of large images: ~19 ms of CPU per 768x1024 image, reference
        image_crops.py:124-167) is cut into pinned staging buffers by the background workers (``prefetch_crops``), so that
        only the first batch's tiling is exposed.  Results are identical to calling ``batch_detect`` per batch."""
        if kind not in ("detect", "point"):
            raise ValueError("kind must be 'detect' or 'point'")
        run = self.batch_detect if kind == "detect" else self.batch_point
        it = iter(batches)
        cur = next(it, None)
        try:
            while cur is not None:
                nxt = next(it, None)
                if nxt is not None:
                    nxt = (list(nxt[0]), list(nxt[1]))   # (the prefetch entry is keyed by these image objects' identities)
                    self.prefetch_crops(nxt[0])          # queued now, cut while the GPU runs ``cur``
                yield run(cur[0], cur[1], settings)
                cur = nxt
        finally:
            self.discard_prefetched_crops()              # a consumer that stops early: buffers returned

    def detect(self, image, object: str, settings: Optional[dict] = None):
        """reference: moondream.py:735-781."""
        return self.batch_detect([image], [object], settings)[0]

    def point(self, image, object: str, settings: Optional[dict] = None):
        """reference: moondream.py:783-829."""
        return self.batch_point([image], [object], settings)[0]
