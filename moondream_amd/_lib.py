"""ctypes binding of libmoondream_hip.so (C ABI: include/moondream_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc,
--offload-arch=gfx950).  There is deliberately NO fallback: if the shared object
is missing, or a launch fails, callers get an exception.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libmoondream_hip.so")
SOURCES = ["gemm_bf16.hip", "gemm_w4.hip", "gemm_fp8w.hip", "gemm_f8.hip", "quant_f8.hip", "attention_f8kv.hip", "decode_b1.hip", "attention.hip", "elementwise.hip", "sampling_region.hip", "api.hip"]

MD_OK = 0
ABI_VERSION = 5  # include/moondream_hip.h MD_ABI_VERSION
MD_EPI_BIAS, MD_EPI_GELU, MD_EPI_RESIDUAL = 0, 1, 2
MD_TILE_BY_SHAPE, MD_TILE_PINNED, MD_TILE_DECODE_TALL = 0, 1, 2  # md_gemm_args / md_vit_model / md_text_model .tile_policy (ABI 5; 2: round 6)
MD_CROPS_U8_HWC, MD_CROPS_BF16_CHW = 0, 1

c_void_p, c_int32, c_int64, c_size_t, c_float = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.c_float


class MdLinear(C.Structure):
    _fields_ = [("w", c_void_p), ("b", c_void_p), ("n", c_int32), ("k", c_int32), ("n_pad", c_int32), ("k_pad", c_int32)]


class MdLayerNorm(C.Structure):
    _fields_ = [("w", c_void_p), ("b", c_void_p)]


class MdGemmArgs(C.Structure):
    _fields_ = [
        ("a", c_void_p), ("lda", c_int64), ("lin", MdLinear), ("c", c_void_p), ("ldc", c_int64),
        ("r", c_void_p), ("ldr", c_int64), ("res_row_mod", c_int32), ("m", c_int32),
        ("epilogue", c_int32), ("store_pad_cols", c_int32), ("gelu_from_col", c_int32), ("splitk_ws", c_void_p),
        ("splitk_ws_bytes", c_size_t), ("tile_policy", c_int32),
    ]


class MdAttnArgs(C.Structure):
    _fields_ = [
        ("q", c_void_p), ("q_bs", c_int64), ("q_ts", c_int64), ("q_hs", c_int64),
        ("k", c_void_p), ("k_bs", c_int64), ("k_ts", c_int64), ("k_hs", c_int64),
        ("v", c_void_p), ("v_bs", c_int64), ("v_ts", c_int64), ("v_hs", c_int64),
        ("o", c_void_p), ("o_bs", c_int64), ("o_ts", c_int64), ("o_hs", c_int64),
        ("batch", c_int32), ("n_heads", c_int32), ("n_kv_heads", c_int32), ("head_dim", c_int32),
        ("q_len", c_int32), ("kv_len_all", c_int32), ("q_pos0", c_void_p), ("kv_len", c_void_p),
        ("prefix_len", c_int32), ("scale", c_float),
        ("o8", c_void_p), ("o8_bs", c_int64), ("o8_ts", c_int64), ("o8_inv_scale", c_float),
    ]


class MdVitBlock(C.Structure):
    _fields_ = [("ln1", MdLayerNorm), ("qkv", MdLinear), ("proj", MdLinear), ("ln2", MdLayerNorm),
                ("fc1", MdLinear), ("fc2", MdLinear)]


class MdLinearF8(C.Structure):
    _fields_ = [("w", c_void_p), ("scale", c_void_p), ("b", c_void_p), ("n", c_int32), ("k", c_int32),
                ("n_pad", c_int32), ("k_pad", c_int32)]


class MdVitBlockF8(C.Structure):
    _fields_ = [("qkv", MdLinearF8), ("proj", MdLinearF8), ("fc1", MdLinearF8), ("fc2", MdLinearF8),
                ("s_ln1", c_float), ("s_att", c_float), ("s_ln2", c_float), ("s_ff", c_float)]


class MdVitF8(C.Structure):
    _fields_ = [("blocks", C.POINTER(MdVitBlockF8)), ("proj_fc1", MdLinearF8), ("proj_fc2", MdLinearF8),
                ("s_cat", c_float), ("s_pff", c_float), ("calib", c_void_p)]


class MdVitModel(C.Structure):
    _fields_ = [
        ("dim", c_int32), ("n_heads", c_int32), ("n_layers", c_int32), ("ff_dim", c_int32),
        ("patch", c_int32), ("crop", c_int32), ("patch_emb", MdLinear), ("pos_emb", c_void_p),
        ("blocks", C.POINTER(MdVitBlock)), ("post_ln", MdLayerNorm), ("proj_fc1", MdLinear),
        ("proj_fc2", MdLinear), ("pixel_lut", c_void_p), ("f8", C.POINTER(MdVitF8)), ("tile_policy", c_int32),
    ]


class MdTextBlock(C.Structure):
    _fields_ = [("ln", MdLayerNorm), ("qkv", MdLinear), ("proj", MdLinear), ("fc1", MdLinear), ("fc2", MdLinear),
                ("qkv_fc1", MdLinear)]


MD_WSTREAM_E4M3, MD_WSTREAM_INT4_G128 = 0, 1


class MdLinearFp8(C.Structure):
    _fields_ = [("w", c_void_p), ("scale", c_void_p), ("b", c_void_p), ("n", c_int32), ("k", c_int32),
                ("n_pad", c_int32), ("k_pad", c_int32), ("format", c_int32)]


class MdGemmF8Args(C.Structure):
    _fields_ = [
        ("a", c_void_p), ("lda", c_int64), ("a_scale", c_float), ("lin", MdLinearF8), ("c", c_void_p), ("ldc", c_int64),
        ("c8", c_void_p), ("ldc8", c_int64), ("c8_inv_scale", c_float), ("f8_from_col", c_int32), ("r", c_void_p), ("ldr", c_int64),
        ("res_row_mod", c_int32), ("m", c_int32), ("epilogue", c_int32), ("store_pad_cols", c_int32), ("gelu_from_col", c_int32),
    ]


class MdTextBlockFp8(C.Structure):
    _fields_ = [("qkv_fc1", MdLinearFp8), ("proj", MdLinearFp8), ("fc2", MdLinearFp8)]


class MdTextFp8(C.Structure):
    _fields_ = [("blocks", C.POINTER(MdTextBlockFp8)), ("lm_head", MdLinearFp8)]


class MdTextBlockF8(C.Structure):
    _fields_ = [("qkv_fc1", MdLinearF8), ("proj", MdLinearF8), ("fc2", MdLinearF8),
                ("s_ln", c_float), ("s_att", c_float), ("s_ff", c_float)]


class MdTextF8(C.Structure):
    _fields_ = [("blocks", C.POINTER(MdTextBlockF8)), ("calib", c_void_p)]


class MdTextModel(C.Structure):
    _fields_ = [
        ("dim", c_int32), ("n_heads", c_int32), ("n_kv_heads", c_int32), ("n_layers", c_int32),
        ("ff_dim", c_int32), ("vocab", c_int32), ("max_context", c_int32), ("prefix_len", c_int32),
        ("rot_dim", c_int32), ("blocks", C.POINTER(MdTextBlock)), ("post_ln", MdLayerNorm),
        ("lm_head", MdLinear), ("wte", c_void_p), ("freqs", c_void_p), ("fp8", C.POINTER(MdTextFp8)),
        ("f8", C.POINTER(MdTextF8)), ("tile_policy", c_int32),
    ]


class MdLoraPair(C.Structure):
    _fields_ = [("a", MdLinear), ("b", MdLinear)]


class MdTextBlockLora(C.Structure):
    _fields_ = [("qkv", MdLoraPair), ("proj", MdLoraPair), ("fc1", MdLoraPair), ("fc2", MdLoraPair)]


class MdKvCache(C.Structure):
    _fields_ = [("k", c_void_p), ("v", c_void_p), ("layer_stride", c_int64), ("batch_stride", c_int64), ("ctx", c_int32),
                ("k8", c_void_p), ("v8", c_void_p), ("k_scale", c_void_p), ("v_scale", c_void_p)]


# name -> (restype, argtypes): every symbol include/moondream_hip.h declares
P = C.POINTER
SIGNATURES = {
    "md_abi_version": (C.c_int, []),
    "md_status_string": (C.c_char_p, [C.c_int]),
    "md_gemm_bf16": (C.c_int, [P(MdGemmArgs), c_void_p]),
    "md_gemm_workspace_bytes": (c_size_t, [P(MdLinear), c_int32, c_int32]),
    "md_gemm_partial_slices": (c_int32, [P(MdLinear)]),
    "md_gemm_partial_f32": (C.c_int, [c_void_p, c_int64, P(MdLinear), c_int32, c_void_p, c_int64, c_int64, c_void_p]),
    "md_gemm_partial_f32_pair": (C.c_int, [c_void_p, c_int64, P(MdLinear), c_void_p, c_void_p, c_int64, P(MdLinear), c_void_p,
                                           c_int32, c_int64, c_int64, c_void_p]),
    "md_reduce_residual_layernorm": (C.c_int, [c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p,
                                               c_int64, c_int64, c_void_p, c_int64, P(MdLayerNorm), c_int32, c_int32, c_float, c_void_p]),
    "md_gemm_fp8w": (C.c_int, [c_void_p, c_int64, P(MdLinearFp8), c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "md_gemm_fp8w_partial_slices": (c_int32, [P(MdLinearFp8)]),
    "md_gemm_fp8w_partial_f32_pair": (C.c_int, [c_void_p, c_int64, P(MdLinearFp8), c_void_p, c_void_p, c_int64, P(MdLinearFp8),
                                                c_void_p, c_int32, c_int64, c_int64, c_void_p]),
    "md_gemm_f8": (C.c_int, [P(MdGemmF8Args), c_void_p]),
    "md_quantize_f8": (C.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "md_layernorm_f8": (C.c_int, [c_void_p, c_int64, c_void_p, c_int64, P(MdLayerNorm), c_int32, c_int32, c_int32, c_float, c_float, c_void_p]),
    "md_amax_bf16": (C.c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "md_attention_decode_rope_f8": (C.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                              c_int32, c_void_p, c_int32, c_int32, c_int32, c_float, c_float, c_float, c_void_p]),
    "md_kv_quantize_f8": (C.c_int, [P(MdKvCache), c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p]),
    "md_gemm_set_tuning": (C.c_int, [C.c_char_p, c_int32]),
    "md_profile_gemm": (None, [c_int32]),
    "md_profile_gemm_read": (C.c_int, [c_int32, P(C.c_double), P(C.c_double), P(c_int64)]),
    "md_profile_gemm_bytes": (C.c_int, [c_int32, P(C.c_double), P(C.c_double)]),
    "md_layernorm_bf16": (C.c_int, [c_void_p, c_int64, c_void_p, c_int64, P(MdLayerNorm), c_int32, c_int32, c_float, c_void_p]),
    "md_patchify_u8": (C.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "md_patchify_bf16": (C.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "md_attention_prefill": (C.c_int, [P(MdAttnArgs), c_void_p]),
    "md_attention_decode": (C.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32,
                                      c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "md_attention_decode_rope": (C.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                                           c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "md_rope_kv_write": (C.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                   c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "md_embed_tokens": (C.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "md_argmax_bf16": (C.c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "md_sample_top_p": (C.c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_float, c_float, c_void_p, c_void_p,
                                  c_void_p, c_int64, c_void_p]),
    "md_fourier_features": (C.c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int64, c_void_p]),
    "md_region_pick_encode": (C.c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                        c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "md_stitch_pool_concat": (C.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "md_vit_workspace_bytes": (c_size_t, [P(MdVitModel), c_int32]),
    "md_vit_encode": (C.c_int, [P(MdVitModel), c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_size_t, c_void_p]),
    "md_vision_project_workspace_bytes": (c_size_t, [P(MdVitModel), c_int32]),
    "md_vision_project": (C.c_int, [P(MdVitModel), c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64,
                                    c_void_p, c_size_t, c_void_p]),
    "md_vision_project_grid": (C.c_int, [P(MdVitModel), c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int64,
                                         c_void_p, c_size_t, c_void_p]),
    "md_text_workspace_bytes": (c_size_t, [P(MdTextModel), c_int32, c_int32]),
    "md_text_forward": (C.c_int, [P(MdTextModel), c_void_p, c_void_p, c_int32, c_int32, c_void_p, P(MdKvCache),
                                  c_void_p, c_size_t, c_void_p]),
    "md_text_lora_workspace_bytes": (c_size_t, [P(MdTextModel), c_int32, c_int32]),
    "md_text_forward_lora": (C.c_int, [P(MdTextModel), c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, P(MdKvCache),
                                       c_void_p, c_size_t, c_void_p]),
    "md_add_bf16": (C.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "md_gelu_bf16": (C.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "md_lm_head_workspace_bytes": (c_size_t, [P(MdTextModel), c_int32]),
    "md_lm_head": (C.c_int, [P(MdTextModel), c_void_p, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "md_decode_b1_workspace_bytes": (c_size_t, [P(MdTextModel)]),
    "md_decode_b1_layers": (C.c_int, [P(MdTextModel), c_void_p, c_void_p, c_void_p, P(MdKvCache), c_void_p, c_size_t, c_void_p, c_void_p]),
    "md_decode_step_b1_workspace_bytes": (c_size_t, [P(MdTextModel)]),
    "md_decode_step_b1_supported": (c_int32, [P(MdTextModel), P(MdKvCache)]),
    "md_decode_step_b1": (C.c_int, [P(MdTextModel), c_void_p, c_void_p, c_void_p, P(MdKvCache), c_int32, c_void_p, c_int64,
                                    c_void_p, c_size_t, c_void_p, c_void_p]),
    "md_decode_workspace_bytes": (c_size_t, [P(MdTextModel), c_int32]),
    "md_decode_step": (C.c_int, [P(MdTextModel), c_void_p, c_void_p, c_void_p, c_int32, P(MdKvCache), c_int32,
                                 c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
}


class MoondreamHipError(RuntimeError):
    pass


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def build_library(force: bool = False, verbose: bool = True) -> str:
    """Compile csrc/*.hip for gfx950 into moondream_amd/libmoondream_hip.so."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in ("md_common.hpp", "gemm_internal.hpp")] + [os.path.join(REPO, "include", "moondream_hip.h")]
    if not force and os.path.exists(LIB_PATH):
        if os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(d) for d in deps + [os.path.abspath(__file__)]):
            return LIB_PATH
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    base = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast"]
    # -amdgpu-mfma-vgpr-form: MFMA accumulators stay in architectural VGPRs.  Every kernel built with it
    # fits in <= 256 registers per wave, and the softmax / epilogue code works on the accumulators with
    # VALU instructions, so the default AGPR placement only adds v_accvgpr_read/write traffic
    # (prefill attention: ~190 moves per 64-key tile, and 216 -> 158 registers).
    # gemm_w4.hip is the exception: its 256 accumulator registers per wave live in a[0:255], named
    # literally by inline asm (512-register waves, one per SIMD); it is compiled with -save-temps and its
    # ISA is audited below.
    vgpr_form = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
    procs = []
    objs = []
    for s in srcs:
        name = os.path.basename(s)
        o = os.path.join(objdir, name + ".o")
        objs.append(o)
        flags = base + (["-save-temps=obj"] if name == "gemm_w4.hip" else vgpr_form)
        if name == "gemm_w4.hip" and os.environ.get("MD_W4_ABLATIONS"):
            flags.append("-DMD_W4_ABLATIONS")  # tools/w4_probe.py's timing-ablation variants
        cmd = [hipcc_path(), *flags, "-c", s, "-o", o]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=objdir)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise MoondreamHipError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode()))
    audit_asm_owned_accumulators(os.path.join(objdir, "gemm_w4-hip-amdgcn-amd-amdhsa-gfx950.s"))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise MoondreamHipError("link failed:\n" + r.stdout.decode())
    return LIB_PATH


def audit_asm_owned_accumulators(asm_path: str) -> None:
    """gemm_w4.hip keeps its accumulators in a[0:255] through inline asm.  That is only sound while
    the compiler itself never touches the accumulator file or spills in those kernels: any
    v_accvgpr_* outside an inline-asm block, or any scratch access, fails the build.  The same for M0,
    which its LDS-DMA groups carry across asm statements."""
    if not os.path.exists(asm_path):
        raise MoondreamHipError(f"{asm_path} missing: gemm_w4.hip must be compiled with -save-temps=obj for the ISA audit")
    import re

    in_asm, kernel, bad, seen = False, None, [], 0
    label = re.compile(r"^(_Z\S*gemm_w4_kernel\S*):")  # the label line may carry a trailing "; @name" comment
    for ln in open(asm_path):
        t = ln.strip()
        m = label.match(t)
        if m:
            kernel = m.group(1)
            seen += 1
        if "ASMSTART" in t:
            in_asm = True
        elif "ASMEND" in t:
            in_asm = False
        elif kernel and not t.startswith(";"):
            if ("v_accvgpr" in t and not in_asm) or t.startswith("scratch_") or ("a[" in t and not in_asm and t.startswith("v_mfma")):
                bad.append(f"{kernel[-40:]}: {t}")
            # M0 carries the LDS block of the running LDS-DMA group from one asm statement to the next
            if not in_asm and re.search(r"\bm0\b", t.split(";")[0]):
                bad.append(f"{kernel[-40:]}: compiler-generated use of m0: {t}")
        if t.startswith("s_endpgm"):
            kernel = None
    if seen == 0:
        raise MoondreamHipError(f"{asm_path}: no gemm_w4_kernel label found -- the ISA audit would be vacuous")
    if bad:
        raise MoondreamHipError("gemm_w4.hip: compiler-generated accumulator-file / scratch traffic in a kernel whose "
                                "a[0:255] are owned by inline asm:\n  " + "\n  ".join(bad[:10]))


_LIB = None


def load() -> C.CDLL:
    """dlopen the library and type every entry point.  Raises if it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("MD_HIP_LIB") or LIB_PATH  # MD_HIP_LIB: another build of the SAME ABI, for same-box A/B runs (tools/ab_lib.sh)
    if not os.path.exists(path):
        raise MoondreamHipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the Moondream hot path)"
        )
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and this table diverge
        fn.restype = res
        fn.argtypes = args
    if lib.md_abi_version() != ABI_VERSION:
        raise MoondreamHipError("libmoondream_hip.so ABI version mismatch")
    _LIB = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != MD_OK:
        msg = load().md_status_string(status).decode()
        raise MoondreamHipError(f"{what or 'libmoondream_hip'}: status {status}: {msg}")


def exported_symbols() -> List[str]:
    return sorted(SIGNATURES)
