"""Kernel-level parity on a real MI355X: every HIP kernel, called through the
C ABI, against a plain fp32 torch restatement of the same op (and against the
CPU oracle's functions where one exists).  Sizes cover the awkward shapes of
the real models: K = 588, head_dim 72, 729 tokens, FF 4304, ragged M."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

from moondream_amd import _lib
from moondream_amd.weights import PackedLinear, PackedLayerNorm, rope_table, reference_pixel_lut
from util import compare, quantize_int4

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a device"
    return _lib.load()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.fixture
def force_tile(lib):
    """Pin the GEMM tile config through the library's measurement hook; automatic again afterwards."""
    def set_tile(tile):
        _lib.check(lib.md_gemm_set_tuning(b"tile", int(tile)))
    yield set_tile
    lib.md_gemm_set_tuning(b"tile", -1)


BIG_TILES = ["20", "1", "2", "11", "15"]  # 20 = four-wave 256x256 (default for big shapes); 11 / 15 = its eight-wave baselines


def randn(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF16).cuda()


_SPLITK_WS = {}  # zeroed once; every launch leaves the ticket area zero again


def gemm(lib, a, lin, epi=0, r=None, res_row_mod=0, store_pad=0, out=None, use_ws=True, tile_policy=0):
    m = a.shape[0]
    width = lin.n_pad if store_pad else lin.n
    c = out if out is not None else torch.full((m, width), float("nan"), dtype=BF16, device="cuda")
    st = lin.struct()
    need = lib.md_gemm_workspace_bytes(C.byref(st), m, store_pad)
    ws = _SPLITK_WS.setdefault(need, torch.zeros(max(need, 16), dtype=torch.uint8, device="cuda")) if use_ws else None
    args = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), st, c.data_ptr(), c.stride(0),
                           r.data_ptr() if r is not None else None, r.stride(0) if r is not None else 0,
                           res_row_mod, m, epi, store_pad, 0, ws.data_ptr() if ws is not None else None,
                           need if ws is not None else 0, tile_policy)
    _lib.check(lib.md_gemm_bf16(C.byref(args), stream()), "gemm")
    torch.cuda.synchronize()
    return c


def ref_linear(a, w, b):
    return (a.float() @ w.float().t() + b.float()).to(BF16)


def pad_k(a, k_pad):
    out = torch.zeros(a.shape[0], k_pad, dtype=BF16, device="cuda")
    out[:, : a.shape[1]] = a
    return out


@pytest.mark.parametrize("tile", BIG_TILES)
def test_gemm_identity_detects_transposes(lib, tile, force_tile):
    """A = I with an ASYMMETRIC W: C must equal W^T bit for bit."""
    force_tile(tile)
    n = k = 512
    w = (torch.arange(n * k, dtype=torch.float32).reshape(n, k) % 251 - 125).to(BF16).cuda()
    a = torch.eye(k, dtype=BF16, device="cuda")
    lin = PackedLinear(w, torch.zeros(n, dtype=BF16), "cuda")
    c = gemm(lib, a, lin)
    assert torch.equal(c, w.t().contiguous())


@pytest.mark.parametrize("tile", BIG_TILES)
@pytest.mark.parametrize("m,k,n", [(300, 588, 1152), (777, 1152, 3456), (1000, 2048, 6144), (64, 2048, 1024), (1, 256, 64)])
def test_gemm_bias(lib, tile, m, k, n, force_tile):
    force_tile(tile)
    a, w, b = randn(m, k, seed=1), randn(n, k, scale=1 / math.sqrt(k), seed=2), randn(n, scale=0.1, seed=3)
    lin = PackedLinear(w, b, "cuda")
    c = gemm(lib, pad_k(a, lin.k_pad), lin)
    compare(f"gemm_bias {m}x{k}x{n} tile{tile}", c, ref_linear(a, w, b), 3e-3, 2e-2)


def test_gemm_tile_configs_agree_bitwise(lib, force_tile):
    """The 32x32x16 tile configs accumulate K in the same order, so a layer's output does not depend on which of them the
    heuristic picks; the four-wave kernel (tile 20) multiplies with 16x16x32 MFMAs since round 4 and agrees with them to
    fp32 rounding of the accumulator (a rare last bf16 bit), and with itself bit for bit.  Repeated launches double as a
    race screen for the hand-synchronised operand rings."""
    m, k, n = 2100, 4096, 2304
    a, w, b = randn(m, k, seed=11), randn(n, k, scale=1 / math.sqrt(k), seed=12), randn(n, scale=0.1, seed=13)
    lin = PackedLinear(w, b, "cuda")
    force_tile(2)
    want = gemm(lib, a, lin)
    for tile in ("1", "11", "15", "16"):  # (16: the decode-regime config, which the single-image regime's few-tile layers take too)
        force_tile(tile)
        for rep in range(6):
            got = gemm(lib, a, lin)
            assert torch.equal(got, want), f"tile {tile} rep {rep}"
    force_tile(20)
    first = gemm(lib, a, lin)
    compare("tile 20 vs tile 2", first, want, 3e-4, 2e-2)
    assert (first != want).float().mean().item() < 0.02  # a last-bit difference is rare
    for rep in range(6):
        assert torch.equal(gemm(lib, a, lin), first), f"tile 20 rep {rep}"


def test_tile_policy_is_a_per_call_field(lib, force_tile):
    """ABI 5: md_gemm_args.tile_policy.  MD_TILE_PINNED makes the tile config of a > 64-row launch a function of the layer alone --
    the 256 x 256 kernel whatever the row count -- so a row gets the same bits in a 100-row and in a 5000-row launch; MD_TILE_BY_SHAPE
    lets small launches take the small-shape configs (another MFMA shape: equal to fp32 rounding, not bitwise).  The policy travels
    with the call: two interleaved callers with different policies do not affect each other, the "w4" A/B knob is honoured under the
    pin (advisor, round 4), launches of <= 64 rows ignore it, and an unknown value is refused."""
    k, n = 2048, 2048
    w, b = randn(n, k, scale=1 / math.sqrt(k), seed=52), randn(n, scale=0.1, seed=53)
    lin = PackedLinear(w, b, "cuda")
    big = randn(5000, k, seed=51)
    small = big[:100].contiguous()
    pinned_big = gemm(lib, big, lin, tile_policy=_lib.MD_TILE_PINNED)
    force_tile(20)
    assert torch.equal(gemm(lib, big, lin), pinned_big)           # the pin IS the four-wave kernel
    force_tile(-1)
    for rep in range(3):  # interleaved callers
        p_small = gemm(lib, small, lin, tile_policy=_lib.MD_TILE_PINNED)
        s_small = gemm(lib, small, lin, tile_policy=_lib.MD_TILE_BY_SHAPE)
        assert torch.equal(p_small, pinned_big[:100])             # same bits alone and in the big launch
        compare("by-shape vs pinned", s_small, p_small, 3e-4, 2e-2)
    _lib.check(lib.md_gemm_set_tuning(b"w4", 0))
    try:
        eight = gemm(lib, small, lin, tile_policy=_lib.MD_TILE_PINNED)
        force_tile(11)
        assert torch.equal(gemm(lib, small, lin), eight)          # pinned + w4 = 0 -> the eight-wave 256 x 256 baseline
    finally:
        force_tile(-1)
        _lib.check(lib.md_gemm_set_tuning(b"w4", 1))
    rows = randn(48, k, seed=54)
    assert torch.equal(gemm(lib, rows, lin, tile_policy=_lib.MD_TILE_PINNED), gemm(lib, rows, lin, tile_policy=_lib.MD_TILE_BY_SHAPE))
    st = lin.struct()
    c = torch.empty(100, n, dtype=BF16, device="cuda")
    bad = _lib.MdGemmArgs(small.data_ptr(), small.stride(0), st, c.data_ptr(), c.stride(0), None, 0, 0, 100, 0, 0, 0, None, 0, 7)
    assert lib.md_gemm_bf16(C.byref(bad), stream()) == 1  # MD_ERR_INVALID_ARG


def test_pinned_policy_cuts_an_oversized_launch_into_row_blocks_instead_of_changing_kernel(lib):
    """The four-wave kernel addresses C with 32-bit byte offsets ((m + 256) * ldc * 2 < 0xfffff000).  Under MD_TILE_PINNED a
    launch beyond that used to fall back SILENTLY to the 32x32x16 family -- other bits than the same row in a small launch
    (advisor, round 5).  Now it is cut into row blocks of the same kernel (bit-identical to a compact launch of the same rows);
    what cannot be cut -- a broadcast residual -- is refused with MD_ERR_UNSUPPORTED; MD_TILE_BY_SHAPE keeps its fallback."""
    k, n, m = 512, 256, 20000
    ldc = 131072                                   # (m + 256) * ldc * 2 = 5.3 GB of C address span -> two row blocks
    w, b = randn(n, k, scale=1 / math.sqrt(k), seed=61), randn(n, scale=0.1, seed=62)
    lin = PackedLinear(w, b, "cuda")
    a = randn(m, k, seed=63)
    compact = gemm(lib, a, lin, tile_policy=_lib.MD_TILE_PINNED)
    wide = torch.empty(m, ldc, dtype=BF16, device="cuda")
    got = gemm(lib, a, lin, out=wide[:, :n], tile_policy=_lib.MD_TILE_PINNED)
    assert got.stride(0) == ldc and torch.equal(got, compact)
    res = randn(m, n, seed=64)
    got_r = gemm(lib, a, lin, epi=_lib.MD_EPI_RESIDUAL, r=res, out=wide[:, :n], tile_policy=_lib.MD_TILE_PINNED)
    assert torch.equal(got_r, gemm(lib, a, lin, epi=_lib.MD_EPI_RESIDUAL, r=res, tile_policy=_lib.MD_TILE_PINNED))
    st = lin.struct()
    pos = randn(729, n, seed=65)
    args = _lib.MdGemmArgs(a.data_ptr(), a.stride(0), st, wide.data_ptr(), ldc, pos.data_ptr(), pos.stride(0), 729, m,
                           _lib.MD_EPI_RESIDUAL, 0, 0, None, 0, _lib.MD_TILE_PINNED)
    assert lib.md_gemm_bf16(C.byref(args), stream()) == 4  # MD_ERR_UNSUPPORTED
    by_shape = gemm(lib, a, lin, out=wide[:, :n], tile_policy=_lib.MD_TILE_BY_SHAPE)   # still served (another MFMA family)
    compare("by-shape fallback vs pinned blocks", by_shape, compact, 3e-4, 2e-2)
    del wide
    torch.cuda.empty_cache()


@pytest.mark.parametrize("m", [64, 33, 1])
def test_decode_regime_configs_agree_bitwise(lib, m):
    """Decode regime (m <= 64): every 64 x 64 config -- 16 (two compute waves + two DMA-only helpers), 17 (round 5: FOUR compute
    waves as 2 x 2 wave tiles of 32 x 32, 128-wide K slices where the layer allows), 18 / 19 (the two ingredients on their own),
    10 (no helpers) -- sums K in the same order per output element, in-launch split-K slices and launch-boundary partial slices
    included: bit-identical outputs for every epilogue, ragged N, K not a multiple of 128, and both partial entry points.
    Repeated launches double as a race screen for the new ring shapes."""
    def set_cfg(c):
        _lib.check(lib.md_gemm_set_tuning(b"decode_cfg", c))
    try:
        for (k, n, epi, gelu_from) in [(2048, 6144 + 8192, 1, 6144), (2048, 2048, 2, 0), (8192, 2048, 0, 0), (2048, 51200, 0, 0),
                                       (704, 256, 0, 0), (1152, 1000, 1, 0), (256, 1024, 2, 0), (4352, 1152, 0, 0)]:
            a, w, b = randn(m, k, seed=31), randn(n, k, scale=1 / math.sqrt(k), seed=32), randn(n, scale=0.1, seed=33)
            lin = PackedLinear(w, b, "cuda")
            ap = pad_k(a, lin.k_pad)
            r = randn(m, n, seed=34) if epi == 2 else None

            def run():
                c = torch.full((m, lin.n), float("nan"), dtype=BF16, device="cuda")
                st = lin.struct()
                need = lib.md_gemm_workspace_bytes(C.byref(st), m, 0)
                ws = torch.zeros(max(need, 16), dtype=torch.uint8, device="cuda")
                args = _lib.MdGemmArgs(ap.data_ptr(), ap.stride(0), st, c.data_ptr(), c.stride(0), r.data_ptr() if r is not None else None,
                                       r.stride(0) if r is not None else 0, 0, m, epi, 0, gelu_from, ws.data_ptr(), need)
                _lib.check(lib.md_gemm_bf16(C.byref(args), stream()), "gemm")
                torch.cuda.synchronize()
                return c
            set_cfg(16)
            want = run()
            ref = ref_linear(a, w, b).float()
            if epi == 1:
                ref[:, gelu_from:] = torch.nn.functional.gelu(ref[:, gelu_from:], approximate="tanh")
            if epi == 2:
                ref = ref + r.float()
            compare(f"decode cfg 16 k={k} n={n} epi={epi}", want, ref.to(BF16), 4e-3, 6e-2)
            for cfg in (17, 18, 19, 10):
                set_cfg(cfg)
                for rep in range(3):
                    assert torch.equal(run(), want), f"cfg {cfg} k={k} n={n} epi={epi} rep {rep}"
        # launch-boundary partial slices: single and paired entry points
        D, FF = 2048, 8192
        a1, a2 = randn(m, D, seed=41), randn(m, FF, seed=42)
        W1, B1 = randn(D, D, scale=1 / math.sqrt(D), seed=43), randn(D, scale=0.1, seed=44)
        l1_bias = lambda _l: B1.float()
        l1 = PackedLinear(W1, B1, "cuda")
        l2 = PackedLinear(randn(D, FF, scale=1 / math.sqrt(FF), seed=45), randn(D, scale=0.1, seed=46), "cuda")
        s1, s2 = l1.struct(), l2.struct()
        n1, n2 = lib.md_gemm_partial_slices(C.byref(s1)), lib.md_gemm_partial_slices(C.byref(s2))

        def partials(pair):
            p1 = torch.full((n1, m, D), float("nan"), dtype=torch.float32, device="cuda")
            p2 = torch.full((n2, m, D), float("nan"), dtype=torch.float32, device="cuda")
            if pair:
                _lib.check(lib.md_gemm_partial_f32_pair(a1.data_ptr(), a1.stride(0), C.byref(s1), p1.data_ptr(), a2.data_ptr(), a2.stride(0),
                                                        C.byref(s2), p2.data_ptr(), m, D, m * D, stream()))
            else:
                _lib.check(lib.md_gemm_partial_f32(a1.data_ptr(), a1.stride(0), C.byref(s1), m, p1.data_ptr(), D, m * D, stream()))
                _lib.check(lib.md_gemm_partial_f32(a2.data_ptr(), a2.stride(0), C.byref(s2), m, p2.data_ptr(), D, m * D, stream()))
            torch.cuda.synchronize()
            return p1, p2
        set_cfg(16)
        w1, w2 = partials(True)
        compare("partial slices sum (proj)", (w1.sum(0) + l1_bias(l1)).to(BF16), ref_linear(a1, W1, B1), 4e-3, 6e-2)
        for cfg in (16, 17, 18, 19, 10):
            set_cfg(cfg)
            for pair in (True, False):
                g1, g2 = partials(pair)
                assert torch.equal(g1, w1) and torch.equal(g2, w2), f"partials cfg {cfg} pair {pair}"
    finally:
        lib.md_gemm_set_tuning(b"decode_cfg", 16)


@pytest.mark.parametrize("epi", [0, 1, 2])
@pytest.mark.parametrize("m,k,n", [(46720 // 8, 2048, 2048), (2 * 729 * 4 + 77, 1152, 4304), (3000, 4352, 1152), (257, 64, 8), (5000, 640, 1152)])
def test_gemm_w4_persistent_stream(lib, force_tile, m, k, n, epi):
    """The four-wave 256x256 kernel (gemm_w4.hip): several tiles per workgroup through ONE continuous
    slice stream (tile boundaries, ragged M and N edges, K = 64 .. 4352), all three epilogues, in-place
    residual; against the fp32 reference, against the 128x128 config (same roundings, another MFMA shape)
    and repeated bit for bit as a race screen for its hand-placed barrier / counted waits."""
    a, w, b = randn(m, k, seed=21), randn(n, k, scale=1 / math.sqrt(k), seed=22), randn(n, scale=0.1, seed=23)
    lin = PackedLinear(w, b, "cuda")
    x = randn(m, lin.n_pad if epi == 1 else n, seed=24)
    store_pad = 1 if epi == 1 else 0

    def run():
        if epi == 2:
            out = x.clone()
            return gemm(lib, pad_k(a, lin.k_pad), lin, epi=2, r=out, out=out)
        return gemm(lib, pad_k(a, lin.k_pad), lin, epi=epi, store_pad=store_pad)

    force_tile(2)
    want = run()
    ref = ref_linear(a, w, b)
    if epi == 1:
        ref = torch.nn.functional.gelu(ref.float(), approximate="tanh").to(BF16)
    if epi == 2:
        ref = (x.float() + ref.float()).to(BF16)
    compare(f"w4 reference m{m} k{k} n{n} epi{epi}", want[:, :n], ref, 3e-3, 2e-2)
    force_tile(20)
    first = run()
    compare(f"w4 m{m} k{k} n{n} epi{epi}", first[:, :n], ref, 3e-3, 2e-2)
    compare(f"w4 vs tile 2 m{m} k{k} n{n} epi{epi}", first, want, 5e-4, 3e-2)  # 16x16x32 vs 32x32x16 MFMAs: fp32 rounding of the accumulator
    if epi == 1:
        assert torch.count_nonzero(first[:, n:]) == 0  # the zero pad columns of a padded output
    for rep in range(4):
        got = run()
        assert torch.equal(got, first), f"rep {rep}: {(got != first).sum().item()} elements differ"


def test_gemm_gelu_writes_zero_pad_columns(lib):
    m, k, n = 515, 1152, 4304
    a, w, b = randn(m, k, seed=4), randn(n, k, scale=1 / math.sqrt(k), seed=5), randn(n, scale=0.1, seed=6)
    lin = PackedLinear(w, b, "cuda")
    assert lin.n_pad == 4352
    c = gemm(lib, a, lin, epi=_lib.MD_EPI_GELU, store_pad=1)
    ref = torch.nn.functional.gelu(ref_linear(a, w, b).float(), approximate="tanh").to(BF16)
    compare("gemm_gelu", c[:, :n], ref, 3e-3, 2e-2)
    assert torch.count_nonzero(c[:, n:]) == 0


def test_gemm_fused_pair_gelu_from_column(lib):
    """[qkv | fc1] over one input as ONE GEMM: bias-only for the first block of
    columns, bias+GELU from column na on; element-for-element what the two
    separate layers give."""
    from moondream_amd.weights import FusedLinear

    for m in (730, 64, 5):
        k, na, nb = 256, 768, 704
        a = randn(m, k, seed=50)
        wa, ba = randn(na, k, scale=1 / 16, seed=51), randn(na, scale=0.1, seed=52)
        wb, bb = randn(nb, k, scale=1 / 16, seed=53), randn(nb, scale=0.1, seed=54)
        f = FusedLinear(wa, ba, wb, bb, "cuda")
        st = f.struct()
        c = torch.full((m, f.n_pad), float("nan"), dtype=BF16, device="cuda")
        need = lib.md_gemm_workspace_bytes(C.byref(st), m, 1)
        ws = torch.zeros(max(need, 16), dtype=torch.uint8, device="cuda")
        args = _lib.MdGemmArgs(a.data_ptr(), k, st, c.data_ptr(), f.n_pad, None, 0, 0, m, _lib.MD_EPI_GELU, 1, na, ws.data_ptr(), need)
        _lib.check(lib.md_gemm_bf16(C.byref(args), stream()))
        torch.cuda.synchronize()
        # the same two layers run separately through their row-range views
        la, lb = f.struct_a(), f.struct_b()
        ca = torch.empty(m, na, dtype=BF16, device="cuda")
        cb = torch.empty(m, f.nb_pad, dtype=BF16, device="cuda")
        for stv, out, epi, sp in ((la, ca, 0, 0), (lb, cb, 1, 1)):
            nd = lib.md_gemm_workspace_bytes(C.byref(stv), m, sp)
            w2 = torch.zeros(max(nd, 16), dtype=torch.uint8, device="cuda")
            ar = _lib.MdGemmArgs(a.data_ptr(), k, stv, out.data_ptr(), out.stride(0), None, 0, 0, m, epi, sp, 0, w2.data_ptr(), nd)
            _lib.check(lib.md_gemm_bf16(C.byref(ar), stream()))
        torch.cuda.synchronize()
        compare("fused qkv part", c[:, :na], ref_linear(a, wa, ba), 3e-3, 2e-2)
        ref_b = torch.nn.functional.gelu(ref_linear(a, wb, bb).float(), approximate="tanh").to(BF16)
        compare("fused fc1 part", c[:, na : na + nb], ref_b, 3e-3, 2e-2)
        if m > 64:  # same tile kernel, same K order -> bit-identical to the separate layers
            assert torch.equal(c[:, :na], ca) and torch.equal(c[:, na:], cb)


def test_gemm_residual_in_place_and_row_mod(lib):
    m, k, n = 2 * 729, 4352, 1152
    a, w, b = randn(m, k, seed=7), randn(n, k, scale=1 / math.sqrt(k), seed=8), randn(n, scale=0.1, seed=9)
    x = randn(m, n, seed=10)
    lin = PackedLinear(w, b, "cuda")
    ref = (x.float() + ref_linear(a, w, b).float()).to(BF16)
    out = x.clone()
    gemm(lib, a, lin, epi=_lib.MD_EPI_RESIDUAL, r=out, out=out)  # r aliases c, like x += f(x)
    compare("gemm_residual_inplace", out, ref, 3e-3, 2e-2)
    pos = randn(729, n, seed=11)  # pos_emb broadcast over crops
    ref2 = (pos.float().repeat(2, 1) + ref_linear(a, w, b).float()).to(BF16)
    c2 = gemm(lib, a, lin, epi=_lib.MD_EPI_RESIDUAL, r=pos, res_row_mod=729)
    compare("gemm_residual_rowmod", c2, ref2, 3e-3, 2e-2)


def test_gemm_batch_invariance(lib):
    """Row i of the result must not depend on M (batched == sequential decode)."""
    k, n = 2048, 2048
    a, w, b = randn(64, k, seed=12), randn(n, k, scale=1 / math.sqrt(k), seed=13), randn(n, scale=0.1, seed=14)
    lin = PackedLinear(w, b, "cuda")
    full = gemm(lib, a, lin)
    for rows in (1, 5, 33):
        assert torch.equal(gemm(lib, a[:rows].contiguous(), lin), full[:rows])
    # big-tile regime: M = 1000 and M = 200 pick different tile configs
    a2 = randn(1000, k, seed=12)
    assert torch.equal(gemm(lib, a2, lin)[:200], gemm(lib, a2[:200].contiguous(), lin))


@pytest.mark.parametrize("m", [1, 7, 33, 64])
@pytest.mark.parametrize("k,n,epi", [(2048, 6144, 0), (2048, 8192, 1), (8192, 2048, 2), (588, 1152, 0), (256, 1024, 2)])
def test_gemm_decode_regime(lib, m, k, n, epi):
    """m <= 64 dispatches to the weight-streaming kernel (in-workgroup split-K)."""
    a, w, b = randn(m, k, seed=40), randn(n, k, scale=1 / math.sqrt(k), seed=41), randn(n, scale=0.1, seed=42)
    lin = PackedLinear(w, b, "cuda")
    r = randn(m, n, seed=43)
    c = gemm(lib, pad_k(a, lin.k_pad), lin, epi=epi, r=r if epi == 2 else None)
    ref = ref_linear(a, w, b)
    if epi == 1:
        ref = torch.nn.functional.gelu(ref.float(), approximate="tanh").to(BF16)
    if epi == 2:
        ref = (r.float() + ref.float()).to(BF16)
    compare(f"skinny m{m} {k}x{n} epi{epi}", c, ref, 3e-3, 2e-2)
    # same scratch, launched again (tickets were left zero), and row-subset invariance
    assert torch.equal(c, gemm(lib, pad_k(a, lin.k_pad), lin, epi=epi, r=r if epi == 2 else None))
    if m > 1:
        sub = gemm(lib, pad_k(a, lin.k_pad)[:1].contiguous(), lin, epi=epi, r=r[:1].contiguous() if epi == 2 else None)
        assert torch.equal(sub, c[:1])
    # without scratch K is not split across workgroups: still correct
    c1 = gemm(lib, pad_k(a, lin.k_pad), lin, epi=epi, r=r if epi == 2 else None, use_ws=False)
    compare(f"skinny(no scratch) m{m} {k}x{n}", c1, ref, 3e-3, 2e-2)


@pytest.mark.parametrize("m", [1, 7, 32, 33, 64])
@pytest.mark.parametrize("k,n,epi,gelu_from", [(2048, 6144 + 8192, 1, 6144), (2048, 51200, 0, 0), (704, 256, 0, 0), (256, 1024, 1, 0)])
def test_gemm_fp8_weights_decode_regime(lib, m, k, n, epi, gelu_from):
    """md_gemm_fp8w: exact products of bf16 activations with the e4m3 weights (the quantisation is the only
    approximation), so it is compared tightly with a torch evaluation over the DEQUANTISED weights; rows do
    not depend on how many rows are in flight; the quantisation error itself is reported against the bf16 layer."""
    from moondream_amd.weights import PackedLinearFp8

    a, w, b = randn(m, k, seed=80), randn(n, k, scale=1 / math.sqrt(k), seed=81), randn(n, scale=0.1, seed=82)
    lin = PackedLinear(w, b, "cuda")
    q = PackedLinearFp8(lin.w, lin.b, n, k)
    assert q.k_pad % 128 == 0 and q.w.numel() == q.n_pad * q.k_pad
    A = pad_k(a, lin.k_pad)

    def run(rows):
        c = torch.full((rows, q.n_pad), float("nan"), dtype=BF16, device="cuda")
        st = q.struct()
        _lib.check(lib.md_gemm_fp8w(A.data_ptr(), A.stride(0), C.byref(st), c.data_ptr(), c.stride(0), rows, epi, 1, gelu_from, stream()))
        torch.cuda.synchronize()
        return c

    c = run(m)
    deq = q.dequantized()[:n, :k]
    ref = (a.float() @ deq.t() + b.float()).to(BF16)
    if epi == 1:
        g = torch.nn.functional.gelu(ref.float(), approximate="tanh").to(BF16)
        ref = torch.cat([ref[:, :gelu_from], g[:, gelu_from:]], 1)
    compare(f"fp8w m{m} {k}x{n} epi{epi}", c[:, :n], ref, 3e-3, 2e-2)
    assert float(c[:, n:].float().abs().max() if q.n_pad > n else 0.0) == 0.0  # pad columns: zero weights, zero bias
    if m > 1:
        assert torch.equal(run(1), c[:1])
    # what the quantisation costs against the bf16 layer (per-channel scale, 3 mantissa bits): a few per cent
    full = ref_linear(a, w, b)
    if epi == 1:
        gfull = torch.nn.functional.gelu(full.float(), approximate="tanh").to(BF16)
        full = torch.cat([full[:, :gelu_from], gfull[:, gelu_from:]], 1)
    compare(f"fp8w vs bf16 layer m{m} {k}x{n}", c[:, :n], full, 6e-2)


@pytest.mark.parametrize("m,k,n,epi,gelu_from", [(64, 2048, 14336, 1, 6144), (1, 2048, 2048, 0, 0), (5, 256, 448, 1, 192), (33, 1024, 1024, 0, 0),
                                                 (8, 8192, 2048, 0, 0), (40, 128, 96, 0, 0)])
def test_gemm_int4_weight_stream_decode_regime(lib, m, k, n, epi, gelu_from):
    """md_gemm_fp8w over the reference's OWN 4-bit checkpoint format (md_linear_fp8.format = MD_WSTREAM_INT4_G128): the kernel
    rebuilds bf16(bf16(q - zero) * scale) in registers, so against an fp32 matmul with dequantize_int4's weights only the
    accumulation order differs (tolerance of the other decode-regime kernels), rows do not depend on the row count, padding
    channels come out as zero, and the result tracks the bf16 decode-regime kernel over the dequantised copy."""
    from moondream_amd.weights import PackedLinearInt4, dequantize_int4

    a = randn(m, k, seed=300)
    w = randn(n, k, scale=1 / math.sqrt(k), seed=301)
    b = randn(n, scale=0.1, seed=302)
    packed, scale, zero = quantize_int4(w, zero_shift=0.3)  # fractional zero points: the first rounding matters
    deq = dequantize_int4(packed, scale, zero, n).cuda()    # the bf16 weights every other launch multiplies with
    n_pad = (n + 63) // 64 * 64
    bp = torch.zeros(n_pad, dtype=BF16, device="cuda")
    bp[:n] = b
    q = PackedLinearInt4([(packed, scale, zero, n)], bp, "cuda")
    assert torch.equal(q.dequantized()[:n].cuda(), deq)

    def run(rows):
        c = torch.full((rows, q.n_pad), float("nan"), dtype=BF16, device="cuda")
        st = q.struct()
        _lib.check(lib.md_gemm_fp8w(a.data_ptr(), a.stride(0), C.byref(st), c.data_ptr(), c.stride(0), rows, epi, 1, gelu_from, stream()))
        torch.cuda.synchronize()
        return c

    c = run(m)
    ref = (a.float() @ deq.float().t() + b.float()).to(BF16)
    if epi == 1:
        g = torch.nn.functional.gelu(ref.float(), approximate="tanh").to(BF16)
        ref = torch.cat([ref[:, :gelu_from], g[:, gelu_from:]], 1)
    compare(f"int4 stream m{m} {k}x{n} epi{epi}", c[:, :n], ref, 3e-3, 2e-2)
    assert float(c[:, n:].float().abs().max() if q.n_pad > n else 0.0) == 0.0
    if m > 1:
        assert torch.equal(run(1), c[:1])
    # the bf16 decode-regime kernel over the dequantised copy: the same weights, another K order
    lin = PackedLinear(deq, b, "cuda")
    c16 = gemm(lib, pad_k(a, lin.k_pad), lin, epi=0)
    if epi == 0:
        compare(f"int4 stream vs bf16 stream m{m} {k}x{n}", c[:, :n], c16[:, :n], 3e-3, 2e-2)


@pytest.mark.parametrize("m,dim,ka,kb", [(64, 2048, 2048, 8192), (5, 1024, 1024, 4096), (33, 256, 256, 768)])
def test_int4_partial_pair_feeds_the_block_tail(lib, m, dim, ka, kb):
    """md_gemm_fp8w_partial_f32_pair with int4 streams: K-slice partials whose sum is the fp32 product with dequantize_int4's
    weights; row-subset invariant."""
    from moondream_amd.weights import PackedLinearInt4, dequantize_int4

    a1, w1 = randn(m, ka, seed=310), randn(dim, ka, scale=1 / math.sqrt(ka), seed=311)
    a2, w2 = randn(m, kb, seed=313), randn(dim, kb, scale=1 / math.sqrt(kb), seed=314)
    t1, t2 = quantize_int4(w1), quantize_int4(w2, zero_shift=-0.4)
    qa, qb = PackedLinearInt4([(*t1, dim)], None, "cuda"), PackedLinearInt4([(*t2, dim)], None, "cuda")
    d1, d2 = dequantize_int4(*t1, dim).cuda(), dequantize_int4(*t2, dim).cuda()
    sa, sb = qa.struct(), qb.struct()
    na, nb = lib.md_gemm_fp8w_partial_slices(C.byref(sa)), lib.md_gemm_fp8w_partial_slices(C.byref(sb))
    assert 1 <= na <= 8 and 1 <= nb <= 8

    def run(rows):
        pa = torch.full((na, rows, dim), float("nan"), dtype=torch.float32, device="cuda")
        pb = torch.full((nb, rows, dim), float("nan"), dtype=torch.float32, device="cuda")
        _lib.check(lib.md_gemm_fp8w_partial_f32_pair(a1.data_ptr(), a1.stride(0), C.byref(sa), pa.data_ptr(), a2.data_ptr(), a2.stride(0),
                                                     C.byref(sb), pb.data_ptr(), rows, dim, rows * dim, stream()))
        torch.cuda.synchronize()
        return pa, pb

    pa, pb = run(m)
    assert torch.isfinite(pa).all() and torch.isfinite(pb).all()
    compare("int4 partials a", pa.sum(0).to(BF16), (a1.float() @ d1.float().t()).to(BF16), 3e-3, 2e-2)
    compare("int4 partials b", pb.sum(0).to(BF16), (a2.float() @ d2.float().t()).to(BF16), 3e-3, 2e-2)
    if m > 1:
        p1a, p1b = run(1)
        assert torch.equal(p1a, pa[:, :1]) and torch.equal(p1b, pb[:, :1])


@pytest.mark.parametrize("m,dim,ka,kb", [(64, 2048, 2048, 8192), (5, 1024, 1024, 4096), (33, 256, 256, 704)])
def test_fp8_partial_pair_feeds_the_block_tail(lib, m, dim, ka, kb):
    """md_gemm_fp8w_partial_f32_pair: scaled K-slice partials whose sum is the fp32 product with the dequantised
    weights; md_reduce_residual_layernorm consumes them unchanged; row-subset invariant."""
    from moondream_amd.weights import PackedLinearFp8

    a1, w1, b1 = randn(m, ka, seed=90), randn(dim, ka, scale=1 / math.sqrt(ka), seed=91), randn(dim, scale=0.1, seed=92)
    a2, w2, b2 = randn(m, kb, seed=93), randn(dim, kb, scale=1 / math.sqrt(kb), seed=94), randn(dim, scale=0.1, seed=95)
    la, lb = PackedLinear(w1, b1, "cuda"), PackedLinear(w2, b2, "cuda")
    qa, qb = PackedLinearFp8(la.w, la.b, dim, ka), PackedLinearFp8(lb.w, lb.b, dim, kb)
    sa, sb = qa.struct(), qb.struct()
    na, nb = lib.md_gemm_fp8w_partial_slices(C.byref(sa)), lib.md_gemm_fp8w_partial_slices(C.byref(sb))
    assert 1 <= na <= 8 and 1 <= nb <= 8

    def run(rows):
        pa = torch.full((na, rows, dim), float("nan"), dtype=torch.float32, device="cuda")
        pb = torch.full((nb, rows, dim), float("nan"), dtype=torch.float32, device="cuda")
        A1, A2 = pad_k(a1[:rows], la.k_pad), pad_k(a2[:rows], lb.k_pad)
        _lib.check(lib.md_gemm_fp8w_partial_f32_pair(A1.data_ptr(), A1.stride(0), C.byref(sa), pa.data_ptr(), A2.data_ptr(), A2.stride(0),
                                                     C.byref(sb), pb.data_ptr(), rows, dim, rows * dim, stream()))
        torch.cuda.synchronize()
        return pa, pb

    pa, pb = run(m)
    assert torch.isfinite(pa).all() and torch.isfinite(pb).all()
    compare("fp8 partials a", pa.sum(0).to(BF16), (a1.float() @ qa.dequantized()[:dim, :ka].t()).to(BF16), 3e-3, 2e-2)
    compare("fp8 partials b", pb.sum(0).to(BF16), (a2.float() @ qb.dequantized()[:dim, :kb].t()).to(BF16), 3e-3, 2e-2)
    if m > 1:
        p1a, p1b = run(1)
        assert torch.equal(p1a, pa[:, :1]) and torch.equal(p1b, pb[:, :1])


@pytest.mark.parametrize("m,dim,ka,kb", [(64, 2048, 2048, 8192), (5, 1024, 1024, 4096), (1, 144, 144, 576), (33, 256, 256, 704)])
def test_block_tail_partials_then_reduce_residual_layernorm(lib, m, dim, ka, kb):
    """Decode-regime block tail: md_gemm_partial_f32 x 2 + md_reduce_residual_layernorm against the
    composition it replaces (two residual linears with bf16 roundings, then layer norm); the
    slice sums are exact fp32 restatements, so x is checked against a torch model of the same
    roundings; rows must not depend on how many rows are in flight."""
    a1, w1, b1 = randn(m, ka, seed=70), randn(dim, ka, scale=1 / math.sqrt(ka), seed=71), randn(dim, scale=0.1, seed=72)
    a2, w2, b2 = randn(m, kb, seed=73), randn(dim, kb, scale=1 / math.sqrt(kb), seed=74), randn(dim, scale=0.1, seed=75)
    x0 = randn(m, dim, seed=76)
    lw, lb = randn(dim, scale=0.1, seed=77) + 1.0, randn(dim, scale=0.1, seed=78)
    la, lbn = PackedLinear(w1, b1, "cuda"), PackedLinear(w2, b2, "cuda")
    ln = PackedLayerNorm(lw, lb, "cuda")

    def run(rows):
        sa, sb = la.struct(), lbn.struct()
        na, nb = lib.md_gemm_partial_slices(C.byref(sa)), lib.md_gemm_partial_slices(C.byref(sb))
        assert 1 <= na <= 8 and 1 <= nb <= 8
        ldp = dim
        pa = torch.full((na, rows, ldp), float("nan"), dtype=torch.float32, device="cuda")
        pb = torch.full((nb, rows, ldp), float("nan"), dtype=torch.float32, device="cuda")
        A1, A2 = pad_k(a1[:rows], la.k_pad), pad_k(a2[:rows], lbn.k_pad)
        _lib.check(lib.md_gemm_partial_f32(A1.data_ptr(), A1.stride(0), C.byref(sa), rows, pa.data_ptr(), ldp, rows * ldp, stream()))
        _lib.check(lib.md_gemm_partial_f32(A2.data_ptr(), A2.stride(0), C.byref(sb), rows, pb.data_ptr(), ldp, rows * ldp, stream()))
        x = x0[:rows].clone()
        ld = (dim + 63) // 64 * 64
        y = torch.zeros(rows, ld, dtype=BF16, device="cuda")
        st = ln.struct()
        _lib.check(lib.md_reduce_residual_layernorm(x.data_ptr(), dim, pa.data_ptr(), na, la.b.data_ptr(), pb.data_ptr(), nb,
                                                    lbn.b.data_ptr(), ldp, rows * ldp, y.data_ptr(), ld, C.byref(st), rows, dim,
                                                    1e-5, stream()))
        torch.cuda.synchronize()
        return x, y, pa, pb

    x, y, pa, pb = run(m)
    assert torch.isfinite(pa).all() and torch.isfinite(pb).all()
    # both layers in one launch: the same partials, bit for bit
    sa, sb = la.struct(), lbn.struct()
    pa2, pb2 = torch.full_like(pa, float("nan")), torch.full_like(pb, float("nan"))
    A1, A2 = pad_k(a1, la.k_pad), pad_k(a2, lbn.k_pad)
    _lib.check(lib.md_gemm_partial_f32_pair(A1.data_ptr(), A1.stride(0), C.byref(sa), pa2.data_ptr(), A2.data_ptr(), A2.stride(0),
                                            C.byref(sb), pb2.data_ptr(), m, dim, m * dim, stream()))
    torch.cuda.synchronize()
    assert torch.equal(pa2, pa) and torch.equal(pb2, pb)
    # the partials add up to the fp32 products
    compare("partials a", pa.sum(0).to(BF16), (a1.float() @ w1.float().t()).to(BF16), 3e-3, 2e-2)
    compare("partials b", pb.sum(0).to(BF16), (a2.float() @ w2.float().t()).to(BF16), 3e-3, 2e-2)
    # exact model of the roundings, from the kernel's own partials summed in slice order
    acc_a = torch.zeros_like(pa[0])
    for s_ in range(pa.shape[0]):
        acc_a = acc_a + pa[s_]
    acc_b = torch.zeros_like(pb[0])
    for s_ in range(pb.shape[0]):
        acc_b = acc_b + pb[s_]
    t1 = (acc_a + b1.float()).to(BF16)
    x1 = (x0[:m].float() + t1.float()).to(BF16)
    t2 = (acc_b + b2.float()).to(BF16)
    x2 = (x1.float() + t2.float()).to(BF16)
    assert torch.equal(x, x2)
    ref = torch.nn.functional.layer_norm(x2.float(), (dim,), lw.float(), lb.float(), 1e-5).to(BF16)
    compare("tail layernorm", y[:, :dim], ref, 2e-3, 1e-2)
    assert torch.count_nonzero(y[:, dim:]) == 0
    # batch invariance: row 0 alone gives the same bits
    xs, ys, _, _ = run(1)
    assert torch.equal(xs[0], x[0]) and torch.equal(ys[0], y[0])



@pytest.mark.parametrize("m", [64, 5])
def test_decode_block_chain_under_graph_replay(lib, m):
    """The decode regime's launch chain -- fused [qkv | fc1] GEMM with its in-launch split-K workspace (GELU on the fc1 columns),
    proj + fc2 as one launch of K-slice partials, the block tail (slice sums, bias, both residual adds, next layer norm) -- twice in
    a row (two "blocks": the second consumes the first's layer-norm output), captured into ONE hipGraph over fixed buffers and
    replayed with CHANGING inputs: replay(A), replay(B), replay(A) must equal the eager launches of the same inputs bit for bit.
    Round 3's rejected fragment-prefetch patch passed every kernel test and broke the model only under graph replay
    (profiles/r03_decode_gemm_where_the_time_goes.txt): this is the kernel-level test of that path -- state that survives a launch
    (split-K tickets, anything a kernel leaves behind for "the next launch") shows up as a replay that differs from eager."""
    dim, ff = 256, 1024
    n_fused = 3 * dim + ff
    blocks = []
    for b in range(2):
        w_f, b_f = randn(n_fused, dim, scale=1 / math.sqrt(dim), seed=900 + 10 * b), randn(n_fused, scale=0.1, seed=901 + 10 * b)
        w_p, b_p = randn(dim, dim, scale=1 / math.sqrt(dim), seed=902 + 10 * b), randn(dim, scale=0.1, seed=903 + 10 * b)
        w_2, b_2 = randn(dim, ff, scale=1 / math.sqrt(ff), seed=904 + 10 * b), randn(dim, scale=0.1, seed=905 + 10 * b)
        lw, lb = randn(dim, scale=0.1, seed=906 + 10 * b) + 1.0, randn(dim, scale=0.1, seed=907 + 10 * b)
        blocks.append((PackedLinear(w_f, b_f, "cuda"), PackedLinear(w_p, b_p, "cuda"), PackedLinear(w_2, b_2, "cuda"), PackedLayerNorm(lw, lb, "cuda")))
    x = torch.zeros(m, dim, dtype=BF16, device="cuda")       # residual stream (updated in place by the tail)
    h = torch.zeros(m, dim, dtype=BF16, device="cuda")       # ln(x): the fused GEMM's operand, rewritten by the tail
    fused = torch.zeros(m, n_fused, dtype=BF16, device="cuda")
    sf = blocks[0][0].struct()
    need = lib.md_gemm_workspace_bytes(C.byref(sf), m, 0)
    ws = torch.zeros(max(need, 16), dtype=torch.uint8, device="cuda")
    sp, s2 = blocks[0][1].struct(), blocks[0][2].struct()
    na, nb = lib.md_gemm_partial_slices(C.byref(sp)), lib.md_gemm_partial_slices(C.byref(s2))
    pa = torch.zeros(na, m, dim, dtype=torch.float32, device="cuda")
    pb = torch.zeros(nb, m, dim, dtype=torch.float32, device="cuda")

    def chain():
        for lf, lp, l2, ln in blocks:
            st = lf.struct()
            args = _lib.MdGemmArgs(h.data_ptr(), h.stride(0), st, fused.data_ptr(), fused.stride(0), None, 0, 0, m, _lib.MD_EPI_GELU, 0,
                                   3 * dim, ws.data_ptr() if need else None, need)
            _lib.check(lib.md_gemm_bf16(C.byref(args), stream()), "fused gemm")
            # (the "attention output" of this toy block is the q section of the fused row)
            spp, s22 = lp.struct(), l2.struct()
            _lib.check(lib.md_gemm_partial_f32_pair(fused.data_ptr(), fused.stride(0), C.byref(spp), pa.data_ptr(),
                                                    fused[:, 3 * dim :].data_ptr(), fused.stride(0), C.byref(s22), pb.data_ptr(),
                                                    m, dim, m * dim, stream()))
            stn = ln.struct()
            _lib.check(lib.md_reduce_residual_layernorm(x.data_ptr(), dim, pa.data_ptr(), na, lp.b.data_ptr(), pb.data_ptr(), nb,
                                                        l2.b.data_ptr(), dim, m * dim, h.data_ptr(), dim, C.byref(stn), m, dim, 1e-5, stream()))

    inputs = {k: (randn(m, dim, seed=950 + i), randn(m, dim, seed=960 + i)) for i, k in enumerate("AB")}

    def load(k):
        x.copy_(inputs[k][0])
        h.copy_(inputs[k][1])

    eager = {}
    for k in "AB":
        load(k)
        chain()
        torch.cuda.synchronize()
        eager[k] = (x.clone(), h.clone(), fused.clone())
    assert not torch.equal(eager["A"][0], eager["B"][0])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        load("A")
        with torch.cuda.graph(g, stream=side):
            chain()
        for rep, k in enumerate("ABAAB"):
            load(k)
            g.replay()
            side.synchronize()
            for name, got, want in zip(("x", "ln(x)", "fused row"), (x, h, fused), eager[k]):
                assert torch.equal(got, want), f"replay {rep} ({k}): {name} differs from the eager launches of the same inputs"
    torch.cuda.current_stream().wait_stream(side)
    # and the chain computes what it claims: block 0's fused row against torch (inputs A)
    load("A")
    chain()
    torch.cuda.synchronize()


@pytest.mark.parametrize("rows,dim", [(7, 144), (1458, 1152), (730, 2048), (3, 720), (5, 256)])
def test_layernorm(lib, rows, dim):
    x = randn(rows, dim, scale=3.0, seed=15) + 0.5
    w, b = randn(dim, scale=0.1, seed=16) + 1.0, randn(dim, scale=0.1, seed=17)
    ln = PackedLayerNorm(w, b, "cuda")
    ld = (dim + 63) // 64 * 64
    y = torch.zeros(rows, ld, dtype=BF16, device="cuda")
    st = ln.struct()
    _lib.check(lib.md_layernorm_bf16(x.data_ptr(), dim, y.data_ptr(), ld, C.byref(st), rows, dim, 1e-5, stream()))
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x.float(), (dim,), w.float(), b.float(), 1e-5).to(BF16)
    compare(f"layernorm {rows}x{dim}", y[:, :dim], ref, 2e-3, 1e-2)
    assert torch.count_nonzero(y[:, dim:]) == 0


def test_patchify_both_inputs(lib):
    from oracle import moondream_oracle as O

    rng = np.random.default_rng(0)
    crops = rng.integers(0, 256, (3, 378, 378, 3), dtype=np.uint8)
    lut = reference_pixel_lut()
    ref = O.patchify(O.normalize_crops(crops), 14)  # [3, 729, 588]
    out = torch.full((3 * 729, 640), float("nan"), dtype=BF16, device="cuda")
    d_crops, d_lut = torch.from_numpy(crops).cuda(), lut.cuda()
    _lib.check(lib.md_patchify_u8(d_crops.data_ptr(), d_lut.data_ptr(), out.data_ptr(), 640, 3, 378, 14, stream()))
    torch.cuda.synchronize()
    assert torch.equal(out[:, :588].cpu(), ref.reshape(-1, 588))
    assert torch.count_nonzero(out[:, 588:]) == 0
    chw = O.normalize_crops(crops).cuda()
    out2 = torch.full((3 * 729, 640), float("nan"), dtype=BF16, device="cuda")
    _lib.check(lib.md_patchify_bf16(chw.data_ptr(), out2.data_ptr(), 640, 3, 378, 14, stream()))
    torch.cuda.synchronize()
    assert torch.equal(out2, out)


def ref_attention(q, k, v, allowed, scale):
    """q [B,H,Tq,d], k/v [B,H,Tk,d] fp32 reference with bf16 probabilities (the oracle's model)."""
    s = (q.float() @ k.float().transpose(-1, -2)) * scale
    if allowed is not None:
        s = s.masked_fill(~allowed, float("-inf"))
    p = torch.exp(s - s.amax(-1, keepdim=True))
    l = p.sum(-1, keepdim=True)
    return ((p.to(BF16).float() @ v.float()) / l).to(BF16)


def run_prefill(lib, q, k, v, q_len, kv_len, prefix, pos0=None, kv_lens=None, o8=None, o8_inv_scale=1.0, bf16_out=True):
    """q [B,Tq,H,d]; k,v [B,H,Tk,d] (slab layout).  o8: uint8 [B,Tq,H*d] for the e4m3 copy of the output."""
    b, tq, h, d = q.shape
    tk = k.shape[2]
    o = torch.full((b, tq, h * d), float("nan"), dtype=BF16, device="cuda")
    a = _lib.MdAttnArgs()
    if o8 is not None:
        a.o8, a.o8_bs, a.o8_ts, a.o8_inv_scale = o8.data_ptr(), tq * h * d, h * d, o8_inv_scale
    a.q, a.q_bs, a.q_ts, a.q_hs = q.data_ptr(), tq * h * d, h * d, d
    a.k, a.k_bs, a.k_ts, a.k_hs = k.data_ptr(), h * tk * d, d, tk * d
    a.v, a.v_bs, a.v_ts, a.v_hs = v.data_ptr(), h * tk * d, d, tk * d
    a.o, a.o_bs, a.o_ts, a.o_hs = (o.data_ptr() if bf16_out else None), tq * h * d, h * d, d
    a.batch, a.n_heads, a.n_kv_heads, a.head_dim = b, h, h, d
    a.q_len, a.kv_len_all = q_len, kv_len
    a.q_pos0 = pos0.data_ptr() if pos0 is not None else None
    a.kv_len = kv_lens.data_ptr() if kv_lens is not None else None
    a.prefix_len, a.scale = prefix, 1.0 / math.sqrt(d)
    _lib.check(lib.md_attention_prefill(C.byref(a), stream()), "attn")
    torch.cuda.synchronize()
    return o.view(b, tq, h, d)


@pytest.mark.parametrize("hd,t", [(72, 729), (72, 100), (64, 730), (64, 64), (64, 129)])
def test_attention_no_mask(lib, hd, t):
    b, h = 2, 3
    q, k, v = randn(b, t, h, hd, seed=20), randn(b, h, t, hd, seed=21), randn(b, h, t, hd, seed=22)
    o = run_prefill(lib, q, k, v, t, t, prefix=t)
    ref = ref_attention(q.permute(0, 2, 1, 3), k, v, None, 1 / math.sqrt(hd)).permute(0, 2, 1, 3)
    compare(f"attn hd{hd} t{t}", o, ref, 6e-3, 4e-2)


@pytest.mark.parametrize("hd,t", [(72, 729), (64, 730), (64, 77)])
def test_attention_prefill_e4m3_output_equals_quantise_pass(lib, hd, t):
    """md_attn_args.o8 (opt-in FP8 mode): the e4m3 copy written by the attention epilogue is bit-identical to md_quantize_f8
    of the bf16 output, with and without the bf16 output itself."""
    b, h = 2, 4
    q, k, v = randn(b, t, h, hd, seed=20), randn(b, h, t, hd, seed=21), randn(b, h, t, hd, seed=22)
    inv = 1.0 / 0.013
    o = run_prefill(lib, q, k, v, t, t, prefix=t)
    want = torch.zeros(b * t, h * hd, dtype=torch.uint8, device="cuda")
    o2 = o.reshape(b * t, h * hd)
    _lib.check(lib.md_quantize_f8(o2.data_ptr(), h * hd, want.data_ptr(), h * hd, b * t, h * hd, h * hd, inv, stream()))
    for bf16_out in (True, False):
        got = torch.full((b, t, h * hd), 0x7F, dtype=torch.uint8, device="cuda")
        o_b = run_prefill(lib, q, k, v, t, t, prefix=t, o8=got, o8_inv_scale=inv, bf16_out=bf16_out)
        torch.cuda.synchronize()
        assert torch.equal(got.view(b * t, h * hd), want), bf16_out
        if bf16_out:
            assert torch.equal(o_b, o)


@pytest.mark.parametrize("hd,t,tq,pos", [(72, 729, 729, 0), (64, 735, 735, 0), (64, 730, 5, 725), (64, 730, 10, 700), (64, 97, 97, 0), (72, 96, 96, 0), (64, 65, 65, 0), (64, 33, 33, 0)])
def test_attention_dead_half_tile_skip_is_exact(lib, hd, t, tq, pos):
    """Round 5: when at most 32 keys of the LAST 64-key tile exist, the kernel skips the tile's second half (its scores are -inf,
    its probabilities exactly 0, its contributions exact zeros).  Same bits as the full computation -- for lengths that end in the
    first half of a tile (729, 735, 97, 65: the skip fires), exactly at the half (96), in the second half (33 + ...: no skip), with
    the prefix-LM rule and a causal tail (730 keys, 10 causal queries at positions 700..709: the bound is what the block's last row
    may see, not kv_len).  Second switch of the same kind: a wave whose 32 query rows all lie past q_len skips its
    arithmetic.  Every combination of the two (3 = default, 0 = round 4's kernel, 1, 2) gives the same bits."""
    b, h = 2, 3
    q, k, v = randn(b, tq, h, hd, seed=30), randn(b, h, t, hd, seed=31), randn(b, h, t, hd, seed=32)
    outs = []
    pos0 = torch.full((b,), pos, dtype=torch.int32, device="cuda") if pos else None
    for skip in (3, 0, 1, 2):
        _lib.check(lib.md_gemm_set_tuning(b"attn_skip_dead", skip))
        outs.append(run_prefill(lib, q, k, v, tq, t, prefix=min(t, 64) if hd == 64 else t, pos0=pos0).clone())
    _lib.check(lib.md_gemm_set_tuning(b"attn_skip_dead", 3))
    assert all(torch.equal(outs[0], o) for o in outs[1:])


def test_attention_spiky_scores_force_rescale(lib):
    """One key dominates late in the sequence: the online-softmax rescale path."""
    b, h, t, hd = 1, 2, 300, 72
    q, k, v = randn(b, t, h, hd, seed=23), randn(b, h, t, hd, seed=24), randn(b, h, t, hd, seed=25)
    k[:, :, 250] = q[:, 7].clone() * 4.0  # q row 7 . k row 250 >> everything else
    o = run_prefill(lib, q, k, v, t, t, prefix=t)
    ref = ref_attention(q.permute(0, 2, 1, 3), k, v, None, 1 / math.sqrt(hd)).permute(0, 2, 1, 3)
    compare("attn spiky", o, ref, 6e-3, 4e-2)


@pytest.mark.parametrize("q_len,pos", [(730, 0), (5, 730), (32, 730), (1, 735), (200, 650)])
def test_attention_prefix_lm_against_slab(lib, q_len, pos):
    """Decoder prefill: queries at pos..pos+q_len-1 against keys [0, pos+q_len) of a
    2048-slot slab with the prefix-LM rule (prefix 730), per-sequence positions."""
    from oracle.moondream_oracle import prefix_lm_allowed

    b, h, hd, ctx, prefix = 2, 4, 64, 2048, 730
    q = randn(b, q_len, h, hd, seed=26)
    k, v = randn(b, h, ctx, hd, seed=27), randn(b, h, ctx, hd, seed=28)
    pos0 = torch.tensor([pos, max(pos - 3, 0)], dtype=torch.int32, device="cuda")
    kv_lens = pos0 + q_len
    o = run_prefill(lib, q, k, v, q_len, 0, prefix, pos0, kv_lens)
    for bi in range(b):
        p0 = int(pos0[bi])
        allowed = prefix_lm_allowed(torch.arange(p0, p0 + q_len), p0 + q_len, prefix).cuda()
        ref = ref_attention(q[bi].permute(1, 0, 2), k[bi, :, : p0 + q_len], v[bi, :, : p0 + q_len], allowed, 1 / 8.0)
        compare(f"prefix-lm q{q_len} pos{p0}", o[bi], ref.permute(1, 0, 2), 6e-3, 4e-2)


def test_attention_decode(lib):
    b, h, hd, ctx = 3, 4, 64, 2048
    q = randn(b, h * hd * 3, seed=29)  # rows with a leading dimension like the fused qkv activation
    k, v = randn(b, h, ctx, hd, seed=30), randn(b, h, ctx, hd, seed=31)
    lens = torch.tensor([736, 1, 2048], dtype=torch.int32, device="cuda")
    o = torch.full((b, h * hd), float("nan"), dtype=BF16, device="cuda")
    _lib.check(lib.md_attention_decode(q.data_ptr(), q.stride(0), o.data_ptr(), h * hd, k.data_ptr(), v.data_ptr(),
                                       h * ctx * hd, ctx, lens.data_ptr(), b, h, h, hd, 0.125, stream()))
    torch.cuda.synchronize()
    for bi in range(b):
        n = int(lens[bi])
        qq = q[bi, : h * hd].view(h, 1, hd)
        ref = ref_attention(qq, k[bi, :, :n], v[bi, :, :n], None, 0.125)
        compare(f"decode attn len{n}", o[bi].view(h, 1, hd), ref, 6e-3, 4e-2)


def test_decode_attention_with_fused_rope_equals_two_kernels(lib):
    """md_attention_decode_rope == md_rope_kv_write + md_attention_decode, bit for bit
    (outputs and the KV rows written at the new position)."""
    b, h, hd, ctx, rot = 5, 4, 64, 2048, 32
    qkv = randn(b, 3 * h * hd + 64, seed=60)  # leading dimension larger than the row
    ld = qkv.stride(0)
    freqs = rope_table(rot, ctx).cuda()
    pos0 = torch.tensor([730, 0, 1, 2047, 915], dtype=torch.int32, device="cuda")
    lens = pos0 + 1
    k0, v0 = randn(b, h, ctx, hd, seed=61), randn(b, h, ctx, hd, seed=62)
    # two kernels
    qa, ka, va = qkv.clone(), k0.clone(), v0.clone()
    oa = torch.zeros(b, h * hd, dtype=BF16, device="cuda")
    _lib.check(lib.md_rope_kv_write(qa.data_ptr(), ld, freqs.data_ptr(), pos0.data_ptr(), ka.data_ptr(), va.data_ptr(),
                                    h * ctx * hd, ctx, b, 1, h, h, hd, rot, stream()))
    _lib.check(lib.md_attention_decode(qa.data_ptr(), ld, oa.data_ptr(), h * hd, ka.data_ptr(), va.data_ptr(),
                                       h * ctx * hd, ctx, lens.data_ptr(), b, h, h, hd, 0.125, stream()))
    # fused
    qb, kb, vb = qkv.clone(), k0.clone(), v0.clone()
    ob = torch.zeros(b, h * hd, dtype=BF16, device="cuda")
    _lib.check(lib.md_attention_decode_rope(qb.data_ptr(), ld, ob.data_ptr(), h * hd, freqs.data_ptr(), kb.data_ptr(),
                                            vb.data_ptr(), h * ctx * hd, ctx, lens.data_ptr(), b, h, hd, rot, 0.125, stream()))
    torch.cuda.synchronize()
    assert torch.equal(oa, ob)
    assert torch.equal(ka, kb) and torch.equal(va, vb)
    assert torch.equal(qb, qkv)  # the fused kernel leaves the activation untouched


def test_decode_attention_launch_shapes_agree_bitwise(lib):
    """The decode kernel runs 16 waves per (sequence, head) for small launches and 4 for large ones; the class-
    partitioned accumulation order makes both the same function: a sequence's output row is bit-identical
    whether it is decoded alone or inside a large batch (fused-rope form too)."""
    h, hd, ctx, rot = 4, 64, 2048, 32
    small, big = 5, 160  # 20 and 640 (sequence, head) pairs: either side of the launch-shape switch
    lens_small = torch.tensor([736, 1, 2048, 129, 915], dtype=torch.int32, device="cuda")
    reps = big // small
    qkv = randn(small, 3 * h * hd, seed=70)
    k, v = randn(small, h, ctx, hd, seed=71), randn(small, h, ctx, hd, seed=72)
    freqs = rope_table(rot, ctx).cuda()

    def run(n_rep, fused):
        q_, k_, v_ = qkv.repeat(n_rep, 1), k.repeat(n_rep, 1, 1, 1), v.repeat(n_rep, 1, 1, 1)
        lens = lens_small.repeat(n_rep)
        o = torch.zeros(small * n_rep, h * hd, dtype=BF16, device="cuda")
        if fused:
            _lib.check(lib.md_attention_decode_rope(q_.data_ptr(), q_.stride(0), o.data_ptr(), h * hd, freqs.data_ptr(),
                                                    k_.data_ptr(), v_.data_ptr(), h * ctx * hd, ctx, lens.data_ptr(),
                                                    small * n_rep, h, hd, rot, 0.125, stream()))
        else:
            _lib.check(lib.md_attention_decode(q_.data_ptr(), q_.stride(0), o.data_ptr(), h * hd, k_.data_ptr(), v_.data_ptr(),
                                               h * ctx * hd, ctx, lens.data_ptr(), small * n_rep, h, h, hd, 0.125, stream()))
        torch.cuda.synchronize()
        return o, k_, v_

    for fused in (False, True):
        o1, k1, v1 = run(1, fused)
        o2, k2, v2 = run(reps, fused)
        assert torch.isfinite(o1.float()).all()
        for r in range(reps):
            assert torch.equal(o2[r * small:(r + 1) * small], o1), (fused, r)
        assert torch.equal(k2[:small], k1) and torch.equal(v2[:small], v1)


def test_rope_kv_write(lib):
    from oracle.moondream_oracle import apply_rope, rope_table as o_rope_table

    b, t, h, hd, ctx, rot = 2, 9, 4, 64, 256, 32
    qkv = randn(b * t, 3 * h * hd, seed=32)
    orig = qkv.clone()
    freqs = rope_table(rot, ctx).cuda()
    pos0 = torch.tensor([100, 7], dtype=torch.int32, device="cuda")
    ks = torch.zeros(b, h, ctx, hd, dtype=BF16, device="cuda")
    vs = torch.zeros_like(ks)
    _lib.check(lib.md_rope_kv_write(qkv.data_ptr(), qkv.stride(0), freqs.data_ptr(), pos0.data_ptr(), ks.data_ptr(),
                                    vs.data_ptr(), h * ctx * hd, ctx, b, t, h, h, hd, rot, stream()))
    torch.cuda.synchronize()
    cos, sin = o_rope_table(rot // 2, ctx)
    assert torch.equal(torch.stack([cos, sin], -1), freqs.cpu())  # same table as the oracle
    o = orig.cpu().view(b, t, 3, h, hd)
    for bi in range(b):
        pos = torch.arange(int(pos0[bi]), int(pos0[bi]) + t)
        qr = apply_rope(o[bi, :, 0].permute(1, 0, 2), cos, sin, pos, rot)
        kr = apply_rope(o[bi, :, 1].permute(1, 0, 2), cos, sin, pos, rot)
        got_q = qkv.cpu().view(b, t, 3, h, hd)[bi, :, 0].permute(1, 0, 2)
        assert torch.equal(got_q, qr)
        assert torch.equal(ks.cpu()[bi][:, pos], kr)
        assert torch.equal(vs.cpu()[bi][:, pos], o[bi, :, 2].permute(1, 0, 2))
    assert torch.equal(qkv.cpu().view(b, t, 3, h, hd)[:, :, 1:], o[:, :, 1:])  # k, v left in place untouched


def test_embed_and_argmax(lib):
    vocab, dim = 51200, 2048
    table = randn(vocab, dim, seed=33)
    ids = torch.tensor([0, 51199, 7, 7, 12345], dtype=torch.int32, device="cuda")
    out = torch.empty(5, dim, dtype=BF16, device="cuda")
    _lib.check(lib.md_embed_tokens(ids.data_ptr(), table.data_ptr(), dim, out.data_ptr(), dim, 5, dim, stream()))
    torch.cuda.synchronize()
    assert torch.equal(out, table[ids.long()])
    logits = randn(4, vocab, seed=34)
    logits[0, 100] = logits[0, 40000] = 50.0  # tie -> lowest index
    logits[1, 3] = 60.0  # suppressed
    logits[1, 77] = 55.0
    logits[2, vocab - 1] = 70.0
    nxt = torch.zeros(4, dtype=torch.int32, device="cuda")
    _lib.check(lib.md_argmax_bf16(logits.data_ptr(), vocab, 4, vocab, 3, nxt.data_ptr(), stream()))
    torch.cuda.synchronize()
    ref = logits.float().clone()
    ref[:, 3] = float("-inf")
    assert nxt.tolist()[:3] == [100, 77, vocab - 1]
    assert int(nxt[3]) == int(torch.argmax(ref[3]))


@pytest.mark.parametrize("th,tw", [(1, 1), (2, 3), (3, 4), (4, 1)])
def test_stitch_pool_concat(lib, th, tw):
    from oracle import moondream_oracle as O

    dim, g, margin = 144, 27, 4
    feats = randn(1 + th * tw, g * g, dim, seed=35)
    out = torch.full((g * g, 2 * dim + 32), float("nan"), dtype=BF16, device="cuda")
    _lib.check(lib.md_stitch_pool_concat(feats.data_ptr(), out.data_ptr(), out.stride(0), dim, g, margin, th, tw, stream()))
    torch.cuda.synchronize()
    f = feats.cpu()
    stitched = O.stitch_local_features(f[1:].reshape(-1, g, g, dim), (th, tw), margin)
    pooled = O.adaptive_avg_pool_hw(stitched, g).reshape(g * g, dim)
    assert torch.equal(out[:, :dim].cpu(), f[0])
    compare(f"stitch_pool {th}x{tw}", out[:, dim : 2 * dim], pooled, 2e-3, 1e-2)


# ---------------------------------------------------------------------------
# sampling and the region head's device-resident step
def _sample(lib, logits, temperature, top_p, u, suppress=-1, want_probs=True):
    b, v = logits.shape
    nxt = torch.full((b,), -1, dtype=torch.int32, device="cuda")
    probs = torch.full((b, v), float("nan"), dtype=BF16, device="cuda") if want_probs else None
    _lib.check(lib.md_sample_top_p(logits.data_ptr(), logits.stride(0), b, v, suppress, temperature, top_p, u.data_ptr(),
                                   nxt.data_ptr(), probs.data_ptr() if want_probs else None, v if want_probs else 0, stream()))
    torch.cuda.synchronize()
    return nxt, probs


def test_sample_top_p_filter_matches_reference(lib):
    """md_sample_top_p's filtered distribution against the REFERENCE's softmax + _apply_top_p
    (moondream.py:270-278,526-527) on fixed logits (tests/golden/sampling_top_p.npz): same support
    and the same bf16 values; the only licence is the order of exactly tied probabilities at the
    nucleus boundary (torch.sort is not stable) and a last-bit difference of exp()."""
    from util import bits_to_bf16

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sampling_top_p.npz"))
    for i in range(int(g["n_cases"])):
        logits = bits_to_bf16(g[f"case{i}.logits"]).cuda()
        want = bits_to_bf16(g[f"case{i}.kept"]).float()
        temp, top_p = float(g[f"case{i}.temperature"]), float(g[f"case{i}.top_p"])
        u = torch.full((logits.shape[0],), 0.5, device="cuda")
        nxt, probs = _sample(lib, logits, temp, top_p, u)
        got = probs.float().cpu()
        for r in range(logits.shape[0]):
            ks, kw = got[r] > 0, want[r] > 0
            # support: identical, or differing only inside one tied probability value / by <= 2 boundary tokens
            if not torch.equal(ks, kw):
                diff = (ks != kw).nonzero().flatten()
                pv = bits_to_bf16(g[f"case{i}.probs"]).float()[r][diff]
                assert int(ks.sum()) == int(kw.sum()) and pv.unique().numel() == 1 or diff.numel() <= 2, (i, r, diff.numel())
            both = ks & kw
            rel = ((got[r][both] - want[r][both]).abs() / want[r][both]).max()
            assert float(rel) <= 2.0 ** -7, (i, r, float(rel))  # one bf16 ulp
            assert bool(ks[int(nxt[r])]), "the drawn token must be inside the nucleus"


def test_sample_top_p_distribution_and_suppression(lib):
    """Draws follow the filtered distribution: 20000 sequences share one logits row whose nucleus has
    a handful of tokens; empirical frequencies within 5 sigma of q.  The suppressed id is never drawn."""
    v = 2048
    g = torch.Generator().manual_seed(3)
    row = (torch.randn(v, generator=g) * 0.5 - 4.0).to(BF16)  # background: ~4 % of the mass
    hot = torch.tensor([5, 77, 300, 301, 1024, 2000])
    row[hot] = torch.tensor([6.0, 5.5, 5.0, 5.0, 4.5, 7.0]).to(BF16)  # id 2000 is the best and will be suppressed
    n = 20000
    logits = row.cuda().repeat(n, 1).contiguous()
    u = torch.rand(n, generator=torch.Generator().manual_seed(4)).cuda()
    nxt, probs = _sample(lib, logits, 1.0, 0.8, u, suppress=2000)
    q = probs[0].float().cpu()
    assert q[2000] == 0 and abs(float(q.sum()) - 1.0) < 2e-2
    support = (q > 0).nonzero().flatten()
    assert 3 <= support.numel() <= 64
    counts = torch.bincount(nxt.cpu().long(), minlength=v).float()
    assert counts[q == 0].sum() == 0
    qn = q / q.sum()
    sigma = (n * qn * (1 - qn)).sqrt()
    assert ((counts - n * qn).abs() <= 5 * sigma + 1)[support].all(), (counts[support], n * qn[support])
    # u -> 0 picks the lowest surviving id, u -> 1 the highest (inverse CDF over ids)
    lo, _ = _sample(lib, logits[:1], 1.0, 0.8, torch.zeros(1, device="cuda"), suppress=2000, want_probs=False)
    hi, _ = _sample(lib, logits[:1], 1.0, 0.8, torch.full((1,), 0.999999, device="cuda"), suppress=2000, want_probs=False)
    assert int(lo[0]) == int(support.min()) and int(hi[0]) == int(support.max())


def test_region_pick_encode_and_fourier(lib):
    """argmax bin (ties -> lowest) -> table value -> [cos | sin] Fourier features, against torch ops
    written like region.py:12-29 / moondream.py:672-713."""
    b, n_bins, half = 5, 1024, 128
    g = torch.Generator().manual_seed(9)
    for groups in (1, 2):
        logits = torch.randn(b, groups * n_bins, generator=g).to(BF16)
        logits[0, 17] = logits[0, 900] = 9.0                      # tie: lowest index wins
        table = (torch.arange(n_bins) / n_bins).to(BF16) if groups == 1 else \
            torch.pow(2.0, (torch.arange(n_bins).float() / 1023.0) * 10.0 - 10.0).to(BF16)
        w = (torch.randn(groups, half, generator=g) * 2.0).to(BF16)
        bins = torch.full((b, groups), -1, dtype=torch.int32, device="cuda")
        feats = torch.empty(b, 2 * half, dtype=BF16, device="cuda")
        lg, tb, wd = logits.cuda(), table.cuda(), w.cuda()
        _lib.check(lib.md_region_pick_encode(lg.data_ptr(), lg.stride(0), b, groups, n_bins, tb.data_ptr(), wd.data_ptr(), half,
                                             bins.data_ptr(), groups, feats.data_ptr(), 2 * half, stream()))
        torch.cuda.synchronize()
        want_bins = torch.stack([logits[:, k * n_bins:(k + 1) * n_bins].float().argmax(-1) for k in range(groups)], 1)
        assert bins.cpu().tolist() == want_bins.tolist()
        assert bins[0, 0] == 17
        x = table[want_bins]                                       # bf16 [b, groups]
        f = 2 * math.pi * x @ w                                    # region.py:28, bf16 ops on the CPU
        want = torch.cat([f.cos(), f.sin()], dim=-1)
        err = (feats.float().cpu() - want.float()).abs().max()
        assert float(err) <= 2.0 ** -7, float(err)                 # cos/sin values, one bf16 ulp at 1.0
        out2 = torch.empty(b, 2 * half, dtype=BF16, device="cuda")
        xd = x.cuda().contiguous()
        _lib.check(lib.md_fourier_features(xd.data_ptr(), xd.stride(0), b, groups, wd.data_ptr(), half, out2.data_ptr(), 2 * half, stream()))
        torch.cuda.synchronize()
        assert torch.equal(out2, feats)


# ---------------------------------------------------------------------------------------------------------------------
# FP8 mode (opt-in; BASELINE configs[4]): md_gemm_f8 and its activation producers.  Kernel-EXACT checks: the products
# of two e4m3 values are exact in fp32, so the kernel must agree with an fp32 matmul of the DEQUANTISED operands up to
# fp32 summation order (then one bf16 rounding).
# ---------------------------------------------------------------------------------------------------------------------
F8 = torch.float8_e4m3fn


def quant_rows(x, scale):
    """what md_quantize_f8 computes: fp8(sat(x / scale))"""
    return (x.float() / scale).clamp(-448.0, 448.0).to(F8)


def gemm_f8(lib, a8, a_scale, lin8, m, epi=0, r=None, res_row_mod=0, store_pad=0, gelu_from=0, c8_from=None, c8_inv_scale=1.0):
    width = lin8.n_pad if store_pad else lin8.n
    c = torch.full((m, width), float("nan"), dtype=BF16, device="cuda")
    c8 = None
    if c8_from is not None:
        c8 = torch.full((m, width - c8_from), 0x7F, dtype=torch.uint8, device="cuda")
    args = _lib.MdGemmF8Args(a8.data_ptr(), a8.stride(0), float(a_scale), lin8.struct(), c.data_ptr(), c.stride(0),
                             c8.data_ptr() if c8 is not None else None, c8.stride(0) if c8 is not None else 0, float(c8_inv_scale),
                             int(c8_from or 0), r.data_ptr() if r is not None else None, r.stride(0) if r is not None else 0,
                             res_row_mod, m, epi, store_pad, gelu_from)
    _lib.check(lib.md_gemm_f8(C.byref(args), stream()), "md_gemm_f8")
    torch.cuda.synchronize()
    return c, c8


def f8_case(m, k, n, seed):
    from moondream_amd.weights import PackedLinearF8

    x = randn(m, k, seed=seed)
    w, b = randn(n, k, scale=1 / math.sqrt(k), seed=seed + 1), randn(n, scale=0.1, seed=seed + 2)
    lin = PackedLinear(w, b, "cuda")
    lin8 = PackedLinearF8(lin.w, lin.b, n, k)
    a_scale = float(x.float().abs().max()) / 448.0
    a8 = torch.zeros(m, lin.k_pad, dtype=torch.uint8, device="cuda")
    a8[:, :k] = quant_rows(x, a_scale).view(torch.uint8)
    a_deq = a8.view(F8).float() * a_scale
    return x, a8, a_scale, a_deq, lin, lin8


@pytest.mark.parametrize("m,k,n", [(300, 588, 1152), (777, 1152, 3456), (1000, 2048, 6144), (2917, 4304, 1152), (64, 2048, 1024), (1, 256, 64)])
def test_gemm_f8_bias_exact_against_dequantised_operands(lib, m, k, n):
    x, a8, a_scale, a_deq, lin, lin8 = f8_case(m, k, n, 40)
    want = (a_deq @ lin8.dequantized()[:n].t() + lin.b[:n].float()).to(BF16)
    got, _ = gemm_f8(lib, a8, a_scale, lin8, m)
    compare(f"gemm_f8 bias {m}x{k}x{n} vs dequantised fp32", got, want, 2e-3, 1.5e-2)
    # and the quantisation itself costs what e4m3 operands cost (3 mantissa bits each): a few percent against bf16
    compare(f"gemm_f8 bias {m}x{k}x{n} vs the bf16 layer", got, ref_linear(x, lin.w[:n, :k], lin.b[:n]), 8e-2)


def test_gemm_f8_identity_detects_transposes(lib):
    from moondream_amd.weights import PackedLinearF8

    n = k = 512
    w = ((torch.arange(n * k, dtype=torch.float32).reshape(n, k) % 13) - 6).to(BF16).cuda()  # small integers: exact in e4m3
    lin8 = PackedLinearF8(w, torch.zeros(n, dtype=BF16, device="cuda"), n, k)
    lin8.q = w.to(F8).contiguous()               # the integers themselves as e4m3 codes ...
    lin8.scale = torch.ones_like(lin8.scale)     # ... with unit channel scales: every product and sum is exact
    a8 = torch.eye(k, device="cuda").to(F8).view(torch.uint8).contiguous()
    got, _ = gemm_f8(lib, a8, 1.0, lin8, k)
    assert torch.equal(got, w.t().contiguous())


def test_gemm_f8_gelu_residual_and_fp8_output(lib):
    m, k, n = 1500, 1152, 4304
    x, a8, a_scale, a_deq, lin, lin8 = f8_case(m, k, n, 50)
    pre = (a_deq @ lin8.dequantized().t() + lin.b.float()).to(BF16)  # padded columns: zero weights, zero bias
    gelu = torch.nn.functional.gelu(pre.float(), approximate="tanh").to(BF16)
    # GELU from column 1152 on (the fused [qkv | fc1] form), bf16 below it and fp8 from it on, padded columns stored
    out_scale = float(gelu.float().abs().max()) / 448.0
    got, got8 = gemm_f8(lib, a8, a_scale, lin8, m, epi=1, store_pad=1, gelu_from=1152, c8_from=1152, c8_inv_scale=1.0 / out_scale)
    compare("gemm_f8 bf16 part (no GELU below gelu_from)", got[:, :1152], pre[:, :1152], 2e-3, 1.5e-2)
    want8 = quant_rows(gelu[:, 1152:], out_scale)
    g8, w8 = got8.view(F8).float(), want8.float()
    assert torch.isfinite(g8).all()
    compare("gemm_f8 fp8 part (dequantised)", g8 * out_scale, w8 * out_scale, 2e-2)  # a bf16 1-ulp flip upstream can move an fp8 code
    assert float((g8 != w8).float().mean()) < 0.02
    assert torch.equal(got8[:, n - 1152 :], torch.zeros_like(got8[:, n - 1152 :]))  # K padding of the consumer: 0x00
    # residual epilogue, rows of the second operand taken modulo 729 (the ViT's position embedding form)
    r = randn(729, lin.n_pad, seed=53)
    got_r, _ = gemm_f8(lib, a8, a_scale, lin8, m, epi=2, r=r, res_row_mod=729)
    want_r = (r[torch.arange(m, device="cuda") % 729][:, :n].float() + pre[:, :n].float()).to(BF16)
    compare("gemm_f8 residual", got_r, want_r, 2e-3, 1.5e-2)


def test_quantize_layernorm_amax_f8(lib):
    rows, dim, dim_pad = 1000, 1152, 1152 + 64
    x = randn(rows, dim, scale=3.0, seed=60)
    x[5, 7] = 1000.0  # saturates
    scale = 0.05
    y = torch.full((rows, dim_pad), 0x7F, dtype=torch.uint8, device="cuda")
    _lib.check(lib.md_quantize_f8(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), rows, dim, dim_pad, 1.0 / scale, stream()))
    torch.cuda.synchronize()
    want = quant_rows(x, scale).view(torch.uint8)
    assert torch.equal(y[:, :dim], want) and int(y[:, dim:].max()) == 0
    assert float(y.view(F8)[5, 7].float()) == 448.0
    ln = PackedLayerNorm(1.0 + randn(dim, scale=0.1, seed=61), randn(dim, scale=0.1, seed=62), "cuda")
    y2 = torch.full((rows, dim_pad), 0x7F, dtype=torch.uint8, device="cuda")
    st = ln.struct()
    _lib.check(lib.md_layernorm_f8(x.data_ptr(), x.stride(0), y2.data_ptr(), y2.stride(0), C.byref(st), rows, dim, dim_pad, 1e-5, 1.0 / scale, stream()))
    yb = torch.empty(rows, dim, dtype=BF16, device="cuda")
    _lib.check(lib.md_layernorm_bf16(x.data_ptr(), x.stride(0), yb.data_ptr(), yb.stride(0), C.byref(st), rows, dim, 1e-5, stream()))
    torch.cuda.synchronize()
    assert torch.equal(y2[:, :dim], quant_rows(yb, scale).view(torch.uint8)) and int(y2[:, dim:].max()) == 0
    # other row widths: two pairs of chunks per lane, an odd chunk count, padding wider than a pair
    for rows_i, dim_i, pad_i in ((77, 2048, 2048), (130, 720, 768), (9, 88, 128), (5, 3000, 3008)):
        xi = randn(rows_i, dim_i, scale=2.0, seed=63)
        lni = PackedLayerNorm(1.0 + randn(dim_i, scale=0.1, seed=64), randn(dim_i, scale=0.1, seed=65), "cuda")
        sti = lni.struct()
        yi = torch.full((rows_i, pad_i), 0x7F, dtype=torch.uint8, device="cuda")
        _lib.check(lib.md_layernorm_f8(xi.data_ptr(), xi.stride(0), yi.data_ptr(), yi.stride(0), C.byref(sti), rows_i, dim_i, pad_i, 1e-5, 1.0 / scale, stream()))
        ybi = torch.empty(rows_i, dim_i, dtype=BF16, device="cuda")
        _lib.check(lib.md_layernorm_bf16(xi.data_ptr(), xi.stride(0), ybi.data_ptr(), ybi.stride(0), C.byref(sti), rows_i, dim_i, 1e-5, stream()))
        torch.cuda.synchronize()
        assert torch.equal(yi[:, :dim_i], quant_rows(ybi, scale).view(torch.uint8)), dim_i
        assert pad_i == dim_i or int(yi[:, dim_i:].max()) == 0, dim_i
    am = torch.zeros(1, dtype=torch.float32, device="cuda")
    x[9, 100] = float("inf")
    _lib.check(lib.md_amax_bf16(x.data_ptr(), x.stride(0), rows, dim, am.data_ptr(), stream()))
    torch.cuda.synchronize()
    assert float(am[0]) == 1000.0


def test_decode_attention_over_e4m3_kv_cache(lib):
    """md_attention_decode_rope_f8 / md_kv_quantize_f8: the e4m3 copy equals torch's conversion of the bf16 slab / scale, the
    new token's row lands in BOTH copies (bf16 rows bit-equal to the bf16 kernel's), and the attention output equals -- to
    fp32 summation order -- the softmax attention computed in torch over the DEQUANTISED cache with the same bf16
    probabilities."""
    b, h, hd, ctx, rot = 5, 4, 64, 2048, 32
    qkv = randn(b, 3 * h * hd + 64, seed=80)
    ld = qkv.stride(0)
    freqs = rope_table(rot, ctx).cuda()
    pos0 = torch.tensor([730, 0, 1, 2047, 915], dtype=torch.int32, device="cuda")
    lens = pos0 + 1
    k0, v0 = randn(b, h, ctx, hd, scale=2.0, seed=81), randn(b, h, ctx, hd, scale=0.5, seed=82)
    ks, vs = float(k0.float().abs().max()) * 1.5 / 448.0, float(v0.float().abs().max()) * 1.5 / 448.0
    k8, v8 = torch.zeros(b, h, ctx, hd, dtype=torch.uint8, device="cuda"), torch.zeros(b, h, ctx, hd, dtype=torch.uint8, device="cuda")
    scales = ((C.c_float * 1)(ks), (C.c_float * 1)(vs))
    kv = _lib.MdKvCache(k0.data_ptr(), v0.data_ptr(), b * h * ctx * hd, h * ctx * hd, ctx, k8.data_ptr(), v8.data_ptr(),
                        C.cast(scales[0], C.c_void_p), C.cast(scales[1], C.c_void_p))
    _lib.check(lib.md_kv_quantize_f8(C.byref(kv), 1, b, h, None, 0, ctx, stream()))
    torch.cuda.synchronize()
    assert torch.equal(k8, quant_rows(k0, ks).view(torch.uint8)) and torch.equal(v8, quant_rows(v0, vs).view(torch.uint8))
    # reference of the bf16 kernel for the new rows
    qa, ka, va = qkv.clone(), k0.clone(), v0.clone()
    _lib.check(lib.md_rope_kv_write(qa.data_ptr(), ld, freqs.data_ptr(), pos0.data_ptr(), ka.data_ptr(), va.data_ptr(),
                                    h * ctx * hd, ctx, b, 1, h, h, hd, rot, stream()))
    o = torch.zeros(b, h * hd, dtype=BF16, device="cuda")
    _lib.check(lib.md_attention_decode_rope_f8(qkv.data_ptr(), ld, o.data_ptr(), h * hd, freqs.data_ptr(), k0.data_ptr(), v0.data_ptr(),
                                               k8.data_ptr(), v8.data_ptr(), h * ctx * hd, ctx, lens.data_ptr(), b, h, rot, 0.125, ks, vs, stream()))
    torch.cuda.synchronize()
    assert torch.equal(k0, ka) and torch.equal(v0, va)  # the bf16 copy got the same new rows as the bf16 path writes
    for bi in range(b):
        p = int(pos0[bi])
        assert torch.equal(k8[bi, :, p], quant_rows(ka[bi, :, p], ks).view(torch.uint8)) and torch.equal(v8[bi, :, p], quant_rows(va[bi, :, p], vs).view(torch.uint8))
        n = p + 1
        kd = k8[bi, :, :n].view(F8).float() * ks
        vd = v8[bi, :, :n].view(F8).float() * vs
        q = qa[bi, : h * hd].view(h, hd).float()  # rotated q, as the bf16 path leaves it in the activation
        s = torch.einsum("hd,hnd->hn", q, kd) * 0.125
        pr = torch.exp(s - s.amax(dim=1, keepdim=True))
        want = (torch.einsum("hn,hnd->hd", pr.to(BF16).float(), vd) / pr.sum(dim=1, keepdim=True)).to(BF16)
        compare(f"decode attention over the e4m3 cache, len {n}", o[bi].view(h, hd), want, 6e-3, 4e-2)
