"""Shared helpers for the parity tests."""
import numpy as np
import torch


def bits_to_bf16(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a).copy()).view(torch.bfloat16)


def bf16_ulp(x: torch.Tensor) -> torch.Tensor:
    """spacing of bf16 at |x| (8 significand bits)."""
    ax = x.float().abs().clamp_min(2.0 ** -120)
    return torch.exp2(torch.floor(torch.log2(ax)) - 7)


def compare(name, got: torch.Tensor, ref: torch.Tensor, rel_rms: float, max_frac: float = None):
    """Tolerance check for bf16 activations of two implementations that share
    rounding points but not accumulation order: the RMS error relative to the
    RMS of the reference must stay below ``rel_rms``; ``max_frac`` optionally
    bounds the largest absolute error as a fraction of the largest |ref|."""
    g, r = got.detach().float().cpu(), ref.detach().float().cpu()
    assert g.shape == r.shape, f"{name}: shape {tuple(g.shape)} vs {tuple(r.shape)}"
    assert torch.isfinite(g).all(), f"{name}: non-finite values"
    err = (g - r)
    rms = float(err.pow(2).mean().sqrt() / r.pow(2).mean().sqrt().clamp_min(1e-30))
    mx = float(err.abs().max() / r.abs().max().clamp_min(1e-30))
    msg = f"{name}: rel-rms {rms:.3e} (tol {rel_rms:.1e}), max-err/max-ref {mx:.3e}"
    print(msg)
    assert rms <= rel_rms, msg
    if max_frac is not None:
        assert mx <= max_frac, msg
    return rms, mx


def margin_aware_mismatches(got, ref, margins, thr=0.5):
    """Greedy ids of two correct bf16 implementations can only part ways at a decision whose
    reference top-1/top-2 logit margin is within bf16 noise.  For every sequence: the first
    position where ``got`` and ``ref`` differ must have reference margin <= thr (margins[i][j] is the
    margin of the decision that produced token j); after that the sequences are unrelated.
    Returns (number of exactly equal sequences, list of violations)."""
    exact, bad = 0, []
    for i, (g, r) in enumerate(zip(got, ref)):
        g, r = list(g), list(r)
        n = min(len(g), len(r))
        j = next((t for t in range(n) if g[t] != r[t]), None)
        if j is None:
            exact += 1
        elif float(margins[i][j]) > thr:
            bad.append((i, j, g[j], r[j], float(margins[i][j])))
    return exact, bad


def leading_wide_objects(margins: np.ndarray, thr: float) -> int:
    """detect/point goldens: number of leading objects all of whose decisions (and every decision
    before them) have a reference margin >= thr bf16 ulps."""
    n = 0
    for row in margins:
        if float(row.min()) < thr:
            break
        n += 1
    return n


def vit_fp64(x_bchw: torch.Tensor, sd, cfg) -> torch.Tensor:
    """The ViT encoder as a function of its bf16 weights and bf16 input, evaluated in float64 with
    NO intermediate rounding (reference structure: vision.py:44-74, layers.py:118-166) -- the
    "truth" both bf16 implementations approximate."""
    v = cfg.vision
    f = lambda k: sd[k].detach().cpu().double()
    x = x_bchw.detach().cpu().double()
    B, C, H, W = x.shape
    P = v.enc_patch_size
    x = x.reshape(B, C, H // P, P, W // P, P).permute(0, 2, 4, 1, 3, 5).reshape(B, (H // P) * (W // P), C * P * P)
    x = x @ f("vision.patch_emb.weight").t() + f("vision.patch_emb.bias") + f("vision.pos_emb")
    ln = lambda t, p: torch.nn.functional.layer_norm(t, (t.shape[-1],), f(p + ".weight"), f(p + ".bias"), 1e-5)
    hd = v.enc_dim // v.enc_n_heads
    for i in range(v.enc_n_layers):
        p = f"vision.blocks.{i}"
        h = ln(x, p + ".ln1")
        qkv = h @ f(p + ".attn.qkv.weight").t() + f(p + ".attn.qkv.bias")
        q, k, vv = [t.reshape(B, -1, v.enc_n_heads, hd).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
        a = torch.softmax(q @ k.transpose(-1, -2) / hd ** 0.5, dim=-1) @ vv
        a = a.transpose(1, 2).reshape(B, -1, v.enc_dim)
        x = x + a @ f(p + ".attn.proj.weight").t() + f(p + ".attn.proj.bias")
        h = ln(x, p + ".ln2")
        h = torch.nn.functional.gelu(h @ f(p + ".mlp.fc1.weight").t() + f(p + ".mlp.fc1.bias"), approximate="tanh")
        x = x + h @ f(p + ".mlp.fc2.weight").t() + f(p + ".mlp.fc2.bias")
    return ln(x, "vision.post_ln")


def quantize_int4(w, group=128, zero_shift=0.0):
    """The checkpoint format dequantize_int4 / the int4 weight stream read (reference layers.py:38-74), from a float weight."""
    rows = w.float().cpu().reshape(-1, group)
    lo, hi = rows.min(1, keepdim=True).values, rows.max(1, keepdim=True).values
    scale = ((hi - lo) / 15).clamp_min(1e-8)
    zero = -lo / scale + zero_shift
    q = torch.clamp(torch.round(rows / scale + zero), 0, 15).to(torch.uint8)
    step = q.shape[0] // 2
    return (q[:step] << 4) | q[step:], scale, zero
