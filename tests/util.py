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
