"""Margin-aware comparison of greedy ids against the reference's, with a MEASURED noise threshold.

Two correct bf16 implementations share rounding points but not GEMM accumulation order, so their
logits differ by a few bf16 ulps.  Greedy ids are an integer output only where the reference's
top-1/top-2 logit margin exceeds that noise: the implementation under test picks another token at
a decision exactly when  (its error on the runner-up) - (its error on the winner) >= margin,
i.e. only where  margin <= 2 * max|logit error|.  So the licence for a divergence is not a flat
constant: it is TWICE THE LARGEST LOGIT ERROR MEASURED ON THIS VERY RUN, teacher-forced on the
reference's ids over every decision of every sequence (``MoondreamModel.teacher_forced_logits``
against the reference's recorded top-k logits, tests/golden/md2b_bench64.npz).  A broken kernel
must not be able to buy itself a wide licence with one outlier, so the licence is BOUNDED three
ways (round 4): it never exceeds ``flat_cap`` (0.5, the flat round-2 licence), the measured errors
are themselves gated (max <= ``max_err_cap``, p99 <= ``p99_ulps_cap`` bf16 ulps), and every first
divergence is checked ON ITS OWN DECISION: the token this run picked must be one of the reference's
recorded top-k candidates there, and the reference's margin between its winner and that token must
be covered by the two errors measured on exactly those two logits.

Used by bench.py (after the timed region) and tests/test_model_gpu.py.  Host-side numpy only.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np

NOISE_FACTOR = 2.0  # a flip needs e(runner-up) - e(winner) >= margin, and each |e| <= max_err


def logit_error_stats(got_topk: np.ndarray, ref_topk: np.ndarray) -> Dict[str, float]:
    """got/ref: float [B, T+1, k] logits at the reference's top-k ids of every decision."""
    err = np.abs(np.asarray(got_topk, dtype=np.float64) - np.asarray(ref_topk, dtype=np.float64))
    err = err[np.isfinite(np.asarray(ref_topk))]
    # bf16 spacing at the reference logit's magnitude (8 significand bits)
    mag = np.abs(np.asarray(ref_topk, dtype=np.float64))[np.isfinite(np.asarray(ref_topk))]
    ulp = np.exp2(np.floor(np.log2(np.maximum(mag, 2.0 ** -120))) - 7)
    return {
        "max": float(err.max()), "p99": float(np.quantile(err, 0.99)), "mean": float(err.mean()),
        "max_ulps": float((err / ulp).max()), "p99_ulps": float(np.quantile(err / ulp, 0.99)),
        "decisions": int(np.asarray(ref_topk).shape[0] * np.asarray(ref_topk).shape[1]),
    }


def first_divergences(got_ids: Sequence[Sequence[int]], ref_ids, margins, tokens: Optional[int] = None):
    """[(sequence, position, got, want, reference margin of that decision)] for every sequence that leaves the
    reference's stream, and the number of sequences that never do."""
    out, exact = [], 0
    for i, (g, r) in enumerate(zip(got_ids, ref_ids)):
        g, r = list(g), list(r)
        n = min(len(g), len(r)) if tokens is None else min(len(g), len(r), tokens)
        j = next((t for t in range(n) if g[t] != r[t]), None)
        if j is None:
            exact += 1
        else:
            out.append((i, j, int(g[j]), int(r[j]), float(margins[i][j])))
    return exact, out


def parity_report(got_ids, ref_ids, margins, got_topk: Optional[np.ndarray], ref_topk: Optional[np.ndarray],
                  tokens: Optional[int] = None, max_err_cap: float = 0.5, min_exact: Optional[int] = None,
                  ref_topk_idx: Optional[np.ndarray] = None, flat_cap: float = 0.5, p99_ulps_cap: float = 10.0) -> Dict[str, object]:
    """The JSON-able verdict.  ``parity_ok`` iff (a) the measured logit error is under its caps (max, p99 in ulps), (b) every
    first divergence sits at a reference margin <= min(NOISE_FACTOR x the measured max error, flat_cap) AND is covered by
    the errors measured on its own two logits (``ref_topk_idx`` given), (c) at least ``min_exact`` sequences are identical
    (when given)."""
    exact, div = first_divergences(got_ids, ref_ids, margins, tokens)
    n = min(len(got_ids), len(ref_ids))
    rep: Dict[str, object] = {"parity_checked": n, "parity_exact": exact}
    worst = max((d[4] for d in div), default=0.0)
    rep["parity_max_divergence_margin"] = worst
    uncovered = []
    if got_topk is not None and ref_topk is not None:
        st = logit_error_stats(got_topk, ref_topk)
        thr = min(NOISE_FACTOR * st["max"], flat_cap)
        rep.update({
            "parity_max_logit_err": st["max"], "parity_p99_logit_err": st["p99"], "parity_max_logit_err_ulps": st["max_ulps"],
            "parity_p99_logit_err_ulps": st["p99_ulps"], "parity_decisions": st["decisions"], "parity_threshold": thr,
        })
        err_ok = st["max"] <= max_err_cap and st["p99_ulps"] <= p99_ulps_cap
        if ref_topk_idx is not None:
            got_a, ref_a, idx_a = np.asarray(got_topk, dtype=np.float64), np.asarray(ref_topk, dtype=np.float64), np.asarray(ref_topk_idx)
            for (i, j, g, r, _m) in div:
                ks = np.nonzero(idx_a[i, j] == g)[0]
                kr = np.nonzero(idx_a[i, j] == r)[0]
                if len(ks) == 0 or len(kr) == 0:
                    uncovered.append((i, j, g, r, "picked a token outside the reference's recorded candidates"))
                    continue
                kg, kr = int(ks[0]), int(kr[0])
                need = ref_a[i, j, kr] - ref_a[i, j, kg]  # the reference's margin between its winner and the token picked here
                have = abs(got_a[i, j, kg] - ref_a[i, j, kg]) + abs(got_a[i, j, kr] - ref_a[i, j, kr])
                if need > have + 1e-6:
                    uncovered.append((i, j, g, r, f"margin {need:.4f} > measured errors {have:.4f}"))
            rep["parity_divergences_checked_on_their_own_logits"] = len(div)
        # TEACHER-FORCED DECISIONS (round 4).  The per-sequence count dies at a sequence's first narrow decision, and the bench
        # fixture has one in most sequences (only 2 of 64 have every margin above the licence).  Teacher-forced on the reference's
        # ids there is no cascade: EVERY decision whose reference margin exceeds the licence must come out as the reference's
        # token, whatever happened earlier in its sequence (the argmax is taken over the reference's recorded top-k candidates: a
        # token outside them would need an error of several units).  On the bench fixture that is ~1940 of 2112 decisions that
        # MUST match instead of 2 sequences.
        got_a, ref_a = np.asarray(got_topk, dtype=np.float64), np.asarray(ref_topk, dtype=np.float64)
        ref_f = np.where(np.isfinite(ref_a), ref_a, -np.inf)
        ref_sorted = np.sort(ref_f, axis=-1)
        ref_margin = ref_sorted[..., -1] - ref_sorted[..., -2]  # winner minus runner-up among the recorded candidates
        must = ref_margin > thr
        agree = np.argmax(np.where(np.isfinite(ref_a), got_a, -np.inf), axis=-1) == np.argmax(ref_f, axis=-1)
        tf_viol = np.argwhere(must & ~agree)
        rep.update({"parity_tf_decisions_must_match": int(must.sum()), "parity_tf_decisions_violations": int(len(tf_viol)),
                    "parity_tf_decisions_agree": int(agree.sum())})
    else:
        thr, err_ok = flat_cap, True  # no logits available: the flat round-2 licence
        rep["parity_threshold"] = thr
        tf_viol = np.zeros((0, 2), dtype=np.int64)
    bad = [d for d in div if d[4] > thr]
    # sequences that MUST be identical: every decision's reference margin above the licence (enforced by `bad` above); the
    # others may tip either way -- the bench fixture has 9 sequences with an exact tie (margin 0) and only 12 whose smallest
    # margin exceeds 0.25, so the COUNT of identical sequences is a noisy statistic (46..52 of 64 across builds that differ in
    # a handful of last-bit roundings).  `min_exact` is a floor calibrated against the second oracle (bench.py).
    mm = np.asarray(margins)[:n, : (tokens if tokens is not None else np.asarray(margins).shape[1])]
    rep["parity_must_match"] = int((mm.min(axis=1) > thr).sum())
    ok = err_ok and not bad and not uncovered and len(tf_viol) == 0 and (min_exact is None or exact >= min_exact)
    rep["parity_ok"] = bool(ok)
    rep["parity_note"] = (
        f"ids vs the reference's (tests/golden/md2b_bench64.npz): {exact}/{n} sequences identical; every first difference must sit at "
        f"a reference top-1/top-2 margin <= min({NOISE_FACTOR:g} x the max |logit error| measured teacher-forced on this run, {flat_cap}) "
        f"= {thr:.4f} and be covered by the errors measured on its own two logits (caps: max error {max_err_cap}, p99 {p99_ulps_cap} ulps); "
        f"largest margin at a first difference {worst:.4f}"
        + (f"; VIOLATIONS {bad[:6]}" if bad else "") + (f"; UNCOVERED {uncovered[:4]}" if uncovered else "")
        + (f"; teacher-forced: {rep['parity_tf_decisions_must_match']} of {rep['parity_decisions']} decisions have a reference margin above the "
           f"licence and must equal the reference's token: {rep['parity_tf_decisions_violations']} violations" if "parity_tf_decisions_must_match" in rep else "")
        + (f"; TEACHER-FORCED VIOLATIONS at (sequence, decision) {tf_viol[:6].tolist()}" if len(tf_viol) else "")
        + ("" if err_ok else "; LOGIT ERROR ABOVE ITS CAPS"))
    return rep


def leading_wide_objects(margins: np.ndarray, thr_ulps: float) -> int:
    """detect / point goldens: number of leading objects all of whose decisions (and every decision before them)
    have a reference top-1/top-2 margin >= thr_ulps bf16 ulps of the top logit."""
    n = 0
    for row in np.asarray(margins).reshape(len(margins), -1):
        if float(row.min()) < thr_ulps:
            break
        n += 1
    return n


def detect_parity(objects_per_image, golden, thr_ulps: float = 4.0) -> Dict[str, object]:
    """``detect`` objects against the reference's (tests/golden/md2b_detect13.npz: unfiltered images, every decision's
    margin recorded).  The region heads' 1024-bin argmaxes are integer decisions like token ids: compared EXACTLY (the
    floats are functions of the bins) for the leading objects whose every decision -- x, y, w, h bins and the next
    token -- has a reference margin >= ``thr_ulps`` bf16 ulps; after the first narrower decision the two streams may
    legitimately part."""
    n_img = min(len(objects_per_image), int(golden["n_images"]))
    compared = mismatched = 0
    detail = []
    for i in range(n_img):
        ref = np.asarray(golden[f"img{i}.objects"]).reshape(-1, 4)
        n_ok = leading_wide_objects(golden[f"img{i}.margins"], thr_ulps) if len(ref) else 0
        got = objects_per_image[i]
        for k in range(n_ok):
            compared += 1
            g = [got[k][f] for f in ("x_min", "y_min", "x_max", "y_max")] if k < len(got) else None
            if g != ref[k].tolist():
                mismatched += 1
                detail.append((i, k, g, ref[k].tolist()))
    return {"images": n_img, "objects_compared": compared, "objects_mismatched": mismatched, "margin_floor_ulps": thr_ulps,
            "ok": mismatched == 0 and compared > 0, "mismatches": detail[:4]}


# ------------------------------------------------------------------ the fp8 mode's accuracy contract (round 5)
#
# e4m3 keeps 3 mantissa bits per operand: ~4 % relative noise per GEMM output whatever the scale, ~10 % after 27 + 24
# residual blocks -- logits of 10..20 move by ~1 on average (profiles/r02_fp8_operand_numerics_study.txt).  That is 10x the
# bf16 path's error and more than most top-1/top-2 margins of the synthetic checkpoint, so the fp8 mode is NOT held to
# "ids equal the reference's".  It is held to a STATED tolerance on its teacher-forced logits against the reference's
# recorded top-8 logits, and to agreement on the decisions whose reference margin exceeds a STATED licence:
FP8_P99_LOGIT_ERR = 3.0      # |logit error| p99 over all recorded candidates (bf16 mode: 0.19)
FP8_MEAN_LOGIT_ERR = 1.5     # mean (bf16 mode: ~0.05)
FP8_LICENCE = 6.0            # = 2 x the p99 cap: a decision whose reference margin exceeds it must come out as the reference's
FP8_REGION_LICENCE_ULPS = 96.0  # the same licence for the region heads' 1024-bin decisions, in bf16 ulps of the top logit
                                # (6.0 at logits of 8..16, where one ulp is 0.0625)


def fp8_contract_report(got_topk: np.ndarray, ref_topk: np.ndarray) -> Dict[str, object]:
    """The fp8 mode teacher-forced on the reference's ids: logits at the reference's top-8 ids of every decision against
    the reference's.  ``ok`` iff the error is inside the stated tolerance AND every decision whose reference margin exceeds
    ``FP8_LICENCE`` picks the reference's token."""
    st = logit_error_stats(got_topk, ref_topk)
    got_a, ref_a = np.asarray(got_topk, dtype=np.float64), np.asarray(ref_topk, dtype=np.float64)
    ref_f = np.where(np.isfinite(ref_a), ref_a, -np.inf)
    srt = np.sort(ref_f, axis=-1)
    margin = srt[..., -1] - srt[..., -2]
    agree = np.argmax(np.where(np.isfinite(ref_a), got_a, -np.inf), axis=-1) == np.argmax(ref_f, axis=-1)
    must = margin > FP8_LICENCE
    viol = int((must & ~agree).sum())
    tol_ok = st["p99"] <= FP8_P99_LOGIT_ERR and st["mean"] <= FP8_MEAN_LOGIT_ERR
    bands = {}
    for lo, hi in ((0.0, 0.5), (0.5, 1.0), (1.0, 2.0), (2.0, 4.0), (4.0, FP8_LICENCE), (FP8_LICENCE, np.inf)):
        sel = (margin > lo) & (margin <= hi)
        bands[f"margin ({lo:g}, {hi:g}]"] = {"decisions": int(sel.sum()), "agree": int((sel & agree).sum())}
    return {
        "contract": f"teacher-forced on the reference's ids, logits at the reference's top-8 ids: p99 |error| <= {FP8_P99_LOGIT_ERR}, mean <= "
                    f"{FP8_MEAN_LOGIT_ERR}; every decision with a reference margin > {FP8_LICENCE} equals the reference's token",
        "max_logit_err": st["max"], "p99_logit_err": st["p99"], "mean_logit_err": st["mean"], "max_logit_err_ulps": st["max_ulps"],
        "p99_logit_err_ulps": st["p99_ulps"], "decisions": st["decisions"], "decisions_agree": int(agree.sum()),
        "must_match": int(must.sum()), "must_match_violations": viol, "agreement_by_reference_margin": bands,
        "tolerance_ok": bool(tol_ok), "ok": bool(tol_ok and viol == 0),
    }


def _object_bins(o) -> Tuple[float, float, float, float]:
    """(x_min, y_min, x_max, y_max) -> the region heads' own units: centre bins (decode_coordinate: bin / 1024, region.py:47-62)
    and size bins (decode_size: 2 ** (bin / 1023 * 10 - 10), region.py:79-93)."""
    x0, y0, x1, y1 = (float(v) for v in o)
    size_bin = lambda s: (np.log2(max(s, 2.0 ** -10)) + 10.0) * 102.3
    return (x0 + x1) * 512.0, (y0 + y1) * 512.0, size_bin(x1 - x0), size_bin(y1 - y0)


def detect_parity_fp8(objects_per_image, golden, licence_ulps: float = FP8_REGION_LICENCE_ULPS) -> Dict[str, object]:
    """The fp8 mode's ``detect`` objects against the reference's (bf16) goldens, under the fp8 licence: objects whose every
    decision has a reference margin >= ``licence_ulps`` must be equal; for ALL objects the centre / size differences are
    reported in the region heads' own bins.  When no object of the fixture clears the licence (the region heads of the
    synthetic checkpoint decide with margins of 0..85 ulps) the verdict is stated as what it is: throughput only."""
    base = detect_parity(objects_per_image, golden, thr_ulps=licence_ulps)
    d_centre, d_size, n_pairs, n_equal = [], [], 0, 0
    for i in range(base["images"]):
        ref = np.asarray(golden[f"img{i}.objects"]).reshape(-1, 4)
        got = objects_per_image[i]
        for k in range(min(len(ref), len(got))):
            g = [got[k][f] for f in ("x_min", "y_min", "x_max", "y_max")]
            gb, rb = _object_bins(g), _object_bins(ref[k].tolist())
            d_centre += [abs(gb[0] - rb[0]), abs(gb[1] - rb[1])]
            d_size += [abs(gb[2] - rb[2]), abs(gb[3] - rb[3])]
            n_pairs += 1
            n_equal += int(g == ref[k].tolist())
    q = lambda v, p: (float(np.quantile(v, p)) if len(v) else None)
    out = dict(base)
    out.update({
        "licence_ulps": licence_ulps, "objects_paired": n_pairs, "objects_equal": n_equal,
        "centre_error_bins_median_p90_max": [q(d_centre, 0.5), q(d_centre, 0.9), q(d_centre, 1.0)],
        "size_error_bins_median_p90_max": [q(d_size, 0.5), q(d_size, 0.9), q(d_size, 1.0)],
    })
    if base["objects_compared"] == 0:
        out["ok"] = None
        out["verdict"] = (f"THROUGHPUT ONLY, outputs unvalidated at the object level: no object of the fixture has every decision above the "
                          f"fp8 licence ({licence_ulps:g} bf16 ulps; the reference's region-head margins are below it), so equality "
                          "with the bf16 reference is not a well-posed requirement for an e4m3 mode; the mode's accuracy contract is the "
                          "teacher-forced logit tolerance of the fp8_full leg / tests/test_model_gpu.py")
    else:
        out["verdict"] = "objects above the fp8 licence equal the reference's" if base["ok"] else "MISMATCH above the fp8 licence"
    return out
