"""Host-side parity logic (moondream_amd/parity.py): the measured-noise licence for greedy ids and the margin-aware
comparison of detect objects, on constructed cases and on the committed reference fixtures themselves."""
import os
import sys

import pytest

import numpy as np

from moondream_amd import parity as P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _case(n=4, t=6, k=8, seed=0):
    rng = np.random.default_rng(seed)
    ref_ids = rng.integers(10, 1000, size=(n, t)).tolist()
    margins = np.full((n, t + 1), 2.0, dtype=np.float32)
    ref_topk = rng.normal(12.0, 2.0, size=(n, t + 1, k)).astype(np.float32)
    return ref_ids, margins, ref_topk


def test_identical_ids_and_small_logit_error_pass():
    ref_ids, margins, ref_topk = _case()
    got_topk = ref_topk + 0.0625
    rep = P.parity_report([list(r) for r in ref_ids], ref_ids, margins, got_topk, ref_topk, tokens=6, min_exact=4)
    assert rep["parity_ok"] and rep["parity_exact"] == 4 and rep["parity_max_divergence_margin"] == 0.0
    assert abs(rep["parity_max_logit_err"] - 0.0625) < 1e-6 and abs(rep["parity_threshold"] - 0.125) < 1e-6
    assert rep["parity_decisions"] == 4 * 7 and rep["parity_must_match"] == 4


def test_divergence_is_licensed_only_below_twice_the_measured_error():
    ref_ids, margins, ref_topk = _case()
    got = [list(r) for r in ref_ids]
    got[1][3] += 1                      # sequence 1 leaves the reference's stream at decision 3
    got_topk = ref_topk.copy()
    got_topk[0, 0, 0] += 0.25           # measured max |logit error| 0.25 -> licence 0.5
    margins[1, 3] = 0.375               # ... and the reference decided that token by 0.375: allowed
    rep = P.parity_report(got, ref_ids, margins, got_topk, ref_topk, tokens=6)
    assert rep["parity_ok"] and rep["parity_exact"] == 3 and abs(rep["parity_max_divergence_margin"] - 0.375) < 1e-6
    assert rep["parity_must_match"] == 3  # sequence 1 has a decision inside the licence
    margins[1, 3] = 0.75                # decided by more than the licence: a real disagreement
    rep = P.parity_report(got, ref_ids, margins, got_topk, ref_topk, tokens=6)
    assert not rep["parity_ok"] and "VIOLATIONS" in rep["parity_note"]
    # only the FIRST difference of a sequence counts: what follows is a different context
    margins[1, 3], margins[1, 4] = 0.375, 5.0
    got[1][4] += 1
    assert P.parity_report(got, ref_ids, margins, got_topk, ref_topk, tokens=6)["parity_ok"]


def test_a_broken_kernel_cannot_buy_itself_a_wide_licence():
    ref_ids, margins, ref_topk = _case()
    got = [list(r) for r in ref_ids]
    got[2][0] += 1
    margins[2, 0] = 1.5
    got_topk = ref_topk + 1.0           # error 1.0 would license margins up to 2.0 -- but it is above the cap
    rep = P.parity_report(got, ref_ids, margins, got_topk, ref_topk, tokens=6, max_err_cap=0.75)
    assert not rep["parity_ok"] and "LOGIT ERROR ABOVE ITS CAPS" in rep["parity_note"]
    # ... and ONE outlier under the cap does not widen the licence past the flat 0.5 either
    got_topk = ref_topk.copy()
    got_topk[0, 0, 0] += 0.45
    margins[2, 0] = 0.75                # 2 x 0.45 = 0.9 would have covered it
    rep = P.parity_report(got, ref_ids, margins, got_topk, ref_topk, tokens=6)
    assert not rep["parity_ok"] and rep["parity_threshold"] == 0.5
    # many small errors that are large in ULPS (tiny logits) trip the p99 gate
    small = np.full_like(ref_topk, 0.01)
    rep = P.parity_report([list(r) for r in ref_ids], ref_ids, margins, small + 0.002, small, tokens=6)
    assert not rep["parity_ok"] and rep["parity_p99_logit_err_ulps"] > 12


def test_a_divergence_must_be_covered_by_the_errors_on_its_own_logits():
    ref_ids, margins, ref_topk = _case()
    n, t1, k = ref_topk.shape
    ref_topk = -np.sort(-ref_topk, axis=2)                      # candidates in descending order, as the fixtures store them
    idx = np.tile(np.arange(100, 100 + k), (n, t1, 1))
    for i in range(n):
        for j in range(t1 - 1):
            idx[i, j, 0] = ref_ids[i][j]                        # the reference's winner is candidate 0
    got = [list(r) for r in ref_ids]
    got[1][3] = int(idx[1, 3, 1])                               # this run picked the reference's runner-up at decision 3
    ref_topk[1, 3, 1] = ref_topk[1, 3, 0] - 0.25                # the reference preferred its winner by 0.25
    margins[1, 3] = 0.25
    got_topk = ref_topk.copy()
    got_topk[0, 0, 0] += 0.25                                   # a global max error of 0.25 (licence 0.5) somewhere else ...
    rep = P.parity_report(got, ref_ids, margins, got_topk, ref_topk, tokens=6, ref_topk_idx=idx)
    assert not rep["parity_ok"] and "UNCOVERED" in rep["parity_note"]   # ... does not explain THIS decision: its own logits are exact
    got_topk[1, 3, 1] += 0.1875
    got_topk[1, 3, 0] -= 0.0625                                 # errors on exactly these two logits: 0.25 in total -> covered
    assert P.parity_report(got, ref_ids, margins, got_topk, ref_topk, tokens=6, ref_topk_idx=idx)["parity_ok"]
    got[1][3] = 7                                               # a token the reference did not even rank
    rep = P.parity_report(got, ref_ids, margins, got_topk, ref_topk, tokens=6, ref_topk_idx=idx)
    assert not rep["parity_ok"] and "outside the reference's recorded candidates" in rep["parity_note"]


def test_exact_count_floor_and_flat_licence_without_logits():
    ref_ids, margins, ref_topk = _case()
    got = [list(r) for r in ref_ids]
    for i in (0, 1, 2):
        got[i][1] += 1
        margins[i, 1] = 0.0             # ties: always licensed
    rep = P.parity_report(got, ref_ids, margins, ref_topk, ref_topk + 0.01, tokens=6, min_exact=2)
    assert not rep["parity_ok"] and rep["parity_exact"] == 1
    assert P.parity_report(got, ref_ids, margins, ref_topk, ref_topk + 0.01, tokens=6, min_exact=1)["parity_ok"]
    rep = P.parity_report(got, ref_ids, margins, None, None, tokens=6)  # no logits: the flat round-2 licence of 0.5
    assert rep["parity_ok"] and rep["parity_threshold"] == 0.5


def test_logit_error_in_bf16_ulps():
    ref = np.array([[[16.0, 8.0, 1.0]]], dtype=np.float32)   # bf16 spacing 0.125, 0.0625, 0.0078125
    got = ref + np.array([[[0.125, 0.125, 0.0078125]]], dtype=np.float32)
    st = P.logit_error_stats(got, ref)
    assert st["max"] == 0.125 and st["max_ulps"] == 2.0 and st["decisions"] == 1


def test_reference_fixture_against_itself():
    """tests/golden/md2b_bench64.npz / md2b_vqa64.npz: the reference's own ids are 64/64 and every decision of both bench
    fixtures is wide-margin."""
    g = np.load(os.path.join(GOLD, "md2b_bench64.npz"))
    ids = g["tokens"].tolist()
    rep = P.parity_report(ids, ids, g["margins"], g["top8_val"], g["top8_val"], tokens=32, min_exact=64)
    assert rep["parity_ok"] and rep["parity_exact"] == 64 and rep["parity_threshold"] == 0.0
    # round 6: the fixture is WELL-CONDITIONED -- every one of the 64 x 33 decisions of the reference has a top-1 / top-2 margin
    # far above the measured logit error (0.31), so "ids equal the reference's" is a requirement on every sequence, not a
    # licensed statistic (rounds 1-5: 9 sequences held an exact tie, 12 had every margin above 0.25)
    m = g["margins"]
    assert m.shape == (64, 33) and float(m.min()) >= 2.0, float(m.min())
    assert rep["parity_must_match"] == 64 and rep["parity_tf_decisions_must_match"] == 64 * 33
    assert len({tuple(t) for t in ids}) >= 48              # and the streams still differ from image to image
    v = np.load(os.path.join(GOLD, "md2b_vqa64.npz"))
    assert v["margins"].shape == (64, 33) and float(v["margins"].min()) >= 2.0, float(v["margins"].min())


def test_detect_parity_compares_leading_wide_objects_exactly():
    g = np.load(os.path.join(GOLD, "md2b_detect13.npz"))
    n = int(g["n_images"])
    objs = []
    for i in range(n):
        ref = np.asarray(g[f"img{i}.objects"]).reshape(-1, 4)
        objs.append([dict(zip(("x_min", "y_min", "x_max", "y_max"), r.tolist())) for r in ref])
    rep = P.detect_parity(objs, g)
    assert rep["ok"] and rep["objects_compared"] > 0 and rep["objects_mismatched"] == 0
    # leading_wide_objects stops at the first object with a narrow decision
    assert P.leading_wide_objects(np.array([[9, 9, 9, 9, 9], [9, 3.9, 9, 9, 9], [9, 9, 9, 9, 9]], dtype=np.float32), 4.0) == 1
    # a changed coordinate of a compared object is a mismatch; a missing object too
    i0 = next(i for i in range(n) if P.leading_wide_objects(g[f"img{i}.margins"], 4.0) > 0)
    bad = [list(o) for o in objs]
    bad[i0] = [dict(bad[i0][0], y_min=bad[i0][0]["y_min"] + 1e-3)] + bad[i0][1:]
    assert P.detect_parity(bad, g)["objects_mismatched"] == 1
    bad[i0] = []
    assert not P.detect_parity(bad, g)["ok"]


def test_teacher_forced_decisions_with_a_wide_margin_must_all_match():
    """Round 4: per-DECISION check on the teacher-forced logits (no cascade): every decision whose reference margin exceeds the
    licence must pick the reference's token, even in sequences whose free-running ids left the stream earlier."""
    ref_ids, margins, ref_topk = _case()
    ref_topk = -np.sort(-ref_topk, axis=-1)          # winner first, as the fixtures store them
    ref_topk[..., 0] = ref_topk[..., 1] + 2.0        # every decision decided by 2.0
    ids = [list(r) for r in ref_ids]
    got_topk = ref_topk + 0.0625
    rep = P.parity_report(ids, ref_ids, margins, got_topk, ref_topk, tokens=6)
    assert rep["parity_ok"] and rep["parity_tf_decisions_must_match"] == 4 * 7 and rep["parity_tf_decisions_violations"] == 0
    assert rep["parity_tf_decisions_agree"] == 4 * 7
    # one decision where this run's logits prefer the runner-up although the reference decided it by 2.0: caught, although the
    # free-running ids are all identical and the measured error stays under its caps
    bad = got_topk.copy()
    bad[3, 5, 1] = bad[3, 5, 0] + 0.01
    bad[3, 5, 0] -= 0.3
    rep = P.parity_report(ids, ref_ids, margins, bad, ref_topk, tokens=6, max_err_cap=5.0, p99_ulps_cap=1e9, flat_cap=0.5)
    assert not rep["parity_ok"] and rep["parity_tf_decisions_violations"] == 1 and "TEACHER-FORCED VIOLATIONS" in rep["parity_note"]
    # a narrow decision (inside the licence) may tip either way
    ref2 = ref_topk.copy()
    ref2[3, 5, 0] = ref2[3, 5, 1] + 0.0625
    got2 = ref2.copy()
    got2[3, 5, 1] += 0.125
    rep = P.parity_report(ids, ref_ids, margins, got2, ref2, tokens=6)
    assert rep["parity_ok"] and rep["parity_tf_decisions_must_match"] == 4 * 7 - 1 and rep["parity_tf_decisions_agree"] == 4 * 7 - 1


# ------------------------------------------------------------------ the seam's mask argument (moondream_amd/integration.py)
def _reference_masks(ctx=2048, prefix=730):
    """The two mask buffers the reference builds: moondream.py:138-146 (prefix-LM) and 571-575 (text-only tril)."""
    import torch

    tril = torch.tril(torch.ones(1, 1, ctx, ctx, dtype=torch.bool))
    prefix_lm = tril.clone()
    prefix_lm[..., :prefix, :prefix] = 1
    return tril, prefix_lm


def test_seam_mask_classifier_on_the_masks_the_reference_passes():
    import torch
    from moondream_amd.integration import MASK_CAUSAL, MASK_EITHER, MASK_PREFIX_LM, classify_attn_mask, consecutive_positions

    ctx, P = 2048, 730
    tril, prefix_lm = _reference_masks(ctx, P)
    ar = lambda a, b: torch.arange(a, b, dtype=torch.long)
    # encode_image (moondream.py:254-257): rows 0..729 of the prefix-LM buffer
    assert classify_attn_mask(prefix_lm[:, :, 0:730, :], ar(0, 730), P, ctx) == MASK_PREFIX_LM
    # prompt prefill after an image (moondream.py:307-309): rows 730.. -- both rules select keys [0, p]
    assert classify_attn_mask(prefix_lm[:, :, 730:735, :], ar(730, 735), P, ctx) == MASK_EITHER
    assert classify_attn_mask(tril[:, :, 730:762, :], ar(730, 762), P, ctx) == MASK_EITHER
    # text-only query (moondream.py:571-575): rows 0..n-1 of a plain tril
    assert classify_attn_mask(tril[:, :, 0:9, :], ar(0, 9), P, ctx) == MASK_CAUSAL
    # rows 0..729 of the tril: causal over the image positions (not what encode_image passes, but one of the two rules)
    assert classify_attn_mask(tril[:, :, 0:730, :], ar(0, 730), P, ctx) == MASK_CAUSAL
    # a pass that straddles the prefix boundary
    assert classify_attn_mask(prefix_lm[:, :, 0:740, :], ar(0, 740), P, ctx) == MASK_PREFIX_LM
    # decode rows (moondream.py:472-474,515): ones on [0, pos]
    for pos, want in ((735, MASK_EITHER), (12, MASK_CAUSAL), (728, MASK_CAUSAL), (729, MASK_EITHER), (730, MASK_EITHER), (ctx - 1, MASK_EITHER)):
        row = torch.zeros(1, 1, ctx, dtype=torch.bool)
        row[:, :, : pos + 1] = 1
        assert classify_attn_mask(row, torch.tensor([pos]), P, ctx) == want, pos
    assert classify_attn_mask(None, ar(3, 8), P, ctx) == MASK_EITHER
    # anything else is refused, loudly
    hole = tril[:, :, 0:9, :].clone()
    hole[0, 0, 5, 2] = False
    stale = torch.zeros(1, 1, ctx, dtype=torch.bool)
    stale[:, :, :800] = 1                      # decode row that exposes slots beyond pos
    for mask, pos in ((hole, ar(0, 9)), (stale, torch.tensor([735])), (tril[:, :, 0:9, :], ar(1, 10)),
                      (tril[:, :, 0:9, :].to(torch.uint8), ar(0, 9)), (tril[:, :, 0:9, :100], ar(0, 9)),
                      (tril[:, :, 0:8, :], ar(0, 9)), (tril[0, 0, 0:9, :], ar(0, 9))):
        with pytest.raises(ValueError):
            classify_attn_mask(mask, pos, P, ctx)
    # positions: consecutive ascending only
    assert consecutive_positions(ar(730, 735)) == 730
    for bad in (torch.tensor([0, 1, 3]), torch.tensor([5, 4, 3]), torch.tensor([2, 2]), torch.tensor([], dtype=torch.long), torch.tensor([-1, 0])):
        with pytest.raises(ValueError):
            consecutive_positions(bad)
    with pytest.raises(ValueError):
        classify_attn_mask(None, ar(2040, 2050), P, ctx)


def test_seam_mask_classifier_on_the_reference_models_own_buffers():
    """Same check on the tensors the unmodified reference really builds (build container only)."""
    import torch
    ref_root = os.environ.get("MOONDREAM_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref_root, "moondream", "torch")):
        pytest.skip("needs the reference checkout (build container)")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import make_golden as mg
    from moondream_amd import synth
    from moondream_amd.config import get_config
    from moondream_amd.integration import MASK_CAUSAL, MASK_EITHER, MASK_PREFIX_LM, classify_attn_mask

    cfg = get_config("tiny")
    model, _ = mg.load_reference(cfg, synth.synthetic_state_dict(cfg, seed=1))
    seen = []
    orig_prefill, orig_decode = model._prefill, model._decode_one_tok

    def prefill_tap(x, mask, pos_ids, lora):
        seen.append(("prefill", classify_attn_mask(mask, pos_ids, cfg.text.prefix_attn, cfg.text.max_context), int(pos_ids[0])))
        return orig_prefill(x, mask, pos_ids, lora)

    def decode_tap(x, mask, pos_ids, lora):
        seen.append(("decode", classify_attn_mask(mask, pos_ids, cfg.text.prefix_attn, cfg.text.max_context), int(pos_ids[0])))
        return orig_decode(x, mask, pos_ids, lora)

    model._prefill, model._decode_one_tok = prefill_tap, decode_tap
    from PIL import Image
    img = Image.fromarray(synth.synthetic_image_array(0, 1, (378, 378)), "RGB")
    model.caption(img, settings={"temperature": 0, "max_tokens": 3, "variant": None})
    model.query(None, "11 12 13", settings={"temperature": 0, "max_tokens": 3})
    kinds = [(k, r) for k, r, _ in seen]
    assert ("prefill", MASK_PREFIX_LM) == kinds[0]          # encode_image
    assert ("prefill", MASK_EITHER) == kinds[1]             # caption prompt at pos 730
    assert ("prefill", MASK_CAUSAL) in kinds                # the text-only query's prompt at pos 0
    assert ("decode", MASK_CAUSAL) in kinds and ("decode", MASK_EITHER) in kinds


def test_vqa64_fixture_is_the_bench_legs_configuration():
    """tests/golden/md2b_vqa64.npz (oracle/make_golden.py vqa64, from the unmodified reference): the 64 seed-1 images x the 32-id
    question prompts bench.py's vqa32 leg runs; internally consistent (ids = argmax of the recorded candidates wherever the
    margin is positive) and statistically like the caption fixture (so the same gates apply)."""
    from moondream_amd import synth
    from moondream_amd.config import get_config

    g = np.load(os.path.join(GOLD, "md2b_vqa64.npz"))
    cfg = get_config("2b")
    assert int(g["seed"]) == 1 and str(g["cfg"]) == "2b"
    assert g["prompt"].shape == (64, 32) and g["tokens"].shape == (64, 32) and g["margins"].shape == (64, 33)
    assert g["top8_idx"].shape == (64, 33, 8) and g["top8_val"].shape == (64, 33, 8)
    assert g["prompt"].tolist() == [synth.synthetic_vqa_prompt(cfg, i, 1) for i in range(64)]
    tpl = cfg.tokenizer.templates["query"]
    assert g["prompt"][0, : len(tpl["prefix"])].tolist() == list(tpl["prefix"])
    top = g["top8_val"]
    assert np.allclose(top[..., 0] - top[..., 1], g["margins"], atol=1e-6)
    pos = g["margins"][:, :32] > 0
    assert (g["top8_idx"][:, :32, 0][pos] == g["tokens"][pos]).all()
    assert (g["margins"] > 0.5).sum() >= 1800   # the must-match decisions the per-decision gate rests on
    # the per-decision report accepts the reference against itself and rejects a swapped pair of candidates
    rep = P.parity_report(g["tokens"].tolist(), g["tokens"].tolist(), g["margins"], g["top8_val"], g["top8_val"], tokens=32,
                          min_exact=64, ref_topk_idx=g["top8_idx"])
    assert rep["parity_ok"] and rep["parity_exact"] == 64
    bad = g["top8_val"].copy()
    i, j = np.argwhere(g["margins"][:, :32] > 2.0)[0]
    bad[i, j, [0, 1]] = bad[i, j, [1, 0]]
    rep = P.parity_report(g["tokens"].tolist(), g["tokens"].tolist(), g["margins"], bad, g["top8_val"], tokens=32, ref_topk_idx=g["top8_idx"])
    assert not rep["parity_ok"]


def test_fp8_contract_and_fp8_detect_parity_helpers():
    g = np.load(os.path.join(GOLD, "md2b_bench64.npz"))
    ref = g["top8_val"]
    ok = P.fp8_contract_report(ref + np.random.default_rng(0).normal(0, 0.8, ref.shape), ref)
    assert ok["tolerance_ok"] and ok["ok"] and ok["must_match"] > 100 and ok["must_match_violations"] == 0
    loud = P.fp8_contract_report(ref + np.random.default_rng(0).normal(0, 3.0, ref.shape), ref)
    assert not loud["tolerance_ok"] and not loud["ok"]
    swapped = ref.copy()
    i, j = np.argwhere((ref[..., 0] - ref[..., 1]) > P.FP8_LICENCE)[0]
    swapped[i, j, [0, 1]] = swapped[i, j, [1, 0]]
    assert P.fp8_contract_report(swapped, ref)["must_match_violations"] == 1
    gd = np.load(os.path.join(GOLD, "md2b_detect13.npz"))
    objs = [[dict(zip(("x_min", "y_min", "x_max", "y_max"), o)) for o in np.asarray(gd[f"img{i}.objects"]).reshape(-1, 4).tolist()]
            for i in range(int(gd["n_images"]))]
    # round 6: every object decision of the fixture clears the fp8 licence (>= 120 of 96 bf16 ulps): the fp8 mode's objects ARE
    # validated at the object level -- ok is True / False, never None
    r = P.detect_parity_fp8(objs, gd)
    assert r["ok"] is True and r["objects_compared"] == r["objects_equal"] == r["objects_paired"] == 32 and "equal the reference" in r["verdict"]
    assert r["centre_error_bins_median_p90_max"] == [0.0, 0.0, 0.0]
    objs[0][0]["y_min"] += 0.01   # 0.01 of the image height: the centre moves by 0.005 x 1024 bins
    r = P.detect_parity_fp8(objs, gd)
    assert r["ok"] is False and r["objects_equal"] == 31 and abs(r["centre_error_bins_median_p90_max"][2] - 5.12) < 1e-6
    # a licence no object of the fixture clears: stated as what it is (throughput only), not as a pass
    r4 = P.detect_parity_fp8(objs, gd, licence_ulps=1e6)
    assert r4["objects_compared"] == 0 and r4["ok"] is None and "THROUGHPUT ONLY" in r4["verdict"]
