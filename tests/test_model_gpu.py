"""End-to-end parity of the HIP path on a real MI355X, through the public API
and the four seam methods, against (a) goldens recorded from the reference
implementation and (b) the CPU oracle on the same seeded inputs."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from moondream_amd import synth
from moondream_amd.config import get_config
from util import bits_to_bf16, compare, leading_wide_objects, margin_aware_mismatches, vit_fp64

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def load_golden(golden_dir, name):
    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated")
    return np.load(path)


def build(cfg_name, seed, max_batch=4):
    max_batch = 128 if cfg_name == "2b" else max_batch  # the 2B test also runs the timed B=64 (x2 slot groups) config
    from moondream_amd.moondream import MoondreamModel, IdTokenizer

    cfg = get_config(cfg_name)
    sd = synth.synthetic_state_dict(cfg, seed=seed, device="cuda")
    return cfg, sd, MoondreamModel(cfg, sd, device="cuda", tokenizer=IdTokenizer(), max_batch=max_batch)


def golden_image(g, idx, kind="cap"):
    src = int(g["image_index"][idx])
    arr = synth.synthetic_image_array(src, int(g["seed"]), tuple(g[f"img{idx}.{kind}.size"]))
    return Image.fromarray(arr, "RGB")


@pytest.fixture(scope="module")
def tiny(golden_dir):
    g = load_golden(golden_dir, "tiny_seed1.npz")
    cfg, sd, model = build("tiny", int(g["seed"]))
    return g, cfg, sd, model


def test_synthetic_weights_identical_on_cpu_and_gpu():
    cfg = get_config("tiny")
    a = synth.synthetic_state_dict(cfg, seed=5, device="cpu")
    b = synth.synthetic_state_dict(cfg, seed=5, device="cuda")
    for k in a:
        assert torch.equal(a[k], b[k].cpu()), k


def test_vis_enc_seam_matches_reference(tiny):
    g, cfg, sd, model = tiny
    from oracle import moondream_oracle as O

    arr = np.array(golden_image(g, 0))
    x = O.normalize_crops(np.stack([arr, arr])).cuda()  # what prepare_crops hands to _vis_enc
    out = model._vis_enc(x)
    # two correct bf16 evaluations of the 27-layer stack sit ~1.0e-2 apart (oracle vs reference: 1.02e-2)
    compare("vit.out vs reference", out, bits_to_bf16(g["img0.vit.out"]), 1.5e-2)
    orc = O.vision_encoder(x.cpu(), {k: v.cpu() for k, v in sd.items()}, cfg)
    compare("vit.out vs oracle", out, orc, 1.5e-2)
    # identical crops must give identical features (no cross-row leakage in any kernel)
    assert torch.equal(out[0], out[1])


def test_vis_proj_seam_matches_reference(tiny):
    g, cfg, sd, model = tiny
    feats = bits_to_bf16(g["img0.vit.out"]).cuda()
    grid = feats[1].view(27, 27, -1)
    out = model._vis_proj(feats[0], grid)
    compare("vis.proj (from reference vit.out)", out, bits_to_bf16(g["img0.vis.proj"]), 5e-3)


def test_encode_image_kv_matches_reference(tiny):
    g, cfg, sd, model = tiny
    enc = model.encode_image(golden_image(g, 0))
    assert enc.pos == int(g["img0.cap.pos"]) == 730
    rs = int(g["kv_row_stride"])
    L = cfg.text.n_layers
    assert len(enc.caches) == L and enc.caches[0][0].shape == (1, cfg.text.n_kv_heads, 730, 64)
    for li in (0, L - 1):
        k, v = enc.caches[li]
        compare(f"k{li}", k[0, :, ::rs], bits_to_bf16(g[f"img0.cap.k{li}"]), 1.5e-2)
        compare(f"v{li}", v[0, :, ::rs], bits_to_bf16(g[f"img0.cap.v{li}"]), 1.5e-2)


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_greedy_token_ids_bit_exact_vs_reference(tiny, idx):
    g, cfg, sd, model = tiny
    ref = g[f"img{idx}.cap.tokens"].tolist()
    got = model.batch_generate_ids([golden_image(g, idx)], [g[f"img{idx}.cap.prompt"].tolist()], max_tokens=len(ref))
    margins = g[f"img{idx}.cap.margins"]
    assert got[0] == ref, f"min reference margin {margins.min():.3f}: {got[0]} vs {ref}"


def test_batched_equals_sequential_and_reference(tiny):
    g, cfg, sd, model = tiny
    images = [golden_image(g, i) for i in range(3)]
    prompts = [g[f"img{i}.cap.prompt"].tolist() for i in range(3)]
    n = len(g["img0.cap.tokens"])
    batched = model.batch_generate_ids(images, prompts, max_tokens=n)
    for i in range(3):
        assert batched[i] == g[f"img{i}.cap.tokens"].tolist()
        assert batched[i] == model.batch_generate_ids([images[i]], [prompts[i]], max_tokens=n)[0]


def test_fused_prefill_equals_two_passes(tiny):
    """batch_generate over raw images prefills [bos | image | prompt] in ONE decoder pass; the reference does two
    (encode_image, then the prompt: moondream.py:228-262, 280-321).  Same ids (== the reference's on the goldens); K / V
    rows agree within bf16 noise, not bit for bit: a 5-row pass runs the decode-regime split-K kernels where the fused
    pass runs the big tiles, and the attention's deferred rescaling is decided per wave of query rows.  Ragged prompts
    included."""
    g, cfg, sd, model = tiny
    images = [golden_image(g, i) for i in range(3)]
    prompts = [g[f"img{i}.cap.prompt"].tolist() for i in range(3)]
    prompts[1] = prompts[1] + prompts[1][:2]  # a second prompt length -> two groups
    n = 6
    try:
        model.fused_prefill = False
        a = model.batch_generate_ids(images, prompts, max_tokens=n, ignore_eos=True)
        ka, va = model._kv_k[:, :3].clone(), model._kv_v[:, :3].clone()
        model.fused_prefill = True
        b = model.batch_generate_ids(images, prompts, max_tokens=n, ignore_eos=True)
        kb, vb = model._kv_k[:, :3], model._kv_v[:, :3]
    finally:
        model.fused_prefill = True
    assert a == b
    assert a[0] == g["img0.cap.tokens"].tolist()[:n] and a[2] == g["img2.cap.tokens"].tolist()[:n]  # == the reference's
    n_pos = 730 + min(len(p) for p in prompts)
    compare("image K rows", kb[:, :, :, :730], ka[:, :, :, :730], 5e-3)
    compare("image V rows", vb[:, :, :, :730], va[:, :, :, :730], 5e-3)
    compare("prompt K rows", kb[:, :, :, 730:n_pos], ka[:, :, :, 730:n_pos], 1e-2)
    compare("prompt V rows", vb[:, :, :, 730:n_pos], va[:, :, :, 730:n_pos], 1e-2)


def test_rope_kv_write_in_gemm_epilogue_is_bit_identical(tiny):
    """Prefill: RoPE + KV-cache write applied in the epilogue of the fused qkv|fc1 GEMM (four-wave kernel, MD_EPI_QKV_ROPE)
    against the separate rope_kv_kernel (rope.py:20-48, text.py:42-46): hidden states, rotated K rows and V rows bit for bit."""
    g, cfg, sd, model = tiny
    t = cfg.text
    gen = torch.Generator().manual_seed(11)
    x = (torch.randn(5, 735, t.dim, generator=gen) * 0.7).to(BF16).cuda()
    model._ensure_batch(8)
    outs = []
    try:
        model.lib.md_gemm_set_tuning(b"tile", 20)  # the four-wave kernel for every layer (the tiny shapes would pick smaller tiles)
        for fuse in (0, 1):
            model.lib.md_gemm_set_tuning(b"rope_fuse", fuse)
            with torch.inference_mode():
                model._kv_k[:, 2:7].zero_()
                model._kv_v[:, 2:7].zero_()
                h = model._text_forward(x, [0, 3, 0, 1, 0], 2)
                torch.cuda.synchronize()
                outs.append((h.clone(), model._kv_k[:, 2:7, :, :740].clone(), model._kv_v[:, 2:7, :, :740].clone()))
    finally:
        model.lib.md_gemm_set_tuning(b"tile", -1)
        model.lib.md_gemm_set_tuning(b"rope_fuse", 1)
    assert float(outs[0][1].float().abs().sum()) > 0
    for a, b, name in zip(outs[0], outs[1], ("hidden", "K slab", "V slab")):
        assert torch.equal(a, b), name


def test_fp8_mode_rope_kv_write_in_gemm_epilogue_is_bit_identical(tiny):
    """The same for the FP8 mode's tile GEMM (gemm_f8_kernel<MD_EPI_QKV_ROPE>): hidden states, the bf16 K / V rows AND their
    e4m3 copies equal the three-kernel path's (GEMM, rope_kv_kernel, kv-quantise pass) bit for bit."""
    g, cfg, sd, model = tiny
    t = cfg.text
    gen = torch.Generator().manual_seed(12)
    x = (torch.randn(5, 735, t.dim, generator=gen) * 0.7).to(BF16).cuda()
    model._ensure_batch(8)
    model.enable_fp8([golden_image(g, i) for i in range(3)], kv_cache=True)
    outs = []
    try:
        for fuse in (0, 1):
            model.lib.md_gemm_set_tuning(b"rope_fuse", fuse)
            with torch.inference_mode():
                for buf in (model._kv_k, model._kv_v, model._kv_k8, model._kv_v8):
                    buf[:, 2:7].zero_()
                h = model._text_forward(x, [0, 3, 0, 1, 0], 2)
                torch.cuda.synchronize()
                outs.append((h.clone(), model._kv_k[:, 2:7, :, :740].clone(), model._kv_v[:, 2:7, :, :740].clone(),
                             model._kv_k8[:, 2:7, :, :740].clone(), model._kv_v8[:, 2:7, :, :740].clone()))
    finally:
        model.lib.md_gemm_set_tuning(b"rope_fuse", 1)
        model.enable_fp8(on=False)
    assert float(outs[0][1].float().abs().sum()) > 0 and int((outs[0][3] != 0).sum()) > 0
    for a, b, name in zip(outs[0], outs[1], ("hidden", "K slab", "V slab", "e4m3 K slab", "e4m3 V slab")):
        assert torch.equal(a, b), name


def test_dedup_identical_crops_is_bit_identical(tiny):
    """An image that fits one crop has a local crop equal to its global crop; with dedup_identical_crops the encoder
    runs once per distinct crop.  The projected embeddings must be the same bits as with both crops encoded, in a batch
    that mixes such images with a multi-crop one."""
    g, cfg, sd, model = tiny
    from moondream_amd import synth

    small = [golden_image(g, i) for i in range(2)]
    assert small[0].size == (378, 378)
    big = synth.synthetic_image(5, 1).resize((700, 500))
    images = [small[0], big, small[1]]
    with torch.inference_mode():
        try:
            model.dedup_identical_crops = False
            a = model._run_vision_encoder_batch(images).clone()
            model.dedup_identical_crops = True
            b = model._run_vision_encoder_batch(images).clone()
        finally:
            model.dedup_identical_crops = False
    assert torch.equal(a, b)


def test_decode_steps_over_more_than_64_sequences(tiny):
    """> 64 sequences per decode step run as blocks of 64 through the decode-regime kernels
    (md_text_forward); every sequence must still produce the reference's ids."""
    g, cfg, sd, model = tiny
    n_seq = 70
    images = [golden_image(g, i % 3) for i in range(n_seq)]
    prompts = [g[f"img{i % 3}.cap.prompt"].tolist() for i in range(n_seq)]
    n = len(g["img0.cap.tokens"])
    out = model.batch_generate_ids(images, prompts, max_tokens=n)
    for i in range(n_seq):
        assert out[i] == g[f"img{i % 3}.cap.tokens"].tolist(), i


def test_hipgraph_decode_equals_eager(tiny):
    """compile() replays the device-resident decode steps from a captured hipGraph:
    same ids as the eager path, on first use (capture) and on replay."""
    g, cfg, sd, model = tiny
    images = [golden_image(g, i) for i in range(3)]
    prompts = [g[f"img{i}.cap.prompt"].tolist() for i in range(3)]
    n = len(g["img0.cap.tokens"])
    eager = model.batch_generate_ids(images, prompts, max_tokens=n)
    model.compile()
    try:
        first = model.batch_generate_ids(images, prompts, max_tokens=n)   # eager chunk + capture
        replay = model.batch_generate_ids(images, prompts, max_tokens=n)  # graph replay
        single = model.batch_generate_ids(images[:1], prompts[:1], max_tokens=n)
        single2 = model.batch_generate_ids(images[:1], prompts[:1], max_tokens=n)
    finally:
        model.use_graphs = False
    assert first == eager and replay == eager
    assert single[0] == eager[0] and single2[0] == eager[0]


def test_pipelined_batches_equal_sequential(tiny):
    """Encode of batch k+1 overlapped with the decode of batch k on two streams:
    every batch's ids equal the sequential path's (and the reference's).  The graph pass runs three times: a decode
    graph's SECOND and later replays, in generator runs after the one that captured it, are the case that broke in
    round 3 (a captured hipMemsetAsync -- the split-K ticket reset -- ran out of order; profiles/r03_stale_graph_replay.txt)."""
    g, cfg, sd, model = tiny
    images = [golden_image(g, i) for i in range(3)]
    prompts = [g[f"img{i}.cap.prompt"].tolist() for i in range(3)]
    n = len(g["img0.cap.tokens"])
    ref = [g[f"img{i}.cap.tokens"].tolist() for i in range(3)]
    batches = [(images, prompts), (images[::-1], prompts[::-1]), (images[:2] + images[:1], prompts[:2] + prompts[:1]), (images, prompts)]
    for use_graphs in (False, True, True, True):
        model.use_graphs = use_graphs
        try:
            outs = list(model.batch_generate_ids_pipelined(batches, max_tokens=n))
        finally:
            model.use_graphs = False
        assert outs[0] == ref and outs[3] == ref
        assert outs[1] == ref[::-1]
        assert outs[2] == ref[:2] + ref[:1]


def test_teacher_forced_logits_vs_reference(tiny):
    g, cfg, sd, model = tiny
    enc = model.encode_image(golden_image(g, 0))
    model.load_encoded_image(enc)
    ref_tokens = g["img0.cap.tokens"].tolist()
    ref_logits = bits_to_bf16(g["img0.cap.step_logits"]).float()
    logits, hidden, pos = model._prefill_prompts([g["img0.cap.prompt"].tolist()], enc.pos, 0)
    compare("prompt hidden", hidden[0], bits_to_bf16(g["img0.cap.prompt_hidden"]), 2e-2)
    for i, tok in enumerate(ref_tokens):
        lg = logits[0].float().cpu()
        if i > 0:
            lg[cfg.tokenizer.answer_id] = float("-inf")
        ok = torch.isfinite(ref_logits[i])
        err = float((lg[ok] - ref_logits[i][ok]).abs().max())
        assert err <= 0.5, (i, err)
        if float(g["img0.cap.margins"][i]) > 0.5:
            assert int(torch.argmax(lg)) == int(g["img0.cap.argmaxes"][i]), i
        emb = model._embed(torch.tensor([[tok]]))
        pos_ids = torch.tensor([pos], dtype=torch.long)
        logits, _ = model._decode_one_tok(emb, None, pos_ids, None)  # the reference's seam signature
        pos += 1


def test_vqa_32_token_prompt_margin_aware(tiny):
    g, cfg, sd, model = tiny
    ref = g["img0.vqa.tokens"].tolist()
    got = model.batch_generate_ids([golden_image(g, 0, "vqa")], [g["img0.vqa.prompt"].tolist()], max_tokens=len(ref))[0]
    margins = g["img0.vqa.margins"]
    for i, (a, b) in enumerate(zip(got, ref)):
        if a != b:
            assert margins[i] <= 0.5, f"step {i}: {a} vs {b} at reference margin {margins[i]}"
            break


def test_text_api_and_streaming(tiny):
    g, cfg, sd, model = tiny
    ref = g["img0.cap.tokens"].tolist()
    settings = {"temperature": 0, "max_tokens": len(ref)}
    out = model.caption(golden_image(g, 0), settings=settings)["caption"]
    assert [int(t) for t in out.split()] == ref
    pieces = list(model.caption(golden_image(g, 0), stream=True, settings=settings)["caption"])
    assert [int(t) for t in "".join(pieces).split()] == ref
    with pytest.raises(ValueError):
        model.caption(golden_image(g, 0), length="epic")
    with pytest.raises(ValueError):
        model.encode_image("not an image")
    q = " ".join(str(t) for t in g["img0.vqa.prompt"].tolist()[3:-2])
    ans = model.query(golden_image(g, 0, "vqa"), q, settings={"temperature": 0, "max_tokens": 4})["answer"]
    assert [int(t) for t in ans.split()] == g["img0.vqa.tokens"].tolist()[:4] or g["img0.vqa.margins"][:5].min() <= 0.5
    sampled = model.caption(golden_image(g, 0), settings={"temperature": 0.5, "top_p": 0.3, "max_tokens": 6})["caption"]
    assert 1 <= len(sampled.split()) <= 6


def test_text_only_query_vs_reference(tiny, golden_dir):
    """query(image=None, question): reference moondream.py:564-575 -- BOS + query prefix at
    position 0 under the plain causal mask; ids recorded from the reference's public API."""
    g0, cfg, sd, model = tiny
    g = load_golden(golden_dir, "tiny_textonly.npz")
    # dirty the KV slabs first: a text-only query must not see an earlier image's keys
    model.caption(golden_image(g0, 0), settings={"temperature": 0, "max_tokens": 2})
    for i in range(int(g["n_cases"])):
        want = g[f"q{i}.tokens"].tolist()
        q = " ".join(str(t) for t in g[f"q{i}.question"].tolist())
        ans = model.query(None, q, settings={"temperature": 0, "max_tokens": len(want)})["answer"]
        assert [int(t) for t in ans.split()] == want
    with pytest.raises(ValueError):
        model.query(None, "1 2 3", spatial_refs=[(0.5, 0.5)])


def test_seam_honours_the_mask_the_reference_passes(tiny, golden_dir):
    """The reference's text-only query goes THROUGH the seam with a plain tril mask slice (moondream.py:571-575 ->
    304-309 -> self._prefill(prompt_emb, mask, pos_ids, lora); decode rows 472-474,515).  Driving _prefill /
    _decode_one_tok with exactly those tensors must reproduce the reference's ids (tiny_textonly.npz) and its
    logits; the prefix-LM slice for the same positions gives another result (so the mask is really read), and a mask
    that is neither rule is refused."""
    g0, cfg, sd, model = tiny
    g = load_golden(golden_dir, "tiny_textonly.npz")
    t = cfg.text
    ctx = t.max_context
    tril = torch.tril(torch.ones(1, 1, ctx, ctx, dtype=torch.bool)).to(model.device)   # moondream.py:573-575
    prefix_lm = tril.clone()
    prefix_lm[..., : t.prefix_attn, : t.prefix_attn] = 1                              # moondream.py:138-146
    tok = cfg.tokenizer
    model.caption(golden_image(g0, 0), settings={"temperature": 0, "max_tokens": 2})   # dirty the slabs
    for i in range(int(g["n_cases"])):
        want = g[f"q{i}.tokens"].tolist()
        ref_logits = bits_to_bf16(g[f"q{i}.step_logits"]).float()
        # moondream.py:564-575 (BOS + prefix), 586-591 (+ question + suffix), 604 (+ suffix again when not reasoning)
        prompt = ([tok.bos_id] + list(tok.templates["query"]["prefix"]) + g[f"q{i}.question"].tolist()
                  + list(tok.templates["query"]["suffix"]) + list(tok.templates["query"]["suffix"]))
        n = len(prompt)
        emb = model._embed(torch.tensor([prompt]))
        pos_ids = torch.arange(0, n, dtype=torch.long)
        hidden = model._prefill(emb, tril[:, :, 0:n, :], pos_ids, None)
        logits = model._lm_head(hidden)
        mask = torch.zeros(1, 1, ctx, dtype=torch.bool, device=model.device)           # moondream.py:472-474
        mask[:, :, :n] = 1
        pos, got = n, []
        for step in range(len(want)):
            lg = logits[0].float().cpu()
            if step > 0:
                lg[tok.answer_id] = float("-inf")
            ok = torch.isfinite(ref_logits[step])
            assert float((lg[ok] - ref_logits[step][ok]).abs().max()) <= 0.5, (i, step)
            nxt = int(torch.argmax(lg))
            got.append(nxt)
            mask[:, :, pos] = 1
            logits, _ = model._decode_one_tok(model._embed(torch.tensor([[nxt]])), mask, torch.tensor([pos], dtype=torch.long), None)
            pos += 1
        assert got == want, (i, got, want)
        # the prefix-LM slice of these rows would let them see keys [n, 730) that the pass does not write: refused
        with pytest.raises(ValueError):
            model._prefill(emb, prefix_lm[:, :, 0:n, :], pos_ids, None)
    # image prefill through the seam with the reference's own slice == the engine's encode_image
    enc = model.encode_image(golden_image(g0, 0))
    img_emb = model._run_vision_encoder(golden_image(g0, 0))
    bos = model._embed(torch.tensor([[tok.bos_id]]))
    x = torch.cat([bos, img_emb[None]], dim=1)
    model._prefill(x, prefix_lm[:, :, 0 : x.shape[1], :], torch.arange(x.shape[1], dtype=torch.long), None)
    assert torch.equal(model._kv_k[:, 0, :, : enc.pos], torch.stack([k[0] for k, _ in enc.caches]))
    # under the causal slice the same rows give another cache (and it is accepted: it is one of the two rules)
    model._prefill(x, tril[:, :, 0 : x.shape[1], :], torch.arange(x.shape[1], dtype=torch.long), None)
    assert not torch.equal(model._kv_k[1:, 0, :, : enc.pos], torch.stack([k[0] for k, _ in enc.caches])[1:])
    bad = tril[:, :, 0:4, :].clone()
    bad[0, 0, 3, 1] = False
    with pytest.raises(ValueError):
        model._prefill(x[:, :4], bad, torch.arange(4, dtype=torch.long), None)
    with pytest.raises(ValueError):
        model._prefill(x[:, :4], None, torch.tensor([0, 1, 3, 4]), None)


def vqa64_vs_reference(model, cfg, sd, imgs64, golden_dir):
    """BASELINE configs[1] / the north star's "32-token prompts" AT BENCH SCALE: the 64 seed-1 images with the 32-id question
    prompts bench.py's vqa32 leg times, B = 64, 32 greedy tokens, against the unmodified reference's ids for exactly these
    (image, prompt) pairs (tests/golden/md2b_vqa64.npz: _generate_answer, the loop behind query(), moondream.py:541-618) --
    the same measured licence, per-decision teacher-forced check and second-oracle floor as the caption configuration."""
    import bench
    from moondream_amd import parity as P

    gv = load_golden(golden_dir, "md2b_vqa64.npz")
    prompts = gv["prompt"].tolist()
    assert prompts == [synth.synthetic_vqa_prompt(cfg, i, int(gv["seed"])) for i in range(64)] and len(prompts[0]) == 32
    got = model.batch_generate_ids(imgs64, prompts, max_tokens=32, ignore_eos=True)
    topk = model.teacher_forced_logits(imgs64, prompts, gv["tokens"], gv["top8_idx"]).numpy()
    second = bench.second_oracle(cfg, sd, int(gv["seed"]), 32, "cuda", fixture="md2b_vqa64")
    floor = bench.exact_floor(64, 32, second)
    rep = P.parity_report(got, gv["tokens"].tolist(), gv["margins"], topk, gv["top8_val"], tokens=32, min_exact=floor, ref_topk_idx=gv["top8_idx"])
    print(f"vqa64 parity (32-id prompts, B=64): {rep['parity_exact']}/64 identical (second oracle {second['exact']}/64 -> floor {floor}); max "
          f"|logit err| {rep['parity_max_logit_err']:.4f} (p99 {rep['parity_p99_logit_err_ulps']:.1f} ulps) -> threshold {rep['parity_threshold']:.4f}; "
          f"teacher-forced: {rep['parity_tf_decisions_must_match']} must-match decisions, {rep['parity_tf_decisions_violations']} violations")
    assert second["max_logit_err"] <= 0.5
    assert rep["parity_ok"], rep["parity_note"]
    assert rep["parity_exact"] == 64 and rep["parity_tf_decisions_must_match"] == 64 * 33 and rep["parity_tf_decisions_violations"] == 0
    # the sequences whose every margin clears the licence are identical outright
    wide = [i for i in range(64) if float(gv["margins"][i].min()) > rep["parity_threshold"]]
    assert wide and all(got[i] == gv["tokens"][i].tolist() for i in wide), wide


def test_multicrop_images_vs_reference(golden_dir):
    g = load_golden(golden_dir, "tiny_multicrop.npz")
    cfg, sd, model = build("tiny", 3)
    for i in range(3):
        size = tuple(g[f"case{i}.size"])
        img = Image.fromarray(synth.synthetic_image_array(i, 3, size), "RGB")
        out = model._run_vision_encoder(img)
        compare(f"multicrop case{i} vis.proj", out, bits_to_bf16(g[f"case{i}.vis.proj"]), 2e-2)
    # two different tilings in one batch
    imgs = [Image.fromarray(synth.synthetic_image_array(i, 3, tuple(g[f"case{i}.size"])), "RGB") for i in range(2)]
    both = model._run_vision_encoder_batch(imgs)
    for i in range(2):
        compare(f"batched multicrop {i}", both[i], bits_to_bf16(g[f"case{i}.vis.proj"]), 2e-2)


def test_vit_error_against_fp64_truth_no_worse_than_reference(tiny):
    """The reference's bf16 ViT output is 9.6e-3 (rel-rms) away from an fp64 evaluation of the same
    network; the HIP path must not be further than 1.2x that (i.e. it is as accurate as the
    reference, not merely 'close to it')."""
    g, cfg, sd, model = tiny
    from oracle import moondream_oracle as O

    arr = np.array(golden_image(g, 0))
    x = O.normalize_crops(np.stack([arr, arr]))
    truth = vit_fp64(x, sd, cfg)
    rel = lambda a: float(((a.double().cpu() - truth) ** 2).mean().sqrt() / (truth ** 2).mean().sqrt())
    e_ref = rel(bits_to_bf16(g["img0.vit.out"]))
    e_hip = rel(model._vis_enc(x.cuda()))
    print(f"rel-rms vs fp64 truth: reference {e_ref:.3e}, hip {e_hip:.3e}")
    assert e_hip <= 1.2 * e_ref


# ------------------------------------------------------------------ region head, pinned to the reference
@pytest.fixture(scope="module")
def detect_gold(golden_dir, tiny):
    return load_golden(golden_dir, "tiny_detect.npz")


@pytest.mark.parametrize("fn", ["decode_coordinate", "encode_coordinate", "decode_size", "encode_size"])
def test_region_functions_match_reference_io_pairs(tiny, detect_gold, fn):
    """Every call the reference made to region.py:32-93 during detect / point, replayed on the HIP heads."""
    g0, cfg, sd, model = tiny
    g = detect_gold
    n = 0
    for case in ("detect0", "detect1", "point0", "point1"):
        key = f"{case}.{fn}.in"
        if key not in g.files:
            continue
        xin, want = bits_to_bf16(g[key]).cuda(), bits_to_bf16(g[f"{case}.{fn}.out"])
        if fn == "encode_coordinate":
            got = model.encode_coordinate(xin.reshape(-1, 1))
        elif fn == "encode_size":
            got = model.encode_size(xin.reshape(-1, 2))
        elif fn == "decode_size":
            got = model.decode_size(xin).reshape(xin.shape[0], -1)
        else:
            got = model.decode_coordinate(xin)
        compare(f"{case}.{fn}", got, want, 1e-2)
        n += xin.shape[0]
    assert n >= 4


@pytest.mark.parametrize("case", ["detect0", "detect1", "point0", "point1"])
def test_detect_point_vs_reference(tiny, detect_gold, case):
    """detect / point through the public API: objects equal the reference's (exact floats) up to the
    first decision whose reference margin is below 4 bf16 ulps."""
    g0, cfg, sd, model = tiny
    g = detect_gold
    kind = "detect" if case.startswith("detect") else "point"
    img = Image.fromarray(synth.synthetic_image_array(int(g[f"{case}.image_index"]), int(g["seed"]), (378, 378)), "RGB")
    obj = " ".join(str(t) for t in g["object_ids"].tolist())
    res = (model.detect if kind == "detect" else model.point)(img, obj, settings={"max_objects": int(g["max_objects"])})
    objs = res["objects" if kind == "detect" else "points"]
    ref = g[f"{case}.objects"]
    n_ok = leading_wide_objects(g[f"{case}.margins"], 4.0)
    assert n_ok >= 1 and len(objs) >= n_ok
    keys = ("x_min", "y_min", "x_max", "y_max") if kind == "detect" else ("x", "y")
    for k in range(n_ok):
        assert [objs[k][f] for f in keys] == ref[k].tolist(), (case, k, objs[k], ref[k])


def test_batch_detect_equals_sequential(tiny, detect_gold):
    """B images in lockstep (ragged object prompts, mixed EncodedImage / PIL inputs) == one at a time."""
    g0, cfg, sd, model = tiny
    g = detect_gold
    imgs = [Image.fromarray(synth.synthetic_image_array(int(g[f"{c}.image_index"]), int(g["seed"]), (378, 378)), "RGB")
            for c in ("detect0", "detect1", "point0")]
    objects = ["7 8", "7 8 9 10", "5"]
    st = {"max_objects": 2}
    seq = [model.detect(im, o, settings=st) for im, o in zip(imgs, objects)]
    enc1 = model.encode_image(imgs[1])
    got = model.batch_detect([imgs[0], enc1, imgs[2]], objects, settings=st)
    assert got == seq
    pts = model.batch_point(imgs, objects, settings=st)
    assert pts == [model.point(im, o, settings=st) for im, o in zip(imgs, objects)]


def test_spatial_ref_query_vs_reference(tiny, detect_gold):
    """query(image, question, spatial_refs=[point, box]): moondream.py:577-604 + 293-301."""
    g0, cfg, sd, model = tiny
    g = detect_gold
    img = Image.fromarray(synth.synthetic_image_array(int(g["spatial.image_index"]), int(g["seed"]), (378, 378)), "RGB")
    refs = [tuple(g["spatial.refs_point"].tolist()), tuple(g["spatial.refs_box"].tolist())]
    want = g["spatial.tokens"].tolist()
    q = " ".join(str(t) for t in g["spatial.question"].tolist())
    ans = model.query(img, q, spatial_refs=refs, settings={"temperature": 0, "max_tokens": len(want)})["answer"]
    assert [int(t) for t in ans.split()] == want


def test_reasoning_query_vs_reference(tiny, golden_dir):
    """query(..., reasoning=True): reasoning text and answer ids recorded from the reference's public API."""
    g0, cfg, sd, model = tiny
    g = load_golden(golden_dir, "tiny_reasoning.npz")
    for i in range(int(g["n_cases"])):
        img = Image.fromarray(synth.synthetic_image_array(int(g[f"case{i}.image_index"]), int(g["seed"]), (378, 378)), "RGB")
        res = model.query(img, "11 12 13", reasoning=True, settings={"temperature": 0, "max_tokens": int(g["max_tokens"])})
        assert [int(t) for t in res["reasoning"]["text"].split()] == g[f"case{i}.reasoning_tokens"].tolist()
        assert [int(t) for t in res["answer"].split()] == g[f"case{i}.answer_tokens"].tolist()
        assert len(res["reasoning"]["grounding"]) == int(g[f"case{i}.n_grounding"])
    # the grounding branch: a prompt whose first generated token is forced to be the coordinate token
    tk = cfg.tokenizer
    enc = model.encode_image(golden_image(g0, 0))
    model.load_encoded_image(enc)
    orig = model._pick
    seq = iter([tk.start_ground_points_id, tk.coord_id, tk.coord_id, tk.end_ground_id, tk.answer_id])
    model._pick = lambda logits, *a, **k: torch.tensor([next(seq)], dtype=torch.int32, device=logits.device)
    try:
        pos, text, grounding = model._generate_reasoning(torch.tensor([[1, 381, 2, 11, 3, tk.thinking_id]]), enc.pos, {"temperature": 0, "max_tokens": 8})
    finally:
        model._pick = orig
    assert [int(t) for t in text.split()] == [tk.start_ground_points_id, tk.coord_id, tk.coord_id, tk.end_ground_id]
    assert len(grounding) == 1 and len(grounding[0]["points"]) == 1 and all(0.0 <= c < 1.0 for c in grounding[0]["points"][0])


def test_lora_variant_vs_reference(tiny, golden_dir):
    """settings={"variant": id}: ids equal the reference's run with the same (seeded) LoRA variant, through caption(),
    the batched engine and detect(); an unknown variant fails loudly instead of downloading."""
    g0, cfg, sd, model = tiny
    g = load_golden(golden_dir, "tiny_lora.npz")
    model.register_variant("synthetic", synth.synthetic_lora(cfg, seed=int(g["seed"]), rank=int(g["rank"]), device="cuda"))
    images = []
    for i in range(int(g["n_cases"])):
        img = Image.fromarray(synth.synthetic_image_array(int(g[f"case{i}.image_index"]), int(g["seed"]), (378, 378)), "RGB")
        images.append(img)
        want = g[f"case{i}.tokens"].tolist()
        st = {"temperature": 0, "max_tokens": len(want), "variant": "synthetic"}
        assert [int(t) for t in model.caption(img, settings=st)["caption"].split()] == want
        base = model.caption(img, settings={"temperature": 0, "max_tokens": len(want)})["caption"]
        assert [int(t) for t in base.split()] == g[f"case{i}.base_tokens"].tolist()
    prompt = cfg.tokenizer.templates["caption"]["normal"]
    n = len(g["case0.tokens"])
    got = model.batch_generate_ids(images, [prompt] * len(images), max_tokens=n, variant="synthetic")
    assert got == [g[f"case{i}.tokens"].tolist() for i in range(len(images))]
    objs = model.detect(images[0], "7 8", settings={"max_objects": 1, "variant": "synthetic"})["objects"]
    assert len(objs) <= 1
    with pytest.raises(FileNotFoundError):
        model.caption(images[0], settings={"variant": "no-such-variant-anywhere"})


def test_rowwise_add_and_gelu_kernels():
    from moondream_amd import _lib
    import ctypes as C

    lib = _lib.load()
    a = torch.randn(77, 704, generator=torch.Generator().manual_seed(1)).to(BF16).cuda()
    b = torch.randn(77, 704, generator=torch.Generator().manual_seed(2)).to(BF16).cuda()
    out = torch.empty_like(a)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.md_add_bf16(a.data_ptr(), 704, b.data_ptr(), 704, out.data_ptr(), 704, 77, 704, st))
    torch.cuda.synchronize()
    assert torch.equal(out, a + b)
    _lib.check(lib.md_gelu_bf16(a.data_ptr(), 704, out.data_ptr(), 704, 77, 704, st))
    torch.cuda.synchronize()
    compare("gelu", out, torch.nn.functional.gelu(a.float(), approximate="tanh").to(BF16), 3e-3, 2e-2)


def test_decode_step_of_128_sequences_is_one_pass_with_the_bits_of_two_passes_of_64(tiny):
    """Round 6: a decode step of 65 .. 128 sequences streams the weights ONCE (128 x 64 weight-streaming tile for the fused
    qkv|fc1 layer, 128-row K-slice partials + fused tail for proj / fc2, by-shape lm_head) instead of once per block of 64.
    Same K order per output element: ids, the step's logits and the K / V rows it writes equal, bit for bit, those of the
    same sequences decoded in batches of 64 -- also for a ragged 100 (one tall pass) and 160 (128 + 32)."""
    g, cfg, sd, model = tiny
    pr = g["img0.cap.prompt"].tolist()
    imgs = [synth.synthetic_image(i, int(g["seed"])) for i in range(160)]
    n = 6
    ref = []
    for i0 in range(0, 160, 64):
        part = imgs[i0 : i0 + 64]
        ref += model.batch_generate_ids(part, [pr] * len(part), max_tokens=n, ignore_eos=True)
    kv_ref = None
    for b in (128, 100, 160):
        got = model.batch_generate_ids(imgs[:b], [pr] * b, max_tokens=n, ignore_eos=True)
        assert got == ref[:b], (b, [i for i in range(b) if got[i] != ref[i]][:5])
    # logits and cache rows of one step: 128 rows at once vs 64 + 64
    p0 = 730 + len(pr)
    model.batch_generate_ids(imgs[:128], [pr] * 128, max_tokens=2, ignore_eos=True)
    k128, lg128 = model._kv_k[:, :128, :, p0 : p0 + 2].clone(), model._decode_logits(128)[:128].clone()
    model.batch_generate_ids(imgs[:64], [pr] * 64, max_tokens=2, ignore_eos=True)
    k_a, lg_a = model._kv_k[:, :64, :, p0 : p0 + 2].clone(), model._decode_logits(64)[:64].clone()
    model.batch_generate_ids(imgs[64:128], [pr] * 64, max_tokens=2, ignore_eos=True)
    k_b, lg_b = model._kv_k[:, :64, :, p0 : p0 + 2].clone(), model._decode_logits(64)[:64].clone()
    assert torch.equal(k128[:, :64], k_a) and torch.equal(k128[:, 64:], k_b)
    assert torch.equal(lg128[:64], lg_a) and torch.equal(lg128[64:], lg_b)


# ------------------------------------------------------------------ per-layer drift profile (round 6)
def _layer_profile(model, cfg, g):
    """HIP activation after every ViT block k (as post_ln(x_k): md_vit_encode with the block list cut after block k) and after
    every decoder block of the image prefill (md_text_forward cut after block k, driven from the REFERENCE's [bos | image]
    embeddings), as relative RMS distances from the reference's tensors in tiny_layers.npz."""
    import ctypes as C
    from moondream_amd import _lib
    from oracle import moondream_oracle as O

    lib, dev = model.lib, model.device
    v, t = cfg.vision, cfg.text
    arr = synth.synthetic_image_array(int(g["image_index"]), int(g["seed"]), (378, 378))
    crops = O.normalize_crops(np.stack([arr, arr])).to(dev).contiguous()
    ts = int(g["vit_token_stride"])
    st = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def rel(a, b):
        a, b = a.detach().float().cpu(), b.float()
        return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())

    vit_d, txt_d = [], []
    for k in range(1, v.enc_n_layers + 1):
        cut = type(model.w.vit).from_buffer_copy(model.w.vit)
        cut.n_layers = k
        out = torch.empty(2, v.n_patches, v.enc_dim, dtype=BF16, device=dev)
        ws = model._workspace(lib.md_vit_workspace_bytes(C.byref(cut), 2))
        _lib.check(lib.md_vit_encode(C.byref(cut), crops.data_ptr(), _lib.MD_CROPS_BF16_CHW, 2, out.data_ptr(), ws.data_ptr(), ws.numel(), st()))
        vit_d.append(rel(out[:, ::ts], bits_to_bf16(g[f"vit.ln_block{k - 1}"])))
    x0 = bits_to_bf16(g["text.input"]).to(dev).reshape(1, -1, t.dim).contiguous()
    pos0 = torch.zeros(1, dtype=torch.int32, device=dev)
    model._ensure_batch(1)
    for k in range(1, t.n_layers + 1):
        cut = type(model.w.text).from_buffer_copy(model.w.text)
        cut.n_layers = k
        hidden = torch.empty_like(x0)
        kv = model._kv_struct(0)
        ws = model._workspace(lib.md_text_workspace_bytes(C.byref(cut), 1, x0.shape[1]))
        _lib.check(lib.md_text_forward(C.byref(cut), x0.data_ptr(), hidden.data_ptr(), 1, x0.shape[1], pos0.data_ptr(), C.byref(kv),
                                       ws.data_ptr(), ws.numel(), st()))
        txt_d.append(rel(hidden[0], bits_to_bf16(g[f"text.block{k - 1}"])))
    return np.array(vit_d), np.array(txt_d)


LAYER_DRIFT_FACTOR = 1.3   # HIP drift at a layer <= 1.3 x the oracle's drift from the reference at that layer ...
LAYER_DRIFT_SLACK = 5e-4   # ... + this (the first layers' drift is a handful of flipped roundings: a ratio of two tiny numbers)


def test_per_layer_drift_profile_and_mutation_is_caught_where_it_happens(tiny, golden_dir):
    """Each ViT block and each decoder block separately (review, round 5): an end-of-stack tolerance of 1.5e-2 after 27 blocks
    forgives one bad layer whose error the following layers dilute.  The reference's activation after EVERY block is in
    tiny_layers.npz together with the oracle's drift from it; the HIP path must stay within 1.3 x that drift at every layer.
    Then the lost-head mutation of the 2B test (one attention head of ViT block 13 lost: its proj input columns zeroed) is
    planted here: the profile must trip AT block 13 -- not before, and visibly (> 2 x the bound) -- and be clean again after."""
    g0, cfg, sd, model = tiny
    g = load_golden(golden_dir, "tiny_layers.npz")
    bound_v = LAYER_DRIFT_FACTOR * g["oracle_drift_vit"] + LAYER_DRIFT_SLACK
    bound_t = LAYER_DRIFT_FACTOR * g["oracle_drift_text"] + LAYER_DRIFT_SLACK
    vit_d, txt_d = _layer_profile(model, cfg, g)
    print("ViT  drift HIP   :", " ".join(f"{d:.2e}" for d in vit_d))
    print("ViT  drift oracle:", " ".join(f"{d:.2e}" for d in g["oracle_drift_vit"]))
    print("text drift HIP   :", " ".join(f"{d:.2e}" for d in txt_d), "| oracle:", " ".join(f"{d:.2e}" for d in g["oracle_drift_text"]))
    assert (vit_d <= bound_v).all(), [(k, vit_d[k], bound_v[k]) for k in range(len(vit_d)) if vit_d[k] > bound_v[k]]
    assert (txt_d <= bound_t).all(), [(k, txt_d[k], bound_t[k]) for k in range(len(txt_d)) if txt_d[k] > bound_t[k]]
    hd = cfg.vision.enc_dim // cfg.vision.enc_n_heads
    w = model.w._vit_packed[13]["proj"].w            # [n_pad][k_pad] bf16: input feature f of the layer = column f
    keep = w.clone()
    try:
        w[:, hd : 2 * hd].zero_()                    # head 1 of block 13 lost
        torch.cuda.synchronize()
        bad_v, _ = _layer_profile(model, cfg, g)
    finally:
        w.copy_(keep)
        torch.cuda.synchronize()
    over = [k for k in range(len(bad_v)) if bad_v[k] > bound_v[k]]
    print("ViT  drift with head 1 of block 13 lost:", " ".join(f"{d:.2e}" for d in bad_v))
    assert over and over[0] == 13 and bad_v[13] > 2 * bound_v[13], (over, bad_v[13], bound_v[13])
    assert np.array_equal(bad_v[:13], vit_d[:13])    # the layers before it are untouched, bit for bit
    again_v, again_t = _layer_profile(model, cfg, g)
    assert np.array_equal(again_v, vit_d) and np.array_equal(again_t, txt_d)


# ------------------------------------------------------------------ batched string API + HF wrapper
def test_batch_generate_strings_ragged_equals_sequential(tiny):
    """batch_generate / batch_query / batch_caption (the names BASELINE.json uses) with questions of
    different token counts: ONE lockstep decode, answers equal query(...) one at a time."""
    g, cfg, sd, model = tiny
    images = [golden_image(g, i) for i in range(3)] + [golden_image(g, 0)]
    questions = ["11 12 13", "21 22 23 24 25 26 27", "31", "11 12 13"]
    st = {"temperature": 0, "max_tokens": 10}
    seq = [model.query(im, q, settings=st)["answer"] for im, q in zip(images, questions)]
    calls = []
    orig = model._text_forward
    model._text_forward = lambda x, *a, **k: (calls.append(tuple(x.shape)), orig(x, *a, **k))[1]
    try:
        got = model.batch_generate(images, questions, st)
    finally:
        model._text_forward = orig
    assert got == seq
    assert got == model.batch_query(images, questions, st)
    prompt_prefills = [c for c in calls if c[1] not in (1, 730)]
    assert len(prompt_prefills) == 3, calls  # one per distinct question length, not one per image
    caps = model.batch_generate(images[:3], None, {"temperature": 0, "max_tokens": 8})
    assert caps == [model.caption(im, settings={"temperature": 0, "max_tokens": 8})["caption"] for im in images[:3]]
    assert caps == model.batch_caption(images[:3], "normal", {"temperature": 0, "max_tokens": 8})


def test_batched_sampled_decode_follows_the_reference_rule(tiny):
    """Round 6: the lockstep engine SAMPLES when temperature > 0 -- what the reference's loop of query() calls does at its
    default settings (temperature 0.5, top_p 0.3: moondream.py:50-53, 313-318, 521-528; hf_moondream.py:99-103).
    (a) temperature -> 0+ with any top_p keeps only the argmax: ids equal the greedy ids;  (b) the draws are seeded
    (same generator state -> same ids), different seeds differ at a hot setting;  (c) DISTRIBUTION: 64 copies of one
    (image, prompt) decoded in lockstep at T = 1.5, top_p = 0.95 draw their FIRST TWO tokens from exactly the distribution
    the reference's rule assigns -- the empirical frequencies of the first token (drawn from the prompt pass's logits) and
    of the second token given the most frequent first one (drawn inside the loop from the decode step's logits) match
    softmax / top-p / renormalise of the logits the greedy path reports, by a chi-square bound;  (d) hipGraph replay and
    eager launches draw the same ids from the same uniforms;  (e) batch_query / batch_caption use the reference's default
    settings when none are given and stay greedy at {"temperature": 0}."""
    g, cfg, sd, model = tiny
    img = golden_image(g, 0)
    prompt = g["img0.cap.prompt"].tolist()
    n = 8
    greedy = model.batch_generate_ids([img] * 2, [prompt] * 2, max_tokens=n)
    cold = model.batch_generate_ids([img] * 2, [prompt] * 2, max_tokens=n, temperature=1e-3, top_p=0.3,
                                    generator=torch.Generator(device="cuda").manual_seed(5))
    assert cold == greedy
    def hot(seed, b=6, **kw):
        return model.batch_generate_ids([img] * b, [prompt] * b, max_tokens=n, temperature=4.0, top_p=0.999,
                                        generator=torch.Generator(device="cuda").manual_seed(seed), **kw)
    a, a2, c = hot(1), hot(1), hot(2)
    assert a == a2 and a != c
    assert len({tuple(s) for s in a}) > 1            # sequences of one batch draw independently
    use_graphs = model.use_graphs
    try:
        model.use_graphs = not use_graphs
        model._graphs.clear()
        assert hot(1) == a                           # (d)
        assert hot(1) == a                           # ... and again from the captured graph
    finally:
        model.use_graphs = use_graphs
        model._graphs.clear()
    # (c) distribution of the first two tokens
    T, P = 1.5, 0.95
    def reference_rule(logits_bf16, suppress=None):
        lg = logits_bf16.clone()
        if suppress is not None:
            lg[suppress] = float("-inf")
        probs = torch.softmax(lg / T, dim=-1)        # bf16 like the reference's logits (moondream.py:316,526)
        srt, idx = torch.sort(probs, descending=True)
        cum = torch.cumsum(srt, dim=-1)
        srt[(cum - srt) > P] = 0.0
        srt.div_(srt.sum())
        return torch.zeros_like(probs).scatter_(0, idx, srt).float().cpu()
    enc = model.encode_image(img)
    model.load_encoded_image(enc)
    with torch.inference_mode():
        logits0, _, nxt, pos = model._prefill_prompt(torch.tensor([prompt]), enc.pos, 0.0, 0.0)
    p0 = reference_rule(logits0[0])
    draws = []
    for seed in range(12):
        draws += model.batch_generate_ids([img] * 64, [prompt] * 64, max_tokens=2, temperature=T, top_p=P,
                                          generator=torch.Generator(device="cuda").manual_seed(100 + seed))
    first = torch.tensor([d[0] for d in draws])
    assert float(p0[first].min()) > 0                 # nothing outside the top-p support is ever drawn
    def chi2_ok(counts, probs, nd):
        """Pearson chi-square of the observed counts against nd x probs: bins with an expectation >= 5 one by one, the rest
        lumped into one bin; accepted up to the mean + 5 standard deviations of the chi-square law (+ slack for the lump)."""
        keep = probs * nd >= 5
        exp = torch.cat([probs[keep] * nd, (probs[~keep].sum() * nd).reshape(1)])
        obs = torch.cat([counts[keep].float(), counts[~keep].sum().float().reshape(1)])
        nz = exp > 0
        chi2 = float(((obs[nz] - exp[nz]) ** 2 / exp[nz]).sum()) + (float("inf") if float(obs[~nz].sum()) > 0 else 0.0)
        dof = max(1, int(nz.sum()) - 1)
        return chi2 <= dof + 5 * (2 * dof) ** 0.5 + 5, (chi2, dof)
    ok, info = chi2_ok(torch.bincount(first, minlength=p0.numel()), p0, len(first))
    assert ok, info
    top = int(torch.bincount(first).argmax())
    with torch.inference_mode():
        emb = model._embed(torch.tensor([[top]]))
        mask = None
        logits1, _ = model._decode_one_tok(emb, mask, torch.tensor([pos]), None)
    p1 = reference_rule(logits1[0], suppress=cfg.tokenizer.answer_id)
    second = torch.tensor([d[1] for d in draws if d[0] == top and len(d) > 1])
    assert len(second) >= 30
    assert float(p1[second].min()) > 0
    ok, info = chi2_ok(torch.bincount(second, minlength=p1.numel()), p1, len(second))
    assert ok, info
    # (e) the string API: reference defaults unless told otherwise
    qs = ["11 12 13"] * 3
    assert model.batch_query([img] * 3, qs, {"temperature": 0, "max_tokens": 6}) == [model.query(img, qs[0], settings={"temperature": 0, "max_tokens": 6})["answer"]] * 3
    sampled = model.batch_query([img] * 3, qs, {"max_tokens": 6, "generator": torch.Generator(device="cuda").manual_seed(3)})
    assert len(sampled) == 3 and all(isinstance(x, str) for x in sampled)


def test_prefetched_crops_are_used_and_identical(tiny):
    """prefetch_crops: the host tiling of a batch started ahead of the call that encodes it -- same ids, the entry is
    consumed, at most two batches are held, unused ones can be dropped."""
    g, cfg, sd, model = tiny
    images = [golden_image(g, i) for i in range(3)]
    prompts = [cfg.tokenizer.templates["caption"]["normal"]] * 3
    base = model.batch_generate_ids(images, prompts, max_tokens=6)
    model.prefetch_crops(images)
    assert len(model._prefetched_crops) == 1
    assert model.batch_generate_ids(images, prompts, max_tokens=6) == base
    assert not model._prefetched_crops
    model.prefetch_crops(images)
    model.prefetch_crops(list(images))  # the same image objects: one entry
    assert len(model._prefetched_crops) == 1
    for _ in range(3):
        model.prefetch_crops([im.copy() for im in images])
    assert len(model._prefetched_crops) == 2
    model.discard_prefetched_crops()
    assert not model._prefetched_crops
    assert model.batch_generate_ids(images, prompts, max_tokens=6) == base


def test_mixed_encoded_and_raw_images_in_one_batch(tiny):
    """an EncodedImage next to raw PIL images must keep its own KV slot."""
    g, cfg, sd, model = tiny
    images = [golden_image(g, i) for i in range(3)]
    prompts = [g[f"img{i}.cap.prompt"].tolist() for i in range(3)]
    n = len(g["img0.cap.tokens"])
    ref = [g[f"img{i}.cap.tokens"].tolist() for i in range(3)]
    enc0, enc2 = model.encode_image(images[0]), model.encode_image(images[2])
    assert model.batch_generate_ids([enc0, images[1], images[2]], prompts, max_tokens=n) == ref
    assert model.batch_generate_ids([images[0], images[1], enc2], prompts, max_tokens=n) == ref
    assert model.batch_generate_ids([enc0, images[1], enc2], prompts, max_tokens=n) == ref


def test_kv_slab_growth_keeps_loaded_slots_and_context_is_bounded(tiny):
    g, cfg, sd, _ = tiny
    from moondream_amd.moondream import MoondreamModel, IdTokenizer

    model = MoondreamModel(cfg, sd, device="cuda", tokenizer=IdTokenizer(), max_batch=1)
    enc1 = model.encode_image(golden_image(g, 1))
    enc = model.encode_image(golden_image(g, 0))   # encode_image works in slot 0: do it before loading slots
    model.load_encoded_image(enc, 0)
    model.load_encoded_image(enc1, 2)               # grows the slabs to 3 slots
    assert model._max_batch >= 3
    assert torch.equal(model._kv_k[0, 0:1, :, :730], enc.caches[0][0])   # slot 0 survived the growth
    with pytest.raises(ValueError):                                       # no silent slab overrun
        model._text_forward(torch.zeros(1, 8, cfg.text.dim, dtype=BF16, device="cuda"), cfg.text.max_context - 4, 0)
    with pytest.raises(ValueError):
        model._embed(torch.tensor([[cfg.text.vocab_size]]))


def test_hf_wrapper_answer_question_and_batch_answer(tiny):
    """HfMoondream (hf_moondream.py:37-183): lazy cache set-up, answer_question == query(...).strip(),
    batch_answer == per-pair greedy query, generate() with the legacy prompt form, embedding accessors."""
    g, cfg, sd, model = tiny
    from moondream_amd.hf_moondream import HfMoondream, extract_question
    from moondream_amd.moondream import IdTokenizer
    import queue

    hf = HfMoondream(cfg, sd, device="cuda", tokenizer=IdTokenizer(), max_batch=2)
    assert hf.model._kv_k is None and not hf._is_kv_cache_setup
    images = [golden_image(g, 0), golden_image(g, 1)]
    questions = ["11 12 13", "21 22 23 24 25"]
    st = {"temperature": 0, "max_tokens": 9}
    want = [model.query(im, q, settings=st)["answer"].strip() for im, q in zip(images, questions)]
    rq = queue.Queue()
    assert hf.answer_question(images[0], questions[0], result_queue=rq, settings=st) == want[0]
    assert hf._is_kv_cache_setup and rq.get_nowait() == want[0]
    assert hf.batch_answer(images, questions, max_new_tokens=9, settings={"temperature": 0}) == want
    assert len(hf.batch_answer(images, questions, max_new_tokens=9)) == 2   # the reference's default sampling settings
    enc = hf.encode_image(images[1])
    assert hf.answer_question(enc, questions[1], settings=st) == want[1]
    sampled = hf.answer_question(images[0], questions[0])          # default sampling settings, like the reference
    assert isinstance(sampled, str)
    legacy = "<image>\n\nQuestion: 11 12 13\n\nAnswer:"
    assert extract_question(legacy) == "11 12 13"
    assert isinstance(hf.generate(images[0], legacy)[0], str)
    cont = hf.generate(images[0], "5 6 7", max_new_tokens=4, settings={"temperature": 0})[0]
    assert 1 <= len(cont.split()) <= 4
    ids = torch.tensor([[1, 2, 3]])
    assert torch.equal(hf.input_embeds(ids).cpu(), hf.get_input_embeddings()(ids.cuda()).cpu())
    assert hf.caption(images[0], settings={"temperature": 0, "max_tokens": 5})["caption"] == model.caption(images[0], settings={"temperature": 0, "max_tokens": 5})["caption"]
    with pytest.raises(NotImplementedError):
        hf._unsupported_exception()


def test_single_sequence_kernel_tracks_batched_kernels(tiny):
    """Batch-1 decode on the persistent kernel (md_decode_step_b1: every decoder block of a token in one launch, grid
    barriers, fp32 matrix-vector products) against the batched kernels at one row: same ids on the wide-margin
    goldens (== the reference's), teacher-forced logits within bf16 noise, K/V rows of the new tokens equal within
    tolerance, and no barrier ever timed out."""
    g, cfg, sd, model = tiny
    try:
        for idx in range(len(g["image_index"])):
            img, prompt = golden_image(g, idx), g[f"img{idx}.cap.prompt"].tolist()
            ref = g[f"img{idx}.cap.tokens"].tolist()
            model.single_sequence_kernel = False
            a = model.batch_generate_ids([img], [prompt], max_tokens=len(ref), ignore_eos=True)[0]
            ka = model._kv_k[:, 0].clone()
            model.single_sequence_kernel = True
            b = model.batch_generate_ids([img], [prompt], max_tokens=len(ref), ignore_eos=True)[0]
            kb = model._kv_k[:, 0].clone()
            assert a == ref and b == ref, (idx, a, b, ref)
            n_pos = 730 + len(prompt) + len(ref) - 1
            compare(f"K rows img{idx}", kb[:, :, 730:n_pos], ka[:, :, 730:n_pos], 1e-2)
        assert int(model._b1_sync[64 * 11]) == 0, "a grid barrier timed out"
        # one teacher-forced step: logits of both paths
        enc = model.encode_image(golden_image(g, 0))
        outs = []
        for flag in (False, True):
            model.single_sequence_kernel = flag
            model.load_encoded_image(enc)
            first = torch.tensor([int(g["img0.cap.tokens"][0])], dtype=torch.int32, device="cuda")
            _, _, pos = model._prefill_prompts([g["img0.cap.prompt"].tolist()], enc.pos, 0)
            model._decode_greedy(first, pos, 1, cfg.tokenizer.answer_id, 0, None)
            outs.append(model._logits_buf[0].float().cpu().clone())
        err = float((outs[0] - outs[1]).abs().max())
        assert err <= 0.25, err
    finally:
        model.single_sequence_kernel = True


def fp8_decode_report(model, images, prompts, ids_bf16, ref_margins, label):
    """Run the same generation with FP8 decode weights; the first decode step's logits (identical image prefix in the
    KV cache, identical input token) must stay within tolerance of the bf16 logits, and token streams may only leave the bf16 stream at
    a decision whose reference margin is small against the measured logit error."""
    n_tok = len(ids_bf16[0])
    base = model.batch_generate_ids(images, prompts, max_tokens=n_tok, ignore_eos=True)
    assert base == [list(x) for x in ids_bf16]
    enc = model.encode_image(images[0])
    model.load_encoded_image(enc)
    logits_b, _, pos = model._prefill_prompts([prompts[0]], enc.pos, 0)
    emb = model._embed(torch.tensor([[int(ids_bf16[0][0])]]))
    step_b, _ = model._decode_one_tok(emb, None, torch.tensor([pos], dtype=torch.long), None)
    model.enable_fp8_decode(True)
    try:
        model.load_encoded_image(enc)
        _, _, pos8 = model._prefill_prompts([prompts[0]], enc.pos, 0)  # <= 64 rows: this launch streams the fp8 weights too
        step_8, _ = model._decode_one_tok(emb, None, torch.tensor([pos8], dtype=torch.long), None)
        ids_fp8 = model.batch_generate_ids(images, prompts, max_tokens=n_tok, ignore_eos=True)
    finally:
        model.enable_fp8_decode(False)
    again = model.batch_generate_ids(images, prompts, max_tokens=n_tok, ignore_eos=True)
    assert again == base  # switching the mode off restores the bf16 path exactly
    lb, l8 = step_b[0].float().cpu(), step_8[0].float().cpu()
    err = float((l8 - lb).abs().max())
    rel = float((l8 - lb).pow(2).mean().sqrt() / lb.pow(2).mean().sqrt())
    same = sum(list(a) == list(b) for a, b in zip(ids_fp8, ids_bf16))
    prefix = []
    worst_margin = 0.0
    for si, (a, b) in enumerate(zip(ids_fp8, ids_bf16)):
        k = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), len(b))
        prefix.append(k)
        if k < len(b) and ref_margins is not None:
            worst_margin = max(worst_margin, float(ref_margins[si][k]))
    print(f"fp8 decode [{label}]: first-step logits rel-RMS {rel:.4f}, max abs err {err:.3f}; {same}/{len(ids_bf16)} sequences "
          f"identical to bf16, mean matching prefix {sum(prefix) / len(prefix):.1f}/{n_tok} tokens, largest reference margin at a "
          f"first divergence {worst_margin:.3f}")
    assert all(0 <= t < model.config.text.vocab_size for seq in ids_fp8 for t in seq)
    # tolerance, not parity: 3 mantissa bits per weight give ~5 % rel-RMS on the logits of one step, and the error
    # compounds through the KV cache, so later decisions with margins of several logit units can flip (reported above)
    assert rel <= 0.12, rel
    assert sum(prefix) / len(prefix) >= 0.25 * n_tok, prefix


def fp8_full_report(model, images, prompts, ids_bf16, label, n_calib=3, ref=None, kv_cache=True):
    """The full fp8 mode (md_gemm_f8 for the ViT blocks, the projector and the prefill + the fp8 decode stream) against the
    bf16 mode on the same inputs: projected image embeddings, K rows of the image prefix and first-token logits within the
    tolerance of e4m3 operands (3 mantissa bits: a few percent per tensor, compounding over 27 + 24 blocks), token streams
    mostly unchanged; switching the mode off restores the bf16 bits."""
    n_tok = len(ids_bf16[0])
    with torch.inference_mode():
        emb_b = model._run_vision_encoder_batch(images[:2]).float().cpu()
    enc_b = model.encode_image(images[0])
    info = model.enable_fp8(images[:n_calib], prompts[0], kv_cache=kv_cache)
    label = f"{label}, e4m3 KV cache {'on' if kv_cache else 'off'}"
    try:
        assert model.w.f8_enabled() and info["kv_cache_fp8"] == kv_cache
        with torch.inference_mode():
            emb_8 = model._run_vision_encoder_batch(images[:2]).float().cpu()
        enc_8 = model.encode_image(images[0])
        ids_8 = model.batch_generate_ids(images, prompts, max_tokens=n_tok, ignore_eos=True)
        if ref is not None:  # logit error of the fp8 mode, teacher-forced on the reference's ids (the parity instrument)
            from moondream_amd import parity as P

            st8 = P.logit_error_stats(model.teacher_forced_logits(images, prompts, ref["tokens"], ref["top8_idx"]).numpy(), ref["top8_val"])
            print(f"fp8 full [{label}]: |logit error| vs the reference's top-8 logits over {st8['decisions']} decisions: max {st8['max']:.3f}, "
                  f"p99 {st8['p99']:.3f}, mean {st8['mean']:.3f} (bf16 mode: max 0.31, p99 0.19)")
            # e4m3 keeps 3 mantissa bits per operand: ~5 % noise per GEMM output whatever the scale (a floating-point format's
            # error is relative), ~10 % after 27 + 24 blocks -- logits of 10..20 move by ~1 on average
            assert st8["p99"] <= 3.0 and st8["mean"] <= 1.5, st8
    finally:
        model.enable_fp8(on=False)
    assert not model.w.f8_enabled()
    assert model.batch_generate_ids(images, prompts, max_tokens=n_tok, ignore_eos=True) == ids_bf16
    rel = lambda a, b: float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
    r_emb = rel(emb_8, emb_b)
    L = model.config.text.n_layers
    r_k0 = rel(enc_8.caches[0][0].float().cpu(), enc_b.caches[0][0].float().cpu())
    r_kl = rel(enc_8.caches[L - 1][0].float().cpu(), enc_b.caches[L - 1][0].float().cpu())
    same = sum(list(a) == list(b) for a, b in zip(ids_8, ids_bf16))
    prefix = [next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), len(b)) for a, b in zip(ids_8, ids_bf16)]
    print(f"fp8 full [{label}]: image embeddings rel-RMS {r_emb:.4f}, K rows layer 0 {r_k0:.4f} / last layer {r_kl:.4f} vs bf16; "
          f"{same}/{len(ids_bf16)} sequences identical, mean matching prefix {sum(prefix) / len(prefix):.1f}/{n_tok}; activation ranges: "
          f"vit max {max(info['vit_amax']):.1f}, text max {max(info['text_amax']):.1f}")
    assert all(0 <= t < model.config.text.vocab_size for seq in ids_8 for t in seq)
    assert torch.isfinite(emb_8).all()
    assert r_emb <= 0.15 and r_k0 <= 0.15 and r_kl <= 0.25, (r_emb, r_k0, r_kl)
    # token streams: the unfiltered 2B bench images decide most tokens at reference margins of 0 .. 0.5 logits (fixture), well
    # inside an e4m3 mode's logit error, so their streams part early; the wide-margin tiny goldens mostly survive
    assert sum(prefix) / len(prefix) >= (0.5 if ref is None else 0.1) * n_tok, prefix


def test_fp8_full_mode_tiny(tiny):
    g, cfg, sd, model = tiny
    n_img = len(g["image_index"])
    images = [golden_image(g, i) for i in range(n_img)]
    prompts = [g[f"img{i}.cap.prompt"].tolist() for i in range(n_img)]
    n = len(g["img0.cap.tokens"])
    ids = model.batch_generate_ids(images, prompts, max_tokens=n, ignore_eos=True)
    fp8_full_report(model, images, prompts, ids, "tiny", kv_cache=False)
    fp8_full_report(model, images, prompts, ids, "tiny", kv_cache=True)
    # an EncodedImage loaded into a slot gets its e4m3 copy rebuilt (load_encoded_image): same ids as the raw image
    enc = model.encode_image(images[0])
    model.enable_fp8(images, prompts[0])
    try:
        a = model.batch_generate_ids([images[0]], [prompts[0]], max_tokens=8, ignore_eos=True)
        model.fused_prefill = False  # raw image through the same two passes as the EncodedImage
        b = model.batch_generate_ids([images[0]], [prompts[0]], max_tokens=8, ignore_eos=True)
        c = model.batch_generate_ids([model.encode_image(images[0])], [prompts[0]], max_tokens=8, ignore_eos=True)
        assert b == c, (a, b, c)
    finally:
        model.fused_prefill = True
        model.enable_fp8(on=False)


def test_fp8_kv_cache_decode_steps_over_more_than_64_sequences(tiny):
    """md_text_forward splits a decode step of more than 64 sequences into blocks of 64 rows; every block must attend over
    (and write) ITS OWN slots of the e4m3 copy of the cache too: sequences 64.. of one batch equal the same images' sequences
    in the first block (same kernels, same launch shapes)."""
    g, cfg, sd, model = tiny
    n_seq = 70
    images = [golden_image(g, i % 3) for i in range(n_seq)]
    prompts = [g[f"img{i % 3}.cap.prompt"].tolist() for i in range(n_seq)]
    model.enable_fp8(images[:3], prompts[0], kv_cache=True)
    try:
        assert model._kv8_scales is not None
        out = model.batch_generate_ids(images, prompts, max_tokens=8, ignore_eos=True)
        again = model.batch_generate_ids(images, prompts, max_tokens=8, ignore_eos=True)  # slots 0..63 were not corrupted
    finally:
        model.enable_fp8(on=False)
    for i in range(n_seq):
        assert out[i] == out[i % 3], (i, out[i], out[i % 3])
    assert out == again


def test_fp8_decode_mode_tiny(tiny):
    g, cfg, sd, model = tiny
    n_img = len(g["image_index"])
    images = [golden_image(g, i) for i in range(n_img)]
    prompts = [g[f"img{i}.cap.prompt"].tolist() for i in range(n_img)]
    n = len(g["img0.cap.tokens"])
    ids = model.batch_generate_ids(images, prompts, max_tokens=n, ignore_eos=True)
    fp8_decode_report(model, images, prompts, ids, None, "tiny")


def detect13_vs_reference(model, golden_dir):
    """BASELINE configs[4]'s workload at full size (run inside the 2B test): 768 x 1024 images -> 13 crops, tiling (3, 4);
    the projected multi-crop embeddings (stitch + adaptive pool to 27 x 27 + projector: moondream.py:206-228, vision.py:77-89)
    and ``detect`` objects against the reference's (tests/golden/md2b_detect13.npz, unfiltered, margin-aware)."""
    from moondream_amd import parity as P

    g = load_golden(golden_dir, "md2b_detect13.npz")
    size, n = tuple(int(x) for x in g["size"]), int(g["n_images"])
    imgs = [synth.synthetic_image(i, int(g["seed"]), size) for i in range(n)]
    crops, tiling = model._crop(imgs[0])
    assert crops.shape[0] == 13 and tuple(tiling) == (3, 4)
    rs, cs = int(g["proj_row_stride"]), int(g["proj_col_stride"])
    with torch.inference_mode():
        emb = model._run_vision_encoder_batch(imgs[:2])
    for i in range(2):
        compare(f"2b 13-crop vis.proj img{i}", emb[i][::rs, ::cs], bits_to_bf16(g[f"img{i}.vis.proj"]), 1.5e-2)
    obj = " ".join(str(t) for t in g["object_ids"].tolist())
    st = {"max_objects": int(g["max_objects"])}
    res = model.batch_detect(imgs, [obj] * n, settings=st)
    rep = P.detect_parity([r["objects"] for r in res], g)
    print(f"detect13 at 2B: {rep['objects_compared']} objects with every decision >= {rep['margin_floor_ulps']} ulps compared exactly, "
          f"{rep['objects_mismatched']} mismatched")
    assert rep["ok"], rep
    # round 6: the region heads decide with >= 120 bf16 ulps on every object of the fixture (planted anchors): ALL objects compare
    assert rep["objects_compared"] == sum(len(g[f"img{i}.objects"]) for i in range(n)) == 32, rep
    for i in range(n):   # and they are the anchors the image's code spells (synth.region_anchor): the planted path, end to end
        bits = synth.image_code_bits(i)
        x, y = synth.region_anchor(bits, "x_first") / 1024, synth.region_anchor(bits, "y") / 1024
        o = res[i]["objects"][0]
        assert abs((o["x_min"] + o["x_max"]) / 2 - x) < 1e-6 and abs((o["y_min"] + o["y_max"]) / 2 - y) < 1e-6, (i, o, x, y)
    one = model.detect(imgs[3], obj, settings=st)
    k = P.leading_wide_objects(g["img3.margins"], 4.0)
    assert one["objects"][:k] == res[3]["objects"][:k]


def side_paths_2b_vs_reference(model, cfg, golden_dir):
    """The LoRA variant side path and query(..., reasoning=True) at the 2B shapes (tests/golden/md2b_lora.npz,
    md2b_reasoning.npz: recorded from the reference's public API, every decision's reference margin >= 0.7, i.e. above
    the measured-noise licence of the bench fixture): ids identical."""
    g = load_golden(golden_dir, "md2b_lora.npz")
    model.register_variant("synthetic", synth.synthetic_lora(cfg, seed=int(g["seed"]), rank=int(g["rank"]), device="cuda"))
    images = []
    for i in range(int(g["n_cases"])):
        img = Image.fromarray(synth.synthetic_image_array(int(g[f"case{i}.image_index"]), int(g["seed"]), (378, 378)), "RGB")
        images.append(img)
        want = g[f"case{i}.tokens"].tolist()
        got = model.caption(img, settings={"temperature": 0, "max_tokens": len(want), "variant": "synthetic"})["caption"]
        assert [int(t) for t in got.split()] == want, (i, got, want, g[f"case{i}.margins"].min())
    prompt = cfg.tokenizer.templates["caption"]["normal"]
    n = len(g["case0.tokens"])
    got = model.batch_generate_ids(images, [prompt] * len(images), max_tokens=n, variant="synthetic")
    assert got == [g[f"case{i}.tokens"].tolist() for i in range(len(images))]
    print(f"2b LoRA variant: {len(images)} captions x {n} tokens identical to the reference's (caption() and the batched engine)")
    g = load_golden(golden_dir, "md2b_reasoning.npz")
    for i in range(int(g["n_cases"])):
        img = Image.fromarray(synth.synthetic_image_array(int(g[f"case{i}.image_index"]), int(g["seed"]), (378, 378)), "RGB")
        res = model.query(img, "11 12 13", reasoning=True, settings={"temperature": 0, "max_tokens": int(g["max_tokens"])})
        assert [int(t) for t in res["reasoning"]["text"].split()] == g[f"case{i}.reasoning_tokens"].tolist()
        assert [int(t) for t in res["answer"].split()] == g[f"case{i}.answer_tokens"].tolist()
        assert len(res["reasoning"]["grounding"]) == int(g[f"case{i}.n_grounding"])
    print(f"2b reasoning query: {int(g['n_cases'])} cases identical to the reference's")


def batch_equals_sequential_unfiltered(model, imgs64, prompt, got64_default, ref_ids, ref_margins, thr):
    """test_batch_equals_sequential_unfiltered (run inside the 2B test: one 2B model per session).  The 64 UNFILTERED bench
    images, Moondream-2B, 32 greedy tokens.
      strict mode (set_strict_batch_invariance): B=64 batch == 64 sequential B=1 calls == caption() BIT FOR BIT;
      default mode: B=64 batch == 64 sequential B=1 calls on the batched kernels BIT FOR BIT (round 4: the four-wave GEMM is
      pinned for every launch of more than 64 rows); against the B=1 LATENCY path (small tiles + persistent kernel): counted, and wherever batch and sequential part
      while both still follow the reference's stream, the reference margin of that decision must be inside the
      measured noise threshold."""
    n = len(imgs64)
    try:
        model.set_strict_batch_invariance(True)
        strict_batch = model.batch_generate_ids(imgs64, [prompt] * n, max_tokens=32, ignore_eos=True)
        strict_seq = [model.batch_generate_ids([im], [prompt], max_tokens=32, ignore_eos=True)[0] for im in imgs64]
        assert strict_batch == strict_seq, [i for i in range(n) if strict_batch[i] != strict_seq[i]]
        cap = model.caption(imgs64[3], settings={"temperature": 0, "max_tokens": 32})["caption"]
        assert [int(t) for t in cap.split()] == strict_seq[3][: len(cap.split())]
    finally:
        model.set_strict_batch_invariance(False)
    # DEFAULT mode, the lone sequence on the batched kernels: the benchmarked batch path == sequential, bit for bit
    try:
        model.single_sequence_kernel = False
        seq_batched = [model.batch_generate_ids([im], [prompt], max_tokens=32, ignore_eos=True)[0] for im in imgs64]
    finally:
        model.single_sequence_kernel = True
    assert got64_default == seq_batched, [i for i in range(n) if got64_default[i] != seq_batched[i]]
    print("batch vs sequential, DEFAULT mode: batch(B=64) == 64 x batch_generate_ids([x]) on the batched kernels, 64/64 bit-identical")
    seq_default = [model.batch_generate_ids([im], [prompt], max_tokens=32, ignore_eos=True)[0] for im in imgs64]
    same_ds = sum(a == b for a, b in zip(got64_default, seq_default))
    same_strict = sum(a == b for a, b in zip(got64_default, strict_batch))
    worst = 0.0
    for i in range(n):
        a, b, r = got64_default[i], seq_default[i], ref_ids[i]
        j = next((t for t in range(32) if a[t] != b[t]), None)
        if j is not None and a[:j] == r[:j]:  # both on the reference's stream up to j: its margin describes this decision
            worst = max(worst, float(ref_margins[i][j]))
    print(f"batch vs sequential, unfiltered 64 images at 2B: strict mode 64/64 bit-identical; default mode batch(B=64, fused prefill) vs "
          f"B=1 (fused prefill + persistent kernel): {same_ds}/64 identical, batch(default) vs batch(strict): {same_strict}/64; largest "
          f"reference margin where default batch and default B=1 part on the reference's stream: {worst:.4f} (threshold {thr:.4f})")
    assert worst <= thr, (worst, thr)
    assert same_ds >= 32 and same_strict >= 32  # quantified above; the hard claims are the strict-mode equality and the margin bound


# End-of-stack activation tolerance at full size (27 ViT blocks / 24 decoder blocks, sampled): relative RMS against the reference.
# The reference itself is 9.6e-3 from an fp64 evaluation at that depth and so is this path (test_vit_error_against_fp64_truth...);
# rounds 1-5 allowed 1.5e-2, the round-5 review asked for 1.25e-2 -- the per-layer profile on the tiny model
# (test_per_layer_drift_profile_...) is what catches a single bad layer.
ACT_TOL = 1.25e-2


def mutation_sensitivity(model, cfg, g, images, imgs64, pr, gb):
    """The parity gates must be able to FAIL.  Each mutation plants ONE local fault of the kind a kernel bug produces into the
    packed weights of the 2B model -- (a) one attention head of one ViT block lost (its 72 input columns of the block's proj
    zeroed), (b) two 64-wide K slices of one decoder fc2 swapped (a mis-addressed operand slice) -- and the gates of
    test_full_size_models_vs_reference are evaluated again on the timed configuration: the ids report of the 64 bench images
    (measured licence, logit-error caps, exact-count floor) and the activation compares against the reference (ViT output,
    KV rows of the first / last decoder block).  Every mutation must trip at least one gate; which ones is printed."""
    from moondream_amd import parity as P
    from oracle import moondream_oracle as O

    ref_ids = gb["tokens"].tolist()
    arr = np.array(images[0])
    ts, fs, rs = int(g["vit_token_stride"]), int(g["vit_feat_stride"]), int(g["kv_row_stride"])

    def rel_rms(a, b):
        a, b = a.detach().float().cpu(), b.detach().float().cpu()
        return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())

    def gates():
        topk = model.teacher_forced_logits(imgs64, [pr] * 64, gb["tokens"], gb["top8_idx"]).numpy()
        got = model.batch_generate_ids(imgs64, [pr] * 64, max_tokens=32, ignore_eos=True)
        rep = P.parity_report(got, ref_ids, gb["margins"], topk, gb["top8_val"], tokens=32, min_exact=40, ref_topk_idx=gb["top8_idx"])
        feats = model._vis_enc(O.normalize_crops(np.stack([arr, arr])).cuda())
        enc = model.encode_image(images[0])
        act = {"vit.out": rel_rms(feats[:, ::ts, ::fs], bits_to_bf16(g["img0.vit.out"]))}
        for li in (0, cfg.text.n_layers - 1):
            act[f"k{li}"] = rel_rms(enc.caches[li][0][0, :, ::rs], bits_to_bf16(g[f"img0.cap.k{li}"]))
        failed = [k for k, v in act.items() if v > ACT_TOL] + ([] if rep["parity_ok"] else ["ids report"])
        return failed, rep, act

    failed, rep, act = gates()
    assert not failed, (failed, rep["parity_note"], act)
    vit_proj = model.w._vit_packed[13]["proj"].w     # [n_pad][k_pad] bf16: input feature f of the layer = column f
    txt_fc2 = model.w._text_packed[11]["fc2"].w
    mutations = [
        ("ViT block 13: head 5 lost (72 proj input columns zeroed)", vit_proj, lambda w: w[:, 5 * 72 : 6 * 72].zero_()),
        ("decoder block 11: K slices [0, 64) and [64, 128) of fc2 swapped", txt_fc2,
         lambda w: w[:, :128].copy_(torch.cat([w[:, 64:128], w[:, :64]], dim=1))),
    ]
    for what, w, mutate in mutations:
        keep = w.clone()
        try:
            mutate(w)
            torch.cuda.synchronize()
            failed, rep, act = gates()
        finally:
            w.copy_(keep)
            torch.cuda.synchronize()
        print(f"mutation [{what}]: gates tripped {failed}; ids {rep['parity_exact']}/64 identical, max |logit err| "
              f"{rep['parity_max_logit_err']:.3f} (p99 {rep['parity_p99_logit_err_ulps']:.1f} ulps), wide-margin teacher-forced decisions that differ "
              f"{rep['parity_tf_decisions_violations']}, activations "
              + ", ".join(f"{k} {v:.3e}" for k, v in act.items()))
        assert failed, f"no parity gate noticed: {what}"
    failed, rep, act = gates()  # restored
    assert not failed, (failed, rep["parity_note"], act)
    # A SYSTEMATIC fault -- the kind a mis-indexed head in the attention kernel or a wrong V stride in the cache would be -- DOES
    # move the ids of the conditioned checkpoint (a local one does not, by construction: DESIGN section 2): in EVERY decoder block
    # the proj input columns of head (l mod 32) and of the next head are swapped.
    t = cfg.text
    hd = t.dim // t.n_heads
    keeps = []
    try:
        for l in range(t.n_layers):
            w = model.w._text_packed[l]["proj"].w
            keeps.append(w.clone())
            h0, h1 = (l % t.n_heads) * hd, ((l + 1) % t.n_heads) * hd
            a, b = w[:, h0 : h0 + hd].clone(), w[:, h1 : h1 + hd].clone()
            w[:, h0 : h0 + hd], w[:, h1 : h1 + hd] = b, a
        torch.cuda.synchronize()
        failed, rep, act = gates()
    finally:
        for l, k in enumerate(keeps):
            model.w._text_packed[l]["proj"].w.copy_(k)
        torch.cuda.synchronize()
    print(f"mutation [every decoder block: two heads' proj input columns swapped]: gates tripped {failed}; ids {rep['parity_exact']}/64 identical, "
          f"max |logit err| {rep['parity_max_logit_err']:.3f}, wide-margin teacher-forced decisions that differ {rep['parity_tf_decisions_violations']}")
    assert "ids report" in failed and rep["parity_exact"] < 32 and rep["parity_tf_decisions_violations"] > 100, (failed, rep["parity_exact"])
    failed, rep, act = gates()  # restored
    assert not failed, (failed, rep["parity_note"], act)


def test_int4_checkpoint_streams_its_own_weights_in_decode():
    """A checkpoint whose decoder blocks are the reference's QuantizedLinear triples (layers.py:47-109; 0.5B shapes, 4-bit
    groups of 128): the decode regime streams the NIBBLES (md_linear_fp8.format = MD_WSTREAM_INT4_G128) and rebuilds the bf16
    weights of the dequantised copy in registers.  Same weights, another K order: against the bf16 stream of the very same
    model (enable_int4_decode(False)) the first decode step's K rows agree within accumulation-order tolerance, the ids of
    every sequence agree up to its first narrow decision (and most of them entirely), a lone sequence and a batch agree, and
    the stream (opt-in: enable_int4_decode) survives enable_fp8_decode(True / False)."""
    from moondream_amd.moondream import MoondreamModel, IdTokenizer
    from util import quantize_int4

    cfg = get_config("0.5b")
    sd = synth.synthetic_state_dict(cfg, seed=1, device="cuda")
    qsd = dict(sd)
    for i in range(cfg.text.n_layers):
        for n in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"):
            p = f"text.blocks.{i}.{n}"
            packed, scale, zero = quantize_int4(qsd.pop(p + ".weight"), zero_shift=0.25)
            qsd[p + ".weight.packed"], qsd[p + ".weight.scale"], qsd[p + ".weight.zero_point"] = packed, scale, zero
    model = MoondreamModel(cfg, qsd, device="cuda", tokenizer=IdTokenizer(), max_batch=4)
    assert model.w.has_int4_source() and not model.int4_decode and not bool(model.w.text.fp8)  # opt-in
    model.enable_int4_decode(True)
    assert model.int4_decode and bool(model.w.text.fp8) and model.w._fp8_is_int4
    images = [synth.synthetic_image(i, 1) for i in range(4)]
    prompt = cfg.tokenizer.templates["caption"]["normal"]
    n_tok = 16

    def run():
        ids = model.batch_generate_ids(images, [prompt] * 4, max_tokens=n_tok, ignore_eos=True)
        p0 = 730 + len(prompt)
        return ids, model._kv_k[:, :4, :, p0 : p0 + 2].clone()

    ids4, k4 = run()
    lone = model.batch_generate_ids(images[:1], [prompt], max_tokens=n_tok, ignore_eos=True)[0]  # batched kernels (no persistent kernel with a stream attached)
    assert model._b1_used is False and lone == ids4[0]
    model.enable_fp8_decode(True)   # the e4m3 copy replaces the int4 stream ...
    assert not model.w._fp8_is_int4
    model.enable_fp8_decode(False)  # ... and the checkpoint's own stream comes back
    assert model.w._fp8_is_int4
    model.enable_int4_decode(False)
    assert not bool(model.w.text.fp8)
    model.single_sequence_kernel = False
    ids16, k16 = run()
    compare("int4 stream vs bf16 stream: K rows of the first decode steps (all layers)", k4, k16, 1e-2)
    same = sum(a == b for a, b in zip(ids4, ids16))
    prefix = [next((t for t in range(n_tok) if a[t] != b[t]), n_tok) for a, b in zip(ids4, ids16)]
    print(f"int4 weight stream vs bf16 stream of the same dequantised weights (0.5B shapes): {same}/4 sequences identical over {n_tok} tokens, "
          f"matching prefixes {prefix}")
    assert same >= 2 and min(prefix) >= 1
    model.enable_int4_decode(True)
    assert run()[0] == ids4  # deterministic, and re-attachable


@pytest.mark.parametrize("name,cfg_name", [("md05b_seed1.npz", "0.5b"), ("md2b_seed1.npz", "2b")])
def test_full_size_models_vs_reference(golden_dir, name, cfg_name):
    """BASELINE.json configs at full size: greedy ids bit-exact against the
    reference on wide-margin images, activations/logits within tolerance."""
    g = load_golden(golden_dir, name)
    cfg, sd, model = build(cfg_name, int(g["seed"]), max_batch=4)
    n_img = len(g["image_index"])
    images = [golden_image(g, i) for i in range(n_img)]
    prompts = [g[f"img{i}.cap.prompt"].tolist() for i in range(n_img)]
    n = len(g["img0.cap.tokens"])
    got = model.batch_generate_ids(images, prompts, max_tokens=n)
    for i in range(n_img):
        ref = g[f"img{i}.cap.tokens"].tolist()
        assert got[i] == ref, f"{cfg_name} img{i}: min margin {g[f'img{i}.cap.margins'].min():.3f}\n{got[i]}\n{ref}"
    # stage activations (sampled)
    from oracle import moondream_oracle as O

    arr = np.array(images[0])
    feats = model._vis_enc(O.normalize_crops(np.stack([arr, arr])).cuda())
    ts, fs = int(g["vit_token_stride"]), int(g["vit_feat_stride"])
    compare(f"{cfg_name} vit.out", feats[:, ::ts, ::fs], bits_to_bf16(g["img0.vit.out"]), ACT_TOL)
    enc = model.encode_image(images[0])
    rs = int(g["kv_row_stride"])
    for li in (0, cfg.text.n_layers - 1):
        compare(f"{cfg_name} k{li}", enc.caches[li][0][0, :, ::rs], bits_to_bf16(g[f"img0.cap.k{li}"]), ACT_TOL)
    # top-8 logits of the prompt prefill
    model.load_encoded_image(enc)
    logits, _, _ = model._prefill_prompts([prompts[0]], enc.pos, 0)
    idx = torch.from_numpy(g["img0.cap.top8_idx"][0]).long()
    ref_top = torch.from_numpy(g["img0.cap.top8_val"][0])
    err = (logits[0].float().cpu()[idx] - ref_top).abs().max()
    assert float(err) <= 0.5, float(err)  # measured cap on every box so far: 0.3125 (the licence of moondream_amd/parity.py is capped at the same 0.5)
    # margin-aware VQA
    refq = g["img0.vqa.tokens"].tolist()
    gotq = model.batch_generate_ids([images[0]], [g["img0.vqa.prompt"].tolist()], max_tokens=len(refq))[0]
    for i, (a, b) in enumerate(zip(gotq, refq)):
        if a != b:
            assert g["img0.vqa.margins"][i] <= 0.5, (i, a, b, g["img0.vqa.margins"][i])
            break
    if cfg_name != "2b":
        return
    # THE TIMED CONFIGURATION (bench.py / BASELINE configs[2]): the 64 seed-1 images, B=64, 32 tokens,
    # against the reference's ids for exactly these images (unfiltered, margin-aware)
    gb = load_golden(golden_dir, "md2b_bench64.npz")
    imgs64 = [synth.synthetic_image(i, int(gb["seed"])) for i in range(gb["tokens"].shape[0])]
    pr = gb["prompt"].tolist()
    from moondream_amd import parity as P

    ref_ids = gb["tokens"].tolist()
    # the licence for a divergence is MEASURED: the HIP path teacher-forced on the reference's ids, its logits at the
    # reference's top-8 ids of all 64 x 33 decisions against the reference's (moondream_amd/parity.py)
    topk = model.teacher_forced_logits(imgs64, [pr] * 64, gb["tokens"], gb["top8_idx"]).numpy()
    # ... and CALIBRATED: the reference's own ATen calls through torch-ROCm on the same images (SURVEY 8c's second oracle) say how
    # many sequences an equally correct evaluation keeps when only the BLAS backend changes; the floor comes from that count
    import bench

    second = bench.second_oracle(cfg, sd, int(gb["seed"]), 32, "cuda")
    floor = bench.exact_floor(64, 32, second)
    print(f"second oracle (reference ATen calls on this GPU): {second['exact']}/64 identical, max |logit err| {second['max_logit_err']:.4f} "
          f"(p99 {second['p99_logit_err']:.4f}), largest margin at a first divergence {second['max_divergence_margin']:.4f} -> floor {floor}")
    assert second["max_logit_err"] <= 0.5 and second["exact"] >= bench.SECOND_ORACLE_SANITY, second
    for pipelined in (False, True):
        if pipelined:
            model.compile()
            outs = list(model.batch_generate_ids_pipelined([(imgs64, [pr] * 64)] * 2, max_tokens=32, ignore_eos=True))
            model.use_graphs = False
            assert outs[0] == outs[1]
            got64 = outs[1]
        else:
            got64 = model.batch_generate_ids(imgs64, [pr] * 64, max_tokens=32, ignore_eos=True)
        rep = P.parity_report(got64, ref_ids, gb["margins"], topk, gb["top8_val"], tokens=32, min_exact=floor, ref_topk_idx=gb["top8_idx"])
        print(f"bench64 parity ({'pipelined+graphs' if pipelined else 'eager'}): {rep['parity_exact']}/64 identical; max |logit err| "
              f"{rep['parity_max_logit_err']:.4f} ({rep['parity_max_logit_err_ulps']:.1f} bf16 ulps, p99 {rep['parity_p99_logit_err']:.4f}) over "
              f"{rep['parity_decisions']} decisions -> threshold {rep['parity_threshold']:.4f}; largest reference margin at a first "
              f"divergence {rep['parity_max_divergence_margin']:.4f}; teacher-forced: {rep['parity_tf_decisions_must_match']} decisions "
              f"with a reference margin above the licence, {rep['parity_tf_decisions_violations']} of them differ "
              f"({rep['parity_tf_decisions_agree']} of {rep['parity_decisions']} decisions agree in all)")
        assert rep["parity_ok"], rep["parity_note"]
        # round 6: the fixture is well-conditioned (every reference margin >= 8, tests/test_parity_cpu.py) -- ALL 64 sequences and
        # ALL 64 x 33 teacher-forced decisions must equal the reference's; the licence machinery above has nothing left to license
        assert rep["parity_exact"] == 64 and rep["parity_must_match"] == 64, (rep["parity_exact"], rep["parity_must_match"])
        assert rep["parity_tf_decisions_must_match"] == 64 * 33 and rep["parity_tf_decisions_violations"] == 0
        for i in g["image_index"].tolist():  # the wide-margin images of md2b_seed1 are among the 64: exact
            assert got64[i] == gb["tokens"][i].tolist(), i
    vqa64_vs_reference(model, cfg, sd, imgs64, golden_dir)
    batch_equals_sequential_unfiltered(model, imgs64, pr, got64, ref_ids, gb["margins"], rep["parity_threshold"])
    mutation_sensitivity(model, cfg, g, images, imgs64, pr, gb)
    detect13_vs_reference(model, golden_dir)
    side_paths_2b_vs_reference(model, cfg, golden_dir)
    # opt-in FP8 weight stream for the decode steps of the same configuration (BASELINE configs[4]): a different
    # numerical mode, judged by tolerance against the bf16 path -- never by bit parity
    fp8_decode_report(model, imgs64, [pr] * 64, got64, gb["margins"], "2b B=64")
    fp8_full_report(model, imgs64, [pr] * 64, got64, "2b B=64", n_calib=8, ref=gb, kv_cache=False)
    fp8_full_report(model, imgs64, [pr] * 64, got64, "2b B=64", n_calib=8, ref=gb, kv_cache=True)
