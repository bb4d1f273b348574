#!/usr/bin/env python3
"""Headline benchmark: images/sec of Moondream-2B bf16 `batch_generate`
(captioning, batch 64 per GPU, synthetic 378x378 images, 5-token caption prompt,
32 greedy decode tokens) -- BASELINE.json configs[2] at N=1, configs[3] shape
(64 images per GPU, data parallel) at N>1.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full pass of the hot path over one batch per rank: host tiling,
H2D of uint8 crops, ViT (2 crops/image, both computed), projector, 730-token
image prefill, prompt prefill, 32 lockstep decode steps, D2H of token ids (and,
for N>1, the RCCL gather of ids on rank 0).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np
import torch

FLOP_VIT_PER_CROP = 666.45e9  # SURVEY.md section 8d / BASELINE.md section 4 (2B)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--tokens", type=int, default=32, help="decode tokens per image")
    ap.add_argument("--model", default="2b")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--vit-chunk", type=int, default=128, help="crops per ViT launch group")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency-runs", type=int, default=5)
    ap.add_argument("--no-graphs", action="store_true", help="launch decode steps eagerly instead of hipGraph replay")
    ap.add_argument("--no-pipeline", action="store_true", help="run each step's encode and decode back to back on one stream")
    ap.add_argument("--prompt", choices=["caption", "vqa32"], default="caption",
                    help="caption: the 5-id caption template; vqa32: 32-id seeded question prompts (SURVEY 8d)")
    ap.add_argument("--no-vqa-leg", action="store_true", help="skip the auxiliary 32-token-prompt measurement")
    ap.add_argument("--selftest-dist", action="store_true",
                    help="only exercise the launch + collective plumbing (gloo on a CPU-only box) and print a JSON line")
    return ap.parse_args()


def selftest_dist(args):
    """The N>1 plumbing without a model: rendezvous, flat broadcast, id gather, barrier, max-over-ranks.
    Runs over gloo where there is no GPU (tests/test_dist_cpu.py drives `bench.py --gpus 2 --selftest-dist`)."""
    from moondream_amd import dist as mdist

    rank, world, local = mdist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local) if torch.cuda.is_available() else torch.device("cpu")
    full = {"w": torch.arange(24, dtype=torch.float32).reshape(4, 6).to(torch.bfloat16)} if rank == 0 else None
    obj = [mdist.state_dict_template(full) if rank == 0 else None]
    if world > 1:
        torch.distributed.broadcast_object_list(obj, src=0)
    sd = mdist.broadcast_state_dict(full, obj[0], dev)
    assert sd["w"].float().sum().item() == 276.0
    mine = mdist.shard_range(3 * world + 1, rank, world)
    ids = torch.tensor([[i, i + 1] for i in mine], dtype=torch.int32, device=dev).reshape(len(mine), 2)
    blocks = mdist.gather_token_ids(ids)
    mdist.barrier()
    t = mdist.max_over_ranks(float(rank), dev)
    if rank == 0:
        got = torch.cat([b.cpu() for b in blocks], 0)[:, 0].tolist()
        assert got == list(range(3 * world + 1)), got
        print(json.dumps({"selftest": "dist", "n_gpus": world, "max_rank": t, "items": len(got)}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def cpu_baseline(cfg, sd, seed, T, budget_s=20.0):
    """The oracle (CPU port of the reference's algorithm) on the host cores, B=1,
    on a STRICTLY bounded sample: single layers are timed (one ViT block on one
    crop, one decoder block over the 730-token prefill, one decoder block for a
    decode step) and scaled by the layer counts; anything the budget does not
    allow is extrapolated from the measured FLOP rate.  Returns images/s."""
    from moondream_amd import synth
    from oracle import moondream_oracle as O

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    torch.set_num_threads(threads)
    v, t = cfg.vision, cfg.text
    names = [k for k in sd if k.startswith("vision.blocks.0.") or k.startswith("text.blocks.0.")]
    w = {k: sd[k].cpu() for k in names}
    O.cache_fp32_weights(w)
    t_begin = time.perf_counter()
    g = torch.Generator().manual_seed(seed)

    def vit_block(x):
        p = "vision.blocks.0"
        a = O.vit_attention(O.layer_norm(x, w[p + ".ln1.weight"], w[p + ".ln1.bias"]), w, p + ".attn", v.enc_n_heads)
        x = (x.float() + a.float()).to(torch.bfloat16)
        m = O.mlp(O.layer_norm(x, w[p + ".ln2.weight"], w[p + ".ln2.bias"]), w, p + ".mlp")
        return (x.float() + m.float()).to(torch.bfloat16)

    x = torch.randn(1, v.n_patches, v.enc_dim, generator=g).to(torch.bfloat16)
    t0 = time.perf_counter()
    vit_block(x)
    t_vlayer = time.perf_counter() - t0
    D, Dv, FFv, Tn = t.dim, v.enc_dim, v.enc_ff_dim, v.n_patches
    flop_vlayer = 2 * Tn * Dv * 3 * Dv + 2 * Tn * Dv * Dv + 4 * v.enc_n_heads * Tn * Tn * v.head_dim + 4 * Tn * Dv * FFv
    rate = flop_vlayer / t_vlayer
    t_vit = 2 * v.enc_n_layers * t_vlayer
    note = [f"1 ViT block on 1 crop {t_vlayer:.2f}s ({rate/1e9:.0f} GFLOP/s) x {v.enc_n_layers} blocks x 2 crops"]

    one = O.Oracle.__new__(O.Oracle)
    one.cfg, one.sd, one.fast = cfg, w, False
    one.cos, one.sin = O.rope_table(t.rot_dim // 2, t.max_context)
    cfg1 = type(cfg)(text=type(t)(**{**t.__dict__, "n_layers": 1}), vision=v, region=cfg.region, tokenizer=cfg.tokenizer)
    kv = O.OracleKV.empty(cfg1)
    P = t.prefix_attn
    flop_player = 2 * P * D * t.qkv_dim + 2 * P * D * D + 4 * t.n_heads * P * P * t.head_dim + 4 * P * D * t.ff_dim
    if (time.perf_counter() - t_begin) + flop_player / rate < budget_s:
        xp = torch.randn(P, D, generator=g).to(torch.bfloat16)
        t0 = time.perf_counter()
        O.text_decoder(xp, w, cfg1, kv, torch.arange(P), one.cos, one.sin)
        t_player = time.perf_counter() - t0
        note.append(f"1 decoder block over the {P}-token prefill {t_player:.2f}s x {t.n_layers}")
    else:
        t_player = flop_player / rate
        note.append(f"prefill block extrapolated at the ViT FLOP rate to {t_player:.2f}s x {t.n_layers}")
    flop_proj = 2 * Tn * (2 * Dv * v.proj_inner_dim + v.proj_inner_dim * v.proj_out_dim)
    t_prefill = t.n_layers * t_player + flop_proj / rate  # + projector at the measured rate
    xd = torch.randn(1, D, generator=g).to(torch.bfloat16)
    reps = []
    for i in range(3):
        t0 = time.perf_counter()
        O.text_decoder(xd, w, cfg1, kv, torch.tensor([P + i]), one.cos, one.sin)
        reps.append(time.perf_counter() - t0)
    t_dlayer = float(np.median(reps))
    per_tok = t.n_layers * t_dlayer + 2 * D * t.vocab_size / rate
    note.append(f"1 decoder block per decode step {t_dlayer*1e3:.1f}ms x {t.n_layers}")
    total = t_vit + t_prefill + per_tok * (T + 1)
    return 1.0 / total, threads, "; ".join(note) + f"; images/s = 1/(ViT {t_vit:.1f}s + prefill {t_prefill:.1f}s + {T + 1} x {per_tok:.2f}s)"


def main():
    args = parse()
    from moondream_amd import dist as mdist

    # `python bench.py --gpus N` with no launcher: become N ranks under torch.distributed.run
    rc = mdist.relaunch_under_torchrun(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if rc is not None:
        sys.exit(rc)
    if args.selftest_dist:
        return selftest_dist(args)
    from moondream_amd import _lib, synth
    from moondream_amd.config import get_config
    from moondream_amd.moondream import MoondreamModel, IdTokenizer

    rank, world, local = mdist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    cfg = get_config(args.model)

    # weights: rank 0 generates, RCCL broadcasts one flat buffer
    sd0 = synth.synthetic_state_dict(cfg, seed=args.seed, device=dev) if rank == 0 else None
    if world > 1:
        template = None
        if rank == 0:
            template = mdist.state_dict_template(sd0)
        obj = [template]
        torch.distributed.broadcast_object_list(obj, src=0)
        sd = mdist.broadcast_state_dict(sd0, obj[0], dev, src=0)
    else:
        sd = sd0
    model = MoondreamModel(cfg, sd, device=dev, tokenizer=IdTokenizer(), max_batch=args.batch, vit_chunk_crops=args.vit_chunk)
    lib = model.lib
    if not args.no_graphs:
        model.compile()  # hipGraph replay of the device-resident decode steps

    B, T = args.batch, args.tokens
    n_total = B * world
    mine = mdist.shard_range(n_total, rank, world)
    images = [synth.synthetic_image(i, args.seed) for i in mine]
    caption_prompt = cfg.tokenizer.templates["caption"]["normal"]
    vqa_prompts = [synth.synthetic_vqa_prompt(cfg, i, args.seed) for i in mine]
    if args.prompt == "caption":
        prompt, prompts = caption_prompt, [caption_prompt] * len(images)
    else:
        prompt, prompts = vqa_prompts[0], vqa_prompts

    def finish(ids):
        local_ids = torch.tensor(ids, dtype=torch.int32, device=dev)
        return mdist.gather_token_ids(local_ids)

    def run_steps(k, prompts=prompts):
        """k steps = k full passes over this rank's batch.  Pipelined mode overlaps the
        decode of step i with the encode of step i+1 (two HIP streams); every step's
        work, including its gather, completes inside the call."""
        if args.no_pipeline:
            outs = [finish(model.batch_generate_ids(images, prompts, max_tokens=T, ignore_eos=True)) for _ in range(k)]
        else:
            gen = model.batch_generate_ids_pipelined(((images, prompts) for _ in range(k)), max_tokens=T, ignore_eos=True)
            outs = [finish(ids) for ids in gen]
        return outs

    if args.warmup:
        # pipelined mode alternates two KV slot groups: warm both (graph capture) before timing
        run_steps(args.warmup if args.no_pipeline else max(2, args.warmup))
    mdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run_steps(args.steps)
    torch.cuda.synchronize()
    mdist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = mdist.max_over_ranks(elapsed, dev)

    # Dominant kernel (bf16 MFMA tile GEMM): algorithmic flops / HIP-event time of its launches,
    # taken on ONE extra, non-overlapped, eagerly launched step right after the timed region:
    # while two streams interleave (pipelined mode) an event bracket also contains the time a
    # launch spends queued behind the other stream's kernels, and graph-replayed launches carry
    # no events at all.  The same step yields the per-phase GPU times.
    graphs_were = model.use_graphs
    model.use_graphs = False
    model.collect_timing = True
    lib.md_profile_gemm(1)
    model.batch_generate_ids(images, prompts, max_tokens=T, ignore_eos=True)
    torch.cuda.synchronize()
    model.collect_timing = False
    model.use_graphs = graphs_were
    phase_ms = {k: round(v, 2) for k, v in model.last_phase_ms.items()}
    step_gpu_s = sum(model.last_phase_ms.values()) * 1e-3
    f, ms, n = C.c_double(), C.c_double(), C.c_int64()
    _lib.check(lib.md_profile_gemm_read(0, C.byref(f), C.byref(ms), C.byref(n)))
    gemm_tflops = f.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
    by, ms1, n1 = C.c_double(), C.c_double(), C.c_int64()
    _lib.check(lib.md_profile_gemm_read(1, C.byref(by), C.byref(ms1), C.byref(n1)))
    lib.md_profile_gemm(0)
    stream_gbs = by.value / (ms1.value * 1e-3) / 1e9 if ms1.value > 0 else 0.0

    if rank != 0:
        return
    # HBM-side bytes per tile-GEMM launch from the committed PMC passes over this command's eager step
    # (tools/gpu_pmc_traffic.sh: separate FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled per the gfx950
    # note in MI355X_MICROARCH.md; WRITE_SIZE uncalibrated).  bench.py cannot collect counters on itself.
    traffic, traffic_note = None, "no profiles/r01_pmc_traffic.json"
    try:
        with open(os.path.join(REPO, "profiles", "r01_pmc_traffic.json")) as f:
            tg = json.load(f)["tile_gemm"]
        traffic = (2.0 * tg["FETCH_SIZE_kb_sum"] + tg["WRITE_SIZE_kb_sum"]) * 1024.0 / tg["launches"]
        traffic_note = ("bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) / launches from profiles/r01_pmc_traffic.json "
                        "(rocprofv3 --pmc passes over one eager bench step, B=64)")
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        pass
    result = {
        "metric": "images_per_sec",
        "value": n_total * args.steps / elapsed,
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": f"Moondream-{args.model.upper()} bf16 batch_generate ({args.prompt}): {B} images/GPU x 378x378 "
                        f"(2 crops each, both encoded), {len(prompt)}-token prompt, {T} greedy decode tokens, "
                        f"seeded synthetic weights",
            "batch_per_gpu": B, "decode_tokens": T, "parallelism": f"dp{world}",
            "step_overlap": "none" if args.no_pipeline else "decode(step i) || encode(step i+1) on two HIP streams",
        },
        "roofline": {
            "bound": "mfma",
            "kernel": "gemm_bf16_kernel<256,256,...> tile GEMM, alternating wave groups (v_mfma_f32_32x32x16_bf16)",
            "achieved": gemm_tflops,
            "peak": 2500.0,
            "unit": "TFLOP/s",
            "frac": gemm_tflops / 2500.0,
            "traffic": traffic,
            "traffic_note": traffic_note,
            "launches": int(n.value),
            "share_of_step": (ms.value * 1e-3) / step_gpu_s if step_gpu_s > 0 else None,
            "measured_on": "one non-overlapped, eagerly launched step after the timed region",
        },
        "decode_gemm": {
            "bound": "hbm", "kernel": "gemm_bf16_kernel / gemm_pair_kernel <64,64> decode-regime configs (m <= 64 weight stream; proj+fc2 as K-slice partials)",
            "achieved": stream_gbs if n1.value else None, "peak": 8000.0, "unit": "GB/s",
            "frac": stream_gbs / 8000.0 if n1.value else None, "launches": int(n1.value),
            "share_of_step": (ms1.value * 1e-3) / step_gpu_s if step_gpu_s > 0 and n1.value else None,
        },
        "phase_ms": phase_ms,
    }
    if args.model == "2b" and phase_ms.get("vision"):
        # the north star's "ViT encoder vs MFMA roofline": whole vision phase (patchify, 27 blocks incl.
        # attention and layer norms, projector) against the algorithmic FLOPs of SURVEY 8d
        vit_flops = len(images) * (2 * FLOP_VIT_PER_CROP + 51.98e9)
        result["vit_encoder"] = {
            "bound": "mfma", "achieved": vit_flops / (phase_ms["vision"] * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
            "frac": vit_flops / (phase_ms["vision"] * 1e-3) / 1e12 / 2500.0,
            "note": "all kernels of the vision phase, 2 crops/image (666.45 GFLOP each) + projector (51.98 GFLOP/image)",
        }

    # p50 single-image caption latency (B=1), outside the timed region
    lat = []
    one = [images[0]]
    for i in range(args.latency_runs + 1):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        model.batch_generate_ids(one, [prompt], max_tokens=T, ignore_eos=True)
        torch.cuda.synchronize()
        if i > 0:
            lat.append(time.perf_counter() - t1)
    if lat:
        result["p50_caption_latency_ms"] = float(np.median(lat) * 1e3)

    # auxiliary leg: the same batch with 32-id question prompts (north star "32-token prompts",
    # BASELINE configs[1] single-image VQA latency); outside the timed region of `value`
    if world == 1 and not args.no_vqa_leg and args.prompt == "caption":
        run_steps(2, vqa_prompts)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(2, vqa_prompts)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / 2
        lat = []
        for i in range(args.latency_runs + 1):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            model.batch_generate_ids(one, [vqa_prompts[0]], max_tokens=T, ignore_eos=True)
            torch.cuda.synchronize()
            if i > 0:
                lat.append(time.perf_counter() - t1)
        result["vqa32"] = {
            "images_per_sec": B / dt, "ms_per_step": dt * 1e3, "prompt_tokens": len(vqa_prompts[0]),
            "p50_latency_ms": float(np.median(lat) * 1e3) if lat else None,
        }

    if world == 1 and not args.no_cpu_baseline:
        est, cores, note = cpu_baseline(cfg, sd, args.seed, T)
        result["cpu_baseline"] = {
            "value": est, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle (CPU port of the reference algorithm, fp32 contractions), B=1, bounded sample: {note}",
        }
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
