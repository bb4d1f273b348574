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
    ap.add_argument("--vit-chunk", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency-runs", type=int, default=5)
    return ap.parse_args()


def cpu_baseline(cfg, sd, seed, budget_s=25.0):
    """The oracle (a CPU port of the reference's algorithm, fp32 contractions with
    the reference's bf16 rounding points) on the host cores, B=1, on a BOUNDED
    sample: the stages are timed one by one and the run stops once ``budget_s``
    is spent; untimed stages are extrapolated by their FLOP ratio and said so."""
    from moondream_amd import synth
    from oracle import moondream_oracle as O

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    O.cache_fp32_weights(sd_cpu)
    orc = O.Oracle(cfg, sd_cpu, fast=False)
    arr = synth.synthetic_image_array(0, seed)
    t_start = time.perf_counter()
    # stage 1: ViT on ONE crop (the second crop is identical work)
    x = O.normalize_crops(arr[None])
    t0 = time.perf_counter()
    feats = O.vision_encoder(x, orc.sd, cfg)
    t_vit1 = time.perf_counter() - t0
    note = [f"ViT 1 crop {t_vit1:.2f}s (x2 crops)"]
    t_vit = 2 * t_vit1
    # stage 2: projector + 730-token image prefill, if the budget allows; else FLOP-ratio estimate
    flop_vit1, flop_rest = 666.45e9, 51.98e9 + 1868.4e9
    if (time.perf_counter() - t_start) + t_vit1 * flop_rest / flop_vit1 < budget_s:
        t0 = time.perf_counter()
        g = cfg.vision.enc_n_layers
        img = O.vision_projection(feats[0], feats[0].reshape(g, g, -1), orc.sd, cfg)
        xx = torch.cat([orc.embed([cfg.tokenizer.bos_id]), img], dim=0)
        kv = O.OracleKV.empty(cfg)
        O.text_decoder(xx, orc.sd, cfg, kv, torch.arange(xx.shape[0]), orc.cos, orc.sin)
        t_rest = time.perf_counter() - t0
        note.append(f"projector + 730-token prefill {t_rest:.2f}s")
        pos = xx.shape[0]
    else:
        t_rest = t_vit1 * flop_rest / flop_vit1
        note.append(f"projector + prefill extrapolated by FLOPs to {t_rest:.2f}s")
        kv, pos = O.OracleKV.empty(cfg), 730
    # stage 3: decode tokens until the budget is spent (at least one)
    tok_times = []
    emb = orc.embed([5])
    while len(tok_times) < 4 and (not tok_times or time.perf_counter() - t_start < budget_s):
        t0 = time.perf_counter()
        orc.decode_token(emb, pos, kv)
        pos += 1
        tok_times.append(time.perf_counter() - t0)
    per_tok = float(np.median(tok_times))
    note.append(f"{len(tok_times)} decode steps, median {per_tok:.2f}s/token")
    return t_vit + t_rest, per_tok, cores, "; ".join(note)


def main():
    args = parse()
    from moondream_amd import _lib, synth, dist as mdist
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

    B, T = args.batch, args.tokens
    n_total = B * world
    mine = mdist.shard_range(n_total, rank, world)
    images = [synth.synthetic_image(i, args.seed) for i in mine]
    prompt = cfg.tokenizer.templates["caption"]["normal"]
    prompts = [prompt] * len(images)

    def step():
        ids = model.batch_generate_ids(images, prompts, max_tokens=T, ignore_eos=True)
        local_ids = torch.tensor(ids, dtype=torch.int32, device=dev)
        return mdist.gather_token_ids(local_ids)

    for _ in range(args.warmup):
        step()
    mdist.barrier()
    torch.cuda.synchronize()
    lib.md_profile_gemm(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    mdist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = mdist.max_over_ranks(elapsed, dev)

    # dominant kernel (bf16 MFMA GEMM): algorithmic flops / HIP-event time over the timed region
    f, ms, n = C.c_double(), C.c_double(), C.c_int64()
    _lib.check(lib.md_profile_gemm_read(0, C.byref(f), C.byref(ms), C.byref(n)))
    gemm_tflops = f.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
    by, ms1, n1 = C.c_double(), C.c_double(), C.c_int64()
    _lib.check(lib.md_profile_gemm_read(1, C.byref(by), C.byref(ms1), C.byref(n1)))
    lib.md_profile_gemm(0)
    stream_gbs = by.value / (ms1.value * 1e-3) / 1e9 if ms1.value > 0 else 0.0

    if rank != 0:
        return
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
            "workload": f"Moondream-{args.model.upper()} bf16 batch_generate (caption): {B} images/GPU x 378x378 "
                        f"(2 crops each, both encoded), 5-token prompt, {T} greedy decode tokens, seeded synthetic weights",
            "batch_per_gpu": B, "decode_tokens": T, "parallelism": f"dp{world}",
        },
        "roofline": {
            "bound": "mfma",
            "kernel": "gemm_bf16_kernel (v_mfma_f32_32x32x16_bf16)",
            "achieved": gemm_tflops,
            "peak": 2500.0,
            "unit": "TFLOP/s",
            "frac": gemm_tflops / 2500.0,
            "traffic": None,
            "launches": int(n.value),
            "share_of_step": (ms.value * 1e-3) / elapsed if elapsed > 0 else None,
        },
        "decode_gemm": {
            "bound": "hbm", "kernel": "gemm_skinny_kernel (m <= 64 weight stream)", "achieved": stream_gbs,
            "peak": 8000.0, "unit": "GB/s", "frac": stream_gbs / 8000.0, "launches": int(n1.value),
            "share_of_step": (ms1.value * 1e-3) / elapsed if elapsed > 0 else None,
        },
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

    if world == 1 and not args.no_cpu_baseline:
        t_enc, per_tok, cores, note = cpu_baseline(cfg, sd, args.seed)
        est = 1.0 / (t_enc + per_tok * (T + 1))
        result["cpu_baseline"] = {
            "value": est, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle (CPU port of the reference algorithm, fp32 contractions), B=1, bounded sample: {note}; "
                      f"images/s = 1 / (encode + {T + 1} x per-token)",
        }
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
