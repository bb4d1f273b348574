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


def cpu_baseline(cfg, sd, seed, tokens_timed=8):
    """The oracle (a port of the reference's algorithm) on the host cores, B=1,
    bounded sample: one image encode + prompt prefill + `tokens_timed` decode steps."""
    from moondream_amd import synth
    from oracle.moondream_oracle import Oracle

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    orc = Oracle(cfg, {k: v.cpu() for k, v in sd.items()}, fast=True)
    arr = synth.synthetic_image_array(0, seed)
    crops = np.stack([arr, arr])
    prompt = cfg.tokenizer.templates["caption"]["normal"]
    t0 = time.perf_counter()
    pos, kv = orc.encode_image(crops, (1, 1))
    t_enc = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.generate(prompt, pos, kv, tokens_timed, keep_logits=False)
    t_gen = time.perf_counter() - t0
    return t_enc, t_gen, cores


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
    _lib.check(lib.md_profile_gemm_read(C.byref(f), C.byref(ms), C.byref(n)))
    lib.md_profile_gemm(0)
    gemm_tflops = f.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0

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
            "gemm_share_of_step": (ms.value * 1e-3) / elapsed if elapsed > 0 else None,
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
        timed = 8
        t_enc, t_gen, cores = cpu_baseline(cfg, sd, args.seed, timed)
        per_tok = t_gen / (timed + 1)
        est = 1.0 / (t_enc + per_tok * (T + 1))
        result["cpu_baseline"] = {
            "value": est, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle (CPU port of the reference algorithm, torch bf16 GEMMs), B=1: 1 image encode "
                      f"({t_enc:.2f}s) + prompt prefill and {timed} decode steps ({t_gen:.2f}s); per-token time "
                      f"extrapolated to {T} tokens",
        }
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
