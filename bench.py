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
    ap.add_argument("--no-second-oracle", action="store_true",
                    help="skip the parity calibration: the reference's own ATen calls through torch-ROCm on the 64 bench images")
    ap.add_argument("--latency-runs", type=int, default=5)
    ap.add_argument("--no-fp8-leg", action="store_true", help="skip the auxiliary FP8-decode-weights leg")
    ap.add_argument("--no-fp8-full-leg", action="store_true", help="skip the auxiliary full-FP8 leg (fp8 MFMA for ViT / projector / prefill + fp8 decode weights)")
    ap.add_argument("--no-dedup-leg", action="store_true", help="skip the auxiliary identical-crop-dedup leg")
    ap.add_argument("--no-graphs", action="store_true", help="launch decode steps eagerly instead of hipGraph replay")
    ap.add_argument("--no-pipeline", action="store_true", help="run each step's encode and decode back to back on one stream")
    ap.add_argument("--prompt", choices=["caption", "vqa32"], default="caption",
                    help="caption: the 5-id caption template; vqa32: 32-id seeded question prompts (SURVEY 8d)")
    ap.add_argument("--no-vqa-leg", action="store_true", help="skip the auxiliary 32-token-prompt measurement")
    ap.add_argument("--no-strict-leg", action="store_true", help="skip the strict-batch-invariance leg (batch == sequential bit for bit, priced)")
    ap.add_argument("--int4-leg", action="store_true", help="also run the int4-checkpoint weight-stream leg (opt-in mode, off in the default run)")
    ap.add_argument("--no-detect13-leg", action="store_true", help="skip the BASELINE configs[4]-shaped leg (768x1024 -> 13 crops, detect)")
    ap.add_argument("--detect13-batch", type=int, default=32, help="images per step of that leg (configs[4]: 256 over 8 GPUs)")
    ap.add_argument("--w4-grid", type=int, default=0,
                    help="persistent workgroups of the four-wave tile GEMM in the pipelined timed region (0 = one per CU): fewer leave "
                         "whole CUs to the decode stream's kernels")
    ap.add_argument("--only-timed-steps", action="store_true",
                    help="exit after the timed region (rocprofv3 --pmc passes: exactly --steps steps of kernels in the trace)")
    ap.add_argument("--leg", choices=["caption", "detect13", "detect13_fp8"], default="caption",
                    help="the workload the ONE JSON line is about, at any --gpus N.  caption (default): BASELINE configs[2] / [3] "
                         "(batch_generate, 64 images/GPU); detect13: 768x1024 images (13 crops), batch_detect, --detect13-batch "
                         "images/GPU; detect13_fp8: the same in the fp8 mode = BASELINE configs[4] (256 images over 8 GPUs)")
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
    # as in main(): every rank builds the (synthetic) weights, rank 0's copy is broadcast, every rank verifies the bytes
    full = {"w": torch.arange(24, dtype=torch.float32).reshape(4, 6).to(torch.bfloat16)}
    sd = mdist.broadcast_state_dict(full if rank == 0 else None, mdist.state_dict_template(full), dev)
    assert sd["w"].float().sum().item() == 276.0
    assert mdist.max_over_ranks(0.0 if torch.equal(sd["w"].cpu(), full["w"]) else 1.0, dev) == 0.0
    # the bench's own weight path -- the product's DataParallelEngine: every rank generates, rank 0's copy is broadcast and
    # verified -- with the tiny config and no model behind it
    from moondream_amd import synth
    from moondream_amd.config import get_config
    from moondream_amd.parallel import DataParallelEngine
    tiny = get_config("tiny")
    eng = DataParallelEngine(tiny, state_dict_fn=lambda d: synth.synthetic_state_dict(tiny, seed=3, device=d), verify_broadcast=True,
                             device=dev, model_factory=lambda cfg, sd, d, **kw: sd)
    wrep = eng.weights_report
    assert (wrep is None) == (not mdist.collectives_on()) and (wrep is None or (wrep["equal_to_local_copy_on_every_rank"] and wrep["bytes"] > 0))
    assert "text.wte" in eng.model
    mine = mdist.shard_range(3 * world + 1, rank, world)
    ids = torch.tensor([[i, i + 1] for i in mine], dtype=torch.int32, device=dev).reshape(len(mine), 2)
    blocks = mdist.gather_token_ids(ids, n_total=3 * world + 1)
    mdist.barrier()
    t = mdist.max_over_ranks(float(rank), dev)
    seen = mdist.ranks_seen(dev)
    per_rank = mdist.gather_floats(float(rank) + 0.5, dev)
    bindings = eng.gather_objects([eng.cpu_binding])   # every rank's core share (dist.bind_rank_cpus), on rank 0
    if os.environ.get("MD_SELFTEST_FAIL_RANK") == str(rank):  # tests: a failing rank must fail the whole launch
        raise RuntimeError(f"selftest: rank {rank} asked to fail")
    if rank == 0:
        got = torch.cat([b.cpu() for b in blocks], 0)[:, 0].tolist()
        assert got == list(range(3 * world + 1)), got
        assert per_rank == [r + 0.5 for r in range(world)], per_rank
        print(json.dumps({"selftest": "dist", "n_gpus": world, "max_rank": t, "items": len(got), "ranks_seen": seen,
                          "per_rank_ms_per_step": per_rank, "weights_broadcast": wrep, "cpu_binding": bindings}), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def decode_gemm_trace_figure(args):
    """The decode-regime GEMMs of one token from the committed rocprofv3 kernel trace of an eager B = 64 step
    (profiles/r05_eager_step_timeline.txt, tools/gpu_r5_eager_step_timeline.sh): a HIP-event bracket around a ~20-us launch contains
    the events' own cost, a kernel trace does not -- the live `achieved` above understates these launches by ~15 %.  None when the
    file is absent or the run is not the configuration it was taken on."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r05_eager_step_timeline.txt")
    non_default = args.w4_grid != 0 or any(k.startswith(("MD_GEMM_", "MD_DECODE_", "MD_ATTN_", "MD_SMALL_M_RULE", "MD_W4_")) for k in os.environ)
    if args.model != "2b" or args.batch != 64 or non_default or not os.path.isfile(path):
        return None
    import re

    us = {}
    for line in open(path):
        m = re.search(r"(gemm_bf16_kernel<64, 64, 2, 1, (\d)|gemm_pair_kernel<64, 64).*n=\s*(\d+)\s+avg=\s*([0-9.]+) us", line)
        if m:
            us["pair" if m.group(1).startswith("gemm_pair") else ("qkv_fc1" if m.group(2) == "1" else "lm_head")] = float(m.group(4))
    if set(us) != {"pair", "qkv_fc1", "lm_head"}:
        return None
    n_layers, token_bytes = 24, 2.627e9
    t = (n_layers * (us["qkv_fc1"] + us["pair"]) + us["lm_head"]) * 1e-6
    return {"source": "profiles/r05_eager_step_timeline.txt (rocprofv3 --kernel-trace of one eager step, committed in round 5)",
            "measured_in_this_run": False,
            "us_per_launch": us, "ms_per_token": t * 1e3, "achieved": token_bytes / t / 1e9, "unit": "GB/s", "frac": token_bytes / t / 8e12,
            "note": "a HIP-event bracket around a ~20-us launch contains the events' own cost; the kernel trace does not"}


def _one_numa_node_physical_cores():
    """The logical CPUs this process may use, narrowed to ONE NUMA node and one hardware thread per core (Linux sysfs; anything
    unreadable -> the process's own affinity mask unchanged).  Returns (cpus, description)."""
    allowed = set(os.sched_getaffinity(0))

    def cpulist(path):
        out = set()
        with open(path) as f:
            for part in f.read().strip().split(","):
                if part:
                    a, _, b = part.partition("-")
                    out.update(range(int(a), int(b or a) + 1))
        return out

    try:
        nodes = sorted(d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())
        best = max(((len(cpulist(f"/sys/devices/system/node/{d}/cpulist") & allowed), d) for d in nodes), default=(0, None))
        if best[0] == 0:
            return sorted(allowed), "no NUMA information: the process's affinity mask"
        cpus = cpulist(f"/sys/devices/system/node/{best[1]}/cpulist") & allowed
        phys = set()
        for c in cpus:
            try:
                sib = cpulist(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") & cpus
            except OSError:
                sib = {c}
            phys.add(min(sib) if sib else c)
        return sorted(phys), f"{best[1]} of {len(nodes)} NUMA node(s), one hardware thread per core"
    except OSError:
        return sorted(allowed), "no NUMA information: the process's affinity mask"


def _set_affinity_all_threads(cpus):
    """sched_setaffinity on every thread of this process (the OpenMP pool exists already and keeps the mask it was created
    under otherwise).  Returns {tid: previous mask} for the restore."""
    prev = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            prev[int(tid)] = os.sched_getaffinity(int(tid))
            os.sched_setaffinity(int(tid), cpus)
        except (OSError, ValueError):
            pass
    return prev


def reference_cpu_timing(cfg, sd, seed, T, n_images=3, threads=None):
    """The UNMODIFIED reference (moondream/torch under MOONDREAM_REFERENCE or /root/reference; tokenizer stubbed, the same seeded
    synthetic weights) on this host's cores: B = 1 sequential, encode_image + the answer generator, wall clock, one warm-up image
    then ``n_images - 1`` timed (method of the reference's sample.py:159-207).  Only where the checkout exists -- never on the
    driver's GPU box; a checker-side measurement, outside every timed region.

    ``threads`` = (encode, decode) torch thread counts and the process is pinned to the physical cores of one NUMA node -- the
    SAME treatment ``cpu_baseline`` gives the port (round 6: with torch's default of one thread per logical CPU the reference
    took 28 s per image on a 256-thread host, 8x slower than the port beside it, which says nothing about either)."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import make_golden as mg
    from moondream_amd import synth

    model, ref_md = mg.load_reference(cfg, {k: v.cpu() for k, v in sd.items()})
    prompt = cfg.tokenizer.templates["caption"]["normal"]
    default_threads = torch.get_num_threads()
    prev_masks = None
    if threads is not None:
        enc_t, dec_t = (max(1, int(x)) for x in threads)

        def with_threads(fn, n):
            def f(*a, **k):
                torch.set_num_threads(n)
                return fn(*a, **k)
            return f

        # instance attributes shadow the class's methods: the reference's own code, run under a thread count per phase
        model._vis_enc, model._vis_proj = with_threads(model._vis_enc, enc_t), with_threads(model._vis_proj, enc_t)
        model._prefill, model._decode_one_tok = with_threads(model._prefill, enc_t), with_threads(model._decode_one_tok, dec_t)
        cpus, _ = _one_numa_node_physical_cores()
        prev_masks = _set_affinity_all_threads(cpus)
    t_enc, t_gen = [], []
    try:
        for i in range(n_images):
            r = mg.run_reference_caption(model, ref_md, synth.synthetic_image_array(i, seed, (378, 378)), prompt, T)
            t_enc.append(r["t_enc"])
            t_gen.append(r["t_gen"])
    finally:
        torch.set_num_threads(default_threads)
        if prev_masks:
            for tid, mask in prev_masks.items():
                try:
                    os.sched_setaffinity(tid, mask)
                except OSError:
                    pass
    enc, gen = float(np.median(t_enc[1:])), float(np.median(t_gen[1:]))
    how = f"{default_threads} torch threads" if threads is None else f"{enc_t} / {dec_t} torch threads (encode / decode), one NUMA node's physical cores"
    return {"images_per_sec": 1.0 / (enc + gen), "threads": default_threads if threads is None else max(enc_t, dec_t), "encode_s": enc, "generate_s": gen,
            "sample": f"the unmodified reference, B=1, {n_images - 1} images after one warm-up: encode_image {enc:.2f}s + {T} greedy tokens "
                      f"{gen:.2f}s, {how}"}


def cpu_baseline(cfg, sd, seed, T, budget_s=30.0):
    """The CPU path of this workload timed on THIS box's host cores.  /root/reference does not exist on the
    GPU box, so what runs is the oracle in its ``fast`` mode: the reference's own ATen calls (bf16 F.linear,
    F.scaled_dot_product_attention under the bool mask over all 2048 cache slots, F.layer_norm, tanh-GELU) in the
    reference's order, B=1 sequential like the reference (it has no batching).

    Made reproducible in round 4 (round 3's figure moved 0.11 .. 0.39 images/s between boxes and came out below the
    unmodified reference on 8 cores): (1) every thread of the process is pinned to the physical cores of ONE NUMA node
    BEFORE the weights are copied to host memory, so the copy's first touch puts them on that node; (2) the thread count is
    picked on the WHOLE phases -- ``encode_image`` for the encode, a whole-model decode step for the tokens (a single decoder
    block's weights sit in L3 and predicted 23 ms/token where the 2.6 GB stream measured 197) -- not on one block; (3) the
    encode is the best of three runs and the spread is reported.  Bounded: the sweep stops when ``budget_s`` is spent.
    The real reference timed in the build container (oracle/make_golden.py reftime) is the committed cross-check:
    profiles/r02_reference_cpu_timing_build_container.json.  Returns (images/s, threads, note, details)."""
    from moondream_amd import synth
    from oracle import moondream_oracle as O

    t_begin = time.perf_counter()
    log = lambda m: print(f"[cpu_baseline +{time.perf_counter() - t_begin:5.1f}s] {m}", file=sys.stderr, flush=True)
    threads_before = torch.get_num_threads()
    cpus, where = _one_numa_node_physical_cores()
    prev_masks = _set_affinity_all_threads(cpus)
    O.ATEN_CALLS = True  # layer norm / GELU / residual adds as the single ATen calls the reference makes
    try:
        n_cpu = len(cpus)
        torch.set_num_threads(min(n_cpu, 32))
        _set_affinity_all_threads(cpus)  # (threads the pool has just created)
        orc = O.Oracle(cfg, {k: x.cpu() for k, x in sd.items()}, fast=True)
        log(f"pinned to {n_cpu} cpus ({where}); weights on the host")
        img = synth.synthetic_image_array(0, seed, (378, 378))
        crops = np.stack([img, img])
        prompt = cfg.tokenizer.templates["caption"]["normal"]
        cands = sorted({min(n_cpu, c) for c in (8, 16, 32, 64, 128)})
        with torch.inference_mode():
            # ---- encode_image: thread count on the whole phase, then best of three
            enc_by_threads = {}
            first = True
            for nt in cands:
                torch.set_num_threads(nt)
                _set_affinity_all_threads(cpus)
                ts = []
                for rep in range(2 if first else 1):  # the very first run pays page faults / thread start-up
                    t0 = time.perf_counter()
                    pos, kv = orc.encode_image(crops, (1, 1))
                    ts.append(time.perf_counter() - t0)
                first = False
                enc_by_threads[nt] = ts[-1]
                log(f"encode_image, {nt} threads: {ts[-1]:.2f}s")
                spent = time.perf_counter() - t_begin
                # more threads stopped helping (on a 2-socket host 64 threads took 10-13 s where 16 take 0.9), or the budget is near
                if ts[-1] > 1.25 * min(enc_by_threads.values()) or spent + 3.5 * min(enc_by_threads.values()) > 0.8 * budget_s:
                    break
            threads = min(enc_by_threads, key=enc_by_threads.get)
            torch.set_num_threads(threads)
            enc_runs = [enc_by_threads[threads]]
            while len(enc_runs) < 3 and time.perf_counter() - t_begin + 1.2 * min(enc_runs) < 0.85 * budget_s:
                t0 = time.perf_counter()
                pos, kv = orc.encode_image(crops, (1, 1))
                enc_runs.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            logits, _, pos = orc.prefill_prompt(prompt, pos, kv)
            t_prompt = time.perf_counter() - t0
            # ---- decode: its own thread count, probed on whole-model steps (the one-row step streams 2.6 GB of weights)
            tok = int(torch.argmax(logits.float()))

            def step():
                nonlocal tok, pos
                t0 = time.perf_counter()
                lg, _ = orc.decode_token(orc.embed([tok]), pos, kv)
                dt = time.perf_counter() - t0
                tok, pos = int(torch.argmax(lg.float())), pos + 1
                return dt

            step()
            tok_by_threads = {}
            for nt in sorted({min(n_cpu, c) for c in (4, 8, 16, 32, 64)}):
                torch.set_num_threads(nt)
                tok_by_threads[nt] = min(step() for _ in range(3))
                log(f"decode step, {nt} threads: {tok_by_threads[nt] * 1e3:.0f}ms")
                if tok_by_threads[nt] > 1.25 * min(tok_by_threads.values()):
                    break
            dec_threads = min(tok_by_threads, key=tok_by_threads.get)
            torch.set_num_threads(dec_threads)
            step()
            groups = [[step() for _ in range(6)] for _ in range(3)]  # best of three groups' medians: the step time is bimodal
            steps = min(groups, key=lambda g: float(np.median(g)))  # (27 .. 110 ms on one box) with the pool's wake-ups
            all_steps = [x for g in groups for x in g]
        t_enc, t_tok = min(enc_runs), float(np.median(steps))
        est_total = t_enc + t_prompt + T * t_tok
        details = {
            "pinned_cpus": n_cpu, "pinning": where, "encode_threads": threads, "decode_threads": dec_threads,
            "encode_image_s_runs": [round(x, 3) for x in enc_runs], "encode_image_s_by_threads": {str(k): round(v, 3) for k, v in enc_by_threads.items()},
            "prompt_prefill_s": round(t_prompt, 3), "decode_ms_per_token_min_median_max": [round(min(all_steps) * 1e3, 1), round(t_tok * 1e3, 1), round(max(all_steps) * 1e3, 1)],
            "decode_ms_per_token_group_medians": [round(float(np.median(g)) * 1e3, 1) for g in groups],
            "decode_ms_by_threads": {str(k): round(v * 1e3, 1) for k, v in tok_by_threads.items()},
            "value_spread": [1.0 / (max(enc_runs) + t_prompt + T * max(float(np.median(g)) for g in groups)),
                             1.0 / (min(enc_runs) + t_prompt + T * min(float(np.median(g)) for g in groups))],
            "seconds": round(time.perf_counter() - t_begin, 1),
        }
        note = (f"WHOLE phases, B=1, pinned to {n_cpu} physical cores of one NUMA node: encode_image {t_enc:.2f}s (best of {len(enc_runs)}, "
                f"{threads} threads, picked on the whole phase), {len(prompt)}-token prompt prefill {t_prompt:.2f}s, decode {t_tok * 1e3:.0f}ms/token "
                f"(best of 3 groups' medians of {len(steps)}, {dec_threads} threads, picked on whole-model steps); images/s = 1 / (encode + prompt + {T} x token)")
        return 1.0 / est_total, threads, note, details
    finally:
        for tid, mask in prev_masks.items():
            try:
                os.sched_setaffinity(tid, mask)
            except OSError:
                pass
        torch.set_num_threads(threads_before)


def second_oracle(cfg, sd, seed, tokens, device, fixture="md2b_bench64"):
    """Parity CALIBRATION (SURVEY 8c's second oracle): the oracle in ``fast`` mode = the reference's own ATen calls (bf16
    F.linear, F.scaled_dot_product_attention under the bool mask over all 2048 slots, F.layer_norm, tanh-GELU) in the
    reference's order, B = 1 sequential like the reference -- executed by torch-ROCm's kernels on THIS GPU instead of the
    CPU kernels that wrote tests/golden/md2b_bench64.npz.  Same code, same weights, same images; only the BLAS / attention
    backend differs.  Teacher-forced on the fixture's ids it yields (a) how many of the 64 sequences that second, equally
    correct evaluation would have kept (every decision's argmax equal to the reference's token) and (b) its logit error at
    the reference's top-8 candidates: the noise floor this build's own count and error are judged against.  A checker,
    outside every timed region."""
    from moondream_amd import synth
    from oracle import moondream_oracle as O

    path = os.path.join(REPO, "tests", "golden", fixture + ".npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    n, t = g["tokens"].shape[0], min(tokens, g["tokens"].shape[1])
    per_image_prompts = g["prompt"].ndim == 2   # md2b_vqa64: one 32-id question prompt per image; md2b_bench64: the caption template
    aten_was, O.ATEN_CALLS = O.ATEN_CALLS, True
    t0 = time.perf_counter()
    try:
        orc = O.Oracle(cfg, sd, fast=True, device=device)
        prompt = g["prompt"].tolist()
        exact, errs, first_div_margins, dec_agree, dec_wide_disagree = 0, [], [], 0, 0
        with torch.inference_mode():
            for i in range(n):
                img = synth.synthetic_image_array(i, seed, (378, 378))
                pos, kv = orc.encode_image(np.stack([img, img]), (1, 1))
                run = orc.generate(g["prompt"][i].tolist() if per_image_prompts else prompt, pos, kv, max_tokens=t, eos_id=-1,
                                   forced=g["tokens"][i, :t].tolist())
                lg = torch.stack([x.float().cpu() for x in run.logits[: t + 1]])  # [t + 1, V]
                got = torch.gather(lg, 1, torch.as_tensor(g["top8_idx"][i, : t + 1], dtype=torch.int64)).numpy()
                errs.append(np.abs(got - g["top8_val"][i, : t + 1])[np.isfinite(g["top8_val"][i, : t + 1])])
                own = lg[:t].argmax(dim=1).numpy()  # (CPU argmax: lowest index among ties, like the reference's)
                diff = np.nonzero(own != g["tokens"][i, :t])[0]
                dec_agree += t - len(diff)
                dec_wide_disagree += int((g["margins"][i, :t][diff] > 0.5).sum())
                if len(diff) == 0:
                    exact += 1
                else:
                    first_div_margins.append(float(g["margins"][i, diff[0]]))
        e = np.concatenate(errs)
        return {"exact": exact, "of": n, "max_logit_err": float(e.max()), "p99_logit_err": float(np.quantile(e, 0.99)),
                "max_divergence_margin": max(first_div_margins, default=0.0), "seconds": round(time.perf_counter() - t0, 1),
                "tf_decisions_agree": dec_agree, "tf_decisions": n * t, "tf_decisions_disagree_at_margin_above_0.5": dec_wide_disagree,
                "what": "oracle fast mode (the reference's own ATen calls, B=1) run by torch-ROCm on this GPU, teacher-forced on the "
                        "reference's ids: sequences whose every argmax equals the reference's token / logit error at its top-8 candidates"}
    finally:
        O.ATEN_CALLS = aten_was


def check_parity(model, images, prompts, ids_per_image, cfg_name, seed, prompt_kind, tokens, floor_from=None):
    """The ids this run generated against the REFERENCE's ids for the same images (tests/golden/md2b_bench64.npz,
    written by oracle/make_golden.py bench64 from the unmodified reference), with a MEASURED licence
    (moondream_amd/parity.py): the HIP path is run teacher-forced on the reference's ids and its logits at the
    reference's top-8 ids of all 64 x 33 decisions are compared with the reference's; a sequence may leave the
    reference's stream only at a decision whose reference margin is <= min(2 x the largest logit error measured there, 0.5)
    and which the errors measured on its own two logits cover; the count of identical sequences has a floor calibrated by the
    second oracle of this very run (``exact_floor``; the fixture has 9 sequences with an exact tie and only 12 whose smallest
    margin exceeds 0.25, so the count is a noisy statistic).  Outside the timed region."""
    from moondream_amd import parity as P

    fixture = {"caption": "md2b_bench64", "vqa32": "md2b_vqa64"}.get(prompt_kind)   # BASELINE configs[2] / configs[1] at bench scale
    path = os.path.join(REPO, "tests", "golden", f"{fixture}.npz")
    if cfg_name != "2b" or seed != 1 or fixture is None or not os.path.exists(path):
        return {"parity_checked": 0, "parity_ok": None, "parity_note": "no reference fixture for this configuration"}
    g = np.load(path)
    n = min(len(ids_per_image), g["tokens"].shape[0], len(images))
    t = min(tokens, g["tokens"].shape[1])
    if g["prompt"].ndim == 2:   # per-image prompts: the run must have used exactly the fixture's
        assert [list(p) for p in prompts[:n]] == g["prompt"][:n].tolist(), "vqa32 prompts differ from the fixture's"
    got_topk = model.teacher_forced_logits(images[:n], prompts[:n], g["tokens"][:n, :t], g["top8_idx"][:n, : t + 1]).numpy()
    return P.parity_report([ids[:t] for ids in ids_per_image[:n]], g["tokens"][:n, :t].tolist(), g["margins"][:n],
                           got_topk, g["top8_val"][:n, : t + 1], tokens=t, min_exact=exact_floor(n, t, floor_from),
                           ref_topk_idx=g["top8_idx"][:n, : t + 1])


STATIC_EXACT_FLOOR = 40   # of 64: the flat floor of round 3; the calibrated floor can only RAISE it (advisor, round 4)
SECOND_ORACLE_SANITY = 44  # of 64: the reference's own ATen calls under torch-ROCm keep 50-51; far below that the calibration
                           # itself is broken (a bad BLAS day or a bug shared with it) and must not lower any gate


def exact_floor(n, t, second):
    """The floor on identical sequences: max(static floor, N2 - 2 sigma).  N2 (of 64) is the second oracle's count from THIS run
    -- a build is not asked to agree with the reference more often than the reference's own code does under another BLAS --
    minus two binomial standard deviations of a 64-trial count at that rate; it can raise the gate above the static 40, never
    lower it, and a second oracle that itself falls below SECOND_ORACLE_SANITY is reported as a failure of the calibration
    (``parity_second_oracle_sane`` in the line), not used."""
    if t != 32:
        return None
    static = (STATIC_EXACT_FLOOR * n) // 64
    if not second or second.get("of") != 64 or n != 64 or second["exact"] < SECOND_ORACLE_SANITY:
        return static
    n2 = second["exact"]
    return max(static, int(n2 - np.ceil(2.0 * np.sqrt(max(n2 * (64 - n2), 1) / 64.0))))


def int4_leg(cfg, sd, images, prompts, T, args, dev):
    from moondream_amd.moondream import MoondreamModel, IdTokenizer

    def quantize(w):  # the checkpoint format dequantize_tensor reads (layers.py:38-44): groups of 128, scale + zero point
        rows = w.float().reshape(-1, 128)
        lo, hi = rows.min(1, keepdim=True).values, rows.max(1, keepdim=True).values
        scale = ((hi - lo) / 15).clamp_min(1e-8)
        zero = -lo / scale
        q = torch.clamp(torch.round(rows / scale + zero), 0, 15).to(torch.uint8)
        step = q.shape[0] // 2
        return (q[:step] << 4) | q[step:], scale, zero

    qsd = dict(sd)
    for i in range(cfg.text.n_layers):
        for n in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"):
            p = f"text.blocks.{i}.{n}"
            qsd[p + ".weight.packed"], qsd[p + ".weight.scale"], qsd[p + ".weight.zero_point"] = quantize(qsd.pop(p + ".weight"))
    m4 = MoondreamModel(cfg, qsd, device=dev, tokenizer=IdTokenizer(), max_batch=args.batch, vit_chunk_crops=args.vit_chunk)
    assert m4.w.has_int4_source()
    m4.single_sequence_kernel = False  # like for like: the lone sequence of the latency figures on the batched kernels in both modes
    out = {}
    ids = {}
    for mode in ("int4_stream", "bf16_stream"):
        m4.enable_int4_decode(mode == "int4_stream")
        m4.batch_generate_ids(images, prompts, max_tokens=T, ignore_eos=True)  # warm-up (arenas)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            ids[mode] = m4.batch_generate_ids(images, prompts, max_tokens=T, ignore_eos=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / 2
        m4.collect_timing = True
        m4.batch_generate_ids(images, prompts, max_tokens=T, ignore_eos=True)
        torch.cuda.synchronize()
        m4.collect_timing = False
        lat = []
        for i in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            m4.batch_generate_ids(images[:1], prompts[:1], max_tokens=T, ignore_eos=True)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t1)
        out[mode] = {"images_per_sec_non_pipelined": len(images) / dt, "ms_per_step": dt * 1e3,
                     "phase_ms": {k: round(v, 2) for k, v in m4.last_phase_ms.items()},
                     "single_image_latency_ms": float(np.median(lat[1:]) * 1e3)}
    m4.single_sequence_kernel = True  # (bf16 stream attached last: the persistent single-sequence kernel, for reference)
    lat = []
    for i in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        m4.batch_generate_ids(images[:1], prompts[:1], max_tokens=T, ignore_eos=True)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t1)
    out["bf16_stream"]["single_image_latency_ms_persistent_kernel"] = float(np.median(lat[1:]) * 1e3)
    same = sum(a == b for a, b in zip(ids["int4_stream"], ids["bf16_stream"]))
    out.update({
        "sequences_identical": same, "of": len(images),
        "decode_phase_speedup": out["bf16_stream"]["phase_ms"]["decode"] / out["int4_stream"]["phase_ms"]["decode"],
        "note": "the synthetic decoder blocks quantised into the reference's QuantizedLinear format (4-bit, groups of 128) and loaded from "
                "the triples; decode launches (<= 64 rows) stream the nibbles and rebuild bf16(bf16(q - zero) * scale) in registers "
                "(md_linear_fp8.format = MD_WSTREAM_INT4_G128) vs the dequantised bf16 copy of the same model: the same weights bit for "
                "bit, sums in another K order; bf16_stream's lone sequence runs on the batched kernels too (like for like); never `value`",
    })
    del m4
    torch.cuda.empty_cache()
    return out


def detect13_leg(model, cfg, args, dev, fp8=False):
    """BASELINE.json configs[4]'s WORKLOAD on one GPU (its fp8 arithmetic is a separate opt-in mode): seeded 768 x 1024
    images -> tiling (3, 4) = 13 crops each (image_crops.py:58-167), ``detect`` with a fixed ``max_objects``, 32 images per
    step.  Reported: images/s over whole steps (host tiling NOT hidden: one step after the other on one stream), the
    per-phase GPU times of one step, the vision phase against the MFMA roofline (13 x 666.45 + 51.98 GFLOP per image),
    and the objects of the first 8 images against the reference's (tests/golden/md2b_detect13.npz)."""
    from moondream_amd import parity as P
    from moondream_amd import synth

    gpath = os.path.join(REPO, "tests", "golden", "md2b_detect13.npz")
    g = np.load(gpath) if os.path.exists(gpath) and args.model == "2b" and args.seed == 1 else None
    size = tuple(int(x) for x in g["size"]) if g is not None else (768, 1024)
    max_objects = int(g["max_objects"]) if g is not None else 4
    obj = " ".join(str(t) for t in (g["object_ids"].tolist() if g is not None else [7, 8]))
    B2 = args.detect13_batch
    imgs = [synth.synthetic_image(i, args.seed, size) for i in range(B2)]
    st = {"max_objects": max_objects, "_run_all_objects": True}
    if fp8:  # BASELINE configs[4] proper: fp8 MFMA for the ViT / projector / prefill linears + fp8 decode weights (region head bf16)
        model.enable_fp8(imgs[:2])
    t_host = time.perf_counter()
    crops = [model._crop(im) for im in imgs[:4]]
    host_ms_per_image = (time.perf_counter() - t_host) / 4 * 1e3
    n_crops = int(crops[0][0].shape[0])
    res = model.batch_detect(imgs, [obj] * B2, settings=st)  # warm-up (arenas, KV slabs)
    torch.cuda.synchronize()
    steps = 2
    t1 = time.perf_counter()
    for _ in range(steps):
        res = model.batch_detect(imgs, [obj] * B2, settings=st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / steps
    # the same through the PIPELINED detect engine (MoondreamModel.batch_detect_pipelined: the next step's host tiling is cut by
    # background workers while the current step runs on the GPU -- what the pipelined caption engine of the headline number
    # does by construction); distinct image objects per step: a prefetched batch is keyed by identity
    copies = [[im.copy() for im in imgs] for _ in range(steps + 1)]
    gen = model.batch_detect_pipelined(((c, [obj] * B2) for c in copies), settings=st)
    next(gen)                      # first batch: its tiling is exposed (and the second batch's is started under it)
    model.wait_prefetched_crops()  # steady state: the batch about to run was tiled during the previous step
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for res_p in gen:
        pass
    torch.cuda.synchronize()
    dt_prefetch = (time.perf_counter() - t2) / steps
    assert [r["objects"] for r in res_p] == [r["objects"] for r in res]
    model.collect_timing = True
    model.batch_detect(imgs, [obj] * B2, settings=st)
    torch.cuda.synchronize()
    model.collect_timing = False
    phase = {k: round(v, 2) for k, v in model.last_phase_ms.items()}
    out = {
        "workload": f"Moondream-{args.model.upper()} bf16 batch_detect: {B2} images/GPU x {size[0]}x{size[1]} ({n_crops} crops each, tiling "
                    f"{tuple(crops[0][1])}), detect prompt, max_objects {max_objects} (every sequence runs all rounds)",
        "images_per_sec": B2 / dt, "ms_per_step": dt * 1e3, "images_per_sec_tiling_prefetched": B2 / dt_prefetch,
        "ms_per_step_tiling_prefetched": dt_prefetch * 1e3, "batch": B2, "crops_per_image": n_crops, "max_objects": max_objects,
        "phase_ms": phase, "host_tiling_ms_per_image_one_thread": host_ms_per_image,
        "note": "steps run back to back on one stream: phase_ms.host_tiling (PIL LANCZOS resize + crop cutting on the thread pool, "
                "reference image_crops.py:124-167) is NOT hidden behind GPU work in images_per_sec; *_tiling_prefetched: the next "
                "step's tiling is started before the current step runs (MoondreamModel.batch_detect_pipelined), same objects",
    }
    if phase.get("vision"):
        fl = B2 * (n_crops * FLOP_VIT_PER_CROP + 51.98e9)
        out["vit_encoder"] = {"achieved": fl / (phase["vision"] * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                              "frac": fl / (phase["vision"] * 1e-3) / 1e12 / 2500.0}
    if g is not None:
        objs = [r["objects"] for r in res]
        out["parity"] = P.detect_parity_fp8(objs, g) if fp8 else P.detect_parity(objs, g)
    if fp8:
        model.enable_fp8(on=False)
        out["workload"] = out["workload"].replace(" bf16 ", " FP8 (md_gemm_f8 for ViT / projector / prefill, e4m3 decode weights; region head bf16) ")
        out["parity_note"] = ("objects vs the bf16 reference's under the fp8 licence (moondream_amd/parity.py: FP8_REGION_LICENCE_ULPS): equality is "
                              "required only of objects whose every decision clears it; centre / size differences reported in bins")
    return out


def detect_job(engine, cfg, args, fp8):
    """``--leg detect13[_fp8] --gpus N``: BASELINE configs[4] (fp8) / its bf16 counterpart as ONE command at any N.  Weak
    scaling: every rank runs ``--detect13-batch`` 768x1024 images per step through the pipelined detect engine
    (``DataParallelEngine.batch_detect_pipelined``: per-rank lockstep object loop, next step's host tiling under the current
    step's GPU time, objects gathered on rank 0 inside the step); W untimed steps, K timed between barrier + synchronize,
    max over ranks.  Rank 0 checks the first images' objects against the reference's (tests/golden/md2b_detect13.npz)."""
    from moondream_amd import parity as P
    from moondream_amd import synth

    model, rank, world, dev = engine.model, engine.rank, engine.world, engine.device
    gpath = os.path.join(REPO, "tests", "golden", "md2b_detect13.npz")
    g = np.load(gpath) if os.path.exists(gpath) and args.model == "2b" and args.seed == 1 else None
    size = tuple(int(x) for x in g["size"]) if g is not None else (768, 1024)
    max_objects = int(g["max_objects"]) if g is not None else 4
    obj = " ".join(str(t) for t in (g["object_ids"].tolist() if g is not None else [7, 8]))
    B2 = args.detect13_batch
    n_total = B2 * world
    mine = engine.shard(n_total)
    imgs = [synth.synthetic_image(i, args.seed, size) for i in mine]
    st = {"max_objects": max_objects, "_run_all_objects": True}
    calib = None
    if fp8:
        calib = model.enable_fp8(imgs[:2])
    n_crops = int(model._crop(imgs[0])[0].shape[0])

    def make_batches(k):  # distinct image objects per step (a prefetched batch is keyed by identity), made OUTSIDE the timed region
        return [([im.copy() for im in imgs], [obj] * len(imgs)) for _ in range(k)]

    def steps(batches):
        return list(engine.batch_detect_pipelined(iter(batches), settings=st))

    steps(make_batches(max(2, args.warmup)))
    timed_batches = make_batches(args.steps)
    engine.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = steps(timed_batches)
    torch.cuda.synchronize()
    engine.barrier()
    elapsed = time.perf_counter() - t0
    per_rank_ms = engine.gather_floats(elapsed / args.steps * 1e3)
    elapsed = engine.max_over_ranks(elapsed)
    seen = engine.ranks_seen()
    model.collect_timing = True
    model.batch_detect(imgs, [obj] * len(imgs), settings=st)
    torch.cuda.synchronize()
    model.collect_timing = False
    phase = {k: round(v, 2) for k, v in model.last_phase_ms.items()}
    if rank != 0:
        return
    res = outs[-1]
    assert len(res) == n_total
    mode = "FP8 (md_gemm_f8 for ViT / projector / prefill, e4m3 decode weights + KV copy; region head bf16)" if fp8 else "bf16"
    line = {
        "metric": "images_per_sec", "value": n_total * args.steps / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": max(2, args.warmup), "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "ranks_seen": seen,
        "weights_broadcast": engine.weights_report, "cpu_binding": engine.cpu_binding, "per_rank_ms_per_step": [round(x, 3) for x in per_rank_ms], "scaling": "weak",
        "vs_baseline": None, "dtype": "fp8 (e4m3 operands, fp32 accumulate; bf16 residual stream)" if fp8 else "bf16", "data": "synthetic",
        "config": {"workload": f"Moondream-{args.model.upper()} {mode} batch_detect: {B2} images/GPU x {size[0]}x{size[1]} ({n_crops} crops "
                               f"each), detect prompt, max_objects {max_objects} (every sequence runs all rounds), seeded synthetic weights",
                   "batch_per_gpu": B2, "global_batch": n_total, "parallelism": f"dp{world}",
                   "host_latency_hiding": "pipelined detect engine: step i+1's PIL tiling on background workers under step i's GPU time"},
        "phase_ms": phase,
    }
    if phase.get("vision"):
        fl = len(imgs) * (n_crops * FLOP_VIT_PER_CROP + 51.98e9)
        tf = fl / (phase["vision"] * 1e-3) / 1e12
        line["roofline"] = {"bound": "mfma", "kernel": "vision phase (ViT blocks + projector) of one non-overlapped step, rank 0", "achieved": tf,
                            "peak": 5000.0 if fp8 else 2500.0, "unit": "TFLOP/s", "frac": tf / (5000.0 if fp8 else 2500.0), "traffic": None}
    if g is not None:
        objs = [r["objects"] for r in res]
        line["parity"] = P.detect_parity_fp8(objs, g) if fp8 else P.detect_parity(objs, g)
    if calib is not None:
        line["fp8_calibration"] = {k: v for k, v in calib.items() if not isinstance(v, (list, dict))}
    print(json.dumps(line), flush=True)


def main():
    args = parse()
    from moondream_amd.parallel import DataParallelEngine

    # `python bench.py --gpus N` with no launcher: become N ranks under torch.distributed.run
    rc = DataParallelEngine.launch(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if rc is not None:
        sys.exit(rc)
    if args.selftest_dist:
        return selftest_dist(args)
    from moondream_amd import _lib, synth
    from moondream_amd.config import get_config
    from moondream_amd.moondream import MoondreamModel, IdTokenizer

    cfg = get_config(args.model)
    # The product's data-parallel runner (moondream_amd/parallel.py): one process per GPU; the checkpoint being SYNTHETIC (a
    # counter-based hash of the seed), every rank generates its own copy at once instead of 7 ranks idling while rank 0 does;
    # the RCCL broadcast of rank 0's flat buffer (what a real checkpoint takes) still runs and every rank checks the received
    # bytes against its own copy: the weight path over xGMI is exercised AND verified on every N > 1 run
    captured = {}

    def build_model(c, sd, d, **kw):
        captured["sd"] = sd
        return MoondreamModel(c, sd, device=d, **kw)

    engine = DataParallelEngine(cfg, state_dict_fn=lambda d: synth.synthetic_state_dict(cfg, seed=args.seed, device=d),
                                verify_broadcast=True, model_factory=build_model, tokenizer=IdTokenizer(), max_batch=args.batch,
                                vit_chunk_crops=args.vit_chunk)
    rank, world, dev, model, sd = engine.rank, engine.world, engine.device, engine.model, captured["sd"]
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    weights_broadcast = engine.weights_report
    lib = model.lib
    if args.leg != "caption":
        detect_job(engine, cfg, args, fp8=args.leg == "detect13_fp8")
        return
    if not args.no_graphs:
        model.compile()  # hipGraph replay of the device-resident decode steps

    B, T = args.batch, args.tokens
    n_total = B * world
    mine = engine.shard(n_total)
    images = [synth.synthetic_image(i, args.seed) for i in mine]
    caption_prompt = cfg.tokenizer.templates["caption"]["normal"]
    vqa_prompts = [synth.synthetic_vqa_prompt(cfg, i, args.seed) for i in mine]
    if args.prompt == "caption":
        prompt, prompts = caption_prompt, [caption_prompt] * len(images)
    else:
        prompt, prompts = vqa_prompts[0], vqa_prompts

    def run_steps(k, prompts=prompts):
        """k steps = k full passes over this rank's batch.  Pipelined mode overlaps the
        decode of step i with the encode of step i+1 (two HIP streams); every step's
        work, including its RCCL id gather (own stream), completes inside the call."""
        if args.no_pipeline:
            outs = [engine.gather_id_blocks(torch.tensor(model.batch_generate_ids(images, prompts, max_tokens=T, ignore_eos=True), dtype=torch.int32), n_total)
                    for _ in range(k)]
        else:
            outs = list(engine.batch_generate_ids_pipelined(((images, prompts) for _ in range(k)), n_total, max_tokens=T, ignore_eos=True))
        return outs

    if args.w4_grid and not args.no_pipeline:
        _lib.check(lib.md_gemm_set_tuning(b"w4_grid", args.w4_grid))
    if args.warmup:
        # pipelined mode alternates two KV slot groups and decodes two consecutive batches as one lockstep of 128 (round 6:
        # MoondreamModel.pair_decode): warm both groups (graph capture of the paired decode at either slot base) before timing
        run_steps(args.warmup if args.no_pipeline else max(4, args.warmup))
    engine.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run_steps(args.steps)
    torch.cuda.synchronize()
    engine.barrier()
    elapsed = time.perf_counter() - t0
    _lib.check(lib.md_gemm_set_tuning(b"w4_grid", 0))
    per_rank_ms = engine.gather_floats(elapsed / args.steps * 1e3)  # every rank's own clock, on rank 0
    elapsed = engine.max_over_ranks(elapsed)
    ranks_seen = engine.ranks_seen()  # an RCCL all-reduce of ones: the ranks that really took part
    # parity of the timed configuration: the LAST timed step's ids (gathered on rank 0, image order)
    parity = None
    if rank == 0 and out and out[-1] is not None and not args.only_timed_steps:
        ids_all = torch.cat([b.cpu() for b in out[-1]], 0).tolist()
        second = None
        if world == 1 and not args.no_second_oracle and args.model == "2b" and args.seed == 1 and args.prompt == "caption":
            second = second_oracle(cfg, sd, args.seed, T, dev)
        parity = check_parity(model, images, prompts, ids_all[: len(images)], args.model, args.seed, args.prompt, T, floor_from=second)
        if second is not None:
            parity["parity_second_oracle"] = second
            parity["parity_second_oracle_exact"] = second["exact"]
            parity["parity_min_exact"] = exact_floor(len(images), T, second)
            parity["parity_second_oracle_sane"] = bool(second["exact"] >= SECOND_ORACLE_SANITY)
            if not parity["parity_second_oracle_sane"]:   # the calibration itself is broken: say so, the static floor stands
                parity["parity_ok"] = False
                parity["parity_note"] = (parity.get("parity_note") or "") + (
                    f"; SECOND ORACLE below its sanity bound ({second['exact']} < {SECOND_ORACLE_SANITY} of 64): calibration rejected")

    if args.only_timed_steps:
        if rank == 0:
            print(json.dumps({"only_timed_steps": args.steps, "ms_per_step": elapsed / args.steps * 1e3}), flush=True)
        return
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
    alg_rd, alg_wr = C.c_double(), C.c_double()
    _lib.check(lib.md_profile_gemm_bytes(0, C.byref(alg_rd), C.byref(alg_wr)))
    by, ms1, n1 = C.c_double(), C.c_double(), C.c_int64()
    _lib.check(lib.md_profile_gemm_read(1, C.byref(by), C.byref(ms1), C.byref(n1)))
    lib.md_profile_gemm(0)
    stream_gbs = by.value / (ms1.value * 1e-3) / 1e9 if ms1.value > 0 else 0.0

    if rank != 0:
        return
    # Fabric-side bytes of the tile GEMMs of ONE B=64 step, from the committed PMC passes over this command's eager
    # step (tools/gpu_pmc_traffic.sh: separate FETCH_SIZE / WRITE_SIZE runs of `bench.py --steps 1 --tokens 1`;
    # FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md; WRITE_SIZE uncalibrated; Infinity-Cache hits
    # are counted, so this is L2-miss traffic, an upper bound on HBM bytes).  bench.py cannot collect counters
    # on itself: the algorithmic bytes beside it are live (this run's launches), the counter figure is the
    # committed pass of the same command.
    traffic = {"algorithmic_read_bytes_per_step": alg_rd.value, "algorithmic_written_bytes_per_step": alg_wr.value,
               "measured": None, "note": "no profiles/r0N_pmc_traffic.json"}
    try:
        pmc_file = next(n for n in ("r06_pmc_traffic.json", "r05_pmc_traffic.json") if os.path.isfile(os.path.join(REPO, "profiles", n)))
        with open(os.path.join(REPO, "profiles", pmc_file)) as f:
            tg = json.load(f)["tile_gemm"]
        rd, wr = 2.0 * tg["FETCH_SIZE_kb_sum"] * 1024.0, tg["WRITE_SIZE_kb_sum"] * 1024.0
        traffic.update({
            "measured": {"read_bytes_per_step": rd, "written_bytes_per_step": wr, "launches": tg["launches"]},
            "read_ratio": rd / alg_rd.value if alg_rd.value else None,
            "write_ratio": wr / alg_wr.value if alg_wr.value else None,
            "note": f"profiles/{pmc_file}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over one eager B=64 step "
                    "(2 x FETCH_SIZE per the gfx950 note; L2-miss side, Infinity-Cache hits included)",
        })
    except (OSError, KeyError, ValueError, ZeroDivisionError, StopIteration):
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
        "ranks_seen": ranks_seen,
        "weights_broadcast": weights_broadcast,
        "cpu_binding": engine.cpu_binding,   # rank 0's share of the host's cores (dist.bind_rank_cpus; every rank binds its own)
        "per_rank_ms_per_step": [round(x, 3) for x in per_rank_ms] if per_rank_ms else None,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": f"Moondream-{args.model.upper()} bf16 batch_generate ({args.prompt}): {B} images/GPU x 378x378 "
                        f"(2 crops each, both encoded), {len(prompt)}-token prompt, {T} greedy decode tokens, "
                        f"seeded synthetic weights",
            "batch_per_gpu": B, "decode_tokens": T, "parallelism": f"dp{world}",
            # two HIP streams (encode of step i+1, decode of step i).  Their kernels do overlap on the GPU, and that is worth
            # 3 % of a step against one in-order stream (same box: 260.5 vs 269 ms, profiles/r04_pipelined_engine_streams_ab.txt);
            # the rest of what the pipelining buys (no-pipeline: 292.8 ms) is host work hidden: tiling, launches, D2H
            "host_latency_hiding": "none" if args.no_pipeline else "step i+1's host tiling (thread pool) + encode launches run while step i "
                                   "decodes on a second, higher-priority HIP stream; step i's ids are collected (pinned, async D2H) after "
                                   "step i+1 is queued; kernels of the two streams overlap on the GPU (3 % of a step vs one in-order stream)"
                                   + ("; the decode of two consecutive steps runs as ONE lockstep of 128 sequences (each step is still "
                                      "encoded on its own, 64 images per launch; md_decode_step takes 65..128 rows in one pass over the "
                                      "weights, bit-identical to two passes of 64)" if model.pair_decode and 2 * args.batch <= 128 else ""),
            # every row of [bos | 729 image embeddings | prompt] goes through every decoder block, as in the reference;
            # the reference does it as two passes over the weights (encode_image, then the prompt)
            "prefill": "image prefix + prompt in one decoder pass" if model.fused_prefill else "image prefix, then prompt (two passes)",
        },
        "roofline": {
            "bound": "mfma",
            "kernel": "gemm_w4_kernel: 256x256 tile GEMM, four waves (one per SIMD) x 128x128 on v_mfma_f32_16x16x32_bf16, operands global -> LDS by LDS-DMA, persistent; "
                      "small shapes on gemm_bf16_kernel<256x128 | 128x128>",
            "achieved": gemm_tflops,
            "peak": 2500.0,
            "unit": "TFLOP/s",
            "frac": gemm_tflops / 2500.0,
            "traffic": traffic,
            "launches": int(n.value),
            "share_of_step": (ms.value * 1e-3) / step_gpu_s if step_gpu_s > 0 else None,
            "mfma_only_ceiling": {"tflops": 2070.0, "frac_of_it": gemm_tflops / 2070.0,
                                  "note": "profiles/r04_mfma_power_probe.txt: a kernel of nothing but v_mfma_f32_16x16x32_bf16 on random register "
                                          "operands sustains 2.07 PF/s on this chip (power-limited clock; 1.83 PF/s with 32x32x16); `frac` above stays "
                                          "against the nominal 2.5 PF/s"},
            "measured_on": "one non-overlapped, eagerly launched step after the timed region",
        },
        "decode_gemm": {
            "bound": "hbm", "kernel": "gemm_bf16_kernel / gemm_pair_kernel <64,64> decode-regime configs (m <= 64 weight stream; proj+fc2 as K-slice partials)",
            "achieved": stream_gbs if n1.value else None, "peak": 8000.0, "unit": "GB/s",
            "frac": stream_gbs / 8000.0 if n1.value else None, "launches": int(n1.value),
            "share_of_step": (ms1.value * 1e-3) / step_gpu_s if step_gpu_s > 0 and n1.value else None,
            # NOT a measurement of this run (advisor, round 5): parsed from a committed rocprofv3 kernel trace of the default bf16
            # configuration, kept apart from the live figures and dropped as soon as a flag changes what runs
            "committed_profile": decode_gemm_trace_figure(args),
        },
        "phase_ms": phase_ms,
    }
    if parity is not None:
        result.update(parity)
    if args.model == "2b" and phase_ms.get("vision"):
        # the north star's "ViT encoder vs MFMA roofline": whole vision phase (patchify, 27 blocks incl.
        # attention and layer norms, projector) against the algorithmic FLOPs of SURVEY 8d
        vit_flops = len(images) * (2 * FLOP_VIT_PER_CROP + 51.98e9)
        result["vit_encoder"] = {
            "bound": "mfma", "achieved": vit_flops / (phase_ms["vision"] * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
            "frac": vit_flops / (phase_ms["vision"] * 1e-3) / 1e12 / 2500.0,
            "note": "all kernels of the vision phase (H2D of the crops, patchify, 27 blocks incl. attention and layer norms, stitch, "
                    "projector), 2 crops/image (666.45 GFLOP each) + projector (51.98 GFLOP/image); the host-side tiling the eager "
                    "step waits for first is phase_ms.host_tiling (hidden behind the previous step's decode in the timed region)",
            "activation_tolerance": "ViT outputs vs the reference's: 1.25e-2 rel-RMS in the -m gpu tests (measured 1.07e-2; per-layer profile on the tiny model within 1.3x of the oracle's drift at every block); the north "
                                    "star's 1e-3 is below what bf16 allows: against an fp64 evaluation of the same weights the reference "
                                    "itself sits at 9.633e-3 rel-RMS and this path at 9.616e-3 (test_vit_error_against_fp64_truth_no_worse_than_reference)",
        }

    if args.model == "2b" and phase_ms.get("decode"):
        # SURVEY 8d's decode roofline: a step reads every decoder + lm_head weight once (2.627 GB) and, per sequence, the
        # K / V rows of its context (196 608 B per position: 24 layers x 32 heads x 64 x 2 x 2 B); step s attends to
        # pos + s + 1 keys.  Over the decode phase of the eager step (all kernels of the phase, launch gaps included).
        p1 = 730 + len(prompt)
        dec_bytes = sum(2.627e9 + B * (p1 + s_ + 1) * 196608.0 for s_ in range(T))
        dec_gbs = dec_bytes / (phase_ms["decode"] * 1e-3) / 1e9
        result["decode_step"] = {
            "bound": "hbm", "achieved": dec_gbs, "peak": 8000.0, "unit": "GB/s", "frac": dec_gbs / 8000.0,
            "ms_per_token": phase_ms["decode"] / T,
            "note": f"{T} lockstep decode steps of {B} sequences: weights 2.627 GB per step + K/V 196608 B per sequence and position "
                    f"(contexts {p1 + 1}..{p1 + T}); time = the decode phase of one eager step (hipGraph replay in the timed region)",
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
        out_vqa = run_steps(2, vqa_prompts)
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
        # parity of THIS configuration against the reference's ids for the same 64 images x 32-id prompts
        # (tests/golden/md2b_vqa64.npz, oracle/make_golden.py vqa64: the unmodified reference's _generate_answer, the loop behind
        # query(), moondream.py:541-618), with the measured licence and the second oracle's calibrated floor like the headline's
        if out_vqa and out_vqa[-1] is not None and not args.only_timed_steps:
            ids_vqa = torch.cat([b.cpu() for b in out_vqa[-1]], 0).tolist()
            second_v = None
            if not args.no_second_oracle and args.model == "2b" and args.seed == 1:
                second_v = second_oracle(cfg, sd, args.seed, T, dev, fixture="md2b_vqa64")
            pv = check_parity(model, images, vqa_prompts, ids_vqa[: len(images)], args.model, args.seed, "vqa32", T, floor_from=second_v)
            if second_v is not None:
                pv["parity_second_oracle_exact"] = second_v["exact"]
                pv["parity_second_oracle_max_logit_err"] = second_v["max_logit_err"]
                pv["parity_min_exact"] = exact_floor(len(images), T, second_v)
            if pv.get("parity_note"):
                pv["parity_note"] = pv["parity_note"].replace("md2b_bench64.npz", "md2b_vqa64.npz")
            result["vqa32"].update(pv)

    # auxiliary leg: the SURVEY 8(b) contract "batch_generate == a loop of caption()" priced.  In strict mode every sequence
    # gets the same bits whatever batch it travels in (two-pass prefill, short prompt passes at <= 64 rows per launch, no
    # persistent single-sequence kernel); the default mode keeps the shortcuts and agrees within bf16 accumulation-order noise.
    if world == 1 and not args.no_strict_leg and args.prompt == "caption":
        # the DEFAULT mode's own contract first: the timed batch == each image alone on the batched kernels, bit for bit
        ids_d = torch.cat([b.cpu() for b in out[-1]], 0).tolist() if out and out[-1] is not None else None
        n_seq = min(8, len(images))
        if ids_d is not None:
            model.single_sequence_kernel = False
            try:
                seq = [model.batch_generate_ids([images[i]], [prompts[i]], max_tokens=T, ignore_eos=True)[0] for i in range(n_seq)]
            finally:
                model.single_sequence_kernel = True
            result["batch_equals_sequential_default_mode"] = {
                "checked": n_seq, "identical": sum(seq[i][:T] == ids_d[i][:T] for i in range(n_seq)),
                "note": "the timed step's ids vs batch_generate_ids([x_i]) with single_sequence_kernel = False (a lone sequence on the batched "
                        "kernels); all 64 in tests/test_model_gpu.py.  The lone sequence's default LATENCY path (persistent kernel) is outside it",
            }
        model.set_strict_batch_invariance(True)
        try:
            run_steps(2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            outs = run_steps(3)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 3
            ids_s = torch.cat([b.cpu() for b in outs[-1]], 0).tolist()
            seq = [model.batch_generate_ids([images[i]], [prompts[i]], max_tokens=T, ignore_eos=True)[0] for i in range(n_seq)]
            result["strict_batch_invariance"] = {
                "images_per_sec": B / dt, "ms_per_step": dt * 1e3, "cost_vs_value": 1.0 - (B / dt) / (n_total * args.steps / elapsed),
                "sequences_identical_to_default_mode": sum(a == b for a, b in zip(ids_s, ids_d)) if ids_d else None, "of": B,
                "batch_equals_sequential": {"checked": n_seq, "identical": sum(seq[i][:T] == ids_s[i][:T] for i in range(n_seq))},
                "note": "MoondreamModel.set_strict_batch_invariance(True): two decoder passes (image prefix, then prompt) like caption() on an "
                        "EncodedImage, so batch_generate_ids(B=64)[i] == caption(x_i) bit for bit "
                        "(all 64 checked in tests/test_model_gpu.py; the first 8 here); never `value`",
            }
        finally:
            model.set_strict_batch_invariance(False)

    # auxiliary leg, NOT the headline: the reference's OWN quantised checkpoint format (layers.py:47-109, QuantizedLinear: 4-bit
    # groups of 128) as the decode regime's weight stream.  The synthetic decoder blocks are quantised into that format, the
    # model is loaded from the triples (prefill multiplies with the dequantised bf16 copy, as the reference's dequantize_tensor
    # produces it) and the SAME model is timed with its decode launches streaming the nibbles vs the bf16 copy: same weights
    # bit for bit, a quarter of the bytes.  Eager, non-pipelined steps (per-phase GPU times), ids compared.
    if world == 1 and args.int4_leg and args.model == "2b":
        result["int4_decode"] = int4_leg(cfg, sd, images, prompts, T, args, dev)

    # BASELINE configs[4]'s workload shape (multi-crop + detect head) as its own leg
    if world == 1 and not args.no_detect13_leg and args.model == "2b":
        result["detect13"] = detect13_leg(model, cfg, args, dev)
        if not args.no_fp8_full_leg:
            try:
                result["detect13_fp8"] = detect13_leg(model, cfg, args, dev, fp8=True)
            finally:
                model.enable_fp8(on=False)

    # auxiliary leg, NOT the headline: identical crops of an image encoded once (the bench's 378 x 378 images have
    # tiling (1, 1): their local crop is byte-identical to the global crop).  The reference encodes both, so `value`
    # above does too; this leg only shows what the opt-in saves, with the ids checked equal to the timed step's.
    if world == 1 and not args.no_dedup_leg:
        model.dedup_identical_crops = True
        try:
            run_steps(2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            outd = run_steps(2)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 2
            same = None
            if outd and outd[-1] is not None and out and out[-1] is not None:
                idsd = torch.cat([b.cpu() for b in outd[-1]], 0).tolist()
                ids16 = torch.cat([b.cpu() for b in out[-1]], 0).tolist()
                same = sum(a == b for a, b in zip(idsd, ids16))
            result["dedup_identical_crops"] = {
                "images_per_sec": B / dt, "ms_per_step": dt * 1e3, "sequences_identical_to_timed_step": same, "of": B,
                "note": "opt-in (MoondreamModel.dedup_identical_crops): 1 ViT pass per image instead of 2 for images that fit one crop; "
                        "bit-identical embeddings (tests/test_model_gpu.py); not the reference's work per image, hence not `value`",
            }
        finally:
            model.dedup_identical_crops = False

    # auxiliary leg, NOT the headline: the opt-in FP8 weight stream for the decode steps (BASELINE configs[4]); a
    # different numerical mode (tolerance-judged in tests/test_model_gpu.py), so it never feeds `value`
    if world == 1 and not args.no_fp8_leg:
        model.enable_fp8_decode(True)
        try:
            run_steps(2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            out8 = run_steps(2)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 2
            lat = []
            for i in range(args.latency_runs + 1):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                model.batch_generate_ids(one, [prompt], max_tokens=T, ignore_eos=True)
                torch.cuda.synchronize()
                if i > 0:
                    lat.append(time.perf_counter() - t1)
            # per-phase GPU time of one eager, non-overlapped step in this mode (compare with phase_ms above)
            graphs_were = model.use_graphs
            model.use_graphs = False
            model.collect_timing = True
            model.batch_generate_ids(images, prompts, max_tokens=T, ignore_eos=True)
            torch.cuda.synchronize()
            model.collect_timing = False
            model.use_graphs = graphs_were
            phase8 = {k: round(v, 2) for k, v in model.last_phase_ms.items()}
            ids8 = torch.cat([b.cpu() for b in out8[-1]], 0).tolist() if out8 and out8[-1] is not None else None
            same = None
            if ids8 is not None and out and out[-1] is not None:
                ids16 = torch.cat([b.cpu() for b in out[-1]], 0).tolist()
                same = sum(a == b for a, b in zip(ids8, ids16))
            result["fp8_decode"] = {
                "images_per_sec": B / dt, "ms_per_step": dt * 1e3,
                "p50_caption_latency_ms": float(np.median(lat) * 1e3) if lat else None,
                "sequences_identical_to_bf16": same, "of": B, "phase_ms": phase8,
                "note": "decode launches (<= 64 rows) stream e4m3 weights with per-channel scales, bf16 activations, fp32 accumulation; "
                        "prefill, vision and KV cache unchanged",
            }
        finally:
            model.enable_fp8_decode(False)

    # auxiliary leg, NOT the headline: the full opt-in FP8 mode (BASELINE configs[4] "fp8 weights, CDNA4 fp8 MFMA") -- every
    # MFMA-bound linear of the ViT blocks, the projector and the prefill on md_gemm_f8 (e4m3 operands, static per-tensor
    # activation scales calibrated on 8 of these images), plus the fp8 decode weight stream.  A different numerical mode
    # (tolerance-judged in tests/test_model_gpu.py), so it never feeds `value`; its ids are compared with this run's bf16 ids.
    if world == 1 and not args.no_fp8_full_leg:
        info = model.enable_fp8(images[:8], prompt)
        try:
            run_steps(2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            out8 = run_steps(3)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 3
            graphs_were = model.use_graphs
            model.use_graphs = False
            model.collect_timing = True
            model.batch_generate_ids(images, prompts, max_tokens=T, ignore_eos=True)
            torch.cuda.synchronize()
            model.collect_timing = False
            model.use_graphs = graphs_were
            phase8 = {k: round(v, 2) for k, v in model.last_phase_ms.items()}
            ids8 = torch.cat([b.cpu() for b in out8[-1]], 0).tolist() if out8 and out8[-1] is not None else None
            same, prefix = None, None
            if ids8 is not None and out and out[-1] is not None:
                ids16 = torch.cat([b.cpu() for b in out[-1]], 0).tolist()
                same = sum(a == b for a, b in zip(ids8, ids16))
                prefix = float(np.mean([next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), len(b)) for a, b in zip(ids8, ids16)]))
            leg8 = {
                "images_per_sec": B / dt, "ms_per_step": dt * 1e3, "speedup_vs_bf16_value": (B / dt) / (n_total * args.steps / elapsed),
                "phase_ms": phase8, "sequences_identical_to_bf16": same, "mean_matching_prefix_tokens": prefix, "of": B,
                "kv_cache_fp8": bool(info.get("kv_cache_fp8")),
                "calibration": {"images": 8, "margin": info["margin"], "vit_amax_max": max(info["vit_amax"]), "text_amax_max": max(info["text_amax"])},
                "note": "md_gemm_f8: OCP e4m3 operands on v_mfma_f32_32x32x64_f8f6f4, fp32 accumulation, per-channel weight scales, one "
                        "static scale per activation tensor; LN -> fp8, GELU epilogue -> fp8, attention epilogue -> fp8 (md_attn_args.o8); "
                        "RoPE + K/V (bf16 and e4m3) written by the qkv GEMM epilogue; patch embedding, prefill attention, residual stream and lm_head at prefill stay bf16; decode "
                        "steps stream e4m3 weights (the fp8_decode leg's mode) and attend over an e4m3 copy of the KV cache (one static "
                        "scale per layer for K and V; the bf16 slabs stay for prefill attention and EncodedImage snapshots)",
            }
            if args.model == "2b" and phase8.get("vision"):
                vit_flops = len(images) * (2 * FLOP_VIT_PER_CROP + 51.98e9)
                leg8["vit_encoder"] = {"achieved": vit_flops / (phase8["vision"] * 1e-3) / 1e12, "unit": "TFLOP/s",
                                       "frac_of_bf16_peak_2500": vit_flops / (phase8["vision"] * 1e-3) / 1e12 / 2500.0,
                                       "frac_of_fp8_peak_5000": vit_flops / (phase8["vision"] * 1e-3) / 1e12 / 5000.0}
            gpath64 = os.path.join(REPO, "tests", "golden", "md2b_bench64.npz")
            if args.model == "2b" and args.seed == 1 and args.prompt == "caption" and os.path.exists(gpath64) and not args.only_timed_steps:
                from moondream_amd import parity as P
                g64 = np.load(gpath64)
                n64 = min(len(images), g64["tokens"].shape[0])
                tf8 = model.teacher_forced_logits(images[:n64], prompts[:n64], g64["tokens"][:n64, :T], g64["top8_idx"][:n64, : T + 1]).numpy()
                leg8["accuracy_vs_reference"] = P.fp8_contract_report(tf8, g64["top8_val"][:n64, : T + 1])
            result["fp8_full"] = leg8
        finally:
            model.enable_fp8(on=False)

    if world == 1 and not args.no_cpu_baseline:
        est, cores, note, cpu_details = cpu_baseline(cfg, sd, args.seed, T)
        try:
            host_cores = len(os.sched_getaffinity(0))
        except AttributeError:
            host_cores = os.cpu_count() or 1
        result["cpu_baseline"] = {
            "value": est, "unit": "images/s", "cores": cores, "host_cores": host_cores, "kind": "port",
            "cores_note": "cores = the torch thread count of the encode (picked on whole encode_image runs: all logical cores is slower "
                          "on a many-core host; the decode step has its own count, in `details`); host_cores = the logical cores this "
                          "process may run on",
            "details": cpu_details,
            "sample": f"oracle in fast mode = the reference's own ATen calls (bf16 F.linear / SDPA over all 2048 slots), {note}",
        }
        # How far the port's clock is from the unmodified reference's (SURVEY 8d asks for the latter; /root/reference does not
        # exist on the GPU box): both were timed back to back in the build container by `oracle/make_golden.py reftime`
        side = os.path.join(REPO, "profiles", "r05_reference_vs_port_cpu_timing_build_container.json")
        if os.path.exists(side):
            with open(side) as f:
                sj = json.load(f)
            result["cpu_baseline"]["port_vs_reference"] = {
                "ratio": sj.get("port_vs_reference"), "reference_images_per_sec": sj.get("images_per_sec"),
                "port_images_per_sec": (sj.get("port") or {}).get("images_per_sec"), "host_cores": (sj.get("host") or {}).get("cores"),
                "where": "build container, same process, back to back (profiles/r05_reference_vs_port_cpu_timing_build_container.json)",
                "reference_equivalent_value": (est / sj["port_vs_reference"]) if sj.get("port_vs_reference") else None,
            }
        # ... and where the reference checkout IS present (the build container, or MOONDREAM_REFERENCE set), it is timed itself
        ref_root = os.environ.get("MOONDREAM_REFERENCE", "/root/reference")
        if os.path.isdir(os.path.join(ref_root, "moondream", "torch")):
            try:
                det = result["cpu_baseline"].get("details") or {}
                thr = (det["encode_threads"], det["decode_threads"]) if "encode_threads" in det and "decode_threads" in det else None
                rt = reference_cpu_timing(cfg, sd, args.seed, T, threads=thr)
                result["cpu_baseline"].update({"kind": "reference", "value": rt["images_per_sec"], "cores": rt["threads"], "port_value": est,
                                               "sample": rt["sample"]})
            except Exception as e:  # a checkout that does not import (missing dependency): the port's figure stands, loudly
                result["cpu_baseline"]["reference_timing_error"] = repr(e)
    print(json.dumps(result), flush=True)
    if result.get("vqa32", {}).get("parity_ok") is False:
        print("bench.py: PARITY VIOLATION in the vqa32 leg -- " + str(result["vqa32"].get("parity_note")), file=sys.stderr, flush=True)
        sys.exit(3)
    if parity is not None and parity.get("parity_ok") is False:
        print("bench.py: generated ids disagree with the reference beyond bf16-noise margins: " + parity["parity_note"], file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
