#!/usr/bin/env python3
"""B=1 caption latency: GPU phase times (events) next to host wall time, plus a cProfile of the host side."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moondream_amd import synth
from moondream_amd.config import get_config
from moondream_amd.moondream import MoondreamModel, IdTokenizer

cfg = get_config("2b")
dev = torch.device("cuda", 0)
sd = synth.synthetic_state_dict(cfg, seed=1, device=dev)
model = MoondreamModel(cfg, sd, device=dev, tokenizer=IdTokenizer(), max_batch=1)
model.compile()
img = [synth.synthetic_image(0, 1)]
prompt = [cfg.tokenizer.templates["caption"]["normal"]]
for _ in range(3):
    model.batch_generate_ids(img, prompt, max_tokens=32, ignore_eos=True)
torch.cuda.synchronize()
model.collect_timing = True
for i in range(3):
    t0 = time.perf_counter()
    model.batch_generate_ids(img, prompt, max_tokens=32, ignore_eos=True)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    print(f"run {i}: wall {wall:.1f} ms, gpu phases {{" + ", ".join(f"{k}: {v:.1f}" for k, v in model.last_phase_ms.items()) + f"}} sum {sum(model.last_phase_ms.values()):.1f} ms", flush=True)
model.collect_timing = False
pr = cProfile.Profile()
pr.enable()
for i in range(5):
    model.batch_generate_ids(img, prompt, max_tokens=32, ignore_eos=True)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
