#!/usr/bin/env python3
"""Driver for a kernel trace of ONE single-image caption (B = 1) after warm-up: tools/gpu_r5_b1_encode_trace.sh runs it under
rocprofv3 --kernel-trace and lists the kernels of the last caption's encode (vision + prefill) by time."""
import os, sys
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
for _ in range(4):
    model.batch_generate_ids(img, prompt, max_tokens=32, ignore_eos=True)
    torch.cuda.synchronize()
