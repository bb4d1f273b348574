"""The thin HF-style wrapper of the reference, on the native engine.

Mirrors ``HfMoondream`` (reference: moondream/torch/hf_moondream.py:37-183): lazy
KV-cache set-up, property pass-throughs to the model's API, ``answer_question``,
``batch_answer``, ``generate`` and the embedding accessors with the reference's
signatures.  The one behavioural difference is the point of this build:
``batch_answer`` runs the images in lockstep on the batched engine instead of a
sequential loop of sampled ``query`` calls (hf_moondream.py:99-103) and is greedy,
so its answers are those of ``query(..., settings={"temperature": 0})`` per pair.

This class is not a ``transformers.PreTrainedModel``: the reference subclasses it only
to be loadable through ``AutoModelForCausalLM(trust_remote_code=True)``, which is Hub
plumbing outside the hot path ("API SURFACE to keep").
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .config import MoondreamConfig
from .moondream import MoondreamModel


def extract_question(text: str) -> Optional[str]:
    """reference: hf_moondream.py:18-25."""
    prefix = "<image>\n\nQuestion: "
    suffix = "\n\nAnswer:"
    if text.startswith(prefix) and text.endswith(suffix):
        return text[len(prefix) : -len(suffix)]
    return None


class HfMoondream:
    def __init__(self, config: MoondreamConfig, state_dict, device="cuda", tokenizer=None, max_batch: int = 1):
        self.model = MoondreamModel(config, state_dict, device=device, setup_caches=False, tokenizer=tokenizer, max_batch=max_batch)
        self._is_kv_cache_setup = False
        self._max_batch = max_batch

    def _setup_caches(self):
        """reference: hf_moondream.py:46-49."""
        if not self._is_kv_cache_setup:
            self.model._setup_caches(self._max_batch)
            self._is_kv_cache_setup = True

    def _passthrough(name):  # noqa: N805
        def getter(self):
            self._setup_caches()
            return getattr(self.model, name)

        return property(getter)

    # reference: hf_moondream.py:51-81 (detect_gaze is post-processing outside the hot path: SURVEY 2.1 row 12)
    encode_image = _passthrough("encode_image")
    query = _passthrough("query")
    caption = _passthrough("caption")
    detect = _passthrough("detect")
    point = _passthrough("point")
    batch_generate = _passthrough("batch_generate")
    batch_caption = _passthrough("batch_caption")
    batch_query = _passthrough("batch_query")
    batch_detect = _passthrough("batch_detect")

    @property
    def device(self):
        return self.model.device

    def answer_question(self, image_embeds, question, tokenizer=None, chat_history="", result_queue=None,
                        max_new_tokens=256, **kwargs) -> str:
        """reference: hf_moondream.py:83-97 -- ``query(image, question)["answer"].strip()`` with the
        model's default sampling settings; ``tokenizer`` / ``chat_history`` / ``max_new_tokens`` are
        ignored there too.  Extension: ``settings=`` is forwarded to ``query`` (e.g. temperature 0)."""
        answer = self.query(image_embeds, question, settings=kwargs.get("settings"))["answer"].strip()
        if result_queue is not None:
            result_queue.put(answer)
        return answer

    def batch_answer(self, images: Sequence, prompts: Sequence[str], tokenizer=None, **kwargs) -> List[str]:
        """reference: hf_moondream.py:99-103 -- a sequential loop of ``query`` calls at the model's DEFAULT sampling settings
        (temperature 0.5, top_p 0.3: moondream.py:50-53).  Here: one lockstep decode over all pairs (questions of different
        token counts share it), every sequence sampling with the same rule and its own uniforms (round 6; rounds 1-5 forced
        greedy).  ``settings={"temperature": 0}`` is the greedy form; ``max_new_tokens`` bounds the answers."""
        self._setup_caches()
        settings = dict(kwargs.get("settings") or {})
        settings.setdefault("max_tokens", kwargs.get("max_new_tokens", kwargs.get("max_tokens", 256)))
        return [a.strip() for a in self.model.batch_query(list(images), list(prompts), settings)]

    def _unsupported_exception(self):
        """reference: hf_moondream.py:105-110."""
        raise NotImplementedError(
            "This method is not supported in the latest version of moondream. "
            "Consider upgrading to the updated API spec, or alternately pin to 'revision=2024-08-26'."
        )

    def generate(self, image_embeds, prompt, tokenizer=None, max_new_tokens=128, **kwargs) -> List[str]:
        """reference: hf_moondream.py:112-142.  A prompt in the legacy "<image>\\n\\nQuestion: ..\\n\\nAnswer:"
        form is a ``query``; anything else is tokenised and continued as is after the image prefix."""
        q = extract_question(prompt)
        if q is not None:
            return [self.model_query(image_embeds, q)]
        enc = self.encode_image(image_embeds)
        self.model.load_encoded_image(enc)
        ids = torch.tensor([self.model.tokenizer.encode(prompt).ids])
        settings = dict(kwargs.get("settings") or {})
        settings.setdefault("max_tokens", max_new_tokens)
        return ["".join(self.model._generate_answer(ids, enc.pos, settings))]

    def model_query(self, image, question: str) -> str:
        return self.query(image=image, question=question, stream=False)["answer"]

    # reference: hf_moondream.py:144-183 -- the embedding table behind an nn.Embedding view (shared storage)
    def get_input_embeddings(self) -> torch.nn.Embedding:
        if not hasattr(self, "_input_embeddings"):
            self._input_embeddings = torch.nn.Embedding.from_pretrained(self.model.w.wte, freeze=True)
        return self._input_embeddings

    def input_embeds(self, input_ids, *, device=None) -> torch.Tensor:
        """token ids [..] -> embeddings [.., D] through the native gather (text.py:12-13)."""
        if not torch.is_tensor(input_ids):
            input_ids = torch.as_tensor(input_ids)
        return self.model._embed(input_ids.cpu() if input_ids.device.type != "cpu" else input_ids)
