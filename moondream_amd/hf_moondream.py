"""The thin HF-style wrapper of the reference, on the native engine.

Mirrors ``HfMoondream`` (reference: moondream/torch/hf_moondream.py:37-183): lazy
KV-cache set-up, property pass-throughs to the model's API, ``answer_question``
and ``batch_answer`` with the reference's signatures.  The one behavioural
difference is the point of this build: ``batch_answer`` runs the images in
lockstep on the batched engine instead of a sequential loop
(hf_moondream.py:99-103), returning the same answers.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

from .config import MoondreamConfig
from .moondream import MoondreamModel


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

    encode_image = _passthrough("encode_image")
    query = _passthrough("query")
    caption = _passthrough("caption")
    detect = _passthrough("detect")
    point = _passthrough("point")
    batch_generate = _passthrough("batch_generate")

    @property
    def device(self):
        return self.model.device

    def answer_question(self, image_embeds, question, tokenizer=None, chat_history="", result_queue=None,
                        max_new_tokens=256, **kwargs) -> str:
        """reference: hf_moondream.py:83-97."""
        answer = self.query(image_embeds, question)["answer"].strip()
        if result_queue is not None:
            result_queue.put(answer)
        return answer

    def batch_answer(self, images: Sequence, prompts: Sequence[str], tokenizer=None, **kwargs) -> List[str]:
        """reference: hf_moondream.py:99-103 (a sequential loop there).  Greedy, lockstep."""
        self._setup_caches()
        settings = {"max_tokens": kwargs.get("max_new_tokens", kwargs.get("max_tokens", 256))}
        return [a.strip() for a in self.model.batch_query(list(images), list(prompts), settings)]

    def _unsupported_exception(self):
        """reference: hf_moondream.py:105-110."""
        raise NotImplementedError(
            "This method is not supported in the latest version of moondream. "
            "Consider upgrading to the updated API spec, or alternately pin to 'revision=2024-08-26'."
        )
