"""Small host helpers kept from the reference surface
(umbrella/speculation/speculation_utils.py:3-14, 316-358)."""
from __future__ import annotations

import re

import torch


def make_causal_mask(input_ids_shape, device):
    """bool [L, L], True on and below the diagonal."""
    _, n = input_ids_shape
    return torch.ones(n, n, dtype=torch.bool, device=device).tril()


def find_first_element_position(tensor: torch.Tensor, elements) -> int:
    wanted = set(int(e) for e in elements)
    for i, v in enumerate(tensor.flatten().tolist()):
        if v in wanted:
            return i
    return -1


def apply_repetition_penalty(input_ids, logits, penalty: float):
    g = torch.gather(logits, 1, input_ids)
    return logits.scatter(1, input_ids, torch.where(g < 0, g * penalty, g / penalty))


def apply_topk(logits, topk: int):
    k = min(topk, logits.size(-1))
    return logits.masked_fill(logits < torch.topk(logits, k)[0][..., -1, None], torch.finfo(logits.dtype).min)


_SENTENCE_END = re.compile(r"[.?!。？！]\s*$")


def is_sentence_complete_regex(text: str) -> bool:
    return bool(_SENTENCE_END.search(text))


class IdTokenizer:
    """Stand-in when no tokenizer files exist (offline, synthetic weights): text is
    space-separated token ids; a BOS id 0 is prepended on encode like HF tokenizers do."""

    def encode(self, text, return_tensors=None, **kw):
        ids = [0] + [int(t) for t in str(text).split()]
        return torch.tensor([ids], dtype=torch.long) if return_tensors == "pt" else ids

    def decode(self, ids, **kw):
        return " ".join(str(int(i)) for i in ids)
