"""AutoModelLM (umbrella/models/auto_model.py:157-182): name -> runtime.

The reference keeps three hub-id dicts (plain / offload / cudagraph) of near-identical
classes; here one HIP-backed ``Llama`` runtime takes ``offload`` / ``cuda_graph`` flags and covers the reference's Llama,
Qwen (q/k/v projection bias, umbrella/models/qwen.py:94-96) and Mistral (free head_dim, mistral.py:28,101) classes.
Accepted names: the Llama, Qwen2.5 / QwQ and Mistral hub ids the reference registers (dims tabulated in
models/config.py -- weights are read from a local HF directory when given one, else
seeded synthetic tensors of the exact shapes), or a local directory with config.json.
"""
from __future__ import annotations

import os

from .config import KNOWN
from .llama import Llama


class AutoModelLM:
    _MODEL_MAPPING = {name: Llama for name in KNOWN}
    _OFFLOAD_MODEL_MAPPING = dict(_MODEL_MAPPING)
    _CUDAGRAPH_MODEL_MAPPING = dict(_MODEL_MAPPING)

    @classmethod
    def from_pretrained(cls, model_name, offload=False, cuda_graph=False, **kwargs):
        if cuda_graph:
            table, what = cls._CUDAGRAPH_MODEL_MAPPING, ""
        elif not offload:
            table, what = cls._MODEL_MAPPING, ""
        else:
            table, what = cls._OFFLOAD_MODEL_MAPPING, " (offload)"
        if model_name in table:
            klass = table[model_name]
        elif os.path.isdir(str(model_name)) and os.path.exists(os.path.join(model_name, "config.json")):
            klass = Llama
        else:
            raise ValueError(f"Model type '{model_name}' is not supported{what}. "
                             f"Supported{what} types: {list(table.keys())}")
        # cuda_graph wins over offload, as in the reference (auto_model.py:165-182)
        return klass(model_name=model_name, offload=bool(offload) and not cuda_graph, cuda_graph=bool(cuda_graph), **kwargs)
