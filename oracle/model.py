"""CPU restatement of the reference Llama model runtime (oracle; test-only).

Follows umbrella/models/llama.py (Llama :11-142, LlamaAwq :222-322,
LlamaCudagraph :412-533), umbrella/models/llama_layer.py and
umbrella/attn/cache.py.  Weights are passed in as a dict of HF-named torch
tensors (``model.layers.{i}.self_attn.q_proj.weight`` ...); AWQ linears as
``<name>.qweight/.qzeros/.scales``.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops


class AppendKV:
    """umbrella/attn/cache.py:5-96 (KV_Cache): NHD cache, append at kv_offset
    (storage_ids' VALUES are ignored, only their length is used, :60-65)."""

    def __init__(self, L, Lmax, Hkv, D, dtype):
        self.k = torch.zeros(L, Lmax, Hkv, D, dtype=dtype)
        self.v = torch.zeros(L, Lmax, Hkv, D, dtype=dtype)
        self.kv_offset = 0

    def update(self, k_new, v_new, layer, storage_ids):
        n = storage_ids.shape[0]
        if layer == 0:
            self.kv_offset += n
        lo = self.kv_offset - n
        self.k[layer, lo:self.kv_offset] = k_new
        self.v[layer, lo:self.kv_offset] = v_new
        return self.k[layer, :self.kv_offset], self.v[layer, :self.kv_offset]

    def attend(self, q, k_new, v_new, layer, storage_ids, mask):
        k, v = self.update(k_new, v_new, layer, storage_ids)
        return ops.masked_attention(q, k, v, mask[:, :self.kv_offset])

    def gather_kv_incremental(self, indices, offset):        # cache.py:41-49
        a = len(indices)
        self.k[:, offset:offset + a] = self.k[:, indices]
        self.v[:, offset:offset + a] = self.v[:, indices]
        self.k[:, offset + a:] = 0.0
        self.v[:, offset + a:] = 0.0
        self.kv_offset = offset + a

    def clear(self):
        self.k.zero_(); self.v.zero_(); self.kv_offset = 0


class SlotKV:
    """umbrella/attn/cache.py:98-192 (StaticKV_Cache): writes at storage_ids,
    attends over all Lmax slots under the mask (graph-safe draft cache)."""

    def __init__(self, L, Lmax, Hkv, D, dtype):
        self.k = torch.zeros(L, Lmax, Hkv, D, dtype=dtype)   # kept NHD here; layout is not semantics
        self.v = torch.zeros(L, Lmax, Hkv, D, dtype=dtype)
        self.kv_offset = 0

    def attend(self, q, k_new, v_new, layer, storage_ids, mask):
        self.k[layer].index_copy_(0, storage_ids, k_new)
        self.v[layer].index_copy_(0, storage_ids, v_new)
        return ops.masked_attention(q, self.k[layer], self.v[layer], mask)

    def gather_kv_incremental(self, indices, offset):        # cache.py:136-144
        a = len(indices)
        self.k[:, offset:offset + a] = self.k[:, indices]
        self.v[:, offset:offset + a] = self.v[:, indices]
        self.k[:, offset + a:] = 0.0
        self.v[:, offset + a:] = 0.0
        self.kv_offset = offset + a

    def clear(self):
        self.k.zero_(); self.v.zero_(); self.kv_offset = 0


class OracleLlama:
    """cfg needs: vocab_size hidden_size intermediate_size num_hidden_layers
    num_attention_heads num_key_value_heads head_dim rms_norm_eps
    tie_word_embeddings.  ``slot_cache=True`` gives the LlamaCudagraph draft
    flavour (StaticKV semantics, optional exit_layer, llama.py:421,450-451)."""

    def __init__(self, cfg, weights: dict, inv_freq, attention_scaling=1.0, max_length=256,
                 dtype=torch.float32, slot_cache=False, exit_layer=-1, awq_group=128):
        self.config = cfg
        self.dtype = dtype
        self.max_length = max_length
        self.G = awq_group
        self.w = {}
        for k, v in weights.items():
            if v.dtype in (torch.int32,):
                self.w[k] = v
            elif k.endswith(".scales"):
                self.w[k] = v.to(torch.float16)
            else:
                self.w[k] = v.to(dtype)
        L = cfg.num_hidden_layers
        if slot_cache and exit_layer > 0:
            L = min(L, exit_layer)
        self.num_layers = L
        self.Hq, self.Hkv, self.D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        cache_cls = SlotKV if slot_cache else AppendKV
        self.kv_cache = cache_cls(L, max_length, self.Hkv, self.D, dtype)
        self.cos, self.sin = ops.rope_cache(inv_freq, attention_scaling, max_length, dtype)
        self.eps = cfg.rms_norm_eps

    # -- linear: dense F.linear (llama.py:89-91) or AwqLinear.apply (awq_utils.py:63-86)
    def _lin(self, x, name):
        if name + ".qweight" in self.w:
            out = ops.awq_linear(x, self.w[name + ".qweight"], self.w[name + ".qzeros"],
                                 self.w[name + ".scales"], self.G)
            b = self.w.get(name + ".bias")
            return out if b is None else out + b
        return F.linear(x, self.w[name + ".weight"], self.w.get(name + ".bias"))       # bias: qwen.py:94-96

    def _layer(self, i, h, position_ids, mask, storage_ids):     # llama.py:75-114 / 262-303
        p = f"model.layers.{i}."
        T = h.shape[0]
        res = h
        x = ops.rmsnorm(h, self.w[p + "input_layernorm.weight"], self.eps)
        q = self._lin(x, p + "self_attn.q_proj").view(T, self.Hq, self.D)
        k = self._lin(x, p + "self_attn.k_proj").view(T, self.Hkv, self.D)
        v = self._lin(x, p + "self_attn.v_proj").view(T, self.Hkv, self.D)
        q, k = ops.apply_rope(q, k, self.cos, self.sin, position_ids)
        a = self.kv_cache.attend(q, k, v, i, storage_ids, mask).reshape(T, self.Hq * self.D)
        h = res + self._lin(a, p + "self_attn.o_proj")
        res = h
        x = ops.rmsnorm(h, self.w[p + "post_attention_layernorm.weight"], self.eps)
        up = self._lin(x, p + "mlp.up_proj")
        gate = F.silu(self._lin(x, p + "mlp.gate_proj"))
        return res + self._lin(gate * up, p + "mlp.down_proj")

    @torch.no_grad()
    def inference(self, input_ids, position_ids, attention_mask, storage_ids):   # llama.py:117-134
        h = F.embedding(input_ids[0], self.w["model.embed_tokens.weight"])
        for i in range(self.num_layers):
            h = self._layer(i, h, position_ids[0], attention_mask, storage_ids)
        h = ops.rmsnorm(h, self.w["model.norm.weight"], self.eps)
        head = self.w["model.embed_tokens.weight"] if self.config.tie_word_embeddings else self.w["lm_head.weight"]
        return F.linear(h, head).float()[None]

    graph_inference = inference                                   # llama.py:522-533 falls back to inference

    def gather_kv_incremental(self, indices, offset):
        self.kv_cache.gather_kv_incremental(indices, offset)

    def clear(self):
        self.kv_cache.clear()
