"""Seeded synthetic Llama checkpoints (no hub / no weights on this box).

``synth_state(cfg, seed)`` yields HF-named tensors.  Small models are drawn with
numpy's RandomState (stable across library versions, used by the golden
fixtures); billion-parameter shapes are drawn tensor-by-tensor with a torch
generator so they can be produced directly at the destination dtype.
AWQ tensors follow the AutoAWQ GEMM format the reference consumes
(umbrella/quantization/awq_utils.py:20-27): qweight [K, N/8] int32,
qzeros [K/G, N/8] int32, scales [K/G, N] fp16.
"""
from __future__ import annotations

import numpy as np
import torch

from .config import LlamaCfg

LINEARS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
           "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")


def linear_shapes(cfg: LlamaCfg) -> dict:
    """name -> (out_features N, in_features K)"""
    H, I = cfg.hidden_size, cfg.intermediate_size
    return {"self_attn.q_proj": (cfg.q_dim, H), "self_attn.k_proj": (cfg.kv_dim, H),
            "self_attn.v_proj": (cfg.kv_dim, H), "self_attn.o_proj": (H, cfg.q_dim),
            "mlp.gate_proj": (I, H), "mlp.up_proj": (I, H), "mlp.down_proj": (H, I)}


def synth_state_small(cfg: LlamaCfg, seed: int, std: float = 0.06) -> dict:
    """fp32 numpy-seeded dense state dict (tiny models; fixtures depend on it)."""
    rs = np.random.RandomState(seed)
    sd = {}

    def nrm(*shape, s=std):
        return torch.from_numpy((rs.standard_normal(shape) * s).astype(np.float32))

    sd["model.embed_tokens.weight"] = nrm(cfg.vocab_size, cfg.hidden_size, s=1.0)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        for name, (n, k) in linear_shapes(cfg).items():
            sd[p + name + ".weight"] = nrm(n, k)
        sd[p + "input_layernorm.weight"] = 1.0 + nrm(cfg.hidden_size, s=0.1)
        sd[p + "post_attention_layernorm.weight"] = 1.0 + nrm(cfg.hidden_size, s=0.1)
    sd["model.norm.weight"] = 1.0 + nrm(cfg.hidden_size, s=0.1)
    if not cfg.tie_word_embeddings:
        sd["lm_head.weight"] = nrm(cfg.vocab_size, cfg.hidden_size, s=0.3)
    if getattr(cfg, "attention_bias", False):          # drawn last: bias-free fixtures keep their stream
        for i in range(cfg.num_hidden_layers):
            for name in ("q_proj", "k_proj", "v_proj"):
                n = linear_shapes(cfg)["self_attn." + name][0]
                sd[f"model.layers.{i}.self_attn.{name}.bias"] = nrm(n, s=0.5)
    return sd


def awq_quantize_small(w: torch.Tensor, group: int = 128, seed: int = 0):
    """Round-to-nearest asymmetric 4-bit quantisation of a dense [N, K] weight
    into AutoAWQ GEMM tensors (numpy; tiny models / tests only).  Returns
    (qweight, qzeros, scales) and is exact w.r.t. its own dequantisation."""
    from .awq_format import pack_rows
    wt = w.t().contiguous().float().numpy()                     # [K, N]
    K, N = wt.shape
    g = wt.reshape(K // group, group, N)
    mx, mn = g.max(1), g.min(1)
    scale = np.maximum((mx - mn) / 15.0, 1e-5).astype(np.float16)
    zero = np.clip(np.round(-mn / scale.astype(np.float32)), 0, 15).astype(np.uint8)
    q = np.clip(np.round(g / scale.astype(np.float32)[:, None, :]) + zero[:, None, :], 0, 15).astype(np.uint8)
    return (torch.from_numpy(pack_rows(q.reshape(K, N))), torch.from_numpy(pack_rows(zero)),
            torch.from_numpy(scale))


def synth_awq_small(cfg: LlamaCfg, seed: int) -> dict:
    """Tiny AWQ checkpoint: dense synth weights quantised per linear."""
    sd = synth_state_small(cfg, seed)
    out = {}
    for k, v in sd.items():
        if k.endswith("_proj.weight"):
            qw, qz, sc = awq_quantize_small(v, cfg.awq_group)
            base = k[:-len(".weight")]
            out[base + ".qweight"], out[base + ".qzeros"], out[base + ".scales"] = qw, qz, sc
        else:
            out[k] = v
    return out


def synth_tensor(shape, std, dtype, device, gen: torch.Generator):
    t = torch.empty(shape, dtype=dtype, device=device)
    t.normal_(0.0, std, generator=gen)
    return t


def synth_awq_tensors(N: int, K: int, group: int, device, gen: torch.Generator, std: float = 0.02):
    """Full-size random AWQ linear straight on ``device``: uniform int4 codes
    and zero points, scales ~ std/4 so that (q - z) * s has std ~= std."""
    qweight = torch.randint(-2**31, 2**31 - 1, (K, N // 8), dtype=torch.int32, device=device, generator=gen)
    qzeros = torch.randint(-2**31, 2**31 - 1, (K // group, N // 8), dtype=torch.int32, device=device, generator=gen)
    scales = (torch.rand((K // group, N), device=device, generator=gen) * 0.5 + 0.75) * (std / 6.5)
    return qweight, qzeros, scales.to(torch.float16)
