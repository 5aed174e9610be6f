"""Llama architecture descriptors and RoPE frequency tables.

The reference resolves models by exact hub id through three dicts
(umbrella/models/auto_model.py:9-182) and reads dims from
``LlamaConfig.from_pretrained`` (umbrella/models/llama.py:24-33).  This box has
no hub access, so the public dims of the BASELINE models are tabulated here;
a local directory holding ``config.json`` is accepted as well.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass, field
from typing import Optional

import torch

LLAMA3_ROPE = {"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0,
               "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}
LLAMA31_ROPE = dict(LLAMA3_ROPE, factor=8.0)


@dataclass
class LlamaCfg:
    vocab_size: int = 128256
    hidden_size: int = 2048
    intermediate_size: int = 8192
    num_hidden_layers: int = 16
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    head_dim: int = 64
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_scaling: Optional[dict] = None
    max_position_embeddings: int = 131072
    tie_word_embeddings: bool = False
    eos_token_id: list = field(default_factory=lambda: [128001, 128008, 128009])
    awq: bool = False                 # 4-bit AWQ (GEMM format, group 128, zero point)
    awq_group: int = 128
    name: str = "llama"
    attention_bias: bool = False      # q/k/v projection bias (Qwen2: umbrella/models/qwen.py:94-96)
    embed_rows: int = 0               # rows of the embedding table when they differ from vocab_size (tensor-parallel shard:
                                      # the table stays whole while the lm_head holds vocab_size = V / P rows); 0 = vocab_size

    @property
    def q_dim(self):
        return self.num_attention_heads * self.head_dim

    @property
    def kv_dim(self):
        return self.num_key_value_heads * self.head_dim

    @classmethod
    def from_dir(cls, path: str) -> "LlamaCfg":
        with open(os.path.join(path, "config.json")) as f:
            c = json.load(f)
        rp = c.get("rope_parameters") or {}
        rs = c.get("rope_scaling") or ({k: v for k, v in rp.items() if k != "rope_theta"} if rp.get("rope_type", "default") != "default" else None)
        eos = c.get("eos_token_id", 2)
        gen = os.path.join(path, "generation_config.json")
        if os.path.exists(gen):                     # engines read eos from GenerationConfig (static:104-108)
            with open(gen) as f:
                eos = json.load(f).get("eos_token_id", eos)
        q = c.get("quantization_config") or {}
        qwen = c.get("model_type") == "qwen2"
        if qwen:                                    # the reference pins Qwen2.5's vocabulary (qwen.py:12,27)
            c["vocab_size"] = min(c["vocab_size"], 151936)
        return cls(attention_bias=bool(c.get("attention_bias", qwen)), vocab_size=c["vocab_size"], hidden_size=c["hidden_size"],
                   intermediate_size=c["intermediate_size"], num_hidden_layers=c["num_hidden_layers"],
                   num_attention_heads=c["num_attention_heads"],
                   num_key_value_heads=c.get("num_key_value_heads", c["num_attention_heads"]),
                   head_dim=c.get("head_dim") or c["hidden_size"] // c["num_attention_heads"],
                   rms_norm_eps=c.get("rms_norm_eps", 1e-6),
                   rope_theta=c.get("rope_theta") or rp.get("rope_theta", 10000.0), rope_scaling=rs,
                   max_position_embeddings=c.get("max_position_embeddings", 8192),
                   tie_word_embeddings=c.get("tie_word_embeddings", False),
                   eos_token_id=eos if isinstance(eos, list) else [eos],
                   awq=q.get("quant_method") == "awq", awq_group=q.get("group_size", 128),
                   name=os.path.basename(os.path.normpath(path)))


def _l(**kw):
    return LlamaCfg(**kw)


# Public HF configs of the hub ids the reference registers for Llama
# (auto_model.py:9-154) and that BASELINE.json's configs name.
KNOWN = {
    "meta-llama/Llama-3.2-1B-Instruct": _l(hidden_size=2048, intermediate_size=8192, num_hidden_layers=16,
        num_attention_heads=32, num_key_value_heads=8, head_dim=64, rope_scaling=LLAMA3_ROPE,
        tie_word_embeddings=True, name="llama-3.2-1b"),
    "meta-llama/Llama-3.1-8B-Instruct": _l(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
        num_attention_heads=32, num_key_value_heads=8, head_dim=128, rope_scaling=LLAMA31_ROPE, name="llama-3.1-8b"),
    "hugging-quants/Meta-Llama-3.1-8B-Instruct-AWQ-INT4": _l(hidden_size=4096, intermediate_size=14336,
        num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8, head_dim=128,
        rope_scaling=LLAMA31_ROPE, awq=True, name="llama-3.1-8b-awq"),
    "hugging-quants/Meta-Llama-3.1-70B-Instruct-AWQ-INT4": _l(hidden_size=8192, intermediate_size=28672,
        num_hidden_layers=80, num_attention_heads=64, num_key_value_heads=8, head_dim=128,
        rope_scaling=LLAMA31_ROPE, awq=True, name="llama-3.1-70b-awq"),
    "casperhansen/llama-3.3-70b-instruct-awq": _l(hidden_size=8192, intermediate_size=28672,
        num_hidden_layers=80, num_attention_heads=64, num_key_value_heads=8, head_dim=128,
        rope_scaling=LLAMA31_ROPE, awq=True, name="llama-3.3-70b-awq"),
}


def _qwen(H, I, L, Hq, Hkv, tie, awq=False, name="qwen2.5"):
    return LlamaCfg(vocab_size=151936, hidden_size=H, intermediate_size=I, num_hidden_layers=L, num_attention_heads=Hq,
                    num_key_value_heads=Hkv, head_dim=128 if H != 896 else 64, rms_norm_eps=1e-6, rope_theta=1000000.0,
                    rope_scaling=None, max_position_embeddings=32768, tie_word_embeddings=tie,
                    eos_token_id=[151645, 151643], awq=awq, attention_bias=True, name=name)


# Qwen2.5 / Mistral families the reference registers (auto_model.py:21-55, 80-154): public HF dims.
_QWEN_DIMS = {"0.5B": (896, 4864, 24, 14, 2, True), "1.5B": (1536, 8960, 28, 12, 2, True),
              "3B": (2048, 11008, 36, 16, 2, True), "7B": (3584, 18944, 28, 28, 4, False),
              "14B": (5120, 13824, 48, 40, 8, False), "32B": (5120, 27648, 64, 40, 8, False),
              "72B": (8192, 29568, 80, 64, 8, False)}
for _size, _d in _QWEN_DIMS.items():
    for _stem in ("Qwen/Qwen2.5-", "Qwen/Qwen2.5-Coder-"):
        KNOWN[f"{_stem}{_size}-Instruct"] = _qwen(*_d, name=f"qwen2.5-{_size.lower()}")
        KNOWN[f"{_stem}{_size}-Instruct-AWQ"] = _qwen(*_d, awq=True, name=f"qwen2.5-{_size.lower()}-awq")
KNOWN["Qwen/QwQ-32B-Preview"] = _qwen(*_QWEN_DIMS["32B"], name="qwq-32b")
KNOWN["KirillR/QwQ-32B-Preview-AWQ"] = _qwen(*_QWEN_DIMS["32B"], awq=True, name="qwq-32b-awq")
KNOWN["casperhansen/deepseek-r1-distill-qwen-32b-awq"] = _qwen(*_QWEN_DIMS["32B"], awq=True, name="r1-distill-qwen-32b-awq")


def _mistral(V, H, I, L, theta, awq=False, name="mistral"):
    return LlamaCfg(vocab_size=V, hidden_size=H, intermediate_size=I, num_hidden_layers=L, num_attention_heads=32,
                    num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-5, rope_theta=theta, rope_scaling=None,
                    max_position_embeddings=32768, tie_word_embeddings=False, eos_token_id=[2], awq=awq, name=name)


KNOWN["mistralai/Mistral-7B-Instruct-v0.3"] = _mistral(32768, 4096, 14336, 32, 1000000.0, name="mistral-7b-v0.3")
KNOWN["solidrust/Mistral-7B-Instruct-v0.3-AWQ"] = _mistral(32768, 4096, 14336, 32, 1000000.0, awq=True, name="mistral-7b-v0.3-awq")
# head_dim 128 with hidden 5120: attention width Hq*D = 4096 != hidden (mistral.py:28,101)
KNOWN["mistralai/Mistral-Small-24B-Instruct-2501"] = _mistral(131072, 5120, 32768, 40, 100000000.0, name="mistral-small-24b")
KNOWN["stelterlab/Mistral-Small-24B-Instruct-2501-AWQ"] = _mistral(131072, 5120, 32768, 40, 100000000.0, awq=True, name="mistral-small-24b-awq")
# further Llama-family ids of the reference registry (auto_model.py:9-20,58-79) with public dims
_L70 = dict(hidden_size=8192, intermediate_size=28672, num_hidden_layers=80, num_attention_heads=64, num_key_value_heads=8,
            head_dim=128)
_L8 = dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8,
           head_dim=128)
for _name in ("ibnzterrell/Meta-Llama-3.3-70B-Instruct-AWQ-INT4", "lambdalabs/Llama-3.3-70B-Instruct-AWQ-4bit",
              "casperhansen/deepseek-r1-distill-llama-70b-awq"):
    KNOWN[_name] = _l(**_L70, rope_scaling=LLAMA31_ROPE, awq=True, name="llama-70b-awq")
for _name in ("meta-llama/Llama-3.3-70B-Instruct", "meta-llama/Llama-3.1-70B-Instruct"):
    KNOWN[_name] = _l(**_L70, rope_scaling=LLAMA31_ROPE, name="llama-3.x-70b")
# Llama 3 (8k context, no rope scaling, eos = <|end_of_text|>, <|eot_id|>)
KNOWN["meta-llama/Meta-Llama-3-70B-Instruct"] = _l(**_L70, max_position_embeddings=8192, eos_token_id=[128001, 128009],
                                                    name="llama-3-70b")
KNOWN["meta-llama/Meta-Llama-3-8B-Instruct"] = _l(**_L8, max_position_embeddings=8192, eos_token_id=[128001, 128009],
                                                   name="llama-3-8b")
KNOWN["meta-llama/Llama-3.2-3B-Instruct"] = _l(hidden_size=3072, intermediate_size=8192, num_hidden_layers=28,
    num_attention_heads=24, num_key_value_heads=8, head_dim=128, rope_scaling=LLAMA3_ROPE, tie_word_embeddings=True,
    name="llama-3.2-3b")
KNOWN["facebook/layerskip-llama3.2-1B"] = KNOWN["meta-llama/Llama-3.2-1B-Instruct"]
KNOWN["Felladrin/Llama-68M-Chat-v1"] = _l(vocab_size=32000, hidden_size=768, intermediate_size=3072, num_hidden_layers=2,
    num_attention_heads=12, num_key_value_heads=12, head_dim=64, rms_norm_eps=1e-6, rope_theta=10000.0,
    max_position_embeddings=2048, eos_token_id=[2], name="llama-68m")
# The reference's small code drafters (Zhuominc/*, InfiniAILab/CodeDrafter-500M) publish no dims here: pass a local
# directory with config.json for those.
KNOWN["meta-llama/Llama-3.2-1B"] = KNOWN["meta-llama/Llama-3.2-1B-Instruct"]
KNOWN["meta-llama/Meta-Llama-3.1-8B-Instruct"] = KNOWN["meta-llama/Llama-3.1-8B-Instruct"]


def rope_inv_freq(cfg: LlamaCfg) -> tuple[torch.Tensor, float]:
    """(inv_freq [D/2] fp32, attention_scaling).  Restates HF's default and
    ``llama3`` rope initialisers, which is where the reference takes them from
    (``hf_model.model.rotary_emb.inv_freq``, umbrella/models/llama.py:48-49)."""
    D = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))
    rs = cfg.rope_scaling
    if rs and rs.get("rope_type", rs.get("type")) == "llama3":
        factor, lo, hi = rs["factor"], rs["low_freq_factor"], rs["high_freq_factor"]
        old = rs["original_max_position_embeddings"]
        low_wl, high_wl = old / lo, old / hi
        wl = 2 * math.pi / inv
        inv_l = torch.where(wl > low_wl, inv / factor, inv)
        smooth = (old / wl - lo) / (hi - lo)
        smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
        mid = ~(wl < high_wl) & ~(wl > low_wl)
        inv = torch.where(mid, smoothed, inv_l)
    return inv, 1.0


def rope_tables(cfg: LlamaCfg, max_length: int, dtype) -> tuple[torch.Tensor, torch.Tensor]:
    """cos/sin [Lmax, D] built in fp32 then cast to the model dtype
    (umbrella/models/llama.py:50-60)."""
    inv, scale = rope_inv_freq(cfg)
    freqs = torch.outer(torch.arange(max_length, dtype=torch.float32), inv)
    emb = torch.cat((freqs, freqs), dim=-1)
    return (emb.cos() * scale).to(dtype), (emb.sin() * scale).to(dtype)
