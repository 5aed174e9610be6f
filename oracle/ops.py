"""Closed-form CPU statements of the L0 ops on the hot path (oracle; test-only).

Each function cites the reference call site (paths relative to /root/reference)
whose behaviour it restates.  Everything is plain torch on CPU tensors.
"""
from __future__ import annotations

import math

import numpy as np
import torch

# ----------------------------------------------------------------------------
# RMSNorm -- umbrella/models/model_utils.py:54-64 (-> flashinfer.rmsnorm)
# y = x * rsqrt(mean(x^2) + eps) * w, accumulated in fp32, one rounding to the
# input dtype at the end (flashinfer's published kernel).
# ----------------------------------------------------------------------------

def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    xf = x.float()
    var = xf.pow(2).mean(dim=-1, keepdim=True)
    return (xf * torch.rsqrt(var + eps) * w.float()).to(x.dtype)


# ----------------------------------------------------------------------------
# RoPE -- cache build umbrella/models/llama.py:48-60, application
# umbrella/models/model_utils.py:17-52 (HF rotate-half, arbitrary position_ids)
# ----------------------------------------------------------------------------

def rope_inv_freq(head_dim: int, rope_theta: float, rope_scaling: dict | None = None):
    """(inv_freq [D/2] fp32, attention_scaling).  The reference reads both off the HF model
    (``hf_model.model.rotary_emb.inv_freq`` / ``.attention_scaling``, umbrella/models/llama.py:48-49, 57-58); this is HF's
    published initialiser restated on its own (numpy, float64 until the last step) -- the ORACLE's copy: tests build the
    oracle model from this one and the product from umbrella_amd.models.config.rope_inv_freq, so a slip in either shows up
    as a parity failure instead of cancelling out.  Pinned to the reference by tests/golden/model_logits.npz `inv_freq`
    (recorded from the HF module, tests/test_oracle_golden.py).  default: theta^(-2i/D); "llama3": wavelengths beyond
    original_max / low_freq_factor are divided by `factor`, those below original_max / high_freq_factor kept, the band
    between interpolated linearly in original_max / wavelength."""
    D = int(head_dim)
    inv = 1.0 / np.power(float(rope_theta), np.arange(0, D, 2, dtype=np.float64) / D)
    inv = inv.astype(np.float32).astype(np.float64)           # HF builds the default frequencies in fp32 first
    rs = rope_scaling or {}
    if rs.get("rope_type", rs.get("type")) == "llama3":
        factor, lo, hi = float(rs["factor"]), float(rs["low_freq_factor"]), float(rs["high_freq_factor"])
        old = float(rs["original_max_position_embeddings"])
        out = np.empty_like(inv)
        for i, f in enumerate(inv):
            wl = 2.0 * math.pi / f
            if wl < old / hi:
                out[i] = f
            elif wl > old / lo:
                out[i] = f / factor
            else:
                t = (old / wl - lo) / (hi - lo)
                out[i] = (1.0 - t) * f / factor + t * f
        inv = out
    return torch.from_numpy(inv.astype(np.float32)), 1.0


def rope_cache(inv_freq: torch.Tensor, attention_scaling: float, max_length: int, dtype):
    pos = torch.arange(max_length, dtype=torch.float32)
    freqs = torch.outer(pos, inv_freq.float())              # [Lmax, D/2]
    emb = torch.cat((freqs, freqs), dim=-1)                   # [Lmax, D]
    cos = (emb.cos() * attention_scaling).to(dtype)
    sin = (emb.sin() * attention_scaling).to(dtype)
    return cos, sin


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(q: torch.Tensor, k: torch.Tensor, cos, sin, position_ids: torch.Tensor):
    """q [T,Hq,D], k [T,Hkv,D] (NHD), position_ids [T].  Arithmetic in the
    tensors' own dtype, exactly as eager torch does in the reference."""
    c = cos[position_ids].unsqueeze(1)
    s = sin[position_ids].unsqueeze(1)
    return q * c + _rot_half(q) * s, k * c + _rot_half(k) * s


# ----------------------------------------------------------------------------
# Masked (tree) attention -- umbrella/attn/cache.py:67-87 (flashinfer custom
# mask prefill) and :169-192 (the reference's own pure-torch statement, which
# is what is restated here): softmax(q k^T / sqrt(D) + mask) v with GQA.
# mask: bool [T, S], True = attend.
# ----------------------------------------------------------------------------

def masked_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask: torch.Tensor):
    T, Hq, D = q.shape
    S, Hkv, _ = k.shape
    g = Hq // Hkv
    qh = q.permute(1, 0, 2).reshape(Hkv, g * T, D)             # [Hkv, g*T, D]  (head-major rows)
    kh = k.permute(1, 0, 2)                                      # [Hkv, S, D]
    vh = v.permute(1, 0, 2)
    w = torch.matmul(qh, kh.transpose(1, 2)) / math.sqrt(D)     # [Hkv, g*T, S]
    m = mask[None, :, :].repeat(1, g, 1)                         # rows ordered (group, t)
    w = w.masked_fill(~m, torch.finfo(w.dtype).min)
    p = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(p, vh)                                      # [Hkv, g*T, D]
    return o.reshape(Hq, T, D).permute(1, 0, 2).contiguous()    # [T, Hq, D]


# ----------------------------------------------------------------------------
# AWQ (AutoAWQ "GEMM" format, autoawq==0.2.7.post3 / autoawq-kernels==0.0.8;
# call sites umbrella/quantization/awq_utils.py:63-86).
#   qweight [K, N/8] int32 : nibble i of word c holds column 8c + ORDER[i]
#   qzeros  [K/G, N/8] int32 : same packing
#   scales  [K/G, N] fp16
#   W[k, n] = (q[k, n] - z[k // G, n]) * s[k // G, n]
# ----------------------------------------------------------------------------

AWQ_ORDER = [0, 2, 4, 6, 1, 3, 5, 7]


def awq_pack(vals: np.ndarray) -> np.ndarray:
    """vals: integer array [R, N] with entries in [0, 15] -> packed int32 [R, N/8]."""
    R, N = vals.shape
    assert N % 8 == 0
    v = vals.astype(np.uint32).reshape(R, N // 8, 8)
    out = np.zeros((R, N // 8), dtype=np.uint32)
    for i, col in enumerate(AWQ_ORDER):
        out |= (v[:, :, col] & 0xF) << np.uint32(4 * i)
    return out.view(np.int32)


def awq_unpack(packed: np.ndarray) -> np.ndarray:
    """packed int32 [R, N/8] -> uint8 [R, N]."""
    p = packed.view(np.uint32)
    R, C = p.shape
    out = np.zeros((R, C, 8), dtype=np.uint8)
    for i, col in enumerate(AWQ_ORDER):
        out[:, :, col] = (p >> np.uint32(4 * i)) & 0xF
    return out.reshape(R, C * 8)


def awq_dequant(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor, group: int = 128):
    """-> W [K, N] in scales.dtype  (awq_ext.dequantize_weights_cuda semantics)."""
    q = torch.from_numpy(awq_unpack(qweight.numpy())).to(torch.float32)      # [K, N]
    z = torch.from_numpy(awq_unpack(qzeros.numpy())).to(torch.float32)       # [K/G, N]
    z = z.repeat_interleave(group, dim=0)
    s = scales.float().repeat_interleave(group, dim=0)
    return ((q - z) * s).to(scales.dtype)


def awq_linear(x: torch.Tensor, qweight, qzeros, scales, group: int = 128, bias=None):
    """AwqLinear.apply (awq_utils.py:63-86): both branches compute x @ dequant(W)."""
    w = awq_dequant(qweight, qzeros, scales, group).to(x.dtype)
    out = torch.matmul(x, w)
    return out + bias if bias is not None else out


# ----------------------------------------------------------------------------
# Draft expand helper -- umbrella/speculation/speculation_utils.py:57-61
# ----------------------------------------------------------------------------

def topk_flatten_gather(logits: torch.Tensor, num_samples: int, indices: torch.Tensor):
    return logits.topk(k=num_samples).indices.flatten().index_select(0, indices)


# ----------------------------------------------------------------------------
# Accept scan -- static_speculation_engine.py:313-325 / dynamic:283-301.
# Integer-exact.  sampled/spec/parents [T]; tree_mask bool [T,T] (row j =
# ancestors of j incl. j); want [T] = number of nodes on the root..j path.
# Returns (accept_path LongTensor sorted ascending, target_token int).
# ----------------------------------------------------------------------------

def accept_scan(sampled: torch.Tensor, spec: torch.Tensor, parents: torch.Tensor,
                tree_mask: torch.Tensor, want: torch.Tensor):
    ok = sampled[parents.long()] == spec
    ok[0] = True
    got = (ok[None, :] & tree_mask).sum(dim=-1)
    path = (got == want).nonzero().squeeze(-1)
    return path, int(sampled[path[-1]])


def first_eos(seq, eos_tokens) -> int:
    """speculation_utils.py:316-338: index of first element of seq in eos_tokens, else -1."""
    for i, t in enumerate(list(seq)):
        if int(t) in eos_tokens:
            return i
    return -1


# speculation_utils.py:340-345
def repetition_penalty(input_ids: torch.Tensor, logits: torch.Tensor, penalty: float):
    g = torch.gather(logits, 1, input_ids)
    g = torch.where(g < 0, g * penalty, g / penalty)
    return logits.scatter(1, input_ids, g)


# speculation_utils.py:347-352
def keep_topk(logits: torch.Tensor, topk: int):
    k = min(topk, logits.size(-1))
    kth = torch.topk(logits, k)[0][..., -1, None]
    return logits.masked_fill(logits < kth, torch.finfo(logits.dtype).min)


# flashinfer.sampling.top_p_renorm_prob (published semantics): keep the
# smallest set of highest-probability entries whose mass reaches top_p,
# zero the rest, renormalise.
def top_p_renorm(probs: torch.Tensor, top_p: float):
    sp, si = torch.sort(probs, dim=-1, descending=True)
    cum = torch.cumsum(sp, dim=-1)
    drop = (cum - sp) >= top_p
    sp = sp.masked_fill(drop, 0.0)
    out = torch.zeros_like(probs).scatter(-1, si, sp)
    return out / out.sum(dim=-1, keepdim=True)


# ----------------------------------------------------------------------------
# flashinfer.sampling.top_k_top_p_sampling_from_logits(logits, uniform_samples, top_k, top_p)
# -- the static engine's verification sampler (static_speculation_engine.py:131,310).
#
# PARITY UNPINNED: the flashinfer wheel is absent here and the reference does not pin its version (install.sh:2); the
# positional `uniform_samples` argument belongs to the 0.2.x API.  What follows restates that release's published
# algorithm (python/flashinfer/sampling.py + include/flashinfer/sampling.cuh), default filter_apply_order
# "top_k_first":  top_k_mask_logits -> softmax -> top_p_sampling_from_probs, the latter a REJECTION sampler driven by
# caller-supplied uniforms u[round][row] (TopPSamplingFromProbKernel):
#     q = 1, pivot = 0
#     for round in 0 .. rounds-1:
#         id    = first index i (vocabulary order) with  sum_{j <= i, p_j > pivot} p_j  >  u[round] * q   (else V - 1)
#         pivot = max(pivot, p_id)
#         q     = sum_{p_j > pivot} p_j              # mass strictly above the drawn token
#         if q < top_p: accept                        # the drawn token lies inside the nucleus
# After the last round the current id is returned whether accepted or not (success = False).  With unlimited rounds the
# output distribution is p restricted to the nucleus {i : mass(p_j > p_i) < top_p}, renormalised -- the same set
# top_p_renorm() keeps; with the reference's 3 rounds at most (1 - top_p)^3 of the mass falls outside it.
# ----------------------------------------------------------------------------
def top_k_mask_logits(logits: torch.Tensor, top_k: int):
    k = min(int(top_k), logits.size(-1))
    kth = torch.topk(logits, k, dim=-1)[0][..., -1, None]
    return logits.masked_fill(logits < kth, -torch.inf)


def top_p_sampling_from_probs(probs: torch.Tensor, uniform_samples: torch.Tensor, top_p: float):
    """probs [B, V] fp32, uniform_samples [rounds, B] in [0, 1).  Returns (ids int64 [B], success bool [B])."""
    B, V = probs.shape
    rounds = uniform_samples.shape[0]
    ids = torch.full((B,), V - 1, dtype=torch.long)
    ok = torch.zeros(B, dtype=torch.bool)
    for b in range(B):
        p = probs[b].float()
        q, pivot = 1.0, 0.0
        for r in range(rounds):
            u = float(uniform_samples[r, b]) * q
            cum = torch.cumsum(torch.where(p > pivot, p, torch.zeros_like(p)), dim=0)
            hit = (cum > u).nonzero()
            sid = int(hit[0]) if hit.numel() else V - 1
            pivot = max(pivot, float(p[sid]))
            q = float(p[p > pivot].sum())
            ids[b] = sid
            if q < top_p:
                ok[b] = True
                break
    return ids, ok


def top_k_top_p_sampling_from_logits(logits: torch.Tensor, uniform_samples: torch.Tensor, top_k: int, top_p: float):
    probs = torch.softmax(top_k_mask_logits(logits.float(), top_k), dim=-1)
    return top_p_sampling_from_probs(probs, uniform_samples, top_p)


def nucleus_distribution(logits: torch.Tensor, top_k: int, top_p: float):
    """The limit distribution of the sampler above (unlimited rounds): softmax over the top-k logits, restricted to
    {i : mass of strictly more probable tokens < top_p}, renormalised.  [B, V] fp32."""
    probs = torch.softmax(top_k_mask_logits(logits.float(), top_k), dim=-1)
    out = torch.zeros_like(probs)
    for b in range(probs.shape[0]):
        p = probs[b]
        above = (p[None, :] > p[:, None]).float() @ p            # mass strictly above each token
        keep = (above < top_p) & (p > 0)
        out[b] = torch.where(keep, p, torch.zeros_like(p))
    return out / out.sum(dim=-1, keepdim=True)
