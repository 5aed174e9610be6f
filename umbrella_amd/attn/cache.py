"""KV cache for tree-structured decoding on MI355X.

Replaces umbrella/attn/cache.py (KV_Cache :5-96, StaticKV_Cache :98-192) with one
layout chosen for the HIP attention kernel's MFMA operand loads: inside each
(layer, kv head) slab the elements are stored in FRAGMENT order (round 4;
umbrella_amd/csrc/common.h `kc_off` / `vt_off` are the device-side statement):

    K    slab of Lmax * D elements: a 32-key tile = 2 * (D / 32) fragments of 1 KiB, fragment (s, ds) lane (j, g) holds
         key 32 * tile + 8 * (j / 4) + 4 * s + j % 4, features 32 * ds + 8 * g .. + 7   (A operand of S^T = K Q^T)
    V^T  slab of D * (Lmax + VT_PAD) elements (the first Lmax * D are used): a 32-key tile = D / 16 fragments, fragment dt
         lane (j, g) holds feature 16 * dt + j, keys 32 * tile + 8 * g .. + 7              (A operand of O^T = V^T P^T)

so every load instruction of the attention kernels reads one contiguous KiB (the row-major [Lmax][D] / [D][Lmax] forms gave
16 rows x 64 B per instruction: profiles/r04_attn_frag_probe.txt).  `k_rows` / `v_rows` / `set_rows` and the two
`*_to_frag` / `*_from_frag` pairs convert to and from the semantic [.., Lmax, D] / [.., D, Lmax] views (tests, tools, the
reference-style `gather_kv_incremental`); the hot path never does.

Slots are addressed explicitly (StaticKV semantics); appending is "slot ==
kv_offset".  ``gather_kv_incremental`` keeps the reference signature; the engines
use the device-side ``compact`` (accepted path read from device memory, no tail
memset -- stale slots are never visible because every read is masked by
prefix_len / the tree mask).
"""
from __future__ import annotations

import torch

from .. import _lib

VT_PAD = 32          # == UMB_VT_PAD (include/umbrella_hip.h): slab stride of the V^T cache is D * (Lmax + VT_PAD)


def k_offsets(pos: torch.Tensor, D: int) -> torch.Tensor:
    """element offsets inside a K slab of (key pos[i], feature d) -> LongTensor [len(pos), D]   (common.h kc_off)"""
    p = pos.to(torch.long)[:, None]
    d = torch.arange(D, device=pos.device, dtype=torch.long)[None, :]
    kk = p & 31
    s, j = (kk >> 2) & 1, ((kk >> 3) << 2) | (kk & 3)
    return ((((p >> 5) * 2 + s) * (D >> 5) + (d >> 5)) * 64 + ((d >> 3) & 3) * 16 + j) * 8 + (d & 7)


def vt_offsets(pos: torch.Tensor, D: int) -> torch.Tensor:
    """element offsets inside a V^T slab of (feature d, key pos[i]) -> LongTensor [D, len(pos)]   (common.h vt_off)"""
    p = pos.to(torch.long)[None, :]
    d = torch.arange(D, device=pos.device, dtype=torch.long)[:, None]
    return (((p >> 5) * (D >> 4) + (d >> 4)) * 64 + ((p >> 3) & 3) * 16 + (d & 15)) * 8 + (p & 7)


def k_to_frag(k: torch.Tensor) -> torch.Tensor:
    """semantic K [..., Lmax, D] -> the same shape in fragment order (what the kernels read)"""
    Lmax, D = k.shape[-2:]
    assert Lmax % 32 == 0 and D % 32 == 0
    off = k_offsets(torch.arange(Lmax, device=k.device), D).reshape(-1)
    out = torch.empty_like(k).reshape(*k.shape[:-2], Lmax * D)
    out[..., off] = k.reshape(*k.shape[:-2], Lmax * D)
    return out.reshape(k.shape)


def k_from_frag(ks: torch.Tensor) -> torch.Tensor:
    Lmax, D = ks.shape[-2:]
    off = k_offsets(torch.arange(Lmax, device=ks.device), D).reshape(-1)
    return ks.reshape(*ks.shape[:-2], Lmax * D)[..., off].reshape(ks.shape)


def vt_to_frag(vt: torch.Tensor) -> torch.Tensor:
    """semantic V^T [..., D, Lmax + VT_PAD] -> the same shape, keys 0 .. Lmax-1 in fragment order, the pad tail zero"""
    D, LV = vt.shape[-2:]
    Lmax = LV - VT_PAD
    assert Lmax % 32 == 0 and D % 16 == 0
    off = vt_offsets(torch.arange(Lmax, device=vt.device), D).reshape(-1)
    out = torch.zeros_like(vt).reshape(*vt.shape[:-2], D * LV)
    out[..., off] = vt[..., :Lmax].reshape(*vt.shape[:-2], D * Lmax)
    return out.reshape(vt.shape)


def vt_from_frag(vs: torch.Tensor) -> torch.Tensor:
    D, LV = vs.shape[-2:]
    Lmax = LV - VT_PAD
    off = vt_offsets(torch.arange(Lmax, device=vs.device), D).reshape(-1)
    out = torch.zeros_like(vs)
    out[..., :Lmax] = vs.reshape(*vs.shape[:-2], D * LV)[..., off].reshape(*vs.shape[:-2], D, Lmax)
    return out


class TreeKVCache:
    def __init__(self, num_layers, num_kv_heads, head_dim, max_length, device, dtype):
        assert max_length % 32 == 0, "the fragment-ordered cache is tiled by 32 keys"
        self.num_layers, self.num_key_value_heads, self.head_dim = num_layers, num_kv_heads, head_dim
        self.max_length, self.device, self.dtype = max_length, device, dtype
        # storage (fragment order inside each [layer, head] slab; the shapes only carry the slab strides)
        self.k = torch.zeros(num_layers, num_kv_heads, max_length, head_dim, device=device, dtype=dtype)
        self.vt = torch.zeros(num_layers, num_kv_heads, head_dim, max_length + VT_PAD, device=device, dtype=dtype)
        self.kv_offset = 0

    # ---- semantic access (tests, tools, the reference-style gather): never on the hot path
    def _flat(self):
        L, H = self.num_layers, self.num_key_value_heads
        return self.k.view(L, H, -1), self.vt.view(L, H, -1)

    def k_rows(self, pos) -> torch.Tensor:
        """K of the keys `pos` -> [L, Hkv, len(pos), D]"""
        pos = torch.as_tensor(pos, device=self.device, dtype=torch.long)
        return self._flat()[0][:, :, k_offsets(pos, self.head_dim)]

    def v_rows(self, pos) -> torch.Tensor:
        """V of the keys `pos` -> [L, Hkv, len(pos), D]"""
        pos = torch.as_tensor(pos, device=self.device, dtype=torch.long)
        return self._flat()[1][:, :, vt_offsets(pos, self.head_dim).t()]

    def set_rows(self, pos, k: torch.Tensor, v: torch.Tensor):
        """write K / V [L, Hkv, len(pos), D] at the keys `pos`"""
        pos = torch.as_tensor(pos, device=self.device, dtype=torch.long)
        kf, vf = self._flat()
        kf[:, :, k_offsets(pos, self.head_dim)] = k.to(self.dtype)
        vf[:, :, vt_offsets(pos, self.head_dim).t()] = v.to(self.dtype)

    # reference API (cache.py:41-49): indices are absolute slots, moved to [offset, offset+len)
    def gather_kv_incremental(self, indices, offset: int):
        idx = torch.as_tensor(indices, device=self.device, dtype=torch.long)
        a = idx.numel()
        if a:
            k, v = self.k_rows(idx).clone(), self.v_rows(idx).clone()
            self.set_rows(torch.arange(offset, offset + a, device=self.device), k, v)
        self.kv_offset = offset + a

    def compact(self, result: torch.Tensor, path: torch.Tensor, max_path: int):
        """Device-side compaction driven by umb_accept_scan's outputs (graph-capturable)."""
        _lib.call("umb_kv_compact", self.k, self.vt, result, path, self.num_layers, self.num_key_value_heads,
                  self.head_dim, self.max_length, max_path, _lib.dtype_code(self.dtype))

    def clear(self):
        self.k.zero_()
        self.vt.zero_()
        self.kv_offset = 0

    def set_kv_len(self, kv_len: int):
        self.kv_offset = kv_len
