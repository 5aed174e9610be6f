"""KV cache for tree-structured decoding on MI355X.

Replaces umbrella/attn/cache.py (KV_Cache :5-96, StaticKV_Cache :98-192) with one
layout chosen for the HIP attention kernel's 16-byte MFMA fragment loads:

    K  [L][Hkv][Lmax][D]        (row = one key, contiguous D)
    V^T[L][Hkv][D][Lmax + VT_PAD]  (transposed: 8 consecutive keys of one d are 16 B; rows padded so the
                                 16 d-rows of a fragment load spread over memory channels)

Slots are addressed explicitly (StaticKV semantics); appending is "slot ==
kv_offset".  ``gather_kv_incremental`` keeps the reference signature; the engines
use the device-side ``compact`` (accepted path read from device memory, no tail
memset -- stale slots are never visible because every read is masked by
prefix_len / the tree mask).
"""
from __future__ import annotations

import torch

from .. import _lib

VT_PAD = 32          # == UMB_VT_PAD (include/umbrella_hip.h)


class TreeKVCache:
    def __init__(self, num_layers, num_kv_heads, head_dim, max_length, device, dtype):
        self.num_layers, self.num_key_value_heads, self.head_dim = num_layers, num_kv_heads, head_dim
        self.max_length, self.device, self.dtype = max_length, device, dtype
        self.k = torch.zeros(num_layers, num_kv_heads, max_length, head_dim, device=device, dtype=dtype)
        self.vt = torch.zeros(num_layers, num_kv_heads, head_dim, max_length + VT_PAD, device=device, dtype=dtype)
        self.kv_offset = 0

    # reference API (cache.py:41-49): indices are absolute slots, moved to [offset, offset+len)
    def gather_kv_incremental(self, indices, offset: int):
        idx = torch.as_tensor(indices, device=self.device, dtype=torch.long)
        a = idx.numel()
        if a:
            self.k[:, :, offset:offset + a, :] = self.k[:, :, idx, :]
            self.vt[:, :, :, offset:offset + a] = self.vt[:, :, :, idx]
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
