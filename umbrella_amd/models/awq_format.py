"""AutoAWQ "GEMM" tensor format helpers (host side, numpy).

Format consumed by the reference's AwqLinear (umbrella/quantization/awq_utils.py:5-86,
produced by autoawq==0.2.7.post3): nibble ``i`` of packed word ``c`` holds logical
column ``8*c + ORDER[i]`` with ORDER = [0,2,4,6,1,3,5,7].
"""
from __future__ import annotations

import numpy as np

ORDER = (0, 2, 4, 6, 1, 3, 5, 7)


def pack_rows(vals: np.ndarray) -> np.ndarray:
    """uint8 [R, N] (values 0..15) -> int32 [R, N/8]."""
    R, N = vals.shape
    v = vals.astype(np.uint32).reshape(R, N // 8, 8)
    out = np.zeros((R, N // 8), dtype=np.uint32)
    for i, col in enumerate(ORDER):
        out |= (v[:, :, col] & 0xF) << np.uint32(4 * i)
    return out.view(np.int32)


def unpack_rows(packed: np.ndarray) -> np.ndarray:
    """int32 [R, N/8] -> uint8 [R, N]."""
    p = packed.view(np.uint32)
    out = np.zeros(p.shape + (8,), dtype=np.uint8)
    for i, col in enumerate(ORDER):
        out[:, :, col] = (p >> np.uint32(4 * i)) & 0xF
    return out.reshape(p.shape[0], -1)
