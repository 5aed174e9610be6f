"""AWQ int4 linear on the HIP path (replaces umbrella/quantization/awq_utils.py:5-86).

Keeps the reference holder's surface (``init_parameters / empty_like / to / copy / apply``)
but stores the weights re-packed into MFMA tile order; ``apply`` runs the int4 skinny GEMM
(no awq_ext, no fp16 dequant + cuBLAS branch: one kernel family covers every T).
"""
from __future__ import annotations

import torch

from ..models.llama import PackedLinear


_TENSORS = ("qweight", "qzeros", "scales")


class AwqLinear:
    """AutoAWQ GEMM-format linear: qweight [K, N/8] i32, qzeros [K/G, N/8] i32, scales [K/G, N] fp16 (+ bias)."""

    def __init__(self):
        self.in_features = self.out_features = 0
        self.w_bit, self.group_size = 4, 128
        self.qweight = self.qzeros = self.scales = self.bias = None
        self.packed: PackedLinear | None = None          # tile-order copy, built when the tensors reach the GPU

    # ---- reference surface -------------------------------------------------------------------------------
    def init_parameters(self, module):
        """Adopt the tensors of any object that looks like an AutoAWQ ``WQLinear_GEMM``."""
        self.in_features, self.out_features = module.in_features, module.out_features
        self.w_bit = getattr(module, "w_bit", 4)
        self.group_size = getattr(module, "group_size", 128)
        for name in _TENSORS:
            setattr(self, name, getattr(module, name).detach())
        bias = getattr(module, "bias", None)
        self.bias = None if bias is None else bias.detach()

    def empty_like(self, module):
        """Same shapes as `module`, zero contents (staging buffer of the offload path)."""
        self.init_parameters(module)
        for name in _TENSORS:
            setattr(self, name, torch.zeros_like(getattr(self, name)))

    def to(self, device, non_blocking=True):
        for name in _TENSORS + ("bias",):
            t = getattr(self, name)
            if t is not None:
                setattr(self, name, t.to(device, non_blocking=non_blocking))
        self._repack_if_on_gpu()

    def copy(self, module: "AwqLinear", non_blocking=True):
        for name in _TENSORS:
            getattr(self, name).copy_(getattr(module, name), non_blocking=non_blocking)
        self._repack_if_on_gpu()

    def apply(self, x: torch.Tensor):
        """x [..., K] fp16 / bf16 -> [1, T, N] (the reference adds a leading 1 to 2-D results, awq_utils.py:83-84)."""
        if self.packed is None:
            raise RuntimeError("AwqLinear.apply needs the weights on the GPU (call .to('cuda:0') first)")
        rows = x.reshape(-1, x.shape[-1]).contiguous()
        y = self.packed.apply(rows).to(x.dtype)
        if self.bias is not None:
            y = y + self.bias
        y = y.reshape(*x.shape[:-1], self.out_features)
        return y[None] if y.dim() == 2 else y

    # ---- HIP side --------------------------------------------------------------------------------------------
    def _repack_if_on_gpu(self):
        if self.qweight is not None and self.qweight.is_cuda:
            self.packed = PackedLinear.from_awq(self.qweight, self.qzeros, self.scales, self.group_size)
