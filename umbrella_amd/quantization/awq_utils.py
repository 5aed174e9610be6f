"""AWQ int4 linear on the HIP path (replaces umbrella/quantization/awq_utils.py:5-86).

Keeps the reference holder's surface (``init_parameters / empty_like / to / copy / apply``)
but stores the weights re-packed into MFMA tile order; ``apply`` runs the int4 skinny GEMM
(no awq_ext, no fp16 dequant + cuBLAS branch: one kernel family covers every T).
"""
from __future__ import annotations

import torch

from ..models.llama import PackedLinear


class AwqLinear:
    def __init__(self):
        self.in_features = self.out_features = 0
        self.w_bit, self.group_size = 4, 128
        self.qweight = self.qzeros = self.scales = self.bias = None
        self.packed: PackedLinear | None = None

    def init_parameters(self, module):
        """module: any object with AutoAWQ GEMM tensors (qweight/qzeros/scales[/bias], in/out_features)."""
        self.in_features, self.out_features = module.in_features, module.out_features
        self.w_bit, self.group_size = getattr(module, "w_bit", 4), getattr(module, "group_size", 128)
        self.qweight, self.qzeros, self.scales = module.qweight.detach(), module.qzeros.detach(), module.scales.detach()
        self.bias = module.bias.detach() if getattr(module, "bias", None) is not None else None

    def empty_like(self, module):
        self.init_parameters(module)
        self.qweight, self.qzeros, self.scales = (torch.zeros_like(t) for t in (self.qweight, self.qzeros, self.scales))

    def to(self, device, non_blocking=True):
        self.qweight = self.qweight.to(device, non_blocking=non_blocking)
        self.qzeros = self.qzeros.to(device, non_blocking=non_blocking)
        self.scales = self.scales.to(device, non_blocking=non_blocking)
        if self.bias is not None:
            self.bias = self.bias.to(device, non_blocking=non_blocking)
        if torch.device(device).type == "cuda":
            self.packed = PackedLinear.from_awq(self.qweight, self.qzeros, self.scales, self.group_size)

    def copy(self, module: "AwqLinear", non_blocking=True):
        self.qweight.copy_(module.qweight, non_blocking=non_blocking)
        self.qzeros.copy_(module.qzeros, non_blocking=non_blocking)
        self.scales.copy_(module.scales, non_blocking=non_blocking)
        if self.qweight.is_cuda:
            self.packed = PackedLinear.from_awq(self.qweight, self.qzeros, self.scales, self.group_size)

    def apply(self, x: torch.Tensor):
        if self.packed is None:
            raise RuntimeError("AwqLinear.apply needs the weights on the GPU (call .to('cuda:0') first)")
        out_shape = x.shape[:-1] + (self.out_features,)
        out = self.packed.apply(x.reshape(-1, x.shape[-1]).contiguous()).to(x.dtype)
        out = out + self.bias if self.bias is not None else out
        out = out.reshape(out_shape)
        return out.unsqueeze(0) if out.dim() == 2 else out
