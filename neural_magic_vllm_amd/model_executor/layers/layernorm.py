"""RMSNorm (reference: vllm/model_executor/layers/layernorm.py:10-77)."""
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from ... import _custom_ops as ops
from .custom_op import CustomOp


class RMSNorm(CustomOp):
    """Root mean square normalisation: w * x / sqrt(E[x^2] + eps)."""

    def __init__(self, hidden_size: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward_native(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None
                       ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        orig_dtype = x.dtype
        x = x.to(torch.float32)
        if residual is not None:
            x = x + residual.to(torch.float32)
            residual = x.to(orig_dtype)
        variance = x.pow(2).mean(dim=-1, keepdim=True)
        x = x * torch.rsqrt(variance + self.variance_epsilon)
        x = x.to(orig_dtype) * self.weight
        return x if residual is None else (x, residual)

    def forward_cuda(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None
                     ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        if residual is not None:
            ops.fused_add_rms_norm(x, residual, self.weight.data, self.variance_epsilon)
            return x, residual
        out = torch.empty_like(x)
        ops.rms_norm(out, x, self.weight.data, self.variance_epsilon)
        return out
