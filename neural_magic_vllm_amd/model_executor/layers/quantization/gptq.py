"""GPTQ (exllama path) quantisation method (reference: quantization/gptq.py:19-231): same parameters
(qweight int32 [K/pack, N], g_idx int32 [K], qzeros int32 [G, N/pack], scales [G, N]) and the same
first-forward shuffle: g_idx <- argsort(g_idx) (act-order) or empty, ops.gptq_shuffle, then
ops.gptq_gemm(..., use_exllama, bits)."""
import enum
from enum import Enum
from fractions import Fraction
from typing import Any, Dict, List, Optional

import torch
from torch.nn.parameter import Parameter

from .... import _custom_ops as ops
from ...utils import set_weight_attrs
from .base_config import LinearMethodBase, QuantizationConfig


class GPTQConfig(QuantizationConfig):
    """Config class for GPTQ (https://arxiv.org/abs/2210.17323)."""

    def __init__(self, weight_bits: int, group_size: int, desc_act: bool,
                 lm_head_quantized: bool = False) -> None:
        self.weight_bits = weight_bits
        self.group_size = group_size
        self.desc_act = desc_act
        self.lm_head_quantized = lm_head_quantized
        self.pack_factor = Fraction(32, self.weight_bits)
        if self.weight_bits not in [2, 3, 4, 8]:
            raise ValueError("Currently, only 2/3/4/8-bit weight quantization is supported for "
                             f"GPTQ, but got {self.weight_bits} bits.")

    def __repr__(self) -> str:
        return (f"GPTQConfig(weight_bits={self.weight_bits}, group_size={self.group_size}, "
                f"desc_act={self.desc_act}), lm_head_quantized={self.lm_head_quantized}")

    @classmethod
    def get_name(cls) -> str:
        return "gptq"

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.half, torch.bfloat16]  # the reference's CUDA kernel is fp16-only

    @classmethod
    def get_min_capability(cls) -> int:
        return 60

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return ["quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "GPTQConfig":
        return cls(cls.get_from_keys(config, ["bits"]), cls.get_from_keys(config, ["group_size"]),
                   cls.get_from_keys(config, ["desc_act"]),
                   cls.get_from_keys_or(config, ["lm_head"], default=False))

    def get_quant_method(self, layer: torch.nn.Module) -> Optional["GPTQLinearMethod"]:
        from ..linear import LinearBase
        from ..vocab_parallel_embedding import ParallelLMHead
        if isinstance(layer, LinearBase) or (isinstance(layer, ParallelLMHead)
                                             and self.lm_head_quantized):
            return GPTQLinearMethod(self)
        return None

    def get_scaled_act_names(self) -> List[str]:
        return []


class ExllamaState(Enum):
    UNUSED = enum.auto()
    UNINITIALIZED = enum.auto()
    READY = enum.auto()


class GPTQLinearMethod(LinearMethodBase):

    def __init__(self, quant_config: GPTQConfig):
        self.quant_config = quant_config

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size,
                       output_size, params_dtype, **extra_weight_attrs):
        del output_size
        cfg = self.quant_config
        if input_size_per_partition % cfg.group_size != 0:
            raise ValueError("The input size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        output_size_per_partition = sum(output_partition_sizes)
        if output_size_per_partition % cfg.pack_factor.numerator != 0:
            raise ValueError("The output size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        group_size = cfg.group_size if cfg.group_size != -1 else input_size
        exllama_state = ExllamaState.UNINITIALIZED
        scale_and_zero_size = input_size // group_size
        scale_and_zero_input_dim = None
        if input_size != input_size_per_partition and cfg.group_size != -1:
            if cfg.desc_act:  # act-order + row parallel: exllama cannot be used (gptq.py:133-136)
                exllama_state = ExllamaState.UNUSED
            else:
                scale_and_zero_size = input_size_per_partition // group_size
                scale_and_zero_input_dim = 0
        pf = cfg.pack_factor   # a Fraction: 32/3 for 3-bit codes (gptq.py:34, :137-160)
        qweight = Parameter(torch.empty(int(input_size_per_partition // pf), output_size_per_partition,
                                        dtype=torch.int32), requires_grad=False)
        set_weight_attrs(qweight, {"input_dim": 0, "output_dim": 1, "packed_dim": 0,
                                   "pack_factor": cfg.pack_factor})
        g_idx = Parameter(torch.tensor([i // cfg.group_size for i in range(input_size_per_partition)],
                                       dtype=torch.int32), requires_grad=False)
        set_weight_attrs(g_idx, {"input_dim": 0, "ignore_warning": True})
        qzeros = Parameter(torch.empty(scale_and_zero_size, int(output_size_per_partition // pf),
                                       dtype=torch.int32), requires_grad=False)
        set_weight_attrs(qzeros, {"input_dim": scale_and_zero_input_dim, "output_dim": 1,
                                  "packed_dim": 1, "pack_factor": cfg.pack_factor})
        scales = Parameter(torch.empty(scale_and_zero_size, output_size_per_partition,
                                       dtype=params_dtype), requires_grad=False)
        set_weight_attrs(scales, {"input_dim": scale_and_zero_input_dim, "output_dim": 1})
        for name, prm in (("qweight", qweight), ("g_idx", g_idx), ("qzeros", qzeros),
                          ("scales", scales)):
            layer.register_parameter(name, prm)
            set_weight_attrs(prm, extra_weight_attrs)
        layer.exllama_state = exllama_state

    def process_weights_after_loading(self, layer) -> None:
        """MI355X-first addition: a 4-bit, group-128, non-act-order layer whose shape allows it
        (K % 256 == 0, N % 64 == 0) is repacked ONCE to the Marlin layout and runs the tuned Marlin
        kernel with per-group zero points (2x the decode throughput of gptq_gemm on the checkpoint
        layout); symmetric checkpoints take GPTQMarlinLinearMethod before they get here, as in the
        reference.  The checkpoint-facing parameters are unchanged."""
        cfg = self.quant_config
        k, n = layer.qweight.shape[0] * int(cfg.pack_factor), layer.qweight.shape[1]
        if (cfg.weight_bits != 4 or cfg.group_size != 128 or cfg.desc_act or k % 256 != 0 or n % 64 != 0
                or layer.exllama_state != ExllamaState.UNINITIALIZED or layer.qweight.device.type != "cuda"
                or layer.scales.dtype not in (torch.half, torch.bfloat16)
                or layer.scales.shape[0] != k // 128):
            return
        dev = layer.qweight.device
        e = torch.empty(0, dtype=torch.int32, device=dev)
        shifts = torch.arange(0, 32, 4, dtype=torch.int32, device=dev)
        zeros = (((layer.qzeros.data.unsqueeze(-1) >> shifts) & 0xF) + 1).reshape(layer.qzeros.shape[0], -1)
        perm = torch.tensor([i + 8 * j for i in range(8) for j in range(8)], device=dev)
        layer.marlin_qweight = ops.gptq_marlin_repack(layer.qweight.data.contiguous(), e, k, n, 4)
        layer.marlin_scales = layer.scales.data.reshape(-1, 64)[:, perm].reshape(-1, n).contiguous()
        layer.marlin_zeros = zeros.to(layer.scales.dtype).reshape(-1, 64)[:, perm].reshape(-1, n).contiguous()
        layer.marlin_workspace = torch.zeros(n // 64 * 16, dtype=torch.int32, device=dev)
        layer.gptq_marlin_kn = (k, n)

    def apply(self, layer, x, bias=None):
        if getattr(layer, "gptq_marlin_kn", None) is not None:
            k, n = layer.gptq_marlin_kn
            x2 = x.reshape(-1, x.shape[-1])
            out = ops.marlin_zp_gemm(x2, layer.marlin_qweight, layer.marlin_scales, layer.marlin_zeros,
                                     layer.marlin_workspace, x2.shape[0], n, k)
            if bias is not None:
                out.add_(bias)
            return out.reshape(x.shape[:-1] + (n, ))
        qweight = layer.qweight
        out_shape = x.shape[:-1] + (qweight.shape[-1], )
        reshaped_x = x.reshape(-1, x.shape[-1])
        if layer.exllama_state == ExllamaState.UNINITIALIZED:
            if self.quant_config.desc_act:
                layer.g_idx.data = torch.argsort(layer.g_idx).to(torch.int)
            else:
                layer.g_idx.data = torch.empty((0, ), dtype=torch.int, device=layer.g_idx.device)
            layer.exllama_state = ExllamaState.READY
            ops.gptq_shuffle(layer.qweight, layer.g_idx, self.quant_config.weight_bits)
        output = ops.gptq_gemm(reshaped_x, layer.qweight, layer.qzeros, layer.scales, layer.g_idx,
                               layer.exllama_state == ExllamaState.READY,
                               self.quant_config.weight_bits)
        if bias is not None:
            output.add_(bias)
        return output.reshape(out_shape)
