"""Legacy Marlin-serialised checkpoints (reference: quantization/marlin.py:17-252): B int32
[K/16, N*16/8], s [K/g, N], workspace int32 [N/64*16]; 4-bit, group -1 / 128."""
from typing import Any, Dict, List, Optional

import torch
from torch.nn.parameter import Parameter

from .... import _custom_ops as ops
from ...utils import set_weight_attrs
from .base_config import LinearMethodBase, QuantizationConfig


class MarlinConfig(QuantizationConfig):
    """Config class for Marlin (https://github.com/IST-DASLab/marlin)."""

    def __init__(self, group_size: int, lm_head_quantized: bool = False) -> None:
        self.group_size = group_size
        self.lm_head_quantized = lm_head_quantized
        if self.group_size != 128 and self.group_size != -1:
            raise ValueError("Currently, only group size 128 and -1 (channelwise) is supported "
                             f"for Marlin, but got group_size of {self.group_size}")
        self.pack_factor = 32 // 4
        self.tile_size = 16
        self.min_n_threads = 64
        self.min_k_threads = 128
        self.max_parallel = 16
        self.perm_len = 1024

    def __repr__(self) -> str:
        return f"MarlinConfig(group_size={self.group_size}, lm_head_quantized={self.lm_head_quantized})"

    @classmethod
    def get_name(cls) -> str:
        return "marlin"

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.half, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 80

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return ["quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "MarlinConfig":
        return cls(cls.get_from_keys(config, ["group_size"]),
                   cls.get_from_keys_or(config, ["lm_head"], default=False))

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant) -> Optional[str]:
        is_marlin_format = (hf_quant_cfg.get("checkpoint_format") == "marlin"
                            or hf_quant_cfg.get("is_marlin_format", False))
        if is_marlin_format and (user_quant is None or user_quant in ("gptq", "marlin")):
            return cls.get_name()
        return None

    def get_quant_method(self, layer: torch.nn.Module) -> Optional["MarlinLinearMethod"]:
        from ..linear import LinearBase
        from ..vocab_parallel_embedding import ParallelLMHead
        if isinstance(layer, LinearBase) or (isinstance(layer, ParallelLMHead)
                                             and self.lm_head_quantized):
            return MarlinLinearMethod(self)
        return None

    def get_scaled_act_names(self) -> List[str]:
        return []


class MarlinLinearMethod(LinearMethodBase):

    def __init__(self, quant_config: MarlinConfig):
        self.quant_config = quant_config

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size,
                       output_size, params_dtype, **extra_weight_attrs):
        del output_size
        cfg = self.quant_config
        if params_dtype not in (torch.float16, torch.bfloat16):
            raise ValueError(f"The params dtype must be float16 or bfloat16, but got {params_dtype}")
        output_size_per_partition = sum(output_partition_sizes)
        if output_size_per_partition % cfg.min_n_threads != 0:
            raise ValueError(f"Weight output_size_per_partition = {output_size_per_partition} is "
                             f"not divisible by min_n_threads = {cfg.min_n_threads}.")
        if output_size_per_partition % cfg.pack_factor != 0:
            raise ValueError(f"Weight output_size_per_partition = {output_size_per_partition} is "
                             f"not divisible by pack_factor = {cfg.pack_factor}.")
        if input_size_per_partition % cfg.min_k_threads != 0:
            raise ValueError(f"Weight input_size_per_partition = {input_size_per_partition} is "
                             f"not divisible by min_k_threads = {cfg.min_k_threads}.")
        if cfg.group_size != -1 and input_size_per_partition % cfg.group_size != 0:
            raise ValueError(f"Weight input_size_per_partition = {input_size_per_partition} is "
                             f"not divisible by group_size = {cfg.group_size}.")
        num_tiles_per_perm = cfg.perm_len // (cfg.tile_size**2)
        if output_size_per_partition % num_tiles_per_perm != 0:
            raise ValueError("Each permutation group must reside on the same gpu")
        qweight = Parameter(torch.empty(input_size_per_partition // cfg.tile_size,
                                        output_size_per_partition * cfg.tile_size // cfg.pack_factor,
                                        dtype=torch.int32), requires_grad=False)
        set_weight_attrs(qweight, {"input_dim": 0, "output_dim": 1, "packed_dim": 1,
                                   "pack_factor": cfg.pack_factor,
                                   "marlin_tile_size": cfg.tile_size})
        input_groups = 1 if cfg.group_size == -1 else input_size_per_partition // cfg.group_size
        scales = Parameter(torch.empty(input_groups, output_size_per_partition, dtype=params_dtype),
                           requires_grad=False)
        set_weight_attrs(scales, {"input_dim": None if input_groups == 1 else 0, "output_dim": 1})
        max_workspace_size = (output_size_per_partition // cfg.min_n_threads) * cfg.max_parallel
        workspace = Parameter(torch.zeros(max_workspace_size, dtype=torch.int), requires_grad=False)
        layer.register_parameter("B", qweight)
        set_weight_attrs(qweight, extra_weight_attrs)
        layer.register_parameter("s", scales)
        set_weight_attrs(scales, extra_weight_attrs)
        layer.register_parameter("workspace", workspace)
        set_weight_attrs(workspace, extra_weight_attrs)

    def apply(self, layer, x, bias=None):
        qweight, scales, workspace = layer.B, layer.s, layer.workspace
        x_2d = x.view(-1, x.shape[-1])
        size_m, size_k = x_2d.shape
        size_n = scales.shape[1]
        output_2d = ops.marlin_gemm(x_2d, qweight, scales, workspace, size_m, size_n, size_k)
        output = output_2d.view(x.shape[:-1] + (output_2d.shape[1], ))
        if bias is not None:
            output.add_(bias)
        return output
