"""FP8 (e4m3fn) quantisation method (reference: quantization/fp8.py:22-379, 563-613).

On gfx950 `current_platform.get_device_capability()` is (9, 5): use_marlin (capability < 89) is
False and cutlass_fp8_supported() is True, so every linear takes
ops.scaled_fp8_quant -> ops.cutlass_scaled_mm (native OCP fp8 MFMA), bias included (the
reference falls back to torch._scaled_mm when a bias is present, fp8.py:347-365; here the
epilogue adds it).  The Marlin weight-only branch is kept for completeness behind use_marlin."""
from typing import Any, Dict, List, Optional, Union

import torch
from torch.nn import Module
from torch.nn.parameter import Parameter

from .... import _custom_ops as ops
from ....platforms import current_platform
from ...utils import set_weight_attrs
from .base_config import LinearMethodBase, QuantizationConfig, QuantizeMethodBase
from .gptq_marlin import (GPTQ_MARLIN_MAX_PARALLEL, GPTQ_MARLIN_MIN_THREAD_N, GPTQMarlinState,
                          marlin_permute_scales)

ACTIVATION_SCHEMES = ["static", "dynamic"]


def cutlass_fp8_supported() -> bool:
    capability = current_platform.get_device_capability()
    return ops.cutlass_scaled_mm_supports_fp8(capability[0] * 10 + capability[1])


def pack_fp8_to_int32(fp8_tensor: torch.Tensor) -> torch.Tensor:
    """4 consecutive rows (K) of fp8 bytes per int32 = GPTQ 8-bit packing (marlin_utils.py:227-247)"""
    assert fp8_tensor.dtype == torch.float8_e4m3fn and fp8_tensor.shape[0] % 4 == 0
    b = fp8_tensor.reshape(-1, 4, *fp8_tensor.shape[1:]).view(torch.uint8).to(torch.int32)
    packed = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16) | (b[:, 3] << 24)
    return packed.view(fp8_tensor.shape[0] // 4, *fp8_tensor.shape[1:]).contiguous()


class Fp8Config(QuantizationConfig):
    """Config class for FP8."""

    def __init__(self, is_checkpoint_fp8_serialized: bool = False,
                 activation_scheme: str = "dynamic") -> None:
        self.is_checkpoint_fp8_serialized = is_checkpoint_fp8_serialized
        if activation_scheme not in ACTIVATION_SCHEMES:
            raise ValueError(f"Unsupported activation scheme {activation_scheme}")
        self.activation_scheme = activation_scheme

    @classmethod
    def get_name(cls) -> str:
        return "fp8"

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.bfloat16, torch.half]

    @classmethod
    def get_min_capability(cls) -> int:
        return 80

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return []

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "Fp8Config":
        quant_method = cls.get_from_keys(config, ["quant_method"])
        return cls(is_checkpoint_fp8_serialized=("fp8" in quant_method),
                   activation_scheme=cls.get_from_keys(config, ["activation_scheme"]))

    def get_quant_method(self, layer: torch.nn.Module) -> Optional[QuantizeMethodBase]:
        from ....attention.layer import Attention
        from ..linear import LinearBase
        if isinstance(layer, LinearBase):
            return Fp8LinearMethod(self)
        if isinstance(layer, Attention):
            return Fp8KVCacheMethod(self)
        return None

    def get_scaled_act_names(self) -> List[str]:
        return []


class Fp8LinearMethod(LinearMethodBase):
    """Per-tensor fp8 weights (checkpoint-serialised or quantised at load), dynamic or static
    per-tensor activation scale."""

    def __init__(self, quant_config: Fp8Config):
        self.quant_config = quant_config
        self.cutlass_fp8_supported = cutlass_fp8_supported()
        capability = current_platform.get_device_capability()
        self.use_marlin = capability[0] * 10 + capability[1] < 89

    def _create_scale_param(self, scale_name, layer, output_partition_sizes, **extra_weight_attrs):
        scale = Parameter(torch.empty(len(output_partition_sizes), dtype=torch.float32),
                          requires_grad=False)
        scale[:] = torch.finfo(torch.float8_e4m3fn).min
        layer.register_parameter(scale_name, scale)
        set_weight_attrs(scale, {**extra_weight_attrs, "needs_scalar_to_array": True})

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size,
                       output_size, params_dtype, **extra_weight_attrs):
        del input_size, output_size
        output_size_per_partition = sum(output_partition_sizes)
        layer.process_after_load = True
        layer.logical_widths = output_partition_sizes
        layer.input_size_per_partition = input_size_per_partition
        layer.output_size_per_partition = output_size_per_partition
        layer.orig_dtype = params_dtype
        weight_dtype = (torch.float8_e4m3fn if self.quant_config.is_checkpoint_fp8_serialized
                        else params_dtype)
        weight = Parameter(torch.empty(output_size_per_partition, input_size_per_partition,
                                       dtype=weight_dtype), requires_grad=False)
        layer.register_parameter("weight", weight)
        set_weight_attrs(weight, {**extra_weight_attrs, "input_dim": 1, "output_dim": 0})
        if self.quant_config.is_checkpoint_fp8_serialized:
            self._create_scale_param("weight_scale", layer, output_partition_sizes,
                                     **extra_weight_attrs)
            if self.quant_config.activation_scheme == "static":
                self._create_scale_param("input_scale", layer, output_partition_sizes,
                                         **extra_weight_attrs)
        if self.use_marlin:
            layer.marlin_state = GPTQMarlinState.REPACK

    def prepare_layer_for_marlin(self, layer: Module) -> None:
        """weight-only fp8 through the Marlin 8-bit layout (fp8.py:195-247)"""
        part_size_n, part_size_k = layer.output_size_per_partition, layer.input_size_per_partition
        layer.marlin_state = GPTQMarlinState.READY
        device = layer.weight.device
        packed = pack_fp8_to_int32(layer.weight)
        marlin_qweight = ops.gptq_marlin_repack(packed, torch.empty(0, dtype=torch.int, device=device),
                                                part_size_k, part_size_n, 8)
        layer.weight = Parameter(marlin_qweight, requires_grad=False)
        scales = layer.weight_scale.repeat(1, part_size_n).to(layer.orig_dtype).to(device)
        layer.weight_scale = Parameter(marlin_permute_scales(scales, part_size_k, part_size_n, -1, 8),
                                       requires_grad=False)
        layer.workspace = torch.zeros((part_size_n // GPTQ_MARLIN_MIN_THREAD_N) *
                                      GPTQ_MARLIN_MAX_PARALLEL, dtype=torch.int, device=device)

    def process_weights_after_loading(self, layer: Module) -> None:
        if not getattr(layer, "process_after_load", False):
            return
        if not self.quant_config.is_checkpoint_fp8_serialized:
            # 16-bit checkpoint: quantise the whole (fused) weight with one dynamic scale
            qweight, weight_scale = ops.scaled_fp8_quant(layer.weight, scale=None)
            layer.weight = Parameter(qweight.t(), requires_grad=False)
            layer.weight_scale = Parameter(weight_scale, requires_grad=False)
            layer.logical_widths = None
            layer.input_scale = None
        else:
            # fp8 checkpoint: requantise the logical shards to the max of their scales
            max_w_scale = layer.weight_scale.max()
            unfused = layer.weight_scale[-1] > torch.finfo(torch.float8_e4m3fn).min
            if unfused:
                start = 0
                for idx, logical_width in enumerate(layer.logical_widths):
                    end = start + logical_width
                    weight_dq = per_tensor_dequantize(layer.weight[start:end, :], layer.weight_scale[idx])
                    layer.weight[start:end, :] = per_tensor_quantize(weight_dq, layer.weight_scale.max())
                    start = end
            layer.weight_scale = Parameter(max_w_scale, requires_grad=False)
            layer.weight = Parameter(layer.weight.t(), requires_grad=False)
            if self.quant_config.activation_scheme == "dynamic":
                layer.input_scale = None
            else:
                layer.input_scale = Parameter(layer.input_scale.max(), requires_grad=False)
        if self.use_marlin:
            self.prepare_layer_for_marlin(layer)
        layer.process_after_load = False

    def apply(self, layer, x, bias=None):
        if self.use_marlin:
            reshaped_x = x.reshape(-1, x.shape[-1])
            out_shape = x.shape[:-1] + (layer.output_size_per_partition, )
            output = ops.fp8_marlin_gemm(reshaped_x, layer.weight, layer.weight_scale,
                                         layer.workspace, 8, reshaped_x.shape[0],
                                         layer.output_size_per_partition,
                                         layer.input_size_per_partition)
            if bias is not None:
                output.add_(bias)
            return output.reshape(out_shape)
        x2 = x.reshape(-1, x.shape[-1])
        qinput, x_scale = ops.scaled_fp8_quant(x2, layer.input_scale)
        output = ops.cutlass_scaled_mm(qinput, layer.weight, scale_a=x_scale,
                                       scale_b=layer.weight_scale, out_dtype=x.dtype, bias=bias)
        return output.reshape(x.shape[:-1] + (output.shape[-1], ))


class Fp8KVCacheMethod(QuantizeMethodBase):
    """kv-cache scaling factor of fp8 checkpoints (fp8.py:563-598).  gfx950 stores OCP e4m3fn, so
    the scale is used as is (no x2 fnuz correction, reference llama.py:503-508)."""

    def __init__(self, quant_config: Fp8Config):
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module):
        layer.kv_scale = Parameter(torch.tensor(1.0), requires_grad=False)

    def apply(self, layer: torch.nn.Module) -> torch.Tensor:
        raise RuntimeError("Fp8KVCacheMethod.apply should not be called.")

    def process_weights_after_loading(self, layer: Module) -> None:
        if layer.kv_cache_dtype != "auto":
            kv_scale = layer.kv_scale.to("cpu").tolist()
            if not isinstance(kv_scale, float):
                raise ValueError("Only support per-tensor scaling factor for fp8 KV cache")
            layer._kv_scale = kv_scale
        del layer.kv_scale


def per_tensor_quantize(tensor: torch.Tensor, inv_scale: Union[float, torch.Tensor]) -> torch.Tensor:
    finfo = torch.finfo(torch.float8_e4m3fn)
    return (tensor / inv_scale).clamp(min=finfo.min, max=finfo.max).to(torch.float8_e4m3fn)


def per_tensor_dequantize(tensor: torch.Tensor, inv_scale: Union[float, torch.Tensor]) -> torch.Tensor:
    return tensor.to(torch.float16) * inv_scale
