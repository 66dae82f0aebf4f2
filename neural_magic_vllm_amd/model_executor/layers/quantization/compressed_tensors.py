"""compressed-tensors checkpoints (reference: quantization/compressed_tensors/compressed_tensors.py,
schemes/compressed_tensors_w8a8.py:14-109, schemes/compressed_tensors_wNa16.py:20-175).

Two schemes on the hot path:
  * W8A8 int8: weight int8 [N, K] (+ per-tensor or per-channel fp32 scale), static per-tensor or
    dynamic per-token activation quantisation -> ops.scaled_int8_quant + ops.cutlass_scaled_mm;
  * WNA16: weight_packed int32 [N, K/pack] + weight_scale [N, K/g] -> transposed to the GPTQ
    layout on first use, ops.gptq_marlin_repack, then ops.gptq_marlin_gemm.
The 2:4-sparse w4a16 scheme of the reference is out of scope (no sparse MFMA path in north_star)."""
from typing import Any, Callable, Dict, List, NamedTuple, Optional

import torch
from torch.nn.parameter import Parameter

from .... import _custom_ops as ops
from ...utils import set_weight_attrs
from .base_config import LinearMethodBase, QuantizationConfig
from .gptq_marlin import (GPTQ_MARLIN_MAX_PARALLEL, GPTQ_MARLIN_MIN_THREAD_N, GPTQMarlinState,
                          marlin_permute_scales)


class Int8Activations(NamedTuple):
    """Activations already quantised per token by a fused producer (ops.rms_norm_dynamic_int8_quant,
    ops.silu_and_mul_dynamic_int8_quant): what ops.scaled_int8_quant(x, None) would have returned."""
    q: torch.Tensor      # int8 [T, K]
    scale: torch.Tensor  # float32 [T, 1]
    dtype: torch.dtype   # the model dtype the GEMM writes


def accepts_int8_activations(linear) -> bool:
    """True for a W8A8 linear with dynamic per-token activation quantisation"""
    scheme = getattr(linear, "scheme", None)
    return isinstance(scheme, CompressedTensorsW8A8) and not scheme.is_static_input_scheme


class CompressedTensorsW8A8:

    def __init__(self, strategy: str, is_static_input_scheme: bool):
        self.strategy = strategy  # "tensor" | "channel"
        self.is_static_input_scheme = is_static_input_scheme

    def create_weights(self, layer, output_partition_sizes: List[int], input_size_per_partition: int,
                       params_dtype: torch.dtype, weight_loader: Callable, **kwargs):
        self.logical_widths = output_partition_sizes
        shape = (sum(output_partition_sizes), 1) if self.strategy == "channel" \
            else (len(output_partition_sizes), )
        weight_scale = Parameter(torch.empty(*shape, dtype=torch.float32), requires_grad=False)
        layer.register_parameter("weight_scale", weight_scale)
        if self.strategy == "channel":
            set_weight_attrs(weight_scale, {"weight_loader": weight_loader, "output_dim": 0})
        else:
            set_weight_attrs(weight_scale, {"weight_loader": weight_loader,
                                            "needs_scalar_to_array": True})
        weight = Parameter(torch.empty(sum(output_partition_sizes), input_size_per_partition,
                                       dtype=torch.int8), requires_grad=False)
        layer.register_parameter("weight", weight)
        set_weight_attrs(weight, {"input_dim": 1, "output_dim": 0, "weight_loader": weight_loader})
        if self.is_static_input_scheme:
            input_scale = Parameter(torch.empty(1, dtype=torch.float32), requires_grad=False)
            layer.register_parameter("input_scale", input_scale)
            set_weight_attrs(input_scale, {"weight_loader": weight_loader, "ignore_warning": True})
        else:
            layer.input_scale = None

    def process_weights_after_loading(self, layer) -> None:
        # fused module with per-tensor scales: expand to per-channel (cutlass has no per-shard mode)
        if self.strategy == "tensor" and len(self.logical_widths) > 1:
            ws = torch.empty((sum(self.logical_widths), 1), dtype=torch.float32,
                             device=layer.weight_scale.device)
            start = 0
            for idx, width in enumerate(self.logical_widths):
                ws[start:start + width, :] = layer.weight_scale[idx]
                start += width
            layer.weight_scale = Parameter(ws, requires_grad=False)
        layer.weight = Parameter(layer.weight.t(), requires_grad=False)  # column-major B

    def apply_weights(self, layer, x):
        if isinstance(x, Int8Activations):
            assert not self.is_static_input_scheme
            x_q, x_scale, out_dtype, lead = x.q.reshape(-1, x.q.shape[-1]), x.scale, x.dtype, x.q.shape[:-1]
        else:
            x2 = x.reshape(-1, x.shape[-1])
            x_q, x_scale = ops.scaled_int8_quant(x2, layer.input_scale)
            out_dtype, lead = x.dtype, x.shape[:-1]
        out = ops.cutlass_scaled_mm(x_q, layer.weight, scale_a=x_scale, scale_b=layer.weight_scale,
                                    out_dtype=out_dtype)
        return out.reshape(lead + (out.shape[-1], ))


class CompressedTensorsWNA16:

    def __init__(self, strategy: str, num_bits: int, group_size: Optional[int] = None):
        self.num_bits = num_bits
        self.strategy = strategy
        self.group_size = group_size
        if self.strategy == "group" and self.group_size is None:
            raise ValueError("group_size must be given when using strategy group")

    def create_weights(self, layer, input_size: int, output_partition_sizes: List[int],
                       input_size_per_partition: int, params_dtype: torch.dtype,
                       weight_loader: Callable, **kwargs):
        pack_factor = 32 // self.num_bits
        output_size_per_partition = sum(output_partition_sizes)
        group_size = self.group_size if self.group_size is not None else input_size
        weight_scale_dim = None
        scales_and_zp_size = input_size // group_size
        if input_size != input_size_per_partition and self.group_size is not None:
            weight_scale_dim = 1
            scales_and_zp_size = input_size_per_partition // group_size
        weight = Parameter(torch.empty(output_size_per_partition,
                                       input_size_per_partition // pack_factor, dtype=torch.int32),
                           requires_grad=False)
        set_weight_attrs(weight, {"input_dim": 1, "output_dim": 0, "packed_dim": 1,
                                  "pack_factor": pack_factor, "weight_loader": weight_loader})
        layer.register_parameter("weight_packed", weight)
        weight_scale = Parameter(torch.empty(output_size_per_partition, scales_and_zp_size,
                                             dtype=params_dtype), requires_grad=False)
        set_weight_attrs(weight_scale, {"weight_loader": weight_loader,
                                        "input_dim": weight_scale_dim, "output_dim": 0})
        layer.register_parameter("weight_scale", weight_scale)
        weight_shape = Parameter(torch.empty(2, dtype=torch.int64), requires_grad=False)
        layer.register_parameter("weight_shape", weight_shape)
        set_weight_attrs(weight_shape, {"weight_loader": weight_loader, "ignore_warning": True})
        layer.input_size_per_partition = input_size_per_partition
        layer.output_size_per_partition = output_size_per_partition
        layer.input_size = input_size
        layer.marlin_state = GPTQMarlinState.REPACK
        layer.is_k_full = True
        layer.group_size = group_size
        layer.workspace = torch.zeros((output_size_per_partition // GPTQ_MARLIN_MIN_THREAD_N) *
                                      GPTQ_MARLIN_MAX_PARALLEL, dtype=torch.int)

    def process_weights_after_loading(self, layer) -> None:
        pass

    def apply_weights(self, layer, x: torch.Tensor):
        reshaped_x = x.reshape(-1, x.shape[-1])
        size_m = reshaped_x.shape[0]
        part_size_n, part_size_k = layer.output_size_per_partition, layer.input_size_per_partition
        out_shape = x.shape[:-1] + (part_size_n, )
        if layer.marlin_state == GPTQMarlinState.REPACK:
            layer.marlin_state = GPTQMarlinState.READY
            dev = layer.weight_packed.device
            layer.workspace = layer.workspace.to(dev)
            layer.g_idx = Parameter(torch.empty(0, dtype=torch.int, device=dev), requires_grad=False)
            layer.g_idx_sort_indices = Parameter(torch.empty(0, dtype=torch.int, device=dev),
                                                 requires_grad=False)
            # [N, K/pack] -> GPTQ [K/pack, N] -> Marlin tiles (compressed_tensors_wNa16.py:150-166)
            marlin_qweight = ops.gptq_marlin_repack(layer.weight_packed.t().contiguous(),
                                                    layer.g_idx_sort_indices, part_size_k,
                                                    part_size_n, self.num_bits)
            layer.weight_packed = Parameter(marlin_qweight, requires_grad=False)
            scales = layer.weight_scale.squeeze().t().contiguous()
            if scales.dim() == 1:
                scales = scales.reshape(1, -1)
            layer.weight_scale = Parameter(marlin_permute_scales(scales, part_size_k, part_size_n,
                                                                 layer.group_size, self.num_bits),
                                           requires_grad=False)
        out = ops.gptq_marlin_gemm(reshaped_x, layer.weight_packed, layer.weight_scale, layer.g_idx,
                                   layer.g_idx_sort_indices, layer.workspace, self.num_bits, size_m,
                                   part_size_n, part_size_k, layer.is_k_full)
        return out.reshape(out_shape)


class CompressedTensorsConfig(QuantizationConfig):
    """Parses the `config_groups` of a compressed-tensors quantization_config for Linear targets
    (compressed_tensors.py:26-209)."""

    def __init__(self, layer_quant_details: Dict[str, Any], ignore: List[str], quant_format: Optional[str] = None):
        self.ignore = ignore
        self.layer_quant_details = layer_quant_details
        # the checkpoint's `format` selects the scheme family as in the reference (compressed_tensors.py
        # :132-160): "int-quantized" -> W8A8, "pack-quantized" -> WNA16 ("marlin-24": 2:4 sparse, not built)
        self.quant_format = quant_format

    def get_name(self) -> str:
        return "compressed_tensors"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        return [torch.float16, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 75

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return []

    def get_scaled_act_names(self) -> List[str]:
        return []

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "CompressedTensorsConfig":
        layer_quant_details: Dict[str, Any] = dict()
        ignore: List[str] = config.get("ignore", [])
        for _, quant_config in config["config_groups"].items():
            for target in quant_config.get("targets"):
                layer_quant_details[target] = {"weights": quant_config.get("weights"),
                                               "input_activations": quant_config.get("input_activations")}
        return cls(layer_quant_details=layer_quant_details, ignore=ignore, quant_format=config.get("format"))

    def _get_schema(self, weight_quant: Dict[str, Any], input_quant: Optional[Dict[str, Any]]):
        wbits, wtype = weight_quant.get("num_bits"), weight_quant.get("type", "int")
        wstrategy = weight_quant.get("strategy", "tensor")
        symmetric = weight_quant.get("symmetric", True)
        if input_quant is None:
            if self.quant_format == "pack-quantized" and wtype == "int" and wbits in (4, 8) and symmetric \
                    and wstrategy in ("group", "channel") and not weight_quant.get("dynamic", False):
                return CompressedTensorsWNA16(wstrategy, wbits, weight_quant.get("group_size"))
            raise NotImplementedError("No compressed-tensors compatible scheme was found.")
        if self.quant_format == "int-quantized" and wbits == 8 and input_quant.get("num_bits") == 8 \
                and wtype == "int" and symmetric and input_quant.get("symmetric", True) \
                and wstrategy in ("tensor", "channel"):
            dynamic = bool(input_quant.get("dynamic", False))
            if not dynamic and input_quant.get("strategy", "tensor") != "tensor":
                raise NotImplementedError("static activation scales must be per tensor")
            return CompressedTensorsW8A8(wstrategy, is_static_input_scheme=not dynamic)
        raise NotImplementedError("No compressed-tensors compatible scheme was found.")

    def get_scheme(self, layer: torch.nn.Module):
        details = self.layer_quant_details.get("Linear")
        if details is None:
            raise ValueError("compressed-tensors config has no 'Linear' target")
        return self._get_schema(details["weights"], details["input_activations"])

    def get_quant_method(self, layer: torch.nn.Module) -> Optional["CompressedTensorsLinearMethod"]:
        from ..linear import LinearBase
        return CompressedTensorsLinearMethod(self) if isinstance(layer, LinearBase) else None


class CompressedTensorsLinearMethod(LinearMethodBase):

    def __init__(self, quantization_config: CompressedTensorsConfig):
        self.quantization_config = quantization_config

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        layer.scheme.process_weights_after_loading(layer)

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size,
                       output_size, params_dtype, **extra_weight_attrs):
        weight_loader = extra_weight_attrs.get("weight_loader")
        scheme = self.quantization_config.get_scheme(layer=layer)
        scheme.create_weights(layer=layer, input_size=input_size,
                              input_size_per_partition=input_size_per_partition,
                              output_partition_sizes=output_partition_sizes,
                              output_size=output_size, params_dtype=params_dtype,
                              weight_loader=weight_loader)
        layer.scheme = scheme

    def apply(self, layer, x, bias=None):
        scheme = layer.scheme
        if scheme is None:
            raise ValueError("A scheme must be defined for each layer")
        out = scheme.apply_weights(layer, x)
        if bias is not None:
            out = out + bias
        return out
