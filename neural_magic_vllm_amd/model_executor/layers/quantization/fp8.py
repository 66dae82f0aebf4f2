"""FP8 (OCP e4m3fn) weights and activations with per-tensor scales.

Interface and checkpoint contract: reference vllm/model_executor/layers/quantization/fp8.py:22-379 (config,
linear method) and :563-613 (KV-cache scale, quantise / dequantise helpers); parameter tables pinned by
tests/golden/linear_method_params.json["fp8_static" | "fp8_dynamic"].

gfx950 reports capability 9.5, so the weight-only Marlin detour the reference takes below capability 8.9 is never
selected here, and `cutlass_scaled_mm_supports_fp8` answers true: a linear is
    scaled_fp8_quant(x, input_scale | dynamic) -> cutlass_scaled_mm(xq, Wq^T, x_scale, w_scale, bias)
on the native fp8 MFMA, bias in the epilogue (the reference leaves a biased layer to torch._scaled_mm,
fp8.py:347-365).  Two checkpoint kinds reach `process_weights_after_loading`:
  * 16-bit weights (is_checkpoint_fp8_serialized = False): the fused weight is quantised as a whole;
  * fp8 weights with one scale per LOGICAL matrix (q / k / v, gate / up): every logical matrix is re-expressed on
    the largest of the scales, so that the fused GEMM needs one weight scale; static activation scales likewise
    collapse to their maximum."""
from typing import Any, Dict, List, Optional, Union

import torch
from torch.nn import Module
from torch.nn.parameter import Parameter

from .... import _custom_ops as ops
from ....platforms import current_platform
from ._schema import Geometry, Slot, build
from .base_config import LinearMethodBase, QuantizationConfig, QuantizeMethodBase
from .gptq_marlin import (GPTQ_MARLIN_MAX_PARALLEL, GPTQ_MARLIN_MIN_THREAD_N, GPTQMarlinState,
                          marlin_permute_scales)

ACTIVATION_SCHEMES = ["static", "dynamic"]
_E4M3 = torch.float8_e4m3fn
_UNSET = torch.finfo(_E4M3).min      # what a scale slot holds until the loader writes it


def _capability() -> int:
    major, minor = current_platform.get_device_capability()
    return major * 10 + minor


def cutlass_fp8_supported() -> bool:
    return ops.cutlass_scaled_mm_supports_fp8(_capability())


def per_tensor_quantize(tensor: torch.Tensor, inv_scale: Union[float, torch.Tensor]) -> torch.Tensor:
    lim = torch.finfo(_E4M3)
    return (tensor / inv_scale).clamp(min=lim.min, max=lim.max).to(_E4M3)


def per_tensor_dequantize(tensor: torch.Tensor, inv_scale: Union[float, torch.Tensor]) -> torch.Tensor:
    return tensor.to(torch.float16) * inv_scale


def pack_fp8_to_int32(fp8_tensor: torch.Tensor) -> torch.Tensor:
    """[K, N] fp8 -> int32 [K/4, N], byte i of a word = row 4r + i (the 8-bit GPTQ packing the Marlin repack
    takes; reference marlin_utils.py:227-247)"""
    assert fp8_tensor.dtype == _E4M3 and fp8_tensor.shape[0] % 4 == 0
    quads = fp8_tensor.view(torch.uint8).reshape(fp8_tensor.shape[0] // 4, 4, *fp8_tensor.shape[1:]).to(torch.int32)
    word = quads[:, 0]
    for i in (1, 2, 3):
        word = word | (quads[:, i] << (8 * i))
    return word.contiguous()


class Fp8Config(QuantizationConfig):

    def __init__(self, is_checkpoint_fp8_serialized: bool = False, activation_scheme: str = "dynamic") -> None:
        if activation_scheme not in ACTIVATION_SCHEMES:
            raise ValueError(f"Unsupported activation scheme {activation_scheme}")
        self.is_checkpoint_fp8_serialized = is_checkpoint_fp8_serialized
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
        method = cls.get_from_keys(config, ["quant_method"])
        return cls(is_checkpoint_fp8_serialized="fp8" in method,
                   activation_scheme=cls.get_from_keys(config, ["activation_scheme"]))

    def get_quant_method(self, layer: torch.nn.Module) -> Optional[QuantizeMethodBase]:
        from ....attention.layer import Attention
        from ..linear import LinearBase
        if isinstance(layer, LinearBase):
            return Fp8LinearMethod(self)
        return Fp8KVCacheMethod(self) if isinstance(layer, Attention) else None

    def get_scaled_act_names(self) -> List[str]:
        return []

    def slots(self) -> List[Slot]:
        """weight [N, K] (fp8 when the checkpoint is, else the model dtype); one fp32 scale per logical matrix for
        the weight -- and for the activations under the static scheme -- when the checkpoint is fp8"""
        serialized, static = self.is_checkpoint_fp8_serialized, self.activation_scheme == "static"
        out = [Slot("weight", lambda g: (g.n, g.k), (lambda g: _E4M3) if serialized else (lambda g: g.dtype),
                    lambda g: {"input_dim": 1, "output_dim": 0})]
        scale_names = (["weight_scale"] + (["input_scale"] if static else [])) if serialized else []
        for name in scale_names:
            out.append(Slot(name, lambda g: (len(g.parts), ), torch.float32,
                            lambda g: {"needs_scalar_to_array": True}, fill=_UNSET))
        return out


class Fp8LinearMethod(LinearMethodBase):

    def __init__(self, quant_config: Fp8Config):
        self.quant_config = quant_config
        self.cutlass_fp8_supported = cutlass_fp8_supported()
        self.use_marlin = _capability() < 89

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        g = Geometry(input_size_per_partition, tuple(output_partition_sizes), params_dtype)
        # the scale slots carry their own attribute first and the loader's second; the weight the other way round is
        # indistinguishable (attribute sets, not order, are the contract)
        build(layer, g, [], self.quant_config.slots(), extra_weight_attrs)
        layer.process_after_load = True
        layer.logical_widths = output_partition_sizes
        layer.input_size_per_partition, layer.output_size_per_partition = g.k, g.n
        layer.orig_dtype = params_dtype
        if self.use_marlin:
            layer.marlin_state = GPTQMarlinState.REPACK

    # ---- after the checkpoint has been read -------------------------------------------------------------------
    def process_weights_after_loading(self, layer: Module) -> None:
        if not getattr(layer, "process_after_load", False):
            return
        finish = self._finish_fp8_checkpoint if self.quant_config.is_checkpoint_fp8_serialized else self._finish_16bit_checkpoint
        finish(layer)
        if self.use_marlin:
            self.prepare_layer_for_marlin(layer)
        layer.process_after_load = False

    @staticmethod
    def _finish_16bit_checkpoint(layer: Module) -> None:
        wq, w_scale = ops.scaled_fp8_quant(layer.weight, scale=None)      # one dynamic scale for the fused weight
        layer.weight = Parameter(wq.t(), requires_grad=False)
        layer.weight_scale = Parameter(w_scale, requires_grad=False)
        layer.logical_widths = None
        layer.input_scale = None

    def _finish_fp8_checkpoint(self, layer: Module) -> None:
        scales = layer.weight_scale
        common = scales.max()
        # a checkpoint that stored the fused module as ONE matrix wrote only the first slot: nothing to re-express
        per_matrix = bool(scales[-1] > _UNSET)
        if per_matrix:
            row = 0
            for i, width in enumerate(layer.logical_widths):
                block = layer.weight[row:row + width, :]
                layer.weight[row:row + width, :] = per_tensor_quantize(per_tensor_dequantize(block, scales[i]), common)
                row += width
        layer.weight_scale = Parameter(common, requires_grad=False)
        layer.weight = Parameter(layer.weight.t(), requires_grad=False)
        if self.quant_config.activation_scheme == "static":
            layer.input_scale = Parameter(layer.input_scale.max(), requires_grad=False)
        else:
            layer.input_scale = None

    def prepare_layer_for_marlin(self, layer: Module) -> None:
        """weight-only fp8 in the 8-bit Marlin layout (devices without fp8 matrix cores; reference fp8.py:195-247)"""
        n, k = layer.output_size_per_partition, layer.input_size_per_partition
        dev = layer.weight.device
        no_perm = torch.empty(0, dtype=torch.int, device=dev)
        layer.weight = Parameter(ops.gptq_marlin_repack(pack_fp8_to_int32(layer.weight), no_perm, k, n, 8),
                                 requires_grad=False)
        channel_scales = layer.weight_scale.repeat(1, n).to(layer.orig_dtype).to(dev)
        layer.weight_scale = Parameter(marlin_permute_scales(channel_scales, k, n, -1, 8), requires_grad=False)
        layer.workspace = torch.zeros((n // GPTQ_MARLIN_MIN_THREAD_N) * GPTQ_MARLIN_MAX_PARALLEL, dtype=torch.int,
                                      device=dev)
        layer.marlin_state = GPTQMarlinState.READY

    # ---- forward --------------------------------------------------------------------------------------------
    def apply(self, layer, x, bias=None):
        rows = x.reshape(-1, x.shape[-1])
        if self.use_marlin:
            n = layer.output_size_per_partition
            y = ops.fp8_marlin_gemm(rows, layer.weight, layer.weight_scale, layer.workspace, 8, rows.shape[0], n,
                                    layer.input_size_per_partition)
            if bias is not None:
                y.add_(bias)
        else:
            xq, x_scale = ops.scaled_fp8_quant(rows, layer.input_scale)
            y = ops.cutlass_scaled_mm(xq, layer.weight, scale_a=x_scale, scale_b=layer.weight_scale, out_dtype=x.dtype,
                                      bias=bias)
        return y.reshape(x.shape[:-1] + (y.shape[-1], ))


class Fp8KVCacheMethod(QuantizeMethodBase):
    """the `kv_scale` of an fp8 checkpoint (reference fp8.py:563-598).  The cache holds OCP e4m3fn bytes on gfx950,
    so the checkpoint's scale applies as it is -- no factor 2 as for the fnuz format of MI300
    (reference models/llama.py:503-508)."""

    def __init__(self, quant_config: Fp8Config):
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module):
        layer.kv_scale = Parameter(torch.tensor(1.0), requires_grad=False)

    def apply(self, layer: torch.nn.Module) -> torch.Tensor:
        raise RuntimeError("Fp8KVCacheMethod.apply should not be called.")

    def process_weights_after_loading(self, layer: Module) -> None:
        if layer.kv_cache_dtype != "auto":
            value = layer.kv_scale.to("cpu").tolist()
            if not isinstance(value, float):
                raise ValueError("Only support per-tensor scaling factor for fp8 KV cache")
            layer._kv_scale = value
        del layer.kv_scale
