"""GPTQ-Marlin (W4A16 / W8A16) quantisation method on the gfx950 kernels.

Plugin-surface mirror of vllm/model_executor/layers/quantization/gptq_marlin.py: GPTQMarlinConfig
(:59-184), GPTQMarlinLinearMethod (:192-466) -- same parameter names / shapes / sharding
attributes (qweight int32 [K/pack, N], g_idx int32 [K], scales [K/g, N], qzeros on `meta`,
workspace int32 [N/64*16]) and the same lazy REPACK -> READY state machine on the first apply().
The two native ops it calls, ops.gptq_marlin_repack and ops.gptq_marlin_gemm, are the HIP
kernels of csrc/w4a16_gemm.hip.
"""
import enum
from enum import Enum
from typing import Any, Dict, List, Optional

import torch
from torch.nn.parameter import Parameter

from .... import _custom_ops as ops
from ....platforms import current_platform
from ...utils import set_weight_attrs
from .base_config import LinearMethodBase, QuantizationConfig

GPTQ_MARLIN_TILE = 16
GPTQ_MARLIN_MIN_THREAD_N = 64
GPTQ_MARLIN_MIN_THREAD_K = 128
GPTQ_MARLIN_MAX_PARALLEL = 16

GPTQ_MARLIN_SUPPORTED_NUM_BITS = [4, 8]
GPTQ_MARLIN_SUPPORTED_GROUP_SIZES = [-1, 32, 64, 128]
GPTQ_MARLIN_SUPPORTED_SYM = [True]


def get_scale_perms(num_bits: int):
    """gptq_marlin.py:26-35: where each output column's scale sits inside a 64- / 32-wide run"""
    scale_perm = [i + 8 * j for i in range(8) for j in range(8)]
    scale_perm_single = [2 * i + j for i in range(4) for j in [0, 1, 8, 9, 16, 17, 24, 25]]
    return scale_perm, scale_perm_single


def get_pack_factor(num_bits: int):
    assert num_bits in GPTQ_MARLIN_SUPPORTED_NUM_BITS, f"Unsupported num_bits = {num_bits}"
    return 32 // num_bits


def marlin_permute_scales(s: torch.Tensor, size_k: int, size_n: int, group_size: int,
                          num_bits: int):
    """gptq_marlin.py:47-56"""
    scale_perm, scale_perm_single = get_scale_perms(num_bits)
    if group_size < size_k and group_size != -1:
        s = s.reshape((-1, len(scale_perm)))[:, scale_perm]
    else:
        s = s.reshape((-1, len(scale_perm_single)))[:, scale_perm_single]
    return s.reshape((-1, size_n)).contiguous()


class GPTQMarlinConfig(QuantizationConfig):
    """Config class for GPTQ Marlin"""

    def __init__(self, weight_bits: int, group_size: int, desc_act: bool, is_sym: bool,
                 lm_head_quantized: bool) -> None:
        if desc_act and group_size == -1:
            # one group per output channel: act_order is a no-op
            desc_act = False
        self.weight_bits = weight_bits
        self.group_size = group_size
        self.desc_act = desc_act
        self.is_sym = is_sym
        self.lm_head_quantized = lm_head_quantized
        if self.weight_bits not in GPTQ_MARLIN_SUPPORTED_NUM_BITS:
            raise ValueError(f"Marlin does not support weight_bits = {self.weight_bits}. "
                             f"Only weight_bits = {GPTQ_MARLIN_SUPPORTED_NUM_BITS} are supported.")
        if self.group_size not in GPTQ_MARLIN_SUPPORTED_GROUP_SIZES:
            raise ValueError(f"Marlin does not support group_size = {self.group_size}. "
                             f"Only group_sizes = {GPTQ_MARLIN_SUPPORTED_GROUP_SIZES} are supported.")
        if self.is_sym not in GPTQ_MARLIN_SUPPORTED_SYM:
            raise ValueError(f"Marlin does not support is_sym = {self.is_sym}. "
                             f"Only sym = {GPTQ_MARLIN_SUPPORTED_SYM} are supported.")
        self.pack_factor = get_pack_factor(weight_bits)
        self.tile_size = GPTQ_MARLIN_TILE
        self.min_thread_n = GPTQ_MARLIN_MIN_THREAD_N
        self.min_thread_k = GPTQ_MARLIN_MIN_THREAD_K
        self.max_parallel = GPTQ_MARLIN_MAX_PARALLEL

    def __repr__(self) -> str:
        return (f"GPTQMarlinConfig(weight_bits={self.weight_bits}, group_size={self.group_size}, "
                f"desc_act={self.desc_act}, lm_head_quantized={self.lm_head_quantized})")

    @classmethod
    def get_name(cls) -> str:
        return "gptq_marlin"

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.half, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 80  # gfx950 reports 95

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return ["quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "GPTQMarlinConfig":
        weight_bits = cls.get_from_keys(config, ["bits"])
        group_size = cls.get_from_keys(config, ["group_size"])
        desc_act = cls.get_from_keys(config, ["desc_act"])
        is_sym = cls.get_from_keys(config, ["sym"])
        lm_head_quantized = cls.get_from_keys_or(config, ["lm_head"], default=False)
        return cls(weight_bits, group_size, desc_act, is_sym, lm_head_quantized)

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant) -> Optional[str]:
        can_convert = cls.is_marlin_compatible(hf_quant_cfg)
        is_valid_user_quant = (user_quant is None or user_quant == "marlin")
        if can_convert and is_valid_user_quant:
            return cls.get_name()
        return None

    def get_quant_method(self, layer: torch.nn.Module) -> Optional["GPTQMarlinLinearMethod"]:
        from ..linear import LinearBase
        from ..vocab_parallel_embedding import ParallelLMHead
        if isinstance(layer, LinearBase) or (isinstance(layer, ParallelLMHead)
                                             and self.lm_head_quantized):
            return GPTQMarlinLinearMethod(self)
        return None

    def get_scaled_act_names(self) -> List[str]:
        return []

    @classmethod
    def is_marlin_compatible(cls, quant_config: Dict[str, Any]):
        num_bits = quant_config.get("bits", None)
        group_size = quant_config.get("group_size", None)
        sym = quant_config.get("sym", None)
        desc_act = quant_config.get("desc_act", None)
        if num_bits is None or group_size is None or sym is None or desc_act is None:
            return False
        major, minor = current_platform.get_device_capability()
        if major * 10 + minor < cls.get_min_capability():
            return False
        return (num_bits in GPTQ_MARLIN_SUPPORTED_NUM_BITS
                and group_size in GPTQ_MARLIN_SUPPORTED_GROUP_SIZES
                and sym in GPTQ_MARLIN_SUPPORTED_SYM)


class GPTQMarlinState(Enum):
    REPACK = enum.auto()
    READY = enum.auto()


class GPTQMarlinLinearMethod(LinearMethodBase):
    """Linear method for GPTQ Marlin."""

    def __init__(self, quant_config: GPTQMarlinConfig) -> None:
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int, output_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs) -> None:
        del output_size
        cfg = self.quant_config
        group_size = cfg.group_size if cfg.group_size != -1 else input_size
        if params_dtype not in [torch.float16, torch.bfloat16]:
            raise ValueError(f"The params dtype must be float16 or bfloat16, but got {params_dtype}")
        output_size_per_partition = sum(output_partition_sizes)
        if output_size_per_partition % cfg.min_thread_n != 0:
            raise ValueError(f"Weight output_size_per_partition = {output_size_per_partition} is "
                             f"not divisible by  min_thread_n = {cfg.min_thread_n}.")
        if input_size_per_partition % cfg.min_thread_k != 0:
            raise ValueError(f"Weight input_size_per_partition = {input_size_per_partition} is "
                             f"not divisible by min_thread_k = {cfg.min_thread_k}.")
        if group_size < input_size and input_size_per_partition % group_size != 0:
            raise ValueError(f"Weight input_size_per_partition = {input_size_per_partition} is "
                             f"not divisible by group_size = {group_size}.")
        # sharding of scales / zero points over the input dim (gptq_marlin.py:246-269)
        scales_and_zp_size = input_size // group_size
        scales_and_zp_input_dim = None
        if cfg.desc_act:
            assert cfg.group_size != -1
            is_k_full = input_size_per_partition == input_size
        else:
            is_k_full = True
            if input_size != input_size_per_partition and cfg.group_size != -1:
                scales_and_zp_size = input_size_per_partition // group_size
                scales_and_zp_input_dim = 0

        qweight = Parameter(torch.empty(input_size_per_partition // cfg.pack_factor,
                                        output_size_per_partition, dtype=torch.int32),
                            requires_grad=False)
        set_weight_attrs(qweight, {**extra_weight_attrs, "input_dim": 0, "output_dim": 1,
                                   "packed_dim": 0, "pack_factor": cfg.pack_factor})
        g_idx = Parameter(torch.empty(input_size_per_partition, dtype=torch.int32),
                          requires_grad=False)
        set_weight_attrs(g_idx, {**extra_weight_attrs, "input_dim": 0, "ignore_warning": True})
        g_idx_sort_indices = torch.empty(g_idx.shape, dtype=torch.int32)
        scales = Parameter(torch.empty(scales_and_zp_size, output_size_per_partition,
                                       dtype=params_dtype), requires_grad=False)
        set_weight_attrs(scales, {**extra_weight_attrs, "input_dim": scales_and_zp_input_dim,
                                  "output_dim": 1})
        qzeros = Parameter(torch.empty(scales_and_zp_size,
                                       output_size_per_partition // cfg.pack_factor,
                                       dtype=torch.int32, device="meta"), requires_grad=False)
        set_weight_attrs(qzeros, {**extra_weight_attrs, "input_dim": scales_and_zp_input_dim,
                                  "output_dim": 1, "packed_dim": 1,
                                  "pack_factor": cfg.pack_factor})
        max_workspace_size = (output_size_per_partition // cfg.min_thread_n) * cfg.max_parallel
        workspace = torch.zeros(max_workspace_size, dtype=torch.int, requires_grad=False)

        layer.register_parameter("qweight", qweight)
        layer.register_parameter("g_idx", g_idx)
        layer.register_parameter("scales", scales)
        layer.register_parameter("qzeros", qzeros)
        layer.g_idx_sort_indices = g_idx_sort_indices
        layer.workspace = workspace
        layer.input_size_per_partition = input_size_per_partition
        layer.output_size_per_partition = output_size_per_partition
        layer.input_size = input_size
        layer.is_k_full = is_k_full
        layer.marlin_state = GPTQMarlinState.REPACK

    def _repack(self, layer: torch.nn.Module) -> None:
        """first-touch GPTQ -> Marlin conversion (gptq_marlin.py:389-447)"""
        cfg = self.quant_config
        part_size_n = layer.output_size_per_partition
        part_size_k = layer.input_size_per_partition

        def replace_tensor(name, new_t):
            # resize_ + copy_ keep the registered parameter's storage
            getattr(layer, name).resize_(new_t.shape)
            getattr(layer, name).copy_(new_t)
            del new_t

        cur_device = layer.qweight.device
        if layer.workspace.device != cur_device:
            layer.workspace = layer.workspace.to(cur_device)
        if cfg.desc_act:
            g_idx_sort_indices = torch.argsort(layer.g_idx).to(torch.int)
            sorted_g_idx = layer.g_idx[g_idx_sort_indices]
            layer.g_idx_sort_indices = layer.g_idx_sort_indices.to(cur_device)
            replace_tensor("g_idx", sorted_g_idx)
            replace_tensor("g_idx_sort_indices", g_idx_sort_indices)
        else:
            layer.g_idx = Parameter(torch.empty(0, dtype=torch.int, device=cur_device),
                                    requires_grad=False)
            layer.g_idx_sort_indices = Parameter(torch.empty(0, dtype=torch.int, device=cur_device),
                                                 requires_grad=False)
        # decode-sized calls run on the MFMA-native tensor (csrc/w4a16_gemm.hip): built here from the same GPTQ
        # words, beside the Marlin tensor that the reference's op (and prefill) consumes
        if self.native_eligible(layer):
            layer.qweight_native = ops.w4_native_repack(layer.qweight.data, None, part_size_k, part_size_n)
            layer.scales_native = layer.scales.data.clone()      # natural [groups, N]
        marlin_qweight = ops.gptq_marlin_repack(layer.qweight, layer.g_idx_sort_indices,
                                                part_size_k, part_size_n, cfg.weight_bits)
        replace_tensor("qweight", marlin_qweight)
        scales_size_k = layer.input_size if cfg.desc_act else part_size_k
        marlin_scales = marlin_permute_scales(layer.scales, scales_size_k, part_size_n,
                                              cfg.group_size, cfg.weight_bits)
        replace_tensor("scales", marlin_scales)

    NATIVE_MAX_M = 64   # rows per call up to which the native kernel is used (its M tiles end at 64 rows)

    def native_eligible(self, layer: torch.nn.Module) -> bool:
        import os
        cfg = self.quant_config
        return (os.environ.get("NMV_W4_NATIVE", "0") == "1" and cfg.weight_bits == 4 and not cfg.desc_act
                and cfg.group_size in (-1, 128) and layer.input_size_per_partition % 256 == 0
                and layer.output_size_per_partition % 64 == 0 and layer.is_k_full and layer.qweight.is_cuda)

    @staticmethod
    def _native(layer, size_m: int) -> bool:
        return size_m <= GPTQMarlinLinearMethod.NATIVE_MAX_M and getattr(layer, "qweight_native", None) is not None

    # ---- deferred split-K: the GEMM leaves fp32 slabs, the following fused_add_rms_norm sums them
    # (ops.gptq_marlin_gemm_partial; not part of the reference's LinearMethod) ----
    def can_defer_reduce(self, layer: torch.nn.Module) -> bool:
        cfg = self.quant_config
        return (cfg.weight_bits == 4 and not cfg.desc_act and cfg.group_size in (-1, 128)
                and layer.input_size_per_partition % 256 == 0 and layer.is_k_full
                and not getattr(layer, "gate_up_interleaved", False))

    def apply_partial(self, layer: torch.nn.Module, x: torch.Tensor) -> torch.Tensor:
        """x @ W as fp32 split-K slabs [splits, T, N]; their sum in split order, rounded to the model
        dtype, is bit-identical to apply()"""
        reshaped_x = x.reshape(-1, x.shape[-1])
        if layer.marlin_state == GPTQMarlinState.REPACK:
            layer.marlin_state = GPTQMarlinState.READY
            self._repack(layer)
        if self._native(layer, reshaped_x.shape[0]):
            return ops.w4_native_gemm(reshaped_x, layer.qweight_native, layer.scales_native, None, reshaped_x.shape[0],
                                      layer.output_size_per_partition, layer.input_size_per_partition, mode=2)
        return ops.gptq_marlin_gemm_partial(reshaped_x, layer.qweight, layer.scales, reshaped_x.shape[0],
                                            layer.output_size_per_partition, layer.input_size_per_partition)

    # ---- gate_up with silu_and_mul folded into the GEMM epilogue (ops.gptq_marlin_gemm_silu_mul;
    # not part of the reference's LinearMethod) ----
    def can_fuse_silu_mul(self, layer: torch.nn.Module) -> bool:
        """a merged [gate | up] projection whose columns can be interleaved per 64-column chunk before
        the (first-touch) Marlin repack, and which is wide enough that running the GEMM without
        split-K (the epilogue needs the whole K in one workgroup) still beats GEMM + silu_and_mul:
        measured on MI355X at K = 4096, the fused form wins from 112 chunks (N = 7168, the TP = 4
        shard of Llama-3-8B) upwards and loses at 56 (TP = 8)"""
        cfg = self.quant_config
        n, k = layer.output_size_per_partition, layer.input_size_per_partition
        if getattr(layer, "gate_up_interleaved", False):
            return True
        return (layer.marlin_state == GPTQMarlinState.REPACK and cfg.weight_bits == 4 and not cfg.desc_act
                and cfg.group_size in (-1, 128) and k % 256 == 0 and n % 128 == 0 and n // 64 >= 112
                and getattr(layer, "bias", None) is None and layer.is_k_full)

    @staticmethod
    def _interleave_gate_up(t: torch.Tensor) -> torch.Tensor:
        """columns [gate 0..I-1 | up 0..I-1] -> per 64-column chunk c: [gate 32c..32c+31 | up 32c..32c+31]"""
        lead, n = t.shape[:-1], t.shape[-1]
        return t.reshape(*lead, 2, n // 64, 32).transpose(-3, -2).reshape(*lead, n).contiguous()

    def apply_silu_mul(self, layer: torch.nn.Module, x: torch.Tensor) -> torch.Tensor:
        """silu(x @ W_gate) * (x @ W_up) -> [.., N/2], bit-identical to apply() + SiluAndMul"""
        reshaped_x = x.reshape(-1, x.shape[-1])
        part_size_n = layer.output_size_per_partition
        part_size_k = layer.input_size_per_partition
        if layer.marlin_state == GPTQMarlinState.REPACK:
            assert self.can_fuse_silu_mul(layer)
            layer.marlin_state = GPTQMarlinState.READY
            layer.qweight.data = self._interleave_gate_up(layer.qweight.data)
            layer.scales.data = self._interleave_gate_up(layer.scales.data)
            layer.gate_up_interleaved = True
            self._repack(layer)
        assert getattr(layer, "gate_up_interleaved", False), "layer was repacked without the interleave"
        if self._native(layer, reshaped_x.shape[0]):
            out = ops.w4_native_gemm(reshaped_x, layer.qweight_native, layer.scales_native, layer.workspace,
                                     reshaped_x.shape[0], part_size_n, part_size_k, mode=1)
        else:
            out = ops.gptq_marlin_gemm_silu_mul(reshaped_x, layer.qweight, layer.scales, layer.workspace,
                                                reshaped_x.shape[0], part_size_n, part_size_k)
        return out.reshape(x.shape[:-1] + (part_size_n // 2, ))

    def apply(self, layer: torch.nn.Module, x: torch.Tensor,
              bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert not getattr(layer, "gate_up_interleaved", False), \
            "this layer's columns are interleaved for apply_silu_mul()"
        reshaped_x = x.reshape(-1, x.shape[-1])
        size_m = reshaped_x.shape[0]
        part_size_n = layer.output_size_per_partition
        part_size_k = layer.input_size_per_partition
        out_shape = x.shape[:-1] + (part_size_n, )
        if layer.marlin_state == GPTQMarlinState.REPACK:
            layer.marlin_state = GPTQMarlinState.READY
            self._repack(layer)
        if self._native(layer, size_m):
            output = ops.w4_native_gemm(reshaped_x, layer.qweight_native, layer.scales_native, layer.workspace, size_m,
                                        part_size_n, part_size_k, mode=0)
        else:
            output = ops.gptq_marlin_gemm(reshaped_x, layer.qweight, layer.scales, layer.g_idx,
                                          layer.g_idx_sort_indices, layer.workspace,
                                          self.quant_config.weight_bits, size_m, part_size_n,
                                          part_size_k, layer.is_k_full)
        if bias is not None:
            output.add_(bias)
        return output.reshape(out_shape)

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        return
