"""GPTQ checkpoints served in their own layout (the "exllama" path; interface and checkpoint contract: reference
vllm/model_executor/layers/quantization/gptq.py:19-231; parameter table pinned by
tests/golden/linear_method_params.json["gptq"]).

Checkpoint tensors of one linear layer, `pack` = 32 / bits codes per int32 (a fraction for 3-bit):
  qweight int32 [K / pack, N]     codes packed along K
  qzeros  int32 [G, N / pack]     zero points minus one, packed along N
  scales        [G, N]            model dtype
  g_idx   int32 [K]               group of every input row (act-order checkpoints permute it)
First forward: the kernel wants the act-order permutation (argsort of g_idx) or nothing, and the codes shuffled for
it (`gptq_shuffle`); after that every call is `gptq_gemm`.  A row-parallel act-order layer cannot be shuffled per
rank and runs the kernel's plain form (`use_exllama` false), as in the reference (gptq.py:133-136).

MI355X addition (not in the reference): a 4-bit, group-128 layer without act-order is additionally repacked once to the
Marlin interchange layout and runs the tuned Marlin kernel with per-group zero points (symmetric checkpoints never get
here: GPTQMarlinConfig claims them first, as in the reference)."""
import enum
from fractions import Fraction
from typing import Any, Dict, List, Optional

import torch

from .... import _custom_ops as ops
from ._schema import Geometry, Require, Slot, build
from .base_config import LinearMethodBase, QuantizationConfig

_TP_HINT = "This can be caused by too large tensor parallel size."


class GPTQConfig(QuantizationConfig):
    """`quantize_config.json`: {"bits": 2|3|4|8, "group_size": g | -1, "desc_act": bool, "lm_head": bool}"""

    SUPPORTED_BITS = (2, 3, 4, 8)

    def __init__(self, weight_bits: int, group_size: int, desc_act: bool, lm_head_quantized: bool = False) -> None:
        if weight_bits not in self.SUPPORTED_BITS:
            raise ValueError("Currently, only 2/3/4/8-bit weight quantization is supported for GPTQ, "
                             f"but got {weight_bits} bits.")
        self.weight_bits, self.group_size, self.desc_act = weight_bits, group_size, desc_act
        self.lm_head_quantized = lm_head_quantized
        self.pack_factor = Fraction(32, weight_bits)

    def __repr__(self) -> str:
        return (f"GPTQConfig(weight_bits={self.weight_bits}, group_size={self.group_size}, desc_act={self.desc_act}), "
                f"lm_head_quantized={self.lm_head_quantized}")

    @classmethod
    def get_name(cls) -> str:
        return "gptq"

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.half, torch.bfloat16]   # the HIP kernels take both (the reference's CUDA kernel: half only)

    @classmethod
    def get_min_capability(cls) -> int:
        return 60

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return ["quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "GPTQConfig":
        bits, group, desc = (cls.get_from_keys(config, [k]) for k in ("bits", "group_size", "desc_act"))
        return cls(bits, group, desc, cls.get_from_keys_or(config, ["lm_head"], default=False))

    def get_quant_method(self, layer: torch.nn.Module) -> Optional["GPTQLinearMethod"]:
        from ..linear import LinearBase
        from ..vocab_parallel_embedding import ParallelLMHead
        takes = isinstance(layer, LinearBase) or (self.lm_head_quantized and isinstance(layer, ParallelLMHead))
        return GPTQLinearMethod(self) if takes else None

    def get_scaled_act_names(self) -> List[str]:
        return []

    # ---- the schema -----------------------------------------------------------------------------------------
    def _group(self, g: Geometry) -> int:
        return g.k_all if self.group_size == -1 else self.group_size

    def _act_order_across_ranks(self, g: Geometry) -> bool:
        return g.row_sharded and self.group_size != -1 and self.desc_act

    def _groups_here(self, g: Geometry) -> int:
        """rows of qzeros / scales on this rank: a row-parallel shard holds its own groups -- unless act-order, whose
        g_idx may point at any group, keeps them all"""
        sliced = g.row_sharded and self.group_size != -1 and not self.desc_act
        return (g.k if sliced else g.k_all) // self._group(g)

    def requirements(self) -> List[Require]:
        return [
            Require(lambda g: g.k % self.group_size == 0,
                    lambda g: "The input size is not aligned with the quantized weight shape. " + _TP_HINT),
            Require(lambda g: g.n % self.pack_factor.numerator == 0,
                    lambda g: "The output size is not aligned with the quantized weight shape. " + _TP_HINT),
        ]

    def slots(self) -> List[Slot]:
        pack = self.pack_factor

        def group_axis(g: Geometry) -> Optional[int]:
            sliced = g.row_sharded and self.group_size != -1 and not self.desc_act
            return 0 if sliced else None

        return [
            Slot("qweight", lambda g: (int(g.k // pack), g.n), torch.int32,
                 lambda g: {"input_dim": 0, "output_dim": 1, "packed_dim": 0, "pack_factor": pack}),
            Slot("g_idx", lambda g: (g.k, ), torch.int32, lambda g: {"input_dim": 0, "ignore_warning": True},
                 init=lambda g: torch.arange(g.k, dtype=torch.int32) // self.group_size),
            Slot("qzeros", lambda g: (self._groups_here(g), int(g.n // pack)), torch.int32,
                 lambda g: {"input_dim": group_axis(g), "output_dim": 1, "packed_dim": 1, "pack_factor": pack}),
            Slot("scales", lambda g: (self._groups_here(g), g.n), lambda g: g.dtype,
                 lambda g: {"input_dim": group_axis(g), "output_dim": 1}),
        ]


class ExllamaState(enum.Enum):
    UNUSED = enum.auto()           # the kernel's plain form, no shuffle
    UNINITIALIZED = enum.auto()    # shuffle pending (first forward)
    READY = enum.auto()


_UNPACK_ORDER = [i + 8 * j for i in range(8) for j in range(8)]   # marlin_permute_scales' column order within 64


class GPTQLinearMethod(LinearMethodBase):

    def __init__(self, quant_config: GPTQConfig):
        self.quant_config = quant_config

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        cfg = self.quant_config
        g = Geometry(input_size_per_partition, tuple(output_partition_sizes), params_dtype, k_total=input_size)
        build(layer, g, cfg.requirements(), cfg.slots(), extra_weight_attrs)
        layer.exllama_state = ExllamaState.UNUSED if cfg._act_order_across_ranks(g) else ExllamaState.UNINITIALIZED

    # ---- Marlin route for 4-bit / group 128 / no act-order (see the module docstring) ------------------------------
    def _marlin_eligible(self, layer) -> bool:
        cfg = self.quant_config
        k, n = layer.qweight.shape[0] * int(cfg.pack_factor), layer.qweight.shape[1]
        return (cfg.weight_bits == 4 and cfg.group_size == 128 and not cfg.desc_act and k % 256 == 0 and n % 64 == 0
                and layer.exllama_state is ExllamaState.UNINITIALIZED and layer.qweight.device.type == "cuda"
                and layer.scales.dtype in (torch.half, torch.bfloat16) and layer.scales.shape[0] == k // 128)

    def process_weights_after_loading(self, layer) -> None:
        if not self._marlin_eligible(layer):
            return
        k, n = layer.qweight.shape[0] * 8, layer.qweight.shape[1]
        dev = layer.qweight.device
        nibble = torch.arange(0, 32, 4, dtype=torch.int32, device=dev)
        zeros = ((layer.qzeros.data.unsqueeze(-1) >> nibble) & 0xF) + 1        # [G, N / 8, 8]: stored minus one
        order = torch.tensor(_UNPACK_ORDER, device=dev)

        def marlin_columns(t: torch.Tensor) -> torch.Tensor:
            return t.reshape(-1, 64)[:, order].reshape(-1, n).contiguous()

        layer.marlin_qweight = ops.gptq_marlin_repack(layer.qweight.data.contiguous(),
                                                      torch.empty(0, dtype=torch.int32, device=dev), k, n, 4)
        layer.marlin_scales = marlin_columns(layer.scales.data)
        layer.marlin_zeros = marlin_columns(zeros.reshape(zeros.shape[0], -1).to(layer.scales.dtype))
        layer.marlin_workspace = torch.zeros(n // 64 * 16, dtype=torch.int32, device=dev)
        layer.gptq_marlin_kn = (k, n)

    # ---- forward ----------------------------------------------------------------------------------------------
    def _first_forward(self, layer) -> None:
        """g_idx becomes what the kernel wants (the act-order permutation, or nothing) and the codes are shuffled"""
        if self.quant_config.desc_act:
            layer.g_idx.data = torch.argsort(layer.g_idx).to(torch.int)
        else:
            layer.g_idx.data = torch.empty((0, ), dtype=torch.int, device=layer.g_idx.device)
        layer.exllama_state = ExllamaState.READY
        ops.gptq_shuffle(layer.qweight, layer.g_idx, self.quant_config.weight_bits)

    def apply(self, layer, x, bias=None):
        rows = x.reshape(-1, x.shape[-1])
        marlin_kn = getattr(layer, "gptq_marlin_kn", None)
        if marlin_kn is not None:
            k, n = marlin_kn
            y = ops.marlin_zp_gemm(rows, layer.marlin_qweight, layer.marlin_scales, layer.marlin_zeros,
                                   layer.marlin_workspace, rows.shape[0], n, k)
        else:
            if layer.exllama_state is ExllamaState.UNINITIALIZED:
                self._first_forward(layer)
            n = layer.qweight.shape[-1]
            y = ops.gptq_gemm(rows, layer.qweight, layer.qzeros, layer.scales, layer.g_idx,
                              layer.exllama_state is ExllamaState.READY, self.quant_config.weight_bits)
        if bias is not None:
            y.add_(bias)
        return y.reshape(x.shape[:-1] + (n, ))
