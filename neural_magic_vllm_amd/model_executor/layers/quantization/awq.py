"""AWQ checkpoints (interface and checkpoint contract: reference vllm/model_executor/layers/quantization/awq.py:14-176;
parameter table pinned by tests/golden/linear_method_params.json["awq"]).

Checkpoint tensors of one linear layer (4-bit, 8 codes per int32, packed along N in AWQ's interleaved nibble order
0 2 4 6 1 3 5 7):
  qweight int32 [K, N / 8]      qzeros int32 [K / g, N / 8]      scales [K / g, N] (model dtype)
Forward, as the reference: from 256 tokens on dequantise once and use the library GEMM (awq.py:166-170), below that
`awq_gemm` on the checkpoint layout.

MI355X addition (not in nm-vllm 0.5.1; later vLLM calls it awq_marlin): a group-128 layer with K % 256 == 0 and
N % 64 == 0 is repacked once after loading to the Marlin interchange layout -- codes, scales and zero points -- and
runs the tuned Marlin kernel with per-group zero points, about 2.5x the decode throughput of `awq_gemm`.  What a
checkpoint loader sees (names, shapes, attributes) does not change."""
from typing import Any, Dict, List, Optional

import torch

from .... import _custom_ops as ops
from ._schema import Geometry, Require, Slot, build
from .base_config import LinearMethodBase, QuantizationConfig

_DEQUANT_FROM_TOKENS = 256     # the reference's FP16_MATMUL_HEURISTIC_CONDITION
_AWQ_NIBBLE_OF_COLUMN = [4 * (((c & 1) << 2) | (c >> 1)) for c in range(8)]   # bit offset of column c within a word
_MARLIN_COLUMN_ORDER = [i + 8 * j for i in range(8) for j in range(8)]         # marlin_permute_scales within 64 columns
_TP_HINT = "This can be caused by too large tensor parallel size."


class AWQConfig(QuantizationConfig):
    """`quant_config.json` / `quantize_config.json`: {"w_bit" | "bits": 4, "q_group_size" | "group_size": g,
    "zero_point": bool}"""

    def __init__(self, weight_bits: int, group_size: int, zero_point: bool) -> None:
        if weight_bits != 4:
            raise ValueError("Currently, only 4-bit weight quantization is supported for AWQ, "
                             f"but got {weight_bits} bits.")
        self.weight_bits, self.group_size, self.zero_point = weight_bits, group_size, zero_point
        self.pack_factor = 32 // weight_bits

    def __repr__(self) -> str:
        return f"AWQConfig(weight_bits={self.weight_bits}, group_size={self.group_size}, zero_point={self.zero_point})"

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
        pick = cls.get_from_keys
        return cls(pick(config, ["w_bit", "bits"]), pick(config, ["q_group_size", "group_size"]),
                   pick(config, ["zero_point"]))

    def get_quant_method(self, layer: torch.nn.Module) -> Optional["AWQLinearMethod"]:
        from ..linear import LinearBase
        return AWQLinearMethod(self) if isinstance(layer, LinearBase) else None

    def get_scaled_act_names(self) -> List[str]:
        return ["gelu", "gelu_fast", "gelu_new", "gelu_pytorch_tanh"]

    # ---- the schema -----------------------------------------------------------------------------------------
    def requirements(self) -> List[Require]:
        return [
            Require(lambda g: g.k % self.group_size == 0,
                    lambda g: "The input size is not aligned with the quantized weight shape. " + _TP_HINT),
            Require(lambda g: g.n % self.pack_factor == 0,
                    lambda g: "The output size is not aligned with the quantized weight shape. " + _TP_HINT),
        ]

    def slots(self) -> List[Slot]:
        pack, group = self.pack_factor, self.group_size
        packed_n = {"input_dim": 0, "output_dim": 1, "packed_dim": 1, "pack_factor": pack}
        return [
            Slot("qweight", lambda g: (g.k, g.n // pack), torch.int32, lambda g: dict(packed_n)),
            Slot("qzeros", lambda g: (g.k // group, g.n // pack), torch.int32, lambda g: dict(packed_n)),
            Slot("scales", lambda g: (g.k // group, g.n), lambda g: g.dtype, lambda g: {"input_dim": 0, "output_dim": 1}),
        ]


class AWQLinearMethod(LinearMethodBase):

    def __init__(self, quant_config: AWQConfig):
        self.quant_config = quant_config

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        build(layer, Geometry(input_size_per_partition, tuple(output_partition_sizes), params_dtype),
              self.quant_config.requirements(), self.quant_config.slots(), extra_weight_attrs)

    # ---- Marlin route (see the module docstring) ------------------------------------------------------------------
    @staticmethod
    def _awq_unpack_cols(packed: torch.Tensor) -> torch.Tensor:
        """int32 [R, N / 8] in AWQ nibble order -> int32 [R, N] in natural column order"""
        offsets = torch.tensor(_AWQ_NIBBLE_OF_COLUMN, dtype=torch.int32, device=packed.device)
        return ((packed.unsqueeze(-1) >> offsets) & 0xF).reshape(packed.shape[0], -1)

    def process_weights_after_loading(self, layer) -> None:
        k, n = layer.qweight.shape[0], layer.qweight.shape[1] * self.quant_config.pack_factor
        dev = layer.qweight.device
        fits = (self.quant_config.group_size == 128 and k % 256 == 0 and n % 64 == 0 and dev.type == "cuda"
                and layer.scales.dtype in (torch.half, torch.bfloat16))
        if not fits:
            return   # the layer stays on awq_gemm / awq_dequantize
        order = torch.tensor(_MARLIN_COLUMN_ORDER, device=dev)

        def marlin_columns(t: torch.Tensor) -> torch.Tensor:
            return t.reshape(-1, 64)[:, order].reshape(-1, n).contiguous()

        layer.marlin_qweight = ops.awq_marlin_repack(layer.qweight.data.contiguous(), k, n)
        layer.marlin_scales = marlin_columns(layer.scales.data)
        layer.marlin_zeros = marlin_columns(self._awq_unpack_cols(layer.qzeros.data).to(layer.scales.dtype))
        layer.marlin_workspace = torch.zeros(n // 64 * 16, dtype=torch.int32, device=dev)
        layer.awq_marlin_kn = (k, n)

    # ---- forward ----------------------------------------------------------------------------------------------
    def apply(self, layer, x, bias=None):
        rows = x.reshape(-1, x.shape[-1])
        marlin_kn = getattr(layer, "awq_marlin_kn", None)
        if marlin_kn is not None:
            k, n = marlin_kn
            y = ops.marlin_zp_gemm(rows, layer.marlin_qweight, layer.marlin_scales, layer.marlin_zeros,
                                   layer.marlin_workspace, rows.shape[0], n, k)
        else:
            n = layer.qweight.shape[-1] * self.quant_config.pack_factor
            if rows.shape[0] >= _DEQUANT_FROM_TOKENS:
                y = rows @ ops.awq_dequantize(layer.qweight, layer.scales, layer.qzeros, 0, 0, 0)
            else:
                y = ops.awq_gemm(rows, layer.qweight, layer.scales, layer.qzeros, self.quant_config.pack_factor)
        if bias is not None:
            y.add_(bias)
        return y.reshape(x.shape[:-1] + (n, ))
