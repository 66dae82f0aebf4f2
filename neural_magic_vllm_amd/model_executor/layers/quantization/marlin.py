"""Legacy Marlin-serialised checkpoints: 4-bit, group 128 or channelwise, no act-order.

The checkpoint already holds the kernel's operands -- `B` int32 [K/16, N*16/8] (the Marlin tile layout),
`s` [K/group, N] in the model dtype -- so this method only has to declare them with the sharding attributes
the loaders read, plus the lock / ticket array `workspace` int32 [N/64 * 16], and hand them to `marlin_gemm`.
Interface and checkpoint contract: reference vllm/model_executor/layers/quantization/marlin.py:17-252 (the
parameter table is pinned by tests/golden/linear_method_params.json["marlin"])."""
from typing import Any, Dict, List, Optional

import torch

from .... import _custom_ops as ops
from ._schema import Geometry, Require, Slot, build
from .base_config import LinearMethodBase, QuantizationConfig

_BITS = 4


class MarlinConfig(QuantizationConfig):
    """`quantize_config.json` of a Marlin checkpoint: {"group_size": 128 | -1, "is_marlin_format": true}"""

    # geometry of the serialised format (attribute names as the reference's config object exposes them)
    pack_factor = 32 // _BITS      # codes per int32
    tile_size = 16                 # k x n tile of the interchange layout
    min_n_threads = 64             # column granularity of a kernel tile
    min_k_threads = 128            # k granularity of a kernel tile
    max_parallel = 16              # lock slots per 64 columns
    perm_len = 1024                # weights covered by one tile permutation

    def __init__(self, group_size: int, lm_head_quantized: bool = False) -> None:
        if group_size not in (128, -1):
            raise ValueError("Currently, only group size 128 and -1 (channelwise) is supported for Marlin, "
                             f"but got group_size of {group_size}")
        self.group_size = group_size
        self.lm_head_quantized = lm_head_quantized

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
        return cls(group_size=cls.get_from_keys(config, ["group_size"]),
                   lm_head_quantized=cls.get_from_keys_or(config, ["lm_head"], default=False))

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant) -> Optional[str]:
        marlin_checkpoint = hf_quant_cfg.get("checkpoint_format") == "marlin" or hf_quant_cfg.get("is_marlin_format", False)
        user_agrees = user_quant in (None, "gptq", "marlin")
        return cls.get_name() if marlin_checkpoint and user_agrees else None

    def get_quant_method(self, layer: torch.nn.Module) -> Optional["MarlinLinearMethod"]:
        from ..linear import LinearBase
        from ..vocab_parallel_embedding import ParallelLMHead
        takes = isinstance(layer, LinearBase) or (self.lm_head_quantized and isinstance(layer, ParallelLMHead))
        return MarlinLinearMethod(self) if takes else None

    def get_scaled_act_names(self) -> List[str]:
        return []

    # ---- the schema -------------------------------------------------------------------------------------
    def requirements(self) -> List[Require]:
        c = self

        def multiple_of(what: str, value, unit_name: str, unit: int) -> Require:
            return Require(lambda g: value(g) % unit == 0,
                           lambda g: f"Weight {what} = {value(g)} is not divisible by {unit_name} = {unit}.")

        req = [
            Require(lambda g: g.dtype in (torch.float16, torch.bfloat16),
                    lambda g: f"The params dtype must be float16 or bfloat16, but got {g.dtype}"),
            multiple_of("output_size_per_partition", lambda g: g.n, "min_n_threads", c.min_n_threads),
            multiple_of("output_size_per_partition", lambda g: g.n, "pack_factor", c.pack_factor),
            multiple_of("input_size_per_partition", lambda g: g.k, "min_k_threads", c.min_k_threads),
        ]
        if c.group_size != -1:
            req.append(multiple_of("input_size_per_partition", lambda g: g.k, "group_size", c.group_size))
        # a tile permutation (perm_len weights = perm_len / tile_size^2 column tiles) may not straddle two ranks
        req.append(Require(lambda g: g.n % (c.perm_len // c.tile_size**2) == 0,
                           lambda g: "Each permutation group must reside on the same gpu"))
        return req

    def slots(self) -> List[Slot]:
        c = self
        groups = (lambda g: 1) if c.group_size == -1 else (lambda g: g.k // c.group_size)
        return [
            Slot("B", lambda g: (g.k // c.tile_size, g.n * c.tile_size // c.pack_factor), torch.int32,
                 lambda g: {"input_dim": 0, "output_dim": 1, "packed_dim": 1, "pack_factor": c.pack_factor,
                            "marlin_tile_size": c.tile_size}),
            Slot("s", lambda g: (groups(g), g.n), lambda g: g.dtype,
                 lambda g: {"input_dim": None if groups(g) == 1 else 0, "output_dim": 1}),
            Slot("workspace", lambda g: ((g.n // c.min_n_threads) * c.max_parallel, ), torch.int, fill="zeros"),
        ]


class MarlinLinearMethod(LinearMethodBase):

    def __init__(self, quant_config: MarlinConfig):
        self.quant_config = quant_config

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        build(layer, Geometry(input_size_per_partition, tuple(output_partition_sizes), params_dtype),
              self.quant_config.requirements(), self.quant_config.slots(), extra_weight_attrs)

    def apply(self, layer, x, bias=None):
        rows = x.reshape(-1, x.shape[-1])
        n = layer.s.shape[1]
        y = ops.marlin_gemm(rows, layer.B, layer.s, layer.workspace, rows.shape[0], n, rows.shape[1])
        if bias is not None:
            y.add_(bias)
        return y.reshape(x.shape[:-1] + (n, ))
