"""The quantisation plugin surface (interface: reference vllm/model_executor/layers/quantization/base_config.py:8-118
and the `LinearMethodBase` of linear.py:69-100).  A format is two objects:

  QuantizationConfig   parsed from the checkpoint's quantisation json; says which layers it takes and with what method
  QuantizeMethodBase   per layer: declares the parameters a checkpoint fills (`create_weights`), optionally rewrites
                       them once after loading (`process_weights_after_loading`), and runs the layer (`apply`)

Method names, arguments and the meaning of every return value are the reference's, so a config / method pair written
against it plugs in here unchanged (tests/golden/linear_method_params.json pins the parameter tables of the eleven
pairs this package ships)."""
import abc
from typing import Any, Dict, List, Optional

import torch
from torch import nn

_MISSING = object()


class QuantizeMethodBase(abc.ABC):

    @abc.abstractmethod
    def create_weights(self, layer: torch.nn.Module, *weight_args, **extra_weight_attrs):
        """register the layer's parameters on `layer` (attributes named as the checkpoint names them)"""

    @abc.abstractmethod
    def apply(self, layer: torch.nn.Module, *args, **kwargs) -> torch.Tensor:
        """the layer's forward on the parameters `create_weights` registered"""

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        """once per layer after the checkpoint is in: repacks, transposes, requantisation.  Default: nothing."""


class LinearMethodBase(QuantizeMethodBase):
    """methods of a linear layer (lives here, not in linear.py, so that a format module never imports linear.py)"""

    @abc.abstractmethod
    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int, output_partition_sizes: List[int],
                       input_size: int, output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        """`input_size_per_partition`: K on this rank; `output_partition_sizes`: this rank's width of every logical
        matrix fused into the layer ([q, k, v] for QKVParallelLinear, [gate, up] for the MLP, one entry otherwise);
        `input_size` / `output_size`: the unsharded dimensions; `extra_weight_attrs`: attributes (the weight loader
        among them) to set on every parameter"""

    @abc.abstractmethod
    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [..., K] -> [..., sum(output_partition_sizes)] (+ bias)"""


class QuantizationConfig(abc.ABC):

    # ---- identity and admission ---------------------------------------------------------------------------
    @abc.abstractmethod
    def get_name(self) -> str:
        """the `quantization=` / `quant_method` string"""

    @abc.abstractmethod
    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        ...

    @classmethod
    @abc.abstractmethod
    def get_min_capability(cls) -> int:
        """lowest device capability (major * 10 + minor) that may run it; gfx950 reports 95 (platforms.py)"""

    @staticmethod
    @abc.abstractmethod
    def get_config_filenames() -> List[str]:
        """files of a checkpoint directory to search for the quantisation json when config.json carries none"""

    # ---- construction -----------------------------------------------------------------------------------------
    @classmethod
    @abc.abstractmethod
    def from_config(cls, config: Dict[str, Any]) -> "QuantizationConfig":
        ...

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant) -> Optional[str]:
        """a format may claim a checkpoint written for another one (gptq -> gptq_marlin): its name, or None"""
        return None

    @staticmethod
    def get_from_keys(config: Dict[str, Any], keys: List[str]) -> Any:
        """the value of the first of `keys` the json has"""
        found = next((config[k] for k in keys if k in config), _MISSING)
        if found is _MISSING:
            raise ValueError(f"Cannot find any of {keys} in the model's quantization config.")
        return found

    @staticmethod
    def get_from_keys_or(config: Dict[str, Any], keys: List[str], default: Any) -> Any:
        return next((config[k] for k in keys if k in config), default)

    # ---- per layer ------------------------------------------------------------------------------------------------
    @abc.abstractmethod
    def get_quant_method(self, layer: torch.nn.Module) -> Optional[QuantizeMethodBase]:
        """the method for `layer`, or None when this format leaves the layer alone"""

    @abc.abstractmethod
    def get_scaled_act_names(self) -> List[str]:
        """activation functions that need a learned post-scale under this format"""
