"""The quantization plugin surface (reference: vllm/model_executor/layers/quantization/
base_config.py:8-118).  Same abstract methods, so a config / linear method written against the
reference plugs in unchanged."""
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional

import torch
from torch import nn


class QuantizeMethodBase(ABC):
    """Base class for different quantized methods."""

    @abstractmethod
    def create_weights(self, layer: torch.nn.Module, *weight_args, **extra_weight_attrs):
        """Create weights for a layer; they are set as attributes of the layer."""
        raise NotImplementedError

    @abstractmethod
    def apply(self, layer: torch.nn.Module, *args, **kwargs) -> torch.Tensor:
        """Apply the weights in layer to the input tensor."""
        raise NotImplementedError

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        """Hook run once the checkpoint is loaded (transposes, requantisation, ...)."""
        return


class LinearMethodBase(QuantizeMethodBase):
    """Base class for (maybe quantized) linear methods (reference: linear.py:69-100; it lives
    here so that quantisation modules can subclass it without importing linear.py)."""

    @abstractmethod
    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int, output_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        """Create the layer's parameters.

        input_size_per_partition: weight input dim on this rank; output_partition_sizes: output
        dim of each logical matrix on this rank (e.g. [q, k, v] widths for QKVParallelLinear);
        input_size / output_size: dims across all ranks."""
        raise NotImplementedError

    @abstractmethod
    def apply(self, layer: torch.nn.Module, x: torch.Tensor,
              bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError


class QuantizationConfig(ABC):
    """Base class for quantization configs."""

    @abstractmethod
    def get_name(self) -> str:
        raise NotImplementedError

    @abstractmethod
    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        raise NotImplementedError

    @classmethod
    @abstractmethod
    def get_min_capability(cls) -> int:
        """Minimum device capability (gfx950 reports 95, see platforms.py)."""
        raise NotImplementedError

    @staticmethod
    @abstractmethod
    def get_config_filenames() -> List[str]:
        raise NotImplementedError

    @classmethod
    @abstractmethod
    def from_config(cls, config: Dict[str, Any]) -> "QuantizationConfig":
        raise NotImplementedError

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant) -> Optional[str]:
        return None

    @staticmethod
    def get_from_keys(config: Dict[str, Any], keys: List[str]) -> Any:
        for key in keys:
            if key in config:
                return config[key]
        raise ValueError(f"Cannot find any of {keys} in the model's quantization config.")

    @staticmethod
    def get_from_keys_or(config: Dict[str, Any], keys: List[str], default: Any) -> Any:
        try:
            return QuantizationConfig.get_from_keys(config, keys)
        except ValueError:
            return default

    @abstractmethod
    def get_quant_method(self, layer: torch.nn.Module) -> Optional[QuantizeMethodBase]:
        raise NotImplementedError

    @abstractmethod
    def get_scaled_act_names(self) -> List[str]:
        raise NotImplementedError
