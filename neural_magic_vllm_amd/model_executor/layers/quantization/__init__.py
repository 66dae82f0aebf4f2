"""Registry of quantisation methods (reference: quantization/__init__.py:21-36).  Only the
methods on the hot path are present; the reference's aqlm / squeezellm / bitsandbytes /
deepspeedfp / gptq_marlin_24 are out of scope."""
from typing import Dict, Type

from .base_config import QuantizationConfig
from .gptq_marlin import GPTQMarlinConfig

QUANTIZATION_METHODS: Dict[str, Type[QuantizationConfig]] = {
    "gptq_marlin": GPTQMarlinConfig,
}


def get_quantization_config(quantization: str) -> Type[QuantizationConfig]:
    if quantization not in QUANTIZATION_METHODS:
        raise ValueError(f"Invalid quantization method: {quantization}")
    return QUANTIZATION_METHODS[quantization]


__all__ = ["QuantizationConfig", "get_quantization_config", "QUANTIZATION_METHODS"]
