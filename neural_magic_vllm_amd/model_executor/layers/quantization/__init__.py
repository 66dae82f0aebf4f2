"""Registry of quantisation methods (reference: quantization/__init__.py:21-36).  Only the
methods on the hot path are present; the reference's aqlm / squeezellm / bitsandbytes /
deepspeedfp / gptq_marlin_24 are out of scope."""
from typing import Dict, Type

from .awq import AWQConfig
from .base_config import QuantizationConfig
from .compressed_tensors import CompressedTensorsConfig
from .fp8 import Fp8Config
from .gptq import GPTQConfig
from .gptq_marlin import GPTQMarlinConfig
from .marlin import MarlinConfig

QUANTIZATION_METHODS: Dict[str, Type[QuantizationConfig]] = {
    "awq": AWQConfig,
    "fp8": Fp8Config,
    # the order of the gptq methods matters for override_quantization_method (config.py)
    "marlin": MarlinConfig,
    "gptq_marlin": GPTQMarlinConfig,
    "gptq": GPTQConfig,
    "compressed-tensors": CompressedTensorsConfig,
}


def get_quantization_config(quantization: str) -> Type[QuantizationConfig]:
    if quantization not in QUANTIZATION_METHODS:
        raise ValueError(f"Invalid quantization method: {quantization}")
    return QUANTIZATION_METHODS[quantization]


__all__ = ["QuantizationConfig", "get_quantization_config", "QUANTIZATION_METHODS"]
