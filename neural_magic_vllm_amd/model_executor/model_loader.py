"""Checkpoint side of the harness: HF-layout Llama checkpoints (safetensors shards + config.json +
the quantisation config) -> the (name, tensor) stream that LlamaForCausalLM.load_weights consumes.

Behavioural references: vllm/model_executor/model_loader/weight_utils.py (safetensors_weights_iterator,
get_quant_config: `quantization_config` inside config.json, else quantize_config.json next to the
weights) and model_loader/loader.py (DefaultModelLoader).  Only local directories: there is no
download path here."""
import glob
import json
import os
from typing import Any, Dict, Iterable, Optional, Tuple

import torch


def safetensors_weights_iterator(model_dir: str) -> Iterable[Tuple[str, torch.Tensor]]:
    """every tensor of every *.safetensors shard in the directory, shard by shard (weight_utils.py)"""
    from safetensors import safe_open
    files = sorted(glob.glob(os.path.join(model_dir, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors files in {model_dir}")
    for path in files:
        with safe_open(path, framework="pt", device="cpu") as f:
            for name in f.keys():
                yield name, f.get_tensor(name)


def read_hf_config(model_dir: str) -> Dict[str, Any]:
    with open(os.path.join(model_dir, "config.json")) as f:
        return json.load(f)


def read_quant_config(model_dir: str, hf_config: Dict[str, Any]) -> Optional[Dict[str, Any]]:
    """weight_utils.get_quant_config: config.json's `quantization_config` (compressed-tensors, fp8, newer
    GPTQ exports) wins, else quantize_config.json (AutoGPTQ / AutoAWQ)"""
    qc = hf_config.get("quantization_config")
    if qc is not None:
        return dict(qc)
    path = os.path.join(model_dir, "quantize_config.json")
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return None


def quant_method_of(qc: Dict[str, Any]) -> str:
    """the reference's method name for a checkpoint's quantisation config; GPTQ checkpoints that the
    Marlin kernels can run are promoted to gptq_marlin as config.py:_verify_quantization does"""
    method = str(qc.get("quant_method", "")).lower()
    if not method:
        method = "awq" if "zero_point" in qc or qc.get("version", "").lower() == "gemm" else "gptq"
    if method == "gptq":
        from .layers.quantization.gptq_marlin import GPTQMarlinConfig
        if GPTQMarlinConfig.is_marlin_compatible(qc):
            method = "gptq_marlin"
    return method


def build_quant_config(model_dir: str, hf_config: Dict[str, Any]):
    """QuantizationConfig instance of the checkpoint, or None for an unquantised one"""
    qc = read_quant_config(model_dir, hf_config)
    if qc is None:
        return None
    from .layers.quantization import get_quantization_config
    return get_quantization_config(quant_method_of(qc)).from_config(qc)
