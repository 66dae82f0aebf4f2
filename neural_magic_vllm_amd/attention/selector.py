"""Backend selection (reference: vllm/attention/selector.py:26-197).  There is exactly one
backend in this build; every other branch of the reference's selector (flash-attn, xformers,
flashinfer, torch-sdpa CPU, openvino, pallas, ipex) belongs to other vendors and is out of scope."""
from typing import Optional, Type

import torch

from .backends.abstract import AttentionBackend


def get_attn_backend(num_heads: int, head_size: int, num_kv_heads: int,
                     sliding_window: Optional[int], dtype: torch.dtype,
                     kv_cache_dtype: Optional[str], block_size: int,
                     is_blocksparse: bool = False) -> Type[AttentionBackend]:
    if is_blocksparse:
        raise NotImplementedError("block-sparse attention is outside the hot-path scope")
    if kv_cache_dtype not in (None, "auto", "fp8", "fp8_e4m3"):
        raise ValueError(f"kv cache dtype {kv_cache_dtype} is not supported on gfx950 "
                         "(OCP e4m3 only)")
    from .backends.rocm_hip_attn import ROCmHipAttentionBackend
    return ROCmHipAttentionBackend
