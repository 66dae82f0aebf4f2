"""Attention layer (reference: vllm/attention/layer.py:16-103)."""
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from .backends.abstract import AttentionMetadata
from .selector import get_attn_backend


class Attention(nn.Module):
    """Multi-head / grouped-query attention over the paged KV cache:
    1. store the new keys/values in the cache, 2. attend, 3. return the output."""

    def __init__(self, num_heads: int, head_size: int, scale: float,
                 num_kv_heads: Optional[int] = None, alibi_slopes: Optional[List[float]] = None,
                 cache_config: Optional[Any] = None, quant_config: Optional[Any] = None,
                 blocksparse_params: Optional[Dict[str, Any]] = None) -> None:
        super().__init__()
        kv_cache_dtype = getattr(cache_config, "cache_dtype", "auto") if cache_config else "auto"
        block_size = getattr(cache_config, "block_size", 16) if cache_config else 16
        sliding_window = getattr(cache_config, "sliding_window", None) if cache_config else None
        if num_kv_heads is None:
            num_kv_heads = num_heads
        # The scaling factor of an fp8 KV cache; loaded from the checkpoint when present
        # (reference: layer.py:49-68, fp8.py Fp8KVCacheMethod).  gfx950 uses OCP e4m3, so there
        # is no x2 correction as on MI300 (llama.py:503-508 of the reference).
        self.kv_cache_dtype = kv_cache_dtype
        self._kv_scale = 1.0
        quant_method = quant_config.get_quant_method(self) if quant_config else None
        if quant_method is not None:
            # fp8 checkpoints may carry a kv-cache scaling factor (Fp8KVCacheMethod)
            if "e5m2" in self.kv_cache_dtype:
                raise ValueError("fp8_e5m2 kv-cache is not supported with fp8 checkpoints.")
            self.quant_method = quant_method
            self.quant_method.create_weights(self)
        dtype = torch.get_default_dtype()
        attn_backend = get_attn_backend(num_heads, head_size, num_kv_heads, sliding_window, dtype,
                                        kv_cache_dtype, block_size, blocksparse_params is not None)
        impl_cls = attn_backend.get_impl_cls()
        self.backend = attn_backend
        self.impl = impl_cls(num_heads, head_size, scale, num_kv_heads, alibi_slopes,
                             sliding_window, kv_cache_dtype, blocksparse_params)

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                kv_cache: Optional[torch.Tensor], attn_metadata: AttentionMetadata,
                cache_written: bool = False) -> torch.Tensor:
        if cache_written:
            return self.impl.forward(query, key, value, kv_cache, attn_metadata, self._kv_scale,
                                     cache_written=True)
        return self.impl.forward(query, key, value, kv_cache, attn_metadata, self._kv_scale)

    def rope_and_cache(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                       value: torch.Tensor, rotary_emb, kv_cache: Optional[torch.Tensor],
                       attn_metadata: AttentionMetadata) -> bool:
        """one-launch rotary embedding + KV-cache write where the backend has it (not part of the
        reference's Attention layer; see ROCmHipAttentionImpl.rope_and_cache)"""
        fused = getattr(self.impl, "rope_and_cache", None)
        if fused is None:
            return False
        return fused(positions, query, key, value, rotary_emb, kv_cache, attn_metadata, self._kv_scale)

    def rope_and_cache_partial(self, positions: torch.Tensor, slab: torch.Tensor, rotary_emb,
                               kv_cache: Optional[torch.Tensor], attn_metadata: AttentionMetadata,
                               dtype: torch.dtype) -> Optional[torch.Tensor]:
        """as rope_and_cache, from the un-reduced split-K slabs of the qkv projection"""
        fused = getattr(self.impl, "rope_and_cache_partial", None)
        if fused is None:
            return None
        return fused(positions, slab, rotary_emb, kv_cache, attn_metadata, self._kv_scale, dtype)

    def decode_rope_partial(self, positions: torch.Tensor, slab: torch.Tensor, rotary_emb,
                            kv_cache: Optional[torch.Tensor], attn_metadata: AttentionMetadata,
                            dtype: torch.dtype) -> Optional[torch.Tensor]:
        """rope + cache write + decode attention in one launch, from the qkv projection's slabs"""
        fused = getattr(self.impl, "decode_rope_partial", None)
        if fused is None:
            return None
        return fused(positions, slab, rotary_emb, kv_cache, attn_metadata, self._kv_scale, dtype)

    def extra_repr(self) -> str:
        return (f"head_size={self.impl.head_size}, num_heads={self.impl.num_heads}, "
                f"num_kv_heads={self.impl.num_kv_heads}, scale={self.impl.scale}, "
                f"backend={self.impl.__class__.__name__}")
