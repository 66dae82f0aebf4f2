"""Rotary positional embedding (reference: vllm/model_executor/layers/rotary_embedding.py:46-189
for the base class; the scaling variants are out of scope except Llama-3's plain rope)."""
from typing import Any, Dict, Optional, Tuple

import torch

from ... import _custom_ops as ops
from .custom_op import CustomOp


def _rotate_neox(x: torch.Tensor) -> torch.Tensor:
    x1 = x[..., :x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def _rotate_gptj(x: torch.Tensor) -> torch.Tensor:
    x1 = x[..., ::2]
    x2 = x[..., 1::2]
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


class RotaryEmbedding(CustomOp):
    """Original rotary positional embedding; cos/sin are precomputed on the host once."""

    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: int,
                 is_neox_style: bool, dtype: torch.dtype) -> None:
        super().__init__()
        self.head_size = head_size
        self.rotary_dim = rotary_dim
        self.max_position_embeddings = max_position_embeddings
        self.base = base
        self.is_neox_style = is_neox_style
        self.dtype = dtype
        cache = self._compute_cos_sin_cache().to(dtype)
        self.register_buffer("cos_sin_cache", cache, persistent=False)

    def _compute_inv_freq(self, base) -> torch.Tensor:
        return 1.0 / (base**(torch.arange(0, self.rotary_dim, 2, dtype=torch.float) /
                             self.rotary_dim))

    def _compute_cos_sin_cache(self) -> torch.Tensor:
        inv_freq = self._compute_inv_freq(self.base)
        t = torch.arange(self.max_position_embeddings, dtype=torch.float)
        freqs = torch.einsum("i,j -> ij", t, inv_freq)
        return torch.cat((freqs.cos(), freqs.sin()), dim=-1)

    def forward_native(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                       offsets: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        query = query.view(*query.shape[:-1], -1, self.head_size)
        key = key.view(*key.shape[:-1], -1, self.head_size)
        query_rot, query_pass = query[..., :self.rotary_dim], query[..., self.rotary_dim:]
        key_rot, key_pass = key[..., :self.rotary_dim], key[..., self.rotary_dim:]
        cos_sin = self.cos_sin_cache.to(positions.device)[
            torch.add(positions, offsets) if offsets is not None else positions]
        cos, sin = cos_sin.chunk(2, dim=-1)
        if self.is_neox_style:
            cos = cos.repeat(1, 1, 2).unsqueeze(-2) if cos.dim() == 3 else cos.repeat(1, 2).unsqueeze(-2)
            sin = sin.repeat(1, 1, 2).unsqueeze(-2) if sin.dim() == 3 else sin.repeat(1, 2).unsqueeze(-2)
            rotate_fn = _rotate_neox
        else:
            cos = cos.repeat_interleave(2, dim=-1).unsqueeze(-2)
            sin = sin.repeat_interleave(2, dim=-1).unsqueeze(-2)
            rotate_fn = _rotate_gptj
        query_rot = query_rot * cos + rotate_fn(query_rot) * sin
        key_rot = key_rot * cos + rotate_fn(key_rot) * sin
        query = torch.cat((query_rot, query_pass), dim=-1).flatten(-2)
        key = torch.cat((key_rot, key_pass), dim=-1).flatten(-2)
        return query, key

    def forward_cuda(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                     offsets: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.cos_sin_cache.device != query.device or self.cos_sin_cache.dtype != query.dtype:
            self.cos_sin_cache = self.cos_sin_cache.to(query.device, dtype=query.dtype)
        # in place on the (possibly strided) q / k slices of the qkv projection
        if offsets is not None:
            ops.batched_rotary_embedding(positions, query, key, self.head_size, self.cos_sin_cache,
                                         self.is_neox_style, self.rotary_dim, offsets)
        else:
            ops.rotary_embedding(positions, query, key, self.head_size, self.cos_sin_cache,
                                 self.is_neox_style)
        return query, key


_ROPE_DICT: Dict[Tuple, RotaryEmbedding] = {}


def get_rope(head_size: int, rotary_dim: int, max_position: int, base: int,
             is_neox_style: bool = True, rope_scaling: Optional[Dict[str, Any]] = None,
             dtype: Optional[torch.dtype] = None) -> RotaryEmbedding:
    """reference get_rope (rotary_embedding.py:726-805), unscaled rope only"""
    if dtype is None:
        dtype = torch.get_default_dtype()
    if rope_scaling is not None:
        kind = rope_scaling.get("rope_type", rope_scaling.get("type", "default"))
        if kind not in (None, "default"):
            raise NotImplementedError(f"rope scaling '{kind}' is outside the hot-path scope")
    key = (head_size, rotary_dim, max_position, base, is_neox_style, dtype)
    if key not in _ROPE_DICT:
        _ROPE_DICT[key] = RotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style,
                                          dtype)
    return _ROPE_DICT[key]
