"""The three classes an attention backend plugs in through (interface: reference
vllm/attention/backends/abstract.py:9-132 -- names, signatures and the prefill / decode split of the metadata are the
reference's, so that `Attention` layers and model code written against it work unchanged):

  AttentionBackend   stateless description: name, impl class, metadata factory, KV-cache geometry and block moves
  AttentionMetadata  what one batch (prompt tokens first, then one decode token per sequence) tells the layers
  AttentionImpl      one layer's attention, constructed once, called per step
"""
import abc
import dataclasses
from typing import Any, Dict, Generic, List, Optional, Set, Tuple, Type, TypeVar

import torch


def _required_static(fn):
    """an abstract static method whose body is never the implementation"""
    return staticmethod(abc.abstractmethod(fn))


class AttentionBackend(abc.ABC):

    @_required_static
    def get_name() -> str:
        ...

    @_required_static
    def get_impl_cls() -> Type["AttentionImpl"]:
        ...

    @_required_static
    def make_metadata(*args, **kwargs) -> "AttentionMetadata":
        ...

    @_required_static
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int, head_size: int) -> Tuple[int, ...]:
        ...

    @_required_static
    def swap_blocks(src_kv_cache: torch.Tensor, dst_kv_cache: torch.Tensor, src_to_dst: torch.Tensor) -> None:
        ...

    @_required_static
    def copy_blocks(kv_caches: List[torch.Tensor], src_to_dists: torch.Tensor) -> None:
        ...


@dataclasses.dataclass
class AttentionMetadata:
    """One batch: `num_prefills` prompts contributing `num_prefill_tokens` tokens, followed by `num_decode_tokens`
    sequences contributing one token each; `slot_mapping[t]` = block * block_size + offset is where token t's key and
    value go in the paged cache (negative: padding)."""
    num_prefills: int
    num_prefill_tokens: int
    num_decode_tokens: int
    slot_mapping: torch.Tensor

    @property
    @abc.abstractmethod
    def prefill_metadata(self) -> Optional["AttentionMetadata"]:
        """the prompt part of the batch as its own metadata object (None when there is none)"""

    @property
    @abc.abstractmethod
    def decode_metadata(self) -> Optional["AttentionMetadata"]:
        """the decode part of the batch (None when there is none)"""

    def asdict_zerocopy(self, skip_fields: Optional[Set[str]] = None) -> Dict[str, Any]:
        """field name -> the field's own object (no copies: tensors are shared), minus `skip_fields`"""
        skip = skip_fields or ()
        return {f.name: getattr(self, f.name) for f in dataclasses.fields(self) if f.name not in skip}


T = TypeVar("T", bound=AttentionMetadata)


class AttentionImpl(abc.ABC, Generic[T]):

    @abc.abstractmethod
    def __init__(self, num_heads: int, head_size: int, scale: float, num_kv_heads: Optional[int] = None,
                 alibi_slopes: Optional[List[float]] = None, sliding_window: Optional[int] = None,
                 kv_cache_dtype: str = "auto", blocksparse_params: Optional[Dict[str, Any]] = None) -> None:
        ...

    @abc.abstractmethod
    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, kv_cache: torch.Tensor,
                attn_metadata: T, kv_scale: float = 1.0) -> torch.Tensor:
        ...
