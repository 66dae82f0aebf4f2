"""Python orchestration of the paged-attention ops (reference: vllm/attention/ops/
paged_attn.py:30-239): cache split, cache write, v1/v2 choice, partition buffers."""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from ... import _custom_ops as ops

# Must equal PA_PARTITION in csrc/attention_kernels.hip (and the reference's PARTITION_SIZE).
_PARTITION_SIZE = 512


@dataclass
class PagedAttentionMetadata:
    """Metadata for PagedAttention."""
    # (batch_size,) length (all tokens seen so far) of each sequence
    seq_lens_tensor: Optional[torch.Tensor]
    # maximum sequence length in the decode batch; 0 for a prefill-only batch
    max_decode_seq_len: int
    # (batch_size, max_blocks_per_seq) physical block numbers of each sequence
    block_tables: Optional[torch.Tensor]


class PagedAttention:

    @staticmethod
    def get_supported_head_sizes() -> List[int]:
        return [64, 80, 96, 112, 128, 192, 256]

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int,
                           head_size: int) -> Tuple[int, ...]:
        return (2, num_blocks, block_size * num_kv_heads * head_size)

    @staticmethod
    def split_kv_cache(kv_cache: torch.Tensor, num_kv_heads: int,
                       head_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
        x = 16 // kv_cache.element_size()
        num_blocks = kv_cache.shape[1]
        key_cache = kv_cache[0].view(num_blocks, num_kv_heads, head_size // x, -1, x)
        value_cache = kv_cache[1].view(num_blocks, num_kv_heads, head_size, -1)
        return key_cache, value_cache

    @staticmethod
    def write_to_paged_cache(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor,
                             value_cache: torch.Tensor, slot_mapping: torch.Tensor,
                             kv_cache_dtype: str, kv_scale: float) -> None:
        ops.reshape_and_cache(key, value, key_cache, value_cache, slot_mapping.flatten(),
                              kv_cache_dtype, kv_scale)

    @staticmethod
    def use_v1(max_seq_len: int, num_seqs: int, num_heads: int) -> bool:
        """the reference's heuristic (paged_attn.py:112-121), kept verbatim in behaviour: one
        partition -> v1; many (seq, head) pairs -> v1; context > 8192 -> v2."""
        max_num_partitions = (max_seq_len + _PARTITION_SIZE - 1) // _PARTITION_SIZE
        return max_seq_len <= 8192 and (max_num_partitions == 1 or num_seqs * num_heads > 512)

    @staticmethod
    def forward_decode(query: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor,
                       block_tables: torch.Tensor, seq_lens: torch.Tensor, max_seq_len: int,
                       kv_cache_dtype: str, num_kv_heads: int, scale: float,
                       alibi_slopes: Optional[torch.Tensor], kv_scale: float, tp_rank: int = 0,
                       blocksparse_local_blocks: int = 0, blocksparse_vert_stride: int = 0,
                       blocksparse_block_size: int = 64,
                       blocksparse_head_sliding_step: int = 0) -> torch.Tensor:
        output = torch.empty_like(query)
        block_size = value_cache.shape[3]
        num_seqs, num_heads, head_size = query.shape
        max_num_partitions = (max_seq_len + _PARTITION_SIZE - 1) // _PARTITION_SIZE
        if PagedAttention.use_v1(max_seq_len, num_seqs, num_heads):
            ops.paged_attention_v1(output, query, key_cache, value_cache, num_kv_heads, scale,
                                   block_tables, seq_lens, block_size, max_seq_len, alibi_slopes,
                                   kv_cache_dtype, kv_scale, tp_rank, blocksparse_local_blocks,
                                   blocksparse_vert_stride, blocksparse_block_size,
                                   blocksparse_head_sliding_step)
        else:
            assert _PARTITION_SIZE % block_size == 0
            tmp_output = torch.empty(size=(num_seqs, num_heads, max_num_partitions, head_size),
                                     dtype=output.dtype, device=output.device)
            exp_sums = torch.empty(size=(num_seqs, num_heads, max_num_partitions),
                                   dtype=torch.float32, device=output.device)
            max_logits = torch.empty_like(exp_sums)
            ops.paged_attention_v2(output, exp_sums, max_logits, tmp_output, query, key_cache,
                                   value_cache, num_kv_heads, scale, block_tables, seq_lens,
                                   block_size, max_seq_len, alibi_slopes, kv_cache_dtype, kv_scale,
                                   tp_rank, blocksparse_local_blocks, blocksparse_vert_stride,
                                   blocksparse_block_size, blocksparse_head_sliding_step)
        return output

    @staticmethod
    def forward_decode_rope_partial(slab: torch.Tensor, positions: torch.Tensor, cos_sin_cache: torch.Tensor,
                                    slot_mapping: torch.Tensor, key_cache: torch.Tensor,
                                    value_cache: torch.Tensor, block_tables: torch.Tensor,
                                    seq_lens: torch.Tensor, max_seq_len: int, kv_cache_dtype: str,
                                    num_heads: int, num_kv_heads: int, head_size: int, scale: float,
                                    kv_scale: float, dtype: torch.dtype) -> torch.Tensor:
        """forward_decode whose query and new key / value are still the qkv projection's split-K slabs:
        rope + cache write + attention in one launch (not in the reference)"""
        num_seqs = slab.shape[-2]   # slabs [S, B, N] fp32, or the finished qkv [B, N] in the model dtype
        output = torch.empty((num_seqs, num_heads, head_size), dtype=dtype, device=slab.device)
        block_size = value_cache.shape[3]
        bufs = None
        if not PagedAttention.use_v1(max_seq_len, num_seqs, num_heads):
            parts = (max_seq_len + _PARTITION_SIZE - 1) // _PARTITION_SIZE
            exp_sums = torch.empty((num_seqs, num_heads, parts), dtype=torch.float32, device=slab.device)
            bufs = (exp_sums, torch.empty_like(exp_sums),
                    torch.empty((num_seqs, num_heads, parts, head_size), dtype=dtype, device=slab.device))
        ops.paged_attention_rope_partial(output, slab, positions, cos_sin_cache, slot_mapping, key_cache,
                                         value_cache, num_heads, num_kv_heads, head_size, scale, block_tables,
                                         seq_lens, block_size, max_seq_len, kv_cache_dtype, kv_scale, bufs)
        return output

    @staticmethod
    def forward_prefix(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                       key_cache: torch.Tensor, value_cache: torch.Tensor,
                       block_tables: torch.Tensor, query_start_loc: torch.Tensor,
                       seq_lens_tensor: torch.Tensor, context_lens: torch.Tensor,
                       max_query_len: int, alibi_slopes: Optional[torch.Tensor],
                       sliding_window: Optional[int], scale: Optional[float] = None) -> torch.Tensor:
        """prefix-enabled prefill (paged_attn.py:184-216).  `key` / `value` (the new tokens) are
        accepted for signature parity; the backend has already written them into the cache, which
        is where the kernel reads every key from (kv cache dtype auto)."""
        output = torch.empty_like(query)
        head_size = query.shape[-1]
        ops.prefix_prefill_attention(output, query, key_cache, value_cache, block_tables,
                                     query_start_loc.to(torch.int32), seq_lens_tensor.to(torch.int32),
                                     context_lens.to(torch.int32), max_query_len,
                                     scale if scale is not None else head_size**-0.5,
                                     alibi_slopes, sliding_window)
        return output

    @staticmethod
    def swap_blocks(src_kv_cache: torch.Tensor, dst_kv_cache: torch.Tensor,
                    src_to_dst: torch.Tensor) -> None:
        ops.swap_blocks(src_kv_cache[0], dst_kv_cache[0], src_to_dst)
        ops.swap_blocks(src_kv_cache[1], dst_kv_cache[1], src_to_dst)

    @staticmethod
    def copy_blocks(kv_caches: List[torch.Tensor], src_to_dists: torch.Tensor) -> None:
        key_caches = [kv_cache[0] for kv_cache in kv_caches]
        value_caches = [kv_cache[1] for kv_cache in kv_caches]
        ops.copy_blocks(key_caches, value_caches, src_to_dists)
