"""`PagedAttention`: the host-side driver of the paged-KV ops -- cache geometry, cache write, the v1 / v2 choice and
the v2 partition buffers (interface: reference vllm/attention/ops/paged_attn.py:30-239; same static methods, argument
order and return conventions, so that an attention backend written against the reference runs on it unchanged).

Geometry (fixed by the boundary; csrc/attention_kernels.hip and csrc/cache_kernels.hip read exactly this):
  one allocation per layer       [2, blocks, block_size * kv_heads * head]          (K plane, V plane)
  K plane viewed as              [blocks, kv_heads, head / x, block_size, x]        x = 16 bytes of cache elements
  V plane viewed as              [blocks, kv_heads, head, block_size]
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from ... import _custom_ops as ops

PARTITION_TOKENS = 512    # PA_PARTITION of csrc/attention_kernels.hip = the reference's _PARTITION_SIZE
_PARTITION_SIZE = PARTITION_TOKENS
_FUSED_V1_MAX_CONTEXT = 896   # use_v1_fused: up to here one launch beats partitions + reduce on MI355X
_V1_MAX_CONTEXT = 8192    # beyond it the reference always partitions (paged_attn.py:112-121)
_V1_MIN_PAIRS = 512       # (sequence, head) pairs above which one launch already fills the GPU


def _partitions(max_seq_len: int) -> int:
    return -(-max_seq_len // PARTITION_TOKENS)


@dataclass
class PagedAttentionMetadata:
    """what a decode batch adds to the backend's metadata (paged_attn.py:16-27)"""
    seq_lens_tensor: Optional[torch.Tensor]    # [batch] tokens seen so far per sequence
    max_decode_seq_len: int                    # longest of them; 0 for a prompt-only batch
    block_tables: Optional[torch.Tensor]       # [batch, max blocks per sequence] physical block ids


class _PartitionBuffers:
    """v2 scratch: per (sequence, head, partition) the partial output, its softmax sum and its running max"""

    def __init__(self, num_seqs: int, num_heads: int, head_size: int, max_seq_len: int, dtype: torch.dtype, device):
        shape = (num_seqs, num_heads, _partitions(max_seq_len))
        self.exp_sums = torch.empty(shape, dtype=torch.float32, device=device)
        self.max_logits = torch.empty(shape, dtype=torch.float32, device=device)
        self.tmp_out = torch.empty(shape + (head_size, ), dtype=dtype, device=device)

    def as_tuple(self):
        return self.exp_sums, self.max_logits, self.tmp_out


class PagedAttention:

    # ---- geometry ------------------------------------------------------------------------------------------
    @staticmethod
    def get_supported_head_sizes() -> List[int]:
        return [64, 80, 96, 112, 128, 192, 256]

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int, head_size: int) -> Tuple[int, ...]:
        per_block = block_size * num_kv_heads * head_size
        return (2, num_blocks, per_block)

    @staticmethod
    def split_kv_cache(kv_cache: torch.Tensor, num_kv_heads: int, head_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
        k_plane, v_plane = kv_cache.unbind(0)
        blocks = k_plane.shape[0]
        x = 16 // kv_cache.element_size()
        return (k_plane.view(blocks, num_kv_heads, head_size // x, -1, x),
                v_plane.view(blocks, num_kv_heads, head_size, -1))

    # ---- cache maintenance -----------------------------------------------------------------------------------
    @staticmethod
    def write_to_paged_cache(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor,
                             slot_mapping: torch.Tensor, kv_cache_dtype: str, kv_scale: float) -> None:
        ops.reshape_and_cache(key, value, key_cache, value_cache, slot_mapping.reshape(-1), kv_cache_dtype, kv_scale)

    @staticmethod
    def swap_blocks(src_kv_cache: torch.Tensor, dst_kv_cache: torch.Tensor, src_to_dst: torch.Tensor) -> None:
        for plane in (0, 1):
            ops.swap_blocks(src_kv_cache[plane], dst_kv_cache[plane], src_to_dst)

    @staticmethod
    def copy_blocks(kv_caches: List[torch.Tensor], src_to_dists: torch.Tensor) -> None:
        k_planes, v_planes = zip(*((c[0], c[1]) for c in kv_caches)) if kv_caches else ((), ())
        ops.copy_blocks(list(k_planes), list(v_planes), src_to_dists)

    # ---- decode ------------------------------------------------------------------------------------------------
    @staticmethod
    def use_v1(max_seq_len: int, num_seqs: int, num_heads: int) -> bool:
        """the reference's rule (paged_attn.py:112-121; its decisions on 189 shapes are pinned by
        tests/golden/pa_heuristic.json): never above 8192 tokens of context; otherwise unpartitioned when there is a
        single partition anyway or when (sequences x heads) alone exceeds 512 workgroups"""
        if max_seq_len > _V1_MAX_CONTEXT:
            return False
        return _partitions(max_seq_len) == 1 or num_seqs * num_heads > _V1_MIN_PAIRS

    @staticmethod
    def use_v1_fused(max_seq_len: int, num_seqs: int, num_heads: int) -> bool:
        """the choice for this repo's own fused launch (forward_decode_rope_partial: not a reference path, so not bound
        to the reference's rule).  Measured on MI355X (tools/bench_attn.py --v2 --fused 2, profiles/r03_attn_v1v2.txt):
        the unpartitioned kernel walks a context with 8 waves per (sequence, kv head) when the grid is small, so up to
        ~900 tokens it beats two 512-token partitions plus the reduce launch (B=1, 530 tokens: 12.0 vs 15.1 us;
        break-even near 1000); beyond that, and whenever the reference's rule says v1, the two agree"""
        if PagedAttention.use_v1(max_seq_len, num_seqs, num_heads):
            return True
        return max_seq_len <= _FUSED_V1_MAX_CONTEXT

    @staticmethod
    def forward_decode(query: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor,
                       block_tables: torch.Tensor, seq_lens: torch.Tensor, max_seq_len: int, kv_cache_dtype: str,
                       num_kv_heads: int, scale: float, alibi_slopes: Optional[torch.Tensor], kv_scale: float,
                       tp_rank: int = 0, blocksparse_local_blocks: int = 0, blocksparse_vert_stride: int = 0,
                       blocksparse_block_size: int = 64, blocksparse_head_sliding_step: int = 0) -> torch.Tensor:
        num_seqs, num_heads, head_size = query.shape
        block_size = value_cache.shape[3]
        out = torch.empty_like(query)
        shared = (query, key_cache, value_cache, num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                  alibi_slopes, kv_cache_dtype, kv_scale, tp_rank, blocksparse_local_blocks, blocksparse_vert_stride,
                  blocksparse_block_size, blocksparse_head_sliding_step)
        if PagedAttention.use_v1(max_seq_len, num_seqs, num_heads):
            ops.paged_attention_v1(out, *shared)
        else:
            assert PARTITION_TOKENS % block_size == 0, "a partition is a whole number of KV blocks"
            bufs = _PartitionBuffers(num_seqs, num_heads, head_size, max_seq_len, out.dtype, out.device)
            ops.paged_attention_v2(out, *bufs.as_tuple(), *shared)
        return out

    @staticmethod
    def forward_decode_rope_partial(slab: torch.Tensor, positions: torch.Tensor, cos_sin_cache: torch.Tensor,
                                    slot_mapping: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor,
                                    block_tables: torch.Tensor, seq_lens: torch.Tensor, max_seq_len: int,
                                    kv_cache_dtype: str, num_heads: int, num_kv_heads: int, head_size: int, scale: float,
                                    kv_scale: float, dtype: torch.dtype) -> torch.Tensor:
        """forward_decode whose query and new key / value are still the qkv projection's split-K slabs [S, B, N]
        (fp32), or the finished qkv row [B, N] in the model dtype: rope + cache write + attention in one launch (not
        in the reference)"""
        num_seqs = slab.shape[-2]
        out = torch.empty((num_seqs, num_heads, head_size), dtype=dtype, device=slab.device)
        bufs = None
        if not PagedAttention.use_v1_fused(max_seq_len, num_seqs, num_heads):
            bufs = _PartitionBuffers(num_seqs, num_heads, head_size, max_seq_len, dtype, slab.device).as_tuple()
        ops.paged_attention_rope_partial(out, slab, positions, cos_sin_cache, slot_mapping, key_cache, value_cache,
                                         num_heads, num_kv_heads, head_size, scale, block_tables, seq_lens,
                                         value_cache.shape[3], max_seq_len, kv_cache_dtype, kv_scale, bufs)
        return out

    # ---- prefix-enabled prompt ------------------------------------------------------------------------------
    @staticmethod
    def forward_prefix(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor,
                       value_cache: torch.Tensor, block_tables: torch.Tensor, query_start_loc: torch.Tensor,
                       seq_lens_tensor: torch.Tensor, context_lens: torch.Tensor, max_query_len: int,
                       alibi_slopes: Optional[torch.Tensor], sliding_window: Optional[int],
                       scale: Optional[float] = None) -> torch.Tensor:
        """new tokens attending to their whole sequence in the paged cache (paged_attn.py:184-216).  `key` / `value`
        are taken for signature parity only: the backend has written them to the cache already, and the kernel
        reads every key from there (cache dtype auto, as the reference's forward_prefix)."""
        as_i32 = [t.to(torch.int32) for t in (query_start_loc, seq_lens_tensor, context_lens)]
        out = torch.empty_like(query)
        ops.prefix_prefill_attention(out, query, key_cache, value_cache, block_tables, *as_i32, max_query_len,
                                     query.shape[-1]**-0.5 if scale is None else scale, alibi_slopes, sliding_window)
        return out
