"""The gfx950 attention backend: KV-cache write + decode on the HIP paged-attention kernels.

Mirror of the ROCm backend of the reference, vllm/attention/backends/rocm_flash_attn.py
(backend :21-54, metadata :57-163, impl :166-456): identical metadata fields, the same
prefill / decode token split (:333-347) and the same call sequence
reshape_and_cache -> prefill attention -> PagedAttention.forward_decode (:441).
Prefill runs the reference's "naive" option (torch SDPA per sequence, :379-403 / :459-491):
the reference's default there is a Triton kernel, which this build does not ship; a gfx950
flash-prefill kernel is row (f)-2 of SURVEY.md section 8 (next).
"""
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple, Type

import torch

from ... import _custom_ops as ops
from ..ops.paged_attn import PagedAttention, PagedAttentionMetadata
from .abstract import AttentionBackend, AttentionImpl, AttentionMetadata


class ROCmHipAttentionBackend(AttentionBackend):

    @staticmethod
    def get_name() -> str:
        return "rocm-hip-attn"

    @staticmethod
    def get_impl_cls() -> Type["ROCmHipAttentionImpl"]:
        return ROCmHipAttentionImpl

    @staticmethod
    def make_metadata(*args, **kwargs) -> "ROCmHipAttentionMetadata":
        return ROCmHipAttentionMetadata(*args, **kwargs)

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int,
                           head_size: int) -> Tuple[int, ...]:
        return PagedAttention.get_kv_cache_shape(num_blocks, block_size, num_kv_heads, head_size)

    @staticmethod
    def swap_blocks(src_kv_cache: torch.Tensor, dst_kv_cache: torch.Tensor,
                    src_to_dst: torch.Tensor) -> None:
        PagedAttention.swap_blocks(src_kv_cache, dst_kv_cache, src_to_dst)

    @staticmethod
    def copy_blocks(kv_caches: List[torch.Tensor], src_to_dists: torch.Tensor) -> None:
        PagedAttention.copy_blocks(kv_caches, src_to_dists)


@dataclass
class ROCmHipAttentionMetadata(AttentionMetadata, PagedAttentionMetadata):
    """Same fields as ROCmFlashAttentionMetadata (rocm_flash_attn.py:57-108).  Python values here
    are frozen into a captured HIP graph; anything that changes per step lives in tensors."""
    seq_lens: Optional[List[int]]
    seq_lens_tensor: Optional[torch.Tensor]
    max_query_len: Optional[int]
    max_prefill_seq_len: int
    max_decode_seq_len: int
    query_start_loc: Optional[torch.Tensor]
    seq_start_loc: Optional[torch.Tensor]
    use_cuda_graph: bool
    context_lens_tensor: Optional[torch.Tensor]
    _cached_prefill_metadata: Optional["ROCmHipAttentionMetadata"] = None
    _cached_decode_metadata: Optional["ROCmHipAttentionMetadata"] = None

    @property
    def prefill_metadata(self) -> Optional["ROCmHipAttentionMetadata"]:
        if self.num_prefills == 0:
            return None
        if self._cached_prefill_metadata is None:
            n = self.num_prefills
            self._cached_prefill_metadata = ROCmHipAttentionMetadata(
                num_prefills=n, num_prefill_tokens=self.num_prefill_tokens, num_decode_tokens=0,
                slot_mapping=self.slot_mapping[:self.num_prefill_tokens],
                seq_lens=self.seq_lens[:n], seq_lens_tensor=self.seq_lens_tensor[:n],
                max_query_len=self.max_query_len, max_prefill_seq_len=self.max_prefill_seq_len,
                max_decode_seq_len=0,
                query_start_loc=None if self.query_start_loc is None else self.query_start_loc[:n + 1],
                seq_start_loc=None if self.seq_start_loc is None else self.seq_start_loc[:n + 1],
                context_lens_tensor=None if self.context_lens_tensor is None
                else self.context_lens_tensor[:n],
                block_tables=None if self.block_tables is None else self.block_tables[:n],
                use_cuda_graph=False)
        return self._cached_prefill_metadata

    @property
    def decode_metadata(self) -> Optional["ROCmHipAttentionMetadata"]:
        if self.num_decode_tokens == 0:
            return None
        if self._cached_decode_metadata is None:
            n = self.num_prefills
            self._cached_decode_metadata = ROCmHipAttentionMetadata(
                num_prefills=0, num_prefill_tokens=0, num_decode_tokens=self.num_decode_tokens,
                slot_mapping=self.slot_mapping[self.num_prefill_tokens:], seq_lens=None,
                seq_lens_tensor=self.seq_lens_tensor[n:], max_query_len=None,
                max_prefill_seq_len=0, max_decode_seq_len=self.max_decode_seq_len,
                query_start_loc=None, seq_start_loc=None, context_lens_tensor=None,
                block_tables=self.block_tables[n:], use_cuda_graph=self.use_cuda_graph)
        return self._cached_decode_metadata


class ROCmHipAttentionImpl(AttentionImpl):
    """Layout: |<-- prefill tokens -->|<-- decode tokens -->| in one flattened batch."""

    def __init__(self, num_heads: int, head_size: int, scale: float,
                 num_kv_heads: Optional[int] = None, alibi_slopes: Optional[List[float]] = None,
                 sliding_window: Optional[int] = None, kv_cache_dtype: str = "auto",
                 blocksparse_params: Optional[Dict[str, Any]] = None) -> None:
        # the decode op takes the block-sparse arguments (PagedAttention.forward_decode); the block-sparse PROMPT path
        # is the reference's separate BlocksparseFlashAttentionBackend, which this backend does not replace
        assert blocksparse_params is None, "block-sparse models need the reference's blocksparse backend for prompts"
        # prompt attention masks the window in the kernel (rocm_flash_attn.py:244-246,394-406 hands it to
        # flash-attention / forward_prefix); decode sees it through the block tables and sequence lengths the
        # model runner builds, as in the reference (paged_attention takes no window)
        self.sliding_window = sliding_window
        self.num_heads = num_heads
        self.head_size = head_size
        self.scale = float(scale)
        self.num_kv_heads = num_heads if num_kv_heads is None else num_kv_heads
        self.alibi_slopes = None if alibi_slopes is None else torch.tensor(alibi_slopes,
                                                                           dtype=torch.float32)
        self.kv_cache_dtype = kv_cache_dtype
        assert self.num_heads % self.num_kv_heads == 0
        self.num_queries_per_kv = self.num_heads // self.num_kv_heads
        supported = PagedAttention.get_supported_head_sizes()
        if head_size not in supported:
            raise ValueError(f"Head size {head_size} is not supported by PagedAttention. "
                             f"Supported head sizes are: {supported}.")

    def rope_and_cache(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                       value: torch.Tensor, rotary_emb, kv_cache: Optional[torch.Tensor],
                       attn_metadata: ROCmHipAttentionMetadata, kv_scale: float = 1.0) -> bool:
        """rotary_embedding + reshape_and_cache in one launch (ops.rotary_embedding_and_cache): the
        two are back-to-back ~5 us launches on the decode path.  Returns False (nothing done) when
        the fused form does not apply; otherwise query / key are rotated in place and the caller
        passes cache_written=True to forward()."""
        if kv_cache is None or query.dim() != 2 or query.dtype not in (torch.float16, torch.bfloat16):
            return False
        cos_sin = rotary_emb.cos_sin_cache
        if cos_sin.device != query.device or cos_sin.dtype != query.dtype:
            cos_sin = rotary_emb.cos_sin_cache = cos_sin.to(query.device, dtype=query.dtype)
        key_cache, value_cache = PagedAttention.split_kv_cache(kv_cache, self.num_kv_heads,
                                                               self.head_size)
        ops.rotary_embedding_and_cache(positions, query, key, value, self.head_size, cos_sin,
                                       rotary_emb.is_neox_style, key_cache, value_cache,
                                       attn_metadata.slot_mapping.flatten(), self.kv_cache_dtype,
                                       kv_scale)
        return True

    def decode_rope_partial(self, positions: torch.Tensor, slab: torch.Tensor, rotary_emb,
                            kv_cache: Optional[torch.Tensor], attn_metadata: ROCmHipAttentionMetadata,
                            kv_scale: float, dtype: torch.dtype) -> Optional[torch.Tensor]:
        """decode-only batch, qkv as the projection's split-K slabs [S, T, N] (fp32) or as the finished
        row [T, N] in the model dtype: rope + cache write + paged attention in ONE launch (PagedAttention.forward_decode_rope_partial).  Returns [T, hidden] or
        None when the fused form does not apply (then rope_and_cache_partial + forward)."""
        dm = attn_metadata.decode_metadata
        if kv_cache is None or dm is None or attn_metadata.prefill_metadata is not None \
                or attn_metadata.num_prefill_tokens != 0 or self.alibi_slopes is not None \
                or not rotary_emb.is_neox_style or rotary_emb.rotary_dim != self.head_size \
                or dtype not in (torch.float16, torch.bfloat16) or slab.shape[-2] != attn_metadata.num_decode_tokens \
                or not slab.is_contiguous():
            return None
        cos_sin = rotary_emb.cos_sin_cache
        if cos_sin.device != slab.device or cos_sin.dtype != dtype:
            cos_sin = rotary_emb.cos_sin_cache = cos_sin.to(slab.device, dtype=dtype)
        key_cache, value_cache = PagedAttention.split_kv_cache(kv_cache, self.num_kv_heads, self.head_size)
        out = PagedAttention.forward_decode_rope_partial(
            slab, positions, cos_sin, attn_metadata.slot_mapping.flatten(), key_cache, value_cache,
            dm.block_tables, dm.seq_lens_tensor, dm.max_decode_seq_len, self.kv_cache_dtype, self.num_heads,
            self.num_kv_heads, self.head_size, self.scale, kv_scale, dtype)
        return out.view(out.shape[0], self.num_heads * self.head_size)

    def rope_and_cache_partial(self, positions: torch.Tensor, slab: torch.Tensor, rotary_emb,
                               kv_cache: Optional[torch.Tensor],
                               attn_metadata: ROCmHipAttentionMetadata, kv_scale: float,
                               dtype: torch.dtype) -> Optional[torch.Tensor]:
        """rope_and_cache() whose qkv row is still the fp32 split-K slabs of the qkv projection
        (ops.gptq_marlin_gemm_partial): returns the rounded, rotated qkv [T, (H + 2 KVH) * D] with k / v
        already in the cache, or None when the fused form does not apply (GPT-J style rope, partial
        rotary dim)."""
        if not rotary_emb.is_neox_style or rotary_emb.rotary_dim != self.head_size \
                or dtype not in (torch.float16, torch.bfloat16):
            return None
        cos_sin = rotary_emb.cos_sin_cache
        if cos_sin.device != slab.device or cos_sin.dtype != dtype:
            cos_sin = rotary_emb.cos_sin_cache = cos_sin.to(slab.device, dtype=dtype)
        key_cache = value_cache = slots = None
        if kv_cache is not None:
            key_cache, value_cache = PagedAttention.split_kv_cache(kv_cache, self.num_kv_heads,
                                                                   self.head_size)
            slots = attn_metadata.slot_mapping.flatten()
        return ops.rotary_embedding_and_cache_partial(positions, slab, self.num_heads, self.num_kv_heads,
                                                      self.head_size, cos_sin, key_cache, value_cache,
                                                      slots, self.kv_cache_dtype, kv_scale, dtype)

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                kv_cache: Optional[torch.Tensor], attn_metadata: ROCmHipAttentionMetadata,
                kv_scale: float = 1.0, cache_written: bool = False) -> torch.Tensor:
        """query [num_tokens, num_heads*head_size], key/value [num_tokens, num_kv_heads*head_size],
        kv_cache [2, num_blocks, block_size*num_kv_heads*head_size] -> [num_tokens, hidden].
        cache_written: rope_and_cache() below has already stored key / value in the cache."""
        num_tokens, hidden_size = query.shape
        query = query.view(-1, self.num_heads, self.head_size)
        key = key.view(-1, self.num_kv_heads, self.head_size)
        value = value.view(-1, self.num_kv_heads, self.head_size)
        if self.alibi_slopes is not None and self.alibi_slopes.device != query.device:
            self.alibi_slopes = self.alibi_slopes.to(query.device)

        key_cache = value_cache = None
        if kv_cache is not None:
            key_cache, value_cache = PagedAttention.split_kv_cache(kv_cache, self.num_kv_heads,
                                                                   self.head_size)
            if not cache_written:
                PagedAttention.write_to_paged_cache(key, value, key_cache, value_cache,
                                                    attn_metadata.slot_mapping, self.kv_cache_dtype,
                                                    kv_scale)

        num_prefill_tokens = attn_metadata.num_prefill_tokens
        num_decode_tokens = attn_metadata.num_decode_tokens
        assert key.shape[0] == num_prefill_tokens + num_decode_tokens

        if attn_metadata.prefill_metadata is None and num_decode_tokens == num_tokens:
            # decode-only batch (the hot path): no scatter into a separate output buffer
            dm = attn_metadata.decode_metadata
            out = PagedAttention.forward_decode(query, key_cache, value_cache, dm.block_tables,
                                                dm.seq_lens_tensor, dm.max_decode_seq_len,
                                                self.kv_cache_dtype, self.num_kv_heads, self.scale,
                                                self.alibi_slopes, kv_scale)
            return out.view(num_tokens, hidden_size)

        output = torch.empty((num_tokens, self.num_heads, self.head_size), dtype=query.dtype,
                             device=query.device)
        decode_query = query[num_prefill_tokens:]
        if (pm := attn_metadata.prefill_metadata) is not None:
            assert pm.seq_lens is not None
            if pm.context_lens_tensor is not None and kv_cache is not None \
                    and pm.block_tables is not None and pm.block_tables.numel() > 0 \
                    and bool((pm.context_lens_tensor > 0).any()):
                # prefix-enabled prefill (chunked prefill / prefix caching): every key comes from
                # the paged cache, which already holds the new tokens (written above)
                if self.kv_cache_dtype != "auto":
                    raise NotImplementedError("prefix-enabled prefill needs kv_cache_dtype auto")
                output[:num_prefill_tokens] = PagedAttention.forward_prefix(
                    query[:num_prefill_tokens], key[:num_prefill_tokens], value[:num_prefill_tokens],
                    key_cache, value_cache, pm.block_tables, pm.query_start_loc, pm.seq_lens_tensor,
                    pm.context_lens_tensor, pm.max_query_len, self.alibi_slopes, self.sliding_window, self.scale)
            elif ops.prefill_attention_supported(self.head_size) and query.dtype in (torch.float16, torch.bfloat16):
                # the hand-written HIP flash-attention forward (csrc/prefill_attention.hip)
                cu = pm.seq_start_loc
                if cu is None or cu.dtype != torch.int32:
                    cu = torch.tensor([0] + list(torch.tensor(pm.seq_lens).cumsum(0).tolist()),
                                      dtype=torch.int32, device=query.device)
                ops.prefill_attention(output[:num_prefill_tokens], query[:num_prefill_tokens],
                                      key[:num_prefill_tokens], value[:num_prefill_tokens], cu,
                                      pm.max_prefill_seq_len or max(pm.seq_lens), self.scale, self.alibi_slopes,
                                      self.sliding_window)
            else:
                if self.sliding_window is not None:
                    raise NotImplementedError("sliding-window prompts need head size 64 or 128")
                # head sizes other than 64 / 128: the reference's "naive" SDPA option
                output[:num_prefill_tokens] = _sdpa_prefill(query[:num_prefill_tokens],
                                                            key[:num_prefill_tokens],
                                                            value[:num_prefill_tokens], pm.seq_lens,
                                                            self.num_queries_per_kv, self.scale,
                                                            self.alibi_slopes)
        if (dm := attn_metadata.decode_metadata) is not None:
            output[num_prefill_tokens:] = PagedAttention.forward_decode(
                decode_query, key_cache, value_cache, dm.block_tables, dm.seq_lens_tensor,
                dm.max_decode_seq_len, self.kv_cache_dtype, self.num_kv_heads, self.scale,
                self.alibi_slopes, kv_scale)
        return output.view(num_tokens, hidden_size)


def _sdpa_prefill(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                  seq_lens: List[int], num_queries_per_kv: int, scale: float,
                  alibi_slopes: Optional[torch.Tensor]) -> torch.Tensor:
    """causal attention per prompt with torch SDPA (rocm_flash_attn.py:459-491)"""
    out = torch.empty_like(query)
    start = 0
    for seq_len in seq_lens:
        end = start + seq_len
        q = query[start:end].movedim(0, 1)  # [H, L, D]
        k = key[start:end].movedim(0, 1)
        v = value[start:end].movedim(0, 1)
        if num_queries_per_kv > 1:
            k = k.repeat_interleave(num_queries_per_kv, dim=0)
            v = v.repeat_interleave(num_queries_per_kv, dim=0)
        mask = None
        if alibi_slopes is not None:
            pos = torch.arange(seq_len, device=query.device)
            bias = (pos[None, :] - pos[:, None]).to(query.dtype)
            mask = alibi_slopes.to(query.dtype)[:, None, None] * bias[None]
            mask = mask + torch.full((seq_len, seq_len), float("-inf"), device=query.device,
                                     dtype=query.dtype).triu(1)[None]
        o = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=mask,
                                                             is_causal=mask is None, scale=scale)
        out[start:end] = o.movedim(0, 1)
        start = end
    return out
