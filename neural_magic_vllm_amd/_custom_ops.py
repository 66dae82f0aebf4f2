"""Drop-in for `vllm._custom_ops` (reference: vllm/_custom_ops.py:49-467).

Same function names, argument order and return conventions; every call goes through
torch.ops._C / _C_cache_ops / _C_cuda_utils, which neural_magic_vllm_amd/_torch_bindings.py
registers on top of libnmvllm_hip.so.  Ops the reference exposes but that are outside the hot
path (aqlm, squeezellm, marlin 2:4, moe, punica, custom all-reduce) are
not defined here.
"""
from typing import List, Optional, Tuple, Type

import torch

from . import _torch_bindings

_torch_bindings.register()


def is_custom_op_supported(op_name: str) -> bool:
    op, overloads = torch._C._jit_get_operation(op_name)
    return op is not None


# activation ops
def silu_and_mul(out: torch.Tensor, x: torch.Tensor) -> None:
    torch.ops._C.silu_and_mul(out, x)


def gelu_and_mul(out: torch.Tensor, x: torch.Tensor) -> None:
    torch.ops._C.gelu_and_mul(out, x)


def gelu_tanh_and_mul(out: torch.Tensor, x: torch.Tensor) -> None:
    torch.ops._C.gelu_tanh_and_mul(out, x)


def gelu_fast(out: torch.Tensor, x: torch.Tensor) -> None:
    torch.ops._C.gelu_fast(out, x)


def gelu_new(out: torch.Tensor, x: torch.Tensor) -> None:
    torch.ops._C.gelu_new(out, x)


def gelu_quick(out: torch.Tensor, x: torch.Tensor) -> None:
    torch.ops._C.gelu_quick(out, x)


# page attention ops
def paged_attention_v1(
    out: torch.Tensor,
    query: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    num_kv_heads: int,
    scale: float,
    block_tables: torch.Tensor,
    seq_lens: torch.Tensor,
    block_size: int,
    max_seq_len: int,
    alibi_slopes: Optional[torch.Tensor],
    kv_cache_dtype: str,
    kv_scale: float,
    tp_rank: int = 0,
    blocksparse_local_blocks: int = 0,
    blocksparse_vert_stride: int = 0,
    blocksparse_block_size: int = 64,
    blocksparse_head_sliding_step: int = 0,
) -> None:
    torch.ops._C.paged_attention_v1(out, query, key_cache, value_cache, num_kv_heads, scale,
                                    block_tables, seq_lens, block_size, max_seq_len,
                                    alibi_slopes, kv_cache_dtype, kv_scale, tp_rank,
                                    blocksparse_local_blocks, blocksparse_vert_stride,
                                    blocksparse_block_size, blocksparse_head_sliding_step)


def paged_attention_v2(
    out: torch.Tensor,
    exp_sum: torch.Tensor,
    max_logits: torch.Tensor,
    tmp_out: torch.Tensor,
    query: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    num_kv_heads: int,
    scale: float,
    block_tables: torch.Tensor,
    seq_lens: torch.Tensor,
    block_size: int,
    max_seq_len: int,
    alibi_slopes: Optional[torch.Tensor],
    kv_cache_dtype: str,
    kv_scale: float,
    tp_rank: int = 0,
    blocksparse_local_blocks: int = 0,
    blocksparse_vert_stride: int = 0,
    blocksparse_block_size: int = 64,
    blocksparse_head_sliding_step: int = 0,
) -> None:
    torch.ops._C.paged_attention_v2(out, exp_sum, max_logits, tmp_out, query, key_cache,
                                    value_cache, num_kv_heads, scale, block_tables, seq_lens,
                                    block_size, max_seq_len, alibi_slopes, kv_cache_dtype,
                                    kv_scale, tp_rank, blocksparse_local_blocks,
                                    blocksparse_vert_stride, blocksparse_block_size,
                                    blocksparse_head_sliding_step)


# pos encoding ops
def rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                     head_size: int, cos_sin_cache: torch.Tensor, is_neox: bool) -> None:
    torch.ops._C.rotary_embedding(positions, query, key, head_size, cos_sin_cache, is_neox)


def batched_rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                             head_size: int, cos_sin_cache: torch.Tensor, is_neox: bool,
                             rot_dim: int, cos_sin_cache_offsets: torch.Tensor) -> None:
    torch.ops._C.batched_rotary_embedding(positions, query, key, head_size, cos_sin_cache,
                                          is_neox, rot_dim, cos_sin_cache_offsets)


# layer norm ops
def rms_norm(out: torch.Tensor, input: torch.Tensor, weight: torch.Tensor,
             epsilon: float) -> None:
    torch.ops._C.rms_norm(out, input, weight, epsilon)


def fused_add_rms_norm(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                       epsilon: float) -> None:
    torch.ops._C.fused_add_rms_norm(input, residual, weight, epsilon)


# gptq_marlin
def gptq_marlin_repack(b_q_weight: torch.Tensor, perm: torch.Tensor, size_k: int, size_n: int,
                       num_bits: int) -> torch.Tensor:
    return torch.ops._C.gptq_marlin_repack(b_q_weight, perm, size_k, size_n, num_bits)


def gptq_marlin_gemm(a: torch.Tensor, b_q_weight: torch.Tensor, b_scales: torch.Tensor,
                     g_idx: torch.Tensor, perm: torch.Tensor, workspace: torch.Tensor,
                     num_bits: int, size_m: int, size_n: int, size_k: int,
                     is_k_full: bool) -> torch.Tensor:
    return torch.ops._C.gptq_marlin_gemm(a, b_q_weight, b_scales, g_idx, perm, workspace,
                                         num_bits, size_m, size_n, size_k, is_k_full)


# awq (vllm/_custom_ops.py:162-177).  NOTE the reference wrapper's parameter NAMES are
# (input, qweight, qzeros, scales) but it forwards them positionally to the native
# awq_gemm(in_feats, kernel, scaling_factors, zeros); AWQLinearMethod passes
# (x, qweight, scales, qzeros) (awq.py:172-173), which is what the native op expects.
def awq_dequantize(qweight: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor,
                   split_k_iters: int, thx: int, thy: int) -> torch.Tensor:
    return torch.ops._C.awq_dequantize(qweight, scales, zeros, split_k_iters, thx, thy)


def awq_gemm(input: torch.Tensor, qweight: torch.Tensor, qzeros: torch.Tensor,
             scales: torch.Tensor, split_k_iters: int) -> torch.Tensor:
    return torch.ops._C.awq_gemm(input, qweight, qzeros, scales, split_k_iters)


# gptq (vllm/_custom_ops.py:180-192)
def gptq_gemm(a: torch.Tensor, b_q_weight: torch.Tensor, b_gptq_qzeros: torch.Tensor,
              b_gptq_scales: torch.Tensor, b_g_idx: torch.Tensor, use_exllama: bool,
              bit: int) -> torch.Tensor:
    return torch.ops._C.gptq_gemm(a, b_q_weight, b_gptq_qzeros, b_gptq_scales, b_g_idx,
                                  use_exllama, bit)


def gptq_shuffle(q_weight: torch.Tensor, q_perm: torch.Tensor, bit: int) -> None:
    torch.ops._C.gptq_shuffle(q_weight, q_perm, bit)


# marlin (vllm/_custom_ops.py:201-206)
def marlin_gemm(a: torch.Tensor, b_q_weight: torch.Tensor, b_scales: torch.Tensor,
                workspace: torch.Tensor, size_m: int, size_n: int, size_k: int) -> torch.Tensor:
    return torch.ops._C.marlin_gemm(a, b_q_weight, b_scales, workspace, size_m, size_n, size_k)


# fp8 marlin (vllm/_custom_ops.py:272-279)
def fp8_marlin_gemm(a: torch.Tensor, b_q_weight: torch.Tensor, b_scales: torch.Tensor,
                    workspace: torch.Tensor, num_bits: int, size_m: int, size_n: int,
                    size_k: int) -> torch.Tensor:
    return torch.ops._C.fp8_marlin_gemm(a, b_q_weight, b_scales, workspace, num_bits, size_m,
                                        size_n, size_k)


# cutlass (vllm/_custom_ops.py:219-238)
def cutlass_scaled_mm_supports_fp8(cuda_device_capability: int) -> bool:
    return torch.ops._C.cutlass_scaled_mm_supports_fp8(cuda_device_capability)


def cutlass_scaled_mm(a: torch.Tensor, b: torch.Tensor, scale_a: torch.Tensor,
                      scale_b: torch.Tensor, out_dtype: Type[torch.dtype],
                      bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert (b.shape[0] % 16 == 0 and b.shape[1] % 16 == 0)
    assert (out_dtype is torch.bfloat16 or out_dtype is torch.float16)
    m = a.shape[0]
    n = b.shape[1]
    out = torch.empty((m, n), dtype=out_dtype, device=a.device)
    torch.ops._C.cutlass_scaled_mm(out, a, b, scale_a, scale_b, bias)
    return out


# fp8 (vllm/_custom_ops.py:282-321)
def scaled_fp8_quant(input: torch.Tensor, scale: Optional[torch.Tensor] = None,
                     batch_dim_padding: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Quantize to FP8 (e4m3fn): static when `scale` is given, dynamic per-tensor otherwise;
    `batch_dim_padding` pads the first dimension of the output."""
    if batch_dim_padding:
        shape = (max(batch_dim_padding, input.shape[0]), *input.shape[1:])
        output = torch.empty(shape, device=input.device, dtype=torch.float8_e4m3fn)
    else:
        output = torch.empty_like(input, dtype=torch.float8_e4m3fn)
    if scale is None:
        scale = torch.zeros(1, device=input.device, dtype=torch.float32)
        torch.ops._C.dynamic_scaled_fp8_quant(output, input, scale)
    else:
        torch.ops._C.static_scaled_fp8_quant(output, input, scale)
    return output, scale


# int8 (vllm/_custom_ops.py:324-351)
def scaled_int8_quant(input: torch.Tensor,
                      scale: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """static per-tensor when `scale` is given, dynamic per-token otherwise"""
    output = torch.empty_like(input, dtype=torch.int8)
    if scale is not None:
        torch.ops._C.static_scaled_int8_quant(output, input, scale)
        return output, scale
    input_scales = torch.empty((input.numel() // input.shape[-1], 1), device=input.device,
                               dtype=torch.float32)
    torch.ops._C.dynamic_scaled_int8_quant(output, input, input_scales)
    return output, input_scales


# cache ops
def reshape_and_cache(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor,
                      value_cache: torch.Tensor, slot_mapping: torch.Tensor,
                      kv_cache_dtype: str, kv_scale: float) -> None:
    torch.ops._C_cache_ops.reshape_and_cache(key, value, key_cache, value_cache, slot_mapping,
                                             kv_cache_dtype, kv_scale)


def reshape_and_cache_flash(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor,
                            value_cache: torch.Tensor, slot_mapping: torch.Tensor,
                            kv_cache_dtype: str) -> None:
    torch.ops._C_cache_ops.reshape_and_cache_flash(key, value, key_cache, value_cache,
                                                   slot_mapping, kv_cache_dtype)


def copy_blocks(key_caches: List[torch.Tensor], value_caches: List[torch.Tensor],
                block_mapping: torch.Tensor) -> None:
    torch.ops._C_cache_ops.copy_blocks(key_caches, value_caches, block_mapping)


def swap_blocks(src: torch.Tensor, dst: torch.Tensor, block_mapping: torch.Tensor) -> None:
    torch.ops._C_cache_ops.swap_blocks(src, dst, block_mapping)


def convert_fp8(output: torch.Tensor, input: torch.Tensor, scale: float = 1.0,
                kv_dtype: str = "fp8") -> None:
    torch.ops._C_cache_ops.convert_fp8(output, input, scale, kv_dtype)


# custom ar (vllm/_custom_ops.py:425-468; csrc/custom_all_reduce.cu).  IPC handles are passed as latin-1 strings
def init_custom_ar(meta: torch.Tensor, rank_data: torch.Tensor, handles: List[str], offsets: List[int],
                   rank: int, full_nvlink: bool) -> int:
    return torch.ops._C_custom_ar.init_custom_ar(meta, rank_data, handles, offsets, rank, full_nvlink)


def should_custom_ar(inp: torch.Tensor, max_size: int, world_size: int, full_nvlink: bool) -> bool:
    return torch.ops._C_custom_ar.should_custom_ar(inp, max_size, world_size, full_nvlink)


def all_reduce_reg(fa: int, inp: torch.Tensor, out: torch.Tensor) -> None:
    torch.ops._C_custom_ar.all_reduce_reg(fa, inp, out)


def all_reduce_unreg(fa: int, inp: torch.Tensor, reg_buffer: torch.Tensor, out: torch.Tensor) -> None:
    torch.ops._C_custom_ar.all_reduce_unreg(fa, inp, reg_buffer, out)


def dispose(fa: int) -> None:
    torch.ops._C_custom_ar.dispose(fa)


def meta_size() -> int:
    return torch.ops._C_custom_ar.meta_size()


def register_buffer(fa: int, t: torch.Tensor, handles: List[str], offsets: List[int]) -> None:
    return torch.ops._C_custom_ar.register_buffer(fa, t, handles, offsets)


def get_graph_buffer_ipc_meta(fa: int) -> Tuple[List[str], List[int]]:
    return torch.ops._C_custom_ar.get_graph_buffer_ipc_meta(fa)


def register_graph_buffers(fa: int, handles: List[str], offsets: List[List[int]]) -> None:
    torch.ops._C_custom_ar.register_graph_buffers(fa, handles, offsets)


def get_device_attribute(attribute: int, device: int) -> int:
    return torch.ops._C_cuda_utils.get_device_attribute(attribute, device)


def get_max_shared_memory_per_block_device_attribute(device: int) -> int:
    return torch.ops._C_cuda_utils.get_max_shared_memory_per_block_device_attribute(device)


# prompt attention: not an op of the reference's _C library (it calls flash-attn / Triton / SDPA
# from Python, rocm_flash_attn.py:349-430); exposed here for the attention backend
def prefill_attention(out: torch.Tensor, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                      cu_seqlens: torch.Tensor, max_seq_len: int, scale: float,
                      alibi_slopes: Optional[torch.Tensor] = None, sliding_window: Optional[int] = None) -> None:
    from neural_magic_vllm_amd import _torch_bindings as tb
    tb.prefill_attention(out, query, key, value, cu_seqlens, max_seq_len, scale, alibi_slopes, sliding_window)


def prefill_attention_supported(head_size: int) -> bool:
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.prefill_attention_supported(head_size)


def prefix_prefill_attention(out: torch.Tensor, query: torch.Tensor, key_cache: torch.Tensor,
                             value_cache: torch.Tensor, block_tables: torch.Tensor,
                             query_start_loc: torch.Tensor, seq_lens: torch.Tensor,
                             context_lens: torch.Tensor, max_query_len: int, scale: float,
                             alibi_slopes: Optional[torch.Tensor] = None,
                             sliding_window: Optional[int] = None) -> None:
    from neural_magic_vllm_amd import _torch_bindings as tb
    tb.prefix_prefill_attention(out, query, key_cache, value_cache, block_tables, query_start_loc,
                                seq_lens, context_lens, max_query_len, scale, alibi_slopes, sliding_window)


# AWQ / asymmetric checkpoints on the Marlin kernel (not ops of nm-vllm 0.5.1; later vLLM: awq_marlin)
def awq_marlin_repack(qweight: torch.Tensor, size_k: int, size_n: int) -> torch.Tensor:
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.awq_marlin_repack(qweight, size_k, size_n)


def marlin_zp_gemm(a: torch.Tensor, b_q_weight: torch.Tensor, b_scales: torch.Tensor,
                   b_zeros: torch.Tensor, workspace: torch.Tensor, size_m: int, size_n: int,
                   size_k: int) -> torch.Tensor:
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.marlin_zp_gemm(a, b_q_weight, b_scales, b_zeros, workspace, size_m, size_n, size_k)


# fused decode-step launches (not ops of nm-vllm 0.5.1; each is bit-identical to the sequence of
# reference ops it replaces -- see include/nmvllm_hip.h)
def rotary_embedding_and_cache(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                               value: torch.Tensor, head_size: int, cos_sin_cache: torch.Tensor,
                               is_neox: bool, key_cache: torch.Tensor, value_cache: torch.Tensor,
                               slot_mapping: torch.Tensor, kv_cache_dtype: str, kv_scale: float) -> None:
    from neural_magic_vllm_amd import _torch_bindings as tb
    tb.rotary_embedding_and_cache(positions, query, key, value, head_size, cos_sin_cache, is_neox,
                                  key_cache, value_cache, slot_mapping, kv_cache_dtype, kv_scale)


def rms_norm_dynamic_int8_quant(input: torch.Tensor, residual: Optional[torch.Tensor],
                                weight: torch.Tensor,
                                epsilon: float) -> Tuple[torch.Tensor, torch.Tensor]:
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.rms_norm_dynamic_int8_quant(input, residual, weight, epsilon)


def silu_and_mul_dynamic_int8_quant(input: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.silu_and_mul_dynamic_int8_quant(input)


def gptq_marlin_gemm_silu_mul(a: torch.Tensor, b_q_weight: torch.Tensor, b_scales: torch.Tensor,
                              workspace: torch.Tensor, size_m: int, size_n: int, size_k: int) -> torch.Tensor:
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.gptq_marlin_gemm_silu_mul(a, b_q_weight, b_scales, workspace, size_m, size_n, size_k)


def greedy_sample_advance(logits: torch.Tensor, input_ids: Optional[torch.Tensor] = None,
                          positions: Optional[torch.Tensor] = None, seq_lens: Optional[torch.Tensor] = None,
                          slot_mapping: Optional[torch.Tensor] = None,
                          block_tables: Optional[torch.Tensor] = None, block_size: int = 0) -> torch.Tensor:
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.greedy_sample_advance(logits, input_ids, positions, seq_lens, slot_mapping, block_tables,
                                    block_size)


# deferred split-K: the GEMM leaves fp32 slabs, the following fused_add_rms_norm sums them
def w4_native_repack(qweight: torch.Tensor, perm: Optional[torch.Tensor], size_k: int, size_n: int) -> torch.Tensor:
    """GPTQ qweight -> the MFMA-native W4 tensor (csrc/w4a16_gemm.hip; not an op of the reference)"""
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.w4_native_repack(qweight, perm, size_k, size_n)


def w4_native_gemm_splits(size_m: int, size_n: int, size_k: int, num_groups: Optional[int] = None) -> int:
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.w4_native_gemm_splits(size_m, size_n, size_k, num_groups)


def w4_native_gemm_slab16(size_m: int, size_n: int, size_k: int) -> bool:
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.w4_native_gemm_slab16(size_m, size_n, size_k)


def w4_native_prefill_plan(size_m: int, size_n: int, size_k: int) -> bool:
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.w4_native_prefill_plan(size_m, size_n, size_k)


def w4_native_gemm(a: torch.Tensor, b_native: torch.Tensor, scales: torch.Tensor, workspace: Optional[torch.Tensor],
                   size_m: int, size_n: int, size_k: int, mode: int = 0) -> torch.Tensor:
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.w4_native_gemm(a, b_native, scales, workspace, size_m, size_n, size_k, mode)


def gptq_marlin_gemm_partial_splits(size_m: int, size_n: int, size_k: int, num_groups=None) -> int:
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.gptq_marlin_gemm_partial_splits(size_m, size_n, size_k, num_groups)


def gptq_marlin_gemm_partial(a: torch.Tensor, b_q_weight: torch.Tensor, b_scales: torch.Tensor,
                             size_m: int, size_n: int, size_k: int) -> torch.Tensor:
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.gptq_marlin_gemm_partial(a, b_q_weight, b_scales, size_m, size_n, size_k)


def fused_add_rms_norm_partial(slab: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                               epsilon: float) -> torch.Tensor:
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.fused_add_rms_norm_partial(slab, residual, weight, epsilon)


def rotary_embedding_and_cache_partial(positions: torch.Tensor, slab: torch.Tensor, num_heads: int,
                                       num_kv_heads: int, head_size: int, cos_sin_cache: torch.Tensor,
                                       key_cache: Optional[torch.Tensor], value_cache: Optional[torch.Tensor],
                                       slot_mapping: Optional[torch.Tensor], kv_cache_dtype: str,
                                       kv_scale: float, dtype: torch.dtype) -> torch.Tensor:
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.rotary_embedding_and_cache_partial(positions, slab, num_heads, num_kv_heads, head_size,
                                                 cos_sin_cache, key_cache, value_cache, slot_mapping,
                                                 kv_cache_dtype, kv_scale, dtype)


def prefetch_l3(t: torch.Tensor, workgroups: int = 0) -> None:
    from neural_magic_vllm_amd import _torch_bindings as tb
    tb.prefetch_l3(t, workgroups)


def greedy_sample_shard(logits: torch.Tensor, index_offset: int) -> torch.Tensor:
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.greedy_sample_shard(logits, index_offset)


def greedy_sample_finish(gathered: torch.Tensor, world: int, num_seqs: int,
                         input_ids: Optional[torch.Tensor] = None, positions: Optional[torch.Tensor] = None,
                         seq_lens: Optional[torch.Tensor] = None, slot_mapping: Optional[torch.Tensor] = None,
                         block_tables: Optional[torch.Tensor] = None, block_size: int = 0) -> torch.Tensor:
    from neural_magic_vllm_amd import _torch_bindings as tb
    return tb.greedy_sample_finish(gathered, world, num_seqs, input_ids, positions, seq_lens, slot_mapping,
                                   block_tables, block_size)


def paged_attention_rope_partial(out: torch.Tensor, slab: torch.Tensor, positions: torch.Tensor,
                                 cos_sin_cache: torch.Tensor, slot_mapping: torch.Tensor,
                                 key_cache: torch.Tensor, value_cache: torch.Tensor, num_heads: int,
                                 num_kv_heads: int, head_size: int, scale: float, block_tables: torch.Tensor,
                                 seq_lens: torch.Tensor, block_size: int, max_seq_len: int, kv_cache_dtype: str,
                                 kv_scale: float, partition_bufs=None) -> None:
    from neural_magic_vllm_amd import _torch_bindings as tb
    tb.paged_attention_rope_partial(out, slab, positions, cos_sin_cache, slot_mapping, key_cache, value_cache,
                                    num_heads, num_kv_heads, head_size, scale, block_tables, seq_lens,
                                    block_size, max_seq_len, kv_cache_dtype, kv_scale, partition_bufs)
