"""torch.ops._C / _C_cache_ops / _C_cuda_utils registered on top of the C ABI.

The reference registers its native ops from C++ (csrc/torch_bindings.cpp:18-259,
`TORCH_LIBRARY_EXPAND(TORCH_EXTENSION_NAME, ops)`).  This module is the drop-in for that file:
the SAME namespaces, op names and schema strings, dispatch key CUDA (HIP tensors carry the CUDA
key on PyTorch-ROCm), but each impl unwraps the tensors to raw pointers / sizes / strides and
calls libnmvllm_hip.so (include/nmvllm_hip.h) on the current HIP stream.  Importing it plays
the role of `import vllm._C`.

Only dispatch key CUDA gets an impl: calling any of these ops with CPU tensors raises
NotImplementedError from the dispatcher -- there is deliberately no CPU path in the product.
"""
from typing import List, Optional

import torch

from . import _lib
from ._lib import check, device_guard, dtype_code, kv_dtype_code, ptr, stream_of

_libs = []  # keep torch.library.Library objects alive
_registered = False


def _req(cond: bool, msg: str) -> None:
    if not cond:
        raise _lib.NmvError(msg)


# ----------------------------------------------------------------------------- attention
def _pa_common(query, key_cache, value_cache, block_tables, seq_lens, blocksparse_vert_stride,
               blocksparse_block_size, block_size):
    # attention_kernels.cu:235: the sparsity block of a KV block is block_idx * BLOCK_SIZE / blocksparse_block_size
    _req(blocksparse_vert_stride <= 1 or (blocksparse_block_size > 0 and blocksparse_block_size % block_size == 0),
         "blocksparse_block_size must be a positive multiple of the KV block size")
    _req(block_tables.dtype == torch.int32 and seq_lens.dtype == torch.int32,
         "block_tables and seq_lens must be int32")
    _req(query.stride(-1) == 1 and query.stride(1) == query.shape[2],
         "query must be [num_seqs, num_heads, head_size] with contiguous heads")
    _req(key_cache.is_contiguous() or key_cache.stride(-1) == 1, "key_cache layout")


def paged_attention_v1(out, query, key_cache, value_cache, num_kv_heads, scale, block_tables,
                       seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype, kv_scale,
                       tp_rank, blocksparse_local_blocks, blocksparse_vert_stride,
                       blocksparse_block_size, blocksparse_head_sliding_step) -> None:
    """csrc/attention/attention_kernels.cu:805-826"""
    _pa_common(query, key_cache, value_cache, block_tables, seq_lens, blocksparse_vert_stride,
               blocksparse_block_size, block_size)
    num_seqs, num_heads, head_size = query.shape
    _req(out.is_contiguous(), "out must be contiguous")
    L = _lib.load()
    with device_guard(query):
        check(L.nmv_paged_attention_v1(
            ptr(out), ptr(query), ptr(key_cache), ptr(value_cache), num_seqs, num_heads,
            head_size, num_kv_heads, scale, ptr(block_tables), ptr(seq_lens), block_size,
            max_seq_len, block_tables.shape[1], ptr(alibi_slopes), query.stride(0),
            key_cache.stride(0), key_cache.stride(1), dtype_code(query.dtype),
            kv_dtype_code(kv_cache_dtype), kv_scale, int(tp_rank), int(blocksparse_local_blocks),
            int(blocksparse_vert_stride), int(blocksparse_block_size), int(blocksparse_head_sliding_step),
            stream_of(query)))


def paged_attention_v2(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache,
                       num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                       alibi_slopes, kv_cache_dtype, kv_scale, tp_rank, blocksparse_local_blocks,
                       blocksparse_vert_stride, blocksparse_block_size,
                       blocksparse_head_sliding_step) -> None:
    """csrc/attention/attention_kernels.cu:966-990"""
    _pa_common(query, key_cache, value_cache, block_tables, seq_lens, blocksparse_vert_stride,
               blocksparse_block_size, block_size)
    num_seqs, num_heads, head_size = query.shape
    _req(out.is_contiguous() and tmp_out.is_contiguous() and exp_sums.is_contiguous()
         and max_logits.is_contiguous(), "out/tmp_out/exp_sums/max_logits must be contiguous")
    max_parts = (max_seq_len + 511) // 512
    _req(exp_sums.shape[-1] >= max_parts and tmp_out.shape[2] >= max_parts,
         "partition buffers too small for max_seq_len")
    _req(exp_sums.shape[-1] == max_parts and tmp_out.shape[2] == max_parts,
         "partition buffers must be sized ceil(max_seq_len / 512)")
    L = _lib.load()
    with device_guard(query):
        check(L.nmv_paged_attention_v2(
            ptr(out), ptr(exp_sums), ptr(max_logits), ptr(tmp_out), ptr(query), ptr(key_cache),
            ptr(value_cache), num_seqs, num_heads, head_size, num_kv_heads, scale,
            ptr(block_tables), ptr(seq_lens), block_size, max_seq_len, block_tables.shape[1],
            ptr(alibi_slopes), query.stride(0), key_cache.stride(0), key_cache.stride(1),
            dtype_code(query.dtype), kv_dtype_code(kv_cache_dtype), kv_scale, int(tp_rank),
            int(blocksparse_local_blocks), int(blocksparse_vert_stride), int(blocksparse_block_size),
            int(blocksparse_head_sliding_step), stream_of(query)))


def paged_attention_rope_partial(out, slab, positions, cos_sin_cache, slot_mapping, key_cache, value_cache,
                                 num_heads, num_kv_heads, head_size, scale, block_tables, seq_lens,
                                 block_size, max_seq_len, kv_cache_dtype, kv_scale, partition_bufs=None) -> None:
    """rotary_embedding + reshape_and_cache + paged_attention_v1 / v2 in one launch, from the qkv
    projection's fp32 split-K slabs (include/nmvllm_hip.h: nmv_paged_attention_v*_rope_partial);
    partition_bufs = (exp_sums, max_logits, tmp_out) selects v2"""
    if slab.dtype == torch.float32:
        _req(slab.dim() == 3 and slab.is_contiguous(), "pa_rope_partial: slab [S, B, N] fp32")
        s_, b, n = slab.shape
    else:  # the finished qkv row in the model dtype
        _req(slab.dim() == 2 and slab.is_contiguous() and slab.dtype == out.dtype,
             "pa_rope_partial: qkv [B, N] contiguous in the model dtype")
        s_, (b, n) = 0, slab.shape
    _req(n == (num_heads + 2 * num_kv_heads) * head_size, "pa_rope_partial: slab width != qkv width")
    _req(out.shape == (b, num_heads, head_size) and out.is_contiguous(), "pa_rope_partial: out [B, H, D]")
    _req(positions.dtype == torch.int64 and slot_mapping.dtype == torch.int64
         and positions.numel() == b and slot_mapping.numel() == b, "pa_rope_partial: int64 [B] positions / slots")
    _req(cos_sin_cache.dtype == out.dtype and cos_sin_cache.shape[1] == head_size and cos_sin_cache.is_contiguous(),
         "pa_rope_partial: cos_sin_cache [max_pos, head_size] in the model dtype (rot_dim == head_size)")
    _req(block_tables.dtype == torch.int32 and seq_lens.dtype == torch.int32, "block_tables / seq_lens must be int32")
    _req(key_cache.is_contiguous() and value_cache.is_contiguous(), "caches must be contiguous")
    L = _lib.load()
    with device_guard(slab):
        if partition_bufs is None:
            check(L.nmv_paged_attention_v1_rope_partial(
                ptr(out), ptr(slab), s_, ptr(positions), ptr(cos_sin_cache), ptr(slot_mapping), ptr(key_cache),
                ptr(value_cache), b, num_heads, head_size, num_kv_heads, scale, ptr(block_tables),
                ptr(seq_lens), block_size, max_seq_len, block_tables.shape[1], key_cache.stride(0),
                key_cache.stride(1), dtype_code(out.dtype), kv_dtype_code(kv_cache_dtype), kv_scale,
                stream_of(slab)))
        else:
            exp_sums, max_logits, tmp_out = partition_bufs
            check(L.nmv_paged_attention_v2_rope_partial(
                ptr(out), ptr(exp_sums), ptr(max_logits), ptr(tmp_out), ptr(slab), s_, ptr(positions),
                ptr(cos_sin_cache), ptr(slot_mapping), ptr(key_cache), ptr(value_cache), b, num_heads,
                head_size, num_kv_heads, scale, ptr(block_tables), ptr(seq_lens), block_size, max_seq_len,
                block_tables.shape[1], key_cache.stride(0), key_cache.stride(1), dtype_code(out.dtype),
                kv_dtype_code(kv_cache_dtype), kv_scale, stream_of(slab)))


# ----------------------------------------------------------------------------- glue
def _rows(t: torch.Tensor) -> int:
    return t.numel() // t.shape[-1] if t.numel() else 0


def rms_norm(out, input, weight, epsilon) -> None:
    """csrc/layernorm_kernels.cu:292-313"""
    _req(input.is_contiguous() and out.is_contiguous(), "rms_norm: tensors must be contiguous")
    with device_guard(input):
        check(_lib.load().nmv_rms_norm(ptr(out), ptr(input), ptr(weight), epsilon, _rows(input),
                                       input.shape[-1], dtype_code(input.dtype),
                                       stream_of(input)))


def fused_add_rms_norm(input, residual, weight, epsilon) -> None:
    """csrc/layernorm_kernels.cu:315-352"""
    _req(input.is_contiguous() and residual.is_contiguous(),
         "fused_add_rms_norm: tensors must be contiguous")
    with device_guard(input):
        check(_lib.load().nmv_fused_add_rms_norm(ptr(input), ptr(residual), ptr(weight), epsilon,
                                                 _rows(input), input.shape[-1],
                                                 dtype_code(input.dtype), stream_of(input)))


def _rope_args(positions, query, key, head_size, cos_sin_cache):
    num_tokens = query.numel() // query.shape[-1]
    _req(positions.dtype == torch.int64, "positions must be int64")
    _req(positions.numel() == num_tokens, "positions / query token count mismatch")
    rot_dim = cos_sin_cache.shape[1]
    num_heads = query.shape[-1] // head_size
    num_kv_heads = key.shape[-1] // head_size
    return num_tokens, rot_dim, num_heads, num_kv_heads, query.stride(-2), key.stride(-2)


def rotary_embedding(positions, query, key, head_size, cos_sin_cache, is_neox) -> None:
    """csrc/pos_encoding_kernels.cu:121-160"""
    nt, rot, nh, nkv, qs, ks = _rope_args(positions, query, key, head_size, cos_sin_cache)
    _req(cos_sin_cache.dtype == query.dtype, "cos_sin_cache dtype must match query")
    with device_guard(query):
        check(_lib.load().nmv_rotary_embedding(
            ptr(positions), ptr(query), ptr(key), nt, nh, nkv, head_size, rot, qs, ks,
            ptr(cos_sin_cache), int(is_neox), dtype_code(query.dtype), stream_of(query)))


def batched_rotary_embedding(positions, query, key, head_size, cos_sin_cache, is_neox, rot_dim,
                             cos_sin_cache_offsets) -> None:
    """csrc/pos_encoding_kernels.cu:162-203"""
    nt, _, nh, nkv, qs, ks = _rope_args(positions, query, key, head_size, cos_sin_cache)
    _req(cos_sin_cache_offsets.dtype == torch.int64, "cos_sin_cache_offsets must be int64")
    with device_guard(query):
        check(_lib.load().nmv_batched_rotary_embedding(
            ptr(positions), ptr(query), ptr(key), nt, nh, nkv, head_size, rot_dim, qs, ks,
            ptr(cos_sin_cache), int(is_neox), ptr(cos_sin_cache_offsets),
            dtype_code(query.dtype), stream_of(query)))


def _act_and_mul(act: int):

    def fn(out, input) -> None:
        d = input.shape[-1] // 2
        _req(input.is_contiguous() and out.is_contiguous(), "act_and_mul: contiguous tensors")
        with device_guard(input):
            check(_lib.load().nmv_act_and_mul(ptr(out), ptr(input), _rows(input), d, act,
                                              dtype_code(input.dtype), stream_of(input)))

    return fn


def _activation(act: int):

    def fn(out, input) -> None:
        _req(input.is_contiguous() and out.is_contiguous(), "activation: contiguous tensors")
        with device_guard(input):
            check(_lib.load().nmv_activation(ptr(out), ptr(input), _rows(input), input.shape[-1],
                                             act, dtype_code(input.dtype), stream_of(input)))

    return fn


# ----------------------------------------------------------------------------- W4A16
def gptq_marlin_repack(b_q_weight, perm, size_k, size_n, num_bits) -> torch.Tensor:
    """csrc/quantization/gptq_marlin/gptq_marlin_repack.cu:267-348"""
    _req(num_bits in (4, 8), f"num_bits must be 4 or 8. Got = {num_bits}")
    pack = 32 // num_bits
    _req(size_k % 16 == 0, f"size_k = {size_k} is not divisible by tile_k_size = 16")
    _req(size_n % 64 == 0, f"size_n = {size_n} is not divisible by tile_n_size = 64")
    _req(tuple(b_q_weight.shape) == (size_k // pack, size_n),
         f"Shape mismatch: b_q_weight {tuple(b_q_weight.shape)} vs "
         f"{(size_k // pack, size_n)}")
    _req(b_q_weight.is_contiguous() and b_q_weight.dtype == torch.int32,
         "b_q_weight must be contiguous int32")
    has_perm = perm.numel() != 0
    if has_perm:
        _req(perm.dtype == torch.int32 and perm.numel() == size_k and perm.is_contiguous(),
             "perm must be contiguous int32 [size_k]")
    out = torch.empty((size_k // 16, size_n * 16 // pack), dtype=torch.int32,
                      device=b_q_weight.device)
    with device_guard(b_q_weight):
        check(_lib.load().nmv_gptq_marlin_repack(ptr(b_q_weight), ptr(perm) if has_perm else None,
                                                 ptr(out), size_k, size_n, num_bits,
                                                 stream_of(b_q_weight)))
    return out


def gptq_marlin_gemm(a, b_q_weight, b_scales, g_idx, perm, workspace, num_bits, size_m, size_n,
                     size_k, is_k_full) -> torch.Tensor:
    """csrc/quantization/gptq_marlin/gptq_marlin.cu:1735-1868 (same argument checks)"""
    _req(num_bits in (4, 8), f"num_bits must be 4 or 8. Got = {num_bits}")
    pack = 32 // num_bits
    _req(a.shape[0] == size_m, f"Shape mismatch: a.size(0) = {a.shape[0]}, size_m = {size_m}")
    _req(a.shape[1] == size_k, f"Shape mismatch: a.size(1) = {a.shape[1]}, size_k = {size_k}")
    _req(size_k % 16 == 0, f"size_k = {size_k} is not divisible by tile_size = 16")
    _req(b_q_weight.shape[0] == size_k // 16,
         f"Shape mismatch: b_q_weight.size(0) = {b_q_weight.shape[0]}, size_k = {size_k}")
    _req(b_q_weight.shape[1] % 16 == 0, "b_q_weight.size(1) is not divisible by tile_size = 16")
    _req(b_q_weight.shape[1] // 16 * pack == size_n,
         f"size_n = {size_n}, actual_size_n = {b_q_weight.shape[1] // 16 * pack}")
    _req(a.is_contiguous(), "A is not contiguous")
    _req(b_q_weight.is_contiguous() and b_q_weight.dtype == torch.int32,
         "b_q_weight must be contiguous int32")
    _req(b_scales.is_contiguous() and b_scales.dtype == a.dtype,
         "b_scales must be contiguous and of A's dtype")
    _req(a.dtype in (torch.float16, torch.bfloat16), "gpt_marlin_gemm only supports bfloat16 and float16")
    _req(size_n % 64 == 0, f"size_n = {size_n} is not divisible by min_thread_n = 64")
    _req(workspace.dtype == torch.int32 and workspace.is_contiguous(), "workspace must be int32")
    _req(workspace.numel() >= size_n // 64 * 16,
         f"workspace.numel = {workspace.numel()} is below min_workspace_size = {size_n // 64 * 16}")
    has_act_order = g_idx.numel() != 0 and perm.numel() != 0
    if has_act_order:
        _req(g_idx.numel() == size_k and perm.numel() == size_k,
             "Unexpected g_idx.size / perm.size")
        _req(g_idx.dtype == torch.int32 and perm.dtype == torch.int32, "g_idx/perm must be int32")
    else:
        _req(g_idx.numel() == 0 and perm.numel() == 0,
             "g_idx and perm must both be empty or both be set")
    num_groups = b_scales.shape[0]
    _req(b_scales.shape[1] == size_n, "b_scales dim 1 != size_n")
    if has_act_order:
        if is_k_full:
            _req(num_groups > 1, "For act_order, num_groups must be > 1")
            _req(size_k % num_groups == 0, f"size_k = {size_k}, is not divisible by num_groups")
    elif num_groups > 1:
        _req(size_k % num_groups == 0, f"size_k = {size_k}, is not divisible by num_groups")
        _req(size_k // num_groups in (32, 64, 128), "group_size must be 32, 64 or 128")
    c = torch.empty((size_m, size_n), dtype=a.dtype, device=a.device)
    if size_m == 0:
        return c
    L = _lib.load()
    nbytes = L.nmv_gptq_marlin_gemm_scratch_bytes(size_m, size_n, size_k,
                                                  int(has_act_order) | (2 if num_bits == 8 else 0))
    scratch = torch.empty((max(nbytes, 16), ), dtype=torch.uint8, device=a.device)
    with device_guard(a):
        check(L.nmv_gptq_marlin_gemm(
            ptr(c), ptr(a), ptr(b_q_weight), ptr(b_scales),
            ptr(g_idx) if has_act_order else None, ptr(perm) if has_act_order else None,
            ptr(workspace), workspace.numel(), ptr(scratch), scratch.numel(), num_bits, size_m,
            size_n, size_k, num_groups, int(is_k_full), dtype_code(a.dtype), stream_of(a)))
    return c


def marlin_zp_gemm(a, b_q_weight, b_scales, b_zeros, workspace, size_m, size_n, size_k) -> torch.Tensor:
    """Marlin-format GEMM with zero points (csrc/w4a16_gemm.hip; not an op of nm-vllm 0.5.1):
    b_zeros holds z in A's dtype, [groups, N] in marlin_permute_scales order; 4-bit, group 128."""
    _req(a.shape[0] == size_m and a.shape[1] == size_k and a.is_contiguous(), "marlin_zp_gemm: bad a")
    _req(b_q_weight.dtype == torch.int32 and b_q_weight.is_contiguous()
         and b_q_weight.shape == (size_k // 16, size_n * 2), "marlin_zp_gemm: b_q_weight must be int32 [K/16, 2N]")
    _req(b_scales.dtype == a.dtype and b_zeros.dtype == a.dtype and b_scales.is_contiguous()
         and b_zeros.is_contiguous() and b_scales.shape == b_zeros.shape and b_scales.shape[1] == size_n,
         "marlin_zp_gemm: scales / zeros must be [groups, N] in A's dtype")
    _req(workspace.dtype == torch.int32 and workspace.numel() >= size_n // 64 * 16, "marlin_zp_gemm: workspace")
    c = torch.empty((size_m, size_n), dtype=a.dtype, device=a.device)
    if size_m == 0:
        return c
    L = _lib.load()
    nbytes = L.nmv_gptq_marlin_gemm_scratch_bytes(size_m, size_n, size_k, 0)
    scratch = torch.empty((max(nbytes, 16), ), dtype=torch.uint8, device=a.device)
    with device_guard(a):
        check(L.nmv_marlin_zp_gemm(ptr(c), ptr(a), ptr(b_q_weight), ptr(b_scales), ptr(b_zeros),
                                   ptr(workspace), workspace.numel(), ptr(scratch), scratch.numel(),
                                   size_m, size_n, size_k, b_scales.shape[0], dtype_code(a.dtype),
                                   stream_of(a)))
    return c


def awq_marlin_repack(qweight, size_k, size_n) -> torch.Tensor:
    """AWQ int32 [K, N/8] -> Marlin int32 [K/16, 2N]"""
    _req(qweight.dtype == torch.int32 and qweight.is_contiguous() and qweight.shape == (size_k, size_n // 8),
         "awq_marlin_repack: qweight must be int32 [K, N/8]")
    out = torch.empty((size_k // 16, size_n * 2), dtype=torch.int32, device=qweight.device)
    with device_guard(qweight):
        check(_lib.load().nmv_awq_marlin_repack(ptr(out), ptr(qweight), size_k, size_n, stream_of(qweight)))
    return out


def marlin_gemm(a, b_q_weight, b_scales, workspace, size_m, size_n, size_k) -> torch.Tensor:
    """csrc/quantization/marlin/dense/marlin_cuda_kernel.cu:1045-1136 (legacy Marlin, 4-bit)"""
    _req(a.shape[0] == size_m and a.shape[1] == size_k, "Shape mismatch: a vs size_m / size_k")
    _req(size_k % 16 == 0 and b_q_weight.shape[0] == size_k // 16,
         f"Shape mismatch: b_q_weight.size(0) = {b_q_weight.shape[0]}, size_k = {size_k}")
    _req(b_q_weight.shape[1] // 16 * 8 == size_n, "size_n does not match b_q_weight")
    _req(a.is_contiguous() and b_q_weight.is_contiguous() and b_scales.is_contiguous(),
         "a, b_q_weight and b_scales must be contiguous")
    _req(a.dtype in (torch.float16, torch.bfloat16) and b_scales.dtype == a.dtype,
         "marlin_gemm supports float16 (and bfloat16) activations with scales of the same dtype")
    _req(workspace.numel() >= size_n // 128 * 16 or workspace.numel() >= size_n // 64,
         "workspace is too small")
    c = torch.empty((size_m, size_n), dtype=a.dtype, device=a.device)
    if size_m == 0:
        return c
    L = _lib.load()
    nbytes = L.nmv_gptq_marlin_gemm_scratch_bytes(size_m, size_n, size_k, 0)
    scratch = torch.empty((max(nbytes, 16), ), dtype=torch.uint8, device=a.device)
    with device_guard(a):
        check(L.nmv_marlin_gemm(ptr(c), ptr(a), ptr(b_q_weight), ptr(b_scales), ptr(workspace),
                                workspace.numel(), ptr(scratch), scratch.numel(), size_m, size_n,
                                size_k, b_scales.shape[0], dtype_code(a.dtype), stream_of(a)))
    return c


def fp8_marlin_gemm(a, b_q_weight, b_scales, workspace, num_bits, size_m, size_n,
                    size_k) -> torch.Tensor:
    """csrc/quantization/fp8/fp8_marlin.cu:1212-1308"""
    _req(a.shape[0] == size_m and a.shape[1] == size_k, "Shape mismatch: a vs size_m / size_k")
    _req(b_q_weight.shape[0] == size_k // 16 and b_q_weight.shape[1] // 16 * 4 == size_n,
         "Shape mismatch: b_q_weight vs size_k / size_n")
    _req(a.is_contiguous() and b_q_weight.is_contiguous() and b_scales.is_contiguous(),
         "a, b_q_weight and b_scales must be contiguous")
    _req(b_scales.dtype == a.dtype and b_scales.shape[-1] == size_n, "b_scales must be [G, size_n] of A's dtype")
    c = torch.empty((size_m, size_n), dtype=a.dtype, device=a.device)
    if size_m == 0:
        return c
    L = _lib.load()
    nbytes = int(L.nmv_fp8_marlin_gemm_scratch_bytes(size_m, size_n, size_k))
    scratch = torch.empty((max(nbytes, 16), ), dtype=torch.uint8, device=a.device)
    with device_guard(a):
        check(L.nmv_fp8_marlin_gemm(ptr(c), ptr(a), ptr(b_q_weight), ptr(b_scales), ptr(workspace),
                                    workspace.numel(), ptr(scratch), scratch.numel(), num_bits, size_m, size_n,
                                    size_k, b_scales.shape[0], dtype_code(a.dtype), stream_of(a)))
    return c


# ----------------------------------------------------------------------------- GPTQ / AWQ
def gptq_gemm(a, b_q_weight, b_gptq_qzeros, b_gptq_scales, b_g_idx, use_exllama, bit) -> torch.Tensor:
    """csrc/quantization/gptq/q_gemm.cu:1823-1846"""
    _req(bit in (2, 3, 4, 8), f"unsupported bit width {bit}")
    size_m, size_k = a.shape[0], a.shape[1]
    size_n = b_q_weight.shape[1]
    # 3-bit: 32 codes per three int32 rows (qweight is [K * 3 / 32, N])
    _req(b_q_weight.shape[0] * 32 == size_k * bit, "b_q_weight rows do not match a's K")
    _req(a.is_contiguous() and b_q_weight.is_contiguous() and b_gptq_scales.is_contiguous()
         and b_gptq_qzeros.is_contiguous(), "gptq_gemm: tensors must be contiguous")
    _req(b_gptq_scales.dtype == a.dtype, "scales must have the activation dtype")
    has_idx = b_g_idx is not None and b_g_idx.numel() > 0 and b_g_idx.device.type != "meta"
    if has_idx:
        _req(b_g_idx.numel() == size_k and b_g_idx.dtype == torch.int32, "g_idx must be int32 [K]")
    c = torch.empty((size_m, size_n), dtype=a.dtype, device=a.device)
    lib = _lib.load()
    nbytes = int(lib.nmv_wq_gemm_scratch_bytes(size_m, size_n, size_k))
    scratch = torch.empty((nbytes, ), dtype=torch.uint8, device=a.device) if nbytes else None
    with device_guard(a):
        check(lib.nmv_gptq_gemm(ptr(c), ptr(a), ptr(b_q_weight), ptr(b_gptq_qzeros),
                                ptr(b_gptq_scales), ptr(b_g_idx) if has_idx else None,
                                int(use_exllama), bit, size_m, size_n, size_k,
                                b_gptq_scales.shape[0], dtype_code(a.dtype),
                                ptr(scratch) if nbytes else None, nbytes, stream_of(a)))
    return c


def gptq_shuffle(q_weight, q_perm, bit) -> None:
    """csrc/quantization/gptq/q_gemm.cu:1848-1856 (in place)"""
    _req(bit in (2, 3, 4, 8), f"gptq_shuffle: {bit}-bit weights are not supported")
    has_perm = q_perm is not None and q_perm.numel() > 0 and q_perm.device.type != "meta"
    if not has_perm:
        return
    size_k = q_weight.shape[0] * 32 // bit
    _req(q_perm.dtype == torch.int32 and q_perm.numel() == size_k, "q_perm must be int32 [K]")
    tmp = torch.empty_like(q_weight)
    with device_guard(q_weight):
        check(_lib.load().nmv_gptq_shuffle(ptr(q_weight), ptr(q_perm), ptr(tmp), size_k, q_weight.shape[1], bit,
                                           stream_of(q_weight)))


def awq_gemm(input, kernel, scaling_factors, zeros, split_k_iters) -> torch.Tensor:
    """csrc/quantization/awq/gemm_kernels.cu:492-549; argument order of csrc/ops.h:66-68"""
    size_m, size_k = input.shape[0], input.shape[1]
    size_n = kernel.shape[1] * 8
    _req(kernel.shape[0] == size_k, "awq_gemm: qweight rows must equal K")
    _req(scaling_factors.shape[1] == size_n and zeros.shape[1] * 8 == size_n, "awq_gemm: scales/zeros shape")
    _req(input.is_contiguous() and kernel.is_contiguous() and scaling_factors.is_contiguous()
         and zeros.is_contiguous(), "awq_gemm: tensors must be contiguous")
    _req(scaling_factors.dtype == input.dtype, "scales must have the activation dtype")
    c = torch.empty((size_m, size_n), dtype=input.dtype, device=input.device)
    lib = _lib.load()
    nbytes = int(lib.nmv_wq_gemm_scratch_bytes(size_m, size_n, size_k))
    scratch = torch.empty((nbytes, ), dtype=torch.uint8, device=input.device) if nbytes else None
    with device_guard(input):
        check(lib.nmv_awq_gemm(ptr(c), ptr(input), ptr(kernel), ptr(scaling_factors),
                               ptr(zeros), size_m, size_n, size_k,
                               scaling_factors.shape[0], dtype_code(input.dtype),
                               ptr(scratch) if nbytes else None, nbytes, stream_of(input)))
    return c


def _alibi_arg(alibi_slopes, num_heads, like):
    if alibi_slopes is None:
        return None
    _req(alibi_slopes.dtype == torch.float32 and alibi_slopes.numel() == num_heads and alibi_slopes.is_contiguous()
         and alibi_slopes.device == like.device, "alibi_slopes must be float32 [num_heads] on the query's device")
    return ptr(alibi_slopes)


def prefill_attention(out, query, key, value, cu_seqlens, max_seq_len, scale, alibi_slopes=None,
                      sliding_window=None) -> None:
    """varlen causal GQA prompt attention (csrc/prefill_attention.hip); q [T, H, D], k/v [T, KVH, D]
    (last dim contiguous, may be slices of qkv), cu_seqlens int32 [num_seqs + 1]; optional ALiBi slopes
    (float32 [H]) and sliding window (keys fewer than that many positions back)"""
    _req(query.dim() == 3 and key.dim() == 3 and value.dim() == 3 and out.dim() == 3, "prefill_attention: [T, H, D] tensors")
    _req(query.stride(2) == 1 and key.stride(2) == 1 and value.stride(2) == 1 and out.stride(2) == 1,
         "prefill_attention: head dim must be contiguous")
    t, h, d = query.shape
    kvh = key.shape[1]
    _req(query.stride(1) == d and key.stride(1) == d and value.stride(1) == d and out.stride(1) == d,
         "prefill_attention: heads must be packed")
    _req(key.stride(0) == value.stride(0), "prefill_attention: k and v must share the token stride")
    _req(cu_seqlens.dtype == torch.int32 and cu_seqlens.is_contiguous(), "cu_seqlens must be int32")
    with device_guard(query):
        check(_lib.load().nmv_prefill_attention(ptr(out), ptr(query), ptr(key), ptr(value), ptr(cu_seqlens),
                                                cu_seqlens.numel() - 1, int(max_seq_len), h, kvh, d,
                                                float(scale), query.stride(0), key.stride(0),
                                                out.stride(0), _alibi_arg(alibi_slopes, h, query),
                                                int(sliding_window or 0), dtype_code(query.dtype),
                                                stream_of(query)))


def prefix_prefill_attention(out, query, key_cache, value_cache, block_tables, query_start_loc,
                             seq_lens, context_lens, max_query_len, scale, alibi_slopes=None,
                             sliding_window=None) -> None:
    """PagedAttention.forward_prefix on the paged cache (csrc/prefill_attention.hip); query/out
    [new tokens, H, D]; key_cache [NB, KVH, D/x, BS, x], value_cache [NB, KVH, D, BS] (16-bit)"""
    _req(query.dim() == 3 and out.dim() == 3 and query.stride(2) == 1 and out.stride(2) == 1,
         "prefix_prefill_attention: [T, H, D] tensors with a contiguous head dim")
    t, h, d = query.shape
    _req(query.stride(1) == d and out.stride(1) == d, "prefix_prefill_attention: heads must be packed")
    _req(key_cache.dtype == query.dtype and value_cache.dtype == query.dtype,
         "prefix_prefill_attention: kv cache dtype must be auto (the model dtype), as for the reference's "
         "forward_prefix (vllm/attention/ops/paged_attn.py:184-216 passes no cache dtype)")
    nb, kvh, _, bs = value_cache.shape
    for x in (block_tables, query_start_loc, seq_lens, context_lens):
        _req(x.dtype == torch.int32 and x.is_contiguous(), "prefix_prefill_attention: int32 index tensors")
    with device_guard(query):
        check(_lib.load().nmv_prefix_prefill_attention(
            ptr(out), ptr(query), ptr(key_cache), ptr(value_cache), ptr(block_tables),
            ptr(query_start_loc), ptr(seq_lens), ptr(context_lens), seq_lens.numel(),
            int(max_query_len), block_tables.shape[1], bs, h, kvh, d, float(scale), query.stride(0),
            out.stride(0), value_cache.stride(0), value_cache.stride(1), _alibi_arg(alibi_slopes, h, query),
            int(sliding_window or 0), dtype_code(query.dtype), stream_of(query)))


def prefill_attention_supported(head_size: int) -> bool:
    return bool(_lib.load().nmv_prefill_attention_supported(int(head_size)))


def awq_dequantize(kernel, scaling_factors, zeros, split_k_iters, thx, thy) -> torch.Tensor:
    """csrc/quantization/awq/gemm_kernels.cu:436-490 -> [K, N]"""
    size_k, size_n = kernel.shape[0], kernel.shape[1] * 8
    out = torch.empty((size_k, size_n), dtype=scaling_factors.dtype, device=kernel.device)
    with device_guard(kernel):
        check(_lib.load().nmv_awq_dequantize(ptr(out), ptr(kernel), ptr(scaling_factors),
                                             ptr(zeros), size_n, size_k, scaling_factors.shape[0],
                                             dtype_code(scaling_factors.dtype), stream_of(kernel)))
    return out


# ----------------------------------------------------------------------------- W8A8
def static_scaled_int8_quant(out, input, scale) -> None:
    """csrc/quantization/compressed_tensors/int8_quant_kernels.cu:77-95"""
    _req(input.is_contiguous() and out.is_contiguous(), "input/out must be contiguous")
    _req(scale.numel() == 1 and scale.dtype == torch.float32, "scale must be one float32")
    _req(out.dtype == torch.int8, "out must be int8")
    with device_guard(input):
        check(_lib.load().nmv_scaled_int8_quant(ptr(out), ptr(input), ptr(scale), _rows(input),
                                                input.shape[-1], 0, dtype_code(input.dtype),
                                                stream_of(input)))


def dynamic_scaled_int8_quant(out, input, scale) -> None:
    """int8_quant_kernels.cu:97-115: per-token scales"""
    _req(input.is_contiguous() and out.is_contiguous(), "input/out must be contiguous")
    _req(scale.dtype == torch.float32 and scale.is_contiguous() and scale.numel() >= _rows(input),
         "scales must be float32 [num_tokens, 1]")
    _req(out.dtype == torch.int8, "out must be int8")
    with device_guard(input):
        check(_lib.load().nmv_scaled_int8_quant(ptr(out), ptr(input), ptr(scale), _rows(input),
                                                input.shape[-1], 1, dtype_code(input.dtype),
                                                stream_of(input)))


def _fp8_quant(dynamic: int):

    def fn(out, input, scale) -> None:
        """csrc/quantization/fp8/common.cu:129-165"""
        _req(input.is_contiguous() and out.is_contiguous(), "input/out must be contiguous")
        _req(out.dtype in (torch.float8_e4m3fn, torch.uint8), "out must be float8_e4m3fn")
        _req(scale.numel() == 1 and scale.dtype == torch.float32, "scale must be one float32")
        _req(out.numel() >= input.numel(), "out is smaller than input")
        with device_guard(input):
            check(_lib.load().nmv_scaled_fp8_quant(ptr(out), ptr(input), ptr(scale), input.numel(),
                                                   dynamic, dtype_code(input.dtype),
                                                   stream_of(input)))

    return fn


def cutlass_scaled_mm_supports_fp8(cuda_device_capability: int) -> bool:
    return bool(_lib.load().nmv_cutlass_scaled_mm_supports_fp8(cuda_device_capability))


def cutlass_scaled_mm(out, a, b, a_scales, b_scales, bias) -> None:
    """csrc/quantization/cutlass_w8a8/scaled_mm_entry.cu:48-100 (same checks)"""
    _req(a.dim() == 2 and b.dim() == 2 and out.dim() == 2, "a, b, c must be 2-D")
    _req(out.shape[0] == a.shape[0] and a.shape[1] == b.shape[0] and b.shape[1] == out.shape[1],
         "shape mismatch between a, b and c")
    _req(a_scales.numel() == 1 or a_scales.numel() == a.shape[0], "a_scales must be scalar or [M]")
    _req(b_scales.numel() == 1 or b_scales.numel() == b.shape[1], "b_scales must be scalar or [N]")
    _req(a.stride(1) == 1 and out.stride(1) == 1, "a and c must be row major")
    _req(b.stride(0) == 1, "b must be column major")
    _req(out.stride(0) % 16 == 0 and b.stride(1) % 16 == 0, "c.stride(0) and b.stride(1) must be 16B aligned")
    _req(a_scales.is_contiguous() and b_scales.is_contiguous(), "scales must be contiguous")
    _req(a_scales.dtype == torch.float32 and b_scales.dtype == torch.float32, "scales must be float32")
    if bias is not None:
        _req(bias.numel() == b.shape[1] and bias.is_contiguous() and bias.dim() == 1,
             "bias must be a contiguous [N] vector")
        _req(bias.dtype == out.dtype, "bias dtype must match the output")
    if a.dtype == torch.int8:
        _req(b.dtype == torch.int8, "a and b must both be int8")
        q = _lib.NMV_I8
    else:
        _req(a.dtype == torch.float8_e4m3fn and b.dtype == torch.float8_e4m3fn,
             "a and b must both be int8 or both float8_e4m3fn")
        q = _lib.NMV_FP8_E4M3
    m, k = a.shape
    n = b.shape[1]
    with device_guard(a):
        lib = _lib.load()
        sb = lib.nmv_scaled_mm_scratch_bytes(m, n, k)
        scratch = torch.empty(sb, dtype=torch.uint8, device=a.device) if sb else None
        check(lib.nmv_scaled_mm(ptr(out), ptr(a), ptr(b), ptr(a_scales), ptr(b_scales),
                                ptr(bias), m, n, k, a.stride(0), b.stride(1),
                                out.stride(0), a_scales.numel(), b_scales.numel(), q,
                                dtype_code(out.dtype), ptr(scratch), sb, stream_of(a)))


# ----------------------------------------------------------------------------- cache ops
def reshape_and_cache(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype,
                      kv_scale) -> None:
    """csrc/cache_kernels.cu:253-278"""
    num_tokens, num_heads, head_size = key.shape
    block_size = key_cache.shape[3]
    _req(slot_mapping.dtype == torch.int64, "slot_mapping must be int64")
    _req(key.stride(-1) == 1 and key.stride(1) == head_size and value.stride(-1) == 1
         and value.stride(1) == head_size, "key/value heads must be contiguous")
    _req(key_cache.is_contiguous() and value_cache.is_contiguous(), "caches must be contiguous")
    with device_guard(key):
        check(_lib.load().nmv_reshape_and_cache(
            ptr(key), ptr(value), ptr(key_cache), ptr(value_cache), ptr(slot_mapping),
            slot_mapping.numel(), num_heads, head_size, block_size, key.stride(0),
            value.stride(0), dtype_code(key.dtype), kv_dtype_code(kv_cache_dtype), kv_scale,
            stream_of(key)))


def reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping,
                            kv_cache_dtype) -> None:
    """csrc/cache_kernels.cu:280-316"""
    _req(kv_cache_dtype == "auto", "FlashAttention does not support FP8 kv-cache")
    num_tokens, num_heads, head_size = key.shape
    block_size = key_cache.shape[1]
    _req(key_cache.stride(0) == value_cache.stride(0), "k/v cache block strides differ")
    with device_guard(key):
        check(_lib.load().nmv_reshape_and_cache_flash(
            ptr(key), ptr(value), ptr(key_cache), ptr(value_cache), ptr(slot_mapping),
            slot_mapping.numel(), num_heads, head_size, block_size, key.stride(0),
            value.stride(0), key_cache.stride(0), dtype_code(key.dtype), stream_of(key)))


# ----------------------------------------------------------------------------- fused decode-step launches
def rotary_embedding_and_cache(positions, query, key, value, head_size, cos_sin_cache, is_neox,
                               key_cache, value_cache, slot_mapping, kv_cache_dtype, kv_scale) -> None:
    """rotary_embedding (pos_encoding_kernels.cu:121-160) + reshape_and_cache (cache_kernels.cu:253-278)
    in one launch; query / key [T, heads*head_size] rotated in place, value [T, kv_heads*head_size]"""
    nt, rot, nh, nkv, qs, ks = _rope_args(positions, query, key, head_size, cos_sin_cache)
    _req(cos_sin_cache.dtype == query.dtype, "cos_sin_cache dtype must match query")
    _req(query.dim() == 2 and key.dim() == 2 and value.dim() == 2 and value.shape == key.shape,
         "rotary_embedding_and_cache: query / key / value must be [T, heads*head_size]")
    _req(query.stride(-1) == 1 and key.stride(-1) == 1 and value.stride(-1) == 1,
         "rotary_embedding_and_cache: heads must be contiguous")
    _req(slot_mapping.dtype == torch.int64 and slot_mapping.numel() == nt, "slot_mapping must be int64 [T]")
    _req(key_cache.is_contiguous() and value_cache.is_contiguous(), "caches must be contiguous")
    _req(key_cache.shape[1] == nkv and value_cache.shape[1] == nkv, "cache kv-head count mismatch")
    block_size = key_cache.shape[3]
    with device_guard(query):
        check(_lib.load().nmv_rotary_embedding_and_cache(
            ptr(positions), ptr(query), ptr(key), ptr(value), nt, nh, nkv, head_size, rot, qs, ks,
            value.stride(0), ptr(cos_sin_cache), int(is_neox), ptr(key_cache), ptr(value_cache),
            ptr(slot_mapping), block_size, dtype_code(query.dtype), kv_dtype_code(kv_cache_dtype),
            kv_scale, stream_of(query)))


def gptq_marlin_gemm_silu_mul(a, b_q_weight, b_scales, workspace, size_m, size_n, size_k) -> torch.Tensor:
    """gate_up GEMM + silu_and_mul in one launch on column-interleaved Marlin weights
    (include/nmvllm_hip.h: nmv_gptq_marlin_gemm_silu_mul); returns [size_m, size_n // 2]"""
    _req(a.dim() == 2 and a.shape == (size_m, size_k) and a.is_contiguous(), "gemm_silu_mul: bad a")
    _req(a.dtype in (torch.float16, torch.bfloat16), "gemm_silu_mul: fp16 / bf16 only")
    _req(b_q_weight.dtype == torch.int32 and b_q_weight.is_contiguous()
         and b_q_weight.shape == (size_k // 16, size_n * 2), "gemm_silu_mul: b_q_weight must be int32 [K/16, 2N]")
    _req(b_scales.dtype == a.dtype and b_scales.is_contiguous() and b_scales.shape[1] == size_n,
         "gemm_silu_mul: scales must be [groups, N] in A's dtype")
    _req(workspace.dtype == torch.int32 and workspace.numel() >= size_n // 64 * 16, "gemm_silu_mul: workspace")
    c = torch.empty((size_m, size_n // 2), dtype=a.dtype, device=a.device)
    if size_m == 0:
        return c
    with device_guard(a):
        check(_lib.load().nmv_gptq_marlin_gemm_silu_mul(
            ptr(c), ptr(a), ptr(b_q_weight), ptr(b_scales), ptr(workspace), workspace.numel(), size_m,
            size_n, size_k, b_scales.shape[0], dtype_code(a.dtype), stream_of(a)))
    return c


def prefetch_l3(t: torch.Tensor, workgroups: int = 0) -> None:
    """read tensor `t` once on the current stream and store nothing (include/nmvllm_hip.h: nmv_prefetch_l3): a cache hint"""
    _req(t.is_cuda and t.is_contiguous(), "prefetch_l3: a contiguous device tensor")
    nbytes = t.numel() * t.element_size()
    with device_guard(t):
        check(_lib.load().nmv_prefetch_l3(ptr(t), nbytes - nbytes % 16, workgroups, stream_of(t)))


def greedy_sample_advance(logits, input_ids=None, positions=None, seq_lens=None, slot_mapping=None,
                          block_tables=None, block_size=0) -> torch.Tensor:
    """argmax over logits [B, V] (ties -> lowest index) -> int64 [B]; with the state tensors also
    the on-device advance of a decode batch (include/nmvllm_hip.h: nmv_greedy_sample_advance)"""
    _req(logits.dim() == 2 and logits.stride(1) == 1, "greedy_sample: logits must be [B, V] with contiguous rows")
    _req(logits.dtype in (torch.float16, torch.bfloat16), "greedy_sample: fp16 / bf16 logits")
    b, v = logits.shape
    out = torch.empty((b, ), dtype=torch.int64, device=logits.device)
    lib = _lib.load()
    sb = lib.nmv_greedy_sample_scratch_bytes(b)
    scratch = torch.empty(max(sb, 1), dtype=torch.uint8, device=logits.device)
    if positions is not None:
        _req(input_ids.dtype == torch.int64 and positions.dtype == torch.int64
             and slot_mapping.dtype == torch.int64 and seq_lens.dtype == torch.int32
             and block_tables.dtype == torch.int32, "greedy_sample: state tensor dtypes")
        _req(all(t.is_contiguous() and t.numel() == b for t in (input_ids, positions, seq_lens, slot_mapping))
             and block_tables.dim() == 2 and block_tables.is_contiguous() and block_tables.shape[0] == b,
             "greedy_sample: state tensors must be contiguous [B]")
    with device_guard(logits):
        check(lib.nmv_greedy_sample_advance(
            ptr(out), ptr(logits), logits.stride(0), b, v, dtype_code(logits.dtype), ptr(scratch), sb,
            ptr(input_ids), ptr(positions), ptr(seq_lens), ptr(slot_mapping), ptr(block_tables),
            block_tables.shape[1] if block_tables is not None else 0, block_size, stream_of(logits)))
    return out


# ---- MFMA-native W4 tensors (include/nmvllm_hip.h: nmv_w4_native_*; not ops of the reference) ----
def w4_native_repack(qweight, perm, size_k, size_n) -> torch.Tensor:
    """GPTQ qweight int32 [K/8, N] (+ optional act-order row gather) -> native int32 [K/8 * N]"""
    _req(qweight.dtype == torch.int32 and qweight.is_contiguous() and tuple(qweight.shape) == (size_k // 8, size_n),
         "w4_native_repack: qweight must be contiguous int32 [K/8, N]")
    has_perm = perm is not None and perm.numel() > 0
    if has_perm:
        _req(perm.dtype == torch.int32 and perm.numel() == size_k, "w4_native_repack: perm must be int32 [K]")
    out = torch.empty((size_k // 8 * size_n, ), dtype=torch.int32, device=qweight.device)
    with device_guard(qweight):
        check(_lib.load().nmv_w4_native_repack(ptr(qweight), ptr(perm) if has_perm else None, ptr(out), size_k, size_n,
                                               stream_of(qweight)))
    return out


def w4_native_gemm_splits(size_m, size_n, size_k, num_groups=None) -> int:
    """slab count of w4_native_gemm mode 2 (0: unsupported); num_groups = rows of the scale tensor (default: group 128)"""
    if num_groups is None:
        num_groups = max(size_k // 128, 1)
    return int(_lib.load().nmv_w4_native_gemm_splits(size_m, size_n, size_k, num_groups))


def w4_native_prefill_plan(size_m, size_n, size_k) -> bool:
    """does the prompt-sized kernel on the native tensor serve a call of these sizes by the library's default rule"""
    return bool(_lib.load().nmv_w4_native_prefill_plan(size_m, size_n, size_k))


def w4_native_gemm_slab16(size_m, size_n, size_k) -> bool:
    """does w4_native_gemm mode 3 (deferred reduction, slabs in the model dtype) exist for a call of these sizes"""
    return bool(_lib.load().nmv_w4_native_gemm_slab16(size_m, size_n, size_k))


def w4_native_gemm(a, b_native, scales, workspace, size_m, size_n, size_k, mode=0) -> torch.Tensor:
    """mode 0: [M, N]; 1: silu(gate) * up -> [M, N/2]; 2: fp32 split-K slabs [splits, M, N] (deferred reduction);
    3: the same slabs in the model dtype (prompt-sized calls: w4_native_gemm_slab16)"""
    _req(a.is_contiguous() and a.shape == (size_m, size_k) and a.dtype in (torch.float16, torch.bfloat16),
         "w4_native_gemm: a must be contiguous fp16 / bf16 [M, K]")
    _req(b_native.dtype == torch.int32 and b_native.numel() == size_k // 8 * size_n, "w4_native_gemm: b is not a native tensor")
    _req(scales.is_contiguous() and scales.dtype == a.dtype and scales.shape[1] == size_n,
         "w4_native_gemm: scales must be the natural [groups, N] tensor in A's dtype")
    L = _lib.load()
    dev = a.device
    if mode in (2, 3):
        splits = L.nmv_w4_native_gemm_splits(size_m, size_n, size_k, scales.shape[0])
        _req(splits >= 1, "w4_native_gemm: shape not supported")
        out = torch.empty((splits, size_m, size_n), dtype=torch.float32 if mode == 2 else a.dtype, device=dev)
        scratch, nbytes, c = out, out.numel() * out.element_size(), None
    else:
        c = torch.empty((size_m, size_n // 2 if mode == 1 else size_n), dtype=a.dtype, device=dev)
        nbytes = 0 if mode == 1 else int(L.nmv_gptq_marlin_gemm_scratch_bytes(size_m, size_n, size_k, 0))
        scratch = torch.empty((max(nbytes, 16), ), dtype=torch.uint8, device=dev)
        out = c
    with device_guard(a):
        check(L.nmv_w4_native_gemm(ptr(c) if c is not None else None, ptr(a), ptr(b_native), ptr(scales),
                                   ptr(workspace) if workspace is not None else None,
                                   workspace.numel() if workspace is not None else 0, ptr(scratch), nbytes, size_m, size_n,
                                   size_k, scales.shape[0], dtype_code(a.dtype), mode, stream_of(a)))
    return out


def gptq_marlin_gemm_partial_splits(size_m, size_n, size_k, num_groups=None) -> int:
    """slab count of gptq_marlin_gemm_partial (0: unsupported); num_groups = rows of b_scales (default: group 128)"""
    if num_groups is None:
        num_groups = max(size_k // 128, 1)
    return int(_lib.load().nmv_gptq_marlin_gemm_partial_splits(size_m, size_n, size_k, num_groups))


def gptq_marlin_gemm_partial(a, b_q_weight, b_scales, size_m, size_n, size_k) -> torch.Tensor:
    """deferred split-K: fp32 slabs [splits, M, N] for nmv_fused_add_rms_norm_partial to sum
    (include/nmvllm_hip.h: nmv_gptq_marlin_gemm_partial)"""
    _req(a.dim() == 2 and a.shape == (size_m, size_k) and a.is_contiguous(), "gemm_partial: bad a")
    _req(b_q_weight.dtype == torch.int32 and b_q_weight.is_contiguous()
         and b_q_weight.shape == (size_k // 16, size_n * 2), "gemm_partial: b_q_weight must be int32 [K/16, 2N]")
    _req(b_scales.dtype == a.dtype and b_scales.is_contiguous() and b_scales.shape[1] == size_n,
         "gemm_partial: scales must be [groups, N] in A's dtype")
    lib = _lib.load()
    splits = lib.nmv_gptq_marlin_gemm_partial_splits(size_m, size_n, size_k, b_scales.shape[0])
    _req(splits >= 1, "gemm_partial: shape not supported")
    slab = torch.empty((splits, size_m, size_n), dtype=torch.float32, device=a.device)
    with device_guard(a):
        check(lib.nmv_gptq_marlin_gemm_partial(ptr(slab), slab.numel() * 4, ptr(a), ptr(b_q_weight),
                                               ptr(b_scales), size_m, size_n, size_k, b_scales.shape[0],
                                               dtype_code(a.dtype), stream_of(a)))
    return slab


def fused_add_rms_norm_partial(slab, residual, weight, epsilon) -> torch.Tensor:
    """residual += round(sum of the slabs); returns rms_norm(residual) * weight"""
    _req(slab.dim() == 3 and slab.dtype in (torch.float32, residual.dtype) and slab.is_contiguous(),
         "norm_partial: slab [S, T, H] fp32, or in the model dtype")
    _req(residual.is_contiguous() and residual.shape[-1] == slab.shape[2]
         and residual.numel() == slab.shape[1] * slab.shape[2], "norm_partial: residual shape")
    out = torch.empty_like(residual)
    lib = _lib.load()
    fn = lib.nmv_fused_add_rms_norm_partial if slab.dtype == torch.float32 else lib.nmv_fused_add_rms_norm_partial16
    with device_guard(residual):
        check(fn(
            ptr(out), ptr(slab), slab.shape[0], ptr(residual), ptr(weight), epsilon, slab.shape[1],
            slab.shape[2], dtype_code(residual.dtype), stream_of(residual)))
    return out


def rotary_embedding_and_cache_partial(positions, slab, num_heads, num_kv_heads, head_size, cos_sin_cache,
                                       key_cache, value_cache, slot_mapping, kv_cache_dtype, kv_scale,
                                       dtype) -> torch.Tensor:
    """qkv = round(sum of the split-K slabs); neox rope on q / k; k / v -> paged cache (skipped when
    key_cache is None); returns the rounded, rotated qkv row [T, (heads + 2 kv_heads) * head_size]"""
    _req(slab.dim() == 3 and slab.dtype in (torch.float32, dtype) and slab.is_contiguous(),
         "rope_partial: slab [S, T, N] fp32, or in the model dtype")
    s_, t, n = slab.shape
    _req(n == (num_heads + 2 * num_kv_heads) * head_size, "rope_partial: slab width != qkv width")
    _req(positions.dtype == torch.int64 and positions.numel() == t, "positions must be int64 [T]")
    _req(cos_sin_cache.dtype == dtype and cos_sin_cache.shape[1] == head_size and cos_sin_cache.is_contiguous(),
         "rope_partial: cos_sin_cache must be [max_pos, head_size] in the model dtype (rot_dim == head_size)")
    out = torch.empty((t, n), dtype=dtype, device=slab.device)
    block_size = 0
    if key_cache is not None:
        _req(slot_mapping.dtype == torch.int64 and slot_mapping.numel() == t, "slot_mapping must be int64 [T]")
        _req(key_cache.is_contiguous() and value_cache.is_contiguous(), "caches must be contiguous")
        _req(key_cache.shape[1] == num_kv_heads and value_cache.shape[1] == num_kv_heads, "cache kv-head count")
        block_size = key_cache.shape[3]
    lib = _lib.load()
    fn = lib.nmv_rotary_embedding_and_cache_partial if slab.dtype == torch.float32 else lib.nmv_rotary_embedding_and_cache_partial16
    with device_guard(slab):
        check(fn(
            ptr(positions), ptr(slab), s_, ptr(out), t, num_heads, num_kv_heads, head_size,
            ptr(cos_sin_cache), ptr(key_cache), ptr(value_cache), ptr(slot_mapping), block_size,
            dtype_code(dtype), kv_dtype_code(kv_cache_dtype), kv_scale, stream_of(slab)))
    return out


def greedy_sample_shard(logits, index_offset: int) -> torch.Tensor:
    """argmax of one vocab shard [B, V_local] -> record (float32 view): value[p], int32 global index[p]"""
    _req(logits.dim() == 2 and logits.stride(1) == 1, "greedy_sample: logits must be [B, V] with contiguous rows")
    _req(logits.dtype in (torch.float16, torch.bfloat16), "greedy_sample: fp16 / bf16 logits")
    b, v = logits.shape
    lib = _lib.load()
    p_ = lib.nmv_greedy_record_elems(b)
    record = torch.zeros(2 * p_, dtype=torch.float32, device=logits.device)
    sb = lib.nmv_greedy_sample_scratch_bytes(b)
    scratch = torch.empty(max(sb, 1), dtype=torch.uint8, device=logits.device)
    with device_guard(logits):
        check(lib.nmv_greedy_sample_shard(ptr(record), ptr(logits), logits.stride(0), b, v, index_offset,
                                          dtype_code(logits.dtype), ptr(scratch), sb, stream_of(logits)))
    return record


def greedy_sample_finish(gathered, world: int, num_seqs: int, input_ids=None, positions=None, seq_lens=None,
                         slot_mapping=None, block_tables=None, block_size=0) -> torch.Tensor:
    """winner over the `world` shard records (ties -> lowest global index) -> int64 [B]; with the state
    tensors also the on-device advance of the decode batch"""
    lib = _lib.load()
    _req(gathered.dtype == torch.float32 and gathered.is_contiguous()
         and gathered.numel() == world * 2 * lib.nmv_greedy_record_elems(num_seqs), "greedy_sample_finish: record size")
    out = torch.empty((num_seqs, ), dtype=torch.int64, device=gathered.device)
    with device_guard(gathered):
        check(lib.nmv_greedy_sample_finish(
            ptr(out), ptr(gathered), world, num_seqs, ptr(input_ids), ptr(positions), ptr(seq_lens),
            ptr(slot_mapping), ptr(block_tables), block_tables.shape[1] if block_tables is not None else 0,
            block_size, stream_of(gathered)))
    return out


def rms_norm_dynamic_int8_quant(input, residual, weight, epsilon):
    """(fused_add_)rms_norm -> dynamic per-token scaled_int8_quant; returns (int8 [T, H], scales [T, 1]);
    residual (or None) is updated in place as fused_add_rms_norm does, input is left untouched"""
    _req(input.is_contiguous() and (residual is None or residual.is_contiguous()),
         "rms_norm_dynamic_int8_quant: tensors must be contiguous")
    out = torch.empty(input.shape, dtype=torch.int8, device=input.device)
    scales = torch.empty((_rows(input), 1), dtype=torch.float32, device=input.device)
    with device_guard(input):
        check(_lib.load().nmv_rms_norm_dynamic_int8_quant(
            ptr(out), ptr(scales), ptr(input), ptr(residual), ptr(weight), epsilon, _rows(input),
            input.shape[-1], dtype_code(input.dtype), stream_of(input)))
    return out, scales


def silu_and_mul_dynamic_int8_quant(input):
    """silu_and_mul -> dynamic per-token scaled_int8_quant; input [T, 2d] -> (int8 [T, d], scales [T, 1])"""
    _req(input.is_contiguous(), "silu_and_mul_dynamic_int8_quant: input must be contiguous")
    d = input.shape[-1] // 2
    out = torch.empty(input.shape[:-1] + (d, ), dtype=torch.int8, device=input.device)
    scales = torch.empty((_rows(input), 1), dtype=torch.float32, device=input.device)
    with device_guard(input):
        check(_lib.load().nmv_silu_and_mul_dynamic_int8_quant(
            ptr(out), ptr(scales), ptr(input), _rows(input), d, dtype_code(input.dtype),
            stream_of(input)))
    return out, scales


def copy_blocks(key_caches: List[torch.Tensor], value_caches: List[torch.Tensor],
                block_mapping: torch.Tensor) -> None:
    """csrc/cache_kernels.cu:101-148 (the pointer tables are built on the host and copied)"""
    num_layers = len(key_caches)
    _req(num_layers == len(value_caches), "key_caches / value_caches length mismatch")
    if num_layers == 0:
        return
    dev = key_caches[0].device
    _req(dev.type == "cuda", "copy_blocks: caches must be on the GPU")
    kptrs = torch.tensor([t.data_ptr() for t in key_caches], dtype=torch.int64).to(dev)
    vptrs = torch.tensor([t.data_ptr() for t in value_caches], dtype=torch.int64).to(dev)
    _req(block_mapping.dtype == torch.int64, "block_mapping must be int64")
    bm = block_mapping.to(dev).contiguous()
    numel_per_block = key_caches[0][0].numel()
    with device_guard(key_caches[0]):
        check(_lib.load().nmv_copy_blocks(ptr(kptrs), ptr(vptrs), ptr(bm), num_layers,
                                          bm.shape[0], numel_per_block,
                                          key_caches[0].element_size(),
                                          stream_of(key_caches[0])))


def swap_blocks(src: torch.Tensor, dst: torch.Tensor, block_mapping: torch.Tensor) -> None:
    """csrc/cache_kernels.cu:24-63"""
    if src.device.type == "cuda" and dst.device.type == "cuda":
        _req(src.device.index == dst.device.index, "src and dst must be on the same GPU")
        kind = 0
    elif src.device.type == "cuda" and dst.device.type == "cpu":
        kind = 2
    elif src.device.type == "cpu" and dst.device.type == "cuda":
        kind = 1
    else:
        raise _lib.NmvError("Invalid device combination")
    _req(block_mapping.device.type == "cpu", "block_mapping must be on CPU")
    bm = block_mapping.to(torch.int64).contiguous()
    block_bytes = src.element_size() * src[0].numel()
    gpu_t = src if src.device.type == "cuda" else dst
    with device_guard(gpu_t):
        check(_lib.load().nmv_swap_blocks(ptr(src), ptr(dst), ptr(bm), bm.shape[0], block_bytes,
                                          kind, stream_of(gpu_t)))


def convert_fp8(dst_cache, src_cache, scale, kv_cache_dtype) -> None:
    """csrc/cache_kernels.cu:339-389"""
    _req(kv_cache_dtype in ("auto", "fp8", "fp8_e4m3"),
         f"Unsupported data type: {kv_cache_dtype}")
    _req(src_cache.device == dst_cache.device and src_cache.device.type == "cuda",
         "src and dst must be on the same GPU")
    num_blocks = src_cache.shape[0]
    block_stride = src_cache.stride(0)
    if dst_cache.dtype == torch.uint8:
        to_fp8, dt = 1, src_cache.dtype
    else:
        _req(src_cache.dtype == torch.uint8, "one of src/dst must be uint8 (fp8 storage)")
        to_fp8, dt = 0, dst_cache.dtype
    with device_guard(src_cache):
        check(_lib.load().nmv_convert_fp8(ptr(dst_cache), ptr(src_cache), num_blocks,
                                          block_stride, dtype_code(dt), to_fp8, scale,
                                          stream_of(src_cache)))


# ----------------------------------------------------------------------------- cuda utils
def get_device_attribute(attribute: int, device_id: int) -> int:
    v = _lib.load().nmv_get_device_attribute(attribute, device_id)
    check(0 if v >= 0 else -1, "get_device_attribute")
    return v


def get_max_shared_memory_per_block_device_attribute(device_id: int) -> int:
    v = _lib.load().nmv_get_max_shared_memory_per_block_device_attribute(device_id)
    check(0 if v >= 0 else -1, "get_max_shared_memory_per_block_device_attribute")
    return v


# ----------------------------------------------------------------------------- registration
_PA_TAIL = ("Tensor value_cache, int num_kv_heads, float scale, Tensor block_tables, "
            "Tensor seq_lens, int block_size, int max_seq_len, Tensor? alibi_slopes, "
            "str kv_cache_dtype, float kv_scale, int tp_rank, int blocksparse_local_blocks, "
            "int blocksparse_vert_stride, int blocksparse_block_size, "
            "int blocksparse_head_sliding_step) -> ()")

# (schema, impl, dispatch key) -- schema strings follow csrc/torch_bindings.cpp line by line
_C_OPS = [
    ("paged_attention_v1(Tensor! out, Tensor query, Tensor key_cache, " + _PA_TAIL,
     paged_attention_v1),
    ("paged_attention_v2(Tensor! out, Tensor exp_sums, Tensor max_logits, Tensor tmp_out, "
     "Tensor query, Tensor key_cache, " + _PA_TAIL, paged_attention_v2),
    ("silu_and_mul(Tensor! out, Tensor input) -> ()", _act_and_mul(0)),
    ("gelu_and_mul(Tensor! out, Tensor input) -> ()", _act_and_mul(1)),
    ("gelu_tanh_and_mul(Tensor! out, Tensor input) -> ()", _act_and_mul(2)),
    ("gelu_new(Tensor! out, Tensor input) -> ()", _activation(0)),
    ("gelu_fast(Tensor! out, Tensor input) -> ()", _activation(1)),
    ("gelu_quick(Tensor! out, Tensor input) -> ()", _activation(2)),
    ("rms_norm(Tensor! out, Tensor input, Tensor weight, float epsilon) -> ()", rms_norm),
    ("fused_add_rms_norm(Tensor! input, Tensor! residual, Tensor weight, float epsilon) -> ()",
     fused_add_rms_norm),
    ("rotary_embedding(Tensor positions, Tensor! query, Tensor! key, int head_size, "
     "Tensor cos_sin_cache, bool is_neox) -> ()", rotary_embedding),
    ("batched_rotary_embedding(Tensor positions, Tensor! query, Tensor! key, int head_size, "
     "Tensor cos_sin_cache, bool is_neox, int rot_dim, Tensor cos_sin_cache_offsets) -> ()",
     batched_rotary_embedding),
    ("marlin_gemm(Tensor a, Tensor b_q_weight, Tensor b_scales, Tensor workspace, int size_m, "
     "int size_n, int size_k) -> Tensor", marlin_gemm),
    ("fp8_marlin_gemm(Tensor a, Tensor b_q_weight, Tensor b_scales, Tensor workspace, "
     "int num_bits, int size_m, int size_n, int size_k) -> Tensor", fp8_marlin_gemm),
    ("gptq_gemm(Tensor a, Tensor b_q_weight, Tensor b_gptq_qzeros, Tensor b_gptq_scales, "
     "Tensor b_g_idx, bool use_exllama, int bit) -> Tensor", gptq_gemm),
    ("gptq_shuffle(Tensor! q_weight, Tensor q_perm, int bit) -> ()", gptq_shuffle),
    ("awq_gemm(Tensor _in_feats, Tensor _kernel, Tensor _scaling_factors, Tensor _zeros, "
     "int split_k_iters) -> Tensor", awq_gemm),
    ("awq_dequantize(Tensor _kernel, Tensor _scaling_factors, Tensor _zeros, int split_k_iters, "
     "int thx, int thy) -> Tensor", awq_dequantize),
    ("cutlass_scaled_mm(Tensor! out, Tensor a, Tensor b, Tensor a_scales, Tensor b_scales, "
     "Tensor? bias) -> ()", cutlass_scaled_mm),
    ("static_scaled_fp8_quant(Tensor! out, Tensor input, Tensor scale) -> ()", _fp8_quant(0)),
    ("dynamic_scaled_fp8_quant(Tensor! out, Tensor input, Tensor! scale) -> ()", _fp8_quant(1)),
    ("static_scaled_int8_quant(Tensor! out, Tensor input, Tensor scale) -> ()",
     static_scaled_int8_quant),
    ("dynamic_scaled_int8_quant(Tensor! out, Tensor input, Tensor! scale) -> ()",
     dynamic_scaled_int8_quant),
    ("gptq_marlin_repack(Tensor b_q_weight, Tensor perm, int size_k, int size_n, int num_bits) "
     "-> Tensor", gptq_marlin_repack),
    ("gptq_marlin_gemm(Tensor a, Tensor b_q_weight, Tensor b_scales, Tensor g_idx, Tensor perm, "
     "Tensor workspace, int num_bits, int size_m, int size_n, int size_k, bool is_k_full) "
     "-> Tensor", gptq_marlin_gemm),
]

_CACHE_OPS = [
    ("swap_blocks(Tensor src, Tensor! dst, Tensor block_mapping) -> ()", swap_blocks),
    ("copy_blocks(Tensor[]! key_caches, Tensor[]! value_caches, Tensor block_mapping) -> ()",
     copy_blocks),
    ("reshape_and_cache(Tensor key, Tensor value, Tensor! key_cache, Tensor! value_cache, "
     "Tensor slot_mapping, str kv_cache_dtype, float kv_scale) -> ()", reshape_and_cache),
    ("reshape_and_cache_flash(Tensor key, Tensor value, Tensor! key_cache, "
     "Tensor! value_cache, Tensor slot_mapping, str kv_cache_dtype) -> ()",
     reshape_and_cache_flash),
    ("convert_fp8(Tensor! dst_cache, Tensor src_cache, float scale, str kv_cache_dtype) -> ()",
     convert_fp8),
]

# ops without tensor arguments (the reference registers them under kCUDA, torch_bindings.cpp:152-156)
_C_NOTENSOR_OPS = [
    ("cutlass_scaled_mm_supports_fp8(int cuda_device_capability) -> bool",
     cutlass_scaled_mm_supports_fp8),
]

_UTIL_OPS = [
    ("get_device_attribute(int attribute, int device_id) -> int", get_device_attribute),
    ("get_max_shared_memory_per_block_device_attribute(int device_id) -> int",
     get_max_shared_memory_per_block_device_attribute),
]


# ---------------------------------------------------------------------------------------------
# _C_custom_ar: the registered-buffer all-reduce protocol (csrc/custom_all_reduce.cu:12-160; the reference
# compiles it out on ROCm, torch_bindings.cpp:261).  `fa` is the C-side state pointer as an int, IPC handles
# travel as latin-1 strings of nmv_ar_handle_bytes() bytes (the schema's `str[]`).
def _hbytes(h) -> bytes:
    return h if isinstance(h, (bytes, bytearray)) else h.encode("latin-1")


def _raw_handle(b: bytes, hb: int) -> bytes:
    """the runtime's hipIpcMemHandle_t inside what the caller passed: either the bare handle, or torch's
    shareable-handle string (`storage._share_cuda_()`[1] since torch 2.5: a version byte, a type byte -- 'c' for
    a plain device allocation -- and then the handle; c10/cuda/CUDACachingAllocator.cpp shareIpcHandle)"""
    if len(b) == hb:
        return b
    _req(len(b) == hb + 2 and b[1:2] == b"c",
         "IPC handle: expected the runtime's handle or torch's shareable handle of a plain device allocation "
         f"(got {len(b)} bytes; expandable segments cannot be shared this way)")
    return b[2:]


def _handle_block(handles) -> bytes:
    hb = _lib.load().nmv_ar_handle_bytes()
    return b"".join(_raw_handle(_hbytes(h), hb) for h in handles)


def car_init_custom_ar(meta, rank_data, handles, offsets, rank, full_nvlink) -> int:
    import ctypes
    world = len(offsets)
    _req(world <= 8, "world size > 8 is not supported")
    _req(world % 2 == 0, "Odd num gpus is not supported for now")
    _req(world == len(handles), "handles length should equal to offsets length")
    _req(0 <= rank < world, "invalid rank passed in")
    st = ctypes.c_void_p()
    offs = (ctypes.c_int64 * world)(*[int(o) for o in offsets])
    with device_guard(meta):
        check(_lib.load().nmv_car_init(ctypes.byref(st), ptr(meta), ptr(rank_data),
                                       rank_data.numel() * rank_data.element_size(), _handle_block(handles), offs,
                                       world, int(rank), int(bool(full_nvlink))))
    return int(st.value)


def _is_weak_contiguous(t: torch.Tensor) -> bool:
    """custom_all_reduce.cu:36-59"""
    return t.is_contiguous() or (t.untyped_storage().nbytes() - t.storage_offset() * t.element_size()
                                 == t.numel() * t.element_size())


def car_should_custom_ar(inp, max_size, world_size, full_nvlink) -> bool:
    """custom_all_reduce.cu:61-71"""
    inp_size = inp.numel() * inp.element_size()
    if inp_size % 16 != 0 or not _is_weak_contiguous(inp):
        return False
    if world_size == 2 or full_nvlink:
        return inp_size <= max_size
    return False


def car_all_reduce_reg(fa, inp, out) -> None:
    _req(inp.dtype == out.dtype, "all_reduce_reg: inp / out dtypes differ")
    _req(inp.numel() == out.numel(), "all_reduce_reg: inp / out sizes differ")
    _req(_is_weak_contiguous(out), "all_reduce_reg: out must be (weakly) contiguous")
    with device_guard(inp):
        check(_lib.load().nmv_car_all_reduce(fa, ptr(inp), ptr(out), out.numel(), dtype_code(out.dtype),
                                             stream_of(inp)))


def car_all_reduce_unreg(fa, inp, reg_buffer, out) -> None:
    nbytes = inp.numel() * inp.element_size()
    _req(inp.dtype == out.dtype and inp.numel() == out.numel(), "all_reduce_unreg: inp / out mismatch")
    _req(nbytes <= reg_buffer.numel() * reg_buffer.element_size(),
         "registered buffer is too small to contain the input")
    # stream-ordered device copy into the registered buffer (cudaMemcpyAsync, custom_all_reduce.cu:121-123)
    reg_buffer.view(torch.uint8).reshape(-1)[:nbytes].copy_(inp.contiguous().view(torch.uint8).reshape(-1))
    staged = reg_buffer.view(torch.uint8).reshape(-1)[:nbytes].view(inp.dtype)
    car_all_reduce_reg(fa, staged, out)


def car_dispose(fa) -> None:
    check(_lib.load().nmv_car_dispose(fa))


def car_meta_size() -> int:
    return int(_lib.load().nmv_car_meta_size())


def car_register_buffer(fa, t, handles, offsets) -> None:
    import ctypes
    offs = (ctypes.c_int64 * len(offsets))(*[int(o) for o in offsets])
    with device_guard(t):
        check(_lib.load().nmv_car_register_buffer(fa, ptr(t), _handle_block(handles), offs))


def car_get_graph_buffer_ipc_meta(fa):
    import ctypes
    L = _lib.load()
    n = L.nmv_car_graph_buffer_count(fa)
    hb = L.nmv_ar_handle_bytes()
    buf = ctypes.create_string_buffer(max(n * hb, 1))
    offs = (ctypes.c_int64 * max(n, 1))()
    check(L.nmv_car_get_graph_buffer_ipc_meta(fa, buf, offs))
    handles = torch.frombuffer(bytearray(buf.raw[:n * hb]), dtype=torch.uint8).clone() if n else \
        torch.empty(0, dtype=torch.uint8)
    return handles, [int(offs[i]) for i in range(n)]


def car_register_graph_buffers(fa, handles, offsets) -> None:
    import ctypes
    L = _lib.load()
    n = L.nmv_car_graph_buffer_count(fa)
    hb = L.nmv_ar_handle_bytes()
    blobs = [_hbytes(h) for h in handles]
    _req(all(len(b) == n * hb for b in blobs) and all(len(o) == n for o in offsets),
         "register_graph_buffers: every rank must send one handle and one offset per recorded buffer")
    flat = (ctypes.c_int64 * max(len(offsets) * n, 1))(*[int(v) for o in offsets for v in o])
    check(L.nmv_car_register_graph_buffers(fa, b"".join(b[:n * hb] for b in blobs), flat))


_CUSTOM_AR_OPS = [
    ("init_custom_ar(Tensor meta, Tensor rank_data, str[] handles, int[] offsets, int rank, bool full_nvlink) -> int",
     car_init_custom_ar, "CUDA"),
    ("should_custom_ar(Tensor inp, int max_size, int world_size, bool full_nvlink) -> bool", car_should_custom_ar, "CUDA"),
    ("all_reduce_reg(int fa, Tensor inp, Tensor! out) -> ()", car_all_reduce_reg, "CUDA"),
    ("all_reduce_unreg(int fa, Tensor inp, Tensor reg_buffer, Tensor! out) -> ()", car_all_reduce_unreg, "CUDA"),
    ("dispose(int _fa) -> ()", car_dispose, "CompositeExplicitAutograd"),
    ("meta_size() -> int", car_meta_size, "CompositeExplicitAutograd"),
    ("register_buffer(int _fa, Tensor t, str[] handles, int[] offsets) -> ()", car_register_buffer, "CUDA"),
    ("get_graph_buffer_ipc_meta(int _fa) -> (Tensor, int[])", car_get_graph_buffer_ipc_meta, "CompositeExplicitAutograd"),
    ("register_graph_buffers(int _fa, str[] handles, int[][] offsets) -> ()", car_register_graph_buffers,
     "CompositeExplicitAutograd"),
]


def _op_name(schema: str) -> str:
    return schema.split("(", 1)[0]


binding = None   # "cpp" (the _C extension, csrc/torch_bindings.cpp) or "python" (this module) once register() ran


def _load_cpp_extension() -> bool:
    """`import neural_magic_vllm_amd._C`: its static TORCH_LIBRARY blocks register all four namespaces.  Skipped when
    NMV_BINDING=python, when the extension has not been built, or when NMV_HIP_LIB selects a development build of the
    C ABI (the extension is linked against the product libnmvllm_hip.so beside it).  NMV_BINDING=cpp makes its
    absence an error."""
    import importlib
    import os
    want = os.environ.get("NMV_BINDING", "auto")
    if want == "python" or (want == "auto" and os.environ.get("NMV_HIP_LIB")):
        return False
    try:
        ext = importlib.import_module(__package__ + "._C")
    except ImportError as e:
        if want == "cpp":
            raise ImportError(f"NMV_BINDING=cpp but the _C extension does not load ({e}); build it with "
                              "`python neural_magic_vllm_amd/csrc/setup_C.py build_ext --inplace`") from e
        return False
    if ext.abi_version() != _lib.ABI_VERSION:
        raise ImportError(f"_C was linked against C ABI version {ext.abi_version()}, this package needs "
                          f"{_lib.ABI_VERSION}: rebuild libnmvllm_hip.so and the _C extension")
    return True


def register() -> None:
    """Register the four namespaces (idempotent): from the C++ extension when it is there, else from this module."""
    global _registered, binding
    if _registered:
        return
    if _load_cpp_extension():
        binding, _registered = "cpp", True
        return
    for ns, table in (("_C", _C_OPS), ("_C_cache_ops", _CACHE_OPS)):
        lib = torch.library.Library(ns, "DEF")
        for schema, fn in table:
            lib.define(schema)
            name = _op_name(schema)
            lib.impl(name, fn, "CUDA")
        if ns == "_C":
            for schema, fn in _C_NOTENSOR_OPS:
                lib.define(schema)
                lib.impl(_op_name(schema), fn, "CompositeExplicitAutograd")
        _libs.append(lib)
    lib = torch.library.Library("_C_cuda_utils", "DEF")
    for schema, fn in _UTIL_OPS:
        lib.define(schema)
        # no tensor arguments: the reference registers these under kCUDA but they are
        # dispatched without a tensor key, so give them the catch-all key.
        lib.impl(_op_name(schema), fn, "CompositeExplicitAutograd")
    _libs.append(lib)
    lib = torch.library.Library("_C_custom_ar", "DEF")
    for schema, fn, key in _CUSTOM_AR_OPS:
        lib.define(schema)
        lib.impl(_op_name(schema), fn, key)
    _libs.append(lib)
    binding, _registered = "python", True


def all_schemas():
    return {"_C": [s for s, _ in _C_OPS] + [s for s, _ in _C_NOTENSOR_OPS], "_C_cache_ops": [s for s, _ in _CACHE_OPS],
            "_C_cuda_utils": [s for s, _ in _UTIL_OPS], "_C_custom_ar": [s for s, _, _ in _CUSTOM_AR_OPS]}
