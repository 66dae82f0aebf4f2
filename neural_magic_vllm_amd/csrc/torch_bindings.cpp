// `_C`: the TORCH_LIBRARY form of the boundary -- what `import vllm._C` loads in the reference
// (csrc/torch_bindings.cpp:18-294 registers the ops, csrc/registration.h:17-22 exports PyInit).
//
// One translation unit, host code only: every op unwraps its tensors to raw pointers / sizes / strides and
// forwards to the C ABI of libnmvllm_hip.so (include/nmvllm_hip.h) on the current HIP stream of the
// tensor's device.  Namespaces, op names and schema strings are the reference's; the argument checks are
// those of the Python binding (neural_magic_vllm_amd/_torch_bindings.py), which stays as the fallback and
// as the home of the ops the reference does not have (the fused decode-step launches).
// tests/test_op_surface.py compares whatever registered the ops against the reference's schemas.
//
// Dispatch keys: tensor ops get CUDA only (HIP tensors carry the CUDA key on PyTorch-ROCm): CPU tensors
// raise NotImplementedError from the dispatcher -- there is no CPU path.
#include <Python.h>

#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <cstring>
#include <initializer_list>
#include <optional>
#include <string>
#include <vector>

#include "nmvllm_hip.h"

namespace {

using at::Tensor;
using Guard = c10::hip::OptionalHIPGuardMasqueradingAsCUDA;

#define NMV_CALL(expr)                                                    \
  do {                                                                    \
    int nmv_rc_ = (expr);                                                 \
    TORCH_CHECK(nmv_rc_ == 0, nmv_last_error()[0] ? nmv_last_error() : #expr " failed"); \
  } while (0)

inline void* stream_of(const Tensor& t) {
  return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream();
}
inline void* P(const Tensor& t) { return t.data_ptr(); }
inline void* P(const std::optional<Tensor>& t) { return t.has_value() ? t->data_ptr() : nullptr; }

nmv_dtype_t dt(at::ScalarType s) {
  switch (s) {
    case at::kHalf: return NMV_F16;
    case at::kBFloat16: return NMV_BF16;
    case at::kFloat: return NMV_F32;
    default: TORCH_CHECK(false, "unsupported data type ", c10::toString(s));
  }
}
nmv_dtype_t dt(const Tensor& t) { return dt(t.scalar_type()); }

// DISPATCH_BY_KV_CACHE_DTYPE, csrc/quantization/fp8/nvidia/quant_utils.cuh:545-571
nmv_kv_dtype_t kvdt(const std::string& s) {
  if (s == "auto") return NMV_KV_AUTO;
  if (s == "fp8" || s == "fp8_e4m3") return NMV_KV_FP8_E4M3;
  TORCH_CHECK(false, "Unsupported data type of kv cache: ", s);
}

inline int64_t rows_of(const Tensor& t) { return t.numel() ? t.numel() / t.size(-1) : 0; }

// Every tensor of a call on one GPU (raw pointers are about to be handed to a kernel: a host pointer there is a
// GPU fault, not an exception); switches to that device for the duration of the call.
struct OnGpu {
  Guard guard;
  OnGpu(const char* op, std::initializer_list<const Tensor*> ts) {
    const Tensor* first = nullptr;
    for (const Tensor* t : ts) {
      if (t == nullptr || !t->defined() || t->is_meta()) continue;   // meta: the reference's placeholder for "no g_idx"
      TORCH_CHECK(t->is_cuda(), op, ": expected a tensor on the GPU, got one on ", t->device(), " (there is no CPU path)");
      if (first == nullptr) first = t;
      TORCH_CHECK(t->device() == first->device(), op, ": tensors on different devices (", first->device(), " and ",
                  t->device(), ")");
    }
    TORCH_CHECK(first != nullptr, op, ": no tensor argument");
    guard.set_device(first->device());
  }
};
inline const Tensor* opt(const std::optional<Tensor>& t) { return t.has_value() ? &*t : nullptr; }

Tensor scratch_like(const Tensor& like, int64_t nbytes) {
  return at::empty({std::max<int64_t>(nbytes, 16)}, like.options().dtype(at::kByte));
}

// ------------------------------------------------------------------------------------------ attention
void pa_common(const Tensor& query, const Tensor& key_cache, const Tensor& block_tables, const Tensor& seq_lens,
               int64_t vert_stride, int64_t bs_block_size, int64_t block_size) {
  // attention_kernels.cu:235: the sparsity block of a KV block is block_idx * BLOCK_SIZE / blocksparse_block_size
  TORCH_CHECK(vert_stride <= 1 || (bs_block_size > 0 && bs_block_size % block_size == 0),
              "blocksparse_block_size must be a positive multiple of the KV block size");
  TORCH_CHECK(block_tables.scalar_type() == at::kInt && seq_lens.scalar_type() == at::kInt,
              "block_tables and seq_lens must be int32");
  TORCH_CHECK(query.dim() == 3 && query.stride(-1) == 1 && query.stride(1) == query.size(2),
              "query must be [num_seqs, num_heads, head_size] with contiguous heads");
  TORCH_CHECK(key_cache.is_contiguous() || key_cache.stride(-1) == 1, "key_cache layout");
}

// csrc/attention/attention_kernels.cu:805-826
void paged_attention_v1(Tensor& out, const Tensor& query, const Tensor& key_cache, const Tensor& value_cache,
                        int64_t num_kv_heads, double scale, const Tensor& block_tables, const Tensor& seq_lens,
                        int64_t block_size, int64_t max_seq_len, const std::optional<Tensor>& alibi_slopes,
                        std::string kv_cache_dtype, double kv_scale, int64_t tp_rank, int64_t bs_local_blocks,
                        int64_t bs_vert_stride, int64_t bs_block_size, int64_t bs_head_sliding_step) {
  pa_common(query, key_cache, block_tables, seq_lens, bs_vert_stride, bs_block_size, block_size);
  TORCH_CHECK(out.is_contiguous(), "out must be contiguous");
  OnGpu g("paged_attention_v1", {&out, &query, &key_cache, &value_cache, &block_tables, &seq_lens, opt(alibi_slopes)});
  NMV_CALL(nmv_paged_attention_v1(
      P(out), P(query), P(key_cache), P(value_cache), query.size(0), query.size(1), query.size(2), num_kv_heads,
      (float)scale, (const int32_t*)P(block_tables), (const int32_t*)P(seq_lens), block_size, max_seq_len,
      block_tables.size(1), (const float*)P(alibi_slopes), query.stride(0), key_cache.stride(0), key_cache.stride(1),
      dt(query), kvdt(kv_cache_dtype), (float)kv_scale, tp_rank, bs_local_blocks, bs_vert_stride, bs_block_size,
      bs_head_sliding_step, stream_of(query)));
}

// csrc/attention/attention_kernels.cu:966-990
void paged_attention_v2(Tensor& out, const Tensor& exp_sums, const Tensor& max_logits, const Tensor& tmp_out,
                        const Tensor& query, const Tensor& key_cache, const Tensor& value_cache, int64_t num_kv_heads,
                        double scale, const Tensor& block_tables, const Tensor& seq_lens, int64_t block_size,
                        int64_t max_seq_len, const std::optional<Tensor>& alibi_slopes, std::string kv_cache_dtype,
                        double kv_scale, int64_t tp_rank, int64_t bs_local_blocks, int64_t bs_vert_stride,
                        int64_t bs_block_size, int64_t bs_head_sliding_step) {
  pa_common(query, key_cache, block_tables, seq_lens, bs_vert_stride, bs_block_size, block_size);
  TORCH_CHECK(out.is_contiguous() && tmp_out.is_contiguous() && exp_sums.is_contiguous() && max_logits.is_contiguous(),
              "out/tmp_out/exp_sums/max_logits must be contiguous");
  const int64_t max_parts = (max_seq_len + 511) / 512;
  TORCH_CHECK(exp_sums.size(-1) >= max_parts && tmp_out.size(2) >= max_parts,
              "partition buffers too small for max_seq_len");
  TORCH_CHECK(exp_sums.size(-1) == max_parts && tmp_out.size(2) == max_parts,
              "partition buffers must be sized ceil(max_seq_len / 512)");
  OnGpu g("paged_attention_v2", {&out, &exp_sums, &max_logits, &tmp_out, &query, &key_cache, &value_cache, &block_tables, &seq_lens, opt(alibi_slopes)});
  NMV_CALL(nmv_paged_attention_v2(
      P(out), (float*)P(exp_sums), (float*)P(max_logits), P(tmp_out), P(query), P(key_cache), P(value_cache),
      query.size(0), query.size(1), query.size(2), num_kv_heads, (float)scale, (const int32_t*)P(block_tables),
      (const int32_t*)P(seq_lens), block_size, max_seq_len, block_tables.size(1), (const float*)P(alibi_slopes),
      query.stride(0), key_cache.stride(0), key_cache.stride(1), dt(query), kvdt(kv_cache_dtype), (float)kv_scale,
      tp_rank, bs_local_blocks, bs_vert_stride, bs_block_size, bs_head_sliding_step, stream_of(query)));
}

// ------------------------------------------------------------------------------------------ glue ops
// csrc/layernorm_kernels.cu:292-313
void rms_norm(Tensor& out, const Tensor& input, const Tensor& weight, double epsilon) {
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous(), "rms_norm: tensors must be contiguous");
  OnGpu g("rms_norm", {&out, &input, &weight});
  NMV_CALL(nmv_rms_norm(P(out), P(input), P(weight), (float)epsilon, rows_of(input), input.size(-1), dt(input),
                        stream_of(input)));
}

// csrc/layernorm_kernels.cu:315-352
void fused_add_rms_norm(Tensor& input, Tensor& residual, const Tensor& weight, double epsilon) {
  TORCH_CHECK(input.is_contiguous() && residual.is_contiguous(), "fused_add_rms_norm: tensors must be contiguous");
  OnGpu g("fused_add_rms_norm", {&input, &residual, &weight});
  NMV_CALL(nmv_fused_add_rms_norm(P(input), P(residual), P(weight), (float)epsilon, rows_of(input), input.size(-1),
                                  dt(input), stream_of(input)));
}

struct RopeArgs {
  int64_t num_tokens, rot_dim, num_heads, num_kv_heads, q_stride, k_stride;
};
RopeArgs rope_args(const Tensor& positions, const Tensor& query, const Tensor& key, int64_t head_size,
                   const Tensor& cos_sin_cache) {
  RopeArgs r;
  r.num_tokens = query.numel() / query.size(-1);
  TORCH_CHECK(positions.scalar_type() == at::kLong, "positions must be int64");
  TORCH_CHECK(positions.numel() == r.num_tokens, "positions / query token count mismatch");
  r.rot_dim = cos_sin_cache.size(1);
  r.num_heads = query.size(-1) / head_size;
  r.num_kv_heads = key.size(-1) / head_size;
  r.q_stride = query.stride(-2);
  r.k_stride = key.stride(-2);
  return r;
}

// csrc/pos_encoding_kernels.cu:121-160
void rotary_embedding(const Tensor& positions, Tensor& query, Tensor& key, int64_t head_size,
                      const Tensor& cos_sin_cache, bool is_neox) {
  RopeArgs r = rope_args(positions, query, key, head_size, cos_sin_cache);
  TORCH_CHECK(cos_sin_cache.scalar_type() == query.scalar_type(), "cos_sin_cache dtype must match query");
  OnGpu g("rotary_embedding", {&positions, &query, &key, &cos_sin_cache});
  NMV_CALL(nmv_rotary_embedding((const int64_t*)P(positions), P(query), P(key), r.num_tokens, r.num_heads,
                                r.num_kv_heads, head_size, r.rot_dim, r.q_stride, r.k_stride, P(cos_sin_cache),
                                is_neox, dt(query), stream_of(query)));
}

// csrc/pos_encoding_kernels.cu:162-203
void batched_rotary_embedding(const Tensor& positions, Tensor& query, Tensor& key, int64_t head_size,
                              const Tensor& cos_sin_cache, bool is_neox, int64_t rot_dim,
                              const Tensor& cos_sin_cache_offsets) {
  RopeArgs r = rope_args(positions, query, key, head_size, cos_sin_cache);
  TORCH_CHECK(cos_sin_cache_offsets.scalar_type() == at::kLong, "cos_sin_cache_offsets must be int64");
  OnGpu g("batched_rotary_embedding", {&positions, &query, &key, &cos_sin_cache, &cos_sin_cache_offsets});
  NMV_CALL(nmv_batched_rotary_embedding((const int64_t*)P(positions), P(query), P(key), r.num_tokens, r.num_heads,
                                        r.num_kv_heads, head_size, rot_dim, r.q_stride, r.k_stride,
                                        P(cos_sin_cache), is_neox, (const int64_t*)P(cos_sin_cache_offsets),
                                        dt(query), stream_of(query)));
}

// csrc/activation_kernels.cu:63-90 (ACT: 0 silu, 1 gelu erf, 2 gelu tanh)
template <int ACT>
void act_and_mul(Tensor& out, const Tensor& input) {
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous(), "act_and_mul: contiguous tensors");
  OnGpu g("act_and_mul", {&out, &input});
  NMV_CALL(nmv_act_and_mul(P(out), P(input), rows_of(input), input.size(-1) / 2, ACT, dt(input), stream_of(input)));
}

// csrc/activation_kernels.cu:96-162 (ACT: 0 gelu_new, 1 gelu_fast, 2 gelu_quick)
template <int ACT>
void activation(Tensor& out, const Tensor& input) {
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous(), "activation: contiguous tensors");
  OnGpu g("activation", {&out, &input});
  NMV_CALL(nmv_activation(P(out), P(input), rows_of(input), input.size(-1), ACT, dt(input), stream_of(input)));
}

// ------------------------------------------------------------------------------------------ W4A16 / W8A16
// csrc/quantization/gptq_marlin/gptq_marlin_repack.cu:267-348
Tensor gptq_marlin_repack(const Tensor& b_q_weight, const Tensor& perm, int64_t size_k, int64_t size_n,
                          int64_t num_bits) {
  TORCH_CHECK(num_bits == 4 || num_bits == 8, "num_bits must be 4 or 8. Got = ", num_bits);
  const int64_t pack = 32 / num_bits;
  TORCH_CHECK(size_k % 16 == 0, "size_k = ", size_k, " is not divisible by tile_k_size = 16");
  TORCH_CHECK(size_n % 64 == 0, "size_n = ", size_n, " is not divisible by tile_n_size = 64");
  TORCH_CHECK(b_q_weight.dim() == 2 && b_q_weight.size(0) == size_k / pack && b_q_weight.size(1) == size_n,
              "Shape mismatch: b_q_weight ", b_q_weight.sizes(), " vs (", size_k / pack, ", ", size_n, ")");
  TORCH_CHECK(b_q_weight.is_contiguous() && b_q_weight.scalar_type() == at::kInt,
              "b_q_weight must be contiguous int32");
  const bool has_perm = perm.numel() != 0;
  if (has_perm)
    TORCH_CHECK(perm.scalar_type() == at::kInt && perm.numel() == size_k && perm.is_contiguous(),
                "perm must be contiguous int32 [size_k]");
  Tensor out = at::empty({size_k / 16, size_n * 16 / pack}, b_q_weight.options());
  OnGpu g("gptq_marlin_repack", {&b_q_weight, &perm});
  NMV_CALL(nmv_gptq_marlin_repack((const int32_t*)P(b_q_weight), has_perm ? (const int32_t*)P(perm) : nullptr,
                                  (int32_t*)P(out), size_k, size_n, num_bits, stream_of(b_q_weight)));
  return out;
}

// csrc/quantization/gptq_marlin/gptq_marlin.cu:1735-1868 (same argument checks)
Tensor gptq_marlin_gemm(const Tensor& a, const Tensor& b_q_weight, const Tensor& b_scales, const Tensor& g_idx,
                        const Tensor& perm, Tensor& workspace, int64_t num_bits, int64_t size_m, int64_t size_n,
                        int64_t size_k, bool is_k_full) {
  TORCH_CHECK(num_bits == 4 || num_bits == 8, "num_bits must be 4 or 8. Got = ", num_bits);
  const int64_t pack = 32 / num_bits;
  TORCH_CHECK(a.size(0) == size_m, "Shape mismatch: a.size(0) = ", a.size(0), ", size_m = ", size_m);
  TORCH_CHECK(a.size(1) == size_k, "Shape mismatch: a.size(1) = ", a.size(1), ", size_k = ", size_k);
  TORCH_CHECK(size_k % 16 == 0, "size_k = ", size_k, " is not divisible by tile_size = 16");
  TORCH_CHECK(b_q_weight.size(0) == size_k / 16, "Shape mismatch: b_q_weight.size(0) = ", b_q_weight.size(0),
              ", size_k = ", size_k);
  TORCH_CHECK(b_q_weight.size(1) % 16 == 0, "b_q_weight.size(1) is not divisible by tile_size = 16");
  TORCH_CHECK(b_q_weight.size(1) / 16 * pack == size_n, "size_n = ", size_n,
              ", actual_size_n = ", b_q_weight.size(1) / 16 * pack);
  TORCH_CHECK(a.is_contiguous(), "A is not contiguous");
  TORCH_CHECK(b_q_weight.is_contiguous() && b_q_weight.scalar_type() == at::kInt,
              "b_q_weight must be contiguous int32");
  TORCH_CHECK(b_scales.is_contiguous() && b_scales.scalar_type() == a.scalar_type(),
              "b_scales must be contiguous and of A's dtype");
  TORCH_CHECK(a.scalar_type() == at::kHalf || a.scalar_type() == at::kBFloat16,
              "gpt_marlin_gemm only supports bfloat16 and float16");
  TORCH_CHECK(size_n % 64 == 0, "size_n = ", size_n, " is not divisible by min_thread_n = 64");
  TORCH_CHECK(workspace.scalar_type() == at::kInt && workspace.is_contiguous(), "workspace must be int32");
  TORCH_CHECK(workspace.numel() >= size_n / 64 * 16, "workspace.numel = ", workspace.numel(),
              " is below min_workspace_size = ", size_n / 64 * 16);
  const bool has_act_order = g_idx.numel() != 0 && perm.numel() != 0;
  if (has_act_order) {
    TORCH_CHECK(g_idx.numel() == size_k && perm.numel() == size_k, "Unexpected g_idx.size / perm.size");
    TORCH_CHECK(g_idx.scalar_type() == at::kInt && perm.scalar_type() == at::kInt, "g_idx/perm must be int32");
  } else {
    TORCH_CHECK(g_idx.numel() == 0 && perm.numel() == 0, "g_idx and perm must both be empty or both be set");
  }
  const int64_t num_groups = b_scales.size(0);
  TORCH_CHECK(b_scales.size(1) == size_n, "b_scales dim 1 != size_n");
  if (has_act_order) {
    if (is_k_full) {
      TORCH_CHECK(num_groups > 1, "For act_order, num_groups must be > 1");
      TORCH_CHECK(size_k % num_groups == 0, "size_k = ", size_k, ", is not divisible by num_groups");
    }
  } else if (num_groups > 1) {
    TORCH_CHECK(size_k % num_groups == 0, "size_k = ", size_k, ", is not divisible by num_groups");
    const int64_t gs = size_k / num_groups;
    TORCH_CHECK(gs == 32 || gs == 64 || gs == 128, "group_size must be 32, 64 or 128");
  }
  Tensor c = at::empty({size_m, size_n}, a.options());
  if (size_m == 0) return c;
  OnGpu g("gptq_marlin_gemm", {&a, &b_q_weight, &b_scales, &g_idx, &perm, &workspace});
  const int64_t nbytes = nmv_gptq_marlin_gemm_scratch_bytes(size_m, size_n, size_k,
                                                            (int)has_act_order | (num_bits == 8 ? 2 : 0));
  Tensor scratch = scratch_like(a, nbytes);
  NMV_CALL(nmv_gptq_marlin_gemm(P(c), P(a), (const int32_t*)P(b_q_weight), P(b_scales),
                                has_act_order ? (const int32_t*)P(g_idx) : nullptr,
                                has_act_order ? (const int32_t*)P(perm) : nullptr, (int32_t*)P(workspace),
                                workspace.numel(), P(scratch), scratch.numel(), num_bits, size_m, size_n, size_k,
                                num_groups, is_k_full, dt(a), stream_of(a)));
  return c;
}

// csrc/quantization/marlin/dense/marlin_cuda_kernel.cu:1045-1136 (legacy Marlin, 4-bit)
Tensor marlin_gemm(Tensor& a, Tensor& b_q_weight, Tensor& b_scales, Tensor& workspace, int64_t size_m,
                   int64_t size_n, int64_t size_k) {
  TORCH_CHECK(a.size(0) == size_m && a.size(1) == size_k, "Shape mismatch: a vs size_m / size_k");
  TORCH_CHECK(size_k % 16 == 0 && b_q_weight.size(0) == size_k / 16,
              "Shape mismatch: b_q_weight.size(0) = ", b_q_weight.size(0), ", size_k = ", size_k);
  TORCH_CHECK(b_q_weight.size(1) / 16 * 8 == size_n, "size_n does not match b_q_weight");
  TORCH_CHECK(a.is_contiguous() && b_q_weight.is_contiguous() && b_scales.is_contiguous(),
              "a, b_q_weight and b_scales must be contiguous");
  TORCH_CHECK((a.scalar_type() == at::kHalf || a.scalar_type() == at::kBFloat16) &&
                  b_scales.scalar_type() == a.scalar_type(),
              "marlin_gemm supports float16 (and bfloat16) activations with scales of the same dtype");
  TORCH_CHECK(workspace.numel() >= size_n / 128 * 16 || workspace.numel() >= size_n / 64, "workspace is too small");
  Tensor c = at::empty({size_m, size_n}, a.options());
  if (size_m == 0) return c;
  OnGpu g("marlin_gemm", {&a, &b_q_weight, &b_scales, &workspace});
  Tensor scratch = scratch_like(a, nmv_gptq_marlin_gemm_scratch_bytes(size_m, size_n, size_k, 0));
  NMV_CALL(nmv_marlin_gemm(P(c), P(a), (const int32_t*)P(b_q_weight), P(b_scales), (int32_t*)P(workspace),
                           workspace.numel(), P(scratch), scratch.numel(), size_m, size_n, size_k, b_scales.size(0),
                           dt(a), stream_of(a)));
  return c;
}

// csrc/quantization/fp8/fp8_marlin.cu:1212-1308
Tensor fp8_marlin_gemm(Tensor& a, Tensor& b_q_weight, Tensor& b_scales, Tensor& workspace, int64_t num_bits,
                       int64_t size_m, int64_t size_n, int64_t size_k) {
  TORCH_CHECK(a.size(0) == size_m && a.size(1) == size_k, "Shape mismatch: a vs size_m / size_k");
  TORCH_CHECK(b_q_weight.size(0) == size_k / 16 && b_q_weight.size(1) / 16 * 4 == size_n,
              "Shape mismatch: b_q_weight vs size_k / size_n");
  TORCH_CHECK(a.is_contiguous() && b_q_weight.is_contiguous() && b_scales.is_contiguous(),
              "a, b_q_weight and b_scales must be contiguous");
  TORCH_CHECK(b_scales.scalar_type() == a.scalar_type() && b_scales.size(-1) == size_n,
              "b_scales must be [G, size_n] of A's dtype");
  Tensor c = at::empty({size_m, size_n}, a.options());
  if (size_m == 0) return c;
  OnGpu g("fp8_marlin_gemm", {&a, &b_q_weight, &b_scales, &workspace});
  Tensor scratch = scratch_like(a, nmv_fp8_marlin_gemm_scratch_bytes(size_m, size_n, size_k));
  NMV_CALL(nmv_fp8_marlin_gemm(P(c), P(a), (const int32_t*)P(b_q_weight), P(b_scales), (int32_t*)P(workspace),
                               workspace.numel(), P(scratch), scratch.numel(), num_bits, size_m, size_n, size_k,
                               b_scales.size(0), dt(a), stream_of(a)));
  return c;
}

// ------------------------------------------------------------------------------------------ GPTQ / AWQ
inline bool usable_index(const Tensor& t) { return t.defined() && t.numel() > 0 && !t.is_meta(); }

// csrc/quantization/gptq/q_gemm.cu:1823-1846
Tensor gptq_gemm(Tensor a, Tensor b_q_weight, Tensor b_gptq_qzeros, Tensor b_gptq_scales, Tensor b_g_idx,
                 bool use_exllama, int64_t bit) {
  TORCH_CHECK(bit == 2 || bit == 3 || bit == 4 || bit == 8, "unsupported bit width ", bit);
  const int64_t size_m = a.size(0), size_k = a.size(1), size_n = b_q_weight.size(1);
  // 3-bit: 32 codes per three int32 rows (qweight is [K * 3 / 32, N])
  TORCH_CHECK(b_q_weight.size(0) * 32 == size_k * bit, "b_q_weight rows do not match a's K");
  TORCH_CHECK(a.is_contiguous() && b_q_weight.is_contiguous() && b_gptq_scales.is_contiguous() &&
                  b_gptq_qzeros.is_contiguous(),
              "gptq_gemm: tensors must be contiguous");
  TORCH_CHECK(b_gptq_scales.scalar_type() == a.scalar_type(), "scales must have the activation dtype");
  const bool has_idx = usable_index(b_g_idx);
  if (has_idx)
    TORCH_CHECK(b_g_idx.numel() == size_k && b_g_idx.scalar_type() == at::kInt, "g_idx must be int32 [K]");
  Tensor c = at::empty({size_m, size_n}, a.options());
  OnGpu g("gptq_gemm", {&a, &b_q_weight, &b_gptq_qzeros, &b_gptq_scales, &b_g_idx});
  const int64_t nbytes = nmv_wq_gemm_scratch_bytes(size_m, size_n, size_k);
  Tensor scratch;
  if (nbytes) scratch = at::empty({nbytes}, a.options().dtype(at::kByte));
  NMV_CALL(nmv_gptq_gemm(P(c), P(a), (const int32_t*)P(b_q_weight), (const int32_t*)P(b_gptq_qzeros),
                         P(b_gptq_scales), has_idx ? (const int32_t*)P(b_g_idx) : nullptr, use_exllama, bit, size_m,
                         size_n, size_k, b_gptq_scales.size(0), dt(a), nbytes ? P(scratch) : nullptr, nbytes,
                         stream_of(a)));
  return c;
}

// csrc/quantization/gptq/q_gemm.cu:1848-1856 (in place)
void gptq_shuffle(Tensor q_weight, Tensor q_perm, int64_t bit) {
  TORCH_CHECK(bit == 2 || bit == 3 || bit == 4 || bit == 8, "gptq_shuffle: ", bit, "-bit weights are not supported");
  if (!usable_index(q_perm)) return;
  const int64_t size_k = q_weight.size(0) * 32 / bit;
  TORCH_CHECK(q_perm.scalar_type() == at::kInt && q_perm.numel() == size_k, "q_perm must be int32 [K]");
  Tensor tmp = at::empty_like(q_weight);
  OnGpu g("gptq_shuffle", {&q_weight, &q_perm});
  NMV_CALL(nmv_gptq_shuffle((int32_t*)P(q_weight), (const int32_t*)P(q_perm), (int32_t*)P(tmp), size_k,
                            q_weight.size(1), bit, stream_of(q_weight)));
}

// csrc/quantization/awq/gemm_kernels.cu:492-549; argument order of csrc/ops.h:66-68
Tensor awq_gemm(Tensor _in_feats, Tensor _kernel, Tensor _scaling_factors, Tensor _zeros, int64_t split_k_iters) {
  const int64_t size_m = _in_feats.size(0), size_k = _in_feats.size(1), size_n = _kernel.size(1) * 8;
  TORCH_CHECK(_kernel.size(0) == size_k, "awq_gemm: qweight rows must equal K");
  TORCH_CHECK(_scaling_factors.size(1) == size_n && _zeros.size(1) * 8 == size_n, "awq_gemm: scales/zeros shape");
  TORCH_CHECK(_in_feats.is_contiguous() && _kernel.is_contiguous() && _scaling_factors.is_contiguous() &&
                  _zeros.is_contiguous(),
              "awq_gemm: tensors must be contiguous");
  TORCH_CHECK(_scaling_factors.scalar_type() == _in_feats.scalar_type(), "scales must have the activation dtype");
  Tensor c = at::empty({size_m, size_n}, _in_feats.options());
  OnGpu g("awq_gemm", {&_in_feats, &_kernel, &_scaling_factors, &_zeros});
  const int64_t nbytes = nmv_wq_gemm_scratch_bytes(size_m, size_n, size_k);
  Tensor scratch;
  if (nbytes) scratch = at::empty({nbytes}, _in_feats.options().dtype(at::kByte));
  NMV_CALL(nmv_awq_gemm(P(c), P(_in_feats), (const int32_t*)P(_kernel), P(_scaling_factors),
                        (const int32_t*)P(_zeros), size_m, size_n, size_k, _scaling_factors.size(0), dt(_in_feats),
                        nbytes ? P(scratch) : nullptr, nbytes, stream_of(_in_feats)));
  return c;
}

// csrc/quantization/awq/gemm_kernels.cu:436-490 -> [K, N]
Tensor awq_dequantize(Tensor _kernel, Tensor _scaling_factors, Tensor _zeros, int64_t split_k_iters, int64_t thx,
                      int64_t thy) {
  const int64_t size_k = _kernel.size(0), size_n = _kernel.size(1) * 8;
  Tensor out = at::empty({size_k, size_n}, _scaling_factors.options());
  OnGpu g("awq_dequantize", {&_kernel, &_scaling_factors, &_zeros});
  NMV_CALL(nmv_awq_dequantize(P(out), (const int32_t*)P(_kernel), P(_scaling_factors), (const int32_t*)P(_zeros),
                              size_n, size_k, _scaling_factors.size(0), dt(_scaling_factors), stream_of(_kernel)));
  return out;
}

// ------------------------------------------------------------------------------------------ W8A8
// csrc/quantization/compressed_tensors/int8_quant_kernels.cu:77-95
void static_scaled_int8_quant(Tensor& out, const Tensor& input, const Tensor& scale) {
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous(), "input/out must be contiguous");
  TORCH_CHECK(scale.numel() == 1 && scale.scalar_type() == at::kFloat, "scale must be one float32");
  TORCH_CHECK(out.scalar_type() == at::kChar, "out must be int8");
  OnGpu g("static_scaled_int8_quant", {&out, &input, &scale});
  NMV_CALL(nmv_scaled_int8_quant(P(out), P(input), (float*)P(scale), rows_of(input), input.size(-1), 0, dt(input),
                                 stream_of(input)));
}

// int8_quant_kernels.cu:97-115: per-token scales
void dynamic_scaled_int8_quant(Tensor& out, const Tensor& input, Tensor& scale) {
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous(), "input/out must be contiguous");
  TORCH_CHECK(scale.scalar_type() == at::kFloat && scale.is_contiguous() && scale.numel() >= rows_of(input),
              "scales must be float32 [num_tokens, 1]");
  TORCH_CHECK(out.scalar_type() == at::kChar, "out must be int8");
  OnGpu g("dynamic_scaled_int8_quant", {&out, &input, &scale});
  NMV_CALL(nmv_scaled_int8_quant(P(out), P(input), (float*)P(scale), rows_of(input), input.size(-1), 1, dt(input),
                                 stream_of(input)));
}

// csrc/quantization/fp8/common.cu:129-165
template <int DYNAMIC>
void scaled_fp8_quant(Tensor& out, const Tensor& input, const Tensor& scale) {
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous(), "input/out must be contiguous");
  TORCH_CHECK(out.scalar_type() == at::kFloat8_e4m3fn || out.scalar_type() == at::kByte,
              "out must be float8_e4m3fn");
  TORCH_CHECK(scale.numel() == 1 && scale.scalar_type() == at::kFloat, "scale must be one float32");
  TORCH_CHECK(out.numel() >= input.numel(), "out is smaller than input");
  OnGpu g("scaled_fp8_quant", {&out, &input, &scale});
  NMV_CALL(nmv_scaled_fp8_quant(P(out), P(input), (float*)P(scale), input.numel(), DYNAMIC, dt(input),
                                stream_of(input)));
}

bool cutlass_scaled_mm_supports_fp8(int64_t cuda_device_capability) {
  return nmv_cutlass_scaled_mm_supports_fp8(cuda_device_capability) != 0;
}

// csrc/quantization/cutlass_w8a8/scaled_mm_entry.cu:48-100 (same checks)
void cutlass_scaled_mm(Tensor& out, const Tensor& a, const Tensor& b, const Tensor& a_scales, const Tensor& b_scales,
                       const std::optional<Tensor>& bias) {
  TORCH_CHECK(a.dim() == 2 && b.dim() == 2 && out.dim() == 2, "a, b, c must be 2-D");
  TORCH_CHECK(out.size(0) == a.size(0) && a.size(1) == b.size(0) && b.size(1) == out.size(1),
              "shape mismatch between a, b and c");
  TORCH_CHECK(a_scales.numel() == 1 || a_scales.numel() == a.size(0), "a_scales must be scalar or [M]");
  TORCH_CHECK(b_scales.numel() == 1 || b_scales.numel() == b.size(1), "b_scales must be scalar or [N]");
  TORCH_CHECK(a.stride(1) == 1 && out.stride(1) == 1, "a and c must be row major");
  TORCH_CHECK(b.stride(0) == 1, "b must be column major");
  TORCH_CHECK(out.stride(0) % 16 == 0 && b.stride(1) % 16 == 0, "c.stride(0) and b.stride(1) must be 16B aligned");
  TORCH_CHECK(a_scales.is_contiguous() && b_scales.is_contiguous(), "scales must be contiguous");
  TORCH_CHECK(a_scales.scalar_type() == at::kFloat && b_scales.scalar_type() == at::kFloat, "scales must be float32");
  if (bias.has_value()) {
    TORCH_CHECK(bias->numel() == b.size(1) && bias->is_contiguous() && bias->dim() == 1,
                "bias must be a contiguous [N] vector");
    TORCH_CHECK(bias->scalar_type() == out.scalar_type(), "bias dtype must match the output");
  }
  nmv_q8_dtype_t q;
  if (a.scalar_type() == at::kChar) {
    TORCH_CHECK(b.scalar_type() == at::kChar, "a and b must both be int8");
    q = NMV_I8;
  } else {
    TORCH_CHECK(a.scalar_type() == at::kFloat8_e4m3fn && b.scalar_type() == at::kFloat8_e4m3fn,
                "a and b must both be int8 or both float8_e4m3fn");
    q = NMV_FP8_E4M3;
  }
  const int64_t m = a.size(0), k = a.size(1), n = b.size(1);
  OnGpu g("cutlass_scaled_mm", {&out, &a, &b, &a_scales, &b_scales, opt(bias)});
  const int64_t sb = nmv_scaled_mm_scratch_bytes(m, n, k);
  Tensor scratch;
  if (sb) scratch = at::empty({sb}, a.options().dtype(at::kByte));
  NMV_CALL(nmv_scaled_mm(P(out), P(a), P(b), (const float*)P(a_scales), (const float*)P(b_scales), P(bias), m, n, k,
                         a.stride(0), b.stride(1), out.stride(0), a_scales.numel(), b_scales.numel(), q, dt(out),
                         sb ? P(scratch) : nullptr, sb, stream_of(a)));
}

// ------------------------------------------------------------------------------------------ cache ops
// csrc/cache_kernels.cu:253-278
void reshape_and_cache(Tensor& key, Tensor& value, Tensor& key_cache, Tensor& value_cache, Tensor& slot_mapping,
                       const std::string& kv_cache_dtype, const double kv_scale) {
  const int64_t num_heads = key.size(1), head_size = key.size(2), block_size = key_cache.size(3);
  TORCH_CHECK(slot_mapping.scalar_type() == at::kLong, "slot_mapping must be int64");
  TORCH_CHECK(key.stride(-1) == 1 && key.stride(1) == head_size && value.stride(-1) == 1 &&
                  value.stride(1) == head_size,
              "key/value heads must be contiguous");
  TORCH_CHECK(key_cache.is_contiguous() && value_cache.is_contiguous(), "caches must be contiguous");
  OnGpu g("reshape_and_cache", {&key, &value, &key_cache, &value_cache, &slot_mapping});
  NMV_CALL(nmv_reshape_and_cache(P(key), P(value), P(key_cache), P(value_cache), (const int64_t*)P(slot_mapping),
                                 slot_mapping.numel(), num_heads, head_size, block_size, key.stride(0),
                                 value.stride(0), dt(key), kvdt(kv_cache_dtype), (float)kv_scale, stream_of(key)));
}

// csrc/cache_kernels.cu:280-316
void reshape_and_cache_flash(Tensor& key, Tensor& value, Tensor& key_cache, Tensor& value_cache,
                             Tensor& slot_mapping, const std::string& kv_cache_dtype) {
  TORCH_CHECK(kv_cache_dtype == "auto", "FlashAttention does not support FP8 kv-cache");
  const int64_t num_heads = key.size(1), head_size = key.size(2), block_size = key_cache.size(1);
  TORCH_CHECK(key_cache.stride(0) == value_cache.stride(0), "k/v cache block strides differ");
  OnGpu g("reshape_and_cache_flash", {&key, &value, &key_cache, &value_cache, &slot_mapping});
  NMV_CALL(nmv_reshape_and_cache_flash(P(key), P(value), P(key_cache), P(value_cache),
                                       (const int64_t*)P(slot_mapping), slot_mapping.numel(), num_heads, head_size,
                                       block_size, key.stride(0), value.stride(0), key_cache.stride(0), dt(key),
                                       stream_of(key)));
}

// csrc/cache_kernels.cu:101-148 (the pointer tables are built on the host and copied)
void copy_blocks(std::vector<Tensor> const& key_caches, std::vector<Tensor> const& value_caches,
                 const Tensor& block_mapping) {
  const int64_t num_layers = key_caches.size();
  TORCH_CHECK(num_layers == (int64_t)value_caches.size(), "key_caches / value_caches length mismatch");
  if (num_layers == 0) return;
  const at::Device dev = key_caches[0].device();
  TORCH_CHECK(dev.is_cuda(), "copy_blocks: caches must be on the GPU");
  TORCH_CHECK(block_mapping.scalar_type() == at::kLong, "block_mapping must be int64");
  Tensor host = at::empty({2, num_layers}, at::TensorOptions().dtype(at::kLong));
  int64_t* hp = host.data_ptr<int64_t>();
  for (int64_t i = 0; i < num_layers; ++i) {
    hp[i] = (int64_t)key_caches[i].data_ptr();
    hp[num_layers + i] = (int64_t)value_caches[i].data_ptr();
  }
  Guard g(dev);
  Tensor ptrs = host.to(dev);
  Tensor bm = block_mapping.to(dev).contiguous();
  NMV_CALL(nmv_copy_blocks((void* const*)ptrs.data_ptr<int64_t>(),
                           (void* const*)(ptrs.data_ptr<int64_t>() + num_layers), (const int64_t*)P(bm), num_layers,
                           bm.size(0), key_caches[0][0].numel(), key_caches[0].element_size(),
                           stream_of(key_caches[0])));
}

// csrc/cache_kernels.cu:24-63
void swap_blocks(Tensor& src, Tensor& dst, const Tensor& block_mapping) {
  int kind;
  if (src.is_cuda() && dst.is_cuda()) {
    TORCH_CHECK(src.get_device() == dst.get_device(), "src and dst must be on the same GPU");
    kind = 0;
  } else if (src.is_cuda() && dst.is_cpu()) {
    kind = 2;
  } else if (src.is_cpu() && dst.is_cuda()) {
    kind = 1;
  } else {
    TORCH_CHECK(false, "Invalid device combination");
  }
  TORCH_CHECK(block_mapping.is_cpu(), "block_mapping must be on CPU");
  Tensor bm = block_mapping.to(at::kLong).contiguous();
  const int64_t block_bytes = src.element_size() * src[0].numel();
  const Tensor& gpu_t = src.is_cuda() ? src : dst;
  Guard g(gpu_t.device());
  NMV_CALL(nmv_swap_blocks(P(src), P(dst), (const int64_t*)P(bm), bm.size(0), block_bytes, kind, stream_of(gpu_t)));
}

// csrc/cache_kernels.cu:339-389
void convert_fp8(Tensor& dst_cache, Tensor& src_cache, const double scale, const std::string& kv_cache_dtype) {
  TORCH_CHECK(kv_cache_dtype == "auto" || kv_cache_dtype == "fp8" || kv_cache_dtype == "fp8_e4m3",
              "Unsupported data type: ", kv_cache_dtype);
  TORCH_CHECK(src_cache.device() == dst_cache.device() && src_cache.is_cuda(), "src and dst must be on the same GPU");
  int to_fp8;
  at::ScalarType t;
  if (dst_cache.scalar_type() == at::kByte) {
    to_fp8 = 1, t = src_cache.scalar_type();
  } else {
    TORCH_CHECK(src_cache.scalar_type() == at::kByte, "one of src/dst must be uint8 (fp8 storage)");
    to_fp8 = 0, t = dst_cache.scalar_type();
  }
  OnGpu g("convert_fp8", {&dst_cache, &src_cache});
  NMV_CALL(nmv_convert_fp8(P(dst_cache), P(src_cache), src_cache.size(0), src_cache.stride(0), dt(t), to_fp8,
                           (float)scale, stream_of(src_cache)));
}

// ------------------------------------------------------------------------------------------ cuda utils
int64_t get_device_attribute(int64_t attribute, int64_t device_id) {
  const int64_t v = nmv_get_device_attribute(attribute, device_id);
  TORCH_CHECK(v >= 0, nmv_last_error()[0] ? nmv_last_error() : "get_device_attribute failed");
  return v;
}
int64_t get_max_shared_memory_per_block_device_attribute(int64_t device_id) {
  const int64_t v = nmv_get_max_shared_memory_per_block_device_attribute(device_id);
  TORCH_CHECK(v >= 0, nmv_last_error()[0] ? nmv_last_error()
                                          : "get_max_shared_memory_per_block_device_attribute failed");
  return v;
}

// ------------------------------------------------------------------------------------------ _C_custom_ar
// The registered-buffer all-reduce protocol (csrc/custom_all_reduce.cu:12-160; the reference compiles it out on
// ROCm, torch_bindings.cpp:261).  `fa` is the C-side state pointer as an int; IPC handles travel as strings of
// nmv_ar_handle_bytes() bytes, or as torch's shareable-handle string (`storage._share_cuda_()`[1] since torch 2.5:
// a version byte, a type byte -- 'c' for a plain device allocation -- and then the runtime's handle).
using fptr_t = int64_t;

// A handle reaches a `str` argument either as Python bytes (raw) or as a Python str holding the bytes as latin-1
// code points, which the argument parser hands over UTF-8 encoded: undo that when the raw length does not fit.
std::string raw_bytes(const std::string& h, size_t expected, size_t expected_alt) {
  if (h.size() == expected || h.size() == expected_alt) return h;
  std::string out;
  out.reserve(h.size());
  for (size_t i = 0; i < h.size(); ++i) {
    const unsigned char c = h[i];
    if (c < 0x80) {
      out.push_back((char)c);
    } else {
      TORCH_CHECK((c == 0xC2 || c == 0xC3) && i + 1 < h.size() && ((unsigned char)h[i + 1] & 0xC0) == 0x80,
                  "IPC handle: neither raw bytes nor a latin-1 string");
      out.push_back((char)(((c & 0x03) << 6) | ((unsigned char)h[++i] & 0x3F)));
    }
  }
  return out;
}

std::string handle_block(const std::vector<std::string>& handles) {
  const size_t hb = nmv_ar_handle_bytes();
  std::string out;
  out.reserve(handles.size() * hb);
  for (const std::string& given : handles) {
    const std::string h = raw_bytes(given, hb, hb + 2);
    if (h.size() == hb) {
      out += h;
    } else {
      TORCH_CHECK(h.size() == hb + 2 && h[1] == 'c',
                  "IPC handle: expected the runtime's handle or torch's shareable handle of a plain device "
                  "allocation (got ", h.size(), " bytes; expandable segments cannot be shared this way)");
      out.append(h, 2, hb);
    }
  }
  return out;
}

fptr_t init_custom_ar(Tensor& meta, Tensor& rank_data, const std::vector<std::string>& handles,
                      const std::vector<int64_t>& offsets, int64_t rank, bool full_nvlink) {
  const int64_t world = offsets.size();
  TORCH_CHECK(world <= 8, "world size > 8 is not supported");
  TORCH_CHECK(world % 2 == 0, "Odd num gpus is not supported for now");
  TORCH_CHECK(world == (int64_t)handles.size(), "handles length should equal to offsets length");
  TORCH_CHECK(rank >= 0 && rank < world, "invalid rank passed in");
  void* st = nullptr;
  const std::string block = handle_block(handles);
  OnGpu g("init_custom_ar", {&meta, &rank_data});
  NMV_CALL(nmv_car_init(&st, P(meta), P(rank_data), rank_data.numel() * rank_data.element_size(), block.data(),
                        offsets.data(), world, rank, full_nvlink));
  return (fptr_t)st;
}

// custom_all_reduce.cu:36-59
bool is_weak_contiguous(const Tensor& t) {
  return t.is_contiguous() || (t.storage().nbytes() - t.storage_offset() * t.element_size() ==
                               (size_t)(t.numel() * t.element_size()));
}

// custom_all_reduce.cu:61-71
bool should_custom_ar(Tensor& inp, int64_t max_size, int64_t world_size, bool full_nvlink) {
  const int64_t inp_size = inp.numel() * inp.element_size();
  if (inp_size % 16 != 0 || !is_weak_contiguous(inp)) return false;
  if (world_size == 2 || full_nvlink) return inp_size <= max_size;
  return false;
}

void all_reduce_reg(fptr_t fa, Tensor& inp, Tensor& out) {
  TORCH_CHECK(inp.scalar_type() == out.scalar_type(), "all_reduce_reg: inp / out dtypes differ");
  TORCH_CHECK(inp.numel() == out.numel(), "all_reduce_reg: inp / out sizes differ");
  TORCH_CHECK(is_weak_contiguous(out), "all_reduce_reg: out must be (weakly) contiguous");
  OnGpu g("all_reduce_reg", {&inp, &out});
  NMV_CALL(nmv_car_all_reduce((void*)fa, P(inp), P(out), out.numel(), dt(out), stream_of(inp)));
}

void all_reduce_unreg(fptr_t fa, Tensor& inp, Tensor& reg_buffer, Tensor& out) {
  const int64_t nbytes = inp.numel() * inp.element_size();
  TORCH_CHECK(inp.scalar_type() == out.scalar_type() && inp.numel() == out.numel(),
              "all_reduce_unreg: inp / out mismatch");
  TORCH_CHECK(nbytes <= reg_buffer.numel() * reg_buffer.element_size(),
              "registered buffer is too small to contain the input");
  // stream-ordered device copy into the registered buffer (cudaMemcpyAsync, custom_all_reduce.cu:121-123)
  Tensor staged_bytes = reg_buffer.view(at::kByte).reshape({-1}).narrow(0, 0, nbytes);
  staged_bytes.copy_(inp.contiguous().view(at::kByte).reshape({-1}));
  Tensor staged = staged_bytes.view(inp.scalar_type());
  all_reduce_reg(fa, staged, out);
}

void dispose(fptr_t _fa) { NMV_CALL(nmv_car_dispose((void*)_fa)); }
int64_t meta_size() { return nmv_car_meta_size(); }

void register_buffer(fptr_t _fa, Tensor& t, const std::vector<std::string>& handles,
                     const std::vector<int64_t>& offsets) {
  const std::string block = handle_block(handles);
  OnGpu g("register_buffer", {&t});
  NMV_CALL(nmv_car_register_buffer((void*)_fa, P(t), block.data(), offsets.data()));
}

std::tuple<Tensor, std::vector<int64_t>> get_graph_buffer_ipc_meta(fptr_t _fa) {
  const int64_t n = nmv_car_graph_buffer_count((void*)_fa);
  const int64_t hb = nmv_ar_handle_bytes();
  Tensor handles = at::empty({n * hb}, at::TensorOptions().dtype(at::kByte));
  std::vector<int64_t> offsets(std::max<int64_t>(n, 1));
  std::vector<char> raw(std::max<int64_t>(n * hb, 1));
  NMV_CALL(nmv_car_get_graph_buffer_ipc_meta((void*)_fa, raw.data(), offsets.data()));
  if (n) std::memcpy(handles.data_ptr(), raw.data(), n * hb);
  offsets.resize(n);
  return {handles, offsets};
}

void register_graph_buffers(fptr_t _fa, const std::vector<std::string>& handles,
                            const std::vector<std::vector<int64_t>>& offsets) {
  const size_t n = nmv_car_graph_buffer_count((void*)_fa);
  const size_t hb = nmv_ar_handle_bytes();
  std::string block;
  for (const std::string& given : handles) {
    const std::string h = raw_bytes(given, n * hb, n * hb);
    TORCH_CHECK(h.size() == n * hb,
                "register_graph_buffers: every rank must send one handle and one offset per recorded buffer");
    block += h;
  }
  std::vector<int64_t> flat;
  for (const auto& o : offsets) {
    TORCH_CHECK(o.size() == n,
                "register_graph_buffers: every rank must send one handle and one offset per recorded buffer");
    flat.insert(flat.end(), o.begin(), o.end());
  }
  if (flat.empty()) flat.push_back(0);
  NMV_CALL(nmv_car_register_graph_buffers((void*)_fa, block.data(), flat.data()));
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// Registration: the schema strings of csrc/torch_bindings.cpp, line by line (ops whose schema the reference
// lets torch infer from the C++ signature are written out -- tests/test_op_surface.py holds both forms).
#define PA_TAIL                                                                                              \
  "Tensor value_cache, int num_kv_heads, float scale, Tensor block_tables, Tensor seq_lens, int block_size, " \
  "int max_seq_len, Tensor? alibi_slopes, str kv_cache_dtype, float kv_scale, int tp_rank, "                 \
  "int blocksparse_local_blocks, int blocksparse_vert_stride, int blocksparse_block_size, "                  \
  "int blocksparse_head_sliding_step) -> ()"

TORCH_LIBRARY(_C, ops) {
  ops.def("paged_attention_v1(Tensor! out, Tensor query, Tensor key_cache, " PA_TAIL);
  ops.impl("paged_attention_v1", c10::kCUDA, &paged_attention_v1);
  ops.def("paged_attention_v2(Tensor! out, Tensor exp_sums, Tensor max_logits, Tensor tmp_out, Tensor query, "
          "Tensor key_cache, " PA_TAIL);
  ops.impl("paged_attention_v2", c10::kCUDA, &paged_attention_v2);

  ops.def("silu_and_mul(Tensor! out, Tensor input) -> ()");
  ops.impl("silu_and_mul", c10::kCUDA, &act_and_mul<0>);
  ops.def("gelu_and_mul(Tensor! out, Tensor input) -> ()");
  ops.impl("gelu_and_mul", c10::kCUDA, &act_and_mul<1>);
  ops.def("gelu_tanh_and_mul(Tensor! out, Tensor input) -> ()");
  ops.impl("gelu_tanh_and_mul", c10::kCUDA, &act_and_mul<2>);
  ops.def("gelu_new(Tensor! out, Tensor input) -> ()");
  ops.impl("gelu_new", c10::kCUDA, &activation<0>);
  ops.def("gelu_fast(Tensor! out, Tensor input) -> ()");
  ops.impl("gelu_fast", c10::kCUDA, &activation<1>);
  ops.def("gelu_quick(Tensor! out, Tensor input) -> ()");
  ops.impl("gelu_quick", c10::kCUDA, &activation<2>);

  ops.def("rms_norm(Tensor! out, Tensor input, Tensor weight, float epsilon) -> ()");
  ops.impl("rms_norm", c10::kCUDA, &rms_norm);
  ops.def("fused_add_rms_norm(Tensor! input, Tensor! residual, Tensor weight, float epsilon) -> ()");
  ops.impl("fused_add_rms_norm", c10::kCUDA, &fused_add_rms_norm);

  ops.def("rotary_embedding(Tensor positions, Tensor! query, Tensor! key, int head_size, Tensor cos_sin_cache, "
          "bool is_neox) -> ()");
  ops.impl("rotary_embedding", c10::kCUDA, &rotary_embedding);
  ops.def("batched_rotary_embedding(Tensor positions, Tensor! query, Tensor! key, int head_size, "
          "Tensor cos_sin_cache, bool is_neox, int rot_dim, Tensor cos_sin_cache_offsets) -> ()");
  ops.impl("batched_rotary_embedding", c10::kCUDA, &batched_rotary_embedding);

  ops.def("marlin_gemm(Tensor a, Tensor b_q_weight, Tensor b_scales, Tensor workspace, int size_m, int size_n, "
          "int size_k) -> Tensor");
  ops.impl("marlin_gemm", c10::kCUDA, &marlin_gemm);
  ops.def("fp8_marlin_gemm(Tensor a, Tensor b_q_weight, Tensor b_scales, Tensor workspace, int num_bits, "
          "int size_m, int size_n, int size_k) -> Tensor");
  ops.impl("fp8_marlin_gemm", c10::kCUDA, &fp8_marlin_gemm);
  ops.def("gptq_gemm(Tensor a, Tensor b_q_weight, Tensor b_gptq_qzeros, Tensor b_gptq_scales, Tensor b_g_idx, "
          "bool use_exllama, int bit) -> Tensor");
  ops.impl("gptq_gemm", c10::kCUDA, &gptq_gemm);
  ops.def("gptq_shuffle(Tensor! q_weight, Tensor q_perm, int bit) -> ()");
  ops.impl("gptq_shuffle", c10::kCUDA, &gptq_shuffle);
  ops.def("awq_gemm(Tensor _in_feats, Tensor _kernel, Tensor _scaling_factors, Tensor _zeros, int split_k_iters) "
          "-> Tensor");
  ops.impl("awq_gemm", c10::kCUDA, &awq_gemm);
  ops.def("awq_dequantize(Tensor _kernel, Tensor _scaling_factors, Tensor _zeros, int split_k_iters, int thx, "
          "int thy) -> Tensor");
  ops.impl("awq_dequantize", c10::kCUDA, &awq_dequantize);

  ops.def("cutlass_scaled_mm(Tensor! out, Tensor a, Tensor b, Tensor a_scales, Tensor b_scales, Tensor? bias) -> ()");
  ops.impl("cutlass_scaled_mm", c10::kCUDA, &cutlass_scaled_mm);
  ops.def("cutlass_scaled_mm_supports_fp8(int cuda_device_capability) -> bool");
  ops.impl("cutlass_scaled_mm_supports_fp8", c10::DispatchKey::CompositeExplicitAutograd,
           &cutlass_scaled_mm_supports_fp8);
  ops.def("static_scaled_fp8_quant(Tensor! out, Tensor input, Tensor scale) -> ()");
  ops.impl("static_scaled_fp8_quant", c10::kCUDA, &scaled_fp8_quant<0>);
  ops.def("dynamic_scaled_fp8_quant(Tensor! out, Tensor input, Tensor! scale) -> ()");
  ops.impl("dynamic_scaled_fp8_quant", c10::kCUDA, &scaled_fp8_quant<1>);
  ops.def("static_scaled_int8_quant(Tensor! out, Tensor input, Tensor scale) -> ()");
  ops.impl("static_scaled_int8_quant", c10::kCUDA, &static_scaled_int8_quant);
  ops.def("dynamic_scaled_int8_quant(Tensor! out, Tensor input, Tensor! scale) -> ()");
  ops.impl("dynamic_scaled_int8_quant", c10::kCUDA, &dynamic_scaled_int8_quant);

  ops.def("gptq_marlin_repack(Tensor b_q_weight, Tensor perm, int size_k, int size_n, int num_bits) -> Tensor");
  ops.impl("gptq_marlin_repack", c10::kCUDA, &gptq_marlin_repack);
  ops.def("gptq_marlin_gemm(Tensor a, Tensor b_q_weight, Tensor b_scales, Tensor g_idx, Tensor perm, "
          "Tensor workspace, int num_bits, int size_m, int size_n, int size_k, bool is_k_full) -> Tensor");
  ops.impl("gptq_marlin_gemm", c10::kCUDA, &gptq_marlin_gemm);
}

TORCH_LIBRARY(_C_cache_ops, cache_ops) {
  cache_ops.def("swap_blocks(Tensor src, Tensor! dst, Tensor block_mapping) -> ()");
  cache_ops.impl("swap_blocks", c10::kCUDA, &swap_blocks);
  cache_ops.def("copy_blocks(Tensor[]! key_caches, Tensor[]! value_caches, Tensor block_mapping) -> ()");
  cache_ops.impl("copy_blocks", c10::kCUDA, &copy_blocks);
  cache_ops.def("reshape_and_cache(Tensor key, Tensor value, Tensor! key_cache, Tensor! value_cache, "
                "Tensor slot_mapping, str kv_cache_dtype, float kv_scale) -> ()");
  cache_ops.impl("reshape_and_cache", c10::kCUDA, &reshape_and_cache);
  cache_ops.def("reshape_and_cache_flash(Tensor key, Tensor value, Tensor! key_cache, Tensor! value_cache, "
                "Tensor slot_mapping, str kv_cache_dtype) -> ()");
  cache_ops.impl("reshape_and_cache_flash", c10::kCUDA, &reshape_and_cache_flash);
  cache_ops.def("convert_fp8(Tensor! dst_cache, Tensor src_cache, float scale, str kv_cache_dtype) -> ()");
  cache_ops.impl("convert_fp8", c10::kCUDA, &convert_fp8);
}

// no tensor arguments: the reference registers these under kCUDA (torch_bindings.cpp:244-256) but a call
// without a tensor carries no backend key, so they get the catch-all key
TORCH_LIBRARY(_C_cuda_utils, cuda_utils) {
  cuda_utils.def("get_device_attribute(int attribute, int device_id) -> int");
  cuda_utils.impl("get_device_attribute", c10::DispatchKey::CompositeExplicitAutograd, &get_device_attribute);
  cuda_utils.def("get_max_shared_memory_per_block_device_attribute(int device_id) -> int");
  cuda_utils.impl("get_max_shared_memory_per_block_device_attribute", c10::DispatchKey::CompositeExplicitAutograd,
                  &get_max_shared_memory_per_block_device_attribute);
}

TORCH_LIBRARY(_C_custom_ar, custom_ar) {
  custom_ar.def("init_custom_ar(Tensor meta, Tensor rank_data, str[] handles, int[] offsets, int rank, "
                "bool full_nvlink) -> int");
  custom_ar.impl("init_custom_ar", c10::kCUDA, &init_custom_ar);
  custom_ar.def("should_custom_ar(Tensor inp, int max_size, int world_size, bool full_nvlink) -> bool");
  custom_ar.impl("should_custom_ar", c10::kCUDA, &should_custom_ar);
  custom_ar.def("all_reduce_reg(int fa, Tensor inp, Tensor! out) -> ()");
  custom_ar.impl("all_reduce_reg", c10::kCUDA, &all_reduce_reg);
  custom_ar.def("all_reduce_unreg(int fa, Tensor inp, Tensor reg_buffer, Tensor! out) -> ()");
  custom_ar.impl("all_reduce_unreg", c10::kCUDA, &all_reduce_unreg);
  custom_ar.def("dispose(int _fa) -> ()");
  custom_ar.impl("dispose", c10::DispatchKey::CompositeExplicitAutograd, &dispose);
  custom_ar.def("meta_size() -> int");
  custom_ar.impl("meta_size", c10::DispatchKey::CompositeExplicitAutograd, &meta_size);
  custom_ar.def("register_buffer(int _fa, Tensor t, str[] handles, int[] offsets) -> ()");
  custom_ar.impl("register_buffer", c10::kCUDA, &register_buffer);
  custom_ar.def("get_graph_buffer_ipc_meta(int _fa) -> (Tensor, int[])");
  custom_ar.impl("get_graph_buffer_ipc_meta", c10::DispatchKey::CompositeExplicitAutograd,
                 &get_graph_buffer_ipc_meta);
  custom_ar.def("register_graph_buffers(int _fa, str[] handles, int[][] offsets) -> ()");
  custom_ar.impl("register_graph_buffers", c10::DispatchKey::CompositeExplicitAutograd, &register_graph_buffers);
}

// csrc/registration.h:17-22 (REGISTER_EXTENSION): an importable module whose import runs the static
// registrations above; `nmv_abi_version` lets the Python side refuse a stale pair of libraries
static PyObject* abi_version(PyObject*, PyObject*) { return PyLong_FromLong(nmv_abi_version()); }
static PyMethodDef module_methods[] = {{"abi_version", abi_version, METH_NOARGS, "C ABI version of libnmvllm_hip.so"},
                                       {nullptr, nullptr, 0, nullptr}};
static struct PyModuleDef module_def = {PyModuleDef_HEAD_INIT, "_C", nullptr, 0, module_methods};
PyMODINIT_FUNC PyInit__C() { return PyModule_Create(&module_def); }
