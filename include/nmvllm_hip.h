/*
 * nmvllm_hip.h -- C ABI of libnmvllm_hip.so: the MI355X (gfx950) hot path of nm-vllm 0.5.1.
 *
 * Every entry point replaces one native op the reference registers in
 * csrc/torch_bindings.cpp (TORCH_LIBRARY _C / _C_cache_ops / _C_cuda_utils) and declares in
 * csrc/ops.h / csrc/cache.h.  The signatures are plain C: raw device pointers, sizes, strides
 * (in ELEMENTS unless noted), enums for dtypes and a hipStream_t passed as void*.  No torch
 * types cross this boundary; the Python host layer (neural_magic_vllm_amd/_torch_bindings.py)
 * re-creates the reference's torch.ops._C.* schemas on top of it.
 *
 * Conventions
 *  - return value: 0 = NMV_OK, <0 = error; nmv_last_error() returns the message of the last
 *    failing call on the calling thread (the analogue of TORCH_CHECK -> RuntimeError).
 *  - all launches are asynchronous on `stream`, never synchronise, never allocate
 *    (hipGraph-capturable), except nmv_swap_blocks which issues hipMemcpyAsync per block like
 *    the reference (csrc/cache_kernels.cu:24-63).
 *  - pointers marked `dev` are device memory; `host` is host memory.
 */
#ifndef NMVLLM_HIP_H_
#define NMVLLM_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NMV_OK 0
#define NMV_ERR_INVALID (-1) /* argument / shape / dtype check failed (TORCH_CHECK analogue) */
#define NMV_ERR_HIP (-2)     /* HIP runtime reported an error at launch */
#define NMV_ERR_UNSUPPORTED (-3)

/* activation / model dtype (scalar_t in the reference) */
typedef enum { NMV_F16 = 0, NMV_BF16 = 1, NMV_F32 = 2 } nmv_dtype_t;
/* kv-cache dtype: "auto" or "fp8"/"fp8_e4m3" (csrc/quantization/fp8/nvidia/quant_utils.cuh:545-571);
 * on gfx950 fp8 is OCP e4m3fn */
typedef enum { NMV_KV_AUTO = 0, NMV_KV_FP8_E4M3 = 1 } nmv_kv_dtype_t;
/* 8-bit operand type of nmv_scaled_mm */
typedef enum { NMV_I8 = 0, NMV_FP8_E4M3 = 1 } nmv_q8_dtype_t;

const char* nmv_last_error(void);
/* ABI version, bumped on any signature change */
int nmv_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * KV-cache ops  (reference: csrc/cache.h:8-32, csrc/cache_kernels.cu)
 * ---------------------------------------------------------------------------------------- */

/* reshape_and_cache  (csrc/cache_kernels.cu:253-278, kernel :152-204)
 * key/value: [num_tokens, num_kv_heads, head_size] with token strides key_stride/value_stride.
 * key_cache: [num_blocks, num_kv_heads, head_size/x, block_size, x], x = 16/sizeof(cache elem)
 * value_cache: [num_blocks, num_kv_heads, head_size, block_size]
 * slot_mapping: int64 [num_tokens]; slot < 0 = padding token, skipped. */
int nmv_reshape_and_cache(const void* key, const void* value, void* key_cache, void* value_cache,
                          const int64_t* slot_mapping, int num_tokens, int num_kv_heads,
                          int head_size, int block_size, int64_t key_stride, int64_t value_stride,
                          nmv_dtype_t dtype, nmv_kv_dtype_t kv_dtype, float kv_scale,
                          void* stream);

/* reshape_and_cache_flash  (csrc/cache_kernels.cu:280-316): caches are
 * [num_blocks, block_size, num_kv_heads, head_size]; block_stride = cache.stride(0). */
int nmv_reshape_and_cache_flash(const void* key, const void* value, void* key_cache,
                                void* value_cache, const int64_t* slot_mapping, int num_tokens,
                                int num_kv_heads, int head_size, int block_size,
                                int64_t key_stride, int64_t value_stride, int64_t block_stride,
                                nmv_dtype_t dtype, void* stream);

/* copy_blocks  (csrc/cache_kernels.cu:101-148): key_cache_ptrs/value_cache_ptrs are DEVICE arrays
 * of num_layers device pointers; block_mapping: dev int64 [num_pairs, 2] (src, dst);
 * numel_per_block elements of elem_size bytes each. */
int nmv_copy_blocks(void* const* key_cache_ptrs, void* const* value_cache_ptrs,
                    const int64_t* block_mapping, int num_layers, int num_pairs,
                    int64_t numel_per_block, int elem_size, void* stream);

/* swap_blocks  (csrc/cache_kernels.cu:24-63): block_mapping is a HOST int64 [num_pairs, 2];
 * kind: 0 = device->device, 1 = host->device, 2 = device->host. */
int nmv_swap_blocks(const void* src, void* dst, const int64_t* block_mapping_host, int num_pairs,
                    int64_t block_bytes, int kind, void* stream);

/* convert_fp8  (csrc/cache_kernels.cu:339-389): dst/src are [num_blocks, block_stride];
 * exactly one side is fp8 (uint8), the other `dtype`.  to_fp8 != 0: dst = fp8(src / scale);
 * else dst = float(src) * scale. */
int nmv_convert_fp8(void* dst, const void* src, int64_t num_blocks, int64_t block_stride,
                    nmv_dtype_t dtype, int to_fp8, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Paged attention  (reference: csrc/ops.h:6-25, csrc/attention/attention_kernels.cu)
 * ---------------------------------------------------------------------------------------- */

/* paged_attention_v1  (attention_kernels.cu:805-826)
 * out, query: [num_seqs, num_heads, head_size]; q_stride = query.stride(0); out is contiguous.
 * block_tables: int32 [num_seqs, max_num_blocks_per_seq]; seq_lens: int32 [num_seqs].
 * kv_block_stride = key_cache.stride(0), kv_head_stride = key_cache.stride(1).
 * alibi_slopes: float [num_heads] or NULL.
 * Block-sparse attention (attention_kernels.cu:209-251,385-393; the last five ints of the reference op):
 * blocksparse_vert_stride <= 1 = dense; else a head attends to the last `blocksparse_local_blocks` blocks of
 * `blocksparse_block_size` tokens and to every block whose id + head offset is a multiple of the stride, the
 * offset sliding with the query head (head_sliding_step >= 0) or the kv head (< 0) and tp_rank. */
int nmv_paged_attention_v1(void* out, const void* query, const void* key_cache,
                           const void* value_cache, int num_seqs, int num_heads, int head_size,
                           int num_kv_heads, float scale, const int32_t* block_tables,
                           const int32_t* seq_lens, int block_size, int max_seq_len,
                           int max_num_blocks_per_seq, const float* alibi_slopes,
                           int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                           nmv_dtype_t dtype, nmv_kv_dtype_t kv_dtype, float kv_scale,
                           int tp_rank, int blocksparse_local_blocks, int blocksparse_vert_stride,
                           int blocksparse_block_size, int blocksparse_head_sliding_step, void* stream);

/* paged_attention_v2  (attention_kernels.cu:966-990): partition size 512.
 * exp_sums, max_logits: float [num_seqs, num_heads, max_num_partitions];
 * tmp_out: [num_seqs, num_heads, max_num_partitions, head_size]. */
int nmv_paged_attention_v2(void* out, float* exp_sums, float* max_logits, void* tmp_out,
                           const void* query, const void* key_cache, const void* value_cache,
                           int num_seqs, int num_heads, int head_size, int num_kv_heads,
                           float scale, const int32_t* block_tables, const int32_t* seq_lens,
                           int block_size, int max_seq_len, int max_num_blocks_per_seq,
                           const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                           int64_t kv_head_stride, nmv_dtype_t dtype, nmv_kv_dtype_t kv_dtype,
                           float kv_scale, int tp_rank, int blocksparse_local_blocks,
                           int blocksparse_vert_stride, int blocksparse_block_size,
                           int blocksparse_head_sliding_step, void* stream);

/* ------------------------------------------------------------------------------------------
 * Glue ops so a whole decoder layer runs without the reference's csrc
 * (csrc/layernorm_kernels.cu, csrc/pos_encoding_kernels.cu, csrc/activation_kernels.cu)
 * ---------------------------------------------------------------------------------------- */

/* rms_norm (layernorm_kernels.cu:22-44,292-313): out/input [num_tokens, hidden] contiguous */
int nmv_rms_norm(void* out, const void* input, const void* weight, float epsilon, int num_tokens,
                 int hidden_size, nmv_dtype_t dtype, void* stream);
/* fused_add_rms_norm (layernorm_kernels.cu:201-290,315-352): in place on input and residual */
int nmv_fused_add_rms_norm(void* input, void* residual, const void* weight, float epsilon,
                           int num_tokens, int hidden_size, nmv_dtype_t dtype, void* stream);
/* rotary_embedding (pos_encoding_kernels.cu:71-93,121-160): positions int64 [num_tokens];
 * query [num_tokens, num_heads*head_size] (stride query_stride), key likewise;
 * cos_sin_cache [max_position, rot_dim] in `dtype`. In place. */
int nmv_rotary_embedding(const int64_t* positions, void* query, void* key, int num_tokens,
                         int num_heads, int num_kv_heads, int head_size, int rot_dim,
                         int64_t query_stride, int64_t key_stride, const void* cos_sin_cache,
                         int is_neox, nmv_dtype_t dtype, void* stream);
/* batched_rotary_embedding (pos_encoding_kernels.cu:95-119,162-203):
 * cos_sin_cache_offsets int64 [num_tokens] added to the position before the cache lookup */
int nmv_batched_rotary_embedding(const int64_t* positions, void* query, void* key, int num_tokens,
                                 int num_heads, int num_kv_heads, int head_size, int rot_dim,
                                 int64_t query_stride, int64_t key_stride,
                                 const void* cos_sin_cache, int is_neox,
                                 const int64_t* cos_sin_cache_offsets, nmv_dtype_t dtype,
                                 void* stream);
/* act_and_mul family (activation_kernels.cu:14-26,63-90): input [num_tokens, 2*d] -> out [.., d].
 * act: 0 silu_and_mul, 1 gelu_and_mul (erf), 2 gelu_tanh_and_mul */
int nmv_act_and_mul(void* out, const void* input, int num_tokens, int d, int act,
                    nmv_dtype_t dtype, void* stream);
/* element-wise activations (activation_kernels.cu:96-162): act: 0 gelu_new, 1 gelu_fast,
 * 2 gelu_quick; input/out [num_tokens, d] */
int nmv_activation(void* out, const void* input, int num_tokens, int d, int act,
                   nmv_dtype_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused decode-step launches.  Not ops of nm-vllm 0.5.1 (which launches the parts separately; later
 * vLLM grew similar fusions): each is bit-identical to the sequence of reference ops it names and
 * exists because a launch on a [B, hidden] tensor is ~5 us of latency, not bandwidth.
 * ------------------------------------------------------------------------------------------ */
/* gate_up GEMM with silu_and_mul folded into the epilogue: b_q_weight / b_scales are the Marlin
 * tensors (gptq_marlin_repack / marlin_permute_scales) of a 4-bit symmetric weight, group 128 or
 * channelwise, no act-order, whose OUTPUT COLUMNS were interleaved before the repack so that
 * 64-column chunk c = [gate 32c..32c+31 | up 32c..32c+31]; c: [size_m, size_n / 2].  K % 256 == 0,
 * N % 128 == 0.  The roundings of gptq_marlin_gemm on the original weight + silu_and_mul; bit-identical to them when
 * that GEMM does not split K across workgroups, otherwise equal up to the order of the fp32 partial sums (the fused
 * launch never splits K): at most an ulp of the model dtype on a few elements. */
int nmv_gptq_marlin_gemm_silu_mul(void* c, const void* a, const int32_t* b_q_weight,
                                  const void* b_scales, int32_t* workspace, int64_t workspace_len,
                                  int size_m, int size_n, int size_k, int num_groups,
                                  nmv_dtype_t dtype, void* stream);
/* Deferred split-K reduction: nmv_gptq_marlin_gemm_partial stores the fp32
 * partial tiles slab[splits, size_m, size_n] (splits = nmv_gptq_marlin_gemm_partial_splits(m, n, k, num_groups),
 * num_groups = rows of b_scales; 0 = shape not supported) and skips the ticket / last-arriver pass; the consumer sums the slabs in
 * split order and rounds to the model dtype, bit-identical to nmv_gptq_marlin_gemm's own output.
 * 4-bit symmetric codes, group 128 or channelwise, no act-order, K % 256 == 0. */
int nmv_gptq_marlin_gemm_partial_splits(int size_m, int size_n, int size_k, int num_groups);
int nmv_gptq_marlin_gemm_partial(float* slab, int64_t slab_bytes, const void* a,
                                 const int32_t* b_q_weight, const void* b_scales, int size_m,
                                 int size_n, int size_k, int num_groups, nmv_dtype_t dtype,
                                 void* stream);
/* fused_add_rms_norm whose input is the slab sum: residual += round(sum_s slab[s]) (in place),
 * out = rms_norm(residual) * weight.  hidden % 8 == 0, hidden <= 8192. */
int nmv_fused_add_rms_norm_partial(void* out, const float* slab, int splits, void* residual,
                                   const void* weight, float epsilon, int num_tokens,
                                   int hidden_size, nmv_dtype_t dtype, void* stream);
/* ... and with the slabs in the MODEL dtype, [splits, num_tokens, hidden_size] of 2-byte elements (mode 3 of
 * nmv_w4_native_gemm: prompt-sized calls, where the slabs are most of the launch's bytes): summed in fp32, rounded once. */
int nmv_fused_add_rms_norm_partial16(void* out, const void* slab, int splits, void* residual,
                                     const void* weight, float epsilon, int num_tokens,
                                     int hidden_size, nmv_dtype_t dtype, void* stream);
/* rotary_embedding_and_cache whose qkv row is the slab sum of nmv_gptq_marlin_gemm_partial
 * (slab [splits, num_tokens, (heads + 2 kv_heads) * head_size] fp32): qkv_out receives the rounded
 * row with q / k rotated (neox style, rot_dim == head_size; cos_sin_cache [max_pos, head_size]); k / v
 * go to the paged cache unless key_cache == NULL. */
int nmv_rotary_embedding_and_cache_partial(const int64_t* positions, const float* slab, int splits,
                                           void* qkv_out, int num_tokens, int num_heads,
                                           int num_kv_heads, int head_size, const void* cos_sin_cache,
                                           void* key_cache, void* value_cache,
                                           const int64_t* slot_mapping, int block_size,
                                           nmv_dtype_t dtype, nmv_kv_dtype_t kv_dtype, float kv_scale,
                                           void* stream);
/* ... and with the slabs in the MODEL dtype (mode 3 of nmv_w4_native_gemm) */
int nmv_rotary_embedding_and_cache_partial16(const int64_t* positions, const void* slab, int splits,
                                             void* qkv_out, int num_tokens, int num_heads,
                                             int num_kv_heads, int head_size, const void* cos_sin_cache,
                                             void* key_cache, void* value_cache,
                                             const int64_t* slot_mapping, int block_size,
                                             nmv_dtype_t dtype, nmv_kv_dtype_t kv_dtype, float kv_scale,
                                             void* stream);
/* paged_attention_v1 / v2 whose query and new key / value are still the fp32 split-K slabs of the
 * qkv projection (slab [splits, num_seqs, (heads + 2 kv_heads) * head_size]): sum + round, neox
 * rotary embedding (rot_dim == head_size), the new token's k / v stored at slot_mapping[seq], then
 * attention over the cache -- rotary_embedding + reshape_and_cache + paged_attention in one launch,
 * bit-identical to them.  splits == 0: `slab` is instead the finished, contiguous qkv row in the
 * model dtype (W8A8 / unquantised projections).  Decode batches only (one new token per sequence,
 * positions[seq] == seq_lens[seq] - 1); no ALiBi. */
int nmv_paged_attention_v1_rope_partial(
    void* out, const float* slab, int splits, const int64_t* positions, const void* cos_sin_cache,
    const int64_t* slot_mapping, void* key_cache, void* value_cache, int num_seqs, int num_heads,
    int head_size, int num_kv_heads, float scale, const int32_t* block_tables, const int32_t* seq_lens,
    int block_size, int max_seq_len, int max_num_blocks_per_seq, int64_t kv_block_stride,
    int64_t kv_head_stride, nmv_dtype_t dtype, nmv_kv_dtype_t kv_dtype, float kv_scale, void* stream);
int nmv_paged_attention_v2_rope_partial(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const float* slab, int splits,
    const int64_t* positions, const void* cos_sin_cache, const int64_t* slot_mapping, void* key_cache,
    void* value_cache, int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int block_size, int max_seq_len,
    int max_num_blocks_per_seq, int64_t kv_block_stride, int64_t kv_head_stride, nmv_dtype_t dtype,
    nmv_kv_dtype_t kv_dtype, float kv_scale, void* stream);
/* greedy sampling (torch.argmax of the reference's Sampler greedy branch; ties -> lowest index) of
 * logits [num_seqs, vocab_size] (row stride in elements) into next_tokens int64[num_seqs], and --
 * when positions != NULL -- the on-device advance of a decode batch: input_ids = token,
 * positions += 1, seq_lens += 1, slot_mapping = block_tables[pos / bs] * bs + pos % bs
 * (worker/model_runner.py:572-580), so that a captured step can be replayed back to back.
 * scratch: nmv_greedy_sample_scratch_bytes(num_seqs) bytes. */
int64_t nmv_greedy_sample_scratch_bytes(int num_seqs);
int nmv_greedy_sample_advance(int64_t* next_tokens, const void* logits, int64_t row_stride,
                              int num_seqs, int vocab_size, nmv_dtype_t dtype, void* scratch,
                              int64_t scratch_bytes, int64_t* input_ids, int64_t* positions,
                              int* seq_lens, int64_t* slot_mapping, const int* block_tables,
                              int max_blocks_per_seq, int block_size, void* stream);
/* Vocab-parallel greedy sampling without gathering the logits: every rank reduces its shard
 * [num_seqs, local_vocab] to a record -- float value[p] then int32 global_index[p], p =
 * nmv_greedy_record_elems(num_seqs) -- the records are all-gathered (nmv_ar_all_gather or any
 * all-gather) and nmv_greedy_sample_finish picks the winner per row (ties -> lowest global index =
 * torch.argmax of the gathered logits) and optionally advances the decode batch like
 * nmv_greedy_sample_advance.  scratch as for nmv_greedy_sample_advance. */
int nmv_greedy_record_elems(int num_seqs);
int nmv_greedy_sample_shard(void* record, const void* logits, int64_t row_stride, int num_seqs,
                            int local_vocab, int index_offset, nmv_dtype_t dtype, void* scratch,
                            int64_t scratch_bytes, void* stream);
int nmv_greedy_sample_finish(int64_t* next_tokens, const void* gathered, int world, int num_seqs,
                             int64_t* input_ids, int64_t* positions, int* seq_lens,
                             int64_t* slot_mapping, const int* block_tables, int max_blocks_per_seq,
                             int block_size, void* stream);
/* rotary_embedding (pos_encoding_kernels.cu:121-160) followed by reshape_and_cache
 * (cache_kernels.cu:253-278): query / key rotated in place, then key / value of every token with
 * slot_mapping[t] >= 0 written to the paged cache. */
int nmv_rotary_embedding_and_cache(const int64_t* positions, void* query, void* key,
                                   const void* value, int num_tokens, int num_heads,
                                   int num_kv_heads, int head_size, int rot_dim, int64_t query_stride,
                                   int64_t key_stride, int64_t value_stride,
                                   const void* cos_sin_cache, int is_neox, void* key_cache,
                                   void* value_cache, const int64_t* slot_mapping, int block_size,
                                   nmv_dtype_t dtype, nmv_kv_dtype_t kv_dtype, float kv_scale,
                                   void* stream);
/* rms_norm (residual == NULL) or fused_add_rms_norm (residual updated in place, input untouched)
 * followed by dynamic per-token scaled_int8_quant (int8_quant_kernels.cu:37-75): out_q int8
 * [num_tokens, hidden], scales float [num_tokens].  hidden % 8 == 0, hidden <= 8192. */
int nmv_rms_norm_dynamic_int8_quant(void* out_q, float* scales, const void* input, void* residual,
                                    const void* weight, float epsilon, int num_tokens,
                                    int hidden_size, nmv_dtype_t dtype, void* stream);
/* silu_and_mul followed by dynamic per-token scaled_int8_quant: input [num_tokens, 2*d] ->
 * out_q int8 [num_tokens, d], scales float [num_tokens].  d % 8 == 0, d <= 32768. */
int nmv_silu_and_mul_dynamic_int8_quant(void* out_q, float* scales, const void* input,
                                        int num_tokens, int d, nmv_dtype_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * W4A16 / W8A16 (GPTQ-Marlin format)  (csrc/ops.h:86-94, csrc/quantization/gptq_marlin/)
 * ---------------------------------------------------------------------------------------- */

/* gptq_marlin_repack  (gptq_marlin_repack.cu:267-348): GPTQ qweight int32 [size_k/pack, size_n]
 * -> Marlin int32 [size_k/16, size_n*16/pack]; perm: int32 [size_k] or NULL (act-order). */
int nmv_gptq_marlin_repack(const int32_t* b_q_weight, const int32_t* perm, int32_t* out,
                           int size_k, int size_n, int num_bits, void* stream);

/* bytes of scratch nmv_gptq_marlin_gemm needs for (size_m, size_n, size_k): fp32 split-K slabs plus, for the
 * variants named in `variant` (bit 0: act-order, bit 1: 8-bit codes), the permuted copy of A
 * (gptq_marlin.cu:1783-1785 allocates the latter) and the slabs of the generic kernel */
int64_t nmv_gptq_marlin_gemm_scratch_bytes(int size_m, int size_n, int size_k, int variant);

/* gptq_marlin_gemm  (gptq_marlin.cu:1735-1868)
 * c[size_m,size_n] = a[size_m,size_k] @ dequant(b_q_weight) ; a, c, b_scales in `dtype`.
 * b_q_weight: Marlin int32 [size_k/16, size_n*16/pack]; b_scales: [num_groups, size_n]
 * (marlin_permute_scales layout, gptq_marlin.py:47-56); g_idx/perm: int32 [size_k] or NULL;
 * workspace: int32 [>= size_n/64*16], zero on entry, zero on exit (split-K tickets);
 * scratch: >= nmv_gptq_marlin_gemm_scratch_bytes(). */
int nmv_gptq_marlin_gemm(void* c, const void* a, const int32_t* b_q_weight, const void* b_scales,
                         const int32_t* g_idx, const int32_t* perm, int32_t* workspace,
                         int64_t workspace_len, void* scratch, int64_t scratch_bytes,
                         int num_bits, int size_m, int size_n, int size_k, int num_groups,
                         int is_k_full, nmv_dtype_t dtype, void* stream);

/* Marlin-format GEMM with per-(group, column) zero points -- not an op of nm-vllm 0.5.1 (later vLLM:
 * awq_marlin): asymmetric AWQ / GPTQ checkpoints repacked once at load run the tuned Marlin kernel.
 * b_zeros: z in the model dtype, [num_groups, N] in marlin_permute_scales order; 4-bit codes, group
 * 128, K % 256 == 0; scratch as nmv_gptq_marlin_gemm.  c = a . ((q - z) * s). */
int nmv_marlin_zp_gemm(void* c, const void* a, const int32_t* b_q_weight, const void* b_scales,
                       const void* b_zeros, int32_t* workspace, int64_t workspace_len, void* scratch,
                       int64_t scratch_bytes, int size_m, int size_n, int size_k, int num_groups,
                       nmv_dtype_t dtype, void* stream);
/* AWQ qweight int32 [K, N/8] -> Marlin int32 [K/16, N*2] (bit-exact code shuffle) */
int nmv_awq_marlin_repack(int32_t* out, const int32_t* qweight, int size_k, int size_n, void* stream);

/* marlin_gemm  (csrc/quantization/marlin/dense/marlin_cuda_kernel.cu:1045-1136): legacy Marlin
 * checkpoints, 4-bit, group -1 / 128; same tensors as nmv_gptq_marlin_gemm without act-order. */
int nmv_marlin_gemm(void* c, const void* a, const int32_t* b_q_weight, const void* b_scales,
                    int32_t* workspace, int64_t workspace_len, void* scratch, int64_t scratch_bytes,
                    int size_m, int size_n, int size_k, int num_groups, nmv_dtype_t dtype,
                    void* stream);

/* fp8_marlin_gemm  (csrc/quantization/fp8/fp8_marlin.cu:1212-1308): fp8-e4m3 weights packed by
 * pack_fp8_to_int32 + gptq_marlin_repack(bits=8), channelwise (or grouped) scales in
 * marlin_permute_scales order.  scratch: >= nmv_fp8_marlin_gemm_scratch_bytes() (split-K slabs).
 * workspace: the reference's zeroed lock array (size_n / 64 * 16 ints; zero again on return) -- with channelwise
 * scales and size_k % 256 == 0 the call runs the tall 8-bit Marlin kernel and takes its split-K tickets from
 * there (a null workspace, grouped scales or another size_k take the generic kernel). */
int64_t nmv_fp8_marlin_gemm_scratch_bytes(int size_m, int size_n, int size_k);
int nmv_fp8_marlin_gemm(void* c, const void* a, const int32_t* b_q_weight, const void* b_scales,
                        int32_t* workspace, int64_t workspace_len, void* scratch, int64_t scratch_bytes,
                        int num_bits, int size_m, int size_n, int size_k, int num_groups, nmv_dtype_t dtype,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * GPTQ (exllama) and AWQ checkpoints  (csrc/ops.h:66-73,119-124)
 * ---------------------------------------------------------------------------------------- */

/* gptq_gemm  (csrc/quantization/gptq/q_gemm.cu:1823-1846): c = a . ((q - (z + 1)) * s).
 * b_q_weight int32 [K/pack, N]; qzeros int32 [G, N/pack]; scales [G, N]; use_exllama != 0:
 * weights were processed by nmv_gptq_shuffle and b_g_idx is the act-order permutation (or NULL);
 * use_exllama == 0: b_g_idx is the per-row group index (or NULL). bit in {2, 4, 8}. */
int nmv_gptq_gemm(void* c, const void* a, const int32_t* b_q_weight, const int32_t* b_gptq_qzeros,
                  const void* b_gptq_scales, const int32_t* b_g_idx, int use_exllama, int bit,
                  int size_m, int size_n, int size_k, int num_groups, nmv_dtype_t dtype,
                  void* scratch, int64_t scratch_bytes, void* stream);

/* fp32 split-K slabs of gptq_gemm / awq_gemm (4-bit, no act-order: the streaming kernel); the
 * reference allocates its own [split_k, M, N] partials per call (gemm_kernels.cu:520-523).
 * 0 when no scratch is needed. */
int64_t nmv_wq_gemm_scratch_bytes(int size_m, int size_n, int size_k);

/* gptq_shuffle  (q_gemm.cu:1848-1856): in place; q_perm NULL = no act-order.  tmp: scratch of
 * the size of q_weight (needed only with q_perm). */
int nmv_gptq_shuffle(int32_t* q_weight, const int32_t* q_perm, int32_t* tmp, int size_k, int size_n,
                     int bit, void* stream);

/* awq_gemm  (csrc/quantization/awq/gemm_kernels.cu:492-549): c = a . ((q - z) * s);
 * qweight int32 [K, N/8], qzeros int32 [G, N/8], scales [G, N]. */
int nmv_awq_gemm(void* c, const void* a, const int32_t* qweight, const void* scales,
                 const int32_t* qzeros, int size_m, int size_n, int size_k, int num_groups,
                 nmv_dtype_t dtype, void* scratch, int64_t scratch_bytes, void* stream);

/* awq_dequantize  (gemm_kernels.cu:436-490): out [K, N] */
int nmv_awq_dequantize(void* out, const int32_t* qweight, const void* scales, const int32_t* qzeros,
                       int size_n, int size_k, int num_groups, nmv_dtype_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Prompt (prefill) attention: varlen causal GQA flash-attention forward.
 * (reference: the prompt branch of vllm/attention/backends/rocm_flash_attn.py:349-430, which
 *  dispatches to Triton / CK flash-attention or torch SDPA; no C++ symbol exists there.)
 * q [tokens, H, D], k / v [tokens, KVH, D], out [tokens, H, D]: token strides in elements (q, k, v
 * may be slices of the fused qkv GEMM output); cu_seqlens int32 [num_seqs + 1] on the device.
 * ---------------------------------------------------------------------------------------- */
int nmv_prefill_attention_supported(int head_size);
/* Prefix-enabled prefill (PagedAttention.forward_prefix, vllm/attention/ops/paged_attn.py:184-216 ->
 * Triton context_attention_fwd): the new tokens [query_start_loc[i], query_start_loc[i+1]) of sequence
 * i attend causally to ALL its keys [0, seq_lens[i]) read from the paged cache (the backend has
 * written the new tokens' K/V before the call); context_lens[i] = seq_lens[i] - new tokens.
 * kv cache dtype auto (fp16 / bf16) only, as in the reference (its forward_prefix takes no cache dtype);
 * layouts as paged_attention.  alibi_slopes: float [H] or NULL -- bias slope * (key - query) on the scaled
 * logits (prefix_prefill.py:552-557); sliding_window > 0: a query sees the keys fewer than that many
 * positions back (:130-144, :201-204); both also on nmv_prefill_attention. */
int nmv_prefix_prefill_attention(void* out, const void* q, const void* key_cache,
                                 const void* value_cache, const int32_t* block_tables,
                                 const int32_t* query_start_loc, const int32_t* seq_lens,
                                 const int32_t* context_lens, int num_seqs, int max_query_len,
                                 int max_blocks_per_seq, int block_size, int num_heads,
                                 int num_kv_heads, int head_size, float scale, int64_t q_stride,
                                 int64_t o_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                                 const float* alibi_slopes, int sliding_window, nmv_dtype_t dtype,
                                 void* stream);
int nmv_prefill_attention(void* out, const void* q, const void* k, const void* v,
                          const int32_t* cu_seqlens, int num_seqs, int max_seq_len, int num_heads,
                          int num_kv_heads, int head_size, float scale, int64_t q_stride,
                          int64_t kv_stride, int64_t o_stride, const float* alibi_slopes,
                          int sliding_window, nmv_dtype_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * W8A8: activation quantisers and the scaled matmul
 * (csrc/ops.h:101-114,126-130; csrc/quantization/compressed_tensors/int8_quant_kernels.cu,
 *  csrc/quantization/fp8/common.cu, csrc/quantization/cutlass_w8a8/scaled_mm_entry.cu)
 * ---------------------------------------------------------------------------------------- */

/* static_scaled_int8_quant (int8_quant_kernels.cu:77-95; dynamic = 0, scale: float[1]) and
 * dynamic_scaled_int8_quant (:97-115; dynamic = 1, scale: float[num_tokens] written).
 * input [num_tokens, hidden] contiguous -> out int8 same shape. */
int nmv_scaled_int8_quant(void* out, const void* input, float* scale, int num_tokens,
                          int hidden_size, int dynamic, nmv_dtype_t dtype, void* stream);

/* static_scaled_fp8_quant / dynamic_scaled_fp8_quant (fp8/common.cu:129-165): per-tensor scale
 * float[1]; dynamic: scale must be 0 on entry and receives absmax/448.  out: e4m3fn bytes. */
int nmv_scaled_fp8_quant(void* out, const void* input, float* scale, int64_t num_elems, int dynamic,
                         nmv_dtype_t dtype, void* stream);

/* cutlass_scaled_mm (scaled_mm_entry.cu:48-100): out[M,N] = a_scales * (b_scales * (a . b)) + bias.
 * a [M,K] row-major (lda), b column-major = [N,K] row-major (ldb = b.stride(1)), out row-major
 * (ldc); scales fp32 with numel 1 (per tensor) or M / N; bias in out_dtype or NULL.
 * scratch: optional device buffer of nmv_scaled_mm_scratch_bytes(M, N, K) bytes for the split-K
 * slabs used at M > 32 on long-K shapes; NULL / too small => the kernel runs unsplit (same result
 * for int8, fp32 summation order differs for fp8).  The reference allocates its CUTLASS workspace
 * inside the op the same way. */
int64_t nmv_scaled_mm_scratch_bytes(int M, int N, int K);
int nmv_scaled_mm(void* out, const void* a, const void* b, const float* a_scales,
                  const float* b_scales, const void* bias, int M, int N, int K, int64_t lda,
                  int64_t ldb, int64_t ldc, int a_scales_numel, int b_scales_numel,
                  nmv_q8_dtype_t in_dtype, nmv_dtype_t out_dtype, void* scratch,
                  int64_t scratch_bytes, void* stream);

/* cutlass_scaled_mm_supports_fp8 (scaled_mm_entry.cu:32-46) */
int nmv_cutlass_scaled_mm_supports_fp8(int64_t cuda_device_capability);

/* ------------------------------------------------------------------------------------------
 * device attributes (csrc/cuda_utils.h:3-5, csrc/cuda_utils_kernels.cu)
 * ---------------------------------------------------------------------------------------- */
int64_t nmv_get_device_attribute(int64_t attribute, int64_t device_id);
int64_t nmv_get_max_shared_memory_per_block_device_attribute(int64_t device_id);

/* ------------------------------------------------------------------------------------------
 * MFMA-native W4 tensors (not ops of nm-vllm 0.5.1; what GPTQMarlinLinearMethod keeps beside the Marlin tensor
 * for decode-sized calls).  nmv_w4_native_repack: GPTQ qweight int32 [K/8, N] (+ optional act-order row gather
 * perm[K]) -> native int32 [K/8 * N] (csrc/w4a16_gemm.hip).  nmv_w4_native_gemm: C = A . ((q - 8) * s), s the
 * natural [groups, N] scales; mode 0 = [M, N] in the model dtype (workspace = zeroed split-K tickets, scratch =
 * fp32 slabs, nmv_gptq_marlin_gemm_scratch_bytes), 1 = silu(gate) * up -> [M, N/2] on column-interleaved
 * gate_up weights, 2 = deferred reduction: fp32 slabs [nmv_w4_native_gemm_splits][M][N] in scratch.
 * Same arithmetic as nmv_gptq_marlin_gemm (fp32 group scaling), M tiles <= 64 rows. */
int nmv_w4_native_repack(const int32_t* qweight, const int32_t* perm, int32_t* out, int size_k, int size_n,
                         void* stream);
int nmv_w4_native_gemm_splits(int size_m, int size_n, int size_k, int num_groups);
/* For a caller that holds BOTH tensors of a weight: 1 when nmv_w4_native_gemm (its prompt-sized kernels, csrc/w4a16_prefill.hip:
 * 256 x 256 or 128 x 128 tiles, codes expanded once per workgroup) was measured ahead of nmv_gptq_marlin_gemm on the Marlin
 * tensor for a call of these sizes, else 0.  nmv_w4_native_gemm itself serves every row count: a caller that keeps only
 * the native tensor (GPTQMarlinLinearMethod by default) never asks. */
int nmv_w4_native_prefill_plan(int size_m, int size_n, int size_k);
int nmv_w4_native_gemm(void* c, const void* a, const int32_t* b_native, const void* b_scales, int32_t* workspace,
                       int64_t workspace_len, void* scratch, int64_t scratch_bytes, int size_m, int size_n,
                       int size_k, int num_groups, nmv_dtype_t dtype, int mode, void* stream);
/* mode 3: as mode 2 (deferred reduction: `scratch` receives slab[splits][size_m][size_n], no ticket, no reduction) with
 * the slabs in the MODEL dtype -- for the *_partial16 consumers.  Prompt-sized calls only: nmv_w4_native_gemm_slab16
 * returns 1 when a call of these sizes supports it (its slab count is nmv_w4_native_gemm_splits). */
int nmv_w4_native_gemm_slab16(int size_m, int size_n, int size_k);
/* Calls of 17..64 rows (group 128) may run csrc/w4a16_ring.hip: loader waves fill an LDS ring by LDS-DMA, consumer waves
 * wait on per-slot words in LDS with bounded spins.  nmv_w4_ring_timeouts: workgroups that gave up on a slot since the
 * library was loaded (0 in a healthy process; their tiles are garbage); synchronises the device; -1 on a HIP error. */
int nmv_w4_ring_timeouts(void);

/* ------------------------------------------------------------------------------------------
 * One-shot / two-shot P2P all-reduce over HIP IPC (the analogue of csrc/custom_all_reduce.cuh:130-250
 * and its dispatch rule :442-451; the `_C_custom_ar` ops of torch_bindings.cpp:262-294 are bound on top of
 * these entry points -- the reference compiles all of it out on ROCm).
 * nmv_ar_create allocates this rank's block (flags + 2 staging + 2 reduced-slice buffers) on the current
 * device and exports its IPC handle (nmv_ar_handle_bytes() bytes); nmv_ar_open maps the peers' blocks
 * (handles: world handles in rank order); nmv_ar_all_reduce sums `numel` fp16 / bf16 elements (bytes % 16
 * == 0, <= max_bytes) across the ranks in fp32, rank order, one kernel launch, capturable into a hipGraph:
 * one-shot (every rank reads all W buffers) or, by the reference's size rule, two-shot (reduce-scatter into
 * the owner's slice + all-gather) -- the same bits either way.  Flag waits are bounded: a call whose wait
 * ran out writes NaN and sets the error word that nmv_ar_error reports (after a device sync); the
 * communicator must not be used after that.
 * ------------------------------------------------------------------------------------------ */
int nmv_ar_handle_bytes(void);
int nmv_ar_create(void** state_out, int rank, int world, int64_t max_bytes, void* handle_out);
int nmv_ar_open(void* state, const void* handles);
int nmv_ar_all_reduce(void* state, const void* inp, void* out, int64_t numel, nmv_dtype_t dtype,
                      void* stream);
/* all-reduce whose local input is still the fp32 split-K slabs [splits, numel] of the row-parallel
 * GEMM (nmv_gptq_marlin_gemm_partial); bit-identical to nmv_gptq_marlin_gemm + nmv_ar_all_reduce */
int nmv_ar_all_reduce_partial(void* state, const float* slab, int splits, void* out, int64_t numel,
                              nmv_dtype_t dtype, void* stream);
/* all-reduce + fused_add_rms_norm in one launch: x = inp [rows, hidden] (model dtype) or, with
 * inp == NULL, the fp32 slabs [splits, rows, hidden]; residual += all_reduce(x) in place, out =
 * rms_norm(residual) * weight.  hidden % 8 == 0, <= 8192.  Bit-identical to the separate launches. */
int nmv_ar_all_reduce_add_rms_norm(void* state, const void* inp, const float* slab, int splits,
                                   void* residual, const void* weight, void* out, float epsilon,
                                   int rows, int hidden, nmv_dtype_t dtype, void* stream);
/* out[q * bytes_per_rank ...] = rank q's inp: all-gather of small per-rank records with the same
 * protocol (bytes_per_rank % 16 == 0, <= max_bytes) */
int nmv_ar_all_gather(void* state, const void* inp, void* out, int64_t bytes_per_rank, void* stream);
int nmv_ar_error(void* state);
/* algo: 0 = size rule of custom_all_reduce.cuh:442-451, 1 = one-shot, 2 = two-shot (same on all ranks) */
int nmv_ar_set_algo(void* state, int algo);
int nmv_ar_is_two_shot(void* state, int64_t bytes);
/* bound of a flag wait in milliseconds (default 2000) */
int nmv_ar_set_timeout_ms(void* state, int64_t ms);
int nmv_ar_destroy(void* state);

/* The reference's registered-buffer protocol (`_C_custom_ar`: init_custom_ar, register_buffer, all_reduce_reg /
 * _unreg, get_graph_buffer_ipc_meta, register_graph_buffers, dispose, meta_size -- csrc/custom_all_reduce.cu:12-160,
 * csrc/torch_bindings.cpp:262-294).  The caller owns `meta` (nmv_car_meta_size() bytes of flags followed by the
 * two-shot scratch, zeroed) and `rank_data` (device scratch for pointer tables) and exchanges the IPC handles of
 * meta and of every input buffer; peers read the inputs in place.  One-shot / two-shot by the reference's rule. */
int64_t nmv_car_meta_size(void);
/* `meta` as an uncached (fine-grained) allocation of the current device, zero-filled, with its IPC handle
 * (nmv_ar_handle_bytes() bytes): what the Python CustomAllreduce wraps in its meta tensor, so that flag stores and polls
 * of peers never sit in a cache another device cannot see.  nmv_car_meta_free releases it (after nmv_car_dispose). */
int nmv_car_meta_alloc(int64_t nbytes, void** ptr_out, void* handle_out);
int nmv_car_meta_free(void* ptr);
int nmv_car_init(void** state_out, void* meta, void* rank_data, int64_t rank_data_bytes, const void* handles,
                 const int64_t* offsets, int world, int rank, int full_link);
int nmv_car_register_buffer(void* state, const void* self, const void* handles, const int64_t* offsets);
int nmv_car_all_reduce(void* state, const void* inp, void* out, int64_t numel, nmv_dtype_t dtype, void* stream);
int nmv_car_graph_buffer_count(void* state);
int nmv_car_get_graph_buffer_ipc_meta(void* state, void* handles_out, int64_t* offsets_out);
int nmv_car_register_graph_buffers(void* state, const void* handles, const int64_t* offsets);
int nmv_car_set_algo(void* state, int algo);
int nmv_car_error(void* state);
int nmv_car_dispose(void* state);

/* Infinity-Cache prefetch (not an op of nm-vllm 0.5.1): reads `bytes` bytes at `ptr` (16-byte aligned) once with
 * `workgroups` (0: 64) workgroups of 256 threads on `stream` and stores nothing -- launched on a side stream it pulls the
 * NEXT decoder layer's weights into the 256 MiB on-die cache while the current layer computes (small batches: the layer
 * leaves HBM idle most of the time).  A hint: results never depend on it. */
int nmv_prefetch_l3(const void* ptr, int64_t bytes, int workgroups, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NMVLLM_HIP_H_ */
