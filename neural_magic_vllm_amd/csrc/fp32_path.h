// Entry points of fp32_path.hip (float models on the paged-KV ops), called from the C ABI functions of
// attention_kernels.hip / cache_kernels.hip when dtype == NMV_F32.
#pragma once
#include "common.h"

namespace nmv {

struct F32Sparse {   // the block-sparse arguments of paged_attention (attention_kernels.cu:209-251); vert_stride <= 1: dense
  int tp_rank, local_blocks, vert_stride, block_size, head_sliding_step;
};

struct F32AttnArgs {
  float* exp_sums; float* max_logits; float* out; float* tmp_out;
  const float* query; const void* key_cache; const void* value_cache;
  int num_seqs, num_heads, head_size, num_kv_heads; float scale;
  const int32_t* block_tables; const int32_t* seq_lens;
  int block_size, max_seq_len, max_num_blocks_per_seq;
  const float* alibi_slopes; int64_t q_stride, kv_block_stride, kv_head_stride;
  float kv_scale; bool partitioned; hipStream_t stream; F32Sparse sparse;
};

int f32_paged_attention(const F32AttnArgs& a, bool fp8_cache);
int f32_reshape_and_cache(const void* key, const void* value, void* key_cache, void* value_cache,
                          const int64_t* slot_mapping, int num_tokens, int num_kv_heads, int head_size, int block_size,
                          int64_t key_stride, int64_t value_stride, bool fp8_cache, float kv_scale, hipStream_t stream);
int f32_convert_fp8(void* dst, const void* src, int64_t num_blocks, int64_t block_stride, bool to_fp8, float scale,
                    hipStream_t stream);

}  // namespace nmv
