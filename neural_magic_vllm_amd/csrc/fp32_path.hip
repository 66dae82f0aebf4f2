// float (fp32) models on the paged-KV ops: reshape_and_cache, convert_fp8, paged_attention_v1 / v2.
//
// The reference instantiates `float` for these ops beside half and bfloat16 (attention_kernels.cu:738-766 with the
// fp8 cache forms of csrc/quantization/fp8/amd/quant_utils.cuh:547-548; cache_kernels.cu:253-278 and :339-389 through
// DISPATCH_BY_KV_CACHE_DTYPE).  No quantized checkpoint of the hot path computes in fp32, so this file is about the
// boundary answering the same dtypes, not about speed: straightforward kernels, one thread per token for Q.K^T and one
// per head dimension for P.V, with the arithmetic of the 16-bit kernels (attention_kernels.hip): fp32 accumulation,
// online softmax, 1 / (sum + 1e-6), probabilities NOT rounded (the compute dtype is float), v2 partitions of 512
// tokens merged exactly like the reference's reduce kernel.  Cache layouts are the boundary's: K
// [blocks, kv_heads, head / x, block_size, x] with x = 16 / sizeof(cache element) = 4 (float cache) or 16 (fp8), V
// [blocks, kv_heads, head, block_size].
#include "common.h"
#include "fp32_path.h"

namespace nmv {

constexpr int F32_THREADS = 256;

template <bool FP8>
__device__ __forceinline__ float cache_elem(const void* base, int64_t idx, float kv_scale) {
  if constexpr (FP8) return fp8_to_f32(reinterpret_cast<const uint8_t*>(base)[idx]) * kv_scale;
  return reinterpret_cast<const float*>(base)[idx];
}

template <bool FP8>
__global__ __launch_bounds__(F32_THREADS) void paged_attention_f32_kernel(
    float* __restrict__ exp_sums, float* __restrict__ max_logits, float* __restrict__ out,
    const float* __restrict__ q, const void* __restrict__ k_cache, const void* __restrict__ v_cache, int num_heads,
    int num_kv_heads, int head_size, int block_size, float scale, const int* __restrict__ block_tables,
    const int* __restrict__ seq_lens, int max_num_blocks_per_seq, const float* __restrict__ alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride, float kv_scale, int partition_size,
    F32Sparse sp) {
  constexpr int X = FP8 ? 16 : 4;
  const int head = blockIdx.x, seq = blockIdx.y, part = blockIdx.z;
  const int max_parts = gridDim.z;
  const int seq_len = seq_lens[seq];
  const int start_tok = partition_size ? part * partition_size : 0;
  if (start_tok >= seq_len) return;
  const int end_tok = partition_size ? min(start_tok + partition_size, seq_len) : seq_len;
  const int kv_head = head / (num_heads / num_kv_heads);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  __shared__ float q_s[256];
  __shared__ float p_s[F32_THREADS];
  __shared__ float red[F32_THREADS / WAVE];
  for (int d = tid; d < head_size; d += F32_THREADS) q_s[d] = q[(int64_t)seq * q_stride + (int64_t)head * head_size + d];
  __syncthreads();

  const int* table = block_tables + (int64_t)seq * max_num_blocks_per_seq;
  const float slope = alibi_slopes ? alibi_slopes[head] : 0.f;
  const bool sparse = sp.vert_stride > 1;
  const int sp_off = sp.head_sliding_step >= 0 ? (sp.tp_rank * num_heads + head) * sp.head_sliding_step + 1
                                                : (sp.tp_rank * num_kv_heads + kv_head) * (-sp.head_sliding_step) + 1;
  float m_run = -INFINITY, l_run = 0.f, acc = 0.f;   // acc: output dimension `tid`

  for (int c0 = start_tok; c0 < end_tok; c0 += F32_THREADS) {
    const int tok = c0 + tid;
    const bool valid = tok < end_tok;
    float logit = -INFINITY;
    if (valid) {
      const int64_t base = (int64_t)table[tok / block_size] * kv_block_stride + (int64_t)kv_head * kv_head_stride;
      const int boff = tok % block_size;
      float a = 0.f;
      for (int d = 0; d < head_size; ++d)
        a += q_s[d] * cache_elem<FP8>(k_cache, base + ((int64_t)(d / X) * block_size + boff) * X + d % X, 1.f);
      logit = a * (FP8 ? scale * kv_scale : scale) + (slope != 0.f ? slope * (float)(tok - seq_len + 1) : 0.f);
      if (sparse) {   // attention_kernels.cu:209-251
        const int kb = (tok / block_size) * block_size / sp.block_size;
        const bool attend = kb > (seq_len - 1) / sp.block_size - sp.local_blocks || (kb + sp_off) % sp.vert_stride == 0;
        logit = attend ? logit : -INFINITY;
      }
    }
    // chunk max over the workgroup
    float cm = wave_max(logit);
    if (lane == 0) red[wave] = cm;
    __syncthreads();
    cm = red[0];
    for (int w = 1; w < F32_THREADS / WAVE; ++w) cm = fmaxf(cm, red[w]);
    __syncthreads();
    const float m_new = fmaxf(m_run, cm);
    const float alpha = m_new == -INFINITY ? 0.f : __expf(m_run - m_new);
    const float p = (logit == -INFINITY) ? 0.f : __expf(logit - m_new);
    p_s[tid] = p;
    float ps = wave_sum(p);
    if (lane == 0) red[wave] = ps;
    __syncthreads();
    ps = 0.f;
    for (int w = 0; w < F32_THREADS / WAVE; ++w) ps += red[w];
    l_run = l_run * alpha + ps;
    m_run = m_new;
    // P.V: thread = output dimension
    if (tid < head_size) {
      acc *= alpha;
      const int n = min(F32_THREADS, end_tok - c0);
      for (int t = 0; t < n; ++t) {
        const int tk = c0 + t;
        const int64_t base = (int64_t)table[tk / block_size] * kv_block_stride + (int64_t)kv_head * kv_head_stride;
        acc += p_s[t] * cache_elem<FP8>(v_cache, base + (int64_t)tid * block_size + tk % block_size, 1.f);
      }
    }
    __syncthreads();
  }
  const float inv = __fdividef(1.f, l_run + 1e-6f);   // attention_kernels.cu:342
  const float kvs = FP8 ? kv_scale : 1.f;
  if (partition_size) {
    const int64_t pidx = ((int64_t)seq * num_heads + head) * max_parts + part;
    if (tid < head_size) out[pidx * head_size + tid] = acc * kvs * inv;
    if (tid == 0) {
      exp_sums[pidx] = l_run;
      max_logits[pidx] = m_run;
    }
  } else if (tid < head_size) {
    out[((int64_t)seq * num_heads + head) * head_size + tid] = acc * kvs * inv;
  }
}

// attention_kernels.cu:564-669 for float
__global__ __launch_bounds__(WAVE) void paged_attention_v2_reduce_f32_kernel(
    float* __restrict__ out, const float* __restrict__ exp_sums, const float* __restrict__ max_logits,
    const float* __restrict__ tmp_out, const int* __restrict__ seq_lens, int max_num_partitions, int head_size) {
  const int num_heads = gridDim.x, head = blockIdx.x, seq = blockIdx.y, lane = threadIdx.x;
  const int num_partitions = (seq_lens[seq] + 511) / 512;
  const int64_t base = ((int64_t)seq * num_heads + head) * max_num_partitions;
  float* o = out + ((int64_t)seq * num_heads + head) * head_size;
  const float* t = tmp_out + base * head_size;
  if (num_partitions <= 1) {
    if (num_partitions == 1)
      for (int i = lane; i < head_size; i += WAVE) o[i] = t[i];
    return;
  }
  extern __shared__ float w_s[];
  float mx = -FLT_MAX;
  for (int i = lane; i < num_partitions; i += WAVE) mx = fmaxf(mx, max_logits[base + i]);
  mx = wave_max(mx);
  float gs = 0.f;
  for (int i = lane; i < num_partitions; i += WAVE) {
    const float r = exp_sums[base + i] * expf(max_logits[base + i] - mx);
    w_s[i] = r;
    gs += r;
  }
  gs = wave_sum(gs);
  __syncthreads();
  const float inv = __fdividef(1.f, gs + 1e-6f);
  for (int i = lane; i < head_size; i += WAVE) {
    float a = 0.f;
    for (int j = 0; j < num_partitions; ++j) a += t[(int64_t)j * head_size + i] * w_s[j] * inv;
    o[i] = a;
  }
}

int f32_paged_attention(const F32AttnArgs& a, bool fp8) {
  if (a.head_size > 256 || a.head_size <= 0) {
    set_error("paged_attention (float): Unsupported head size: %d", a.head_size);
    return NMV_ERR_INVALID;
  }
  if (a.head_size % (fp8 ? 16 : 4) != 0) {
    set_error("paged_attention (float): head size %d is not a multiple of x", a.head_size);
    return NMV_ERR_INVALID;
  }
  const int parts = a.partitioned ? (a.max_seq_len + 511) / 512 : 1;
  dim3 grid(a.num_heads, a.num_seqs, parts);
#define NMV_F32_PA(FP8_)                                                                                          \
  hipLaunchKernelGGL((paged_attention_f32_kernel<FP8_>), grid, dim3(F32_THREADS), 0, a.stream, a.exp_sums,         \
                     a.max_logits, a.partitioned ? a.tmp_out : a.out, a.query, a.key_cache, a.value_cache,         \
                     a.num_heads, a.num_kv_heads, a.head_size, a.block_size, a.scale, a.block_tables, a.seq_lens,  \
                     a.max_num_blocks_per_seq, a.alibi_slopes, a.q_stride, a.kv_block_stride, a.kv_head_stride,    \
                     a.kv_scale, a.partitioned ? 512 : 0, a.sparse)
  if (fp8) NMV_F32_PA(true); else NMV_F32_PA(false);
#undef NMV_F32_PA
  if (a.partitioned)
    hipLaunchKernelGGL(paged_attention_v2_reduce_f32_kernel, dim3(a.num_heads, a.num_seqs), dim3(WAVE),
                       parts * sizeof(float), a.stream, a.out, a.exp_sums, a.max_logits, a.tmp_out, a.seq_lens, parts,
                       a.head_size);
  return NMV_OK;
}

// ---- reshape_and_cache (cache_kernels.cu:152-204) for float key / value: one workgroup per token ----
template <bool FP8>
__global__ void reshape_and_cache_f32_kernel(const float* __restrict__ key, const float* __restrict__ value,
                                             void* __restrict__ key_cache, void* __restrict__ value_cache,
                                             const int64_t* __restrict__ slot_mapping, int64_t key_stride,
                                             int64_t value_stride, int num_heads, int head_size, int block_size,
                                             float kv_scale) {
  constexpr int X = FP8 ? 16 : 4;
  const int64_t token = blockIdx.x, slot = slot_mapping[token];
  if (slot < 0) return;   // padding token
  const int64_t blk = slot / block_size, off = slot % block_size;
  for (int i = threadIdx.x; i < num_heads * head_size; i += blockDim.x) {
    const int h = i / head_size, d = i % head_size;
    const int64_t kt = (((blk * num_heads + h) * (head_size / X) + d / X) * block_size + off) * X + d % X;
    const int64_t vt = ((blk * num_heads + h) * head_size + d) * block_size + off;
    const float k = key[token * key_stride + i], v = value[token * value_stride + i];
    if constexpr (FP8) {
      reinterpret_cast<uint8_t*>(key_cache)[kt] = f32_to_fp8(k / kv_scale);
      reinterpret_cast<uint8_t*>(value_cache)[vt] = f32_to_fp8(v / kv_scale);
    } else {
      reinterpret_cast<float*>(key_cache)[kt] = k;
      reinterpret_cast<float*>(value_cache)[vt] = v;
    }
  }
}

int f32_reshape_and_cache(const void* key, const void* value, void* key_cache, void* value_cache,
                          const int64_t* slot_mapping, int num_tokens, int num_kv_heads, int head_size, int block_size,
                          int64_t key_stride, int64_t value_stride, bool fp8, float kv_scale, hipStream_t stream) {
  dim3 grid(num_tokens), block(std::min(num_kv_heads * head_size, 512));
  if (fp8)
    hipLaunchKernelGGL(reshape_and_cache_f32_kernel<true>, grid, block, 0, stream, (const float*)key,
                       (const float*)value, key_cache, value_cache, slot_mapping, key_stride, value_stride, num_kv_heads,
                       head_size, block_size, kv_scale);
  else
    hipLaunchKernelGGL(reshape_and_cache_f32_kernel<false>, grid, block, 0, stream, (const float*)key,
                       (const float*)value, key_cache, value_cache, slot_mapping, key_stride, value_stride, num_kv_heads,
                       head_size, block_size, kv_scale);
  return NMV_OK;
}

// ---- convert_fp8 (cache_kernels.cu:339-389) with a float side ----
template <bool TO_FP8>
__global__ void convert_fp8_f32_kernel(void* __restrict__ dst, const void* __restrict__ src, float scale,
                                       int64_t block_stride) {
  for (int64_t i = threadIdx.x; i < block_stride; i += blockDim.x) {
    const int64_t idx = (int64_t)blockIdx.x * block_stride + i;
    if constexpr (TO_FP8)
      reinterpret_cast<uint8_t*>(dst)[idx] = f32_to_fp8(reinterpret_cast<const float*>(src)[idx] / scale);
    else
      reinterpret_cast<float*>(dst)[idx] = fp8_to_f32(reinterpret_cast<const uint8_t*>(src)[idx]) * scale;
  }
}

int f32_convert_fp8(void* dst, const void* src, int64_t num_blocks, int64_t block_stride, bool to_fp8, float scale,
                    hipStream_t stream) {
  dim3 grid(num_blocks), block((unsigned)std::min<int64_t>(block_stride, 512));
  if (to_fp8)
    hipLaunchKernelGGL(convert_fp8_f32_kernel<true>, grid, block, 0, stream, dst, src, scale, block_stride);
  else
    hipLaunchKernelGGL(convert_fp8_f32_kernel<false>, grid, block, 0, stream, dst, src, scale, block_stride);
  return NMV_OK;
}

}  // namespace nmv
