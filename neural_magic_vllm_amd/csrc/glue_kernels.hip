// Glue ops of a decoder layer: rms_norm, fused_add_rms_norm, rotary_embedding (+batched),
// act_and_mul (silu / gelu / gelu_tanh), element-wise gelu_new / gelu_fast / gelu_quick.
// Behavioural references: /root/reference/csrc/layernorm_kernels.cu:22-44,201-290,
// csrc/pos_encoding_kernels.cu:10-119, csrc/activation_kernels.cu:14-150.
// All of them are HBM-bound streaming kernels: 16-byte vectors per lane, fp32 math inside,
// and the reference's intermediate roundings to the model dtype are reproduced exactly
// (e.g. rms_norm rounds x*rsqrt(var) to the model dtype BEFORE multiplying by the weight).
#include "common.h"
#include "cache_write.h"

// The reference rounds every intermediate to the model dtype (c10::Half / c10::BFloat16
// operators).  With contraction on, hipcc folds the fp16 paths into v_fma_f16 and skips one of
// those roundings -- keep the arithmetic exactly as written.
#pragma clang fp contract(off)

namespace nmv {

// The rounded bits pass through an empty asm so that the compiler has to materialise them: hipcc
// was seen to drop the float -> _Float16 -> float round trip altogether when the rounded value
// never reached memory (the fused norm + int8-quant kernel, fp16 only: a few quantised values per
// 100k moved by one step).
__device__ __forceinline__ uint32_t pin(uint32_t bits) {
  asm volatile("" : "+v"(bits));
  return bits;
}

template <typename T>
__device__ __forceinline__ float rnd(float f) {  // round-trip through the model dtype
  return T::to_float((uint16_t)pin(T::from_float(f)));
}

__device__ __forceinline__ float block_sum_256(float v, float* red /*[4]*/) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  const int nw = (blockDim.x + 63) >> 6;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

__device__ __forceinline__ float block_max_pos(float v, float* red /*[16]*/) {  // v >= 0
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  const int nw = (blockDim.x + 63) >> 6;
  for (int i = 0; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

// dynamic per-token int8 (int8_quant_kernels.cu:37-75): x * (127 / absmax), round to nearest even,
// saturate -- the arithmetic of int8_quant_kernel<T, true> in quant_kernels.hip
__device__ __forceinline__ uint32_t q8(float x, float mul) {
  float d = __builtin_nearbyintf(x * mul);
  d = fminf(fmaxf(d, -128.f), 127.f);
  return (uint32_t)(uint8_t)(int8_t)d;
}

// ---------------------------------------------------------------- RMSNorm
template <typename T, bool FUSED_ADD>
__global__ __launch_bounds__(256) void rms_norm_kernel(uint16_t* out,       // [T, H] (== input when fused: no restrict)
                                                       uint16_t* input,     // [T, H]
                                                       uint16_t* residual,  // [T, H] (fused only)
                                                       const uint16_t* __restrict__ weight,
                                                       float epsilon, int hidden) {
  __shared__ float red[4];
  const int64_t row = (int64_t)blockIdx.x * hidden;
  const bool vec = (hidden % 8 == 0) &&
                   (((reinterpret_cast<uintptr_t>(input) | reinterpret_cast<uintptr_t>(out) |
                      reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(weight)) & 15) == 0);
  float var = 0.f;
  if (vec) {
    const int nv = hidden / 8;
    for (int v = threadIdx.x; v < nv; v += blockDim.x) {
      uint4 x = ld16(input + row + v * 8);
      uint32_t xs[4] = {x.x, x.y, x.z, x.w};
      if constexpr (FUSED_ADD) {
        uint4 r = ld16(residual + row + v * 8);
        uint32_t rs[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // z = x + residual rounded to the model dtype (layernorm_kernels.cu:271-274)
          const float lo = rnd<T>(lo_f<T>(xs[j]) + lo_f<T>(rs[j]));
          const float hi = rnd<T>(hi_f<T>(xs[j]) + hi_f<T>(rs[j]));
          xs[j] = T::pack2(lo, hi);
          var += lo * lo + hi * hi;
        }
        st16(residual + row + v * 8, make_uint4(xs[0], xs[1], xs[2], xs[3]));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = lo_f<T>(xs[j]), hi = hi_f<T>(xs[j]);
          var += lo * lo + hi * hi;
        }
      }
    }
  } else {
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) {
      float x = T::to_float(input[row + i]);
      if constexpr (FUSED_ADD) {
        x = rnd<T>(x + T::to_float(residual[row + i]));
        residual[row + i] = T::from_float(x);
      }
      var += x * x;
    }
  }
  var = block_sum_256(var, red);
  const float s = rsqrtf(var / hidden + epsilon);
  const uint16_t* src = FUSED_ADD ? residual : input;
  if (vec) {
    const int nv = hidden / 8;
    for (int v = threadIdx.x; v < nv; v += blockDim.x) {
      const uint4 x = ld16(src + row + v * 8);
      const uint4 w = ld16(weight + v * 8);
      const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
      const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // ((scalar_t)(x * s_variance)) * weight  (layernorm_kernels.cu:41-42)
        const float lo = rnd<T>(lo_f<T>(xs[j]) * s) * lo_f<T>(ws[j]);
        const float hi = rnd<T>(hi_f<T>(xs[j]) * s) * hi_f<T>(ws[j]);
        o[j] = T::pack2(lo, hi);
      }
      st16(out + row + v * 8, make_uint4(o[0], o[1], o[2], o[3]));
    }
  } else {
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) {
      const float x = T::to_float(src[row + i]);
      out[row + i] = T::from_float(rnd<T>(x * s) * T::to_float(weight[i]));
    }
  }
}

// Register-resident form for 16-byte-aligned rows of up to 256 * 8 * NV elements (the decode
// step: [B, 4096]): the row is read once, the weight vectors are requested together with it, and
// only the block reduction sits between the loads and the stores -- the generic kernel above pays a
// second dependent read of the row, ~1 us of the ~4.6 us such a launch takes.  Per-lane summation
// order is the generic kernel's, so the results are bit-identical.
// QUANT: the normalised row (rounded to the model dtype exactly as the stand-alone op does) is not
// stored but quantised to int8 with a dynamic per-token scale, i.e. rms_norm -> scaled_int8_quant
// (dynamic) in one launch for the W8A8 linears.
// SLABS: the input row is the sum of `splits` fp32 split-K slabs of the preceding GEMM
// (nmv_gptq_marlin_gemm_partial), added in split order from +0 and rounded to the model dtype --
// exactly what the GEMM's own last-arriver pass would have stored.
template <typename T, bool FUSED_ADD, int NV, bool QUANT, bool SLABS = false, int THREADS = 256>
__global__ __launch_bounds__(THREADS) void rms_norm_reg_kernel(void* out_v, uint16_t* input,
                                                           uint16_t* residual,
                                                           const uint16_t* __restrict__ weight,
                                                           float* __restrict__ q_scale,
                                                           float epsilon, int hidden,
                                                           const float* __restrict__ slab = nullptr,
                                                           int splits = 0, int64_t slab_stride = 0) {
  __shared__ float red[16];
  const int64_t row = (int64_t)blockIdx.x * hidden;
  const int nv = hidden / 8;
  uint32_t xs[NV][4];
  uint4 w[NV];
  float var = 0.f;
  float terms[4] = {0.f, 0.f, 0.f, 0.f};  // THREADS > 256 (NV == 1): the four addends of this lane's vector
  static_assert(THREADS == 256 || (NV == 1 && FUSED_ADD), "the wide form holds one vector per lane");
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = threadIdx.x + i * THREADS;
    const bool ok = v < nv;
    const int vc = ok ? v : nv - 1;
    w[i] = ld16(weight + vc * 8);
    if constexpr (SLABS) {
      f32x4_t lo4 = {0.f, 0.f, 0.f, 0.f}, hi4 = {0.f, 0.f, 0.f, 0.f};
      if (splits < 0) {   // uniform: -splits slabs in the MODEL dtype (the prompt-sized GEMM's mode 3), summed in fp32
        const uint16_t* src = reinterpret_cast<const uint16_t*>(slab) + row + vc * 8;
#pragma unroll 4
        for (int sp = 0; sp < -splits; ++sp) {
          const uint4 x = ld16(src + sp * slab_stride);
          lo4[0] += lo_f<T>(x.x), lo4[1] += hi_f<T>(x.x), lo4[2] += lo_f<T>(x.y), lo4[3] += hi_f<T>(x.y);
          hi4[0] += lo_f<T>(x.z), hi4[1] += hi_f<T>(x.z), hi4[2] += lo_f<T>(x.w), hi4[3] += hi_f<T>(x.w);
        }
      } else {
        const float* src = slab + row + vc * 8;
#pragma unroll 4
        for (int sp = 0; sp < splits; ++sp) {
          lo4 += *reinterpret_cast<const f32x4_t*>(src + sp * slab_stride);
          hi4 += *reinterpret_cast<const f32x4_t*>(src + sp * slab_stride + 4);
        }
      }
      xs[i][0] = T::pack2(lo4[0], lo4[1]), xs[i][1] = T::pack2(lo4[2], lo4[3]);
      xs[i][2] = T::pack2(hi4[0], hi4[1]), xs[i][3] = T::pack2(hi4[2], hi4[3]);
    } else {
      const uint4 x = ld16(input + row + vc * 8);
      xs[i][0] = x.x, xs[i][1] = x.y, xs[i][2] = x.z, xs[i][3] = x.w;
    }
    if constexpr (FUSED_ADD) {
      const uint4 r = ld16(residual + row + vc * 8);
      const uint32_t rs[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // z = x + residual rounded to the model dtype (layernorm_kernels.cu:271-274)
        const float lo = rnd<T>(lo_f<T>(xs[i][j]) + lo_f<T>(rs[j]));
        const float hi = rnd<T>(hi_f<T>(xs[i][j]) + hi_f<T>(rs[j]));
        xs[i][j] = T::pack2(lo, hi);
        if (ok) var += lo * lo + hi * hi;
        if constexpr (THREADS > 256) terms[j] = ok ? lo * lo + hi * hi : 0.f;
      }
      if (ok) st16(residual + row + v * 8, make_uint4(xs[i][0], xs[i][1], xs[i][2], xs[i][3]));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = lo_f<T>(xs[i][j]), hi = hi_f<T>(xs[i][j]);
        if (ok) var += lo * lo + hi * hi;
      }
    }
  }
  if constexpr (THREADS > 256) {
    // more lanes than the 256-lane form: replay that form's per-lane accumulation (the four addends
    // of vector t, then of t + 256, ...) on lanes 0..255 so that the variance is bit-identical to it
    __shared__ float fold[THREADS][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fold[threadIdx.x][j] = terms[j];
    __syncthreads();
    var = 0.f;
    if (threadIdx.x < 256) {
#pragma unroll
      for (int q = 0; q < THREADS / 256; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) var += fold[threadIdx.x + q * 256][j];
    }
  }
  var = block_sum_256(var, red);
  const float s = rsqrtf(var / hidden + epsilon);
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = threadIdx.x + i * THREADS;
    const uint32_t ws[4] = {w[i].x, w[i].y, w[i].z, w[i].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // ((scalar_t)(x * s_variance)) * weight  (layernorm_kernels.cu:41-42)
      const float lo = rnd<T>(lo_f<T>(xs[i][j]) * s) * lo_f<T>(ws[j]);
      const float hi = rnd<T>(hi_f<T>(xs[i][j]) * s) * hi_f<T>(ws[j]);
      xs[i][j] = T::pack2(lo, hi);
      if constexpr (QUANT) xs[i][j] = pin(xs[i][j]);
      if constexpr (QUANT)
        if (v < nv) amax = fmaxf(amax, fmaxf(fabsf(lo_f<T>(xs[i][j])), fabsf(hi_f<T>(xs[i][j]))));
    }
    if constexpr (!QUANT)
      if (v < nv)
        st16(reinterpret_cast<uint16_t*>(out_v) + row + v * 8,
             make_uint4(xs[i][0], xs[i][1], xs[i][2], xs[i][3]));
  }
  if constexpr (QUANT) {
    amax = block_max_pos(amax, red);
    if (threadIdx.x == 0) q_scale[blockIdx.x] = amax / 127.0f;
    const float mul = 127.0f / amax;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = threadIdx.x + i * THREADS;
      uint2 o;
      o.x = q8(lo_f<T>(xs[i][0]), mul) | (q8(hi_f<T>(xs[i][0]), mul) << 8) |
            (q8(lo_f<T>(xs[i][1]), mul) << 16) | (q8(hi_f<T>(xs[i][1]), mul) << 24);
      o.y = q8(lo_f<T>(xs[i][2]), mul) | (q8(hi_f<T>(xs[i][2]), mul) << 8) |
            (q8(lo_f<T>(xs[i][3]), mul) << 16) | (q8(hi_f<T>(xs[i][3]), mul) << 24);
      if (v < nv) *reinterpret_cast<uint2*>(reinterpret_cast<int8_t*>(out_v) + row + v * 8) = o;
    }
  }
}

// ---------------------------------------------------------------- rotary embedding
// One workgroup per token.  Every intermediate is rounded to the model dtype exactly as the
// reference's scalar_t arithmetic does (pos_encoding_kernels.cu:10-37: x*cos - y*sin with
// c10::BFloat16/Half operators).
template <typename T, bool IS_NEOX>
__device__ __forceinline__ void rope_one(uint16_t* __restrict__ arr, const uint16_t* __restrict__ cos_ptr,
                                         const uint16_t* __restrict__ sin_ptr, int rot_offset,
                                         int embed_dim) {
  int x_index, y_index;
  float c, s;
  if constexpr (IS_NEOX) {
    x_index = rot_offset;
    y_index = embed_dim + rot_offset;
    c = T::to_float(cos_ptr[x_index]);
    s = T::to_float(sin_ptr[x_index]);
  } else {
    x_index = 2 * rot_offset;
    y_index = 2 * rot_offset + 1;
    c = T::to_float(cos_ptr[x_index / 2]);
    s = T::to_float(sin_ptr[x_index / 2]);
  }
  const float x = T::to_float(arr[x_index]);
  const float y = T::to_float(arr[y_index]);
  arr[x_index] = T::from_float(rnd<T>(x * c) - rnd<T>(y * s));
  arr[y_index] = T::from_float(rnd<T>(y * c) + rnd<T>(x * s));
}

template <typename T, bool IS_NEOX>
__global__ void rotary_embedding_kernel(const int64_t* __restrict__ positions,
                                        uint16_t* __restrict__ query, uint16_t* __restrict__ key,
                                        const uint16_t* __restrict__ cos_sin_cache,
                                        const int64_t* __restrict__ cache_offsets, int rot_dim,
                                        int64_t query_stride, int64_t key_stride, int num_heads,
                                        int num_kv_heads, int head_size) {
  const int token_idx = blockIdx.x;
  int64_t pos = positions[token_idx];
  if (cache_offsets) pos += cache_offsets[token_idx];
  const uint16_t* cache_ptr = cos_sin_cache + pos * rot_dim;
  const int embed_dim = rot_dim / 2;
  const uint16_t* cos_ptr = cache_ptr;
  const uint16_t* sin_ptr = cache_ptr + embed_dim;
  const int nq = num_heads * embed_dim;
  for (int i = threadIdx.x; i < nq; i += blockDim.x) {
    const int head_idx = i / embed_dim;
    const int rot_offset = i % embed_dim;
    rope_one<T, IS_NEOX>(query + token_idx * query_stride + (int64_t)head_idx * head_size, cos_ptr,
                         sin_ptr, rot_offset, embed_dim);
  }
  const int nk = num_kv_heads * embed_dim;
  for (int i = threadIdx.x; i < nk; i += blockDim.x) {
    const int head_idx = i / embed_dim;
    const int rot_offset = i % embed_dim;
    rope_one<T, IS_NEOX>(key + token_idx * key_stride + (int64_t)head_idx * head_size, cos_ptr,
                         sin_ptr, rot_offset, embed_dim);
  }
}

// rotary_embedding + reshape_and_cache in one launch (the decode step calls them back to back on
// [B, heads*head_size] tensors of a few hundred KB: each launch is ~5 us of latency, not bandwidth).
// Same arithmetic and the same in-place update of query / key as the two separate ops.  The lane
// that rotates a key pair also stores the two rotated elements (the rounded bits it writes back to
// `key`) at their place in the paged cache, so there is no barrier and no second read: one
// load -> rotate -> store chain per lane, as in rotary_embedding alone.
template <typename T, bool IS_NEOX, bool FP8>
__global__ void rope_and_cache_kernel(const int64_t* __restrict__ positions, uint16_t* query,
                                      uint16_t* key, const uint16_t* value,
                                      const uint16_t* __restrict__ cos_sin_cache, int rot_dim,
                                      int64_t query_stride, int64_t key_stride,
                                      int64_t value_stride, int num_heads, int num_kv_heads,
                                      int head_size, void* key_cache, void* value_cache,
                                      const int64_t* __restrict__ slot_mapping, int block_size,
                                      float kv_scale) {
  const int token_idx = blockIdx.x;
  // the pairs / elements of a token are independent: spread over gridDim.y workgroups
  const int tid0 = blockIdx.y * blockDim.x + threadIdx.x, tstride = blockDim.x * gridDim.y;
  const int64_t pos = positions[token_idx];
  const int64_t slot_idx = slot_mapping[token_idx];
  const bool cached = slot_idx >= 0;  // padding tokens are rotated but not cached (cache_kernels.cu:166-169)
  const int64_t block_idx = cached ? slot_idx / block_size : 0;
  const int64_t block_offset = cached ? slot_idx % block_size : 0;
  const uint16_t* cache_ptr = cos_sin_cache + pos * rot_dim;
  const int embed_dim = rot_dim / 2;
  const uint16_t* cos_ptr = cache_ptr;
  const uint16_t* sin_ptr = cache_ptr + embed_dim;
  // ---- K: rotate, write back, store in the cache ----
  const int nk = num_kv_heads * embed_dim;
  for (int i = tid0; i < nk; i += tstride) {
    const int head_idx = i / embed_dim;
    const int rot_offset = i % embed_dim;
    uint16_t* arr = key + token_idx * key_stride + (int64_t)head_idx * head_size;
    const int x_index = IS_NEOX ? rot_offset : 2 * rot_offset;
    const int y_index = IS_NEOX ? embed_dim + rot_offset : 2 * rot_offset + 1;
    const float c = T::to_float(cos_ptr[rot_offset]);
    const float s = T::to_float(sin_ptr[rot_offset]);
    const float x = T::to_float(arr[x_index]);
    const float y = T::to_float(arr[y_index]);
    const uint16_t xo = T::from_float(rnd<T>(x * c) - rnd<T>(y * s));
    const uint16_t yo = T::from_float(rnd<T>(y * c) + rnd<T>(x * s));
    arr[x_index] = xo;
    arr[y_index] = yo;
    if (cached) {
      const int64_t hb = block_idx * num_kv_heads + head_idx;
      cache_store_k<T, FP8>(key_cache, hb, head_size, block_size, block_offset, x_index, xo, kv_scale);
      cache_store_k<T, FP8>(key_cache, hb, head_size, block_size, block_offset, y_index, yo, kv_scale);
    }
  }
  if (cached) {
    // K dims beyond rot_dim pass through unrotated
    const int pass = head_size - rot_dim;
    for (int i = tid0; i < num_kv_heads * pass; i += tstride) {
      const int head_idx = i / pass, d = rot_dim + i % pass;
      cache_store_k<T, FP8>(key_cache, block_idx * num_kv_heads + head_idx, head_size, block_size,
                            block_offset, d, key[token_idx * key_stride + (int64_t)head_idx * head_size + d],
                            kv_scale);
    }
    // ---- V: element scatter, as write_token_to_cache (cache_write.h) ----
    const int n = num_kv_heads * head_size;
    for (int i = tid0; i < n; i += tstride) {
      const int head_idx = i / head_size;
      const int head_off = i % head_size;
      const int64_t tgt =
          ((block_idx * num_kv_heads + head_idx) * head_size + head_off) * block_size + block_offset;
      const uint16_t v = value[token_idx * value_stride + i];
      if constexpr (!FP8)
        reinterpret_cast<uint16_t*>(value_cache)[tgt] = v;
      else
        reinterpret_cast<uint8_t*>(value_cache)[tgt] = f32_to_fp8(T::to_float(v) / kv_scale);
    }
  }
  // ---- Q ----
  const int nq = num_heads * embed_dim;
  for (int i = tid0; i < nq; i += tstride) {
    const int head_idx = i / embed_dim;
    const int rot_offset = i % embed_dim;
    rope_one<T, IS_NEOX>(query + token_idx * query_stride + (int64_t)head_idx * head_size, cos_ptr,
                         sin_ptr, rot_offset, embed_dim);
  }
}

// rope_and_cache_kernel whose q / k / v come from the fp32 split-K slabs of the qkv projection
// (nmv_gptq_marlin_gemm_partial): each element is the slab sum in split order from +0, rounded to
// the model dtype -- what the GEMM's own last-arriver pass would have stored -- then exactly the
// arithmetic above.  A lane owns 4 consecutive head dims (16-byte slab reads): for neox pairs
// (d, d + rot/2) that is 4 pairs of a q or k head, or 4 v elements.  The rounded qkv row is also
// written out ([T, (heads + 2 kv_heads) * head_size]) for the attention kernels.  neox style,
// rot_dim == head_size.
template <typename T, bool FP8>
__global__ __launch_bounds__(256) void rope_and_cache_slab_kernel(
    const int64_t* __restrict__ positions, const float* __restrict__ slab, int splits,
    int64_t slab_stride, uint16_t* __restrict__ qkv_out, const uint16_t* __restrict__ cos_sin_cache,
    int num_heads, int num_kv_heads, int head_size, void* key_cache, void* value_cache,
    const int64_t* __restrict__ slot_mapping, int block_size, float kv_scale) {
  const int token_idx = blockIdx.x;
  const int64_t pos = positions[token_idx];
  const int64_t slot_idx = (key_cache != nullptr) ? slot_mapping[token_idx] : -1;
  const bool cached = slot_idx >= 0;
  const int64_t block_idx = cached ? slot_idx / block_size : 0;
  const int64_t block_offset = cached ? slot_idx % block_size : 0;
  const int embed_dim = head_size / 2;
  const uint16_t* cos_ptr = cos_sin_cache + pos * head_size;
  const uint16_t* sin_ptr = cos_ptr + embed_dim;
  const int row_elems = (num_heads + 2 * num_kv_heads) * head_size;
  const float* src_row = slab + (int64_t)token_idx * row_elems;
  uint16_t* dst_row = qkv_out + (int64_t)token_idx * row_elems;
  auto slab_sum4 = [&](int col, float (&o)[4]) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    if (splits < 0) {   // uniform: -splits slabs in the MODEL dtype (the prompt-sized GEMM's mode 3)
      const uint16_t* src16 = reinterpret_cast<const uint16_t*>(slab) + (int64_t)token_idx * row_elems + col;
#pragma unroll 4
      for (int sp = 0; sp < -splits; ++sp) {
        const uint2 x = *reinterpret_cast<const uint2*>(src16 + sp * slab_stride);
        acc[0] += lo_f<T>(x.x), acc[1] += hi_f<T>(x.x), acc[2] += lo_f<T>(x.y), acc[3] += hi_f<T>(x.y);
      }
    } else {
#pragma unroll 4
      for (int sp = 0; sp < splits; ++sp)
        acc += *reinterpret_cast<const f32x4_t*>(src_row + sp * slab_stride + col);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = rnd<T>(acc[i]);  // the GEMM's output rounding
  };
  const int quads = embed_dim / 4;                 // items per rotated head
  const int n_rot = (num_heads + num_kv_heads) * quads;
  const int n_v = num_kv_heads * head_size / 4;
  // the items of a token are independent: they are spread over gridDim.y workgroups, one item per
  // lane, so that the slab reads of a row are all in flight at once (at B = 1 a single workgroup
  // per token would be the whole grid)
  const int it = blockIdx.y * blockDim.x + threadIdx.x;
  if (it < n_rot + n_v) {
    if (it < n_rot) {
      const int head = it / quads;                 // q heads first, then k heads (qkv column order)
      const int d0 = (it % quads) * 4;
      const int col = head * head_size + d0;
      float x[4], y[4];
      slab_sum4(col, x);
      slab_sum4(col + embed_dim, y);
      const uint2 cw = *reinterpret_cast<const uint2*>(cos_ptr + d0);
      const uint2 sw = *reinterpret_cast<const uint2*>(sin_ptr + d0);
      const uint32_t cs[2] = {cw.x, cw.y}, sn[2] = {sw.x, sw.y};
      uint16_t xo[4], yo[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float c = (i & 1) ? hi_f<T>(cs[i >> 1]) : lo_f<T>(cs[i >> 1]);
        const float s = (i & 1) ? hi_f<T>(sn[i >> 1]) : lo_f<T>(sn[i >> 1]);
        xo[i] = T::from_float(rnd<T>(x[i] * c) - rnd<T>(y[i] * s));
        yo[i] = T::from_float(rnd<T>(y[i] * c) + rnd<T>(x[i] * s));
      }
      uint2 px, py;
      px.x = xo[0] | ((uint32_t)xo[1] << 16), px.y = xo[2] | ((uint32_t)xo[3] << 16);
      py.x = yo[0] | ((uint32_t)yo[1] << 16), py.y = yo[2] | ((uint32_t)yo[3] << 16);
      *reinterpret_cast<uint2*>(dst_row + col) = px;
      *reinterpret_cast<uint2*>(dst_row + col + embed_dim) = py;
      if (cached && head >= num_heads) {
        const int64_t hb = block_idx * num_kv_heads + (head - num_heads);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          cache_store_k<T, FP8>(key_cache, hb, head_size, block_size, block_offset, d0 + i, xo[i], kv_scale);
          cache_store_k<T, FP8>(key_cache, hb, head_size, block_size, block_offset, d0 + embed_dim + i, yo[i], kv_scale);
        }
      }
    } else {
      const int e0 = (it - n_rot) * 4;             // element of the v part
      const int col = (num_heads + num_kv_heads) * head_size + e0;
      float v[4];
      slab_sum4(col, v);
      uint16_t vb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) vb[i] = T::from_float(v[i]);
      uint2 pv;
      pv.x = vb[0] | ((uint32_t)vb[1] << 16), pv.y = vb[2] | ((uint32_t)vb[3] << 16);
      *reinterpret_cast<uint2*>(dst_row + col) = pv;
      if (cached) {
        const int head_idx = e0 / head_size, head_off = e0 % head_size;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t tgt =
              ((block_idx * num_kv_heads + head_idx) * head_size + head_off + i) * block_size + block_offset;
          if constexpr (!FP8)
            reinterpret_cast<uint16_t*>(value_cache)[tgt] = vb[i];
          else
            reinterpret_cast<uint8_t*>(value_cache)[tgt] = f32_to_fp8(T::to_float(vb[i]) / kv_scale);
        }
      }
    }
  }
}

// ---------------------------------------------------------------- activations
template <typename T, int ACT>
__device__ __forceinline__ float gate_act(float f) {
  if constexpr (ACT == 0) {  // silu
    return rnd<T>(f / (1.0f + expf(-f)));
  } else if constexpr (ACT == 1) {  // gelu (erf)
    return rnd<T>(f * 0.5f * (1.0f + erff(f * 0.70710678118654752440f)));
  } else {  // gelu tanh
    constexpr float BETA = 1.41421356237309504880f * 1.12837916709551257390f * 0.5f;
    constexpr float KAPPA = 0.044715f;
    const float inner = BETA * (f + KAPPA * (f * f * f));
    return rnd<T>(0.5f * f * (1.0f + tanhf(inner)));
  }
}

template <typename T, int ACT>
__global__ void act_and_mul_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ input,
                                   int d) {
  const int64_t token_idx = blockIdx.x;
  const uint16_t* xr = input + token_idx * 2 * d;
  const uint16_t* yr = xr + d;
  uint16_t* o = out + token_idx * d;
  const bool vec = (d % 8 == 0) &&
                   (((reinterpret_cast<uintptr_t>(input) | reinterpret_cast<uintptr_t>(out)) & 15) == 0);
  if (vec) {
    for (int v = threadIdx.x; v < d / 8; v += blockDim.x) {
      const uint4 x = ld16(xr + v * 8), y = ld16(yr + v * 8);
      const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
      uint32_t r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        r[j] = T::pack2(gate_act<T, ACT>(lo_f<T>(xs[j])) * lo_f<T>(ys[j]),
                        gate_act<T, ACT>(hi_f<T>(xs[j])) * hi_f<T>(ys[j]));
      st16(o + v * 8, make_uint4(r[0], r[1], r[2], r[3]));
    }
  } else {
    for (int i = threadIdx.x; i < d; i += blockDim.x)
      o[i] = T::from_float(gate_act<T, ACT>(T::to_float(xr[i])) * T::to_float(yr[i]));
  }
}

template <typename T, int ACT>
__device__ __forceinline__ float unary_act(float x) {
  if constexpr (ACT == 0) {  // gelu_new (activation_kernels.cu:120-125), model-dtype intermediates
    const float x3 = rnd<T>(rnd<T>(x * x) * x);
    const float inner = rnd<T>(x + rnd<T>(0.044715f * x3));
    const float t = rnd<T>(tanhf(rnd<T>(0.79788456f * inner)));
    return rnd<T>(rnd<T>(0.5f * x) * rnd<T>(1.0f + t));
  } else if constexpr (ACT == 1) {  // gelu_fast (:127-133)
    const float a = rnd<T>(x * 0.79788456f);
    const float b = rnd<T>(1.0f + rnd<T>(rnd<T>(0.044715f * x) * x));
    const float t = rnd<T>(tanhf(rnd<T>(a * b)));
    return rnd<T>(rnd<T>(0.5f * x) * rnd<T>(1.0f + t));
  } else {  // gelu_quick (:135-139)
    return rnd<T>(x / (1.0f + expf(-1.702f * x)));
  }
}

// silu_and_mul -> scaled_int8_quant (dynamic per token) in one launch: the products, rounded to
// the model dtype as act_and_mul_kernel stores them, stay in registers (NV 16-byte vectors per
// lane, 1024 lanes) while the row maximum is reduced.
template <typename T, int ACT, int NV>
__global__ __launch_bounds__(1024) void act_and_mul_quant_kernel(int8_t* __restrict__ out,
                                                                 float* __restrict__ q_scale,
                                                                 const uint16_t* __restrict__ input,
                                                                 int d) {
  __shared__ float red[16];
  const int64_t token_idx = blockIdx.x;
  const uint16_t* xr = input + token_idx * 2 * d;
  const uint16_t* yr = xr + d;
  const int nv = d / 8;
  uint32_t r[NV][4];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = threadIdx.x + i * 1024;
    const int vc = v < nv ? v : nv - 1;
    const uint4 x = ld16(xr + vc * 8), y = ld16(yr + vc * 8);
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r[i][j] = pin(T::pack2(gate_act<T, ACT>(lo_f<T>(xs[j])) * lo_f<T>(ys[j]),
                             gate_act<T, ACT>(hi_f<T>(xs[j])) * hi_f<T>(ys[j])));
      if (v < nv) amax = fmaxf(amax, fmaxf(fabsf(lo_f<T>(r[i][j])), fabsf(hi_f<T>(r[i][j]))));
    }
  }
  amax = block_max_pos(amax, red);
  if (threadIdx.x == 0) q_scale[token_idx] = amax / 127.0f;
  const float mul = 127.0f / amax;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = threadIdx.x + i * 1024;
    uint2 o;
    o.x = q8(lo_f<T>(r[i][0]), mul) | (q8(hi_f<T>(r[i][0]), mul) << 8) |
          (q8(lo_f<T>(r[i][1]), mul) << 16) | (q8(hi_f<T>(r[i][1]), mul) << 24);
    o.y = q8(lo_f<T>(r[i][2]), mul) | (q8(hi_f<T>(r[i][2]), mul) << 8) |
          (q8(lo_f<T>(r[i][3]), mul) << 16) | (q8(hi_f<T>(r[i][3]), mul) << 24);
    if (v < nv) *reinterpret_cast<uint2*>(out + token_idx * d + v * 8) = o;
  }
}

template <typename T, int ACT>
__global__ void activation_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ input,
                                  int d) {
  const int64_t token_idx = blockIdx.x;
  for (int i = threadIdx.x; i < d; i += blockDim.x)
    out[token_idx * d + i] = T::from_float(unary_act<T, ACT>(T::to_float(input[token_idx * d + i])));
}

}  // namespace nmv

using namespace nmv;

#define NMV_HALF_ONLY(name)                                                              \
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16, name ": unsupported dtype %d", (int)dtype)

// kind: 0 rms_norm, 1 fused_add_rms_norm; q_out/q_scale non-null: int8 output with per-token scales
template <typename T>
static void rms_launch(void* out, uint16_t* input, uint16_t* residual, const uint16_t* weight,
                       float* q_scale, float epsilon, int num_tokens, int hidden, bool quant,
                       hipStream_t s) {
  dim3 grid(num_tokens), block(256);
  const bool fused = residual != nullptr;
  const bool aligned = hidden % 8 == 0 &&
                       (((uintptr_t)input | (uintptr_t)out | (uintptr_t)residual | (uintptr_t)weight) & 15) == 0;
  const int nv = aligned ? (hidden / 8 + 255) / 256 : 0;
#define RMS_REG(NV_)                                                                               \
  {                                                                                                \
    if (fused) {                                                                                   \
      if (quant) hipLaunchKernelGGL((rms_norm_reg_kernel<T, true, NV_, true>), grid, block, 0, s, out, input, residual, weight, q_scale, epsilon, hidden); \
      else hipLaunchKernelGGL((rms_norm_reg_kernel<T, true, NV_, false>), grid, block, 0, s, out, input, residual, weight, q_scale, epsilon, hidden); \
    } else {                                                                                       \
      if (quant) hipLaunchKernelGGL((rms_norm_reg_kernel<T, false, NV_, true>), grid, block, 0, s, out, input, residual, weight, q_scale, epsilon, hidden); \
      else hipLaunchKernelGGL((rms_norm_reg_kernel<T, false, NV_, false>), grid, block, 0, s, out, input, residual, weight, q_scale, epsilon, hidden); \
    }                                                                                              \
  }
  if (nv == 1) RMS_REG(1)
  else if (nv == 2) RMS_REG(2)
  else if (nv >= 3 && nv <= 4) RMS_REG(4)
  else if (fused)
    hipLaunchKernelGGL((rms_norm_kernel<T, true>), grid, block, 0, s, (uint16_t*)out, input, residual, weight, epsilon, hidden);
  else
    hipLaunchKernelGGL((rms_norm_kernel<T, false>), grid, block, 0, s, (uint16_t*)out, input, residual, weight, epsilon, hidden);
#undef RMS_REG
}

extern "C" int nmv_rms_norm(void* out, const void* input, const void* weight, float epsilon,
                            int num_tokens, int hidden_size, nmv_dtype_t dtype, void* stream) {
  NMV_HALF_ONLY("rms_norm");
  if (num_tokens == 0) return NMV_OK;
  if (dtype == NMV_F16)
    rms_launch<F16>(out, (uint16_t*)input, nullptr, (const uint16_t*)weight, nullptr, epsilon,
                    num_tokens, hidden_size, false, (hipStream_t)stream);
  else
    rms_launch<BF16>(out, (uint16_t*)input, nullptr, (const uint16_t*)weight, nullptr, epsilon,
                     num_tokens, hidden_size, false, (hipStream_t)stream);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_fused_add_rms_norm(void* input, void* residual, const void* weight,
                                      float epsilon, int num_tokens, int hidden_size,
                                      nmv_dtype_t dtype, void* stream) {
  NMV_HALF_ONLY("fused_add_rms_norm");
  NMV_CHECK(residual != nullptr, "fused_add_rms_norm: null residual");
  if (num_tokens == 0) return NMV_OK;
  if (dtype == NMV_F16)
    rms_launch<F16>(input, (uint16_t*)input, (uint16_t*)residual, (const uint16_t*)weight, nullptr,
                    epsilon, num_tokens, hidden_size, false, (hipStream_t)stream);
  else
    rms_launch<BF16>(input, (uint16_t*)input, (uint16_t*)residual, (const uint16_t*)weight, nullptr,
                     epsilon, num_tokens, hidden_size, false, (hipStream_t)stream);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_rms_norm_dynamic_int8_quant(void* out_q, float* scales, const void* input,
                                               void* residual, const void* weight, float epsilon,
                                               int num_tokens, int hidden_size, nmv_dtype_t dtype,
                                               void* stream) {
  NMV_HALF_ONLY("rms_norm_dynamic_int8_quant");
  NMV_CHECK(hidden_size % 8 == 0 && hidden_size <= 8192 &&
                (((uintptr_t)input | (uintptr_t)out_q | (uintptr_t)residual | (uintptr_t)weight) & 15) == 0,
            "rms_norm_dynamic_int8_quant: hidden_size must be a multiple of 8, <= 8192, 16-byte aligned rows");
  if (num_tokens == 0) return NMV_OK;
  if (dtype == NMV_F16)
    rms_launch<F16>(out_q, (uint16_t*)input, (uint16_t*)residual, (const uint16_t*)weight, scales,
                    epsilon, num_tokens, hidden_size, true, (hipStream_t)stream);
  else
    rms_launch<BF16>(out_q, (uint16_t*)input, (uint16_t*)residual, (const uint16_t*)weight, scales,
                     epsilon, num_tokens, hidden_size, true, (hipStream_t)stream);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

static int fused_add_rms_norm_partial_impl(void* out, const float* slab, int splits, bool slab16, void* residual,
                                           const void* weight, float epsilon, int num_tokens,
                                           int hidden_size, nmv_dtype_t dtype, void* stream) {
  NMV_HALF_ONLY("fused_add_rms_norm_partial");
  NMV_CHECK(splits >= 1 && residual != nullptr && slab != nullptr, "fused_add_rms_norm_partial: bad arguments");
  NMV_CHECK(hidden_size % 8 == 0 && hidden_size <= 8192 &&
                (((uintptr_t)slab | (uintptr_t)out | (uintptr_t)residual | (uintptr_t)weight) & 15) == 0,
            "fused_add_rms_norm_partial: hidden_size must be a multiple of 8, <= 8192, 16-byte aligned rows");
  if (num_tokens == 0) return NMV_OK;
  // one 16-byte vector (8 elements = 2 x 16 bytes of every slab) per lane: the slab reads are the
  // work here, so the row gets as many waves as it has vectors (up to 16) to keep them in flight
  dim3 grid(num_tokens);
  hipStream_t s = (hipStream_t)stream;
  const int vecs = hidden_size / 8;
  const int64_t stride = (int64_t)num_tokens * hidden_size;
#define RMS_SLAB(T, TH_)                                                                               \
  hipLaunchKernelGGL((rms_norm_reg_kernel<T, true, 1, false, true, TH_>), grid, dim3(TH_), 0, s, out,  \
                     (uint16_t*)nullptr, (uint16_t*)residual, (const uint16_t*)weight,                 \
                     (float*)nullptr, epsilon, hidden_size, slab, slab16 ? -splits : splits, stride)
  if (dtype == NMV_F16) {
    if (vecs <= 256) RMS_SLAB(F16, 256); else if (vecs <= 512) RMS_SLAB(F16, 512); else RMS_SLAB(F16, 1024);
  } else {
    if (vecs <= 256) RMS_SLAB(BF16, 256); else if (vecs <= 512) RMS_SLAB(BF16, 512); else RMS_SLAB(BF16, 1024);
  }
#undef RMS_SLAB
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_fused_add_rms_norm_partial(void* out, const float* slab, int splits, void* residual,
                                              const void* weight, float epsilon, int num_tokens,
                                              int hidden_size, nmv_dtype_t dtype, void* stream) {
  return fused_add_rms_norm_partial_impl(out, slab, splits, false, residual, weight, epsilon, num_tokens, hidden_size, dtype,
                                         stream);
}

// the same with the slabs in the model dtype, [splits][num_tokens][hidden_size] of 2-byte elements (mode 3 of
// nmv_w4_native_gemm): summed in fp32 in split order, rounded once
extern "C" int nmv_fused_add_rms_norm_partial16(void* out, const void* slab, int splits, void* residual,
                                                const void* weight, float epsilon, int num_tokens,
                                                int hidden_size, nmv_dtype_t dtype, void* stream) {
  return fused_add_rms_norm_partial_impl(out, (const float*)slab, splits, true, residual, weight, epsilon, num_tokens,
                                         hidden_size, dtype, stream);
}

extern "C" int nmv_silu_and_mul_dynamic_int8_quant(void* out_q, float* scales, const void* input,
                                                   int num_tokens, int d, nmv_dtype_t dtype,
                                                   void* stream) {
  NMV_HALF_ONLY("silu_and_mul_dynamic_int8_quant");
  NMV_CHECK(d % 8 == 0 && d <= 32768 && (((uintptr_t)input | (uintptr_t)out_q) & 15) == 0,
            "silu_and_mul_dynamic_int8_quant: d must be a multiple of 8, <= 32768, 16-byte aligned rows");
  if (num_tokens == 0 || d == 0) return NMV_OK;
  dim3 grid(num_tokens), block(1024);
  hipStream_t s = (hipStream_t)stream;
  const int nv = (d / 8 + 1023) / 1024;
#define LAUNCH_AQ(T, NV_)                                                                   \
  hipLaunchKernelGGL((act_and_mul_quant_kernel<T, 0, NV_>), grid, block, 0, s, (int8_t*)out_q, \
                     scales, (const uint16_t*)input, d)
  if (dtype == NMV_F16) {
    if (nv == 1) LAUNCH_AQ(F16, 1); else if (nv == 2) LAUNCH_AQ(F16, 2); else LAUNCH_AQ(F16, 4);
  } else {
    if (nv == 1) LAUNCH_AQ(BF16, 1); else if (nv == 2) LAUNCH_AQ(BF16, 2); else LAUNCH_AQ(BF16, 4);
  }
#undef LAUNCH_AQ
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

static int rope_launch(const int64_t* positions, void* query, void* key, int num_tokens,
                       int num_heads, int num_kv_heads, int head_size, int rot_dim,
                       int64_t query_stride, int64_t key_stride, const void* cos_sin_cache,
                       int is_neox, const int64_t* offsets, nmv_dtype_t dtype, void* stream) {
  NMV_HALF_ONLY("rotary_embedding");
  NMV_CHECK(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size,
            "rotary_embedding: bad rot_dim %d for head_size %d", rot_dim, head_size);
  if (num_tokens == 0) return NMV_OK;
  dim3 grid(num_tokens), block(std::min(num_heads * rot_dim / 2, 512));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH_ROPE(T, NEOX)                                                                    \
  hipLaunchKernelGGL((rotary_embedding_kernel<T, NEOX>), grid, block, 0, s, positions,          \
                     (uint16_t*)query, (uint16_t*)key, (const uint16_t*)cos_sin_cache, offsets, \
                     rot_dim, query_stride, key_stride, num_heads, num_kv_heads, head_size)
  if (dtype == NMV_F16) {
    if (is_neox) LAUNCH_ROPE(F16, true); else LAUNCH_ROPE(F16, false);
  } else {
    if (is_neox) LAUNCH_ROPE(BF16, true); else LAUNCH_ROPE(BF16, false);
  }
#undef LAUNCH_ROPE
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_rotary_embedding(const int64_t* positions, void* query, void* key,
                                    int num_tokens, int num_heads, int num_kv_heads,
                                    int head_size, int rot_dim, int64_t query_stride,
                                    int64_t key_stride, const void* cos_sin_cache, int is_neox,
                                    nmv_dtype_t dtype, void* stream) {
  return rope_launch(positions, query, key, num_tokens, num_heads, num_kv_heads, head_size,
                     rot_dim, query_stride, key_stride, cos_sin_cache, is_neox, nullptr, dtype,
                     stream);
}

extern "C" int nmv_batched_rotary_embedding(const int64_t* positions, void* query, void* key,
                                            int num_tokens, int num_heads, int num_kv_heads,
                                            int head_size, int rot_dim, int64_t query_stride,
                                            int64_t key_stride, const void* cos_sin_cache,
                                            int is_neox, const int64_t* cos_sin_cache_offsets,
                                            nmv_dtype_t dtype, void* stream) {
  NMV_CHECK(cos_sin_cache_offsets != nullptr, "batched_rotary_embedding: null offsets");
  return rope_launch(positions, query, key, num_tokens, num_heads, num_kv_heads, head_size,
                     rot_dim, query_stride, key_stride, cos_sin_cache, is_neox,
                     cos_sin_cache_offsets, dtype, stream);
}

extern "C" int nmv_rotary_embedding_and_cache(const int64_t* positions, void* query, void* key,
                                              const void* value, int num_tokens, int num_heads,
                                              int num_kv_heads, int head_size, int rot_dim,
                                              int64_t query_stride, int64_t key_stride,
                                              int64_t value_stride, const void* cos_sin_cache,
                                              int is_neox, void* key_cache, void* value_cache,
                                              const int64_t* slot_mapping, int block_size,
                                              nmv_dtype_t dtype, nmv_kv_dtype_t kv_dtype,
                                              float kv_scale, void* stream) {
  NMV_HALF_ONLY("rotary_embedding_and_cache");
  NMV_CHECK(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size,
            "rotary_embedding_and_cache: bad rot_dim %d for head_size %d", rot_dim, head_size);
  NMV_CHECK(kv_dtype == NMV_KV_AUTO || kv_dtype == NMV_KV_FP8_E4M3,
            "rotary_embedding_and_cache: unsupported kv cache dtype %d", (int)kv_dtype);
  const int x = kv_dtype == NMV_KV_AUTO ? 8 : 16;
  NMV_CHECK(head_size % x == 0, "rotary_embedding_and_cache: head_size %d not a multiple of x=%d",
            head_size, x);
  NMV_CHECK(block_size > 0 && num_kv_heads > 0 && num_heads > 0, "rotary_embedding_and_cache: bad shape");
  if (num_tokens == 0) return NMV_OK;
  const int pairs = num_heads * rot_dim / 2;
  dim3 grid(num_tokens, std::min(std::max(pairs / 512, 1), 8)), block(std::min(std::max(pairs, 64), 512));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH_RC(T, NEOX, FP8)                                                                  \
  hipLaunchKernelGGL((rope_and_cache_kernel<T, NEOX, FP8>), grid, block, 0, s, positions,        \
                     (uint16_t*)query, (uint16_t*)key, (const uint16_t*)value,                   \
                     (const uint16_t*)cos_sin_cache, rot_dim, query_stride, key_stride,          \
                     value_stride, num_heads, num_kv_heads, head_size, key_cache, value_cache,   \
                     slot_mapping, block_size, kv_scale)
#define LAUNCH_RC_T(T)                                                       \
  if (kv_dtype == NMV_KV_AUTO) {                                             \
    if (is_neox) LAUNCH_RC(T, true, false); else LAUNCH_RC(T, false, false); \
  } else {                                                                   \
    if (is_neox) LAUNCH_RC(T, true, true); else LAUNCH_RC(T, false, true);   \
  }
  if (dtype == NMV_F16) LAUNCH_RC_T(F16) else LAUNCH_RC_T(BF16)
#undef LAUNCH_RC_T
#undef LAUNCH_RC
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

static int rotary_embedding_and_cache_partial_impl(const int64_t* positions, const float* slab,
                                                   int splits, bool slab16, void* qkv_out, int num_tokens,
                                                      int num_heads, int num_kv_heads, int head_size,
                                                      const void* cos_sin_cache, void* key_cache,
                                                      void* value_cache, const int64_t* slot_mapping,
                                                      int block_size, nmv_dtype_t dtype,
                                                      nmv_kv_dtype_t kv_dtype, float kv_scale,
                                                      void* stream) {
  NMV_HALF_ONLY("rotary_embedding_and_cache_partial");
  NMV_CHECK(splits >= 1 && slab != nullptr && qkv_out != nullptr, "rotary_embedding_and_cache_partial: bad arguments");
  NMV_CHECK(head_size % 16 == 0 && num_heads > 0 && num_kv_heads > 0,
            "rotary_embedding_and_cache_partial: head_size must be a multiple of 16");
  NMV_CHECK(kv_dtype == NMV_KV_AUTO || kv_dtype == NMV_KV_FP8_E4M3,
            "rotary_embedding_and_cache_partial: unsupported kv cache dtype %d", (int)kv_dtype);
  NMV_CHECK((key_cache == nullptr) == (value_cache == nullptr) &&
                (key_cache == nullptr || (slot_mapping != nullptr && block_size > 0)),
            "rotary_embedding_and_cache_partial: caches / slot_mapping");
  if (num_tokens == 0) return NMV_OK;
  const int64_t stride = (int64_t)num_tokens * (num_heads + 2 * num_kv_heads) * head_size;
  const int items = ((num_heads + num_kv_heads) * (head_size / 2) + num_kv_heads * head_size) / 4;
  dim3 grid(num_tokens, (items + 255) / 256), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH_RS(T, FP8)                                                                          \
  hipLaunchKernelGGL((rope_and_cache_slab_kernel<T, FP8>), grid, block, 0, s, positions, slab,     \
                     slab16 ? -splits : splits, stride, (uint16_t*)qkv_out, (const uint16_t*)cos_sin_cache, num_heads, \
                     num_kv_heads, head_size, key_cache, value_cache, slot_mapping, block_size,    \
                     kv_scale)
  if (dtype == NMV_F16) {
    if (kv_dtype == NMV_KV_AUTO) LAUNCH_RS(F16, false); else LAUNCH_RS(F16, true);
  } else {
    if (kv_dtype == NMV_KV_AUTO) LAUNCH_RS(BF16, false); else LAUNCH_RS(BF16, true);
  }
#undef LAUNCH_RS
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_rotary_embedding_and_cache_partial(const int64_t* positions, const float* slab,
                                                      int splits, void* qkv_out, int num_tokens,
                                                      int num_heads, int num_kv_heads, int head_size,
                                                      const void* cos_sin_cache, void* key_cache,
                                                      void* value_cache, const int64_t* slot_mapping,
                                                      int block_size, nmv_dtype_t dtype,
                                                      nmv_kv_dtype_t kv_dtype, float kv_scale,
                                                      void* stream) {
  return rotary_embedding_and_cache_partial_impl(positions, slab, splits, false, qkv_out, num_tokens, num_heads, num_kv_heads,
                                                 head_size, cos_sin_cache, key_cache, value_cache, slot_mapping, block_size,
                                                 dtype, kv_dtype, kv_scale, stream);
}

// the same with the slabs in the model dtype (mode 3 of nmv_w4_native_gemm)
extern "C" int nmv_rotary_embedding_and_cache_partial16(const int64_t* positions, const void* slab,
                                                        int splits, void* qkv_out, int num_tokens,
                                                        int num_heads, int num_kv_heads, int head_size,
                                                        const void* cos_sin_cache, void* key_cache,
                                                        void* value_cache, const int64_t* slot_mapping,
                                                        int block_size, nmv_dtype_t dtype,
                                                        nmv_kv_dtype_t kv_dtype, float kv_scale,
                                                        void* stream) {
  return rotary_embedding_and_cache_partial_impl(positions, (const float*)slab, splits, true, qkv_out, num_tokens, num_heads,
                                                 num_kv_heads, head_size, cos_sin_cache, key_cache, value_cache, slot_mapping,
                                                 block_size, dtype, kv_dtype, kv_scale, stream);
}

extern "C" int nmv_act_and_mul(void* out, const void* input, int num_tokens, int d, int act,
                               nmv_dtype_t dtype, void* stream) {
  NMV_HALF_ONLY("act_and_mul");
  NMV_CHECK(act >= 0 && act <= 2, "act_and_mul: unknown activation %d", act);
  if (num_tokens == 0 || d == 0) return NMV_OK;
  dim3 grid(num_tokens), block(std::min(std::max(d / 8, 64), 1024));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH_AM(T, A)                                                                    \
  hipLaunchKernelGGL((act_and_mul_kernel<T, A>), grid, block, 0, s, (uint16_t*)out,       \
                     (const uint16_t*)input, d)
  if (dtype == NMV_F16) {
    if (act == 0) LAUNCH_AM(F16, 0); else if (act == 1) LAUNCH_AM(F16, 1); else LAUNCH_AM(F16, 2);
  } else {
    if (act == 0) LAUNCH_AM(BF16, 0); else if (act == 1) LAUNCH_AM(BF16, 1); else LAUNCH_AM(BF16, 2);
  }
#undef LAUNCH_AM
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_activation(void* out, const void* input, int num_tokens, int d, int act,
                              nmv_dtype_t dtype, void* stream) {
  NMV_HALF_ONLY("activation");
  NMV_CHECK(act >= 0 && act <= 2, "activation: unknown activation %d", act);
  if (num_tokens == 0 || d == 0) return NMV_OK;
  dim3 grid(num_tokens), block(std::min(std::max(d, 64), 1024));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH_ACT(T, A)                                                                   \
  hipLaunchKernelGGL((activation_kernel<T, A>), grid, block, 0, s, (uint16_t*)out,        \
                     (const uint16_t*)input, d)
  if (dtype == NMV_F16) {
    if (act == 0) LAUNCH_ACT(F16, 0); else if (act == 1) LAUNCH_ACT(F16, 1); else LAUNCH_ACT(F16, 2);
  } else {
    if (act == 0) LAUNCH_ACT(BF16, 0); else if (act == 1) LAUNCH_ACT(BF16, 1); else LAUNCH_ACT(BF16, 2);
  }
#undef LAUNCH_ACT
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}
