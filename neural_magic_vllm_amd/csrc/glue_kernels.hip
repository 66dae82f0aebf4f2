// Glue ops of a decoder layer: rms_norm, fused_add_rms_norm, rotary_embedding (+batched),
// act_and_mul (silu / gelu / gelu_tanh), element-wise gelu_new / gelu_fast / gelu_quick.
// Behavioural references: /root/reference/csrc/layernorm_kernels.cu:22-44,201-290,
// csrc/pos_encoding_kernels.cu:10-119, csrc/activation_kernels.cu:14-150.
// All of them are HBM-bound streaming kernels: 16-byte vectors per lane, fp32 math inside,
// and the reference's intermediate roundings to the model dtype are reproduced exactly
// (e.g. rms_norm rounds x*rsqrt(var) to the model dtype BEFORE multiplying by the weight).
#include "common.h"

// The reference rounds every intermediate to the model dtype (c10::Half / c10::BFloat16
// operators).  With contraction on, hipcc folds the fp16 paths into v_fma_f16 and skips one of
// those roundings -- keep the arithmetic exactly as written.
#pragma clang fp contract(off)

namespace nmv {

template <typename T>
__device__ __forceinline__ float rnd(float f) {  // round-trip through the model dtype
  return T::to_float(T::from_float(f));
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

extern "C" int nmv_rms_norm(void* out, const void* input, const void* weight, float epsilon,
                            int num_tokens, int hidden_size, nmv_dtype_t dtype, void* stream) {
  NMV_HALF_ONLY("rms_norm");
  if (num_tokens == 0) return NMV_OK;
  dim3 grid(num_tokens), block(256);
  if (dtype == NMV_F16)
    hipLaunchKernelGGL((rms_norm_kernel<F16, false>), grid, block, 0, (hipStream_t)stream,
                       (uint16_t*)out, (uint16_t*)input, (uint16_t*)nullptr,
                       (const uint16_t*)weight, epsilon, hidden_size);
  else
    hipLaunchKernelGGL((rms_norm_kernel<BF16, false>), grid, block, 0, (hipStream_t)stream,
                       (uint16_t*)out, (uint16_t*)input, (uint16_t*)nullptr,
                       (const uint16_t*)weight, epsilon, hidden_size);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_fused_add_rms_norm(void* input, void* residual, const void* weight,
                                      float epsilon, int num_tokens, int hidden_size,
                                      nmv_dtype_t dtype, void* stream) {
  NMV_HALF_ONLY("fused_add_rms_norm");
  if (num_tokens == 0) return NMV_OK;
  dim3 grid(num_tokens), block(256);
  if (dtype == NMV_F16)
    hipLaunchKernelGGL((rms_norm_kernel<F16, true>), grid, block, 0, (hipStream_t)stream,
                       (uint16_t*)input, (uint16_t*)input, (uint16_t*)residual,
                       (const uint16_t*)weight, epsilon, hidden_size);
  else
    hipLaunchKernelGGL((rms_norm_kernel<BF16, true>), grid, block, 0, (hipStream_t)stream,
                       (uint16_t*)input, (uint16_t*)input, (uint16_t*)residual,
                       (const uint16_t*)weight, epsilon, hidden_size);
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
