// Device-side body of reshape_and_cache (reference csrc/cache_kernels.cu:152-204).  The fused
// rotary + cache kernel (glue_kernels.hip) stores the same bytes element by element from the lanes
// that rotate them.
#pragma once
#include "common.h"

namespace nmv {

// One workgroup per token.  K: the paged layout keeps x = 16/sizeof(cache_t) consecutive head
// elements contiguous, so each lane moves one x-element chunk (16 B store).  V: the layout is
// [head, d, block_offset], a 2-byte (1-byte for fp8) scatter per element -- unavoidable for a
// single token, the bytes involved are tiny.
template <typename T, bool FP8>
__device__ __forceinline__ void write_token_to_cache(const uint16_t* __restrict__ key,
                                                     const uint16_t* __restrict__ value,
                                                     void* __restrict__ key_cache_v,
                                                     void* __restrict__ value_cache_v,
                                                     int64_t token_idx, int64_t slot_idx,
                                                     int64_t key_stride, int64_t value_stride,
                                                     int num_heads, int head_size, int block_size,
                                                     float kv_scale) {
  const int64_t block_idx = slot_idx / block_size;
  const int64_t block_offset = slot_idx % block_size;
  constexpr int X = FP8 ? 16 : 8;  // elements per 16-byte K chunk
  const int n = num_heads * head_size;

  // ---- K: chunks of X elements ----
  const int n_chunks = n / X;  // head_size % X == 0 is checked on the host
  for (int c = threadIdx.x; c < n_chunks; c += blockDim.x) {
    const int e0 = c * X;
    const int head_idx = e0 / head_size;
    const int head_off = e0 % head_size;
    const int x_idx = head_off / X;
    const uint16_t* src = key + token_idx * key_stride + e0;
    const int64_t tgt = (((block_idx * num_heads + head_idx) * (head_size / X) + x_idx) * block_size +
                         block_offset) * X;
    if constexpr (!FP8) {
      uint16_t* kc = reinterpret_cast<uint16_t*>(key_cache_v);
      // source may be only 2-byte aligned in principle (a strided qkv slice); in practice the
      // offsets are multiples of head_size.  Use 16-B vectors when aligned, else scalars.
      if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        st16(kc + tgt, ld16(src));
      } else {
#pragma unroll
        for (int j = 0; j < X; ++j) kc[tgt + j] = src[j];
      }
    } else {
      uint8_t* kc = reinterpret_cast<uint8_t*>(key_cache_v);
      uint32_t w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t b = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          b |= (uint32_t)f32_to_fp8(T::to_float(src[j * 4 + t]) / kv_scale) << (8 * t);
        w[j] = b;
      }
      st16(kc + tgt, make_uint4(w[0], w[1], w[2], w[3]));
    }
  }
  // ---- V: element scatter ----
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int head_idx = i / head_size;
    const int head_off = i % head_size;
    const int64_t tgt =
        ((block_idx * num_heads + head_idx) * head_size + head_off) * block_size + block_offset;
    const uint16_t v = value[token_idx * value_stride + i];
    if constexpr (!FP8) {
      reinterpret_cast<uint16_t*>(value_cache_v)[tgt] = v;
    } else {
      reinterpret_cast<uint8_t*>(value_cache_v)[tgt] = f32_to_fp8(T::to_float(v) / kv_scale);
    }
  }
}

// one element of a token's key into the paged K cache ([block][head][d / x][offset][x]); the fused
// rope + cache kernels store the rounded bits they have just produced
template <typename T, bool FP8>
__device__ __forceinline__ void cache_store_k(void* key_cache, int64_t head_base /*(block*heads+head)*/,
                                              int head_size, int block_size, int64_t block_offset,
                                              int d, uint16_t bits, float kv_scale) {
  constexpr int X = FP8 ? 16 : 8;
  const int64_t tgt = ((head_base * (head_size / X) + d / X) * block_size + block_offset) * X + d % X;
  if constexpr (FP8)
    reinterpret_cast<uint8_t*>(key_cache)[tgt] = f32_to_fp8(T::to_float(bits) / kv_scale);
  else
    reinterpret_cast<uint16_t*>(key_cache)[tgt] = bits;
}

}  // namespace nmv
