// KV-cache ops for gfx950: reshape_and_cache, reshape_and_cache_flash, copy_blocks, swap_blocks,
// convert_fp8.  Behavioural reference: /root/reference/csrc/cache_kernels.cu (layouts and index
// arithmetic :152-204, :206-237, :68-94, :318-329).  The work is pure byte movement: every kernel
// here moves 16-byte vectors wherever the paged layout has 16 contiguous bytes.
#include "common.h"
#include "cache_write.h"
#include "fp32_path.h"

namespace nmv {

template <typename T, bool FP8>
__global__ void reshape_and_cache_kernel(const uint16_t* __restrict__ key,
                                         const uint16_t* __restrict__ value,
                                         void* __restrict__ key_cache_v,
                                         void* __restrict__ value_cache_v,
                                         const int64_t* __restrict__ slot_mapping,
                                         int64_t key_stride, int64_t value_stride, int num_heads,
                                         int head_size, int block_size, float kv_scale) {
  const int64_t token_idx = blockIdx.x;
  const int64_t slot_idx = slot_mapping[token_idx];
  if (slot_idx < 0) return;  // padding token (cache_kernels.cu:166-169)
  write_token_to_cache<T, FP8>(key, value, key_cache_v, value_cache_v, token_idx, slot_idx,
                               key_stride, value_stride, num_heads, head_size, block_size, kv_scale);
}

// flash layout: [num_blocks, block_size, num_heads, head_size] -- a straight row copy.
__global__ void reshape_and_cache_flash_kernel(const uint16_t* __restrict__ key,
                                               const uint16_t* __restrict__ value,
                                               uint16_t* __restrict__ k_cache,
                                               uint16_t* __restrict__ v_cache,
                                               const int64_t* __restrict__ slot_mapping,
                                               int64_t block_stride, int64_t key_stride,
                                               int64_t value_stride, int num_heads, int head_size,
                                               int block_size) {
  const int64_t token_idx = blockIdx.x;
  const int64_t slot_idx = slot_mapping[token_idx];
  if (slot_idx < 0) return;
  const int64_t block_idx = slot_idx / block_size;
  const int64_t block_offset = slot_idx % block_size;
  const int n = num_heads * head_size;
  const int64_t tgt0 = block_idx * block_stride + block_offset * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    k_cache[tgt0 + i] = key[token_idx * key_stride + i];
    v_cache[tgt0 + i] = value[token_idx * value_stride + i];
  }
}

// grid (num_layers, num_pairs); bytes per block are moved as 16-byte vectors when possible.
__global__ void copy_blocks_kernel(void* const* __restrict__ key_cache_ptrs,
                                   void* const* __restrict__ value_cache_ptrs,
                                   const int64_t* __restrict__ block_mapping,
                                   int64_t bytes_per_block) {
  const int layer_idx = blockIdx.x;
  const int pair_idx = blockIdx.y;
  char* kc = reinterpret_cast<char*>(key_cache_ptrs[layer_idx]);
  char* vc = reinterpret_cast<char*>(value_cache_ptrs[layer_idx]);
  const int64_t src = block_mapping[2 * pair_idx] * bytes_per_block;
  const int64_t dst = block_mapping[2 * pair_idx + 1] * bytes_per_block;
  if ((bytes_per_block & 15) == 0 && ((reinterpret_cast<uintptr_t>(kc) | reinterpret_cast<uintptr_t>(vc)) & 15) == 0) {
    const int64_t nvec = bytes_per_block >> 4;
    for (int64_t i = threadIdx.x; i < nvec; i += blockDim.x) {
      st16(kc + dst + (i << 4), ld16(kc + src + (i << 4)));
    }
    for (int64_t i = threadIdx.x; i < nvec; i += blockDim.x) {
      st16(vc + dst + (i << 4), ld16(vc + src + (i << 4)));
    }
  } else {
    for (int64_t i = threadIdx.x; i < bytes_per_block; i += blockDim.x) kc[dst + i] = kc[src + i];
    for (int64_t i = threadIdx.x; i < bytes_per_block; i += blockDim.x) vc[dst + i] = vc[src + i];
  }
}

template <typename T, bool TO_FP8>
__global__ void convert_fp8_kernel(void* __restrict__ dst_v, const void* __restrict__ src_v,
                                   float scale, int64_t block_stride) {
  const int64_t block_idx = blockIdx.x;
  for (int64_t i = threadIdx.x; i < block_stride; i += blockDim.x) {
    const int64_t idx = block_idx * block_stride + i;
    if constexpr (TO_FP8) {
      const uint16_t s = reinterpret_cast<const uint16_t*>(src_v)[idx];
      reinterpret_cast<uint8_t*>(dst_v)[idx] = f32_to_fp8(T::to_float(s) / scale);
    } else {
      const uint8_t s = reinterpret_cast<const uint8_t*>(src_v)[idx];
      reinterpret_cast<uint16_t*>(dst_v)[idx] = T::from_float(fp8_to_f32(s) * scale);
    }
  }
}

}  // namespace nmv

using namespace nmv;

extern "C" int nmv_reshape_and_cache(const void* key, const void* value, void* key_cache,
                                     void* value_cache, const int64_t* slot_mapping,
                                     int num_tokens, int num_kv_heads, int head_size,
                                     int block_size, int64_t key_stride, int64_t value_stride,
                                     nmv_dtype_t dtype, nmv_kv_dtype_t kv_dtype, float kv_scale,
                                     void* stream) {
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16 || dtype == NMV_F32, "reshape_and_cache: unsupported dtype %d",
            (int)dtype);
  NMV_CHECK(kv_dtype == NMV_KV_AUTO || kv_dtype == NMV_KV_FP8_E4M3,
            "reshape_and_cache: unsupported kv cache dtype %d", (int)kv_dtype);
  const int x = kv_dtype == NMV_KV_AUTO ? (dtype == NMV_F32 ? 4 : 8) : 16;
  NMV_CHECK(head_size % x == 0, "reshape_and_cache: head_size %d not a multiple of x=%d",
            head_size, x);
  NMV_CHECK(block_size > 0 && num_kv_heads > 0, "reshape_and_cache: bad shape");
  if (num_tokens == 0) return NMV_OK;
  if (dtype == NMV_F32) {   // float models: fp32_path.hip
    f32_reshape_and_cache(key, value, key_cache, value_cache, slot_mapping, num_tokens, num_kv_heads, head_size,
                          block_size, key_stride, value_stride, kv_dtype == NMV_KV_FP8_E4M3, kv_scale,
                          (hipStream_t)stream);
    NMV_LAUNCH_CHECK();
    return NMV_OK;
  }
  dim3 grid(num_tokens);
  dim3 block(std::min(num_kv_heads * head_size, 512));
  hipStream_t s = (hipStream_t)stream;
  const uint16_t* k = (const uint16_t*)key;
  const uint16_t* v = (const uint16_t*)value;
#define LAUNCH_RC(T, FP8)                                                                      \
  hipLaunchKernelGGL((reshape_and_cache_kernel<T, FP8>), grid, block, 0, s, k, v, key_cache,   \
                     value_cache, slot_mapping, key_stride, value_stride, num_kv_heads,        \
                     head_size, block_size, kv_scale)
  if (kv_dtype == NMV_KV_AUTO) {
    if (dtype == NMV_F16) LAUNCH_RC(F16, false); else LAUNCH_RC(BF16, false);
  } else {
    if (dtype == NMV_F16) LAUNCH_RC(F16, true); else LAUNCH_RC(BF16, true);
  }
#undef LAUNCH_RC
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_reshape_and_cache_flash(const void* key, const void* value, void* key_cache,
                                           void* value_cache, const int64_t* slot_mapping,
                                           int num_tokens, int num_kv_heads, int head_size,
                                           int block_size, int64_t key_stride,
                                           int64_t value_stride, int64_t block_stride,
                                           nmv_dtype_t dtype, void* stream) {
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16,
            "reshape_and_cache_flash: unsupported dtype %d", (int)dtype);
  if (num_tokens == 0) return NMV_OK;
  dim3 grid(num_tokens);
  dim3 block(std::min(num_kv_heads * head_size, 512));
  hipLaunchKernelGGL(reshape_and_cache_flash_kernel, grid, block, 0, (hipStream_t)stream,
                     (const uint16_t*)key, (const uint16_t*)value, (uint16_t*)key_cache,
                     (uint16_t*)value_cache, slot_mapping, block_stride, key_stride, value_stride,
                     num_kv_heads, head_size, block_size);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_copy_blocks(void* const* key_cache_ptrs, void* const* value_cache_ptrs,
                               const int64_t* block_mapping, int num_layers, int num_pairs,
                               int64_t numel_per_block, int elem_size, void* stream) {
  NMV_CHECK(elem_size == 1 || elem_size == 2 || elem_size == 4, "copy_blocks: bad elem_size %d",
            elem_size);
  if (num_layers == 0 || num_pairs == 0) return NMV_OK;
  dim3 grid(num_layers, num_pairs);
  dim3 block(256);
  hipLaunchKernelGGL(copy_blocks_kernel, grid, block, 0, (hipStream_t)stream, key_cache_ptrs,
                     value_cache_ptrs, block_mapping, numel_per_block * elem_size);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_swap_blocks(const void* src, void* dst, const int64_t* block_mapping_host,
                               int num_pairs, int64_t block_bytes, int kind, void* stream) {
  NMV_CHECK(kind >= 0 && kind <= 2, "swap_blocks: Invalid device combination (kind=%d)", kind);
  const hipMemcpyKind k = kind == 0 ? hipMemcpyDeviceToDevice
                                    : (kind == 1 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost);
  const char* s = (const char*)src;
  char* d = (char*)dst;
  for (int i = 0; i < num_pairs; ++i) {
    const int64_t sb = block_mapping_host[2 * i], db = block_mapping_host[2 * i + 1];
    hipError_t e = hipMemcpyAsync(d + db * block_bytes, s + sb * block_bytes, block_bytes, k,
                                  (hipStream_t)stream);
    if (e != hipSuccess) {
      set_error("swap_blocks: hipMemcpyAsync failed: %s", hipGetErrorString(e));
      return NMV_ERR_HIP;
    }
  }
  return NMV_OK;
}

extern "C" int nmv_convert_fp8(void* dst, const void* src, int64_t num_blocks,
                               int64_t block_stride, nmv_dtype_t dtype, int to_fp8, float scale,
                               void* stream) {
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16 || dtype == NMV_F32, "convert_fp8: unsupported dtype %d",
            (int)dtype);
  if (num_blocks == 0) return NMV_OK;
  if (dtype == NMV_F32) {
    f32_convert_fp8(dst, src, num_blocks, block_stride, to_fp8 != 0, scale, (hipStream_t)stream);
    NMV_LAUNCH_CHECK();
    return NMV_OK;
  }
  dim3 grid(num_blocks);
  dim3 block((unsigned)std::min<int64_t>(block_stride, 512));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH_CV(T, TO)                                                                   \
  hipLaunchKernelGGL((convert_fp8_kernel<T, TO>), grid, block, 0, s, dst, src,            \
                     scale, block_stride)
  if (to_fp8) {
    if (dtype == NMV_F16) LAUNCH_CV(F16, true); else LAUNCH_CV(BF16, true);
  } else {
    if (dtype == NMV_F16) LAUNCH_CV(F16, false); else LAUNCH_CV(BF16, false);
  }
#undef LAUNCH_CV
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}
