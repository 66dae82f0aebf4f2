// Activation quantisers: static / dynamic-per-token int8 and static / dynamic-per-tensor fp8.
// Behavioural references: /root/reference/csrc/quantization/compressed_tensors/
// int8_quant_kernels.cu:6-75 (x/scale, round-to-nearest-even, saturate; dynamic: scale = absmax/127,
// x * (127/absmax)) and csrc/quantization/fp8/common.cu:12-127 (x * (1/scale), clamp +-448, e4m3fn;
// dynamic: scale = absmax/448 through an atomic max on a pre-zeroed scalar).
// HBM-bound streaming kernels: 16-byte loads, 8-byte (int8 x8) / 8-byte (fp8 x8) stores.
#include "common.h"

namespace nmv {

__device__ __forceinline__ int8_t float_to_int8_rn(float x) {
  float d = __builtin_nearbyintf(x);  // v_rndne_f32
  d = fminf(fmaxf(d, -128.f), 127.f);
  return (int8_t)d;
}

__device__ __forceinline__ float block_max(float v, float* red) {
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

template <typename T, bool DYNAMIC>
__global__ __launch_bounds__(256) void int8_quant_kernel(const uint16_t* __restrict__ input,
                                                         int8_t* __restrict__ out,
                                                         float* __restrict__ scale, int hidden) {
  __shared__ float red[4];
  const int64_t row = (int64_t)blockIdx.x * hidden;
  const bool vec = (hidden % 8 == 0) && ((reinterpret_cast<uintptr_t>(input) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(out) & 7) == 0);
  float mul, div;
  if constexpr (DYNAMIC) {
    float amax = 0.f;
    if (vec) {
      for (int v = threadIdx.x; v < hidden / 8; v += blockDim.x) {
        const uint4 x = ld16(input + row + v * 8);
        const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(lo_f<T>(xs[j])), fabsf(hi_f<T>(xs[j]))));
      }
    } else {
      for (int i = threadIdx.x; i < hidden; i += blockDim.x) amax = fmaxf(amax, fabsf(T::to_float(input[row + i])));
    }
    amax = block_max(amax, red);
    if (threadIdx.x == 0) scale[blockIdx.x] = amax / 127.0f;
    mul = 127.0f / amax;
    div = 1.f;
  } else {
    mul = 1.f;
    div = *scale;
  }
  auto q = [&](float x) -> uint32_t {
    return (uint32_t)(uint8_t)float_to_int8_rn(DYNAMIC ? x * mul : x / div);
  };
  if (vec) {
    for (int v = threadIdx.x; v < hidden / 8; v += blockDim.x) {
      const uint4 x = ld16(input + row + v * 8);
      uint2 o;
      o.x = q(lo_f<T>(x.x)) | (q(hi_f<T>(x.x)) << 8) | (q(lo_f<T>(x.y)) << 16) | (q(hi_f<T>(x.y)) << 24);
      o.y = q(lo_f<T>(x.z)) | (q(hi_f<T>(x.z)) << 8) | (q(lo_f<T>(x.w)) << 16) | (q(hi_f<T>(x.w)) << 24);
      *reinterpret_cast<uint2*>(out + row + v * 8) = o;
    }
  } else {
    for (int i = threadIdx.x; i < hidden; i += blockDim.x)
      out[row + i] = (int8_t)q(T::to_float(input[row + i]));
  }
}

// atomic max on a non-negative float (fp8/common.cu:12-19)
__device__ __forceinline__ void atomic_max_pos(float* addr, float value) {
  atomicMax(reinterpret_cast<int*>(addr), __float_as_int(value));
}

template <typename T>
__global__ __launch_bounds__(256) void fp8_absmax_kernel(float* __restrict__ scale,
                                                         const uint16_t* __restrict__ input,
                                                         int64_t num_elems) {
  __shared__ float red[4];
  float amax = 0.f;
  const int64_t stride = (int64_t)blockDim.x * gridDim.x;
  const bool vec = (reinterpret_cast<uintptr_t>(input) & 15) == 0;
  const int64_t nvec = vec ? num_elems / 8 : 0;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
    const uint4 x = ld16(input + v * 8);
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(lo_f<T>(xs[j])), fabsf(hi_f<T>(xs[j]))));
  }
  for (int64_t i = nvec * 8 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_elems; i += stride)
    amax = fmaxf(amax, fabsf(T::to_float(input[i])));
  amax = block_max(amax, red);
  if (threadIdx.x == 0) atomic_max_pos(scale, amax / 448.0f);
}

template <typename T>
__global__ __launch_bounds__(256) void fp8_quant_kernel(uint8_t* __restrict__ out,
                                                        const uint16_t* __restrict__ input,
                                                        const float* __restrict__ scale,
                                                        int64_t num_elems) {
  const float inv = 1.0f / (*scale);  // the reference multiplies by the inverted scale (:104)
  const int64_t stride = (int64_t)blockDim.x * gridDim.x;
  const bool vec = ((reinterpret_cast<uintptr_t>(input) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 7) == 0);
  const int64_t nvec = vec ? num_elems / 8 : 0;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
    const uint4 x = ld16(input + v * 8);
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
    uint32_t b[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      b[2 * j] = f32_to_fp8(lo_f<T>(xs[j]) * inv);
      b[2 * j + 1] = f32_to_fp8(hi_f<T>(xs[j]) * inv);
    }
    uint2 o;
    o.x = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
    o.y = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
    *reinterpret_cast<uint2*>(out + v * 8) = o;
  }
  for (int64_t i = nvec * 8 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_elems; i += stride)
    out[i] = f32_to_fp8(T::to_float(input[i]) * inv);
}

}  // namespace nmv

using namespace nmv;

extern "C" int nmv_scaled_int8_quant(void* out, const void* input, float* scale, int num_tokens,
                                     int hidden_size, int dynamic, nmv_dtype_t dtype, void* stream) {
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16, "scaled_int8_quant: unsupported dtype %d", (int)dtype);
  NMV_CHECK(scale != nullptr, "scaled_int8_quant: null scale");
  if (num_tokens == 0 || hidden_size == 0) return NMV_OK;
  dim3 grid(num_tokens), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH_I8(T, DYN)                                                                     \
  hipLaunchKernelGGL((int8_quant_kernel<T, DYN>), grid, block, 0, s, (const uint16_t*)input,  \
                     (int8_t*)out, scale, hidden_size)
  if (dynamic) { if (dtype == NMV_F16) LAUNCH_I8(F16, true); else LAUNCH_I8(BF16, true); }
  else { if (dtype == NMV_F16) LAUNCH_I8(F16, false); else LAUNCH_I8(BF16, false); }
#undef LAUNCH_I8
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_scaled_fp8_quant(void* out, const void* input, float* scale, int64_t num_elems,
                                    int dynamic, nmv_dtype_t dtype, void* stream) {
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16, "scaled_fp8_quant: unsupported dtype %d", (int)dtype);
  NMV_CHECK(scale != nullptr, "scaled_fp8_quant: null scale");
  if (num_elems == 0) return NMV_OK;
  const int64_t nvec = cdiv64(num_elems, 8);
  dim3 grid((unsigned)std::min<int64_t>(cdiv64(nvec, 256), 2048)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dynamic) {
    // `scale` must be zero on entry (vllm/_custom_ops.py:316 allocates it with torch.zeros)
    if (dtype == NMV_F16)
      hipLaunchKernelGGL((fp8_absmax_kernel<F16>), grid, block, 0, s, scale, (const uint16_t*)input, num_elems);
    else
      hipLaunchKernelGGL((fp8_absmax_kernel<BF16>), grid, block, 0, s, scale, (const uint16_t*)input, num_elems);
  }
  if (dtype == NMV_F16)
    hipLaunchKernelGGL((fp8_quant_kernel<F16>), grid, block, 0, s, (uint8_t*)out, (const uint16_t*)input, scale, num_elems);
  else
    hipLaunchKernelGGL((fp8_quant_kernel<BF16>), grid, block, 0, s, (uint8_t*)out, (const uint16_t*)input, scale, num_elems);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}
