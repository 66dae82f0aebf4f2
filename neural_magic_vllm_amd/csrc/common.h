// Shared device helpers for the gfx950 (CDNA4, wave64) kernels in this directory.
// Everything here is written for MI355X only: 64-lane wavefronts, OCP fp8 (e4m3fn),
// v_dot2c_f32_bf16 / v_dot2_f32_f16 packed dot products.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "../../include/nmvllm_hip.h"

namespace nmv {

constexpr int WAVE = 64;

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

// ---- error plumbing for the C ABI -------------------------------------------------
void set_error(const char* fmt, ...);
void append_error(const char* fmt, ...);
#define NMV_CHECK(cond, ...)              \
  do {                                    \
    if (!(cond)) {                        \
      ::nmv::set_error(__VA_ARGS__);      \
      return NMV_ERR_INVALID;             \
    }                                     \
  } while (0)
#define NMV_LAUNCH_CHECK()                                                    \
  do {                                                                        \
    hipError_t e_ = hipGetLastError();                                        \
    if (e_ != hipSuccess) {                                                   \
      ::nmv::set_error("HIP launch failed: %s", hipGetErrorString(e_));       \
      return NMV_ERR_HIP;                                                     \
    }                                                                         \
  } while (0)

// ---- scalar types -----------------------------------------------------------------
// Storage is always raw 16-bit; the tag type selects the arithmetic.
struct F16 {
  using raw = _Float16;
  static __device__ __forceinline__ float to_float(uint16_t b) {
    return (float)__builtin_bit_cast(_Float16, b);
  }
  static __device__ __forceinline__ uint16_t from_float(float f) {
    return __builtin_bit_cast(uint16_t, (_Float16)f);
  }
  // packed pair dot product with fp32 accumulate: c + a.x*b.x + a.y*b.y
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b), c,
                                  false);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    f16x2_t v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
  }
};

struct BF16 {
  using raw = __bf16;
  static __device__ __forceinline__ float to_float(uint16_t b) {
    return __uint_as_float(((uint32_t)b) << 16);
  }
  static __device__ __forceinline__ uint16_t from_float(float f) {
    // plain cast -> v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
    return __builtin_bit_cast(uint16_t, (__bf16)f);
  }
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a),
                                           __builtin_bit_cast(bf16x2_t, b), c, false);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
  }
};

template <typename T>
__device__ __forceinline__ float lo_f(uint32_t pair) {
  return T::to_float((uint16_t)(pair & 0xffffu));
}
template <typename T>
__device__ __forceinline__ float hi_f(uint32_t pair) {
  return T::to_float((uint16_t)(pair >> 16));
}

// float -> model dtype -> float with the rounded bits passed through an empty asm, so that the
// rounding is materialised even when the value never reaches memory (hipcc was seen to drop the
// float -> _Float16 -> float round trip otherwise; glue_kernels.hip has the story)
template <typename T>
__device__ __forceinline__ float round_trip(float f) {
  uint32_t b = T::from_float(f);
  asm volatile("" : "+v"(b));
  return T::to_float((uint16_t)b);
}

// ---- OCP fp8 e4m3fn (gfx950 native) -----------------------------------------------
// two fp8 bytes (selected 16-bit half of `w`) -> two floats
template <bool HI>
__device__ __forceinline__ f32x2_t fp8x2_to_f32(uint32_t w) {
  return __builtin_amdgcn_cvt_pk_f32_fp8((int)w, HI);
}
__device__ __forceinline__ float fp8_to_f32(uint8_t b) {
  return __builtin_amdgcn_cvt_f32_fp8((int)b, 0);
}
// fp8 x4 (one dword) -> two packed T pairs: (byte 0, byte 1) and (byte 2, byte 3), exact (3 mantissa bits)
template <typename T>
__device__ __forceinline__ void fp8x4_to_pairs_t(uint32_t w, uint32_t& p0, uint32_t& p1) {
  const f32x2_t a = fp8x2_to_f32<false>(w);
  const f32x2_t b = fp8x2_to_f32<true>(w);
  p0 = T::pack2(a.x, a.y);
  p1 = T::pack2(b.x, b.y);
}
// float -> fp8 e4m3fn byte, round-to-nearest-even, saturating to +-448 (NaN stays NaN)
__device__ __forceinline__ uint8_t f32_to_fp8(float f) {
  float c = __builtin_fminf(__builtin_fmaxf(f, -448.f), 448.f);
  c = (f != f) ? f : c;
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(c, c, 0, false);
  return (uint8_t)(r & 0xff);
}

// ---- wave64 reductions ------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// 16-byte global load / store helpers
__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void st16(void* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
__device__ __forceinline__ uint2 ld8(const void* p) { return *reinterpret_cast<const uint2*>(p); }

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace nmv
