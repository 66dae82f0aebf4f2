// W8A8 scaled matmul (the reference's cutlass_scaled_mm) for gfx950:
//   out[M,N] = a_scales[M|1] * (b_scales[N|1] * (A[M,K] . B[K,N])) (+ bias[N])
// A row-major int8 or fp8-e4m3; B column-major (b.stride(0) == 1), i.e. B^T [N, K] is row-major.
// Behavioural reference: /root/reference/csrc/quantization/cutlass_w8a8/scaled_mm_entry.cu:48-100
// (argument checks), scaled_mm_c2x.cu:75-140 (ScaledEpilogue: fp32 scale multiply, one rounding to
// the output type); the CUTLASS 3.5.0 mainloop itself is not in the reference tree -- the int8
// path accumulates exactly in int32, the fp8 path in fp32, as CUTLASS does.
//
// Both operands are K-contiguous, which is exactly what MFMA wants: the weights are the MFMA "A"
// operand (16 rows = 16 output columns n), the activations the "B" operand (16 columns = 16
// tokens); lane (r = l&15, g = l>>4) loads 16 bytes at row r, k = k0 + 16 g of each -- no LDS, no
// shuffles.  int8: v_mfma_i32_16x16x64_i8 (one per 16-byte load); fp8: two v_mfma_f32_16x16x32_fp8_fp8
// per load (low / high 8 bytes; the k permutation is the same on both operands, so it cancels).
// Decode is weight-streaming: a wave owns 16 columns x all (<=64) rows and one K slice, the 4 waves
// of a workgroup split K and reduce through LDS; no cross-workgroup reduction.
// HBM-bound: algorithmic bytes K*N + M*K + 2*M*N + 4*(M+N).
#include "common.h"

namespace nmv {

typedef __attribute__((ext_vector_type(4))) int i32x4_t;

struct MMParams {
  const uint8_t* a;     // [M, K] row-major, lda
  const uint8_t* bt;    // [N, K] row-major (= column-major B), ldb
  void* out;            // [M, N] row-major, ldc (elements)
  const float* a_scales;
  const float* b_scales;
  const void* bias;     // out dtype, or null
  int M, N, K;
  int64_t lda, ldb, ldc;
  int a_per_row, b_per_col;  // 1: per-token / per-channel scales, 0: per-tensor
};

template <bool FP8> struct Acc;
template <> struct Acc<false> {
  using type = i32x4_t;
  static __device__ __forceinline__ type zero() { return type{0, 0, 0, 0}; }
  static __device__ __forceinline__ type mma(uint4 w, uint4 x, type c) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4_t, w),
                                                 __builtin_bit_cast(i32x4_t, x), c, 0, 0, 0);
  }
};
template <> struct Acc<true> {
  using type = f32x4_t;
  static __device__ __forceinline__ type zero() { return type{0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ type mma(uint4 w, uint4 x, type c) {
    const long wl = (long)(((uint64_t)w.y << 32) | w.x), wh = (long)(((uint64_t)w.w << 32) | w.z);
    const long xl = (long)(((uint64_t)x.y << 32) | x.x), xh = (long)(((uint64_t)x.w << 32) | x.z);
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wl, xl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wh, xh, c, 0, 0, 0);
  }
};

constexpr int MM_THREADS = 256;
// A wave owns NT 16-column tiles x MT 16-row tiles: every activation fragment it loads is used for
// NT column tiles and every weight fragment for MT row tiles, so at M = 64 the activations are
// re-read once per 64 columns instead of once per 16 (that re-read, 4x the weight bytes on the
// wide projections, was what held the first version at 1.1-1.6 TB/s at M = 64).  UN 64-byte
// k-steps are in flight per lane.
template <typename T, bool FP8, int MT, int NT, int UN>
__global__ __launch_bounds__(MM_THREADS) void scaled_mm_kernel(const MMParams p) {
  using A = Acc<FP8>;
  __shared__ float red[3][NT][MT][4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * (16 * NT);
  const int m0 = blockIdx.y * (16 * MT);
  // K split over the 4 waves in multiples of 64
  const int ksteps = (p.K + 63) / 64;
  const int per_wave = (ksteps + 3) / 4;
  const int ks0 = wave * per_wave, ks1 = min(ks0 + per_wave, ksteps);

  const uint8_t* wp[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n_row = min(n0 + j * 16 + r, p.N - 1);
    wp[j] = p.bt + (int64_t)n_row * p.ldb + g * 16;
  }
  const uint8_t* ap[MT];
  bool a_ok[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = m0 + t * 16 + r;
    a_ok[t] = m < p.M;
    ap[t] = p.a + (int64_t)min(m, p.M - 1) * p.lda + g * 16;
  }
  typename A::type acc[NT][MT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[j][t] = A::zero();

  for (int ks = ks0; ks < ks1; ks += UN) {
    uint4 w[UN][NT], x[UN][MT];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = (ks + u) * 64 + g * 16;
      const bool ok = (ks + u) < ks1 && k < p.K;  // K % 16 == 0: a 16-byte chunk is all-or-nothing
#pragma unroll
      for (int j = 0; j < NT; ++j)
        w[u][j] = ok ? ld16(wp[j] + (int64_t)(ks + u) * 64) : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < MT; ++t)
        x[u][t] = (ok && a_ok[t]) ? ld16(ap[t] + (int64_t)(ks + u) * 64) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[j][t] = A::mma(w[u][j], x[u][t], acc[j][t]);
  }

  // cross-wave K reduction in the accumulator's own type (int32 stays exact)
  using elem_t = decltype(acc[0][0][0] + acc[0][0][0]);
  elem_t (*redt)[NT][MT][4][64] = reinterpret_cast<elem_t (*)[NT][MT][4][64]>(red);
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) redt[wave - 1][j][t][i][lane] = acc[j][t][i];
  }
  __syncthreads();
  if (wave != 0) return;

  // epilogue: D[row = n_idx][col = m]; lane (m = l&15, g) holds n = n0 + 16 j + 4g + i
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int nb = n0 + j * 16 + 4 * g;
    if (nb >= p.N) continue;
    float bs[4], bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = min(nb + i, p.N - 1);
      bs[i] = p.b_scales[p.b_per_col ? n : 0];
      bv[i] = p.bias ? T::to_float(reinterpret_cast<const uint16_t*>(p.bias)[n]) : 0.f;
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int m = m0 + t * 16 + r;
      if (m >= p.M) continue;
      const float as = p.a_scales[p.a_per_row ? m : 0];
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        elem_t v = acc[j][t][i];
#pragma unroll
        for (int ww = 0; ww < 3; ++ww) v += redt[ww][j][t][i][lane];
        o[i] = fmaf(as, bs[i] * (float)v, bv[i]);
      }
      uint16_t* dst = reinterpret_cast<uint16_t*>(p.out) + (int64_t)m * p.ldc + nb;
      if (nb + 3 < p.N && (reinterpret_cast<uintptr_t>(dst) & 7) == 0) {
        uint2 pk;
        pk.x = T::pack2(o[0], o[1]);
        pk.y = T::pack2(o[2], o[3]);
        *reinterpret_cast<uint2*>(dst) = pk;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (nb + i < p.N) dst[i] = T::from_float(o[i]);
      }
    }
  }
}

template <typename T, bool FP8>
static void launch_mm(const MMParams& p, hipStream_t s) {
#define MM_CASE(MT_, NT_, UN_)                                                                       \
  {                                                                                                  \
    dim3 grid((p.N + 16 * NT_ - 1) / (16 * NT_), (p.M + 16 * MT_ - 1) / (16 * MT_));                 \
    hipLaunchKernelGGL((scaled_mm_kernel<T, FP8, MT_, NT_, UN_>), grid, dim3(MM_THREADS), 0, s, p);  \
  }
  // wide enough to keep >= ~256 workgroups: more columns per wave (activation reuse)
  const int n16 = (p.N + 15) / 16;
  // measured on the Llama-3-8B shapes: one column tile per wave is best at M <= 16 (weight
  // streaming, 4.9 TB/s on gate_up), reuse pays from M = 32 on the wide projections
  if (p.M <= 16) {
    MM_CASE(1, 1, 4)
  } else if (p.M <= 32) {
    if (n16 >= 1024) MM_CASE(2, 2, 4) else MM_CASE(2, 1, 4)
  } else {
    if (n16 >= 1024) MM_CASE(4, 4, 2) else if (n16 >= 384) MM_CASE(4, 2, 2) else MM_CASE(4, 1, 4)
  }
#undef MM_CASE
}

}  // namespace nmv

using namespace nmv;

extern "C" int nmv_cutlass_scaled_mm_supports_fp8(int64_t cuda_device_capability) {
  // scaled_mm_entry.cu:32-46 gates fp8 on sm89+; gfx950 (capability 95) has native OCP fp8 MFMA
  return cuda_device_capability >= 89 ? 1 : 0;
}

extern "C" int nmv_scaled_mm(void* out, const void* a, const void* b, const float* a_scales,
                             const float* b_scales, const void* bias, int M, int N, int K,
                             int64_t lda, int64_t ldb, int64_t ldc, int a_scales_numel,
                             int b_scales_numel, nmv_q8_dtype_t in_dtype, nmv_dtype_t out_dtype,
                             void* stream) {
  NMV_CHECK(out_dtype == NMV_F16 || out_dtype == NMV_BF16, "cutlass_scaled_mm: out must be fp16/bf16");
  NMV_CHECK(in_dtype == NMV_I8 || in_dtype == NMV_FP8_E4M3, "cutlass_scaled_mm: a/b must be int8 or fp8_e4m3");
  NMV_CHECK(a_scales_numel == 1 || a_scales_numel == M, "cutlass_scaled_mm: a_scales.numel() must be 1 or M");
  NMV_CHECK(b_scales_numel == 1 || b_scales_numel == N, "cutlass_scaled_mm: b_scales.numel() must be 1 or N");
  NMV_CHECK(K % 16 == 0 && ldb % 16 == 0 && ldc % 16 == 0 && lda % 16 == 0,
            "cutlass_scaled_mm: K, lda, b.stride(1) and c.stride(0) must be multiples of 16");
  NMV_CHECK(in_dtype == NMV_FP8_E4M3 || K <= 131072, "cutlass_scaled_mm: K too large for int32 accumulation");
  if (M == 0 || N == 0) return NMV_OK;
  MMParams p{(const uint8_t*)a, (const uint8_t*)b, out, a_scales, b_scales, bias, M, N, K,
             lda, ldb, ldc, a_scales_numel == M && M > 1 ? 1 : (a_scales_numel == M ? 1 : 0),
             b_scales_numel == N && N > 1 ? 1 : (b_scales_numel == N ? 1 : 0)};
  hipStream_t s = (hipStream_t)stream;
  if (in_dtype == NMV_I8) {
    if (out_dtype == NMV_F16) launch_mm<F16, false>(p, s); else launch_mm<BF16, false>(p, s);
  } else {
    if (out_dtype == NMV_F16) launch_mm<F16, true>(p, s); else launch_mm<BF16, true>(p, s);
  }
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}
