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
#include <type_traits>
#include <cstdlib>

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
  // split-K over blockIdx.z (M > 32): raw int32 / fp32 partials in slab[splits][M][N]
  int splits, k_per_split;
  void* slab;
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
  // K split over blockIdx.z (k_per_split, a multiple of 64) and then over the 4 waves, in
  // multiples of 64
  const int split_s0 = blockIdx.z * (p.k_per_split / 64);
  const int ksteps = min((p.K + 63) / 64, split_s0 + p.k_per_split / 64) - split_s0;
  const int per_wave = (ksteps + 3) / 4;
  const int ks0 = split_s0 + wave * per_wave, ks1 = min(ks0 + per_wave, split_s0 + ksteps);

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
    if (p.splits > 1) {  // raw partial sums; scaled_mm_reduce_kernel finishes
      elem_t* slab = reinterpret_cast<elem_t*>(p.slab) + (int64_t)blockIdx.z * p.M * p.N;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int m = m0 + t * 16 + r;
        if (m >= p.M) continue;
        typename A::type v = acc[j][t];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int ww = 0; ww < 3; ++ww) v[i] += redt[ww][j][t][i][lane];
        elem_t* dst = slab + (int64_t)m * p.N + nb;
        if (nb + 3 < p.N && (p.N & 3) == 0) {
          *reinterpret_cast<typename A::type*>(dst) = v;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (nb + i < p.N) dst[i] = v[i];
        }
      }
      continue;
    }
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
        // multiplies(a_scales, multiplies(b_scales, acc)); with a bias the outer node is multiply_add
        // (scaled_mm_c2x.cu:117-131, :157-171) -- no "+ 0" without one: it would turn -0 into +0
        const float tmp = bs[i] * (float)v;
        o[i] = p.bias ? fmaf(as, tmp, bv[i]) : as * tmp;
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


// ---------------------------------------------------------------------------------------------
// 17 .. 64 rows: the waves of a workgroup split N, not K, and share the activations through LDS.
// In scaled_mm_kernel every wave re-reads its own k slice of all rows from L2: a workgroup covers 16 NT columns, so the
// activation bytes through the CU's vector-memory path are 64 / (16 NT) x the weight bytes (o_proj at M = 64: 4 x) -- and
// that path, not HBM, is what the kernel runs at (weights + activations = 5.2 TB/s on every projection, the same number
// as the weights alone at M = 1).  Here a workgroup covers 64 NT columns (wave w: tiles w NT .. w NT + NT - 1)
// and one k range; a 256-byte chunk of all its rows is staged ONCE (global -> registers one chunk ahead -> LDS, XOR
// swizzled on 16-byte slots so that the 16 lanes of an MFMA operand read hit 16 different slots), read as MFMA operands
// by the four waves, while each wave streams its own weights straight into operand registers one chunk ahead.
// Activation bytes / weight bytes = 16 MT / (64 NT): 1/4 at NT = 4.  No cross-wave reduction; split-K as above.
template <typename T, bool FP8, int MT, int NT>
__global__ __launch_bounds__(MM_THREADS) void scaled_mm_wide_kernel(const MMParams p) {
  constexpr int D = NT == 4 ? 2 : 3;   // chunks of weights in registers: the one being multiplied and D - 1 on their way
  using A = Acc<FP8>;
  constexpr int ROWS = 16 * MT, STAGE = ROWS * 256, XPT = ROWS / 16;   // 16-byte pieces per thread and chunk
  __shared__ __attribute__((aligned(16))) uint8_t act[2][STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * (64 * NT) + wave * (16 * NT);
  const int m0 = blockIdx.y * ROWS;
  const int k_begin = blockIdx.z * p.k_per_split;
  const int nch = (min(p.K, k_begin + p.k_per_split) - k_begin) >> 8;   // 256-byte chunks (K % 256 == 0, k_per_split too)

  const uint8_t* wp[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) wp[j] = p.bt + (int64_t)min(n0 + j * 16 + r, p.N - 1) * p.ldb + k_begin + g * 16;
  // staging: thread -> (row tid >> 4 + 16 u, slot tid & 15): 16 threads move one row's 256 bytes
  const int s_row = tid >> 4, s_c = tid & 15;
  const uint8_t* ap[XPT];
  bool a_ok[XPT];
#pragma unroll
  for (int u = 0; u < XPT; ++u) {
    const int m = m0 + s_row + 16 * u;
    a_ok[u] = m < p.M;
    ap[u] = p.a + (int64_t)min(m, p.M - 1) * p.lda + k_begin + s_c * 16;
  }
  const int s_dst = s_row * 256 + ((s_c ^ (s_row & 15)) << 4);    // + u * 4096
  const int x_src = r * 256;                                       // + t * 4096 + (((4 ks + g) ^ r) << 4)

  typename A::type acc[NT][MT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[j][t] = A::zero();

  uint4 xs[XPT], w[D][4][NT];
#pragma unroll
  for (int u = 0; u < XPT; ++u) {
    xs[u] = a_ok[u] ? ld16(ap[u]) : make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(act[0] + s_dst + u * 4096) = xs[u];
  }
#pragma unroll
  for (int d = 0; d < D - 1; ++d)
    if (d < nch) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < NT; ++j) w[d][ks][j] = ld16(wp[j] + d * 256 + ks * 64);
    }
  __syncthreads();

  for (int c0 = 0; c0 < nch; c0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int c = c0 + d;
      if (c >= nch) break;   // uniform
      // requested before this chunk is multiplied: the next chunk's activations, the weights D - 1 chunks ahead (into the
      // ring slot the previous chunk left)
      const bool more = c + 1 < nch;
      if (more) {
#pragma unroll
        for (int u = 0; u < XPT; ++u) xs[u] = a_ok[u] ? ld16(ap[u] + (int64_t)(c + 1) * 256) : make_uint4(0, 0, 0, 0);
      }
      if (c + D - 1 < nch) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int j = 0; j < NT; ++j) w[(d + D - 1) % D][ks][j] = ld16(wp[j] + (int64_t)(c + D - 1) * 256 + ks * 64);
      }
      const uint8_t* buf = act[c & 1] + x_src;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        uint4 x[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) x[t] = *reinterpret_cast<const uint4*>(buf + t * 4096 + (((4 * ks + g) ^ r) << 4));
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[j][t] = A::mma(w[d][ks][j], x[t], acc[j][t]);
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < XPT; ++u) *reinterpret_cast<uint4*>(act[(c + 1) & 1] + s_dst + u * 4096) = xs[u];
      }
      __syncthreads();   // the other buffer is complete, and every wave is done with this one
    }
  }

  // epilogue (scaled_mm_kernel's, without the cross-wave sum): D[row = n_idx][col = m]; lane (m = l & 15, g) holds n = 16 j + 4 g + i
  using elem_t = decltype(acc[0][0][0] + acc[0][0][0]);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int nb = n0 + j * 16 + 4 * g;
    if (nb >= p.N) continue;
    if (p.splits > 1) {  // raw partial sums; scaled_mm_reduce_kernel finishes
      elem_t* slab = reinterpret_cast<elem_t*>(p.slab) + (int64_t)blockIdx.z * p.M * p.N;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int m = m0 + t * 16 + r;
        if (m >= p.M) continue;
        elem_t* dst = slab + (int64_t)m * p.N + nb;
        if (nb + 3 < p.N && (p.N & 3) == 0) {
          *reinterpret_cast<typename A::type*>(dst) = acc[j][t];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (nb + i < p.N) dst[i] = acc[j][t][i];
        }
      }
      continue;
    }
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
        // multiplies(a_scales, multiplies(b_scales, acc)); with a bias the outer node is multiply_add
        // (scaled_mm_c2x.cu:117-131, :157-171) -- no "+ 0" without one: it would turn -0 into +0
        const float tmp = bs[i] * (float)acc[j][t][i];
        o[i] = p.bias ? fmaf(as, tmp, bv[i]) : as * tmp;
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


// sums the split-K slabs in the accumulator's own type, then the same epilogue arithmetic
template <typename T, bool FP8>
__global__ __launch_bounds__(256) void scaled_mm_reduce_kernel(const MMParams p) {
  using elem_t = typename std::conditional<FP8, float, int>::type;
  typedef elem_t vec4_t __attribute__((ext_vector_type(4)));
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int n4 = (p.N + 3) / 4;
  if (idx >= (int64_t)p.M * n4) return;
  const int m = (int)(idx / n4), nb = (int)(idx % n4) * 4;
  const elem_t* src = reinterpret_cast<const elem_t*>(p.slab) + (int64_t)m * p.N + nb;
  const int64_t slab_stride = (int64_t)p.M * p.N;
  const float as = p.a_scales[p.a_per_row ? m : 0];
  uint16_t* dst = reinterpret_cast<uint16_t*>(p.out) + (int64_t)m * p.ldc + nb;
  elem_t v[4] = {0, 0, 0, 0};
  if (nb + 3 < p.N && (p.N & 3) == 0) {
    // whole 16-byte vectors, the loads of up to 8 slabs in flight together (the launch is latency, not bytes); the
    // additions stay in split order (fp8: the fp32 sum is the order-dependent one)
    for (int s0 = 0; s0 < p.splits; s0 += 8) {
      vec4_t q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (s0 + u < p.splits) q[u] = *reinterpret_cast<const vec4_t*>(src + (s0 + u) * slab_stride);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (s0 + u < p.splits) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += q[u][i];
        }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (nb + i >= p.N) break;
      for (int s = 0; s < p.splits; ++s) v[i] += src[s * slab_stride + i];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (nb + i >= p.N) break;
    const float bs = p.b_scales[p.b_per_col ? nb + i : 0];
    const float bv = p.bias ? T::to_float(reinterpret_cast<const uint16_t*>(p.bias)[nb + i]) : 0.f;
    const float tmp = bs * (float)v[i];
    dst[i] = T::from_float(p.bias ? fmaf(as, tmp, bv) : as * tmp);
  }
}

struct MMPlan {
  int mt, nt, un;  // tiles per wave, 64-byte k-steps in flight
  int splits, k_per_split;
  int wide;        // scaled_mm_wide_kernel (waves split N, activations through LDS)
};

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// Tile and split-K choice, by shape (Llama-3-8B projections measured on MI355X).
static MMPlan mm_plan(int M, int N, int K, int64_t scratch_bytes) {
  MMPlan pl{1, 1, 4, 1, K, 0};
  // 17 .. 64 rows, K in whole 256-byte chunks: the wide kernel where it was measured ahead (MI355X, int8, us,
  // tools/sweep_w8a8_wide.py -> profiles/r04_w8a8_wide.txt): many columns -- gate_up M = 64: 36.4 against 44.9 with two
  // column tiles per wave, unsliced (M = 32: 32.1 against 37.1 in two slices) -- and a long K with few columns -- down:
  // 23.6 against 29.0 with one tile per wave in eight slices.  A short K with few columns (qkv, o_proj) has too few
  // workgroups per byte either way and stays with the K-splitting kernel.  NMV_MM_WIDE=0 / NMV_MM_NT / NMV_MM_SPLITS force.
  if (M > 16 && M <= 64 && K % 256 == 0 && env_int("NMV_MM_WIDE", 1)) {
    const int forced_nt = env_int("NMV_MM_NT", 0), forced_s = env_int("NMV_MM_SPLITS", 0);
    int nt = 0, sp = 1;
    if (N >= 16384) {
      nt = 2;
      sp = M <= 32 ? 2 : 1;
    } else if (K >= 8192 && N <= 8192) {
      nt = 1;
      sp = 8;
    }
    if (forced_nt || forced_s) {   // development / tests: any valid form of the wide kernel
      nt = (forced_nt == 1 || forced_nt == 2 || forced_nt == 4) ? forced_nt : (nt ? nt : 2);
      sp = forced_s > 0 ? forced_s : sp;
      if ((sp & (sp - 1)) != 0 || sp > 8) nt = 0;
    }
    while (nt && sp > 1 && ((K / 256) % sp != 0 || K / sp < 512 || (int64_t)sp * M * N * 4 > scratch_bytes)) sp >>= 1;
    if (nt) {
      pl.wide = 1;
      pl.mt = M <= 32 ? 2 : 4;
      pl.nt = nt;
      pl.un = 4;
      pl.splits = sp;
      pl.k_per_split = K / sp;
      return pl;
    }
  }
  const int n16 = (N + 15) / 16;
  int splits = 1;
  if (M <= 32) {
    // M = 1: one column tile per wave is pure weight streaming (4.8 TB/s on gate_up).  With more rows every
    // workgroup re-reads the whole activation block from L2 -- at 16 columns per workgroup that is as many
    // bytes as the weights -- so wider column tiles (the fragment is reused NT times) win as soon as they
    // leave enough workgroups
    pl.mt = M <= 16 ? 1 : 2;
    // measured (int8, us, NT = 1 / 2 / 4): gate_up M=8 30.3 / 31.5 / 32.0, M=16 36.0 / 35.0 / 33.7, M=32 48.3 / 42.7 /
    // 36.3; qkv M=32 16.0 / 14.3 / 19.2; o and down (256 column tiles) are fastest at NT = 1 up to M = 32
    pl.nt = 1;
    if (n16 >= 1024 && M > 8) pl.nt = 4;
    else if (n16 >= 384 && n16 < 1024 && M > 16) pl.nt = 2;
    const int nt = env_int("NMV_MM_NT", 0);
    if (nt == 1 || nt == 2 || nt == 4) pl.nt = nt;
    pl.un = pl.nt == 4 ? 2 : 4;
  } else {
    // Every activation fragment a wave loads is reused for NT column tiles, so the L2 traffic for
    // activations is 4/NT x the weight bytes: take the widest NT that still leaves >= 192
    // workgroups; a long K (down_proj) keeps NT = 4 and reaches that count with split-K slabs
    // instead (the reduce launch costs ~3 us, which a short K does not repay).
    const int m_blocks = (M + 63) / 64;
    pl.mt = 4;
    for (pl.nt = 4; pl.nt > 1; pl.nt >>= 1) {
      const int base = ((n16 + pl.nt - 1) / pl.nt) * m_blocks;
      if (base >= 192) break;
      if (pl.nt == 4 && K >= 8192) {
        while (splits < 4 && base * splits < 256) splits *= 2;
        break;
      }
    }
    const int nt = env_int("NMV_MM_NT", 0);
    if (nt == 1 || nt == 2 || nt == 4) pl.nt = nt;
    splits = max(1, env_int("NMV_MM_SPLITS", splits));
    pl.un = pl.nt == 1 ? 4 : 2;
  }
  if ((int64_t)splits * M * N * 4 > scratch_bytes) splits = 1;
  int kps = (K + splits - 1) / splits;
  kps = (kps + 63) / 64 * 64;
  pl.splits = (K + kps - 1) / kps;
  pl.k_per_split = kps;
  return pl;
}

template <typename T, bool FP8>
static void launch_mm(MMParams p, const MMPlan& pl, hipStream_t s) {
  p.splits = pl.splits;
  p.k_per_split = pl.k_per_split;
  if (pl.wide) {
#define MMW_CASE(MT_, NT_)                                                                            \
    if (pl.mt == MT_ && pl.nt == NT_) {                                                               \
      dim3 grid((p.N + 64 * NT_ - 1) / (64 * NT_), (p.M + 16 * MT_ - 1) / (16 * MT_), pl.splits);     \
      hipLaunchKernelGGL((scaled_mm_wide_kernel<T, FP8, MT_, NT_>), grid, dim3(MM_THREADS), 0, s, p); \
    }
    MMW_CASE(2, 1)
    MMW_CASE(2, 2)
    MMW_CASE(2, 4)
    MMW_CASE(4, 1)
    MMW_CASE(4, 2)
    MMW_CASE(4, 4)
#undef MMW_CASE
    if (pl.splits > 1) {
      const int64_t items = (int64_t)p.M * ((p.N + 3) / 4);
      hipLaunchKernelGGL((scaled_mm_reduce_kernel<T, FP8>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, p);
    }
    return;
  }
#define MM_CASE(MT_, NT_, UN_)                                                                       \
  if (pl.mt == MT_ && pl.nt == NT_) {                                                                \
    dim3 grid((p.N + 16 * NT_ - 1) / (16 * NT_), (p.M + 16 * MT_ - 1) / (16 * MT_), pl.splits);      \
    hipLaunchKernelGGL((scaled_mm_kernel<T, FP8, MT_, NT_, UN_>), grid, dim3(MM_THREADS), 0, s, p);  \
  }
  MM_CASE(1, 1, 4)
  MM_CASE(1, 2, 4)
  MM_CASE(1, 4, 2)
  MM_CASE(2, 1, 4)
  MM_CASE(2, 2, 4)
  MM_CASE(2, 4, 2)
  MM_CASE(4, 1, 4)
  MM_CASE(4, 2, 2)
  MM_CASE(4, 4, 2)
#undef MM_CASE
  if (pl.splits > 1) {
    const int64_t items = (int64_t)p.M * ((p.N + 3) / 4);
    hipLaunchKernelGGL((scaled_mm_reduce_kernel<T, FP8>), dim3((unsigned)((items + 255) / 256)),
                       dim3(256), 0, s, p);
  }
}

}  // namespace nmv

using namespace nmv;

extern "C" int nmv_cutlass_scaled_mm_supports_fp8(int64_t cuda_device_capability) {
  // scaled_mm_entry.cu:32-46 gates fp8 on sm89+; gfx950 (capability 95) has native OCP fp8 MFMA
  return cuda_device_capability >= 89 ? 1 : 0;
}

// split-K slabs (M > 32): at most 4 x [M, N] x 4 bytes; without scratch (null) the kernels run
// unsplit
extern "C" int64_t nmv_scaled_mm_scratch_bytes(int M, int N, int K) {
  const MMPlan pl = mm_plan(M, N, K, INT64_MAX);
  return pl.splits > 1 ? (int64_t)pl.splits * M * N * 4 : 0;
}

extern "C" int nmv_scaled_mm(void* out, const void* a, const void* b, const float* a_scales,
                             const float* b_scales, const void* bias, int M, int N, int K,
                             int64_t lda, int64_t ldb, int64_t ldc, int a_scales_numel,
                             int b_scales_numel, nmv_q8_dtype_t in_dtype, nmv_dtype_t out_dtype,
                             void* scratch, int64_t scratch_bytes, void* stream) {
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
             b_scales_numel == N && N > 1 ? 1 : (b_scales_numel == N ? 1 : 0), 1, K, scratch};
  hipStream_t s = (hipStream_t)stream;
  const MMPlan pl = mm_plan(M, N, K, scratch ? scratch_bytes : 0);
  if (in_dtype == NMV_I8) {
    if (out_dtype == NMV_F16) launch_mm<F16, false>(p, pl, s); else launch_mm<BF16, false>(p, pl, s);
  } else {
    if (out_dtype == NMV_F16) launch_mm<F16, true>(p, pl, s); else launch_mm<BF16, true>(p, pl, s);
  }
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}
