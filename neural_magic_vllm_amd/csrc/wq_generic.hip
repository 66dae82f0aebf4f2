// Generic weight-only-quantised GEMM for the checkpoint formats that are not (yet) served by the
// tuned Marlin kernel of w4a16_gemm.hip:
//   * GPTQ / exllama  (gptq_gemm, gptq_shuffle)      reference csrc/quantization/gptq/q_gemm.cu:191-327,
//                                                     :1387-1419 (reconstruct), :1823-1856
//   * AWQ             (awq_gemm, awq_dequantize)      reference csrc/quantization/awq/gemm_kernels.cu:29-549,
//                                                     dequantize.cuh:20-101
//   * Marlin 8-bit, Marlin act-order on a K shard (is_k_full = False), fp8-Marlin
//                                                     reference gptq_marlin.cu:396-1363, fp8_marlin.cu:1212-1308
// One kernel, one format functor per layout: a workgroup owns 64 output columns; every 32-deep k
// step its 256 threads dequantise a 64 x 32 weight tile (one thread = 8 consecutive k of one
// column = exactly one MFMA operand) into LDS in MFMA-A-operand order, each wave then multiplies
// its 16 columns against all (<= 64) rows with v_mfma_f32_16x16x32; activations come straight
// from global memory in natural k order.  w = (q - z) * s is rounded to the model dtype, fp32
// accumulation, one rounding of the result -- the reference's reconstruct-then-GEMM numerics
// (q_gemm.cu:1496-1499, awq.py:166-170).  Correctness-first path: bounded by the LDS round trip,
// not tuned to the HBM roofline like the Marlin kernel; DESIGN.md section 3.5.
#include <algorithm>

#include "common.h"

namespace nmv {

enum WqFormat { WQ_GPTQ = 0, WQ_AWQ = 1, WQ_MARLIN = 2, WQ_MARLIN_FP8 = 3 };

struct WqParams {
  const uint16_t* a;        // [M, K]
  const uint32_t* qweight;  // format specific
  const uint32_t* qzeros;   // GPTQ [G, N/pack] (stored zero-1) / AWQ [G, N/8]; null = symmetric
  const uint16_t* scales;   // [G, N] (natural order, or marlin_permute_scales order for WQ_MARLIN*)
  const int* g_idx;         // [K] group of every (possibly permuted) row, or null -> k / group_size
  const int* perm;          // [K] activation column gather, or null
  uint16_t* c;              // [M, N]
  int M, N, K, bits, group_size, num_groups;
};

// integer code of element (k, n) -------------------------------------------------------------
template <int FMT>
__device__ __forceinline__ uint32_t wq_code(const WqParams& p, int k, int n) {
  const int bits = p.bits;
  const uint32_t mask = (1u << bits) - 1;
  if constexpr (FMT == WQ_GPTQ) {
    if (bits == 3) {
      // 32 consecutive k of a column are a 96-bit little-endian stream over three int32 rows: code i sits at
      // bits [3 i, 3 i + 3), codes 10 and 21 straddle a word (q_gemm.cu:1438-1459)
      const int bit0 = (k & 31) * 3, w = bit0 >> 5, sh = bit0 & 31;
      const uint32_t* col = p.qweight + (int64_t)((k >> 5) * 3) * p.N + n;
      const uint64_t lo = col[(int64_t)w * p.N];
      const uint64_t hi = (sh > 29) ? col[(int64_t)(w + 1) * p.N] : 0;
      return (uint32_t)(((hi << 32) | lo) >> sh) & 7u;
    }
    // qweight [K/pack, N]: `pack` consecutive k share one int32, low bits first (quant_utils.py:125-146)
    const int pack = 32 / bits;
    return (p.qweight[(int64_t)(k / pack) * p.N + n] >> (bits * (k % pack))) & mask;
  } else if constexpr (FMT == WQ_AWQ) {
    // qweight [K, N/8]: column c of a word sits at nibble {0,4,1,5,2,6,3,7}[c] (dequantize.cuh:31-62)
    const int c = n & 7;
    const int nib = ((c & 1) << 2) | (c >> 1);
    return (p.qweight[(int64_t)k * (p.N / 8) + (n >> 3)] >> (4 * nib)) & 0xf;
  } else {
    // Marlin tile order (marlin_perms.py:16-43), inverse of csrc/w4a16_gemm.hip's header comment
    const int kt = k >> 4, k_in = k & 15;
    const int chunk = n >> 6, c64 = n & 63;
    const int j = c64 >> 4, blk = (c64 >> 3) & 1, n_in = c64 & 7;
    const int q = (k_in & 7) >> 1, odd = k_in & 1, hi8 = k_in >> 3;
    const int i = n_in * 4 + q;
    if (bits == 4) {
      const int pz = (odd << 2) | (blk << 1) | hi8;
      return (p.qweight[(int64_t)kt * (p.N * 2) + chunk * 128 + i * 4 + j] >> (4 * pz)) & 0xf;
    }
    const int pz = (odd << 1) | hi8;
    return (p.qweight[(int64_t)kt * (p.N * 4) + chunk * 256 + i * 8 + j * 2 + blk] >> (8 * pz)) & 0xff;
  }
}

template <typename T, int FMT>
__device__ __forceinline__ float wq_scale(const WqParams& p, int g, int n) {
  if constexpr (FMT == WQ_MARLIN || FMT == WQ_MARLIN_FP8) {
    int pos;
    if (p.num_groups > 1) { const int c = n & 63; pos = (n & ~63) + (c & 7) * 8 + (c >> 3); }
    else { const int c = n & 31; pos = (n & ~31) + ((c & 7) >> 1) * 8 + 2 * (c >> 3) + (c & 1); }
    return T::to_float(p.scales[(int64_t)g * p.N + pos]);
  } else {
    return T::to_float(p.scales[(int64_t)g * p.N + n]);
  }
}

template <int FMT>
__device__ __forceinline__ float wq_zero(const WqParams& p, int g, int n) {
  if constexpr (FMT == WQ_GPTQ) {
    if (p.qzeros == nullptr) return (float)(1 << (p.bits - 1));
    if (p.bits == 3) {
      // the same 3-bit stream along N: 32 columns per three int32 (matrix_view.cuh:204-232)
      const int bit0 = (n & 31) * 3, w = bit0 >> 5, sh = bit0 & 31;
      const uint32_t* row = p.qzeros + (int64_t)g * (p.N * 3 / 32) + (n >> 5) * 3;
      const uint64_t lo = row[w];
      const uint64_t hi = (sh > 29) ? row[w + 1] : 0;
      return (float)(((uint32_t)(((hi << 32) | lo) >> sh) & 7u) + 1);
    }
    const int pack = 32 / p.bits;
    const uint32_t z = (p.qzeros[(int64_t)g * (p.N / pack) + n / pack] >> (p.bits * (n % pack))) & ((1u << p.bits) - 1);
    return (float)(z + 1);  // GPTQ stores zero - 1 (q_gemm.cu:1410-1417)
  } else if constexpr (FMT == WQ_AWQ) {
    const int c = n & 7;
    const int nib = ((c & 1) << 2) | (c >> 1);
    return (float)((p.qzeros[(int64_t)g * (p.N / 8) + (n >> 3)] >> (4 * nib)) & 0xf);
  } else {
    return (float)(1 << (p.bits - 1));  // symmetric Marlin: 8 or 128
  }
}

// dequantised weight, rounded to the model dtype
template <typename T, int FMT>
__device__ __forceinline__ float wq_weight(const WqParams& p, int k, int n) {
  const int g = p.g_idx ? p.g_idx[k] : (p.group_size > 0 ? k / p.group_size : 0);
  const uint32_t code = wq_code<FMT>(p, k, n);
  const float s = wq_scale<T, FMT>(p, g, n);
  if constexpr (FMT == WQ_MARLIN_FP8) {
    return T::to_float(T::from_float(fp8_to_f32((uint8_t)code) * s));
  } else {
    return T::to_float(T::from_float(((float)code - wq_zero<FMT>(p, g, n)) * s));
  }
}

template <typename T> struct Mfma16;
template <> struct Mfma16<BF16> {
  static __device__ __forceinline__ f32x4_t run(uint4 w, uint4 a, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, a), c, 0, 0, 0);
  }
};
template <> struct Mfma16<F16> {
  static __device__ __forceinline__ f32x4_t run(uint4 w, uint4 a, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w), __builtin_bit_cast(f16x8_t, a), c, 0, 0, 0);
  }
};

// ---- the same element definitions, word by word: a thread's dequantisation cell is 8 consecutive k of one
// column; its packed words, group ids, scale and zero point are fetched as raw words AHEAD of their use (the
// generic kernel below runs them through a three-deep register pipeline), and decoded when they are consumed ----
struct WqRaw {
  uint32_t w[8];   // packed words holding the 8 codes (GPTQ 1-2, Marlin 4, AWQ 8)
  int gi[8];       // group of every k (g_idx formats), else gi[0] only
  bool uniform;    // the 8 k share one group (always without g_idx): scale / zero are those of gi[0]
};
struct WqSz { uint32_t sw; uint32_t zw[2]; };   // raw scale element and zero-point word(s) of (group, column)

template <int FMT>
__device__ __forceinline__ void wq_load_raw(const WqParams& p, int k8, int n, WqRaw& r) {
#pragma unroll
  for (int i = 0; i < 8; ++i) r.w[i] = 0;
  if constexpr (FMT == WQ_GPTQ) {
    if (p.bits == 4) {
      r.w[0] = p.qweight[(int64_t)(k8 >> 3) * p.N + n];
    } else if (p.bits == 8) {
      r.w[0] = p.qweight[(int64_t)(k8 >> 2) * p.N + n];
      r.w[1] = p.qweight[(int64_t)((k8 >> 2) + 1) * p.N + n];
    } else if (p.bits == 2) {
      r.w[0] = p.qweight[(int64_t)(k8 >> 4) * p.N + n];
    } else {  // 3 bits: the cell is 24 bits of the 96-bit stream of its 32-k run, from bit 0 / 24 / 48 / 72
      const int wi = ((k8 & 31) * 3) >> 5;
      const uint32_t* col = p.qweight + (int64_t)((k8 >> 5) * 3) * p.N + n;
      r.w[0] = col[(int64_t)wi * p.N];
      r.w[1] = wi < 2 ? col[(int64_t)(wi + 1) * p.N] : 0;
    }
  } else if constexpr (FMT == WQ_AWQ) {
#pragma unroll
    for (int e = 0; e < 8; ++e) r.w[e] = p.qweight[(int64_t)(k8 + e) * (p.N / 8) + (n >> 3)];
  } else {
    const int kt = k8 >> 4, chunk = n >> 6, c64 = n & 63;
    const int j = c64 >> 4, blk = (c64 >> 3) & 1, n_in = c64 & 7;
    if (p.bits == 4) {
      const uint32_t* base = p.qweight + (int64_t)kt * (p.N * 2) + chunk * 128 + n_in * 16 + j;
#pragma unroll
      for (int q = 0; q < 4; ++q) r.w[q] = base[q * 4];
    } else {
      const uint32_t* base = p.qweight + (int64_t)kt * (p.N * 4) + chunk * 256 + n_in * 32 + j * 2 + blk;
#pragma unroll
      for (int q = 0; q < 4; ++q) r.w[q] = base[q * 8];
    }
  }
  if (p.g_idx) {
    const int4 g0 = *reinterpret_cast<const int4*>(p.g_idx + k8);
    const int4 g1 = *reinterpret_cast<const int4*>(p.g_idx + k8 + 4);
    r.gi[0] = g0.x; r.gi[1] = g0.y; r.gi[2] = g0.z; r.gi[3] = g0.w;
    r.gi[4] = g1.x; r.gi[5] = g1.y; r.gi[6] = g1.z; r.gi[7] = g1.w;
    r.uniform = (g0.x == g0.y) & (g0.x == g0.z) & (g0.x == g0.w) & (g0.x == g1.x) & (g0.x == g1.y) &
                (g0.x == g1.z) & (g0.x == g1.w);
  } else {
    r.gi[0] = p.group_size > 0 ? k8 / p.group_size : 0;   // groups are multiples of 8 k
    r.uniform = true;
  }
}

// code of element e (k = k8 + e) of the cell
template <int FMT>
__device__ __forceinline__ uint32_t wq_raw_code(const WqParams& p, const WqRaw& r, int k8, int n, int e) {
  if constexpr (FMT == WQ_GPTQ) {
    if (p.bits == 4) return (r.w[0] >> (4 * e)) & 0xf;
    if (p.bits == 8) return (r.w[e >> 2] >> (8 * (e & 3))) & 0xff;
    if (p.bits == 2) return (r.w[0] >> (2 * ((k8 & 15) + e))) & 3u;
    const uint64_t v = ((uint64_t)r.w[1] << 32) | r.w[0];
    return (uint32_t)(v >> ((((k8 & 31) * 3) & 31) + 3 * e)) & 7u;
  } else if constexpr (FMT == WQ_AWQ) {
    const int c = n & 7;
    return (r.w[e] >> (4 * (((c & 1) << 2) | (c >> 1)))) & 0xf;
  } else {
    const int blk = (n >> 3) & 1, hi8 = (k8 >> 3) & 1, odd = e & 1;
    if (p.bits == 4) return (r.w[e >> 1] >> (4 * ((odd << 2) | (blk << 1) | hi8))) & 0xf;
    return (r.w[e >> 1] >> (8 * ((odd << 1) | hi8))) & 0xff;
  }
}

template <int FMT>
__device__ __forceinline__ void wq_load_sz(const WqParams& p, int g, int n, WqSz& z) {
  int pos = n;
  if constexpr (FMT == WQ_MARLIN || FMT == WQ_MARLIN_FP8) {
    if (p.num_groups > 1) { const int c = n & 63; pos = (n & ~63) + (c & 7) * 8 + (c >> 3); }
    else { const int c = n & 31; pos = (n & ~31) + ((c & 7) >> 1) * 8 + 2 * (c >> 3) + (c & 1); }
  }
  z.sw = p.scales[(int64_t)g * p.N + pos];
  z.zw[0] = z.zw[1] = 0;
  if constexpr (FMT == WQ_GPTQ) {
    if (p.qzeros != nullptr) {
      if (p.bits == 3) {
        const int wi = ((n & 31) * 3) >> 5;
        const uint32_t* row = p.qzeros + (int64_t)g * (p.N * 3 / 32) + (n >> 5) * 3;
        z.zw[0] = row[wi];
        z.zw[1] = wi < 2 ? row[wi + 1] : 0;
      } else {
        z.zw[0] = p.qzeros[(int64_t)g * (p.N / (32 / p.bits)) + n / (32 / p.bits)];
      }
    }
  } else if constexpr (FMT == WQ_AWQ) {
    z.zw[0] = p.qzeros[(int64_t)g * (p.N / 8) + (n >> 3)];
  }
}

template <typename T, int FMT>
__device__ __forceinline__ void wq_decode_sz(const WqParams& p, const WqSz& z, int n, float& scale, float& zero) {
  scale = T::to_float((uint16_t)z.sw);
  if constexpr (FMT == WQ_GPTQ) {
    if (p.qzeros == nullptr) zero = (float)(1 << (p.bits - 1));
    else if (p.bits == 3) {
      const uint64_t v = ((uint64_t)z.zw[1] << 32) | z.zw[0];
      zero = (float)(((uint32_t)(v >> (((n & 31) * 3) & 31)) & 7u) + 1);
    } else {
      const int pack = 32 / p.bits;
      zero = (float)(((z.zw[0] >> (p.bits * (n % pack))) & ((1u << p.bits) - 1)) + 1);
    }
  } else if constexpr (FMT == WQ_AWQ) {
    const int c = n & 7;
    zero = (float)((z.zw[0] >> (4 * (((c & 1) << 2) | (c >> 1)))) & 0xf);
  } else {
    zero = (float)(1 << (p.bits - 1));
  }
}

// Generic kernel: any format of this file, any group structure (g_idx runs of any length), M <= 64 rows per
// workgroup.  grid = (N / 64, split-K, M blocks); K is split across workgroups into fp32 slabs that
// wq_reduce_kernel sums in split order (slab == nullptr: one split, direct store).  A workgroup's four waves
// dequantise a 64-column x 32-k tile per step into LDS in MFMA-operand order (w = (q - z) * s rounded to the
// model dtype: the reference's reconstruct-then-GEMM numerics) and multiply it against all its rows.  Every
// global operand is fetched ahead of its use: the packed words and group ids three steps ahead, the scale / zero
// point of a cell whose 8 k share a group two steps ahead, the activation fragment one step ahead; only cells
// that straddle a group boundary look their scales up element by element.
template <typename T, int FMT, int MT>
__global__ __launch_bounds__(256) void wq_gemm_kernel(const WqParams p, float* __restrict__ slab, int k_per_wg) {
  // [2 buffers][64 columns][32 k] in the model dtype; row = one column's 32 k values (64 B)
  __shared__ __attribute__((aligned(16))) uint16_t w_s[2][64][32 + 8];  // +8: spread rows over banks
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 64;
  const int m0 = blockIdx.z * (16 * MT);
  const int k_lo = blockIdx.y * k_per_wg;
  const int k_hi = min(k_lo + k_per_wg, p.K);
  // dequantisation cell of this thread: column dn, k group dk (8 consecutive k)
  const int dn = threadIdx.x & 63, dk = threadIdx.x >> 6;
  const int n_d = min(n0 + dn, p.N - 1);

  f32x4_t acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int ksteps = (k_hi - k_lo) / 32;
  auto cell_k = [&](int ks) { return k_lo + min(ks, ksteps - 1) * 32 + dk * 8; };  // look-ahead past the end re-reads
  auto load_sz = [&](const WqRaw& raw, WqSz& z) {
    if (raw.uniform) wq_load_sz<FMT>(p, raw.gi[0], n_d, z);
  };
  auto dequant = [&](int buf, int ks, const WqRaw& raw, const WqSz& z) {
    const int k8 = cell_k(ks);
    float sc = 0.f, zr = 0.f;
    if (raw.uniform) wq_decode_sz<T, FMT>(p, z, n_d, sc, zr);
    float wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (!raw.uniform) {  // the cell straddles a group boundary (act-order runs)
        WqSz ze;
        wq_load_sz<FMT>(p, raw.gi[e], n_d, ze);
        wq_decode_sz<T, FMT>(p, ze, n_d, sc, zr);
      }
      const uint32_t code = wq_raw_code<FMT>(p, raw, k8, n_d, e);
      if constexpr (FMT == WQ_MARLIN_FP8) wv[e] = T::to_float(T::from_float(fp8_to_f32((uint8_t)code) * sc));
      else wv[e] = T::to_float(T::from_float(((float)code - zr) * sc));
    }
    *reinterpret_cast<uint4*>(&w_s[buf][dn][dk * 8]) =
        make_uint4(T::pack2(wv[0], wv[1]), T::pack2(wv[2], wv[3]), T::pack2(wv[4], wv[5]), T::pack2(wv[6], wv[7]));
  };
  // activation fragments: rows m0 + 16 t + r, k = k0 + 8 g .. + 8 (act-order: the caller has gathered the columns)
  auto load_a = [&](int ks, uint4 (&af)[MT]) {
    const int k0 = k_lo + min(ks, ksteps - 1) * 32;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int m = m0 + t * 16 + r;
      af[t] = m < p.M ? ld16(p.a + (int64_t)m * p.K + k0 + g * 8) : make_uint4(0, 0, 0, 0);
    }
  };

  if (ksteps > 0) {
    WqRaw ra, rb, rc;          // raw cells of steps ks+1, ks+2, ks+3
    WqSz sa, sb;               // scale / zero of steps ks+1, ks+2
    uint4 af[MT], af_next[MT];
    {
      WqRaw r0;
      WqSz s0;
      wq_load_raw<FMT>(p, cell_k(0), n_d, r0);
      wq_load_raw<FMT>(p, cell_k(1), n_d, ra);
      wq_load_raw<FMT>(p, cell_k(2), n_d, rb);
      load_a(0, af);
      load_sz(r0, s0);
      load_sz(ra, sa);
      dequant(0, 0, r0, s0);
    }
    __syncthreads();
    for (int ks = 0; ks < ksteps; ++ks) {
      const int buf = ks & 1;
      wq_load_raw<FMT>(p, cell_k(ks + 3), n_d, rc);
      load_sz(rb, sb);
      load_a(ks + 1, af_next);
      if (ks + 1 < ksteps) dequant(buf ^ 1, ks + 1, ra, sa);
      const uint4 wf = *reinterpret_cast<const uint4*>(&w_s[buf][wave * 16 + r][g * 8]);
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = Mfma16<T>::run(wf, af[t], acc[t]);
      __syncthreads();
      ra = rb; rb = rc; sa = sb;
#pragma unroll
      for (int t = 0; t < MT; ++t) af[t] = af_next[t];
    }
  }
  // D[row = column index][col = m]: lane (m = r, g) holds columns n0 + 16 wave + 4 g + i
  const int nb = n0 + wave * 16 + 4 * g;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = m0 + t * 16 + r;
    if (m >= p.M) continue;
    if (slab != nullptr) {
      float* dst = slab + ((int64_t)blockIdx.y * p.M + m) * p.N + nb;
      if (nb + 3 < p.N) *reinterpret_cast<f32x4_t*>(dst) = acc[t];
      else
#pragma unroll
        for (int i = 0; i < 4; ++i) if (nb + i < p.N) dst[i] = acc[t][i];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (nb + i < p.N) p.c[(int64_t)m * p.N + nb + i] = T::from_float(acc[t][i]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Streaming kernel for the two checkpoint layouts that matter in practice: 4-bit GPTQ without
// act-order (exllama layout: int32 [K/8, N], 8 consecutive k of one column per word) and AWQ
// (int32 [K, N/8], 8 columns of one k per word, nibble order {0,4,1,5,2,6,3,7}).
// The packed words are streamed ONCE with 16-byte coalesced loads into a raw LDS image
// (double-buffered, 128 k x 64 columns = 4 KiB per stage); each lane pulls the 8 codes of its MFMA
// operand out of LDS (GPTQ: one word; AWQ: one nibble of 8 words, a broadcast read) and expands
// them with the exponent trick of w4a16_gemm.hip -- a nibble dropped into the top mantissa bits
// of a bf16/fp16 with exponent 2^4 is exactly 16 + q.  Codes i and i+4 of a GPTQ word are 16 bits
// apart, so ONE rotate + ONE v_and_or_b32 yields the pair (k_i, k_i+4): the operand's k order is
// (0,4,1,5,2,6,3,7) and the activation fragment (natural order from global memory) is brought to
// the same order with four v_perm_b32.  Zero point and scale are applied in fp32 per group on the
// accumulators:  acc += s[n] * (acc_group - (16 + z[n]) * S_group[m]),  S from an all-ones MFMA
// (more accurate than the reference's reconstruct-to-half-then-GEMM, q_gemm.cu:1496-1499).
// K is split over workgroups (fp32 slabs summed in split order by wq_reduce_kernel: deterministic).
constexpr int WS_K = 128;  // k per stage

template <typename T> struct WqMagic;
template <> struct WqMagic<BF16> {   // nibble -> mantissa bits [6:3] of each half, value 16 + q
  static constexpr uint32_t MASK = 0x00780078u, MAGIC = 0x41804180u, ONES = 0x3F803F80u;
  static constexpr int POS = 3;
};
template <> struct WqMagic<F16> {    // mantissa bits [9:6]
  static constexpr uint32_t MASK = 0x03C003C0u, MAGIC = 0x4C004C00u, ONES = 0x3C003C00u;
  static constexpr int POS = 6;
};

template <typename T, int FMT, int MT>
__global__ __launch_bounds__(256) void wq_stream_kernel(const WqParams p, float* __restrict__ slab,
                                                        int k_per_wg) {
  static_assert(FMT == WQ_GPTQ || FMT == WQ_AWQ, "4-bit GPTQ / AWQ only");
  using MG = WqMagic<T>;
  __shared__ __attribute__((aligned(16))) uint32_t raw[2][1024];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 64;
  const int m0 = blockIdx.z * (16 * MT);
  const int kb = blockIdx.y * k_per_wg;
  const int ke = min(kb + k_per_wg, p.K);
  const int n_stages = (ke - kb) / WS_K;
  const int col = wave * 16 + r;       // weight column of this lane (MFMA A-operand row)
  const int nout = n0 + wave * 16 + 4 * g;   // first of this lane's 4 OUTPUT columns

  // ---- raw stage loader: one 16-byte piece per thread ----
  const uint32_t* src;
  int64_t src_stage_stride;
  int lds_idx;
  if constexpr (FMT == WQ_GPTQ) {
    src = p.qweight + (int64_t)(kb / 8 + (threadIdx.x >> 4)) * p.N + n0 + (threadIdx.x & 15) * 4;
    src_stage_stride = (int64_t)(WS_K / 8) * p.N;
    lds_idx = (threadIdx.x >> 4) * 64 + (threadIdx.x & 15) * 4;     // [16 word rows][64 columns]
  } else {
    src = p.qweight + (int64_t)(kb + (threadIdx.x >> 1)) * (p.N / 8) + n0 / 8 + (threadIdx.x & 1) * 4;
    src_stage_stride = (int64_t)WS_K * (p.N / 8);
    lds_idx = (threadIdx.x >> 1) * 8 + (threadIdx.x & 1) * 4;       // [128 k rows][8 words]
  }
  const uint32_t kmask = __builtin_amdgcn_readfirstlane(MG::MASK);
  uint32_t kmagic = MG::MAGIC;
  asm volatile("" : "+v"(kmagic));   // keep mask in an SGPR and magic in a VGPR: v_and_or_b32
  // AWQ: rotate amounts that bring this column's nibble to the low / high half's mantissa field
  const int awq_sh = 4 * (((col & 1) << 2) | ((col & 7) >> 1));
  const uint32_t awq_rlo = (uint32_t)(awq_sh - MG::POS) & 31u;
  const uint32_t awq_rhi = (uint32_t)(awq_sh - MG::POS - 16) & 31u;
  const uint32_t mask_lo = kmask & 0xffffu, mask_hi = kmask & 0xffff0000u;
  const uint16_t* a_row[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
    a_row[t] = p.a + (int64_t)min(m0 + t * 16 + r, p.M - 1) * p.K + kb + g * 8;

  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4_t acc[MT], accg[MT], accs[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) { acc[t] = zero4; accg[t] = zero4; accs[t] = zero4; }
  const uint4 ones = make_uint4(MG::ONES, MG::ONES, MG::ONES, MG::ONES);
  const int flush_ks = min(p.group_size, WS_K) / 32;   // 1, 2 or 4 k-steps per scale group (uniform)

  // acc += s[n] * (accg - (16 + z[n]) * S[m]) for this lane's 4 output columns, then reset
  auto flush = [&](int k_abs) {
    const int grp = k_abs / p.group_size;
    const uint2 s2 = *reinterpret_cast<const uint2*>(p.scales + (int64_t)grp * p.N + nout);
    const uint32_t zw = p.qzeros ? p.qzeros[(int64_t)grp * (p.N / 8) + (nout >> 3)] : 0x77777777u;
    const float sc[4] = {lo_f<T>(s2.x), hi_f<T>(s2.x), lo_f<T>(s2.y), hi_f<T>(s2.y)};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = (nout & 7) + i;   // column inside the zero-point word (nout % 8 is 0 or 4)
      int z;
      if constexpr (FMT == WQ_GPTQ) z = (int)((zw >> (4 * c)) & 0xf) + 1;   // stored as zero - 1
      else z = (int)((zw >> (4 * (((c & 1) << 2) | (c >> 1)))) & 0xf);
      const float zc = -(16.0f + (float)z);
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t][i] = fmaf(sc[i], fmaf(zc, accs[t][0], accg[t][i]), acc[t][i]);
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) { accg[t] = zero4; accs[t] = zero4; }
  };

  uint4 nxt = make_uint4(0, 0, 0, 0);
  if (n_stages > 0) {
    const uint4 first = *reinterpret_cast<const uint4*>(src);
    *reinterpret_cast<uint4*>(&raw[0][lds_idx]) = first;
  }
  __syncthreads();
  for (int st = 0; st < n_stages; ++st) {
    const int buf = st & 1;
    const int k0 = kb + st * WS_K;
    if (st + 1 < n_stages) nxt = *reinterpret_cast<const uint4*>(src + (st + 1) * src_stage_stride);
    uint4 af[4][MT];   // activation fragments of the 4 k-steps: all loads issued up front
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int t = 0; t < MT; ++t) af[ks][t] = ld16(a_row[t] + st * WS_K + ks * 32);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 wf;
      if constexpr (FMT == WQ_GPTQ) {
        const uint32_t x = raw[buf][(ks * 4 + g) * 64 + col];
        // code i sits at bit 4 i and must reach bit POS: rotate right by (4 i - POS) mod 32; code
        // i + 4 follows 16 bits higher into the upper half
        wf = make_uint4((__builtin_amdgcn_alignbit(x, x, (0 - MG::POS) & 31) & kmask) | kmagic,
                        (__builtin_amdgcn_alignbit(x, x, (4 - MG::POS) & 31) & kmask) | kmagic,
                        (__builtin_amdgcn_alignbit(x, x, (8 - MG::POS) & 31) & kmask) | kmagic,
                        (__builtin_amdgcn_alignbit(x, x, (12 - MG::POS) & 31) & kmask) | kmagic);
      } else {
        uint32_t w8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w8[i] = raw[buf][(ks * 32 + g * 8 + i) * 8 + (col >> 3)];
        uint32_t d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t lo = (__builtin_amdgcn_alignbit(w8[i], w8[i], awq_rlo) & mask_lo) | kmagic;
          d[i] = (__builtin_amdgcn_alignbit(w8[i + 4], w8[i + 4], awq_rhi) & mask_hi) | lo;
        }
        wf = make_uint4(d[0], d[1], d[2], d[3]);
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        // activations to the operand's k order (0,4,1,5,2,6,3,7)
        const uint4 x = af[ks][t];
        const uint4 ap = make_uint4(__builtin_amdgcn_perm(x.z, x.x, 0x05040100u), __builtin_amdgcn_perm(x.z, x.x, 0x07060302u),
                                    __builtin_amdgcn_perm(x.w, x.y, 0x05040100u), __builtin_amdgcn_perm(x.w, x.y, 0x07060302u));
        accg[t] = Mfma16<T>::run(wf, ap, accg[t]);
        accs[t] = Mfma16<T>::run(ones, ap, accs[t]);
      }
      if ((ks + 1) % flush_ks == 0) flush(k0 + ks * 32);   // uniform
    }
    if (st + 1 < n_stages) *reinterpret_cast<uint4*>(&raw[buf ^ 1][lds_idx]) = nxt;
    __syncthreads();
  }
  // D[row = column index][col = m]: lane (m = r, g) holds columns nout + i
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = m0 + t * 16 + r;
    if (m >= p.M) continue;
    if (gridDim.y == 1) {
      uint2 pk;
      pk.x = T::pack2(acc[t][0], acc[t][1]);
      pk.y = T::pack2(acc[t][2], acc[t][3]);
      *reinterpret_cast<uint2*>(p.c + (int64_t)m * p.N + nout) = pk;
    } else {
      *reinterpret_cast<f32x4_t*>(slab + ((int64_t)blockIdx.y * p.M + m) * p.N + nout) = acc[t];
    }
  }
}

// out[m][n] = sum over splits (ascending) of slab[split][m][n], rounded once
template <typename T>
__global__ void wq_reduce_kernel(const float* __restrict__ slab, uint16_t* __restrict__ out,
                                 int64_t mn, int splits) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= mn) return;
  f32x4_t s = *reinterpret_cast<const f32x4_t*>(slab + i);
  for (int k = 1; k < splits; ++k) s += *reinterpret_cast<const f32x4_t*>(slab + (int64_t)k * mn + i);
  uint2 pk;
  pk.x = T::pack2(s[0], s[1]);
  pk.y = T::pack2(s[2], s[3]);
  *reinterpret_cast<uint2*>(out + i) = pk;
}

struct WqStreamPlan { int mt, splits, k_per_wg; };
static WqStreamPlan wq_stream_plan(int M, int N, int K) {
  WqStreamPlan pl;
  pl.mt = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
  const int base = (N / 64) * ((M + 16 * pl.mt - 1) / (16 * pl.mt));
  const int k_units = K / WS_K;
  int splits = std::max(1, std::min(512 / std::max(base, 1), 16));
  splits = std::min(splits, std::max(1, k_units / 2));
  pl.k_per_wg = ((k_units + splits - 1) / splits) * WS_K;
  pl.splits = (K + pl.k_per_wg - 1) / pl.k_per_wg;
  return pl;
}
// the streaming kernel's domain: 4 bits, whole stages and tiles, groups that are whole k-steps.  Act-order in
// the exllama form is inside it: gptq_shuffle has sorted the weight rows by group, so the groups are contiguous
// again and only the activation columns have to be gathered -- once per call, by permute_cols_kernel, instead
// of per weight tile (the reference's Marlin path does the same, gptq_marlin.cu:345-394)
static bool wq_stream_ok(const WqParams& p) {
  return p.bits == 4 && p.g_idx == nullptr && p.K % WS_K == 0 && p.N % 64 == 0 &&
         p.group_size % 32 == 0 && p.group_size > 0;
}

// out[m, k] = a[m, perm[k]]: one thread per 8 consecutive output elements (16-byte store)
__global__ void permute_cols_kernel(const uint16_t* __restrict__ a, const int* __restrict__ perm,
                                    uint16_t* __restrict__ out, int M, int K) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_row = K / 8;
  if (idx >= (int64_t)M * per_row) return;
  const int m = idx / per_row, k0 = (idx % per_row) * 8;
  const uint16_t* row = a + (int64_t)m * K;
  uint16_t e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) e[j] = row[perm[k0 + j]];
  st16(out + (int64_t)m * K + k0, make_uint4(e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16),
                                             e[4] | ((uint32_t)e[5] << 16), e[6] | ((uint32_t)e[7] << 16)));
}

template <typename T, int FMT>
static int launch_wq_stream(const WqParams& p_in, float* slab, int64_t slab_bytes, hipStream_t s) {
  WqParams p = p_in;
  const WqStreamPlan pl = wq_stream_plan(p.M, p.N, p.K);
  // scratch = [gathered activations (act-order only)] [split-K slabs]
  const int64_t a_bytes = p.perm ? (((int64_t)p.M * p.K * 2 + 255) & ~(int64_t)255) : 0;
  const int64_t need = a_bytes + (pl.splits > 1 ? (int64_t)pl.splits * p.M * p.N * 4 : 0);
  if (need > 0 && (slab == nullptr || slab_bytes < need)) return -2;
  if (p.perm) {
    uint16_t* a_perm = reinterpret_cast<uint16_t*>(slab);
    const int64_t cells = (int64_t)p.M * (p.K / 8);
    hipLaunchKernelGGL(permute_cols_kernel, dim3((unsigned)cdiv64(cells, 256)), dim3(256), 0, s, p.a, p.perm, a_perm,
                       p.M, p.K);
    p.a = a_perm;
    p.perm = nullptr;
    slab = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(slab) + a_bytes);
  }
  dim3 grid(p.N / 64, pl.splits, (p.M + 16 * pl.mt - 1) / (16 * pl.mt));
  if (pl.mt == 1) hipLaunchKernelGGL((wq_stream_kernel<T, FMT, 1>), grid, dim3(256), 0, s, p, slab, pl.k_per_wg);
  else if (pl.mt == 2) hipLaunchKernelGGL((wq_stream_kernel<T, FMT, 2>), grid, dim3(256), 0, s, p, slab, pl.k_per_wg);
  else hipLaunchKernelGGL((wq_stream_kernel<T, FMT, 4>), grid, dim3(256), 0, s, p, slab, pl.k_per_wg);
  if (pl.splits > 1) {
    const int64_t mn = (int64_t)p.M * p.N;
    hipLaunchKernelGGL((wq_reduce_kernel<T>), dim3((unsigned)cdiv64(mn / 4, 256)), dim3(256), 0, s,
                       slab, p.c, mn, pl.splits);
  }
  return 0;
}

// [K, N] dense reconstruction (awq_dequantize; also a debugging aid for the other formats)
template <typename T, int FMT>
__global__ void wq_dequant_kernel(const WqParams p, uint16_t* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)p.K * p.N) return;
  const int k = idx / p.N, n = idx % p.N;
  out[idx] = T::from_float(wq_weight<T, FMT>(p, k, n));
}

// gptq_shuffle with an act-order permutation: rows are re-ordered to perm order, packing unchanged.
// (The reference additionally interleaves nibbles for its own dequant trick, q_gemm.cu:1543-1822;
// that order is private to its kernel pair and not observable through the op contract.)
__global__ void gptq_permute_rows_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                         const int* __restrict__ perm, int K, int N, int bits) {
  const int pack = 32 / bits;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)(K / pack) * N) return;
  const int row = idx / N, n = idx % N;
  const uint32_t mask = (1u << bits) - 1;
  uint32_t res = 0;
  for (int e = 0; e < pack; ++e) {
    const int ks = perm[row * pack + e];
    res |= ((src[(int64_t)(ks / pack) * N + n] >> (bits * (ks % pack))) & mask) << (bits * e);
  }
  dst[idx] = res;
}

// 3-bit: a thread owns one column of one 32-row group of the OUTPUT (three words) and gathers its 32 codes
__global__ void gptq_permute_rows_3bit_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                              const int* __restrict__ perm, int K, int N) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)(K / 32) * N) return;
  const int grp = idx / N, n = idx % N;
  uint64_t acc = 0;   // bit stream under construction
  int filled = 0, w = 0;
  for (int e = 0; e < 32; ++e) {
    const int ks = perm[grp * 32 + e];
    const int bit0 = (ks & 31) * 3, sw = bit0 >> 5, sh = bit0 & 31;
    const uint32_t* col = src + (int64_t)((ks >> 5) * 3) * N + n;
    const uint64_t lo = col[(int64_t)sw * N];
    const uint64_t hi = (sh > 29) ? col[(int64_t)(sw + 1) * N] : 0;
    acc |= (uint64_t)((uint32_t)(((hi << 32) | lo) >> sh) & 7u) << filled;
    filled += 3;
    if (filled >= 32) {
      dst[(int64_t)(grp * 3 + w) * N + n] = (uint32_t)acc;
      acc >>= 32;
      filled -= 32;
      ++w;
    }
  }
}

// plan of the generic kernel: ~512 workgroups, at least 4 k-steps per split
struct WqGenericPlan { int mt, splits, k_per_wg; };
static WqGenericPlan wq_generic_plan(int M, int N, int K) {
  WqGenericPlan pl;
  pl.mt = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
  const int base = ((N + 63) / 64) * ((M + 16 * pl.mt - 1) / (16 * pl.mt));
  const int k_units = K / 32;
  int splits = std::max(1, std::min(512 / std::max(base, 1), 16));
  splits = std::min(splits, std::max(1, k_units / 4));
  pl.k_per_wg = ((k_units + splits - 1) / splits) * 32;
  pl.splits = (K + pl.k_per_wg - 1) / pl.k_per_wg;
  return pl;
}
static int64_t wq_generic_scratch_bytes(int M, int N, int K) {
  const WqGenericPlan pl = wq_generic_plan(M, N, K);
  const int64_t a_bytes = ((int64_t)M * K * 2 + 255) & ~(int64_t)255;   // gathered activations (act-order)
  return a_bytes + (pl.splits > 1 ? (int64_t)pl.splits * M * N * 4 : 0);
}

// scratch = [gathered activations (act-order only)] [split-K slabs]; without enough scratch for the slabs the
// launch runs unsplit, without room for the gathered activations it fails (-2)
template <typename T, int FMT>
static int launch_wq(const WqParams& p_in, void* scratch, int64_t scratch_bytes, hipStream_t s) {
  WqParams p = p_in;
  WqGenericPlan pl = wq_generic_plan(p.M, p.N, p.K);
  uint8_t* cur = reinterpret_cast<uint8_t*>(scratch);
  int64_t left = scratch ? scratch_bytes : 0;
  if (p.perm) {
    const int64_t a_bytes = ((int64_t)p.M * p.K * 2 + 255) & ~(int64_t)255;
    if (left < a_bytes) return -2;
    uint16_t* a_perm = reinterpret_cast<uint16_t*>(cur);
    const int64_t cells = (int64_t)p.M * (p.K / 8);
    hipLaunchKernelGGL(permute_cols_kernel, dim3((unsigned)cdiv64(cells, 256)), dim3(256), 0, s, p.a, p.perm, a_perm,
                       p.M, p.K);
    p.a = a_perm;
    p.perm = nullptr;
    cur += a_bytes;
    left -= a_bytes;
  }
  const int64_t mn = (int64_t)p.M * p.N;
  if (pl.splits > 1 && (left < (int64_t)pl.splits * mn * 4 || mn % 4 != 0)) {
    pl.splits = 1;
    pl.k_per_wg = p.K;
  }
  float* slab = pl.splits > 1 ? reinterpret_cast<float*>(cur) : nullptr;
  const int nblk = (p.N + 63) / 64;
  dim3 grid(nblk, pl.splits, (p.M + 16 * pl.mt - 1) / (16 * pl.mt));
  if (pl.mt == 1) hipLaunchKernelGGL((wq_gemm_kernel<T, FMT, 1>), grid, dim3(256), 0, s, p, slab, pl.k_per_wg);
  else if (pl.mt == 2) hipLaunchKernelGGL((wq_gemm_kernel<T, FMT, 2>), grid, dim3(256), 0, s, p, slab, pl.k_per_wg);
  else hipLaunchKernelGGL((wq_gemm_kernel<T, FMT, 4>), grid, dim3(256), 0, s, p, slab, pl.k_per_wg);
  if (pl.splits > 1)
    hipLaunchKernelGGL((wq_reduce_kernel<T>), dim3((unsigned)cdiv64(mn / 4, 256)), dim3(256), 0, s, slab, p.c, mn,
                       pl.splits);
  return 0;
}

template <typename T>
static int launch_wq_fmt(int fmt, const WqParams& p, void* scratch, int64_t scratch_bytes, hipStream_t s) {
  switch (fmt) {
    case WQ_GPTQ: return launch_wq<T, WQ_GPTQ>(p, scratch, scratch_bytes, s);
    case WQ_AWQ: return launch_wq<T, WQ_AWQ>(p, scratch, scratch_bytes, s);
    case WQ_MARLIN: return launch_wq<T, WQ_MARLIN>(p, scratch, scratch_bytes, s);
    case WQ_MARLIN_FP8: return launch_wq<T, WQ_MARLIN_FP8>(p, scratch, scratch_bytes, s);
  }
  return -1;
}

// used by w4a16_gemm.hip for the Marlin variants its tuned kernel does not cover
int64_t wq_marlin_fallback_scratch_bytes(int size_m, int size_n, int size_k) {
  return wq_generic_scratch_bytes(size_m, size_n, size_k);
}
int wq_marlin_fallback(void* c, const void* a, const int32_t* b_q_weight, const void* b_scales,
                       const int32_t* g_idx, const int32_t* perm, int num_bits, int size_m,
                       int size_n, int size_k, int num_groups, int is_fp8, nmv_dtype_t dtype,
                       void* scratch, int64_t scratch_bytes, hipStream_t stream) {
  WqParams p{(const uint16_t*)a, (const uint32_t*)b_q_weight, nullptr, (const uint16_t*)b_scales,
             g_idx, perm, (uint16_t*)c, size_m, size_n, size_k, num_bits,
             num_groups > 1 ? size_k / num_groups : 0, num_groups};
  const int fmt = is_fp8 ? WQ_MARLIN_FP8 : WQ_MARLIN;
  return dtype == NMV_F16 ? launch_wq_fmt<F16>(fmt, p, scratch, scratch_bytes, stream)
                          : launch_wq_fmt<BF16>(fmt, p, scratch, scratch_bytes, stream);
}

}  // namespace nmv

using namespace nmv;

extern "C" int64_t nmv_wq_gemm_scratch_bytes(int size_m, int size_n, int size_k) {
  if (size_m <= 0 || size_n <= 0 || size_k <= 0) return 0;
  // the larger of what the two kernels want (the caller does not say which format / act-order it will pass):
  // the gathered activations of an act-order call + the split-K slabs
  int64_t need = wq_generic_scratch_bytes(size_m, size_n, size_k);
  if (size_k % WS_K == 0 && size_n % 64 == 0) {
    const WqStreamPlan pl = wq_stream_plan(size_m, size_n, size_k);
    const int64_t a_bytes = ((int64_t)size_m * size_k * 2 + 255) & ~(int64_t)255;
    need = std::max(need, a_bytes + (pl.splits > 1 ? (int64_t)pl.splits * size_m * size_n * 4 : 0));
  }
  return need;
}

extern "C" int nmv_gptq_gemm(void* c, const void* a, const int32_t* b_q_weight,
                             const int32_t* b_gptq_qzeros, const void* b_gptq_scales,
                             const int32_t* b_g_idx, int use_exllama, int bit, int size_m,
                             int size_n, int size_k, int num_groups, nmv_dtype_t dtype,
                             void* scratch, int64_t scratch_bytes, void* stream) {
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16, "gptq_gemm: unsupported dtype %d", (int)dtype);
  NMV_CHECK(bit == 2 || bit == 3 || bit == 4 || bit == 8, "gptq_gemm: %d-bit weights are not supported", bit);
  NMV_CHECK(size_k % 32 == 0 && num_groups >= 1 && size_k % num_groups == 0, "gptq_gemm: bad K / groups");
  NMV_CHECK(size_n % (bit == 3 ? 32 : 32 / bit) == 0, "gptq_gemm: N must be a multiple of the pack factor");
  if (size_m == 0) return NMV_OK;
  // exllama: weights were row-permuted by gptq_shuffle, b_g_idx is that permutation (gathers A);
  // otherwise b_g_idx is the per-row group index of the unshuffled weights (q_gemm.cu:1823-1846)
  WqParams p{(const uint16_t*)a, (const uint32_t*)b_q_weight, (const uint32_t*)b_gptq_qzeros,
             (const uint16_t*)b_gptq_scales, use_exllama ? nullptr : b_g_idx,
             use_exllama ? b_g_idx : nullptr, (uint16_t*)c, size_m, size_n, size_k, bit,
             size_k / num_groups, num_groups};
  int rc;
  if (wq_stream_ok(p)) {
    rc = dtype == NMV_F16
             ? launch_wq_stream<F16, WQ_GPTQ>(p, (float*)scratch, scratch_bytes, (hipStream_t)stream)
             : launch_wq_stream<BF16, WQ_GPTQ>(p, (float*)scratch, scratch_bytes, (hipStream_t)stream);
    NMV_CHECK(rc != -2, "gptq_gemm: scratch too small (see nmv_wq_gemm_scratch_bytes)");
  } else {
    rc = dtype == NMV_F16 ? launch_wq_fmt<F16>(WQ_GPTQ, p, scratch, scratch_bytes, (hipStream_t)stream)
                          : launch_wq_fmt<BF16>(WQ_GPTQ, p, scratch, scratch_bytes, (hipStream_t)stream);
    NMV_CHECK(rc != -2, "gptq_gemm: scratch too small (see nmv_wq_gemm_scratch_bytes)");
  }
  NMV_CHECK(rc == 0, "gptq_gemm: launch failed");
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_gptq_shuffle(int32_t* q_weight, const int32_t* q_perm, int32_t* tmp, int size_k,
                                int size_n, int bit, void* stream) {
  NMV_CHECK(bit == 2 || bit == 3 || bit == 4 || bit == 8, "gptq_shuffle: %d-bit weights are not supported", bit);
  if (q_perm == nullptr) return NMV_OK;  // nothing observable to do without act-order
  NMV_CHECK(tmp != nullptr, "gptq_shuffle: scratch required with a permutation");
  NMV_CHECK(bit != 3 || size_k % 32 == 0, "gptq_shuffle: 3-bit rows come in groups of 32");
  const int64_t total = bit == 3 ? (int64_t)(size_k / 32) * 3 * size_n : (int64_t)(size_k / (32 / bit)) * size_n;
  if (total == 0) return NMV_OK;
  hipStream_t s = (hipStream_t)stream;
  if (bit == 3)
    hipLaunchKernelGGL(gptq_permute_rows_3bit_kernel, dim3((unsigned)cdiv64((int64_t)(size_k / 32) * size_n, 256)),
                       dim3(256), 0, s, (const uint32_t*)q_weight, (uint32_t*)tmp, q_perm, size_k, size_n);
  else
    hipLaunchKernelGGL(gptq_permute_rows_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s,
                       (const uint32_t*)q_weight, (uint32_t*)tmp, q_perm, size_k, size_n, bit);
  hipError_t e = hipMemcpyAsync(q_weight, tmp, total * 4, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) { set_error("gptq_shuffle: copy failed: %s", hipGetErrorString(e)); return NMV_ERR_HIP; }
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_awq_gemm(void* c, const void* a, const int32_t* qweight, const void* scales,
                            const int32_t* qzeros, int size_m, int size_n, int size_k,
                            int num_groups, nmv_dtype_t dtype, void* scratch,
                            int64_t scratch_bytes, void* stream) {
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16, "awq_gemm: unsupported dtype %d", (int)dtype);
  NMV_CHECK(size_n % 8 == 0 && size_k % 32 == 0 && num_groups >= 1 && size_k % num_groups == 0,
            "awq_gemm: bad shape (N %% 8, K %% 32, K %% groups)");
  if (size_m == 0) return NMV_OK;
  WqParams p{(const uint16_t*)a, (const uint32_t*)qweight, (const uint32_t*)qzeros,
             (const uint16_t*)scales, nullptr, nullptr, (uint16_t*)c, size_m, size_n, size_k, 4,
             size_k / num_groups, num_groups};
  int rc;
  if (wq_stream_ok(p)) {
    rc = dtype == NMV_F16
             ? launch_wq_stream<F16, WQ_AWQ>(p, (float*)scratch, scratch_bytes, (hipStream_t)stream)
             : launch_wq_stream<BF16, WQ_AWQ>(p, (float*)scratch, scratch_bytes, (hipStream_t)stream);
    NMV_CHECK(rc != -2, "awq_gemm: scratch too small (see nmv_wq_gemm_scratch_bytes)");
  } else {
    rc = dtype == NMV_F16 ? launch_wq_fmt<F16>(WQ_AWQ, p, scratch, scratch_bytes, (hipStream_t)stream)
                          : launch_wq_fmt<BF16>(WQ_AWQ, p, scratch, scratch_bytes, (hipStream_t)stream);
  }
  NMV_CHECK(rc == 0, "awq_gemm: launch failed");
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_awq_dequantize(void* out, const int32_t* qweight, const void* scales,
                                  const int32_t* qzeros, int size_n, int size_k, int num_groups,
                                  nmv_dtype_t dtype, void* stream) {
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16, "awq_dequantize: unsupported dtype %d", (int)dtype);
  NMV_CHECK(size_n % 8 == 0 && num_groups >= 1 && size_k % num_groups == 0, "awq_dequantize: bad shape");
  const int64_t total = (int64_t)size_k * size_n;
  if (total == 0) return NMV_OK;
  WqParams p{nullptr, (const uint32_t*)qweight, (const uint32_t*)qzeros, (const uint16_t*)scales,
             nullptr, nullptr, nullptr, 0, size_n, size_k, 4, size_k / num_groups, num_groups};
  dim3 grid((unsigned)cdiv64(total, 256)), block(256);
  if (dtype == NMV_F16)
    hipLaunchKernelGGL((wq_dequant_kernel<F16, WQ_AWQ>), grid, block, 0, (hipStream_t)stream, p, (uint16_t*)out);
  else
    hipLaunchKernelGGL((wq_dequant_kernel<BF16, WQ_AWQ>), grid, block, 0, (hipStream_t)stream, p, (uint16_t*)out);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

int marlin_tall_fp8(void* c, const void* a, const int32_t* b_q_weight, const void* b_scales, int32_t* workspace,
                    int64_t workspace_len, void* scratch, int64_t scratch_bytes, int size_m, int size_n, int size_k,
                    int num_groups, nmv_dtype_t dtype, void* stream);   // w4a16_gemm.hip

extern "C" int64_t nmv_fp8_marlin_gemm_scratch_bytes(int size_m, int size_n, int size_k) {
  if (size_m <= 0 || size_n <= 0 || size_k <= 0) return 0;
  // the generic kernel's gathered activations + slabs, or the tall 8-bit kernel's split-K slabs
  return std::max(wq_generic_scratch_bytes(size_m, size_n, size_k), nmv_gptq_marlin_gemm_scratch_bytes(size_m, size_n, size_k, 0));
}

extern "C" int nmv_fp8_marlin_gemm(void* c, const void* a, const int32_t* b_q_weight,
                                   const void* b_scales, int32_t* workspace, int64_t workspace_len,
                                   void* scratch, int64_t scratch_bytes, int num_bits, int size_m, int size_n,
                                   int size_k, int num_groups, nmv_dtype_t dtype, void* stream) {
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16, "fp8_marlin_gemm only supports bfloat16 and float16");
  NMV_CHECK(num_bits == 8, "num_bits must be 8 for fp8. Got = %d", num_bits);
  NMV_CHECK(size_n % 64 == 0 && size_k % 32 == 0, "fp8_marlin_gemm: N %% 64 and K %% 32 required");
  if (size_m == 0) return NMV_OK;
  // channelwise scales and K in whole 256-k rings: the tall 8-bit Marlin kernel with the fp8 byte conversion
  // (4-5 x the generic kernel); `workspace` is the reference's zeroed lock array = its split-K tickets
  if (num_groups == 1 && workspace != nullptr) {
    const int rc = marlin_tall_fp8(c, a, b_q_weight, b_scales, workspace, workspace_len, scratch, scratch_bytes, size_m,
                                   size_n, size_k, num_groups, dtype, stream);
    if (rc != -1000) return rc;
  }
  const int rc = wq_marlin_fallback(c, a, b_q_weight, b_scales, nullptr, nullptr, 8, size_m, size_n,
                                    size_k, num_groups, 1, dtype, scratch, scratch_bytes, (hipStream_t)stream);
  NMV_CHECK(rc == 0, "fp8_marlin_gemm: launch failed");
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}
