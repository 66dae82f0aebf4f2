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

template <typename T, int FMT, int MT>
__global__ __launch_bounds__(256) void wq_gemm_kernel(const WqParams p) {
  // [2 buffers][64 columns][32 k] in the model dtype; row = one column's 32 k values (64 B)
  __shared__ __attribute__((aligned(16))) uint16_t w_s[2][64][32 + 8];  // +8: spread rows over banks
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 64;
  const int m0 = blockIdx.y * (16 * MT);
  // dequantisation cell of this thread: column dn, k group dk (8 consecutive k)
  const int dn = threadIdx.x & 63, dk = threadIdx.x >> 6;
  const int n_d = min(n0 + dn, p.N - 1);

  f32x4_t acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto dequant = [&](int buf, int k0) {
    uint32_t pk[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = wq_weight<T, FMT>(p, k0 + dk * 8 + 2 * e, n_d);
      const float hi = wq_weight<T, FMT>(p, k0 + dk * 8 + 2 * e + 1, n_d);
      pk[e] = T::pack2(lo, hi);
    }
    *reinterpret_cast<uint4*>(&w_s[buf][dn][dk * 8]) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
  };

  const int ksteps = p.K / 32;
  dequant(0, 0);
  __syncthreads();
  for (int ks = 0; ks < ksteps; ++ks) {
    const int buf = ks & 1;
    const int k0 = ks * 32;
    if (ks + 1 < ksteps) dequant(buf ^ 1, k0 + 32);
    // activation fragments: rows m0 + 16 t + r, k = k0 + 8 g .. + 8 (gathered through perm if given)
    uint4 af[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int m = m0 + t * 16 + r;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m < p.M) {
        const uint16_t* row = p.a + (int64_t)m * p.K;
        if (p.perm == nullptr) {
          v = ld16(row + k0 + g * 8);
        } else {
          uint16_t e[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) e[j] = row[p.perm[k0 + g * 8 + j]];
          v = make_uint4(e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16),
                         e[4] | ((uint32_t)e[5] << 16), e[6] | ((uint32_t)e[7] << 16));
        }
      }
      af[t] = v;
    }
    const uint4 wf = *reinterpret_cast<const uint4*>(&w_s[buf][wave * 16 + r][g * 8]);
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = Mfma16<T>::run(wf, af[t], acc[t]);
    __syncthreads();
  }
  // D[row = column index][col = m]: lane (m = r, g) holds columns n0 + 16 wave + 4 g + i
  const int nb = n0 + wave * 16 + 4 * g;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = m0 + t * 16 + r;
    if (m >= p.M) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (nb + i < p.N) p.c[(int64_t)m * p.N + nb + i] = T::from_float(acc[t][i]);
  }
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

template <typename T, int FMT>
static void launch_wq(const WqParams& p, hipStream_t s) {
  const int nblk = (p.N + 63) / 64;
  if (p.M <= 16) {
    hipLaunchKernelGGL((wq_gemm_kernel<T, FMT, 1>), dim3(nblk, (p.M + 15) / 16), dim3(256), 0, s, p);
  } else if (p.M <= 32) {
    hipLaunchKernelGGL((wq_gemm_kernel<T, FMT, 2>), dim3(nblk, (p.M + 31) / 32), dim3(256), 0, s, p);
  } else {
    hipLaunchKernelGGL((wq_gemm_kernel<T, FMT, 4>), dim3(nblk, (p.M + 63) / 64), dim3(256), 0, s, p);
  }
}

template <typename T>
static int launch_wq_fmt(int fmt, const WqParams& p, hipStream_t s) {
  switch (fmt) {
    case WQ_GPTQ: launch_wq<T, WQ_GPTQ>(p, s); return 0;
    case WQ_AWQ: launch_wq<T, WQ_AWQ>(p, s); return 0;
    case WQ_MARLIN: launch_wq<T, WQ_MARLIN>(p, s); return 0;
    case WQ_MARLIN_FP8: launch_wq<T, WQ_MARLIN_FP8>(p, s); return 0;
  }
  return -1;
}

// used by w4a16_gemm.hip for the Marlin variants its tuned kernel does not cover
int wq_marlin_fallback(void* c, const void* a, const int32_t* b_q_weight, const void* b_scales,
                       const int32_t* g_idx, const int32_t* perm, int num_bits, int size_m,
                       int size_n, int size_k, int num_groups, int is_fp8, nmv_dtype_t dtype,
                       hipStream_t stream) {
  WqParams p{(const uint16_t*)a, (const uint32_t*)b_q_weight, nullptr, (const uint16_t*)b_scales,
             g_idx, perm, (uint16_t*)c, size_m, size_n, size_k, num_bits,
             num_groups > 1 ? size_k / num_groups : 0, num_groups};
  const int fmt = is_fp8 ? WQ_MARLIN_FP8 : WQ_MARLIN;
  return dtype == NMV_F16 ? launch_wq_fmt<F16>(fmt, p, stream) : launch_wq_fmt<BF16>(fmt, p, stream);
}

}  // namespace nmv

using namespace nmv;

extern "C" int nmv_gptq_gemm(void* c, const void* a, const int32_t* b_q_weight,
                             const int32_t* b_gptq_qzeros, const void* b_gptq_scales,
                             const int32_t* b_g_idx, int use_exllama, int bit, int size_m,
                             int size_n, int size_k, int num_groups, nmv_dtype_t dtype,
                             void* stream) {
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16, "gptq_gemm: unsupported dtype %d", (int)dtype);
  NMV_CHECK(bit == 2 || bit == 4 || bit == 8, "gptq_gemm: %d-bit weights are not supported on gfx950 (2, 4, 8 are)", bit);
  NMV_CHECK(size_k % 32 == 0 && num_groups >= 1 && size_k % num_groups == 0, "gptq_gemm: bad K / groups");
  NMV_CHECK(size_n % (32 / bit) == 0, "gptq_gemm: N must be a multiple of the pack factor");
  if (size_m == 0) return NMV_OK;
  // exllama: weights were row-permuted by gptq_shuffle, b_g_idx is that permutation (gathers A);
  // otherwise b_g_idx is the per-row group index of the unshuffled weights (q_gemm.cu:1823-1846)
  WqParams p{(const uint16_t*)a, (const uint32_t*)b_q_weight, (const uint32_t*)b_gptq_qzeros,
             (const uint16_t*)b_gptq_scales, use_exllama ? nullptr : b_g_idx,
             use_exllama ? b_g_idx : nullptr, (uint16_t*)c, size_m, size_n, size_k, bit,
             size_k / num_groups, num_groups};
  const int rc = dtype == NMV_F16 ? launch_wq_fmt<F16>(WQ_GPTQ, p, (hipStream_t)stream)
                                  : launch_wq_fmt<BF16>(WQ_GPTQ, p, (hipStream_t)stream);
  NMV_CHECK(rc == 0, "gptq_gemm: launch failed");
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_gptq_shuffle(int32_t* q_weight, const int32_t* q_perm, int32_t* tmp, int size_k,
                                int size_n, int bit, void* stream) {
  NMV_CHECK(bit == 2 || bit == 4 || bit == 8, "gptq_shuffle: %d-bit weights are not supported on gfx950", bit);
  if (q_perm == nullptr) return NMV_OK;  // nothing observable to do without act-order
  NMV_CHECK(tmp != nullptr, "gptq_shuffle: scratch required with a permutation");
  const int pack = 32 / bit;
  const int64_t total = (int64_t)(size_k / pack) * size_n;
  if (total == 0) return NMV_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gptq_permute_rows_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s,
                     (const uint32_t*)q_weight, (uint32_t*)tmp, q_perm, size_k, size_n, bit);
  hipError_t e = hipMemcpyAsync(q_weight, tmp, total * 4, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) { set_error("gptq_shuffle: copy failed: %s", hipGetErrorString(e)); return NMV_ERR_HIP; }
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_awq_gemm(void* c, const void* a, const int32_t* qweight, const void* scales,
                            const int32_t* qzeros, int size_m, int size_n, int size_k,
                            int num_groups, nmv_dtype_t dtype, void* stream) {
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16, "awq_gemm: unsupported dtype %d", (int)dtype);
  NMV_CHECK(size_n % 8 == 0 && size_k % 32 == 0 && num_groups >= 1 && size_k % num_groups == 0,
            "awq_gemm: bad shape (N %% 8, K %% 32, K %% groups)");
  if (size_m == 0) return NMV_OK;
  WqParams p{(const uint16_t*)a, (const uint32_t*)qweight, (const uint32_t*)qzeros,
             (const uint16_t*)scales, nullptr, nullptr, (uint16_t*)c, size_m, size_n, size_k, 4,
             size_k / num_groups, num_groups};
  const int rc = dtype == NMV_F16 ? launch_wq_fmt<F16>(WQ_AWQ, p, (hipStream_t)stream)
                                  : launch_wq_fmt<BF16>(WQ_AWQ, p, (hipStream_t)stream);
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

extern "C" int nmv_fp8_marlin_gemm(void* c, const void* a, const int32_t* b_q_weight,
                                   const void* b_scales, int32_t* workspace, int64_t workspace_len,
                                   int num_bits, int size_m, int size_n, int size_k, int num_groups,
                                   nmv_dtype_t dtype, void* stream) {
  (void)workspace; (void)workspace_len;
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16, "fp8_marlin_gemm only supports bfloat16 and float16");
  NMV_CHECK(num_bits == 8, "num_bits must be 8 for fp8. Got = %d", num_bits);
  NMV_CHECK(size_n % 64 == 0 && size_k % 32 == 0, "fp8_marlin_gemm: N %% 64 and K %% 32 required");
  if (size_m == 0) return NMV_OK;
  const int rc = wq_marlin_fallback(c, a, b_q_weight, b_scales, nullptr, nullptr, 8, size_m, size_n,
                                    size_k, num_groups, 1, dtype, (hipStream_t)stream);
  NMV_CHECK(rc == 0, "fp8_marlin_gemm: launch failed");
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}
