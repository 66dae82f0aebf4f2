// Definitions shared by the W4A16 kernels (w4a16_gemm.hip: the 16-row and "tall" kernels and the host entry
// points; w4a16_stream.hip: the resident / streamed-activation kernel): nibble -> model-dtype expansion,
// the launch parameter block, and the split-K last-arriver reduction.
#pragma once
#include <type_traits>

#include "common.h"

namespace nmv {

constexpr int GT = 256;        // threads per workgroup
constexpr int STAGE_K = 128;   // k per pipeline stage (= 4 MFMA k-steps of 32)
constexpr int KSTEPS = STAGE_K / 32;

// (x & mask) | magic in ONE VALU op (v_and_or_b32).  gfx9 VOP3 encodings take no literals, so
// hipcc splits the expression into v_and_b32 + v_or_b32 when mask and magic are constants; callers
// therefore pin the mask in an SGPR (readfirstlane) and the magic in a VGPR (an empty asm), and the
// compiler then selects the three-operand form itself.  This must stay a compiler-visible
// expression: an inline-asm v_and_or_b32 whose result fed the next v_mfma directly produced
// garbage accumulator tiles (the hazard recogniser does not look inside asm statements).
__device__ __forceinline__ uint32_t and_or(uint32_t x, uint32_t mask_sgpr, uint32_t magic_vgpr) {
  return (x & mask_sgpr) | magic_vgpr;
}

// Weight vectors that a launch reads once (a single block of rows: M <= 16 mt) are streamed with the
// non-temporal policy so that they do not displace the activation tile and the scales in the XCD's L2
// (measured: -2..4 % at M <= 16); with two row blocks the second one finds them in L2 and nt costs 10 %.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// NT is a template parameter of the kernels: a run-time choice is a uniform branch around the load, which
// makes hipcc fall back to `vmcnt(0)` waits and costs the whole software pipeline (measured: +10 %).
template <bool NT>
__device__ __forceinline__ uint4 ld_stream(const uint4* ptr) {
  if constexpr (NT) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(ptr));
    return make_uint4(v.x, v.y, v.z, v.w);
  } else {
    return *ptr;
  }
}

template <typename T> struct W4;
template <> struct W4<BF16> {
  static constexpr uint32_t MASK = 0x00780078u, MAGIC = 0x41804180u, ONES = 0x3F803F80u;
  // the same four shifts as rotate-right amounts (the wrapped bits fall outside MASK)
  static constexpr uint32_t ROT_LO0 = 29, ROT_HI0 = 1, ROT_LO1 = 5, ROT_HI1 = 9;
  // nibble -> mantissa bits [6:3]
  static __device__ __forceinline__ uint32_t lo0(uint32_t x, uint32_t m, uint32_t g) { return and_or(x << 3, m, g); }
  static __device__ __forceinline__ uint32_t hi0(uint32_t x, uint32_t m, uint32_t g) { return and_or(x >> 1, m, g); }
  static __device__ __forceinline__ uint32_t lo1(uint32_t x, uint32_t m, uint32_t g) { return and_or(x >> 5, m, g); }
  static __device__ __forceinline__ uint32_t hi1(uint32_t x, uint32_t m, uint32_t g) { return and_or(x >> 9, m, g); }
  static __device__ __forceinline__ f32x4_t mfma(uint4 w, uint4 a, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w),
                                                   __builtin_bit_cast(bf16x8_t, a), c, 0, 0, 0);
  }
};
template <> struct W4<F16> {
  static constexpr uint32_t MASK = 0x03C003C0u, MAGIC = 0x4C004C00u, ONES = 0x3C003C00u;
  static constexpr uint32_t ROT_LO0 = 26, ROT_HI0 = 30, ROT_LO1 = 2, ROT_HI1 = 6;
  // nibble -> mantissa bits [9:6]
  static __device__ __forceinline__ uint32_t lo0(uint32_t x, uint32_t m, uint32_t g) { return and_or(x << 6, m, g); }
  static __device__ __forceinline__ uint32_t hi0(uint32_t x, uint32_t m, uint32_t g) { return and_or(x << 2, m, g); }
  static __device__ __forceinline__ uint32_t lo1(uint32_t x, uint32_t m, uint32_t g) { return and_or(x >> 2, m, g); }
  static __device__ __forceinline__ uint32_t hi1(uint32_t x, uint32_t m, uint32_t g) { return and_or(x >> 6, m, g); }
  static __device__ __forceinline__ f32x4_t mfma(uint4 w, uint4 a, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w),
                                                  __builtin_bit_cast(f16x8_t, a), c, 0, 0, 0);
  }
};
constexpr float W4_ZP = 24.0f;  // (16 + q) - 24 = q - 8

// The MFMA-native tensor (w4a16_gemm.hip "The MFMA-native weight layout"): pair p of a dword = nibbles p and p + 4
template <typename T> struct W4N;
template <> struct W4N<BF16> {   // nibble stays in mantissa bits [3:0]: 128 + q
  static constexpr uint32_t MASK = 0x000F000Fu, MAGIC = 0x43004300u, ONES = 0x3F803F80u;
  static constexpr int POS = 0;
  static constexpr float ZPC = 136.0f;
};
template <> struct W4N<F16> {    // nibble in mantissa bits [9:6]: 16 + q
  static constexpr uint32_t MASK = 0x03C003C0u, MAGIC = 0x4C004C00u, ONES = 0x3C003C00u;
  static constexpr int POS = 6;
  static constexpr float ZPC = 24.0f;
};


struct GemmParams {
  const uint16_t* a;      // [M, K]
  const uint4* b;         // Marlin int32 [K/16, N*2] viewed as uint4 [K/16, N/2]
  const uint16_t* s;      // [num_groups, N] (marlin_permute_scales layout)
  const uint16_t* zp;     // zero points z in the model dtype, layout of s; null = symmetric (8)
  const int* perm;        // [K] or null: A columns are gathered through it (act-order)
  uint16_t* c;            // [M, N] (used when splits == 1)
  float* slab;            // [splits, M, N] fp32 (used when splits > 1)
  int* tickets;           // [n_blocks * m_blocks] zero on entry / exit (the Marlin `workspace`)
  int M, N, K;
  int bits;               // 4, or 8 (tall kernel only)
  int group_size;         // 32/64/128, or 0 = channelwise (one scale row)
  int k_per_wg;           // k range of one workgroup (multiple of WK*STAGE_K)
  int splits;
  int fp8;                // bits == 8: the bytes are fp8-e4m3 numbers (fp8_marlin_gemm), not unsigned codes with zero point 128
  int native;             // b is the MFMA-native tensor of nmv_w4_native_repack, s / zp are natural [groups, N]
  int epi;                // 1: silu(gate) * up epilogue on column-interleaved gate_up weights (tall
                          //    kernel, splits == 1): c is [M, N/2]
                          // 2: deferred reduction: every workgroup stores its fp32 partial tile in
                          //    slab[split] (also when splits == 1) and returns -- no ticket, no
                          //    last-arriver pass; the consumer kernel sums the slabs
  // w4a16_stream_kernel only (w4a16_stream.hip): k_per_wg = (k groups of the workgroup) * n_stages * g_stage * 128
  int g_stage;            // 128-k scale groups per activation stage and k group
  int n_stages;           // activation stages (1 = the whole k range of the workgroup stays in LDS)
  // w4a16_prefill.hip only, with epi == 2: the slabs are written in the MODEL dtype (slab[splits][M][N] of 2-byte
  // elements) -- half the bytes the consumer launch has to read back at prompt size
  int slab16;
};


// ---------------------------------------------------------------------------------------------
// Split-K: the workgroup that drew the last ticket of a tile sums the fp32 slabs.
// The sum is latency-bound (every slab read is an sc1 load that goes past the XCD's L2), so the
// only thing that matters is how many loads are in flight: a tile with few elements (decode: 1..16
// rows) is spread over the split range as well -- thread = (element, partition of the splits),
// up to 16 loads in flight each, partial sums combined through LDS -- and a tile with many
// elements keeps 16 split loads in flight per element.  The order of the additions is fixed by
// (split index, partition index) only, never by arrival: bit-reproducible.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
constexpr int RED_INFLIGHT = 16;

template <typename T, int NTHR = GT>
__device__ __forceinline__ void splitk_reduce_tile(const GemmParams& p, __amdgpu_buffer_rsrc_t rs,
                                                   int m0, int tile_rows, int n_base,
                                                   int tile_cols, f32x4_t* red /* LDS, NTHR entries */) {
  const int rows = min(tile_rows, p.M - m0);
  const int cols = min(tile_cols, p.N - n_base);
  if (rows <= 0 || cols <= 0) return;  // uniform
  const int f4_per_row = cols >> 2;
  const int n_elems = rows * f4_per_row;
  const int64_t split_stride = (int64_t)p.M * p.N * 4;  // bytes
  // uniform: either one split per partition or no partitioning, so that the association is always
  // ((0 + s0) + s1) + ... -- the order the consumers of deferred slabs (nmv_fused_add_rms_norm_partial)
  // reproduce bit for bit
  const int parts = (p.splits * n_elems <= NTHR) ? p.splits : 1;
  const int per_part = (p.splits + parts - 1) / parts;
  // exactly the loads that are needed, in batches of 16 / 8 / 4 / 2 / 1 all in flight together
  auto sum_range = [&](int off0, int s_begin, int s_end) -> f32x4_t {
    f32x4_t sum = {0.f, 0.f, 0.f, 0.f};
    int sb = s_begin;
    auto batch = [&](auto n_tag) {
      constexpr int NB = decltype(n_tag)::value;
      u32x4_t q[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u)
        q[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, off0 + (int)((sb + u) * split_stride), 0, 16);
#pragma unroll
      for (int u = 0; u < NB; ++u) sum += __builtin_bit_cast(f32x4_t, q[u]);  // fixed order
      sb += NB;
    };
    while (s_end - sb >= RED_INFLIGHT) batch(std::integral_constant<int, RED_INFLIGHT>{});
    if (s_end - sb >= 8) batch(std::integral_constant<int, 8>{});
    if (s_end - sb >= 4) batch(std::integral_constant<int, 4>{});
    if (s_end - sb >= 2) batch(std::integral_constant<int, 2>{});
    if (s_end - sb >= 1) batch(std::integral_constant<int, 1>{});
    return sum;
  };
  auto store_out = [&](int m, int n, f32x4_t sum) {
    uint2 pk;
    pk.x = T::pack2(sum[0], sum[1]);
    pk.y = T::pack2(sum[2], sum[3]);
    *reinterpret_cast<uint2*>(p.c + (int64_t)m * p.N + n) = pk;
  };
  if (parts == 1) {
    for (int e = threadIdx.x; e < n_elems; e += NTHR) {
      const int m = m0 + e / f4_per_row;
      const int n = n_base + (e % f4_per_row) * 4;
      store_out(m, n, sum_range((int)(((int64_t)m * p.N + n) * 4), 0, p.splits));
    }
    return;
  }
  const int part = threadIdx.x / n_elems;
  const int e = threadIdx.x - part * n_elems;
  const int m = m0 + e / f4_per_row;
  const int n = n_base + (e % f4_per_row) * 4;
  const bool active = part < parts;
  if (active) {
    const int s_begin = part * per_part;
    const int s_end = min(s_begin + per_part, p.splits);
    f32x4_t sum = {0.f, 0.f, 0.f, 0.f};
    if (s_begin < s_end) sum = sum_range((int)(((int64_t)m * p.N + n) * 4), s_begin, s_end);
    red[threadIdx.x] = sum;
  }
  __syncthreads();
  if (part == 0) {
    f32x4_t sum = red[e];
    for (int q = 1; q < parts; ++q) sum += red[q * n_elems + e];  // partition order
    store_out(m, n, sum);
  }
}



// ---- w4a16_stream.hip: plan and launch of the resident / streamed-activation kernel ----
struct W4StreamPlan {
  int mt, nw, cpw, d, gst;      // kernel shape: 16 mt rows, nw waves = cpw chunks x (nw / cpw) k groups, ring depth, stage
  int g_stage, n_stages;        // groups per stage and k group, stages
  int splits, k_per_wg, n_blocks, m_blocks;
  int lds_bytes;
};
// false: the shape is outside the kernel's domain (the caller takes the tall kernel)
// deferred: the launch leaves its split-K slabs to the next one (nmv_gptq_marlin_gemm_partial)
bool w4s_make_plan(int M, int N, int K, int64_t tickets_len, bool unsplit, bool deferred, W4StreamPlan* out);
int w4s_launch(const W4StreamPlan& pl, const GemmParams& p, bool f16, hipStream_t s);

// ---- w4a16_ring.hip: loader / consumer waves over an LDS-DMA ring (native tensor, 17..64 rows) ----
struct W4RingPlan {
  int mt;                       // 16 mt rows per workgroup (2, 4)
  int splits, k_per_wg, n_blocks, m_blocks;
  int lds_bytes;
};
// false: outside the kernel's domain (the caller takes the stream kernel)
bool w4r_make_plan(int M, int N, int K, int64_t tickets_len, bool unsplit, W4RingPlan* out);
int w4r_launch(const W4RingPlan& pl, const GemmParams& p, bool f16, hipStream_t s);
// ---- w4a16_prefill.hip: 256 x 128 tiles, codes expanded once per workgroup (native tensor, prompt-sized calls) ----
struct W4PrefillPlan {
  int small;                    // 0: 256 x 256 tiles, 1: 128 x 128 tiles
  int splits, k_per_wg, n_blocks, m_blocks;
  int lds_bytes;
};
// false: outside the kernel's domain
bool w4p_make_plan(int M, int N, int K, int64_t tickets_len, bool unsplit, W4PrefillPlan* out);
int w4p_launch(const W4PrefillPlan& pl, const GemmParams& p, bool f16, hipStream_t s);
// true: measured ahead of the tall kernel on a Marlin tensor of the same weights (what a caller that holds both should do)
bool w4p_wins(int M, int N, int K);
// the > 64 KiB dynamic-LDS opt-in (hipFuncSetAttribute) holds for the device that is current when it is made: one bit
// per device ordinal in *mask; true = this device has not opted in yet (ordinals past 63: always true)
bool lds_optin_needed(unsigned long long* mask, int device);

}  // namespace nmv
