// W4A16 GEMM on the GPTQ-Marlin interchange format, for gfx950 (MFMA 16x16x32, wave64).
//
// Behavioural reference: /root/reference/csrc/quantization/gptq_marlin/gptq_marlin.cu
//   (entry :1735-1868, host logic :1577-1731, kernel :396-1363) and the layout produced by
//   gptq_marlin_repack.cu / marlin_utils.py:25-57 + marlin_perms.py:16-50.
//   C[M,N] = A[M,K] . ((q - 8) * s[group(k), n]),  q = 4-bit code, fp32 accumulation.
//
// The Marlin tensor is consumed AS IS (no cached re-tile): in a row of 16 k ("k-tile") each
// 64-column chunk is 32 x 16 bytes; 16-byte vector i holds, for n_in = i/4 and q = i%4, the four
// n-tiles j=0..3 as one int32 each = nibbles {k 2q, 2q+8 | n_in} {2q, 2q+8 | n_in+8}
// {2q+1, 2q+9 | n_in} {2q+1, 2q+9 | n_in+8}.  That is NVIDIA mma.m16n8k16 fragment order, but it
// also factors onto v_mfma_f32_16x16x32: use the WEIGHTS as the MFMA "A" operand (16 rows = 16
// columns n of W) and the activations as "B" (16 columns = 16 tokens m).  Lane (r = l&15,
// g = l>>4) loads vector i = (r&7)*4 + g of chunk (r>>3) from two consecutive k-tiles (2 x 16 B,
// a wave covers 1 KiB contiguous per k-tile) and owns, for each of the 8 "variants" (j, n_in+8?)
// of its 8 columns, eight k values {2g,2g+1,2g+8,2g+9} of both k-tiles: exactly one MFMA
// operand.  The k order inside an MFMA is a fixed permutation, matched on the activation side
// when A is staged into LDS.
//
// Dequantisation costs ONE VALU op per two weights (+ shifts): (x >> s) & mask | magic places a
// nibble in the top mantissa bits of a bf16/fp16 whose exponent is 2^4, i.e. the exact value
// (16 + q).  Zero point and group scale are applied OUTSIDE the MFMA in fp32:
//      sum_k (q-8) a  =  sum_k (16+q) a  -  24 * sum_k a
// the second sum comes from one extra MFMA per k-step with an all-ones operand, and at every
// group boundary   acc_main += s[g,n] * (acc_group - 24 * S_group[m]).
// fp32 scaling is strictly more accurate than the reference's half-precision (q-8)*s products.
//
// Decomposition: a 256-thread workgroup = 4 waves arranged WN x WM x WK; a wave owns
// 128 columns x (16*MT) rows and walks its k range in 128-deep stages (8 x 16-byte weight loads
// per lane per stage, the next stage's loads are issued before the current stage is consumed).
// Activations of a stage go through LDS once per workgroup in MFMA-operand order.  Split-K
// across workgroups writes fp32 slabs that a second tiny kernel sums in a fixed order
// (bit-reproducible, unlike the reference's lock-based fp16 global reduce :1054-1110).
// HBM-bound for M <= 64: algorithmic bytes K*N/2 + (K/g)*N*2 + 2*M*K + 2*M*N.
#include "common.h"

namespace nmv {

constexpr int GT = 256;        // threads per workgroup
constexpr int STAGE_K = 128;   // k per pipeline stage (= 4 MFMA k-steps of 32)
constexpr int KSTEPS = STAGE_K / 32;

// (x & mask) | magic in ONE VALU op.  hipcc splits the C expression into v_and_b32 + v_or_b32 because
// gfx9 VOP3 encodings take no literals; with the mask in an SGPR and the magic in a VGPR the
// three-operand form is legal (one constant-bus read).
__device__ __forceinline__ uint32_t and_or(uint32_t x, uint32_t mask_sgpr, uint32_t magic_vgpr) {
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(mask_sgpr), "v"(magic_vgpr));
  return r;
}

template <typename T> struct W4;
template <> struct W4<BF16> {
  static constexpr uint32_t MASK = 0x00780078u, MAGIC = 0x41804180u, ONES = 0x3F803F80u;
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

struct GemmParams {
  const uint16_t* a;      // [M, K]
  const uint4* b;         // Marlin int32 [K/16, N*2] viewed as uint4 [K/16, N/2]
  const uint16_t* s;      // [num_groups, N] (marlin_permute_scales layout)
  const int* perm;        // [K] or null: A columns are gathered through it (act-order)
  uint16_t* c;            // [M, N] (used when splits == 1)
  float* slab;            // [splits, M, N] fp32 (used when splits > 1)
  int* tickets;           // [n_blocks * m_blocks] zero on entry / exit (the Marlin `workspace`)
  int M, N, K;
  int group_size;         // 32/64/128, or 0 = channelwise (one scale row)
  int k_per_wg;           // k range of one workgroup (multiple of WK*STAGE_K)
  int splits;
};

// ---------------------------------------------------------------------------------------------
template <typename T, int MT, int WN, int WM, int WK, int GS /* 0 = channelwise */>
__global__ __launch_bounds__(GT) void w4a16_gemm_kernel(const GemmParams p) {
  static_assert(WN * WM * WK == 4, "4 waves per workgroup");
  static_assert(GS == 0 || GS == 32 || GS == 64 || GS == 128, "group size");
  constexpr int FLUSH_EVERY = GS == 0 ? KSTEPS : GS / 32;  // k-steps per group
  constexpr int NG = GS == 0 ? 0 : STAGE_K / GS;           // scale groups per stage
  constexpr int NGA = NG > 0 ? NG : 1;
  constexpr int MP = 16 * MT * WM;          // activation rows staged per workgroup
  constexpr int A_STAGE_U4 = KSTEPS * 4 * MP;  // uint4 per (stage, k-group)
  // LDS: activation stages [2][WK][A_STAGE_U4]; re-used for the cross-wave reduction at the end
  constexpr int RED_U4 = (WK > 1) ? (WK - 1) * WN * WM * 8 * MT * 64 : 0;
  constexpr int LDS_U4 = (2 * WK * A_STAGE_U4 > RED_U4) ? 2 * WK * A_STAGE_U4 : RED_U4;
  __shared__ __attribute__((aligned(16))) uint4 lds[LDS_U4];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave % WN;
  const int wm = (wave / WN) % WM;
  const int wk = wave / (WN * WM);
  const int r = lane & 15, g = lane >> 4;

  const int n_chunks = p.N >> 6;
  const int chunk0 = (blockIdx.x * WN + wn) * 2;       // first 64-col chunk of this wave
  const int m0 = blockIdx.z * (16 * MT * WM);          // first row of this workgroup
  const int split = blockIdx.y;
  const int k_wg0 = split * p.k_per_wg;
  const int k_wg1 = min(k_wg0 + p.k_per_wg, p.K);
  const int k_per_wave = p.k_per_wg / WK;
  const int k_w0 = k_wg0 + wk * k_per_wave;             // this wave's k range
  const int k_w1 = min(k_w0 + k_per_wave, k_wg1);
  // all k-groups run the same number of stages so that barriers match
  const int n_stages = (min(k_per_wave, max(k_wg1 - k_wg0, 0)) + STAGE_K - 1) / STAGE_K;

  // ---- weight stream addressing (uint4 units) ----
  const int my_chunk = chunk0 + (r >> 3);
  const bool chunk_ok = my_chunk < n_chunks;
  const int64_t row_u4 = p.N >> 1;  // uint4 per k-tile row
  const uint4* bp = p.b + (int64_t)(chunk_ok ? my_chunk : 0) * 32 + ((r & 7) * 4 + g);

  auto load_stage_w = [&](int st, uint4 (&w)[2 * KSTEPS]) {
    const int kb = k_w0 + st * STAGE_K;
#pragma unroll
    for (int i = 0; i < 2 * KSTEPS; ++i) {
      const int k = kb + i * 16;
      if (chunk_ok && k < k_w1) w[i] = bp[(int64_t)(k >> 4) * row_u4];
      else w[i] = make_uint4(0, 0, 0, 0);
    }
  };

  // ---- activation staging: global -> registers -> LDS in MFMA operand order ----
  // chunk id -> (k-group, row, 8-wide k chunk c16); thread t handles ids t, t+256, ...
  constexpr int A_CHUNKS = WK * MP * 16;
  constexpr int A_PER_THREAD = (A_CHUNKS + GT - 1) / GT;
  auto load_stage_a = [&](int st, uint4 (&av)[A_PER_THREAD]) {
#pragma unroll
    for (int i = 0; i < A_PER_THREAD; ++i) {
      const int id = threadIdx.x + i * GT;
      const int c16 = id & 15;
      const int row = (id >> 4) % MP;
      const int kg = (id >> 4) / MP;
      const int m = m0 + row;
      const int kw0 = k_wg0 + kg * k_per_wave;
      const int k = kw0 + st * STAGE_K + c16 * 8;
      const int kend = min(kw0 + k_per_wave, k_wg1);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (id < A_CHUNKS && m < p.M && k < kend) {
        if (p.perm == nullptr) {
          v = ld16(p.a + (int64_t)m * p.K + k);
        } else {
          // act-order: A columns are gathered through perm (the reference materialises this
          // in a separate permute_cols_kernel, gptq_marlin.cu:345-394)
          uint16_t e[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) e[j] = p.a[(int64_t)m * p.K + p.perm[k + j]];
          v = make_uint4(e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16),
                         e[4] | ((uint32_t)e[5] << 16), e[6] | ((uint32_t)e[7] << 16));
        }
      }
      av[i] = v;
    }
  };
  auto store_stage_a = [&](int buf, const uint4 (&av)[A_PER_THREAD]) {
    uint32_t* base = reinterpret_cast<uint32_t*>(lds);
#pragma unroll
    for (int i = 0; i < A_PER_THREAD; ++i) {
      const int id = threadIdx.x + i * GT;
      if (id >= A_CHUNKS) continue;
      const int c16 = id & 15;
      const int row = (id >> 4) % MP;
      const int kg = (id >> 4) / MP;
      const int ks = c16 >> 2, cc = c16 & 3;
      // pair p of this 8-wide chunk belongs to lane group g = p, dword cc of its 16-byte entry
      const int e0 = ((buf * WK + kg) * A_STAGE_U4 + (ks * 4 + 0) * MP + row) * 4 + cc;
      base[e0] = av[i].x;
      base[e0 + 4 * MP] = av[i].y;
      base[e0 + 8 * MP] = av[i].z;
      base[e0 + 12 * MP] = av[i].w;
    }
  };

  // ---- accumulators ----
  f32x4_t accm[8][MT], accg[8][MT], accs[MT];
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int v = 0; v < 8; ++v)
#pragma unroll
    for (int t = 0; t < MT; ++t) { accm[v][t] = zero4; accg[v][t] = zero4; }
#pragma unroll
  for (int t = 0; t < MT; ++t) accs[t] = zero4;

  // scale addressing for the OUTPUT fragment of this lane: rows (g*4 + reg) of the MFMA tile
  // = columns chunk (chunk0 + (g>>1)), c64 = j*16 + blk*8 + (g&1)*4 + reg.  In the grouped
  // marlin_permute_scales layout the 8 variants (2j+blk) of one (g, reg) are 8 consecutive
  // elements: 4 x 16-byte loads per group and lane.
  const int out_chunk = chunk0 + (g >> 1);
  const bool out_ok = out_chunk < n_chunks;
  const uint16_t* sp = p.s + (int64_t)(out_ok ? out_chunk : 0) * 64 + (g & 1) * 32;
  // scales of one stage: NG groups x 4 x 16 B per lane, issued EARLY (with the weight prefetch
  // for one group per stage, else at the start of the stage) so that no flush waits on memory
  auto load_scales = [&](int st, uint4 (&sv)[NGA][4]) {
    if constexpr (NG > 0) {
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        const int k_abs = k_w0 + st * STAGE_K + gi * GS;
        const uint16_t* sg = sp + (int64_t)(min(k_abs, p.K - 1) / GS) * p.N;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) sv[gi][reg] = ld16(sg + reg * 8);
      }
    }
  };

  auto flush = [&](const uint4 (&sv)[NGA][4], int gi) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const float zs = -W4_ZP * accs[t][0];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const uint32_t d[4] = {sv[gi][reg].x, sv[gi][reg].y, sv[gi][reg].z, sv[gi][reg].w};
#pragma unroll
        for (int v = 0; v < 8; ++v) {
          const float dlt = accg[v][t][reg] + zs;
          if constexpr (GS != 0) {
            const float scv = (v & 1) ? hi_f<T>(d[v >> 1]) : lo_f<T>(d[v >> 1]);
            accm[v][t][reg] = fmaf(scv, dlt, accm[v][t][reg]);
          } else {
            accm[v][t][reg] += dlt;
          }
        }
      }
    }
  };

  // ---- prologue ----
  uint4 w0[2 * KSTEPS], w1[2 * KSTEPS];
  uint4 s0[NGA][4], s1[NGA][4];
  uint4 areg[A_PER_THREAD];
  if (n_stages > 0) {
    load_stage_w(0, w0);
    if constexpr (NG == 1) load_scales(0, s0);
    load_stage_a(0, areg);
    store_stage_a(0, areg);
  }
  __syncthreads();

  const uint4 ones = make_uint4(W4<T>::ONES, W4<T>::ONES, W4<T>::ONES, W4<T>::ONES);
  const uint32_t kmask = __builtin_amdgcn_readfirstlane(W4<T>::MASK);  // SGPR
  uint32_t kmagic = W4<T>::MAGIC;
  asm volatile("" : "+v"(kmagic));  // pin the magic constant in a VGPR
  const int a_rd_base = (wm * MT) * 16 + r;  // row of M-tile 0 for this lane

  // one pipeline stage: prefetch stage st+1 into (wn, sn), consume stage st from (wc, sc)
  auto stage = [&](int st, uint4 (&wc)[2 * KSTEPS], uint4 (&wn)[2 * KSTEPS], uint4 (&sc)[NGA][4],
                   uint4 (&sn)[NGA][4]) {
    const int buf = st & 1;
    const bool more = st + 1 < n_stages;
    if constexpr (NG > 1) load_scales(st, sc);  // before the prefetch: waits on it stay counted
    if (more) {
      load_stage_w(st + 1, wn);
      if constexpr (NG == 1) load_scales(st + 1, sn);
      load_stage_a(st + 1, areg);
    }
    const int kb = k_w0 + st * STAGE_K;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int k = kb + ks * 32;
      if (k < k_w1) {  // wave-uniform
        // activation fragments (B operand): one ds_read_b128 per M-tile
        uint4 af[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
          af[t] = lds[(buf * WK + wk) * A_STAGE_U4 + (ks * 4 + g) * MP + a_rd_base + t * 16];
        const uint4 x = wc[2 * ks], y = wc[2 * ks + 1];
        const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
        const uint32_t ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint4 wv0 = make_uint4(W4<T>::lo0(xs[j], kmask, kmagic), W4<T>::hi0(xs[j], kmask, kmagic),
                                       W4<T>::lo0(ys[j], kmask, kmagic), W4<T>::hi0(ys[j], kmask, kmagic));
          const uint4 wv1 = make_uint4(W4<T>::lo1(xs[j], kmask, kmagic), W4<T>::hi1(xs[j], kmask, kmagic),
                                       W4<T>::lo1(ys[j], kmask, kmagic), W4<T>::hi1(ys[j], kmask, kmagic));
          const bool first = (ks % FLUSH_EVERY) == 0;  // compile-time after unrolling
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            accg[2 * j][t] = W4<T>::mfma(wv0, af[t], first ? zero4 : accg[2 * j][t]);
            accg[2 * j + 1][t] = W4<T>::mfma(wv1, af[t], first ? zero4 : accg[2 * j + 1][t]);
          }
        }
#pragma unroll
        for (int t = 0; t < MT; ++t)
          accs[t] = W4<T>::mfma(ones, af[t], (ks % FLUSH_EVERY) == 0 ? zero4 : accs[t]);
        if ((ks + 1) % FLUSH_EVERY == 0) flush(sc, ks / FLUSH_EVERY);
      }
    }
    if (more) store_stage_a(buf ^ 1, areg);
    __syncthreads();
  };

  for (int st = 0; st < n_stages; st += 2) {
    stage(st, w0, w1, s0, s1);
    if (st + 1 < n_stages) stage(st + 1, w1, w0, s1, s0);
  }

  // channelwise + a trailing partial stage (K % 128 != 0): fold what is still pending
  if constexpr (GS == 0) {
    // only when the last stage ended before its 4th k-step (K % 128 != 0 inside this wave's range)
    const int done = k_w1 > k_w0 ? (k_w1 - k_w0) : 0;
    if (done % STAGE_K != 0) flush(s0, 0);
  }

  // ---- channelwise scales are applied once, on the fp32 result ----
  if (GS == 0 && out_ok) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const int j = v >> 1, blk = v & 1;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        // scale_perm_single (marlin_perms.py:45-49) position of column c64 inside its chunk
        const int c8 = (g & 1) * 4 + reg;                       // c % 8
        const int pos = (j >> 1) * 32 + (c8 >> 1) * 8 + 2 * (2 * (j & 1) + blk) + (c8 & 1);
        const float sv = T::to_float(p.s[(int64_t)out_chunk * 64 + pos]);
#pragma unroll
        for (int t = 0; t < MT; ++t) accm[v][t][reg] *= sv;
      }
    }
  }

  // ---- cross-wave (intra-workgroup) k reduction ----
  if constexpr (WK > 1) {
    float* red = reinterpret_cast<float*>(lds);
    // layout [(wk-1)][wn, wm][v][t][reg][lane]
    if (wk > 0) {
      float* dst = red + (((wk - 1) * WN * WM + (wm * WN + wn)) * 8 * MT * 4) * 64;
#pragma unroll
      for (int v = 0; v < 8; ++v)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) dst[((v * MT + t) * 4 + reg) * 64 + lane] = accm[v][t][reg];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int kk = 1; kk < WK; ++kk) {
        const float* src = red + (((kk - 1) * WN * WM + (wm * WN + wn)) * 8 * MT * 4) * 64;
#pragma unroll
        for (int v = 0; v < 8; ++v)
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) accm[v][t][reg] += src[((v * MT + t) * 4 + reg) * 64 + lane];
      }
    }
  }
  const bool writer = (wk == 0) && out_ok;  // this wave/lane owns output fragments

  // ---- epilogue: lane holds, per variant, 4 consecutive columns of row m ----
  if (p.splits == 1) {
    if (!writer) return;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int m = m0 + (wm * MT + t) * 16 + r;
      if (m >= p.M) continue;
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const int n = out_chunk * 64 + (v >> 1) * 16 + (v & 1) * 8 + (g & 1) * 4;
        const f32x4_t o = accm[v][t];
        uint2 pk;
        pk.x = T::pack2(o[0], o[1]);
        pk.y = T::pack2(o[2], o[3]);
        *reinterpret_cast<uint2*>(p.c + (int64_t)m * p.N + n) = pk;
      }
    }
    return;
  }

  // ---- split-K across workgroups: slabs + "last arriver reduces", in one launch ----
  // Hand-off per the CDNA4 rules for inter-workgroup data (per-CU L1 is never refreshed, XCD L2s
  // are not coherent): every slab byte is stored WRITE-THROUGH (sc1), each storing wave drains
  // vmcnt, the workgroup barrier orders them before ONE lane's agent-scope ticket add; the
  // workgroup that draws the last ticket reads every slab with sc1 loads (L1 bypass) and sums
  // them in split order -> the result is bit-reproducible, whichever workgroup arrives last.
  const int64_t slab_bytes = (int64_t)p.splits * p.M * p.N * 4;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.slab, 0, (int)slab_bytes, 0x00020000);
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
  if (writer) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int m = m0 + (wm * MT + t) * 16 + r;
      if (m >= p.M) continue;
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const int n = out_chunk * 64 + (v >> 1) * 16 + (v & 1) * 8 + (g & 1) * 4;
        const int off = (int)((((int64_t)split * p.M + m) * p.N + n) * 4);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, accm[v][t]), rs, off, 0, 16);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores
  __shared__ int ticket_s;
  __syncthreads();
  const int tile = blockIdx.z * gridDim.x + blockIdx.x;
  if (threadIdx.x == 0)
    ticket_s = __hip_atomic_fetch_add(p.tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (ticket_s != p.splits - 1) return;  // uniform for the workgroup
  if (threadIdx.x == 0)  // leave the ticket array zeroed for the next call
    __hip_atomic_store(p.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // all 256 threads of the last workgroup sum the tile: float4 columns strided over threads,
  // the split loop unrolled so that 8 independent sc1 loads are in flight per element
  constexpr int TILE_COLS = WN * 128;
  constexpr int TILE_ROWS = 16 * MT * WM;
  constexpr int F4_PER_ROW = TILE_COLS / 4;
  const int n_base = blockIdx.x * TILE_COLS;
  const int64_t split_stride = (int64_t)p.M * p.N * 4;  // bytes
  for (int e = threadIdx.x; e < TILE_ROWS * F4_PER_ROW; e += GT) {
    const int m = m0 + e / F4_PER_ROW;
    const int n = n_base + (e % F4_PER_ROW) * 4;
    if (m >= p.M || n >= p.N) continue;
    const int off0 = (int)(((int64_t)m * p.N + n) * 4);
    f32x4_t sum = {0.f, 0.f, 0.f, 0.f};
    int sidx = 0;
    for (; sidx + 8 <= p.splits; sidx += 8) {
      u32x4_t q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        q[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, off0 + (int)((sidx + u) * split_stride), 0, 16);
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += __builtin_bit_cast(f32x4_t, q[u]);  // fixed order
    }
    for (; sidx < p.splits; ++sidx) {
      const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rs, off0 + (int)(sidx * split_stride), 0, 16);
      sum += __builtin_bit_cast(f32x4_t, q);
    }
    uint2 pk;
    pk.x = T::pack2(sum[0], sum[1]);
    pk.y = T::pack2(sum[2], sum[3]);
    *reinterpret_cast<uint2*>(p.c + (int64_t)m * p.N + n) = pk;
  }
}

// ---------------------------------------------------------------------------------------------
// gptq_marlin_repack: GPTQ [K/pack, N] -> Marlin [K/16, N*16/pack].  One thread per output int32.
template <int BITS>
__global__ void marlin_repack_kernel(const uint32_t* __restrict__ qw, const int* __restrict__ perm,
                                     uint32_t* __restrict__ out, int K, int N) {
  constexpr int PACK = 32 / BITS;
  const int64_t row_words = (int64_t)N * 16 / PACK;
  const int64_t total = (int64_t)(K / 16) * row_words;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int kt = idx / row_words;
  const int col = idx % row_words;
  constexpr int WPC = 1024 / PACK;  // words per 64-column chunk (128 for 4-bit, 256 for 8-bit)
  const int chunk = col / WPC;
  const int rr = col % WPC;
  int i, j, blk_fixed;
  if constexpr (BITS == 4) { i = rr >> 2; j = rr & 3; blk_fixed = -1; }
  else { i = rr >> 3; j = (rr >> 1) & 3; blk_fixed = rr & 1; }
  const int q = i & 3, n_in = i >> 2;
  uint32_t res = 0;
#pragma unroll
  for (int pz = 0; pz < PACK; ++pz) {
    int k_in, blk;
    if constexpr (BITS == 4) {
      // nibble order after the {0,2,4,6,1,3,5,7} interleave (marlin_perms.py:33-41)
      k_in = 2 * q + ((pz & 1) ? 8 : 0) + (pz >> 2);
      blk = (pz >> 1) & 1;
    } else {
      // byte order after the {0,2,1,3} interleave
      k_in = 2 * q + ((pz & 1) ? 8 : 0) + (pz >> 1);
      blk = blk_fixed;
    }
    const int n = chunk * 64 + j * 16 + blk * 8 + n_in;
    const int k = kt * 16 + k_in;
    const int ks = perm ? perm[k] : k;
    const uint32_t w = qw[(int64_t)(ks / PACK) * N + n];
    const uint32_t val = (w >> (BITS * (ks % PACK))) & ((1u << BITS) - 1);
    res |= val << (BITS * pz);
  }
  out[idx] = res;
}

// ---------------------------------------------------------------------------------------------
struct GemmPlan {
  int mt, wn, wm, wk;  // kernel shape
  int splits, k_per_wg, m_blocks, n_blocks;
};

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// Pick the workgroup shape and the split-K factor.  Decode-sized GEMMs (M <= 64) last only a
// few microseconds at HBM speed, so the plan aims at >= ~2 workgroups per CU while keeping the
// fp32 partial traffic (splits * M * N * 4 B) well below the weight bytes (K * N / 2).
static GemmPlan make_plan(int M, int N, int K, int64_t tickets_len) {
  GemmPlan pl;
  const int n_chunks = N / 64;
  // Measured on MI355X (tools/bench_gemm.py --sweep): the 16-row tile (MT = 1, 2 workgroups per
  // CU) beats the 32/64-row tiles at every M <= 64, even though 64 rows re-dequantise the weights
  // four times; larger M walks blockIdx.z.  ~512 workgroups (one resident set) is the sweet spot.
  pl.wm = 1;
  pl.mt = env_int("NMV_W4_MT", 1);
  const int rows_per_wg = 16 * pl.mt;
  pl.m_blocks = (M + rows_per_wg - 1) / rows_per_wg;
  if (n_chunks >= 256) pl.wn = (pl.m_blocks == 1 && M <= 8) ? 4 : 2;
  else pl.wn = pl.m_blocks >= 2 ? 2 : 1;
  if (K < (4 / pl.wn) * STAGE_K) pl.wn = 4;
  pl.wn = env_int("NMV_W4_WN", pl.wn);
  if (pl.mt == 4 && pl.wn == 1) pl.wn = 2;  // the 64-row tile keeps WK <= 2 (LDS / registers)
  pl.wk = 4 / pl.wn;
  pl.n_blocks = (n_chunks + 2 * pl.wn - 1) / (2 * pl.wn);
  const int unit = pl.wk * STAGE_K;  // k granularity of a workgroup
  const int k_units = (K + unit - 1) / unit;
  const int base_wgs = pl.n_blocks * pl.m_blocks;
  int splits = std::max(1, 512 / base_wgs);
  splits = std::min(splits, 32);
  splits = env_int("NMV_W4_SPLITS", splits);
  splits = std::max(1, std::min(splits, k_units));
  if ((int64_t)base_wgs > tickets_len) splits = 1;  // no ticket per output tile available
  const int units_per_wg = (k_units + splits - 1) / splits;
  pl.k_per_wg = units_per_wg * unit;
  pl.splits = (K + pl.k_per_wg - 1) / pl.k_per_wg;
  return pl;
}

template <typename T, int GS>
static int launch_gemm_gs(const GemmPlan& pl, const GemmParams& p, hipStream_t s) {
  dim3 grid(pl.n_blocks, pl.splits, pl.m_blocks), block(GT);
#define NMV_W4_CASE(MT_, WN_, WM_, WK_)                                                         \
  if (pl.mt == MT_ && pl.wn == WN_ && pl.wm == WM_ && pl.wk == WK_) {                           \
    hipLaunchKernelGGL((w4a16_gemm_kernel<T, MT_, WN_, WM_, WK_, GS>), grid, block, 0, s, p);   \
    return 0;                                                                                   \
  }
  NMV_W4_CASE(1, 4, 1, 1)
  NMV_W4_CASE(1, 2, 1, 2)
  NMV_W4_CASE(1, 1, 1, 4)
#ifdef NMV_W4_TALL_TILES  // 32- and 64-row tiles: slower than 16 rows x blockIdx.z on MI355X
  NMV_W4_CASE(2, 4, 1, 1)
  NMV_W4_CASE(2, 2, 1, 2)
  NMV_W4_CASE(2, 1, 1, 4)
  NMV_W4_CASE(4, 4, 1, 1)
  NMV_W4_CASE(4, 2, 1, 2)
#endif
#undef NMV_W4_CASE
  return -1;
}

template <typename T>
static int launch_gemm(const GemmPlan& pl, const GemmParams& p, hipStream_t s) {
  switch (p.group_size) {
    case 0: return launch_gemm_gs<T, 0>(pl, p, s);
    case 32: return launch_gemm_gs<T, 32>(pl, p, s);
    case 64: return launch_gemm_gs<T, 64>(pl, p, s);
    case 128: return launch_gemm_gs<T, 128>(pl, p, s);
    default: return -1;
  }
}

// csrc/wq_generic.hip: LDS-staged generic path for the Marlin variants not covered above
int wq_marlin_fallback(void* c, const void* a, const int32_t* b_q_weight, const void* b_scales,
                       const int32_t* g_idx, const int32_t* perm, int num_bits, int size_m,
                       int size_n, int size_k, int num_groups, int is_fp8, nmv_dtype_t dtype,
                       hipStream_t stream);

}  // namespace nmv

using namespace nmv;

extern "C" int nmv_gptq_marlin_repack(const int32_t* b_q_weight, const int32_t* perm, int32_t* out,
                                      int size_k, int size_n, int num_bits, void* stream) {
  NMV_CHECK(num_bits == 4 || num_bits == 8, "num_bits must be 4 or 8. Got = %d", num_bits);
  NMV_CHECK(size_k % 16 == 0, "size_k = %d is not divisible by tile_k_size = 16", size_k);
  NMV_CHECK(size_n % 64 == 0, "size_n = %d is not divisible by tile_n_size = 64", size_n);
  const int pack = 32 / num_bits;
  const int64_t total = (int64_t)(size_k / 16) * ((int64_t)size_n * 16 / pack);
  if (total == 0) return NMV_OK;
  dim3 grid((unsigned)cdiv64(total, 256)), block(256);
  if (num_bits == 4)
    hipLaunchKernelGGL((marlin_repack_kernel<4>), grid, block, 0, (hipStream_t)stream,
                       (const uint32_t*)b_q_weight, perm, (uint32_t*)out, size_k, size_n);
  else
    hipLaunchKernelGGL((marlin_repack_kernel<8>), grid, block, 0, (hipStream_t)stream,
                       (const uint32_t*)b_q_weight, perm, (uint32_t*)out, size_k, size_n);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int64_t nmv_gptq_marlin_gemm_scratch_bytes(int size_m, int size_n, int size_k,
                                                      int has_act_order) {
  (void)has_act_order;  // the A gather is fused into the LDS staging: no a_tmp copy
  if (size_m <= 0 || size_n <= 0 || size_k <= 0) return 0;
  // upper bound over every plan the entry point may pick (the ticket array only lowers splits)
  const GemmPlan pl = make_plan(size_m, size_n, size_k, INT64_MAX);
  return pl.splits > 1 ? (int64_t)pl.splits * size_m * size_n * 4 : 0;
}

extern "C" int nmv_gptq_marlin_gemm(void* c, const void* a, const int32_t* b_q_weight,
                                    const void* b_scales, const int32_t* g_idx,
                                    const int32_t* perm, int32_t* workspace,
                                    int64_t workspace_len, void* scratch, int64_t scratch_bytes,
                                    int num_bits, int size_m, int size_n, int size_k,
                                    int num_groups, int is_k_full, nmv_dtype_t dtype,
                                    void* stream) {
  // `workspace` (the reference's lock array, zero on entry and exit) holds the split-K tickets
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16,
            "gpt_marlin_gemm only supports bfloat16 and float16");
  NMV_CHECK(num_bits == 4 || num_bits == 8, "num_bits must be 4 or 8. Got = %d", num_bits);
  NMV_CHECK(size_m > 0 && size_n > 0 && size_k > 0, "Invalid MNK = [%d, %d, %d]", size_m, size_n,
            size_k);
  NMV_CHECK(size_n % 64 == 0, "size_n = %d is not divisible by min_thread_n = 64", size_n);
  NMV_CHECK(size_k % 32 == 0, "size_k = %d is not divisible by 32", size_k);
  NMV_CHECK(num_groups >= 1, "num_groups must be >= 1");
  const bool has_act_order = (g_idx != nullptr && perm != nullptr);
  int group_size;
  if (num_groups > 1 && has_act_order && !is_k_full) {
    group_size = -1;  // a K shard carries the scales of every group; rows are mapped by g_idx
  } else if (num_groups > 1) {
    NMV_CHECK(size_k % num_groups == 0, "size_k = %d, is not divisible by b_scales.size(0) = %d",
              size_k, num_groups);
    group_size = size_k / num_groups;
  } else {
    group_size = 0;
  }
  NMV_CHECK(group_size == 0 || group_size == 32 || group_size == 64 || group_size == 128 ||
                (has_act_order && !is_k_full),
            "Unsupported group_size = %d", group_size);
  if (num_bits == 8 || (has_act_order && !is_k_full)) {
    // 8-bit codes, and act-order on a K shard (irregular group runs: one scale row per k through
    // g_idx), take the generic LDS-staged kernel; same math, not HBM-tuned yet
    const int rc = wq_marlin_fallback(c, a, b_q_weight, b_scales, has_act_order ? g_idx : nullptr,
                                      has_act_order ? perm : nullptr, num_bits, size_m, size_n,
                                      size_k, num_groups, 0, dtype, (hipStream_t)stream);
    NMV_CHECK(rc == 0, "gptq_marlin_gemm: generic path launch failed");
    NMV_LAUNCH_CHECK();
    return NMV_OK;
  }
  const GemmPlan pl = make_plan(size_m, size_n, size_k, workspace ? workspace_len : 0);
  const int64_t need = pl.splits > 1 ? (int64_t)pl.splits * size_m * size_n * 4 : 0;
  NMV_CHECK(need < (int64_t)1 << 31, "gptq_marlin_gemm: split-K slab too large");
  NMV_CHECK(scratch_bytes >= need && (need == 0 || scratch != nullptr),
            "gptq_marlin_gemm: scratch too small (%lld < %lld)", (long long)scratch_bytes,
            (long long)need);
  GemmParams p;
  p.a = (const uint16_t*)a;
  p.b = (const uint4*)b_q_weight;
  p.s = (const uint16_t*)b_scales;
  p.perm = has_act_order ? perm : nullptr;
  p.c = (uint16_t*)c;
  p.slab = (float*)scratch;
  p.tickets = workspace;
  p.M = size_m; p.N = size_n; p.K = size_k;
  p.group_size = group_size;
  p.k_per_wg = pl.k_per_wg;
  p.splits = pl.splits;
  hipStream_t s = (hipStream_t)stream;
  const int rc = dtype == NMV_F16 ? launch_gemm<F16>(pl, p, s) : launch_gemm<BF16>(pl, p, s);
  NMV_CHECK(rc == 0, "gptq_marlin_gemm: no kernel for plan mt=%d wn=%d wm=%d wk=%d", pl.mt, pl.wn,
            pl.wm, pl.wk);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

/* legacy Marlin checkpoints (csrc/quantization/marlin/dense/marlin_cuda_kernel.cu:1045-1136):
 * 4-bit, group -1 / 128, no act-order -- the same tile and scale layout as gptq_marlin */
extern "C" int nmv_marlin_gemm(void* c, const void* a, const int32_t* b_q_weight,
                               const void* b_scales, int32_t* workspace, int64_t workspace_len,
                               void* scratch, int64_t scratch_bytes, int size_m, int size_n,
                               int size_k, int num_groups, nmv_dtype_t dtype, void* stream) {
  NMV_CHECK(size_k % 128 == 0, "size_k = %d is not divisible by min_thread_k = 128", size_k);
  NMV_CHECK(num_groups >= 1 && size_k % num_groups == 0, "marlin_gemm: bad number of scale groups");
  const int gs = num_groups > 1 ? size_k / num_groups : -1;
  NMV_CHECK(gs == -1 || gs == 128, "Unexpected groupsize = %d", gs);
  return nmv_gptq_marlin_gemm(c, a, b_q_weight, b_scales, nullptr, nullptr, workspace,
                              workspace_len, scratch, scratch_bytes, 4, size_m, size_n, size_k,
                              num_groups, 1, dtype, stream);
}
