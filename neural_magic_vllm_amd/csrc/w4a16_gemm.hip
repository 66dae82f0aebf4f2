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
// Three kernels live here:
//   * w4a16_gemm_tall_kernel -- the default: one wave = one 64-column chunk x 16/32/64/128 rows,
//     weights straight from registers for all its row tiles (a DPP exchange between lane pairs
//     replaces the second load), 4-slot register ring, in-workgroup K split; 4- and 8-bit codes;
//     group 128 / channelwise, no act-order, K in whole rings.
//   * w4a16_gemm_kernel -- the first kernel of this round (a wave = 128 columns x 16 rows, the
//     lane mapping described above): groups of 32 / 64, act-order with the full K, K % 256 != 0.
//   * w4n_gemm_kernel -- the tall kernel's schedule on an MFMA-native weight tensor (one contiguous 1 KB
//     wave-load per k-step, no lane exchange, 30 % fewer VALU instructions): opt-in (NMV_W4_NATIVE=1 in
//     GPTQMarlinLinearMethod) because it measured within +-3 % of the tall kernel and needs a second copy
//     of the weights (DESIGN.md 3.2 "What was tried").
// All stage the activations of a stage through LDS once per workgroup in MFMA-operand order and
// split K across workgroups into fp32 slabs that the LAST workgroup of a tile (ticket in the
// Marlin `workspace`) sums in a fixed order inside the same launch: bit-reproducible, unlike the
// reference's lock-based fp16 global reduce (:1054-1110).
// Two optional epilogues of the tall kernel (GemmParams::epi; neither is an op of the reference, both
// are bit-identical to the op sequence they replace): 1 = silu(gate) * up on column-interleaved
// gate_up weights, 2 = deferred reduction -- the slabs are left for the next launch of the layer
// (residual-add + RMSNorm, rope + cache write) to sum, which takes the ticket round trip and the
// slab re-read (2-3 us per call at decode sizes) off the critical path.
// HBM-bound for M <= 64: algorithmic bytes K*N/2 + (K/g)*N*2 + 2*M*K + 2*M*N.
#include <type_traits>

#include "common.h"
#include "w4a16_common.h"

namespace nmv {

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
  __syncthreads();  // the LDS stages are dead: reuse them for the partial sums
  splitk_reduce_tile<T>(p, rs, m0, 16 * MT * WM, blockIdx.x * (WN * 128), WN * 128,
                        reinterpret_cast<f32x4_t*>(lds));
}

// ---------------------------------------------------------------------------------------------
// Epilogue shared by the tall and the native-layout kernel: a wave holds accm[j][t] = its 64-column chunk x
// 16 MT rows over its k range (lane (r, g): columns chunk*64 + 16 j + 4 g + reg, rows m0 + 16 t + r).
// Channelwise scales, the in-workgroup K reduction through LDS, then one of: model-dtype store, the
// silu(gate) * up store, fp32 slabs + ticket + last-arriver sum, or slabs only (deferred reduction).
template <typename T, int MT, int WN, int WK, int GS, bool NAT = false>
__device__ __forceinline__ void w4_tall_epilogue(const GemmParams& p, f32x4_t (&accm)[4][MT], uint4* lds,
                                                 int wn, int wk, int lane, int r, int g, int chunk,
                                                 bool chunk_ok, int m0, int split) {
  constexpr int MP = 16 * MT;
  // ---- channelwise scales are applied once, on the fp32 result ----
  if (GS == 0 && chunk_ok) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int c64 = j * 16 + 4 * g + reg;
        const int c = c64 & 31;
        // Marlin tensors carry channelwise scales in scale_perm_single order, native ones in column order
        const int pos = NAT ? c64 : (c64 >> 5) * 32 + ((c & 7) >> 1) * 8 + 2 * (c >> 3) + (c & 1);
        const float sv = T::to_float(p.s[(int64_t)chunk * 64 + pos]);
#pragma unroll
        for (int t = 0; t < MT; ++t) accm[j][t][reg] *= sv;
      }
    }
  }

  // ---- cross-wave (intra-workgroup) k reduction ----
  if constexpr (WK > 1) {
    float* red = reinterpret_cast<float*>(lds);
    __syncthreads();  // nobody reads the operand images any more
    if (wk > 0) {
      float* dst = red + (int64_t)((wk - 1) * WN + wn) * (4 * MT * 4) * 64;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) dst[((j * MT + t) * 4 + reg) * 64 + lane] = accm[j][t][reg];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int kk = 1; kk < WK; ++kk) {
        const float* src = red + (int64_t)((kk - 1) * WN + wn) * (4 * MT * 4) * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) accm[j][t][reg] += src[((j * MT + t) * 4 + reg) * 64 + lane];
      }
    }
  }
  const bool writer = (wk == 0) && chunk_ok;

  // ---- epilogue ----
  if (p.splits == 1 && p.epi != 2) {
    if (!writer) return;
    if (p.epi) {
      // gate_up with silu_and_mul folded in: the weight columns were interleaved at load time so
      // that a chunk holds gate[32 c .. 32 c + 31] in tiles 0, 1 and up[32 c .. 32 c + 31] in tiles
      // 2, 3 -- the lane that holds a gate element holds its up element.  Same roundings as the two
      // ops it replaces: both GEMM outputs rounded to the model dtype, silu in fp32 rounded to the
      // model dtype (activation_kernels.cu:14-26), product rounded.
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int m = m0 + t * 16 + r;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float o[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float gb = round_trip<T>(accm[j][t][i]), ub = round_trip<T>(accm[j + 2][t][i]);
            o[i] = round_trip<T>(gb / (1.0f + expf(-gb))) * ub;
          }
          uint2 pk;
          pk.x = T::pack2(o[0], o[1]);
          pk.y = T::pack2(o[2], o[3]);
          *reinterpret_cast<uint2*>(p.c + (int64_t)m * (p.N >> 1) + chunk * 32 + j * 16 + 4 * g) = pk;
        }
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int m = m0 + t * 16 + r;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = chunk * 64 + j * 16 + 4 * g;
        const f32x4_t o = accm[j][t];
        uint2 pk;
        pk.x = T::pack2(o[0], o[1]);
        pk.y = T::pack2(o[2], o[3]);
        *reinterpret_cast<uint2*>(p.c + (int64_t)m * p.N + n) = pk;
      }
    }
    return;
  }
  // split-K across workgroups: sc1 slabs, ticket, the last workgroup of the tile reduces (see above)
  const int64_t slab_bytes = (int64_t)p.splits * p.M * p.N * 4;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.slab, 0, (int)slab_bytes, 0x00020000);
  if (writer) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int m = m0 + t * 16 + r;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = chunk * 64 + j * 16 + 4 * g;
        const int off = (int)((((int64_t)split * p.M + m) * p.N + n) * 4);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, accm[j][t]), rs, off, 0, 16);
      }
    }
  }
  if (p.epi == 2) return;  // deferred: the next kernel in the stream sums the slabs (kernel boundary = release)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __shared__ int ticket_s;
  __syncthreads();
  const int tile = blockIdx.z * gridDim.x + blockIdx.x;
  if (threadIdx.x == 0)
    ticket_s = __hip_atomic_fetch_add(p.tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (ticket_s != p.splits - 1) return;
  if (threadIdx.x == 0)
    __hip_atomic_store(p.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  splitk_reduce_tile<T>(p, rs, m0, MP, blockIdx.x * (WN * 64), WN * 64, reinterpret_cast<f32x4_t*>(lds));
}

// ---------------------------------------------------------------------------------------------
// The default kernel: "tall" register tile -- one wave = ONE 64-column chunk x 16 MT rows
// (MT = 1 for M <= 16, 2 up to 64, 4 beyond).
//
// The 16-row kernel above re-expands every weight once per 16 rows and prefetches one 128-k stage.
// A workgroup-shared expansion through LDS (tried: 128 x 64 tile, weights expanded once per
// workgroup into an MFMA-operand image) removes the redundancy but pays a serial
// wait -> ds_read -> MFMA -> expand -> ds_write -> barrier chain per 64 k and ended up no faster
// (DESIGN.md 3.3).  Here the expansion is shared the cheap way: the wave that expanded a weight
// multiplies it against ALL its MT row tiles, straight from registers -- no LDS round trip and no
// barrier for the weights.  A 64-column chunk has 16-column MFMA tiles j = 0..3 whose rows are
// r = 8 blk + n_in, so lane (r, g) needs the blk half of Marlin vector (n_in, q = g) from both
// k-tiles of the 32-deep step.  Lanes r and r^8 want the two halves of the same two vectors: the
// blk-0 lane loads the even k-tile, the blk-1 lane the odd one (every byte is loaded once, one
// 16-byte load per lane and k-step), they swap through one DPP row rotate (row_ror:8), and each
// lane expands its own half with per-lane rotate amounts (v_alignbit_b32 + v_and_or_b32 = 2 ops
// per 2 weights, as everywhere).  Weights run 3 stages ahead in a 4-slot register ring of 4-8
// VGPRs per slot; activations are staged through LDS as in the 16-row kernel and the group's scale
// rows travel with them.  The 4 waves split N (WN) and K (WK); the in-workgroup K reduction goes
// through LDS, so narrow projections need few split-K slabs.  Main loop: branch-free, whole rings
// (the plan only picks this kernel when every wave's k range is a multiple of 4 stages).
// Stage = 64 k (two MFMA k-steps) for the 32-row tile, 32 k for the 64-row tile: a 64-row k-step
// carries 20 MFMAs, so the same look-ahead in time needs half the k -- and half the ring and
// staging registers, which is what lets 144 accumulator VGPRs fit under 256.

// PS ("prescale", the prefill variant): the group scale and the zero point are applied to the
// expanded weights in fp32 and the product is rounded to the model dtype -- the reference Marlin
// kernel's own semantics (gptq_marlin.cu:269-278) -- instead of in fp32 on the per-group
// accumulators.  That costs 5 more VALU ops per weight pair but needs no group accumulators and
// no sum-of-activations MFMA, which is what lets a wave carry 128 rows (MT = 8): in prefill the
// expansion is then shared by 128 rows and the MFMA pipe, not the VALU, is the busy one.
// BITS = 8 (W8A16 GPTQ-Marlin): the same kernel on the 8-bit Marlin tensor (int32 [K/16, N*4], a
// vector = 32 bytes = words (tile j, block) of one k-tile, bytes = k {2q, 2q+8, 2q+1, 2q+9}): a lane
// loads the whole vector of its k-tile, the two lanes of a pair trade the words of the other block
// through the same bank-masked DPP rotate, and a byte becomes a model-dtype number with one
// v_perm_b32 per pair (fp16: 0x6400 | b = 1024 + b) or v_cvt_f32_ubyte + pack (bf16: b itself);
// the zero point (128, resp. 1024 + 128) leaves through the sum-of-activations correction.
// ZP: per-(group, column) zero points (asymmetric AWQ / GPTQ checkpoints repacked to the Marlin
// layout): p.zp holds z in the model dtype in the layout of the scales; they travel through LDS
// with the scales (threads 32..63 stage them) and replace the constant 8 in the correction term.
// F8 (BITS = 8, channelwise scales: fp8_marlin_gemm, fp8_marlin.cu:1212-1308): the bytes of the 8-bit Marlin tensor are
// fp8-e4m3 numbers; a pair of them becomes a model-dtype pair through the hardware's exact conversion (v_cvt_pk_f32_fp8 +
// pack: subnormals included, no exponent-bias multiply), there is no zero point and hence no sum-of-activations MFMA.
template <typename T, int MT, int WN, int WK, int GS, bool PS = false, int BITS = 4, bool ZP = false, bool NT = false,
          bool F8 = false>
__global__ __launch_bounds__(GT, 2) void w4a16_gemm_tall_kernel(const GemmParams p) {
  static_assert(!F8 || (BITS == 8 && GS == 0 && !ZP), "fp8 bytes: 8-bit Marlin tensor, channelwise scales");
  static_assert(!ZP || (BITS == 4 && GS == 128), "zero points: 4-bit, group 128");
  static_assert(WN * WK == 4, "4 waves per workgroup");
  static_assert(BITS == 4 || (BITS == 8 && !PS), "4-bit, or 8-bit without the prescale variant");
  constexpr int WV = BITS / 4;                 // 16-byte loads per lane and k-step
  // what the expanded numbers are offset by: 16 + 8 (4-bit), 128 (8-bit bf16), 1024 + 128 (8-bit fp16)
  constexpr float ZPC = F8 ? 0.0f : BITS == 4 ? W4_ZP : (std::is_same<T, F16>::value ? 1152.0f : 128.0f);
  static_assert(MT <= 4 || PS, "128-row tiles only fit without the group accumulators");
  static_assert(GS == 0 || GS == 128, "one scale group per two stages, or channelwise");
  constexpr int MP = 16 * MT;                  // rows per workgroup
  constexpr int KSS = MT >= 4 ? 1 : 2;         // MFMA k-steps per stage
  constexpr int TS_K = 32 * KSS;               // k per stage
  constexpr int SPG = 4 / KSS;                 // stages per 128-k scale group
  constexpr int PPR = 4 * KSS;                 // 16-byte activation pieces per row and stage
  // [k-step][g] planes of MP rows x 16 B (a plane read back by ds_read_b128 is conflict-free exactly with this
  // 256-byte row pitch: the instruction's 16-lane groups are {0-3,12-15,20-27}, ...).  With two k-steps per stage a
  // staging write instruction stores one dword of rows 0..3 into plane (k-step 0, g) AND (k-step 1, g) -- a multiple
  // of 256 B apart, the same 16 of the 32 write banks twice, four times with the paired ds_write2 -- so the row index
  // of k-step 1's planes is XORed with 4: rows 0..3 go to banks 16..31 there (PMC: SQ_LDS_BANK_CONFLICT was 30 % of
  // SQ_LDS_IDX_ACTIVE); the read applies the same XOR and stays a permutation of a plane's 16 rows.
  constexpr int PSTR = MP;
  constexpr int A_U4 = KSS * 4 * PSTR;         // uint4 per (stage, k-group): [k-step][g][row]
  constexpr int SC_U4 = 4 * 8;                 // one scale group: [k-group * WN + wn] x 128 B
  constexpr int MAIN_U4 = 2 * WK * A_U4 + (ZP ? 4 : 2) * SC_U4;
  constexpr int RED_U4 = (WK > 1) ? (WK - 1) * WN * MT * 256 : 0;
  constexpr int LDS_U4 = MAIN_U4 > RED_U4 ? (MAIN_U4 > GT ? MAIN_U4 : GT) : (RED_U4 > GT ? RED_U4 : GT);
  __shared__ __attribute__((aligned(16))) uint4 lds[LDS_U4];
  uint4* a_s = lds;
  uint4* sc_s = lds + 2 * WK * A_U4;
  uint4* zp_s = sc_s + 2 * SC_U4;              // [2][SC_U4] zero points (ZP only)

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave % WN, wk = wave / WN;
  const int r = lane & 15, g = lane >> 4;
  const int blk = r >> 3, n_in = r & 7;
  const int n_chunks = p.N >> 6;
  const int chunk = blockIdx.x * WN + wn;
  const bool chunk_ok = chunk < n_chunks;
  const int m0 = blockIdx.z * MP;
  const int split = blockIdx.y;
  const int k_wg0 = split * p.k_per_wg;
  const int k_wg1 = min(k_wg0 + p.k_per_wg, p.K);
  const int k_per_wave = (k_wg1 - k_wg0) / WK;      // multiple of 4 stages (make_plan)
  const int k_w0 = k_wg0 + wk * k_per_wave;
  const int n_stages = k_per_wave / TS_K;           // multiple of 4
  // a stage index past the end (the ring's look-ahead) re-reads the last stage; never consumed
  const int st_last = __builtin_amdgcn_readfirstlane(n_stages - 1);

  // ---- weights: lane (blk, n_in, q = g) streams vector n_in*4+q of k-tile 2 ks + blk ----
  const int64_t row_u4 = (int64_t)(p.N >> 1) * WV;
  const uint4* bp = p.b + ((int64_t)(chunk_ok ? chunk : 0) * 32 + (n_in * 4 + g)) * WV +
                    ((int64_t)(k_w0 >> 4) + blk) * row_u4;
  auto load_w = [&](int st, uint4 (&w)[KSS * WV]) {
    const uint4* q = bp + (int64_t)min(st, st_last) * (2 * KSS * row_u4);
#pragma unroll
    for (int ks = 0; ks < KSS; ++ks)
#pragma unroll
      for (int h = 0; h < WV; ++h) w[ks * WV + h] = ld_stream<NT>(q + ks * 2 * row_u4 + h);
  };
  const uint32_t kmask = __builtin_amdgcn_readfirstlane(W4<T>::MASK);
  uint32_t kmagic = W4<T>::MAGIC;
  asm volatile("" : "+v"(kmagic));
  const uint32_t rot_lo = blk ? W4<T>::ROT_LO1 : W4<T>::ROT_LO0;
  const uint32_t rot_hi = blk ? W4<T>::ROT_HI1 : W4<T>::ROT_HI0;

  // ---- activations (+ the scale rows of the group) -> registers -> LDS ----
  constexpr int A_CHUNKS = WK * MP * PPR;                  // 16-byte pieces per stage
  constexpr int APT = (A_CHUNKS + GT - 1) / GT;
  // fewer pieces than threads (16 rows, one k group): the upper threads repeat the lower ones'
  // pieces -- same value to the same LDS word, no branch in the loop
  static_assert(A_CHUNKS % GT == 0 || A_CHUNKS < GT, "no ragged tail");
  const uint16_t* ap[APT];
#pragma unroll
  for (int i = 0; i < APT; ++i) {
    const int id = (threadIdx.x + i * GT) % A_CHUNKS;
    const int c8 = id % PPR, row = (id / PPR) % MP, kg = (id / PPR) / MP;
    ap[i] = p.a + (int64_t)min(m0 + row, p.M - 1) * p.K + (k_wg0 + kg * k_per_wave) + c8 * 8;
  }
  auto load_a = [&](int st, uint4 (&av)[APT]) {
    // rows past M are clamped duplicates that only feed rows which are never stored
    const int off = min(st, st_last) * TS_K;
#pragma unroll
    for (int i = 0; i < APT; ++i) av[i] = ld16(ap[i] + off);
  };
  auto store_a = [&](int buf, const uint4 (&av)[APT]) {
    uint32_t* base = reinterpret_cast<uint32_t*>(a_s);
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      const int id = (threadIdx.x + i * GT) % A_CHUNKS;
      const int c8 = id % PPR, row = (id / PPR) % MP, kg = (id / PPR) / MP;
      const int ks = c8 >> 2, cc = c8 & 3;
      const int e0 = ((buf * WK + kg) * A_U4 + (ks * 4 + 0) * PSTR + (row ^ (ks << 2))) * 4 + cc;
      base[e0] = av[i].x;
      base[e0 + 4 * PSTR] = av[i].y;
      base[e0 + 8 * PSTR] = av[i].z;
      base[e0 + 12 * PSTR] = av[i].w;
    }
  };
  // scale rows: threads 0..31 = (k-group*WN + wn') x 8 pieces of 16 B (64 columns x 2 B)
  const int s_combo = (threadIdx.x >> 3) & 3, s_piece = threadIdx.x & 7;
  const int s_chunk = min(blockIdx.x * WN + (s_combo % WN), n_chunks - 1);
  // threads 0..31 fetch scales, threads 32..63 the zero points (same geometry)
  const bool s_is_zp = ZP && threadIdx.x >= 32;
  const uint16_t* s_src = (s_is_zp ? p.zp : p.s) + (int64_t)s_chunk * 64 + s_piece * 8;
  const int s_k0 = k_wg0 + (s_combo / WN) * k_per_wave;
  auto load_sc = [&](int st) -> uint4 {
    if constexpr (GS == 0) return make_uint4(0, 0, 0, 0);
    const int k_abs = min(s_k0 + st * TS_K, p.K - 1);
    return ld16(s_src + (int64_t)(k_abs / 128) * p.N);
  };
  auto store_sc = [&](int gbuf, uint4 v) {
    if constexpr (GS != 0) {
      if (threadIdx.x < 32) sc_s[gbuf * SC_U4 + threadIdx.x] = v;
      if constexpr (ZP) {
        if (threadIdx.x >= 32 && threadIdx.x < 64) zp_s[gbuf * SC_U4 + threadIdx.x - 32] = v;
      }
    }
  };

  // ---- accumulators ----
  constexpr int GT_ = PS ? 1 : MT;             // group accumulators exist only without PS
  f32x4_t accm[4][MT], accg[4][GT_], accs[GT_];
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < MT; ++t) accm[j][t] = zero4;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < GT_; ++t) accg[j][t] = zero4;
#pragma unroll
  for (int t = 0; t < GT_; ++t) accs[t] = zero4;
  // PS: scale and -24 * scale of this lane's four weight rows (columns 16 j + r) for the open group
  float ps_s[4] = {1.f, 1.f, 1.f, 1.f}, ps_z[4] = {-W4_ZP, -W4_ZP, -W4_ZP, -W4_ZP};

  // output fragment of this lane: columns chunk*64 + 16 j + 4 g + reg, rows m0 + 16 t + r.
  // grouped scale layout: element (4 (g&1) + reg) * 8 + 2 j + (g >> 1) of the chunk's 64
  const uint32_t sc_shift = (g >> 1) * 16;
  auto flush = [&](int gbuf) {
    if constexpr (PS) return;
    float zs[GT_];
#pragma unroll
    for (int t = 0; t < GT_; ++t) zs[t] = F8 ? 0.0f : -ZPC * accs[t][0];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      uint4 d4 = make_uint4(0, 0, 0, 0), z4 = make_uint4(0, 0, 0, 0);
      if constexpr (GS != 0) d4 = sc_s[gbuf * SC_U4 + (wk * WN + wn) * 8 + (g & 1) * 4 + reg];
      if constexpr (ZP) z4 = zp_s[gbuf * SC_U4 + (wk * WN + wn) * 8 + (g & 1) * 4 + reg];
      const uint32_t d[4] = {d4.x, d4.y, d4.z, d4.w};
      const uint32_t zd[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float scv = 1.f;
        if constexpr (GS != 0) scv = T::to_float((uint16_t)(d[j] >> sc_shift));
        float nzc = 0.f;   // -(16 + z) of this column
        if constexpr (ZP) nzc = -(16.0f + T::to_float((uint16_t)(zd[j] >> sc_shift)));
#pragma unroll
        for (int t = 0; t < GT_; ++t) {
          const float dlt = ZP ? fmaf(nzc, accs[t][0], accg[j][t][reg]) : accg[j][t][reg] + zs[t];
          if constexpr (GS != 0) accm[j][t][reg] = fmaf(scv, dlt, accm[j][t][reg]);
          else accm[j][t][reg] += dlt;
        }
      }
    }
  };

  // ---- prologue ----
  uint4 w0[KSS * WV], w1[KSS * WV], w2[KSS * WV], w3[KSS * WV];
  // activations travel global -> registers -> LDS TWO stages ahead of their use (round 3: with one stage, a
  // 128-row stage is ~1000 cycles of work against ~2000 of L2 latency and every stage began by waiting for its
  // own activations -- PMC at M = 512: 45 % of the wave cycles in s_waitcnt, profiles/r03_prefill_gemm_pmc.txt)
  // Only the prompt-sized tiles that have the registers for a second set (64 / 128 rows, 4-bit, no zero points, at
  // most two k groups); the others keep one stage of distance (A2 false: ar1 is ar0's alias in the code below).
  constexpr bool A2 = MT >= 4 && WK <= 2 && BITS == 4 && !ZP && (MT < 8 || std::is_same<T, BF16>::value);   // fp16 at 128 rows: spills
  uint4 ar0[APT], ar1[APT];
  uint4 scr = make_uint4(0, 0, 0, 0);
  load_a(0, ar0);
  scr = load_sc(0);
  load_w(0, w0);
  load_w(1, w1);
  load_w(2, w2);
  if constexpr (A2) load_a(1, ar1);
  store_a(0, ar0);
  store_sc(0, scr);
  __syncthreads();
  const uint4 ones = make_uint4(W4<T>::ONES, W4<T>::ONES, W4<T>::ONES, W4<T>::ONES);

  // one stage (U = position in the ring: buffer parity and the scale-group schedule are static):
  // fetch the activations of stage st+2 and the weights of stage st+3 (into the slots stage st-1
  // just released), multiply stage st, park stage st+1's activations (fetched one stage ago), barrier
  auto stage = [&](auto u_tag, int st, const uint4 (&wc)[KSS * WV], uint4 (&wfree)[KSS * WV]) {
    constexpr int U = decltype(u_tag)::value;
    constexpr int buf = U & 1;
    constexpr bool closes = (U + 1) % SPG == 0;   // last stage of a scale group
    // scale buffer of this stage's group: static for 2 groups per ring, else by ring parity
    const int gbuf = SPG == 2 ? ((U >> 1) & 1) : ((st >> 2) & 1);
    if constexpr (!A2) load_a(st + 1, ar0);
    else if constexpr ((U & 1) == 0) load_a(st + 2, ar0);
    else load_a(st + 2, ar1);
    if constexpr (closes) scr = load_sc(st + 1);  // the next stage opens a group
    load_w(st + 3, wfree);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KSS; ++ks) {
      const int kstep = (U % SPG) * KSS + ks;     // k-step inside the group
      const bool first = kstep == 0;              // static after unrolling
      if constexpr (PS && GS != 0) {
        if (first) {  // static: the group's scales of columns 16 j + r: element (r&7)*8 + 2 j + (r>>3)
          const uint4 d4 = sc_s[gbuf * SC_U4 + (wk * WN + wn) * 8 + (r & 7)];
          const uint32_t d[4] = {d4.x, d4.y, d4.z, d4.w};
          uint4 z4 = make_uint4(0, 0, 0, 0);
          if constexpr (ZP) z4 = zp_s[gbuf * SC_U4 + (wk * WN + wn) * 8 + (r & 7)];
          const uint32_t zd[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            ps_s[j] = T::to_float((uint16_t)(d[j] >> ((r >> 3) * 16)));
            const float zc = ZP ? 16.0f + T::to_float((uint16_t)(zd[j] >> ((r >> 3) * 16))) : W4_ZP;
            ps_z[j] = -zc * ps_s[j];
          }
        }
      }
      uint4 af[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) af[t] = a_s[(buf * WK + wk) * A_U4 + (ks * 4 + g) * PSTR + ((t * 16 + r) ^ (ks << 2))];
      uint32_t own[4 * WV];
#pragma unroll
      for (int h = 0; h < WV; ++h) {
        own[4 * h] = wc[ks * WV + h].x; own[4 * h + 1] = wc[ks * WV + h].y;
        own[4 * h + 2] = wc[ks * WV + h].z; own[4 * h + 3] = wc[ks * WV + h].w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // partner lane r^8 holds the other k-tile of the same vector: one DPP move per k-tile,
        // row_ror:8 with a bank mask so that only the half-row that needs the partner's word takes
        // it (banks 2,3 = lanes 8..15 = blk 1 for the even k-tile, banks 0,1 for the odd one)
        uint4 wv;
        if constexpr (BITS == 4) {
          const uint32_t e = (uint32_t)__builtin_amdgcn_update_dpp((int)own[j], (int)own[j], 0x128, 0xf, 0xc, false);
          const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)own[j], (int)own[j], 0x128, 0xf, 0x3, false);
          wv = make_uint4(and_or(__builtin_amdgcn_alignbit(e, e, rot_lo), kmask, kmagic),
                          and_or(__builtin_amdgcn_alignbit(e, e, rot_hi), kmask, kmagic),
                          and_or(__builtin_amdgcn_alignbit(o, o, rot_lo), kmask, kmagic),
                          and_or(__builtin_amdgcn_alignbit(o, o, rot_hi), kmask, kmagic));
        } else {
          // words (tile j, block 0 / 1) of the lane's own k-tile: the even k-tile's word of MY block
          // is my own (block 0 lanes) or the partner's word 2j+1 (block 1 lanes), and vice versa
          const uint32_t e = (uint32_t)__builtin_amdgcn_update_dpp((int)own[2 * j], (int)own[2 * j + 1], 0x128, 0xf, 0xc, false);
          const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)own[2 * j + 1], (int)own[2 * j], 0x128, 0xf, 0x3, false);
          if constexpr (F8) {
            // bytes (0, 2) and (1, 3) are the k pairs: bring them together, then two exact fp8 -> f32 -> T conversions
            uint32_t p0, p1, p2, p3;
            fp8x4_to_pairs_t<T>(__builtin_amdgcn_perm(e, e, 0x03010200u), p0, p1);
            fp8x4_to_pairs_t<T>(__builtin_amdgcn_perm(o, o, 0x03010200u), p2, p3);
            wv = make_uint4(p0, p1, p2, p3);
          } else if constexpr (std::is_same<T, F16>::value) {
            // bytes {0,2} / {1,3} next to 0x64: fp16 1024 + b, one v_perm_b32 per pair
            wv = make_uint4(__builtin_amdgcn_perm(0x64646464u, e, 0x04020400u), __builtin_amdgcn_perm(0x64646464u, e, 0x04030401u),
                            __builtin_amdgcn_perm(0x64646464u, o, 0x04020400u), __builtin_amdgcn_perm(0x64646464u, o, 0x04030401u));
          } else {
            auto pair = [](uint32_t x, int lo_byte, int hi_byte) -> uint32_t {
              return BF16::pack2((float)((x >> (8 * lo_byte)) & 0xffu), (float)((x >> (8 * hi_byte)) & 0xffu));
            };
            wv = make_uint4(pair(e, 0, 2), pair(e, 1, 3), pair(o, 0, 2), pair(o, 1, 3));
          }
        }
        if constexpr (PS) {
          // w = (16 + q) * s - 24 * s in fp32, one rounding to the model dtype
          const uint32_t xs4[4] = {wv.x, wv.y, wv.z, wv.w};
          uint32_t ws4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            ws4[e] = T::pack2(fmaf(lo_f<T>(xs4[e]), ps_s[j], ps_z[j]), fmaf(hi_f<T>(xs4[e]), ps_s[j], ps_z[j]));
          const uint4 ws = make_uint4(ws4[0], ws4[1], ws4[2], ws4[3]);
#pragma unroll
          for (int t = 0; t < MT; ++t) accm[j][t] = W4<T>::mfma(ws, af[t], accm[j][t]);
        } else {
#pragma unroll
          for (int t = 0; t < GT_; ++t) accg[j][t] = W4<T>::mfma(wv, af[t], first ? zero4 : accg[j][t]);
        }
        // 64+-row tiles: at the register limit -- keep the scheduler from expanding all four column
        // tiles ahead of the MFMAs (that costs 16+ live VGPRs and spills)
        if constexpr (MT >= 4) __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (!PS) {
#pragma unroll
        for (int t = 0; t < GT_; ++t)
          if constexpr (!F8) accs[t] = W4<T>::mfma(ones, af[t], first ? zero4 : accs[t]);
        if (kstep == 3) flush(gbuf);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!A2 || (U & 1) == 1) store_a(buf ^ 1, ar0);
    else store_a(buf ^ 1, ar1);
    if constexpr (closes) store_sc(gbuf ^ 1, scr);
    __syncthreads();
  };
  for (int st = 0; st < n_stages; st += 4) {
    stage(std::integral_constant<int, 0>{}, st, w0, w3);
    stage(std::integral_constant<int, 1>{}, st + 1, w1, w0);
    stage(std::integral_constant<int, 2>{}, st + 2, w2, w1);
    stage(std::integral_constant<int, 3>{}, st + 3, w3, w2);
  }

  w4_tall_epilogue<T, MT, WN, WK, GS>(p, accm, lds, wn, wk, lane, r, g, chunk, chunk_ok, m0, split);
}

// ---------------------------------------------------------------------------------------------
// The MFMA-native weight layout and its kernel (M <= 64, 4-bit, group 128 / channelwise).
//
// rocprofv3 PMC and the ablations of the tall kernel put decode at ~70 VALU instructions per
// KiB of weights, issued by ~1.75 waves per SIMD: the wave program's own instruction issue, not HBM, sets the
// time.  Three parts of that count are owed to the Marlin tensor being an NVIDIA fragment order: the DPP pair
// exchange (8 per k-step), per-lane rotate amounts instead of plain shifts (16), and activations that must be
// re-ordered to the Marlin k order on their way through LDS (four ds_write_b32 per 16 bytes).  The op
// gptq_marlin_gemm has to take that tensor; the LinearMethod does not: at load time it also builds
//     native[kstep][chunk][lane] (uint4),  lane = r + 16 g:
//       dword j  <->  column n = 64 chunk + 16 j + r   (the MFMA A-operand row of tile j)
//       nibble (s >> 1) + 4 (s & 1) of that dword  <->  k = 32 kstep + 8 g + s,  s = 0..7
// i.e. exactly one MFMA operand per lane, k in NATURAL order (slot s of lane group g is k 8g + s: the B
// operand is the activation row's own 16 bytes), pairs (2p, 2p+1) 16 bits apart so that one v_and_or_b32 of
// x >> 4p makes two numbers, and a wave's k-step is 1 KiB contiguous; a k-step of all chunks is one
// contiguous band of N * 16 bytes, so the waves of a launch, which advance through k together, sweep memory
// front to back as they do on the Marlin tensor (the chunk-major order, every wave in a stream of its own
// 128 KiB apart, measured 10-15 % slower).  bf16 keeps the nibble where it is (128 + q, exponent 2^7: 3 shifts + 4 and_or per dword), fp16
// moves it up to the top mantissa bits (16 + q, as in the Marlin kernels) to spare its 10-bit sum of
// activations three more bits of cancellation.  Scales stay the checkpoint's natural [groups, N] tensor and are
// parked in LDS as fp32 in output-fragment order; zero point and group scale are applied in fp32 on the group
// accumulators exactly as in the tall kernel.  Everything around the operand path -- 4 waves = WN chunks x WK
// k groups, 4-slot register ring three stages ahead, one barrier per stage, LDS K reduction, slabs / ticket /
// deferred / silu epilogues -- is the tall kernel's.
// W4N<T> (mask / magic / nibble position / zero-point constant of the native tensor): w4a16_common.h

template <typename T, int MT, int WN, int WK, int GS, bool NT = false>
__global__ __launch_bounds__(GT, 2) void w4n_gemm_kernel(const GemmParams p) {
  static_assert(WN * WK == 4, "4 waves per workgroup");
  static_assert(MT == 1 || MT == 2 || MT == 4, "16, 32 or 64 rows");
  static_assert(GS == 0 || GS == 128, "group 128 or channelwise");
  using N4 = W4N<T>;
  constexpr int MP = 16 * MT;
  constexpr int KSS = MT >= 4 ? 1 : 2;         // MFMA k-steps per stage
  constexpr int TS_K = 32 * KSS;
  constexpr int SPG = 4 / KSS;                 // stages per 128-k scale group
  constexpr int PPR = 4 * KSS;                 // 16-byte activation pieces per row and stage
  constexpr int A_U4 = KSS * 4 * MP;           // uint4 per (stage, k-group): [k-step][g][row]
  constexpr int SC_ROW_U4 = 16;                // one chunk's 64 scales as fp32, [g][reg][j]
  constexpr int SC_U4 = 4 * SC_ROW_U4;
  constexpr int MAIN_U4 = 2 * WK * A_U4 + 2 * SC_U4;
  constexpr int RED_U4 = (WK > 1) ? (WK - 1) * WN * MT * 256 : 0;
  constexpr int LDS_U4 = MAIN_U4 > RED_U4 ? (MAIN_U4 > GT ? MAIN_U4 : GT) : (RED_U4 > GT ? RED_U4 : GT);
  __shared__ __attribute__((aligned(16))) uint4 lds[LDS_U4];
  uint4* a_s = lds;
  uint4* sc_s = lds + 2 * WK * A_U4;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave % WN, wk = wave / WN;
  const int r = lane & 15, g = lane >> 4;
  const int n_chunks = p.N >> 6;
  const int chunk = blockIdx.x * WN + wn;
  const bool chunk_ok = chunk < n_chunks;
  const int m0 = blockIdx.z * MP;
  const int split = blockIdx.y;
  const int k_wg0 = split * p.k_per_wg;
  const int k_wg1 = min(k_wg0 + p.k_per_wg, p.K);
  const int k_per_wave = (k_wg1 - k_wg0) / WK;      // multiple of 4 stages (make_plan)
  const int k_w0 = k_wg0 + wk * k_per_wave;
  const int n_stages = k_per_wave / TS_K;
  const int st_last = __builtin_amdgcn_readfirstlane(n_stages - 1);

  // ---- weights: one 16-byte load per lane and k-step, a wave reads 1 KiB contiguous ----
  const int64_t kstep_u4 = (int64_t)n_chunks * 64;     // one k-step of every chunk: N * 16 bytes, contiguous
  const uint4* bp = p.b + ((int64_t)(k_w0 >> 5) * n_chunks + (chunk_ok ? chunk : 0)) * 64 + lane;
  auto load_w = [&](int st, uint4 (&w)[KSS]) {
    const uint4* q = bp + (int64_t)min(st, st_last) * (KSS * kstep_u4);
#pragma unroll
    for (int ks = 0; ks < KSS; ++ks) w[ks] = ld_stream<NT>(q + ks * kstep_u4);
  };
  const uint32_t kmask = __builtin_amdgcn_readfirstlane(N4::MASK);
  uint32_t kmagic = N4::MAGIC;
  asm volatile("" : "+v"(kmagic));

  // ---- activations -> registers -> LDS: a 16-byte piece (row, 8 k) IS the B operand of lane (row, g) ----
  constexpr int A_CHUNKS = WK * MP * PPR;
  constexpr int APT = (A_CHUNKS + GT - 1) / GT;
  static_assert(A_CHUNKS % GT == 0 || A_CHUNKS < GT, "no ragged tail");
  const uint16_t* ap[APT];
  int a_dst[APT];
#pragma unroll
  for (int i = 0; i < APT; ++i) {
    const int id = (threadIdx.x + i * GT) % A_CHUNKS;
    const int c8 = id % PPR, row = (id / PPR) % MP, kg = (id / PPR) / MP;
    ap[i] = p.a + (int64_t)min(m0 + row, p.M - 1) * p.K + (k_wg0 + kg * k_per_wave) + c8 * 8;
    // c8 = 4 kstep + g.  The eight pieces of a row go to eight [kstep][g] planes MP * 16 bytes apart -- the same
    // LDS banks -- so the row index is XOR-swizzled with the plane: a store wave-instruction's groups of 8 lanes
    // (one row, c8 = 0..7) then cover 8 different bank quads, and a plane read back by 16 lanes (r = 0..15) is
    // still a permutation of 16 consecutive vectors
    a_dst[i] = kg * A_U4 + c8 * MP + (row ^ (c8 & 7));
  }
  auto load_a = [&](int st, uint4 (&av)[APT]) {
    const int off = min(st, st_last) * TS_K;
#pragma unroll
    for (int i = 0; i < APT; ++i) av[i] = ld16(ap[i] + off);
  };
  auto store_a = [&](int buf, const uint4 (&av)[APT]) {
#pragma unroll
    for (int i = 0; i < APT; ++i) a_s[buf * WK * A_U4 + a_dst[i]] = av[i];
  };
  // scale rows: threads 0..31 = (k-group*WN + wn') x 8 pieces of 16 B (64 columns x 2 B, natural order)
  const int s_combo = (threadIdx.x >> 3) & 3, s_piece = threadIdx.x & 7;
  const int s_chunk = min(blockIdx.x * WN + (s_combo % WN), n_chunks - 1);
  const uint16_t* s_src = p.s + (int64_t)s_chunk * 64 + s_piece * 8;
  const int s_k0 = k_wg0 + (s_combo / WN) * k_per_wave;
  auto load_sc = [&](int st) -> uint4 {
    if constexpr (GS == 0) return make_uint4(0, 0, 0, 0);
    const int k_abs = min(s_k0 + st * TS_K, p.K - 1);
    return ld16(s_src + (int64_t)(k_abs / 128) * p.N);
  };
  auto store_sc = [&](int gbuf, uint4 v) {
    if constexpr (GS != 0) {
      if (threadIdx.x < 32) {
        // piece pc holds columns 8 pc .. 8 pc + 7 = tile j = pc / 2, lane group g = 2 (pc & 1) + (i >> 2),
        // reg = i & 3: fp32 destination index (g * 4 + reg) * 4 + j
        const int slot = threadIdx.x >> 3, pc = threadIdx.x & 7;
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
        float* dst = reinterpret_cast<float*>(sc_s + gbuf * SC_U4 + slot * SC_ROW_U4);
        const int j = pc >> 1, g0 = 2 * (pc & 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float f = (i & 1) ? hi_f<T>(d[i >> 1]) : lo_f<T>(d[i >> 1]);
          dst[((g0 + (i >> 2)) * 4 + (i & 3)) * 4 + j] = f;
        }
      }
    }
  };

  f32x4_t accm[4][MT], accg[4][MT], accs[MT];
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < MT; ++t) { accm[j][t] = zero4; accg[j][t] = zero4; }
#pragma unroll
  for (int t = 0; t < MT; ++t) accs[t] = zero4;

  auto flush = [&](int gbuf) {
    float zs[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) zs[t] = -N4::ZPC * accs[t][0];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      f32x4_t d4 = {1.f, 1.f, 1.f, 1.f};
      if constexpr (GS != 0)
        d4 = reinterpret_cast<const f32x4_t*>(sc_s + gbuf * SC_U4 + (wk * WN + wn) * SC_ROW_U4)[g * 4 + reg];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const float dlt = accg[j][t][reg] + zs[t];
          if constexpr (GS != 0) accm[j][t][reg] = fmaf(d4[j], dlt, accm[j][t][reg]);
          else accm[j][t][reg] += dlt;
        }
    }
  };

  // ---- prologue ----
  uint4 w0[KSS], w1[KSS], w2[KSS], w3[KSS];
  uint4 ar[APT];
  uint4 scr = make_uint4(0, 0, 0, 0);
  load_a(0, ar);
  scr = load_sc(0);
  load_w(0, w0);
  load_w(1, w1);
  load_w(2, w2);
  store_a(0, ar);
  store_sc(0, scr);
  __syncthreads();
  const uint4 ones = make_uint4(N4::ONES, N4::ONES, N4::ONES, N4::ONES);

  auto stage = [&](auto u_tag, int st, const uint4 (&wc)[KSS], uint4 (&wfree)[KSS]) {
    constexpr int U = decltype(u_tag)::value;
    constexpr int buf = U & 1;
    constexpr bool closes = (U + 1) % SPG == 0;
    const int gbuf = SPG == 2 ? ((U >> 1) & 1) : ((st >> 2) & 1);
    load_a(st + 1, ar);
    if constexpr (closes) scr = load_sc(st + 1);
    load_w(st + 3, wfree);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KSS; ++ks) {
      const int kstep = (U % SPG) * KSS + ks;
      const bool first = kstep == 0;
      uint4 af[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t)
        af[t] = a_s[(buf * WK + wk) * A_U4 + (ks * 4 + g) * MP + ((t * 16 + r) ^ ((ks * 4 + g) & 7))];
      const uint32_t x4[4] = {wc[ks].x, wc[ks].y, wc[ks].z, wc[ks].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t x = x4[j];
        // pair p = nibbles p and p + 4: bring nibble p to bit POS of the low half (p + 4 follows in the high half)
        auto pair = [&](int pz) -> uint32_t {
          constexpr int dummy = 0; (void)dummy;
          const int sh = 4 * pz - N4::POS;
          const uint32_t y = sh == 0 ? x : (sh > 0 ? (x >> sh) : (x << (-sh)));
          return and_or(y, kmask, kmagic);
        };
        const uint4 wv = make_uint4(pair(0), pair(1), pair(2), pair(3));
#pragma unroll
        for (int t = 0; t < MT; ++t) accg[j][t] = W4<T>::mfma(wv, af[t], first ? zero4 : accg[j][t]);
        if constexpr (MT >= 4) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) accs[t] = W4<T>::mfma(ones, af[t], first ? zero4 : accs[t]);
      if (kstep == 3) flush(gbuf);
    }
    __builtin_amdgcn_sched_barrier(0);
    store_a(buf ^ 1, ar);
    if constexpr (closes) store_sc(gbuf ^ 1, scr);
    __syncthreads();
  };
  for (int st = 0; st < n_stages; st += 4) {
    stage(std::integral_constant<int, 0>{}, st, w0, w3);
    stage(std::integral_constant<int, 1>{}, st + 1, w1, w0);
    stage(std::integral_constant<int, 2>{}, st + 2, w2, w1);
    stage(std::integral_constant<int, 3>{}, st + 3, w3, w2);
  }
  w4_tall_epilogue<T, MT, WN, WK, GS, true>(p, accm, lds, wn, wk, lane, r, g, chunk, chunk_ok, m0, split);
}

// GPTQ int32 [K/8, N] (8 consecutive k of a column per word, low nibble first; optional act-order row gather
// `perm`) -> the native tensor above.  One thread per output dword.
__global__ void w4_native_repack_kernel(const uint32_t* __restrict__ qw, const int* __restrict__ perm,
                                        uint32_t* __restrict__ out, int K, int N) {
  const int64_t total = (int64_t)(K / 8) * N;       // dwords
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int j = idx & 3;
  const int lane = (idx >> 2) & 63;
  const int64_t blk = idx >> 8;                      // kstep * (N/64) + chunk
  const int chunks = N >> 6;
  const int ks = blk / chunks, chunk = blk % chunks;
  const int r = lane & 15, g = lane >> 4;
  const int n = chunk * 64 + 16 * j + r;
  const int k0 = ks * 32 + 8 * g;
  uint32_t res = 0;
  if (perm == nullptr) {
    const uint32_t w = qw[(int64_t)(k0 >> 3) * N + n];
#pragma unroll
    for (int sidx = 0; sidx < 8; ++sidx) res |= ((w >> (4 * sidx)) & 0xfu) << (4 * ((sidx >> 1) + 4 * (sidx & 1)));
  } else {
#pragma unroll
    for (int sidx = 0; sidx < 8; ++sidx) {
      const int ksrc = perm[k0 + sidx];
      const uint32_t code = (qw[(int64_t)(ksrc >> 3) * N + n] >> (4 * (ksrc & 7))) & 0xfu;
      res |= code << (4 * ((sidx >> 1) + 4 * (sidx & 1)));
    }
  }
  out[idx] = res;
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
// AWQ checkpoint layout (int32 [K, N/8]: 8 columns of one k per word, nibble of column c =
// {0,4,1,5,2,6,3,7}[c], dequantize.cuh:31-62) -> Marlin tensor, the same target as above.
__global__ void awq_marlin_repack_kernel(const uint32_t* __restrict__ qw, uint32_t* __restrict__ out,
                                         int K, int N) {
  const int64_t row_words = (int64_t)N * 2;
  const int64_t total = (int64_t)(K / 16) * row_words;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int kt = idx / row_words;
  const int col = idx % row_words;
  const int chunk = col / 128, rr = col % 128;
  const int i = rr >> 2, j = rr & 3;
  const int q = i & 3, n_in = i >> 2;
  uint32_t res = 0;
#pragma unroll
  for (int pz = 0; pz < 8; ++pz) {
    const int k = kt * 16 + 2 * q + ((pz & 1) ? 8 : 0) + (pz >> 2);
    const int n = chunk * 64 + j * 16 + ((pz >> 1) & 1) * 8 + n_in;
    const int c = n & 7;
    const uint32_t w = qw[(int64_t)k * (N / 8) + (n >> 3)];
    res |= ((w >> (4 * (((c & 1) << 2) | (c >> 1)))) & 0xfu) << (4 * pz);
  }
  out[idx] = res;
}

struct GemmPlan {
  int mt, wn, wm, wk;  // kernel shape
  int splits, k_per_wg, m_blocks, n_blocks;
  int tall;      // 1: tall register tile (64 columns x 16 mt rows per wave); wn, wk, mt say which
};

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// Pick the workgroup shape and the split-K factor.  Decode-sized GEMMs (M <= 64) last only a
// few microseconds at HBM speed, so the plan aims at >= ~2 workgroups per CU while keeping the
// fp32 partial traffic (splits * M * N * 4 B) well below the weight bytes (K * N / 2).
static GemmPlan make_plan(int M, int N, int K, int64_t tickets_len, bool allow_tall = false,
                          int bits = 4, bool unsplit = false, bool has_zp = false, bool native = false) {
  GemmPlan pl;
  const int n_chunks = N / 64;
  pl.tall = 0;
  // K in whole 256-k rings per wave, group 128 / channelwise, no act-order: the tall register
  // tile (measured faster than the kernels below at every M on the Llama-3-8B shapes)
  if (allow_tall && M >= env_int("NMV_W4_TALL_MIN_M", 1) && K % 256 == 0 && env_int("NMV_W4_TALL", 1)) {
    pl.tall = 1;
    pl.wm = 1;
    // 128-row prescale tile: measured to win from M = 1024 on every Llama-3-8B projection and from
    // M = 384 on the wide ones (gate_up: 153 vs 176 us at M = 512, 481 vs 666 us at M = 2048)
    // (round 2, tools/sweep_tall.py at M = 128 / 256 / 512): the prescale tile already wins at M = 256 on the wide
    // projections (gate_up 88.9 vs 98.6 us); the narrow ones (qkv, o: < 256 chunks) are faster with the 32-row
    // tile and four k groups up to M = 256 (o 14.2 vs 17.6 us at M = 128, 20.5 vs 23.6 at 256), a long K (down)
    // only up to M = 128 (32.8 vs 39.3 us)
    const bool big = M >= 1024 || (M >= 256 && n_chunks >= 256);
    const bool narrow32 = n_chunks < 256 && (M <= 128 || (M <= 256 && K <= 8192));
    pl.mt = env_int("NMV_W4_TALL_MT", M <= 16 ? 1 : (M <= 64 || narrow32) ? 2 : big ? 8 : 4);
    if (bits == 8 && pl.mt > 4) pl.mt = 4;       // no prescale variant for 8-bit codes  // measured: 64-row tile wins only past M = 64
    const int rows = 16 * pl.mt;
    pl.m_blocks = (M + rows - 1) / rows;
    // wide N: two chunks per workgroup, two k groups; narrow N: one chunk, four k groups (the
    // in-workgroup k reduction goes through LDS and saves split-K slabs)
    int wk = (M > 32 && n_chunks * pl.m_blocks >= 256 && !(M > 64 && pl.mt == 2)) ? 2 : 4;
    wk = env_int("NMV_W4_TALL_WK", wk);
    if (pl.mt == 8) wk = 1;                     // the 128-row tile exists as 4 chunks x 1 k group only
    if (bits == 8 && pl.mt == 4 && wk == 4) wk = 2;  // 8-bit 64-row tile: registers allow 4x1 / 2x2
    const int ring_k = pl.mt >= 4 ? 128 : 256;  // 4 stages of 32 / 64 k
    while (wk > 1 && K % (ring_k * wk) != 0) wk >>= 1;
    pl.wk = wk;
    pl.wn = 4 / wk;
    pl.n_blocks = (n_chunks + pl.wn - 1) / pl.wn;
    const int unit = ring_k * wk;
    const int k_units = K / unit;
    const int base_wgs = pl.n_blocks * pl.m_blocks;
    int splits = std::max(1, env_int("NMV_W4_TALL_WGS", 512) / base_wgs);
    splits = std::min(splits, env_int("NMV_W4_TALL_MAX_SPLITS", 8));
    splits = env_int("NMV_W4_SPLITS", splits);
    splits = std::max(1, std::min(splits, k_units));
    if ((int64_t)base_wgs > tickets_len || unsplit) splits = 1;
    pl.k_per_wg = ((k_units + splits - 1) / splits) * unit;
    pl.splits = (K + pl.k_per_wg - 1) / pl.k_per_wg;
    return pl;
  }
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
  if constexpr (GS == 0 || GS == 128) {
    if (pl.tall && p.bits == 8) {
#define NMV_W8_TALL_CASE(mt_, wn_, wk_)                                                                  \
  if (pl.mt == mt_ && pl.wn == wn_ && pl.wk == wk_) {                                                    \
    if constexpr (GS == 0) {                                                                             \
      if (p.fp8) {                                                                                       \
        hipLaunchKernelGGL((w4a16_gemm_tall_kernel<T, mt_, wn_, wk_, 0, false, 8, false, false, true>), grid, block, 0, s, p); \
        return 0;                                                                                        \
      }                                                                                                  \
    }                                                                                                    \
    if (p.fp8) return -1;                                                                                \
    hipLaunchKernelGGL((w4a16_gemm_tall_kernel<T, mt_, wn_, wk_, GS, false, 8>), grid, block, 0, s, p);  \
    return 0;                                                                                            \
  }
      NMV_W8_TALL_CASE(1, 4, 1) NMV_W8_TALL_CASE(1, 2, 2) NMV_W8_TALL_CASE(1, 1, 4)
      NMV_W8_TALL_CASE(2, 4, 1) NMV_W8_TALL_CASE(2, 2, 2) NMV_W8_TALL_CASE(2, 1, 4)
      NMV_W8_TALL_CASE(4, 4, 1) NMV_W8_TALL_CASE(4, 2, 2)
#undef NMV_W8_TALL_CASE
      return -1;
    }
    if constexpr (GS == 128) {
      if (pl.tall && p.zp != nullptr) {
#define NMV_ZP_TALL_CASE(mt_, wn_, wk_, ps_)                                                                \
  if (pl.mt == mt_ && pl.wn == wn_ && pl.wk == wk_) {                                                       \
    hipLaunchKernelGGL((w4a16_gemm_tall_kernel<T, mt_, wn_, wk_, 128, ps_, 4, true>), grid, block, 0, s, p); \
    return 0;                                                                                               \
  }
        NMV_ZP_TALL_CASE(1, 4, 1, false) NMV_ZP_TALL_CASE(1, 2, 2, false) NMV_ZP_TALL_CASE(1, 1, 4, false)
        NMV_ZP_TALL_CASE(2, 4, 1, false) NMV_ZP_TALL_CASE(2, 2, 2, false) NMV_ZP_TALL_CASE(2, 1, 4, false)
        NMV_ZP_TALL_CASE(4, 4, 1, false) NMV_ZP_TALL_CASE(4, 2, 2, false) NMV_ZP_TALL_CASE(4, 1, 4, false)
        NMV_ZP_TALL_CASE(8, 4, 1, true)
#undef NMV_ZP_TALL_CASE
        return -1;
      }
    }
    if (p.native) {
      if (!pl.tall || p.bits != 4 || p.zp != nullptr) return -1;
#define NMV_W4N_CASE(mt_, wn_, wk_)                                                        \
  if (pl.mt == mt_ && pl.wn == wn_ && pl.wk == wk_) {                                      \
    hipLaunchKernelGGL((w4n_gemm_kernel<T, mt_, wn_, wk_, GS>), grid, block, 0, s, p);     \
    return 0;                                                                              \
  }
      NMV_W4N_CASE(1, 4, 1) NMV_W4N_CASE(1, 2, 2) NMV_W4N_CASE(1, 1, 4)
      NMV_W4N_CASE(2, 4, 1) NMV_W4N_CASE(2, 2, 2) NMV_W4N_CASE(2, 1, 4)
      NMV_W4N_CASE(4, 4, 1) NMV_W4N_CASE(4, 2, 2) NMV_W4N_CASE(4, 1, 4)
#undef NMV_W4N_CASE
      return -1;
    }
    if (pl.tall) {
      // weights read once by the launch (a single block of rows) stream non-temporally; the 64-row tile is
      // only planned for M > 64, i.e. several row blocks
      const bool nt = pl.m_blocks == 1 && pl.mt <= 2 && env_int("NMV_W4_NT", 1);
#define NMV_W4_TALL_CASE(mt_, wn_, wk_)                                                        \
  if (pl.mt == mt_ && pl.wn == wn_ && pl.wk == wk_) {                                          \
    if (nt) hipLaunchKernelGGL((w4a16_gemm_tall_kernel<T, mt_, wn_, wk_, GS, false, 4, false, true>), grid, block, 0, s, p); \
    else hipLaunchKernelGGL((w4a16_gemm_tall_kernel<T, mt_, wn_, wk_, GS>), grid, block, 0, s, p);  \
    return 0;                                                                                  \
  }
      NMV_W4_TALL_CASE(1, 4, 1) NMV_W4_TALL_CASE(1, 2, 2) NMV_W4_TALL_CASE(1, 1, 4)
      NMV_W4_TALL_CASE(2, 4, 1) NMV_W4_TALL_CASE(2, 2, 2) NMV_W4_TALL_CASE(2, 1, 4)
      NMV_W4_TALL_CASE(4, 4, 1) NMV_W4_TALL_CASE(4, 2, 2) NMV_W4_TALL_CASE(4, 1, 4)
#undef NMV_W4_TALL_CASE
      if (pl.mt == 8 && pl.wn == 4 && pl.wk == 1) {
        hipLaunchKernelGGL((w4a16_gemm_tall_kernel<T, 8, 4, 1, GS, true>), grid, block, 0, s, p);
        return 0;
      }

      return -1;
    }
  }
  if (pl.tall) return -1;
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
int64_t wq_marlin_fallback_scratch_bytes(int size_m, int size_n, int size_k);
int wq_marlin_fallback(void* c, const void* a, const int32_t* b_q_weight, const void* b_scales,
                       const int32_t* g_idx, const int32_t* perm, int num_bits, int size_m,
                       int size_n, int size_k, int num_groups, int is_fp8, nmv_dtype_t dtype,
                       void* scratch, int64_t scratch_bytes, hipStream_t stream);

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

extern "C" int64_t nmv_gptq_marlin_gemm_scratch_bytes(int size_m, int size_n, int size_k, int variant) {
  const bool has_act_order = (variant & 1) != 0;
  // the A gather is fused into the LDS staging: no a_tmp copy, only the split-K slabs
  if (size_m <= 0 || size_n <= 0 || size_k <= 0) return 0;
  // upper bound over every plan the entry point may pick (the ticket array only lowers splits)
  int max_splits = make_plan(size_m, size_n, size_k, INT64_MAX, false).splits;
  if (!has_act_order) {
    max_splits = std::max(max_splits, make_plan(size_m, size_n, size_k, INT64_MAX, true, 4).splits);
    max_splits = std::max(max_splits, make_plan(size_m, size_n, size_k, INT64_MAX, true, 8).splits);
  }
  // the resident / streamed-activation kernel (w4a16_stream.hip) takes group-128 4-bit launches of up to 64 rows
  if (!has_act_order) {
    W4StreamPlan sp;
    if (w4s_make_plan(size_m, size_n, size_k, INT64_MAX, false, false, &sp)) max_splits = std::max(max_splits, sp.splits);
    if (w4s_make_plan(size_m, size_n, size_k, INT64_MAX, false, true, &sp)) max_splits = std::max(max_splits, sp.splits);
    W4RingPlan rp;   // the native tensor's calls size their scratch with this function too
    if (w4r_make_plan(size_m, size_n, size_k, INT64_MAX, false, &rp)) max_splits = std::max(max_splits, rp.splits);
    W4PrefillPlan pp;
    if (w4p_make_plan(size_m, size_n, size_k, INT64_MAX, false, &pp)) max_splits = std::max(max_splits, pp.splits);
  }
  // ... and of the generic kernel that takes the remaining variants (8-bit groups of 32 / 64, act-order on a K
  // shard): its gathered activations + slabs
  const int64_t tuned = max_splits > 1 ? (int64_t)max_splits * size_m * size_n * 4 : (int64_t)0;
  return variant ? std::max(tuned, wq_marlin_fallback_scratch_bytes(size_m, size_n, size_k)) : tuned;
}

constexpr int NMV_NOT_TALL = -1000;   // marlin_gemm_impl(fp8 = 1): shape outside the tall kernel (no error recorded)
static int marlin_gemm_impl(void* c, const void* a, const int32_t* b_q_weight, const void* b_scales,
                            const void* b_zeros, const int32_t* g_idx, const int32_t* perm,
                            int32_t* workspace, int64_t workspace_len, void* scratch,
                            int64_t scratch_bytes, int num_bits, int size_m, int size_n, int size_k,
                            int num_groups, int is_k_full, nmv_dtype_t dtype, void* stream,
                            int epi = 0, int native = 0, int fp8 = 0, int slab16 = 0) {
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
  // 8-bit codes run the tall kernel when its conditions hold (group 128 / channelwise, no act-order,
  // K in whole rings, at most 128 rows per launch block: M-tiles up to 64 rows exist for 8 bits)
  const bool tall8 = num_bits == 8 && !has_act_order && (group_size == 0 || (group_size == 128 && !fp8)) &&
                     size_k % 256 == 0 && env_int("NMV_W4_TALL", 1);
  if (fp8 && !tall8) return NMV_NOT_TALL;   // fp8_marlin_gemm outside the tall kernel's domain: the caller's generic kernel
  if ((num_bits == 8 && !tall8) || (has_act_order && !is_k_full)) {
    // the remaining 8-bit cases, and act-order on a K shard (irregular group runs: one scale row per k
    // through g_idx), take the generic LDS-staged kernel; same math, not HBM-tuned
    const int rc = wq_marlin_fallback(c, a, b_q_weight, b_scales, has_act_order ? g_idx : nullptr,
                                      has_act_order ? perm : nullptr, num_bits, size_m, size_n,
                                      size_k, num_groups, 0, dtype, scratch, scratch_bytes, (hipStream_t)stream);
    NMV_CHECK(rc != -2, "gptq_marlin_gemm: scratch too small (see nmv_gptq_marlin_gemm_scratch_bytes)");
    NMV_CHECK(rc == 0, "gptq_marlin_gemm: generic path launch failed");
    NMV_LAUNCH_CHECK();
    return NMV_OK;
  }
  // prompt-sized calls on the native tensor: 256 x 128 tiles with the codes expanded once per workgroup (w4a16_prefill.hip)
  if (native && num_bits == 4 && !has_act_order && group_size == 128 && b_zeros == nullptr) {
    W4PrefillPlan pp;
    if (w4p_make_plan(size_m, size_n, size_k, epi == 2 ? INT64_MAX : (workspace ? workspace_len : 0), epi == 1, &pp) &&
        (epi != 1 || (pp.splits == 1 && size_n % 128 == 0))) {
      const int64_t need = (pp.splits > 1 || epi == 2) ? (int64_t)pp.splits * size_m * size_n * (slab16 ? 2 : 4) : 0;
      NMV_CHECK(need < (int64_t)1 << 31, "w4_native_gemm: split-K slab too large");
      NMV_CHECK(scratch_bytes >= need && (need == 0 || scratch != nullptr),
                "w4_native_gemm: scratch too small (%lld < %lld)", (long long)scratch_bytes, (long long)need);
      GemmParams p;
      p.a = (const uint16_t*)a;
      p.b = (const uint4*)b_q_weight;
      p.s = (const uint16_t*)b_scales;
      p.zp = nullptr;
      p.perm = nullptr;
      p.c = (uint16_t*)c;
      p.slab = (float*)scratch;
      p.tickets = workspace;
      p.M = size_m; p.N = size_n; p.K = size_k;
      p.bits = 4;
      p.group_size = 128;
      p.k_per_wg = pp.k_per_wg;
      p.splits = pp.splits;
      p.native = 1;
      p.fp8 = 0;
      p.epi = epi;
      p.g_stage = 0;
      p.n_stages = 0;
      p.slab16 = slab16;
      const int rc = w4p_launch(pp, p, dtype == NMV_F16, (hipStream_t)stream);
      NMV_CHECK(rc == 0, "w4_native_gemm: prefill kernel launch failed for splits=%d (rc %d)", pp.splits, rc);
      NMV_LAUNCH_CHECK();
      return NMV_OK;
    }
  }
  NMV_CHECK(!slab16, "w4_native_gemm: 16-bit slabs (mode 3) exist for prompt-sized calls of the prefill kernels only "
                     "(nmv_w4_native_gemm_slab16)");
  // 17..64 rows on the native tensor: loader / consumer waves over an LDS-DMA ring (w4a16_ring.hip)
  if (native && num_bits == 4 && !has_act_order && group_size == 128 && b_zeros == nullptr) {
    W4RingPlan rp;
    if (w4r_make_plan(size_m, size_n, size_k, epi == 2 ? INT64_MAX : (workspace ? workspace_len : 0), epi == 1, &rp) &&
        (epi != 1 || (rp.splits == 1 && size_n % 128 == 0))) {
      const int64_t need = (rp.splits > 1 || epi == 2) ? (int64_t)rp.splits * size_m * size_n * 4 : 0;
      NMV_CHECK(need < (int64_t)1 << 31, "w4_native_gemm: split-K slab too large");
      NMV_CHECK(scratch_bytes >= need && (need == 0 || scratch != nullptr),
                "w4_native_gemm: scratch too small (%lld < %lld)", (long long)scratch_bytes, (long long)need);
      GemmParams p;
      p.a = (const uint16_t*)a;
      p.b = (const uint4*)b_q_weight;
      p.s = (const uint16_t*)b_scales;
      p.zp = nullptr;
      p.perm = nullptr;
      p.c = (uint16_t*)c;
      p.slab = (float*)scratch;
      p.tickets = workspace;
      p.M = size_m; p.N = size_n; p.K = size_k;
      p.bits = 4;
      p.group_size = 128;
      p.k_per_wg = rp.k_per_wg;
      p.splits = rp.splits;
      p.native = 1;
      p.fp8 = 0;
      p.epi = epi;
      p.g_stage = 0;
      p.n_stages = 0;
      p.slab16 = 0;
      const int rc = w4r_launch(rp, p, dtype == NMV_F16, (hipStream_t)stream);
      NMV_CHECK(rc == 0, "w4_native_gemm: ring kernel launch failed for mt=%d splits=%d (rc %d)", rp.mt, rp.splits, rc);
      NMV_LAUNCH_CHECK();
      return NMV_OK;
    }
  }
  // decode batches of the prevalent format (4-bit symmetric, group 128, no act-order): w4a16_stream.hip
  if (num_bits == 4 && !has_act_order && group_size == 128 && b_zeros == nullptr) {
    W4StreamPlan sp;
    if (w4s_make_plan(size_m, size_n, size_k, epi == 2 ? INT64_MAX : (workspace ? workspace_len : 0), epi == 1, epi == 2, &sp) &&
        (epi != 1 || (sp.splits == 1 && size_n % 128 == 0))) {
      const int64_t need = (sp.splits > 1 || epi == 2) ? (int64_t)sp.splits * size_m * size_n * 4 : 0;
      NMV_CHECK(need < (int64_t)1 << 31, "gptq_marlin_gemm: split-K slab too large");
      NMV_CHECK(scratch_bytes >= need && (need == 0 || scratch != nullptr),
                "gptq_marlin_gemm: scratch too small (%lld < %lld)", (long long)scratch_bytes, (long long)need);
      GemmParams p;
      p.a = (const uint16_t*)a;
      p.b = (const uint4*)b_q_weight;
      p.s = (const uint16_t*)b_scales;
      p.zp = nullptr;
      p.perm = nullptr;
      p.c = (uint16_t*)c;
      p.slab = (float*)scratch;
      p.tickets = workspace;
      p.M = size_m; p.N = size_n; p.K = size_k;
      p.bits = 4;
      p.group_size = 128;
      p.k_per_wg = sp.k_per_wg;
      p.splits = sp.splits;
      p.native = native;
      p.fp8 = 0;
      p.epi = epi;
      p.g_stage = sp.g_stage;
      p.n_stages = sp.n_stages;
      p.slab16 = 0;
      const int rc = w4s_launch(sp, p, dtype == NMV_F16, (hipStream_t)stream);
      NMV_CHECK(rc == 0, "gptq_marlin_gemm: no stream kernel for plan mt=%d nw=%d cpw=%d d=%d gst=%d (rc %d)", sp.mt,
                sp.nw, sp.cpw, sp.d, sp.gst, rc);
      NMV_LAUNCH_CHECK();
      return NMV_OK;
    }
    NMV_CHECK(!env_int("NMV_W4S_STRICT", 0), "gptq_marlin_gemm: NMV_W4S_STRICT is set and the stream kernel has no plan for this launch");
  }
  // the tall kernel covers the prevalent formats (group 128 / channelwise, no act-order gather);
  // groups of 32 / 64 and act-order stay on the 16-row kernel
  const bool allow_tall = !has_act_order && (group_size == 0 || group_size == 128);
  const GemmPlan pl = make_plan(size_m, size_n, size_k,
                                epi == 2 ? INT64_MAX : (workspace ? workspace_len : 0), allow_tall,
                                num_bits, epi == 1, b_zeros != nullptr, native != 0);
  NMV_CHECK(!native || (pl.tall && pl.mt <= 4 && num_bits == 4 && b_zeros == nullptr && !has_act_order),
            "w4_native_gemm: needs 4-bit symmetric codes, group 128 or channelwise, K %% 256 == 0, M tiles <= 64 rows");
  NMV_CHECK(epi != 2 || (pl.tall && num_bits == 4 && b_zeros == nullptr),
            "gptq_marlin_gemm_partial: needs 4-bit symmetric codes, group 128 or channelwise, no "
            "act-order, K %% 256 == 0");
  NMV_CHECK(epi != 1 || (pl.tall && pl.splits == 1 && num_bits == 4 && b_zeros == nullptr && size_n % 128 == 0),
            "gptq_marlin_gemm_silu_mul: needs 4-bit symmetric codes, group 128 or channelwise, no "
            "act-order, K %% 256 == 0, N %% 128 == 0");
  const int64_t need = (pl.splits > 1 || epi == 2) ? (int64_t)pl.splits * size_m * size_n * 4 : 0;
  NMV_CHECK(need < (int64_t)1 << 31, "gptq_marlin_gemm: split-K slab too large");
  NMV_CHECK(scratch_bytes >= need && (need == 0 || scratch != nullptr),
            "gptq_marlin_gemm: scratch too small (%lld < %lld)", (long long)scratch_bytes,
            (long long)need);
  GemmParams p;
  p.a = (const uint16_t*)a;
  p.b = (const uint4*)b_q_weight;
  p.s = (const uint16_t*)b_scales;
  p.perm = has_act_order ? perm : nullptr;
  p.zp = (const uint16_t*)b_zeros;
  NMV_CHECK(b_zeros == nullptr || (pl.tall && num_bits == 4 && group_size == 128),
            "marlin_zp_gemm: zero points need 4-bit codes, group 128, K %% 256 == 0, no act-order");
  p.c = (uint16_t*)c;
  p.slab = (float*)scratch;
  p.tickets = workspace;
  p.M = size_m; p.N = size_n; p.K = size_k;
  p.bits = num_bits;
  p.group_size = group_size;
  p.k_per_wg = pl.k_per_wg;
  p.splits = pl.splits;
  p.epi = epi;
  p.native = native;
  p.fp8 = fp8;
  p.slab16 = 0;
  hipStream_t s = (hipStream_t)stream;
  const int rc = dtype == NMV_F16 ? launch_gemm<F16>(pl, p, s) : launch_gemm<BF16>(pl, p, s);
  NMV_CHECK(rc == 0, "gptq_marlin_gemm: no kernel for plan mt=%d wn=%d wm=%d wk=%d", pl.mt, pl.wn,
            pl.wm, pl.wk);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

// fp8_marlin_gemm on the tall 8-bit kernel (channelwise scales, K % 256 == 0); NMV_NOT_TALL (-1000) when the shape is
// outside its domain -- wq_generic.hip then takes the generic kernel
int marlin_tall_fp8(void* c, const void* a, const int32_t* b_q_weight, const void* b_scales, int32_t* workspace,
                    int64_t workspace_len, void* scratch, int64_t scratch_bytes, int size_m, int size_n, int size_k,
                    int num_groups, nmv_dtype_t dtype, void* stream) {
  return marlin_gemm_impl(c, a, b_q_weight, b_scales, nullptr, nullptr, nullptr, workspace, workspace_len, scratch,
                          scratch_bytes, 8, size_m, size_n, size_k, num_groups, 1, dtype, stream, 0, 0, 1);
}

/* gate_up projection with silu_and_mul folded into the epilogue (not an op of nm-vllm 0.5.1).
 * b_q_weight / b_scales: the Marlin tensors of a weight whose OUTPUT COLUMNS were interleaved before
 * the repack -- 64-column chunk c = [gate 32c..32c+31 | up 32c..32c+31] -- c: [size_m, size_n / 2].
 * The roundings of gptq_marlin_gemm on the original weight followed by silu_and_mul: bit-identical when that GEMM does
 * not split K across workgroups, otherwise equal up to the order of the fp32 partial sums. */
extern "C" int nmv_gptq_marlin_gemm_silu_mul(void* c, const void* a, const int32_t* b_q_weight,
                                             const void* b_scales, int32_t* workspace,
                                             int64_t workspace_len, int size_m, int size_n,
                                             int size_k, int num_groups, nmv_dtype_t dtype,
                                             void* stream) {
  return marlin_gemm_impl(c, a, b_q_weight, b_scales, nullptr, nullptr, nullptr, workspace,
                          workspace_len, nullptr, 0, 4, size_m, size_n, size_k, num_groups, 1, dtype,
                          stream, 1);
}

/* Deferred split-K (not an op of nm-vllm 0.5.1): the GEMM stores its fp32 partial tiles in
 * slab[splits, size_m, size_n] and returns; the consumer (nmv_fused_add_rms_norm_partial) sums them
 * in split order -- the same association and rounding as the last-arriver pass of
 * nmv_gptq_marlin_gemm, which this removes from the critical path (ticket round trip + slab
 * re-read: 2-3 us per call at decode sizes).  4-bit symmetric codes, group 128 or channelwise, no
 * act-order, K % 256 == 0.  nmv_gptq_marlin_gemm_partial_splits returns the slab count (>= 1) the
 * call will write for a shape, or 0 when the shape is not supported. */
extern "C" int nmv_gptq_marlin_gemm_partial_splits(int size_m, int size_n, int size_k, int num_groups) {
  if (size_m <= 0 || size_n <= 0 || size_k <= 0 || size_n % 64 != 0 || size_k % 256 != 0) return 0;
  // the plan (and with it the slab count) depends on the grouping: group 128 takes w4a16_stream.hip
  W4StreamPlan sp;
  if (num_groups > 1 && size_k / num_groups == 128 && w4s_make_plan(size_m, size_n, size_k, INT64_MAX, false, true, &sp))
    return sp.splits;
  const GemmPlan pl = make_plan(size_m, size_n, size_k, INT64_MAX, true, 4, false);
  return pl.tall ? pl.splits : 0;
}

extern "C" int nmv_gptq_marlin_gemm_partial(float* slab, int64_t slab_bytes, const void* a,
                                            const int32_t* b_q_weight, const void* b_scales,
                                            int size_m, int size_n, int size_k, int num_groups,
                                            nmv_dtype_t dtype, void* stream) {
  return marlin_gemm_impl(nullptr, a, b_q_weight, b_scales, nullptr, nullptr, nullptr, nullptr, 0, slab,
                          slab_bytes, 4, size_m, size_n, size_k, num_groups, 1, dtype, stream, 2);
}

/* legacy Marlin checkpoints (csrc/quantization/marlin/dense/marlin_cuda_kernel.cu:1045-1136):
 * 4-bit, group -1 / 128, no act-order -- the same tile and scale layout as gptq_marlin */
extern "C" int nmv_gptq_marlin_gemm(void* c, const void* a, const int32_t* b_q_weight,
                                    const void* b_scales, const int32_t* g_idx,
                                    const int32_t* perm, int32_t* workspace,
                                    int64_t workspace_len, void* scratch, int64_t scratch_bytes,
                                    int num_bits, int size_m, int size_n, int size_k,
                                    int num_groups, int is_k_full, nmv_dtype_t dtype,
                                    void* stream) {
  return marlin_gemm_impl(c, a, b_q_weight, b_scales, nullptr, g_idx, perm, workspace, workspace_len,
                          scratch, scratch_bytes, num_bits, size_m, size_n, size_k, num_groups,
                          is_k_full, dtype, stream);
}

extern "C" int nmv_marlin_zp_gemm(void* c, const void* a, const int32_t* b_q_weight,
                                  const void* b_scales, const void* b_zeros, int32_t* workspace,
                                  int64_t workspace_len, void* scratch, int64_t scratch_bytes,
                                  int size_m, int size_n, int size_k, int num_groups,
                                  nmv_dtype_t dtype, void* stream) {
  NMV_CHECK(b_zeros != nullptr, "marlin_zp_gemm: b_zeros is required");
  return marlin_gemm_impl(c, a, b_q_weight, b_scales, b_zeros, nullptr, nullptr, workspace,
                          workspace_len, scratch, scratch_bytes, 4, size_m, size_n, size_k,
                          num_groups, 1, dtype, stream);
}

extern "C" int nmv_awq_marlin_repack(int32_t* out, const int32_t* qweight, int size_k, int size_n,
                                     void* stream) {
  NMV_CHECK(size_k % 16 == 0 && size_n % 64 == 0, "awq_marlin_repack: K %% 16 and N %% 64 required");
  const int64_t total = (int64_t)(size_k / 16) * size_n * 2;
  if (total == 0) return NMV_OK;
  hipLaunchKernelGGL(awq_marlin_repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const uint32_t*)qweight, (uint32_t*)out, size_k, size_n);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_marlin_gemm(void* c, const void* a, const int32_t* b_q_weight,
                               const void* b_scales, int32_t* workspace, int64_t workspace_len,
                               void* scratch, int64_t scratch_bytes, int size_m, int size_n,
                               int size_k, int num_groups, nmv_dtype_t dtype, void* stream) {
  NMV_CHECK(size_k % 128 == 0, "size_k = %d is not divisible by min_thread_k = 128", size_k);
  NMV_CHECK(num_groups >= 1 && size_k % num_groups == 0, "marlin_gemm: bad number of scale groups");
  const int gs = num_groups > 1 ? size_k / num_groups : -1;
  NMV_CHECK(gs == -1 || gs == 128, "Unexpected groupsize = %d", gs);
  return marlin_gemm_impl(c, a, b_q_weight, b_scales, nullptr, nullptr, nullptr, workspace,
                          workspace_len, scratch, scratch_bytes, 4, size_m, size_n, size_k,
                          num_groups, 1, dtype, stream);
}

/* ------------------------------------------------------------------------------------------------
 * MFMA-native W4 tensors (not ops of nm-vllm 0.5.1: what GPTQMarlinLinearMethod keeps beside the Marlin
 * tensor for decode-sized calls; w4a16_gemm.hip "The MFMA-native weight layout").
 * nmv_w4_native_repack: GPTQ qweight int32 [K/8, N] (+ optional act-order row gather perm[K]) -> native int32
 * [N/64 * K/32 * 256].  nmv_w4_native_gemm: C = A . ((q - 8) * s) with s the NATURAL [groups, N] scale tensor;
 * mode 0 = model-dtype output [M, N] (split-K inside the launch: workspace = zeroed tickets, scratch = slabs),
 * mode 1 = silu(gate) * up on column-interleaved gate_up weights, output [M, N/2], mode 2 = deferred reduction:
 * fp32 slabs [splits, M, N] in `scratch`, no output.  M tiles up to 64 rows, K % 256 == 0, N % 64 == 0. */
extern "C" int nmv_w4_native_repack(const int32_t* qweight, const int32_t* perm, int32_t* out, int size_k,
                                    int size_n, void* stream) {
  NMV_CHECK(size_k % 32 == 0 && size_n % 64 == 0, "w4_native_repack: K %% 32 and N %% 64 required");
  const int64_t total = (int64_t)(size_k / 8) * size_n;
  if (total == 0) return NMV_OK;
  hipLaunchKernelGGL(w4_native_repack_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint32_t*)qweight, perm, (uint32_t*)out, size_k, size_n);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_w4_native_gemm_splits(int size_m, int size_n, int size_k, int num_groups) {
  if (size_m <= 0 || size_n <= 0 || size_k <= 0 || size_n % 64 != 0 || size_k % 256 != 0) return 0;
  // group 128 takes w4a16_ring.hip (17..64 rows) or w4a16_stream.hip (as nmv_gptq_marlin_gemm_partial_splits)
  W4PrefillPlan pp;
  if (num_groups > 1 && size_k / num_groups == 128 && w4p_make_plan(size_m, size_n, size_k, INT64_MAX, false, &pp))
    return pp.splits;
  W4RingPlan rp;
  if (num_groups > 1 && size_k / num_groups == 128 && w4r_make_plan(size_m, size_n, size_k, INT64_MAX, false, &rp))
    return rp.splits;
  W4StreamPlan sp;
  if (num_groups > 1 && size_k / num_groups == 128 && w4s_make_plan(size_m, size_n, size_k, INT64_MAX, false, true, &sp))
    return sp.splits;
  const GemmPlan pl = make_plan(size_m, size_n, size_k, INT64_MAX, true, 4, false, false, true);
  return (pl.tall && pl.mt <= 4) ? pl.splits : 0;
}

extern "C" int nmv_w4_native_prefill_plan(int size_m, int size_n, int size_k) {
  return (size_m > 0 && size_n > 0 && size_k > 0 && w4p_wins(size_m, size_n, size_k)) ? 1 : 0;
}

extern "C" int nmv_w4_native_gemm(void* c, const void* a, const int32_t* b_native, const void* b_scales,
                                  int32_t* workspace, int64_t workspace_len, void* scratch, int64_t scratch_bytes,
                                  int size_m, int size_n, int size_k, int num_groups, nmv_dtype_t dtype, int mode,
                                  void* stream) {
  NMV_CHECK(mode >= 0 && mode <= 3, "w4_native_gemm: mode must be 0, 1, 2 or 3");
  const bool deferred = mode >= 2;
  return marlin_gemm_impl(c, a, b_native, b_scales, nullptr, nullptr, nullptr, deferred ? nullptr : workspace,
                          deferred ? 0 : workspace_len, scratch, scratch_bytes, 4, size_m, size_n, size_k, num_groups, 1,
                          dtype, stream, deferred ? 2 : mode, 1, 0, mode == 3 ? 1 : 0);
}

/* mode 3 of nmv_w4_native_gemm -- deferred reduction with the slabs in the MODEL dtype, slab[splits][M][N] of 2-byte
 * elements, for nmv_fused_add_rms_norm_partial16 / nmv_rotary_embedding_and_cache_partial16 to sum -- is served by the
 * prompt-sized kernels only: 1 when a call of these sizes takes them (then nmv_w4_native_gemm_splits is its slab count). */
extern "C" int nmv_w4_native_gemm_slab16(int size_m, int size_n, int size_k) {
  W4PrefillPlan pp;
  return (size_m > 0 && size_n > 0 && size_k > 0 && w4p_make_plan(size_m, size_n, size_k, INT64_MAX, false, &pp)) ? 1 : 0;
}
