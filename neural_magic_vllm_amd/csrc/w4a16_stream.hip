// W4A16 GEMM for decode batches (M <= 64) on the GPTQ-Marlin interchange tensor: the round-3 kernel.
//
// Behavioural reference: /root/reference/csrc/quantization/gptq_marlin/gptq_marlin.cu (kernel :396-1363; its
// striped partitioning :423-470 and its reduction at stripe boundaries :995-1110 are the reference's own
// answer to "few rows, wide N, few SMs") -- same op, same tensors, C[M,N] = A[M,K] . ((q - 8) * s[k/128, n]).
//
// What rounds 1-2 measured on the "tall" kernel (w4a16_gemm.hip): at M <= 64 the launch is
// instruction-issue bound, not HBM bound -- ~70 vector instructions per KiB of weights at M <= 16, twice the
// expansion work at M = 64 (two 32-row blocks) -- and every 64-k stage ends in a workgroup barrier that waits for
// activation loads issued at the top of the same stage.  This kernel changes the structure instead of the tile:
//
//   * A wave owns 64 columns x 16 MT rows (MT = 1, 2, 4): a weight is expanded ONCE for up to 64 rows.
//   * Activations: either the workgroup's whole k range is staged into LDS once, in MFMA-operand order, and the
//     main loop has NO barrier and no activation traffic at all ("resident", M <= 16 / 32), or they are streamed
//     in 128-k or 256-k stages per k group through a double buffer (one barrier per 64-128 MFMAs of a wave, loads
//     issued a whole stage ahead of their LDS write; M = 33..64).
//   * 8 (or 4) waves = CPW column chunks x P k groups; the k groups meet in LDS after the loop, so narrow
//     projections need few or no split-K slabs and two waves that share a k group share its activations.
//   * Weights go straight from a bounds-checked buffer load to registers in a ring PER = D + 1 scale groups
//     deep (D = 3: 12 KiB in flight per wave); a load past the wave's range is turned off by an out-of-range
//     offset (no memory request, no branch in the loop).
//   * Zero point without a second MFMA chain and without an add in the flush:  sum (16 + q) a - 24 sum a :  S (the
//     group's sum of activations per row) is computed with v_dot2 by the threads that stage the activations and kept in
//     LDS; after the main loop ONE more MFMA k-step per 16 groups adds sum_g s[g, n] * (-24 S[g, m]).  The flush is
//     one fma per accumulator element and group:  acc += s[group, n] * acc_group.
//   * Two forms of the same kernel (template flag NV): on the Marlin interchange tensor (the reference op), and on the
//     MFMA-native tensor of nmv_w4_native_repack with natural scales (what the decode step uses: no lane exchange, no
//     byte gather, no activation transpose).  The 64-row streamed stage is written out step by step (HS), see there.
//   * What the stage is bound by, and the probes that did not move it: DESIGN.md 3.2, profiles/r03_gemm_ablation.txt
//     (development switches NMV_W4S_ABL_*, NMV_W4S_PARK_KS below; tools/debug/abl_w4s.sh).
// The epilogue forms (model-dtype store, silu(gate) * up, fp32 slabs + ticket + last-arriver sum, deferred slabs)
// are those of the tall kernel, bit-compatible with its consumers (slabs are summed in split order from +0).
#include <climits>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "w4a16_common.h"

namespace nmv {

namespace {

template <int CTRL>
__device__ __forceinline__ float dpp_mov_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row (every lane gets it): xor 1, xor 2, mirror inside 8, mirror inside 16
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov_f<0xB1>(v);
  v += dpp_mov_f<0x4E>(v);
  v += dpp_mov_f<0x141>(v);
  v += dpp_mov_f<0x140>(v);
  return v;
}

// Diagnostic build (-DNMV_W4S_STAMPS, a separate .so loaded through NMV_HIP_LIB): every wave records the 100 MHz
// wall clock at phase boundaries into a device array that nmv_dbg_w4s_stamps copies out (tools/debug/w4s_timeline.py).
#ifdef NMV_W4S_STAMPS
__device__ unsigned long long g_w4s_stamps[1 << 18];
#define W4S_STAMP(i)                                                                                              \
  do {                                                                                                            \
    if (lane == 0)                                                                                                \
      g_w4s_stamps[(((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NW + wave) * 8 + (i)] =    \
          __builtin_amdgcn_s_memrealtime();                                                                       \
  } while (0)
#else
#define W4S_STAMP(i)
#endif

constexpr uint32_t OOB_OFF = 0x7ffffff0u;   // a voffset no buffer of ours reaches: the load returns 0, no request

template <int AUX>
__device__ __forceinline__ uint4 buf_ld16(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, AUX);
  return make_uint4(v.x, v.y, v.z, v.w);
}

}  // namespace

// LDS (dynamic), NBUF = 1 (resident) or 2:
//   a_s  [NBUF][P * g_stage groups][4 k-steps][4 g][MP rows] uint4   the B operand of lane (row, g)
//   ns_s [NBUF][P * g_stage][MP] float                                -24 * sum of the group's activations
//   sc_s [NBUF][P * g_stage][CPW][4 g][4 reg][4 j] float              scales of the lane's output fragment
// after the loop the front of it is reused for the k-group reduction: (P - 1) * CPW tiles of MT * 4 KiB.
// Ring: RS = 4 D slots of one k-step (16 bytes per lane) each; a slot is refilled with the k-step RS ahead as soon
// as its content has been copied out, so 4 D - 1 .. 4 D k-steps are in flight per wave and slot numbers repeat
// every D groups (the loops below run whole periods so that they are static).
// NV: `b` is the MFMA-native tensor of nmv_w4_native_repack (w4a16_gemm.hip: native[kstep][chunk][lane], a lane's 16
// bytes ARE its four A operands, k in natural order, pair p = nibbles p and p + 4) and `s` the natural [groups, N]
// scale tensor -- no lane exchange, no byte gather, one pair per dword needs no shift (bf16: 128 + q), and a 16-byte
// piece of an activation row is the B operand as it stands (no 4 x 4 dword transpose on the way into LDS).
template <typename T, int MT, int NW, int CPW, int D, int GST /* groups per stage; 0 = resident */, bool NV = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void w4a16_stream_kernel(const GemmParams p) {
  static_assert(NW % CPW == 0, "waves = chunks x k groups");
  constexpr int P = NW / CPW, MP = 16 * MT, NTHR = NW * 64, RS = 4 * D;
  constexpr bool RES = GST == 0;
  constexpr int NBUF = RES ? 1 : 2;
#ifdef NMV_W4S_NO_HS
  constexpr bool HS = false;
#else
  // the 64-row streamed stage is issued in a hand-written order (see "hand-scheduled stage" below)
  constexpr bool HS = MT == 4 && GST == 1 && D == 1;
#endif
  // scale image in LDS: [g][j][reg] (one float4 per 16-column tile: native scales arrive in that order, and the
  // hand-scheduled stage flushes tile by tile) or [g][reg][j] (one permuted 16-byte piece = two float4)
  constexpr bool SC_JR = NV || HS;
  constexpr int LOG_MP = MT == 1 ? 4 : MT == 2 ? 5 : 6;
  constexpr int LOG_CPW = CPW == 1 ? 0 : CPW == 2 ? 1 : 2;
  static_assert(CPW == 1 || CPW == 2 || CPW == 4, "chunks per workgroup");
  static_assert(RES || GST % D == 0, "a stage is whole ring periods");
  static_assert(RES || MT >= 2, "streamed activations: 32 or 64 rows");
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = wave % CPW, kp = wave / CPW;
  const int r = lane & 15, g = lane >> 4, blk = r >> 3, n_in = r & 7;
  const int n_chunks = p.N >> 6;
  const int chunk = blockIdx.x * CPW + c;
  const bool chunk_ok = chunk < n_chunks;
  const int m0 = blockIdx.z * MP;
  const int split = blockIdx.y;
  const int k_wg0 = split * p.k_per_wg;
  const int g_stage = RES ? p.g_stage : GST;
  const int n_stages = RES ? 1 : p.n_stages;
  const int GW = g_stage * n_stages;          // scale groups of one wave
  const int k_w0 = k_wg0 + kp * GW * 128;
  W4S_STAMP(0);

  const int ngrp_buf = P * g_stage;
  const int a_buf_u4 = ngrp_buf * 16 * MP;
  const int sc_buf = ngrp_buf * CPW * 64;
  uint4* a_s = smem;
  float* ns_s = reinterpret_cast<float*>(a_s + NBUF * a_buf_u4);   // [P][GW][MP]: -24 * sum(a) of EVERY group (kept for the end)
  float* sc_s = ns_s + P * GW * MP;

  // ---- buffer resources: bounds-checked on the per-lane offset, so rows past M, chunks past N and ring
  //      look-ahead past the k range cost neither a branch nor a memory request ----
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint4*>(p.b), 0, (int)(((int64_t)p.K * p.N) >> 1), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.a), 0, (int)((int64_t)p.M * p.K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.s), 0, (int)((int64_t)(p.K >> 7) * p.N * 2), 0x00020000);

  // ---- weights: lane (blk, n_in, q = g) streams vector n_in*4+q of k-tile 2 ks + blk (w4a16_gemm.hip) ----
  const uint32_t row_bytes = (uint32_t)p.N * 8;               // one k-tile row of the Marlin tensor
  // native: a k-step of a chunk is 1 KiB contiguous (lane's vector at 16 lane), a k-step of all chunks N * 16 bytes
  const uint32_t w_voff = !chunk_ok ? OOB_OFF
                          : NV      ? ((uint32_t)chunk * 64 + lane) * 16
                                    : ((uint32_t)chunk * 32 + n_in * 4 + g) * 16 + blk * row_bytes;
  const uint32_t w_s0 = (uint32_t)(k_w0 >> 4) * row_bytes;    // uniform
  const uint32_t kmask = __builtin_amdgcn_readfirstlane(NV ? W4N<T>::MASK : W4<T>::MASK);
  uint32_t kmagic = NV ? W4N<T>::MAGIC : W4<T>::MAGIC;
  constexpr float ZPC = NV ? W4N<T>::ZPC : W4_ZP;   // dequantised code = q + (ZPC - 8)
  asm volatile("" : "+v"(kmagic));
  // v_perm selector ({own, partner} = bytes 7..4, 3..0): block 0 lanes hold the even k-tile and take bytes 0, 2 of
  // both words, block 1 lanes hold the odd k-tile and take bytes 1, 3
  const uint32_t bsel = blk ? 0x07030501u : 0x02060004u;

  uint4 wq[RS];
  // k-step `kstep` of this wave (flat) into ring slot SL; `vo` = w_voff, or OOB_OFF past the wave's range
  auto load_w = [&](auto slot_tag, int kstep, uint32_t vo) {
    constexpr int SL = decltype(slot_tag)::value;
    wq[SL] = buf_ld16<0>(rs_w, vo, w_s0 + (uint32_t)kstep * 2 * row_bytes);
  };

  // ---- activation staging: a thread moves UNITS of (row, group, k-step) = 64 contiguous bytes = four 16-byte
  //      pieces; dword i of piece cc is slot cc of lane group g = i, so the four planes of the k-step each get ONE
  //      16-byte store.  Lanes run along k (coalesced loads: 16 lanes = 1 KiB of a row); the row index inside a
  //      plane is XORed with (k-step << 1 | group & 1) so that the 8 lanes of a store phase (2 rows x 4 k-steps,
  //      or 2 groups x 4 k-steps of one row) hit 8 different bank quads; the reader applies the same XOR.
  //      The unit's 32 activations are summed (v_dot2) and the 4 lanes of a group combine: -24 * sum -> ns_s ----
  const int kst_ = tid & 3, rq0 = tid >> 2;
  constexpr int RQ_STEP = NTHR / 4;
  const uint32_t ones2 = W4<T>::ONES;
  // zg: flat index (k group * GW + group of the wave) of the unit's group in ns_s
  auto a_store = [&](int row, int gb, int buf, int zg, const uint4 (&v)[4]) {
    uint4* base = a_s + buf * a_buf_u4 + gb * 16 * MP + (kst_ * 4) * MP + (row ^ ((kst_ << 1) | (gb & 1)));
    if constexpr (NV) {   // piece cc = k 8 cc .. 8 cc + 7 of the k-step = the operand of lane group cc
      base[0] = v[0];
      base[MP] = v[1];
      base[2 * MP] = v[2];
      base[3 * MP] = v[3];
    } else {
      base[0] = make_uint4(v[0].x, v[1].x, v[2].x, v[3].x);
      base[MP] = make_uint4(v[0].y, v[1].y, v[2].y, v[3].y);
      base[2 * MP] = make_uint4(v[0].z, v[1].z, v[2].z, v[3].z);
      base[3 * MP] = make_uint4(v[0].w, v[1].w, v[2].w, v[3].w);
    }
    float sum = 0.f;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      sum = T::dot2(v[cc].x, ones2, sum);
      sum = T::dot2(v[cc].y, ones2, sum);
      sum = T::dot2(v[cc].z, ones2, sum);
      sum = T::dot2(v[cc].w, ones2, sum);
    }
    sum += dpp_mov_f<0xB1>(sum);   // the four k-steps of the group sit in one quad of lanes
    sum += dpp_mov_f<0x4E>(sum);
    if (kst_ == 0 && zg >= 0) ns_s[zg * MP + row] = -ZPC * sum;
  };
  // scale rows: 16-byte piece pc of (group gb, chunk cc) -> two float4 of the fragment image
  auto s_store = [&](int id2, int buf, uint4 v) {
    const int pc = id2 & 7, gbc = id2 >> 3;        // gbc = gb * CPW + cc
    f32x4_t* dst = reinterpret_cast<f32x4_t*>(sc_s + buf * sc_buf + gbc * 64);
    if constexpr (NV) {
      // natural order: the piece is columns 8 pc .. 8 pc + 7 = tile j = pc / 2, lane groups 2 (pc & 1) and + 1, reg 0..3;
      // the image is [g][j][reg] (the flush reads one float4 per tile)
      const f32x4_t h0 = {lo_f<T>(v.x), hi_f<T>(v.x), lo_f<T>(v.y), hi_f<T>(v.y)};
      const f32x4_t h1 = {lo_f<T>(v.z), hi_f<T>(v.z), lo_f<T>(v.w), hi_f<T>(v.w)};
      const int j = pc >> 1, g0 = 2 * (pc & 1);
      dst[g0 * 4 + j] = h0;
      dst[(g0 + 1) * 4 + j] = h1;
    } else if constexpr (SC_JR) {
      // permuted piece (g_lo = pc / 4, reg = pc % 4; dword j = tile j, halves = lane groups g_lo, g_lo + 2) into [g][j][reg]
      float* d1 = reinterpret_cast<float*>(dst);
      const int g_lo = pc >> 2, reg = pc & 3;
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        d1[(g_lo * 4 + j) * 4 + reg] = lo_f<T>(w[j]);
        d1[((g_lo + 2) * 4 + j) * 4 + reg] = hi_f<T>(w[j]);
      }
    } else {
      const f32x4_t h0 = {lo_f<T>(v.x), lo_f<T>(v.y), lo_f<T>(v.z), lo_f<T>(v.w)};
      const f32x4_t h1 = {hi_f<T>(v.x), hi_f<T>(v.y), hi_f<T>(v.z), hi_f<T>(v.w)};
      const int g_lo = pc >> 2, reg = pc & 3;
      dst[g_lo * 4 + reg] = h0;
      dst[(g_lo + 2) * 4 + reg] = h1;
    }
  };
  // per-thread part of a scale piece's offset (chunk, piece); OOB for chunks past N
  auto s_voff_thread = [&](int id2) -> uint32_t {
    const int pc = id2 & 7, cc = (id2 >> 3) & (CPW - 1);
    const int ch = blockIdx.x * CPW + cc;
    return ch < n_chunks ? (uint32_t)((ch * 64 + pc * 8) * 2) : OOB_OFF;
  };

  // ---- accumulators ----
  f32x4_t accm[4][MT];
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < MT; ++t) accm[j][t] = zero4;

  // one 128-k scale group (flat index gi of this wave) from ring slots 4 U .. 4 U + 3; gb = its group inside
  // activation buffer `buf`.  `mid` runs between k-steps 1 and 2 (the streamed form parks half a stage there),
  // `pre_flush` before the scales are read (the first group parks the wave's own scale rows there).
  auto group_compute = [&](auto u_tag, int gi, int gb, int buf, auto&& mid, auto&& pre_flush) {
    constexpr int U = decltype(u_tag)::value;
    const uint4* a_g = a_s + buf * a_buf_u4 + gb * 16 * MP;
    const uint32_t vo_next = (gi + D < GW) ? w_voff : OOB_OFF;   // uniform condition
    const int gsw = gb & 1;
    f32x4_t accg[4][MT];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks == 2) mid();
      uint4 af[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) af[t] = a_g[(ks * 4 + g) * MP + ((t * 16 + r) ^ ((ks << 1) | gsw))];
      const uint32_t own[4] = {wq[4 * U + ks].x, wq[4 * U + ks].y, wq[4 * U + ks].z, wq[4 * U + ks].w};
      // 16 / 32-row tiles: the slot is free as soon as it has been copied out -- fetch the k-step RS ahead into it;
      // the 64-row tile has no registers for the copy and refills the slot behind the k-step's last expansion
      if constexpr (MT < 4) {
        if (ks == 0) load_w(std::integral_constant<int, 4 * U + 0>{}, (gi + D) * 4 + 0, vo_next);
        if (ks == 1) load_w(std::integral_constant<int, 4 * U + 1>{}, (gi + D) * 4 + 1, vo_next);
        if (ks == 2) load_w(std::integral_constant<int, 4 * U + 2>{}, (gi + D) * 4 + 2, vo_next);
        if (ks == 3) load_w(std::integral_constant<int, 4 * U + 3>{}, (gi + D) * 4 + 3, vo_next);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // lanes r and r^8 hold the two k-tiles of the same vector.  A word is [blk0 | blk1 | blk0 | blk1] bytes
        // (nibbles k 2q, 2q+8 | same | k 2q+1, 2q+9 | same): fetch the partner's word with one DPP row rotate and
        // gather MY block's four bytes of both k-tiles with one v_perm -> [E.x, O.x, E.y, O.y] (E / O = even / odd
        // k-tile), so that the four rotate amounts below are the same for every lane
        uint4 wv;
        if constexpr (NV) {
          // pair p = nibbles p and p + 4 of the dword: bring nibble p to bit POS of the low half
          const uint32_t x = own[j];
          constexpr int P0 = W4N<T>::POS;
          wv = make_uint4(and_or(P0 == 0 ? x : x << P0, kmask, kmagic),
                          and_or(4 >= P0 ? x >> (4 - P0) : x << (P0 - 4), kmask, kmagic),
                          and_or(x >> (8 - P0), kmask, kmagic), and_or(x >> (12 - P0), kmask, kmagic));
        } else {
          const uint32_t pw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)own[j], 0x128, 0xf, 0xf, true);
          const uint32_t mw = __builtin_amdgcn_perm(own[j], pw, bsel);
          wv = make_uint4(and_or(__builtin_amdgcn_alignbit(mw, mw, W4<T>::ROT_LO0), kmask, kmagic),
                          and_or(__builtin_amdgcn_alignbit(mw, mw, W4<T>::ROT_HI0), kmask, kmagic),
                          and_or(__builtin_amdgcn_alignbit(mw, mw, W4<T>::ROT_LO1), kmask, kmagic),
                          and_or(__builtin_amdgcn_alignbit(mw, mw, W4<T>::ROT_HI1), kmask, kmagic));
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) accg[j][t] = W4<T>::mfma(wv, af[t], ks == 0 ? zero4 : accg[j][t]);
        if constexpr (MT >= 4) __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (MT >= 4) {
        if (ks == 0) load_w(std::integral_constant<int, 4 * U + 0>{}, (gi + D) * 4 + 0, vo_next);
        if (ks == 1) load_w(std::integral_constant<int, 4 * U + 1>{}, (gi + D) * 4 + 1, vo_next);
        if (ks == 2) load_w(std::integral_constant<int, 4 * U + 2>{}, (gi + D) * 4 + 2, vo_next);
        if (ks == 3) load_w(std::integral_constant<int, 4 * U + 3>{}, (gi + D) * 4 + 3, vo_next);
      }
    }
    pre_flush();
    const f32x4_t* sc_g = reinterpret_cast<const f32x4_t*>(sc_s + buf * sc_buf + (gb * CPW + c) * 64);
#pragma unroll
    for (int x = 0; x < 4; ++x) {     // [g][reg][j]: x = reg, the float4 runs over the tiles; [g][j][reg]: x = tile, over reg
      const f32x4_t s4 = sc_g[g * 4 + x];
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        const int reg = SC_JR ? y : x, j = SC_JR ? x : y;
#pragma unroll
        for (int t = 0; t < MT; ++t) accm[j][t][reg] = fmaf(s4[y], accg[j][t][reg], accm[j][t][reg]);
      }
    }
  };
  auto nop = [] {};

  // first k of group gb of a buffer holding stage st
  auto grp_k0 = [&](int gb, int st) -> int {
    if constexpr (RES) return k_wg0 + gb * 128;
    else return k_wg0 + ((gb / GST) * GW + st * GST + (gb % GST)) * 128;
  };
  // ---- prologue: EVERY load of the prologue is issued before the first wait -- activations of stage 0 (L2),
  //      then the ring (HBM), then the wave's own scale rows of stage 0 (HBM; parked by the first group just
  //      before its flush, so they are not on the path to the first MFMA) -- straight-line code; loads that are
  //      not needed are switched off by their offset ----
  constexpr int MAXG = (128 * 1024) / (MP * 256);
  constexpr int WSB = RES ? (MAXG * 8 + 63) / 64 : (GST * 8 + 63) / 64;   // own scale pieces per lane
  uint4 sv[WSB];
  auto park_own_scales = [&] {
#pragma unroll
    for (int i = 0; i < WSB; ++i) {
      const int l2 = lane + 64 * i;
      const int gi = l2 >> 3, pc = l2 & 7;
      if (gi < g_stage) s_store(((kp * g_stage + gi) * CPW + c) * 8 + pc, 0, sv[i]);
    }
  };
  {
    // resident: only rows < M are staged; (group, row) pairs are numbered with the row count padded to a power of two
    const int rows_valid = min(MP, p.M - m0);
    const int lr = rows_valid > 1 ? 32 - __builtin_clz(rows_valid - 1) : 0;     // uniform
    constexpr int UB_RES = (MAXG * MP * 4) / NTHR < 4 ? (MAXG * MP * 4) / NTHR : 4;   // the plan keeps g_wg * rows within it
    constexpr int UB = RES ? UB_RES : (P * GST * MP * 4) / NTHR;                // units per thread
    uint4 av[UB][4];
    // unit i of this thread: (row, group of the buffer), or group -1 when it lies outside the staged rows / groups
    auto unit = [&](int i, int& row, int& gb) {
      const int rq = rq0 + i * RQ_STEP;
      if constexpr (RES) { row = rq & ((1 << lr) - 1); gb = rq >> lr; }
      else { row = rq & (MP - 1); gb = rq >> LOG_MP; }
      if (gb >= ngrp_buf || row >= rows_valid) gb = -1;
    };
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      int row, gb;
      unit(i, row, gb);
      const uint32_t vo = gb >= 0 ? (uint32_t)(((m0 + row) * p.K + grp_k0(gb, 0) + kst_ * 32) * 2) : OOB_OFF;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) av[i][cc] = buf_ld16<0>(rs_a, vo, cc * 16);
    }
    // the ring's first group only: the memory pipeline of a CU serves requests in arrival order, so a deep ring
    // issued by eight waves at once (96 KiB of HBM misses) would stand between the last waves' activation loads
    // and the barrier; slots 4 .. RS-1 are issued after the activations have been parked
    auto pro = [&](auto s_tag) {
      constexpr int SL = decltype(s_tag)::value;
      if constexpr (SL < RS) load_w(s_tag, SL, SL < 4 * GW ? w_voff : OOB_OFF);
    };
    pro(std::integral_constant<int, 0>{});  pro(std::integral_constant<int, 1>{});
    pro(std::integral_constant<int, 2>{});  pro(std::integral_constant<int, 3>{});
    {
      const int ch_ok = chunk_ok ? 1 : 0;
#pragma unroll
      for (int i = 0; i < WSB; ++i) {
        const int l2 = lane + 64 * i;
        const int gi = l2 >> 3, pc = l2 & 7;
        const bool ok = ch_ok && gi < g_stage;
        sv[i] = buf_ld16<0>(rs_s, ok ? (uint32_t)((chunk * 64 + pc * 8) * 2 + ((grp_k0(kp * g_stage + gi, 0) >> 7) * p.N) * 2) : OOB_OFF, 0);
      }
    }
    W4S_STAMP(1);
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      int row, gb;
      unit(i, row, gb);
      if (gb >= 0) a_store(row, gb, 0, RES ? gb : (gb / (RES ? 1 : GST)) * GW + (gb % (RES ? 1 : GST)), av[i]);   // rows past M are never read back
    }
    pro(std::integral_constant<int, 4>{});  pro(std::integral_constant<int, 5>{});
    pro(std::integral_constant<int, 6>{});  pro(std::integral_constant<int, 7>{});
    pro(std::integral_constant<int, 8>{});  pro(std::integral_constant<int, 9>{});
    pro(std::integral_constant<int, 10>{}); pro(std::integral_constant<int, 11>{});
    pro(std::integral_constant<int, 12>{}); pro(std::integral_constant<int, 13>{});
    pro(std::integral_constant<int, 14>{}); pro(std::integral_constant<int, 15>{});
  }
  if constexpr (!RES) park_own_scales();   // streamed launches are long: one instance of the group body matters more
  W4S_STAMP(2);
  __syncthreads();
  W4S_STAMP(3);

  if constexpr (RES) {
    // resident: no barrier, no activation traffic; D groups per trip with static ring slots
    group_compute(std::integral_constant<int, 0>{}, 0, kp * g_stage, 0, nop, park_own_scales);
    {
      auto body = [&](auto u_tag) {
        constexpr int U = decltype(u_tag)::value;
        if constexpr (U >= 1 && U < D) {
          if (U < GW) group_compute(u_tag, U, kp * g_stage + U, 0, nop, nop);
        }
      };
      body(std::integral_constant<int, 1>{});
      body(std::integral_constant<int, 2>{});
      body(std::integral_constant<int, 3>{});
    }
    for (int G0 = D; G0 < GW; G0 += D) {
      auto body = [&](auto u_tag) {
        constexpr int U = decltype(u_tag)::value;
        if constexpr (U < D) {
          if (G0 + U < GW) group_compute(u_tag, G0 + U, kp * g_stage + G0 + U, 0, nop, nop);
        }
      };
      body(std::integral_constant<int, 0>{});
      body(std::integral_constant<int, 1>{});
      body(std::integral_constant<int, 2>{});
      body(std::integral_constant<int, 3>{});
    }
  } else {
    // streamed: one stage per trip (ring slots repeat every D groups and GST % D == 0: static; buffer = st & 1).
    // A thread moves UPT units per stage, in AH batches: a batch is fetched two k-steps (>= 32 MFMAs of the wave)
    // before it is parked in the other buffer; the unit's (row, group) is (rq0 mod MP, wave-uniform group).
    constexpr int UPT = (P * GST * MP * 4) / NTHR;
    static_assert((P * GST * MP * 4) % NTHR == 0 && RQ_STEP % MP == 0, "whole units per thread, one row per thread");
#ifdef NMV_W4S_AH2
    constexpr int AH = (MT == 4 && GST == 1 && UPT == 2) ? 2 : 1;          // batches per stage
#else
    constexpr int AH = 1;   // the whole stage is fetched at its top and parked at its bottom: a stage of latency cover
#endif
    constexpr int UPB = UPT / AH;                                          // units per batch
    constexpr int SPT = (P * GST * CPW * 8 + NTHR - 1) / NTHR;
    const int s_row = rq0 & (MP - 1);
    const int gb0 = __builtin_amdgcn_readfirstlane(rq0 >> LOG_MP);        // uniform in a wave: RQ_STEP / MP groups apart per unit
    constexpr int GB_STEP = RQ_STEP >> LOG_MP;
    const uint32_t a_vo = m0 + s_row < p.M ? (uint32_t)(((m0 + s_row) * p.K + kst_ * 32) * 2) : OOB_OFF;
    // scale pieces of later stages: thread-dependent group -> its row offset goes into the vector offset
    uint32_t s_vo[SPT];
#pragma unroll
    for (int i = 0; i < SPT; ++i) {
      const int id2 = tid + i * NTHR;
      const int gb = id2 >> (3 + LOG_CPW);
      const uint32_t vt = s_voff_thread(id2);
      s_vo[i] = (id2 < P * GST * CPW * 8 && vt != OOB_OFF) ? vt + (uint32_t)((grp_k0(gb, 0) >> 7) * p.N * 2) : OOB_OFF;
    }
    const uint32_t s_stage_bytes = (uint32_t)(GST * p.N * 2);   // one stage further: GST scale rows
    if constexpr (HS) {
      // ---- hand-scheduled stage (64 rows, one 128-k group per stage and k group) ----
      // rocprofv3 counters of the compiler-scheduled stage (profiles/r03_gemm_pmc.txt): MFMA pipe 22 % busy, vector
      // issue 38 %, a third of the wave-cycles in s_waitcnt -- neither unit is the limit, the order is: every k-step
      // opened with four LDS reads and waited for them, a word was expanded and THEN its four MFMAs issued, and
      // parking the next stage (stores, 32 dependent v_dot2) and the flush ran with the MFMA pipe idle, in all eight
      // waves at once because the barrier aligns them.  Here the source order IS the schedule (a sched_barrier after
      // every MFMA): each MFMA is followed by the vector work that fits in its shadow --
      //   * the expansion of the NEXT step's word (a step = one 16-column tile of one k-step: 4 MFMAs),
      //   * k-step 2: the next stage's activation pieces go to LDS and into the running sums, piece by piece,
      //   * k-step 3: the flush of the tile whose group chain ended one step earlier, with scales read a step ahead --
      // and the B operands of k-step ks + 1 are requested during step (ks, 1).  What is left at the end of a stage is
      // the flush of the last tile, two DPP adds and the barrier.
      static_assert(UPT == 1 || UPT == 2 || UPT == 4, "activation units per thread and stage");
#ifndef NMV_W4S_PARK_KS
#define NMV_W4S_PARK_KS 2
#endif
      constexpr int PARK_KS = NMV_W4S_PARK_KS;   // the k-step in whose MFMA shadow the next stage is parked (probe: 1 / 2 / 3)
      const int gsw = kp & 1;                       // group of the buffer = kp (GST == 1)
      const uint4* a_rd[4];                         // plane (ks, g) of buffer 0, swizzled row r; + 16 t is an immediate
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a_rd[ks] = a_s + kp * 16 * MP + (ks * 4 + g) * MP + (r ^ ((ks << 1) | gsw));
      uint4* a_wr[UPT];
      int z_idx[UPT];
#pragma unroll
      for (int u = 0; u < UPT; ++u) {
        const int gb = gb0 + u * GB_STEP;
        a_wr[u] = a_s + gb * 16 * MP + (kst_ * 4) * MP + (s_row ^ ((kst_ << 1) | (gb & 1)));
        z_idx[u] = gb * GW * MP + s_row;
      }
      auto word = [&](int sl, int jj) -> uint32_t {
        return jj == 0 ? wq[sl].x : jj == 1 ? wq[sl].y : jj == 2 ? wq[sl].z : wq[sl].w;
      };
      // quarter `part` of the expansion of word x into w (tmp: the gathered word of the Marlin form)
      auto xpart = [&](int part, uint32_t x, uint4& w, uint32_t& tmp) {
        if constexpr (NV) {
          constexpr int P0 = W4N<T>::POS;
          if (part == 0) w.x = and_or(P0 == 0 ? x : x << P0, kmask, kmagic);
          if (part == 1) w.y = and_or(4 >= P0 ? x >> (4 - P0) : x << (P0 - 4), kmask, kmagic);
          if (part == 2) w.z = and_or(x >> (8 - P0), kmask, kmagic);
          if (part == 3) w.w = and_or(x >> (12 - P0), kmask, kmagic);
        } else {
          if (part == 0) {
            const uint32_t pw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, true);
            tmp = __builtin_amdgcn_perm(x, pw, bsel);
            w.x = and_or(__builtin_amdgcn_alignbit(tmp, tmp, W4<T>::ROT_LO0), kmask, kmagic);
          }
          if (part == 1) w.y = and_or(__builtin_amdgcn_alignbit(tmp, tmp, W4<T>::ROT_HI0), kmask, kmagic);
          if (part == 2) w.z = and_or(__builtin_amdgcn_alignbit(tmp, tmp, W4<T>::ROT_LO1), kmask, kmagic);
          if (part == 3) w.w = and_or(__builtin_amdgcn_alignbit(tmp, tmp, W4<T>::ROT_HI1), kmask, kmagic);
        }
      };
      uint4 wvn;
      {
        uint32_t tmp = 0;
#pragma unroll
        for (int part = 0; part < 4; ++part) xpart(part, word(0, 0), wvn, tmp);
      }
      for (int st = 0; st < n_stages; ++st) {
        const int SP = st & 1;
        const bool more = st + 1 < n_stages;             // uniform; loads past the end are switched off by their offset
#ifdef NMV_W4S_ABL_A      // development builds (results garbage, times valid): no activation traffic after stage 0
        const uint32_t a_vo_st = OOB_OFF;
#else
        const uint32_t a_vo_st = more ? a_vo : OOB_OFF;
#endif
#ifdef NMV_W4S_ABL_W      // no weight traffic after the first ring fill
        const uint32_t vo_next = OOB_OFF;
#else
        const uint32_t vo_next = more ? w_voff : OOB_OFF;
#endif
        uint4 ar[UPT][4], sr[SPT];
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
          const uint32_t so = (uint32_t)(grp_k0(gb0 + u * GB_STEP, st + 1) * 2);
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) ar[u][cc] = buf_ld16<0>(rs_a, a_vo_st, so + cc * 16);
        }
#pragma unroll
        for (int i = 0; i < SPT; ++i) sr[i] = buf_ld16<0>(rs_s, more ? s_vo[i] : OOB_OFF, (uint32_t)(st + 1) * s_stage_bytes);
        const int rd_off = SP * a_buf_u4, wr_off = (SP ^ 1) * a_buf_u4;
        const f32x4_t* sc_g = reinterpret_cast<const f32x4_t*>(sc_s + SP * sc_buf + (kp * CPW + c) * 64) + g * 4;
        uint4 afc[4], afn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) afc[t] = a_rd[0][rd_off + t * 16];   // (16 t + r) ^ x = 16 t + (r ^ x): x < 8
        f32x4_t accg[4][4];
        f32x4_t s4c = zero4, s4n = zero4;
        float psum[UPT];
#pragma unroll
        for (int u = 0; u < UPT; ++u) psum[u] = 0.f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int sn = (ks * 4 + j + 1) & 15;        // the step whose word is expanded in this one
            const uint32_t xw = word(sn >> 2, sn & 3);
            const uint4 wv = wvn;
            uint32_t tmp = 0;
            if (j == 1 && ks < 3) {
#ifndef NMV_W4S_ABL_LDSRD  // no B-operand reads after k-step 0
#pragma unroll
              for (int t = 0; t < 4; ++t) afn[t] = a_rd[ks + 1][rd_off + t * 16];
#else
#pragma unroll
              for (int t = 0; t < 4; ++t) afn[t] = afc[t];
#endif
            }
            if (ks == 3) {
              s4c = s4n;
              s4n = sc_g[j];
            }
#pragma unroll
            for (int part = 0; part < 4; ++part) {
#ifndef NMV_W4S_ABL_MFMA   // no MFMA: the operands are kept alive by an empty asm
              accg[j][part] = W4<T>::mfma(wv, afc[part], ks == 0 ? zero4 : accg[j][part]);
#else
              asm volatile("" :: "v"(wv.x), "v"(wv.y), "v"(wv.z), "v"(wv.w), "v"(afc[part].x), "v"(afc[part].w));
              if (ks == 0) accg[j][part] = zero4;
#endif
#ifndef NMV_W4S_ABL_EXP    // no expansion: the raw word is the operand
              xpart(part, xw, wvn, tmp);
#else
              wvn = make_uint4(xw, xw, xw, xw);
#endif
#ifndef NMV_W4S_ABL_PARK   // the next stage is fetched but neither stored nor summed
              if (ks == PARK_KS) {
                // pieces q = j UPT .. + UPT - 1 of this thread's stage: dwords d = part UPT .. of the step's 4 UPT
#pragma unroll
                for (int i = 0; i < UPT; ++i) {
                  const int d = part * UPT + i;                 // dword of the step
                  const int q = j * UPT + (d >> 2);             // piece of the stage
                  const int u = q >> 2, cc = q & 3, e = d & 3;  // unit, piece of the unit, dword of the piece
                  if (e == 0) {
                    if constexpr (NV) {
                      a_wr[u][wr_off + cc * MP] = ar[u][cc];
                    } else {   // plane cc of the k-step takes dword cc of the unit's four pieces
                      const uint4 tr = cc == 0   ? make_uint4(ar[u][0].x, ar[u][1].x, ar[u][2].x, ar[u][3].x)
                                       : cc == 1 ? make_uint4(ar[u][0].y, ar[u][1].y, ar[u][2].y, ar[u][3].y)
                                       : cc == 2 ? make_uint4(ar[u][0].z, ar[u][1].z, ar[u][2].z, ar[u][3].z)
                                                 : make_uint4(ar[u][0].w, ar[u][1].w, ar[u][2].w, ar[u][3].w);
                      a_wr[u][wr_off + cc * MP] = tr;
                    }
                  }
                  const uint32_t dv = e == 0 ? ar[u][cc].x : e == 1 ? ar[u][cc].y : e == 2 ? ar[u][cc].z : ar[u][cc].w;
                  psum[u] = T::dot2(dv, ones2, psum[u]);
                }
              }
#endif
              if (ks == 3 && j >= 1) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                  accm[j - 1][part][reg] = fmaf(s4c[reg], accg[j - 1][part][reg], accm[j - 1][part][reg]);
                asm volatile("" : "+v"(accm[j - 1][part]));   // the flush stays in this step (it would sink to the stage's end)
              }
            }
            if (j == 3) {
              // the k-step's slot was expanded for the last time one step ago: refill it with the next group's k-step
              if (ks == 0) load_w(std::integral_constant<int, 0>{}, (st + 1) * 4 + 0, vo_next);
              if (ks == 1) load_w(std::integral_constant<int, 1>{}, (st + 1) * 4 + 1, vo_next);
              if (ks == 2) load_w(std::integral_constant<int, 2>{}, (st + 1) * 4 + 2, vo_next);
              if (ks == 3) load_w(std::integral_constant<int, 3>{}, (st + 1) * 4 + 3, vo_next);
#pragma unroll
              for (int t = 0; t < 4; ++t) afc[t] = afn[t];
            }
            // the order of the step for the machine scheduler: LDS reads first, then four times [one MFMA, its share of
            // the vector work, at most one LDS store], the ring refill last; nothing crosses into the next step
            constexpr int VX = NV ? 2 : 3;                                  // expansion: 7 / 10 ops in four shares
            constexpr int VP = 2 * UPT + (NV ? 0 : 4);                      // sums (and transposes) of the park slices
            constexpr int V2 = VX + VP, V3 = VX + 4 + (PARK_KS == 3 ? VP : 0);   // park step / flush step
            if ((j == 1 && ks < 3) || ks == 3) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int part = 0; part < 4; ++part) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              if (ks == 3) __builtin_amdgcn_sched_group_barrier(0x002, V3, 0);
              else if (ks == PARK_KS) __builtin_amdgcn_sched_group_barrier(0x002, V2, 0);
              else __builtin_amdgcn_sched_group_barrier(0x002, VX, 0);
              if (ks == PARK_KS) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
            if (j == 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // tail: the last tile's flush, the sums, the scale rows of the next stage
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) accm[3][t][reg] = fmaf(s4n[reg], accg[3][t][reg], accm[3][t][reg]);
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
          float sum = psum[u];
          sum += dpp_mov_f<0xB1>(sum);   // the four k-steps of the group sit in one quad of lanes
          sum += dpp_mov_f<0x4E>(sum);
          if (kst_ == 0 && more) ns_s[z_idx[u] + (st + 1) * MP] = -ZPC * sum;
        }
#pragma unroll
        for (int i = 0; i < SPT; ++i)
          if (tid + i * NTHR < P * GST * CPW * 8) s_store(tid + i * NTHR, SP ^ 1, sr[i]);
#ifndef NMV_W4S_ABL_BAR    // no barrier between the stages
        __syncthreads();
#else
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      }
    } else {
  #ifdef NMV_W4S_STAMPS
      unsigned long long acc_compute = 0, acc_park = 0, acc_bar = 0;
  #endif
      for (int st = 0; st < n_stages; ++st) {
        const int SP = st & 1;
  #ifdef NMV_W4S_STAMPS
        const unsigned long long t_a = __builtin_amdgcn_s_memrealtime();
  #endif
        uint4 ar[UPB][4], sr[SPT];
        const bool more = st + 1 < n_stages;             // uniform; loads past the end are switched off by their offset
        const uint32_t a_vo_st = more ? a_vo : OOB_OFF;
        auto fetch = [&](int b) {
  #pragma unroll
          for (int i = 0; i < UPB; ++i) {
            const uint32_t so = (uint32_t)(grp_k0(gb0 + (b * UPB + i) * GB_STEP, st + 1) * 2);
  #pragma unroll
            for (int cc = 0; cc < 4; ++cc) ar[i][cc] = buf_ld16<0>(rs_a, a_vo_st, so + cc * 16);
          }
        };
        auto park = [&](int b) {
  #pragma unroll
          for (int i = 0; i < UPB; ++i) {
            const int gb = gb0 + (b * UPB + i) * GB_STEP;
            a_store(s_row, gb, SP ^ 1, more ? (gb / GST) * GW + (st + 1) * GST + (gb % GST) : -1, ar[i]);   // past the end: zeros, no sum
          }
        };
  #ifndef NMV_W4S_ABLATE_A
        fetch(0);
  #endif
  #pragma unroll
        for (int i = 0; i < SPT; ++i) sr[i] = buf_ld16<0>(rs_s, more ? s_vo[i] : OOB_OFF, (uint32_t)(st + 1) * s_stage_bytes);
        auto mid = [&] {
  #ifndef NMV_W4S_ABLATE_A
          if constexpr (AH == 2) { park(0); fetch(1); }
  #endif
        };
        auto body = [&](auto gi_tag) {
          constexpr int GI = decltype(gi_tag)::value;
          if constexpr (GI < GST) {
            constexpr int U = GI % D;
            if constexpr (GI == 0) {
              group_compute(std::integral_constant<int, U>{}, st * GST + GI, kp * GST + GI, SP, mid, nop);
            } else {
              group_compute(std::integral_constant<int, U>{}, st * GST + GI, kp * GST + GI, SP, nop, nop);
            }
          }
        };
        body(std::integral_constant<int, 0>{});
        body(std::integral_constant<int, 1>{});
  #ifdef NMV_W4S_STAMPS
        const unsigned long long t_b = __builtin_amdgcn_s_memrealtime();
  #endif
  #ifndef NMV_W4S_ABLATE_A
        park(AH - 1);
  #endif
  #pragma unroll
        for (int i = 0; i < SPT; ++i)
          if (tid + i * NTHR < P * GST * CPW * 8) s_store(tid + i * NTHR, SP ^ 1, sr[i]);
  #ifdef NMV_W4S_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t_c = __builtin_amdgcn_s_memrealtime();
  #endif
        __syncthreads();
  #ifdef NMV_W4S_STAMPS
        const unsigned long long t_d = __builtin_amdgcn_s_memrealtime();
        acc_compute += t_b - t_a; acc_park += t_c - t_b; acc_bar += t_d - t_c;
  #endif
      }
  #ifdef NMV_W4S_STAMPS
      if (lane == 0) {
        unsigned long long* q = g_w4s_stamps + (((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NW + wave) * 8;
        q[6] = acc_compute | (acc_park << 32);
        q[7] = acc_bar;
      }
  #endif
    }
  }

  // ---- zero point: acc += sum_g s[g, n] * (-ZPC * S[g, m]) as ONE more MFMA k-step per 16 groups: the k slots are
  //      (group, hi / lo half of the fp32 value split into two bf16), the weight-side operand holds s[g, n] in both
  //      halves.  The split keeps 16 significant bits of a term that is O(24 |sum a| s) -- 2^-17 relative, far below
  //      the model dtype's rounding -- and a one-hot activation row stays exact (-24 has no low half).  The native
  //      bf16 form (128 + q: the term is 2^7 above the signal) splits three ways, the third part in a second MFMA ----
  {
    const float* zs = ns_s + kp * GW * MP;
    constexpr bool IS_F16 = std::is_same<T, F16>::value;
    constexpr int NPASS = (NV && !IS_F16) ? 2 : 1;
    for (int gq = 0; gq < GW; gq += 16) {
      // weight side: lane (r, g) = column 16 j + r, groups gq + 4 g + e.  Marlin: one 16-byte piece of the permuted
      // scale row holds the column's four tiles j (dword j, half r >> 3); native: the dword of columns (n & ~1, + 1)
      uint32_t sraw[4][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int gi = gq + 4 * g + e;
        const bool ok = chunk_ok && gi < GW;
        const uint32_t row_off = (uint32_t)(((k_w0 >> 7) + gi) * p.N * 2);
        if constexpr (NV) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            sraw[e][j] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(
                rs_s, ok ? (int)((uint32_t)((chunk * 64 + 16 * j + (r & ~1)) * 2) + row_off) : (int)OOB_OFF, 0, 0);
        } else {
          const uint4 sp = buf_ld16<0>(rs_s, ok ? (uint32_t)((chunk * 64 + (r & 7) * 8) * 2) + row_off : OOB_OFF, 0);
          sraw[e][0] = sp.x; sraw[e][1] = sp.y; sraw[e][2] = sp.z; sraw[e][3] = sp.w;
        }
      }
      const uint32_t hsel = (NV ? (r & 1) : (r >> 3)) ? 0x03020302u : 0x01000100u;
      // fp16: the sum can leave the type's range (24 * 128 * |a|), so the value travels as z / 16 beside 16 * s
      // (both exact); bf16: high half by truncation
      uint4 zb[NPASS][MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        uint32_t d[NPASS][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int gi = gq + 4 * g + e;
          float z = gi < GW ? zs[gi * MP + t * 16 + r] : 0.f;
          if constexpr (IS_F16) {
            z *= 0.0625f;
            const uint16_t zh = T::from_float(z);
            d[0][e] = (uint32_t)zh | ((uint32_t)T::from_float(z - T::to_float(zh)) << 16);
          } else {
            const uint32_t zh = __float_as_uint(z) & 0xffff0000u;
            const float rem = z - __uint_as_float(zh);
            const uint16_t zm = T::from_float(rem);
            d[0][e] = (zh >> 16) | ((uint32_t)zm << 16);
            if constexpr (NPASS == 2) d[1][e] = (uint32_t)T::from_float(rem - T::to_float(zm));
          }
        }
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) zb[ps][t] = make_uint4(d[ps][0], d[ps][1], d[ps][2], d[ps][3]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t sd[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sd[e] = __builtin_amdgcn_perm(sraw[e][j], sraw[e][j], hsel);   // the column's scale in both halves
          if constexpr (IS_F16) {
            const uint16_t s16 = T::from_float(16.0f * T::to_float((uint16_t)(sd[e] & 0xffffu)));
            sd[e] = (uint32_t)s16 | ((uint32_t)s16 << 16);
          }
        }
        const uint4 sa = make_uint4(sd[0], sd[1], sd[2], sd[3]);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
          for (int t = 0; t < MT; ++t) accm[j][t] = W4<T>::mfma(sa, zb[ps][t], accm[j][t]);
      }
    }
  }
  W4S_STAMP(4);
  // ---- k groups meet in LDS (fixed order: bit-reproducible) ----
  if constexpr (P > 1) {
    float* red = reinterpret_cast<float*>(smem);
    __syncthreads();   // nobody reads the operand images any more
    if (kp > 0) {
      float* dst = red + (int64_t)((kp - 1) * CPW + c) * (4 * MT * 4) * 64;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) dst[((j * MT + t) * 4 + reg) * 64 + lane] = accm[j][t][reg];
    }
    __syncthreads();
    if (kp == 0) {
#pragma unroll
      for (int kk = 1; kk < P; ++kk) {
        const float* src = red + (int64_t)((kk - 1) * CPW + c) * (4 * MT * 4) * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) accm[j][t][reg] += src[((j * MT + t) * 4 + reg) * 64 + lane];
      }
    }
  }
  W4S_STAMP(5);
  const bool writer = (kp == 0) && chunk_ok;

  // ---- epilogue (the tall kernel's forms): lane (r, g) holds columns chunk*64 + 16 j + 4 g + reg of rows m0 + 16 t + r ----
  if (p.splits == 1 && p.epi != 2) {
    if (!writer) return;
    if (p.epi) {
      // silu(gate) * up on column-interleaved gate_up weights (w4a16_gemm.hip, w4_tall_epilogue): same roundings as
      // the two ops it replaces (activation_kernels.cu:14-26)
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
        const f32x4_t o = accm[j][t];
        uint2 pk;
        pk.x = T::pack2(o[0], o[1]);
        pk.y = T::pack2(o[2], o[3]);
        *reinterpret_cast<uint2*>(p.c + (int64_t)m * p.N + chunk * 64 + j * 16 + 4 * g) = pk;
      }
    }
    return;
  }
  // split-K across workgroups: write-through fp32 slabs, ticket, the last workgroup of the tile sums them in
  // split order (w4a16_common.h); deferred mode leaves the slabs to the next launch of the layer
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
  if (p.epi == 2) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int* ticket_s = reinterpret_cast<int*>(smem) + (NTHR * 4);   // behind the NTHR float4 of the reduction
  __syncthreads();
  const int tile = blockIdx.z * gridDim.x + blockIdx.x;
  if (tid == 0)
    *ticket_s = __hip_atomic_fetch_add(p.tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*ticket_s != p.splits - 1) return;
  if (tid == 0) __hip_atomic_store(p.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  splitk_reduce_tile<T, NTHR>(p, rs, m0, MP, blockIdx.x * (CPW * 64), CPW * 64, reinterpret_cast<f32x4_t*>(smem));
}

// ---------------------------------------------------------------------------------------------------------
// Host side: plan and launch.  Returns false when the shape is outside the kernel's domain (the caller then
// takes the tall kernel): 4-bit symmetric codes, group 128, no act-order, M <= 64 per row tile.
static int env_i(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// Plan (defaults from tools/sweep_stream.py on MI355X in the forms the decode step issues -- deferred reduction for
// qkv / o / down, silu epilogue for gate_up -- profiles/r03_sweep_stream.txt; NMV_W4S_* override every choice):
//   wide launches (>= 224 column chunks): one k range per workgroup, no split-K.  M <= 16: 8 waves = 2 chunks x
//     4 k groups, all of K resident in LDS, ring 2 groups deep; M <= 32: 32-row tile, activations streamed in 256-k
//     stages; M <= 64: 64-row tile, 128-k stages;
//   narrow launches: split-K to about 256 workgroups, activations resident.  M <= 16: 4 waves = 2 chunks x 2 k
//     groups (several workgroups per CU overlap each other's prologue and epilogue); M > 16: 32-row tiles (two row
//     blocks at M = 64: the second reads the weights from L2), 8 waves = 2 (M <= 32) or 4 chunks per workgroup.
bool w4s_make_plan(int M, int N, int K, int64_t tickets_len, bool unsplit, bool deferred, W4StreamPlan* out) {
  if (!env_i("NMV_W4S", 1)) return false;
  if (M <= 0 || M > env_i("NMV_W4S_MAX_M", 64) || N % 64 != 0 || K % 128 != 0) return false;
  W4StreamPlan pl;
  const int n_chunks = N / 64, groups = K / 128;
  const bool wide = n_chunks >= 224;
  // a self-contained narrow launch pays for every split in its own last-arriver pass: 16-row tiles (row blocks
  // instead of split-K: the later ones read the weights from L2), one chunk x 8 k groups per workgroup, and only
  // as many splits as LDS demands (K <= 4096: none); the deferred form leaves the slabs to the next launch
  // (measured -- profiles/r03_sweep_stream.txt -- but OFF by default, NMV_W4S_SOLO=1: a plan that depends on the
  // mode would break the bit-identity of gemm_partial + consumer with the self-contained op)
  const bool solo = !wide && !deferred && K <= 8192 && env_i("NMV_W4S_SOLO", 0);
  // small shards (the qkv / o projections of Llama-3-8B at TP = 4 / 8: <= 2.5 MB of codes) keep the 16-row tile at every
  // M -- row blocks instead of taller tiles: more workgroups, the later blocks read the weights from L2 (tools/sweep_stream.py
  // --native --shapes qkv_tp8,o_tp8,o_tp4 at M = 64: 5.4 / 4.7 / 5.5 us against 6.9 / 6.7 / 7.0 with the 32-row tile)
  const bool small = !wide && (int64_t)K * N / 2 <= (5 << 19);
  const int mt = env_i("NMV_W4S_MT", (M <= 16 || solo || small) ? 1 : (M <= 32 || !wide) ? 2 : 4);
  if (mt != 1 && mt != 2 && mt != 4) return false;
  pl.mt = mt;
  const int mp = 16 * mt;
  pl.m_blocks = (M + mp - 1) / mp;
  pl.nw = env_i("NMV_W4S_NW", (!wide && mt == 1 && !solo) ? 4 : 8);
  const int gst = env_i("NMV_W4S_GST", !wide ? 0 : mt == 4 ? 1 : mt == 2 ? 2 : 0);
  // two row blocks of 32 (M = 33..64) on a narrow projection: four chunks per workgroup halve the activation staging, but
  // need twice the split-K slabs to reach 256 workgroups; o_proj-sized launches (< 96 chunks, K <= 8192) are as fast
  // with two chunks and half the slabs (profiles/r03_sweep_stream.txt: 9.2 vs 9.3 us), and their consumer reads half
  // (two chunks only where K then splits evenly into 4 slices of whole k-group sets: K % 2048 == 0 -- at K = 3584, the
  // TP = 4 shard of down_proj, the only other split is 7 and the launch goes from 8.8 to 12.3 us)
  const bool two_chunks_ok = n_chunks < 96 && K <= 8192 && groups % 16 == 0;
  const int cpw = env_i("NMV_W4S_CPW", solo ? 1 : (!wide && mt == 2 && M > 32 && !two_chunks_ok) ? 4 : 2);
  if (cpw != 1 && cpw != 2 && cpw != 4) return false;
  if (pl.nw != 4 && pl.nw != 8 && pl.nw != 16) return false;
  if (pl.nw % cpw != 0) return false;
  const int P = pl.nw / cpw;
  pl.cpw = cpw;
  pl.gst = gst;
  pl.d = env_i("NMV_W4S_D", gst != 0 ? 1 : (wide || solo) ? 2 : 3);
  if (gst != 0 && gst % pl.d != 0) return false;
  pl.n_blocks = (n_chunks + cpw - 1) / cpw;
  const int base_wgs = pl.n_blocks * pl.m_blocks;
  // groups per workgroup: a divisor of the group count, whole k groups (and whole stages when streamed), activations
  // within LDS and within the prologue's staging capacity when resident; among those the split count whose workgroup
  // count is closest to the target (wide launches: no split)
  const int unit = gst == 0 ? P : P * gst;
  const int max_g_wg = gst == 0 ? (128 * 1024) / (mp * 256) : groups;
  const int target = env_i("NMV_W4S_WGS", solo ? 1 : 256);
  const int forced = (unsplit || wide) ? (unsplit ? 1 : env_i("NMV_W4S_SPLITS", 1)) : env_i("NMV_W4S_SPLITS", 0);
  int rows_pad = 1;
  while (rows_pad < std::min(mp, M)) rows_pad *= 2;
  const int nthr = pl.nw * 64;
  int best_splits = 0, best_dist = INT32_MAX;
  for (int splits = 1; splits <= groups; ++splits) {
    if (groups % splits != 0) continue;
    const int g_wg = groups / splits;
    if (g_wg % unit != 0 || g_wg > max_g_wg) continue;
    if (gst == 0 && g_wg * rows_pad * 4 > std::min(4, max_g_wg * mp * 4 / nthr) * nthr) continue;   // staging: <= UB units per thread
    if (splits > 1 && (int64_t)base_wgs > tickets_len) break;
    if (forced) {
      if (splits == forced) { best_splits = splits; break; }
      continue;
    }
    // past 16 rows the slabs weigh in (splits * M * N * 4 bytes each way): no more than 8 of them
    if (M > 16 && splits > 8 && best_splits != 0) break;
    const int dist = std::abs(base_wgs * splits - target);
    if (dist < best_dist) { best_dist = dist; best_splits = splits; }
  }
  if (best_splits == 0) return false;
  // two 32-row blocks whose k slices cannot fill the chip (Llama-3-70B at TP = 8: qkv 8192 x 1280, down 3584 x 8192 --
  // 160 / 128 workgroups): the tall kernel's 64-column workgroups do better there (7.6 vs 8.4 us, 11.4 vs 14.3 us at M = 64)
  if (!forced && !wide && mt == 2 && M > 32 && base_wgs * best_splits < 192 && !env_i("NMV_W4S_STRICT", 0)) return false;
  pl.splits = best_splits;
  const int g_wg = groups / best_splits;
  pl.k_per_wg = g_wg * 128;
  if (gst == 0) {
    pl.g_stage = g_wg / P;
    pl.n_stages = 1;
  } else {
    pl.g_stage = gst;
    pl.n_stages = g_wg / (P * gst);
  }
  const int nbuf = gst == 0 ? 1 : 2;
  const int ngrp = P * pl.g_stage;
  const int64_t main_b = (int64_t)nbuf * ngrp * (16 * mp * 16 + cpw * 256) + (int64_t)P * (pl.g_stage * pl.n_stages) * mp * 4;
  const int64_t red_b = std::max<int64_t>((int64_t)(P - 1) * cpw * mt * 4096, (int64_t)pl.nw * 64 * 16 + 16);
  pl.lds_bytes = (int)std::max(main_b, red_b);
  if (pl.lds_bytes > 160 * 1024) return false;
  *out = pl;
  return true;
}

template <typename T, int MT, int NW, int CPW, int D, int GST, bool NV>
static int w4s_launch_one(const W4StreamPlan& pl, const GemmParams& p, hipStream_t s) {
  auto kern = w4a16_stream_kernel<T, MT, NW, CPW, D, GST, NV>;
  // > 64 KiB of dynamic LDS needs an opt-in that holds for the device current at the call: tracked per device ordinal
  // (w4a16_ring.hip: lds_optin_needed), so the first launch on a second GPU of the process opts in too
  static unsigned long long optin = 0;   // per instantiation
  if (pl.lds_bytes > 64 * 1024) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -2;
    if (lds_optin_needed(&optin, dev) &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return -2;
  }
  dim3 grid(pl.n_blocks, pl.splits, pl.m_blocks), block(NW * 64);
  hipLaunchKernelGGL(kern, grid, block, pl.lds_bytes, s, p);
  return 0;
}

template <typename T, bool NV>
static int w4s_launch_t(const W4StreamPlan& pl, const GemmParams& p, hipStream_t s) {
#define NMV_W4S_CASE_(mt_, nw_, cpw_, d_, gst_)                                                  \
  if (pl.mt == mt_ && pl.nw == nw_ && pl.cpw == cpw_ && pl.d == d_ && pl.gst == gst_)           \
    return w4s_launch_one<T, mt_, nw_, cpw_, d_, gst_, NV>(pl, p, s);
#define NMV_W4S_CASE(...) NMV_W4S_CASE_(__VA_ARGS__)
#ifdef NMV_W4S_PROBE_CASE   // development: one instantiation only (tools/kernel_resources.py ... -DNMV_W4S_PROBE_CASE=4,8,2,1,1)
  NMV_W4S_CASE(NMV_W4S_PROBE_CASE)
#else
  if constexpr (NV) {   // the native tensor: the plan's defaults and what tools/sweep_stream.py visits around them
    NMV_W4S_CASE(1, 8, 2, 2, 0) NMV_W4S_CASE(1, 8, 2, 3, 0) NMV_W4S_CASE(1, 4, 2, 3, 0) NMV_W4S_CASE(1, 4, 2, 2, 0)
    NMV_W4S_CASE(2, 8, 2, 3, 0) NMV_W4S_CASE(2, 8, 4, 3, 0) NMV_W4S_CASE(2, 8, 2, 2, 0)
    NMV_W4S_CASE(4, 8, 2, 1, 1) NMV_W4S_CASE(4, 8, 4, 1, 1) NMV_W4S_CASE(2, 8, 2, 2, 2) NMV_W4S_CASE(2, 8, 2, 1, 2)
  } else {
    // resident
    NMV_W4S_CASE(1, 8, 2, 2, 0) NMV_W4S_CASE(1, 8, 2, 3, 0) NMV_W4S_CASE(1, 8, 1, 2, 0) NMV_W4S_CASE(1, 8, 1, 3, 0)
    NMV_W4S_CASE(1, 8, 4, 3, 0)
    NMV_W4S_CASE(1, 4, 1, 3, 0) NMV_W4S_CASE(1, 4, 2, 3, 0) NMV_W4S_CASE(1, 4, 2, 2, 0)
    NMV_W4S_CASE(1, 16, 2, 2, 0) NMV_W4S_CASE(1, 16, 4, 2, 0)
    NMV_W4S_CASE(2, 8, 1, 3, 0) NMV_W4S_CASE(2, 8, 2, 3, 0) NMV_W4S_CASE(2, 8, 4, 3, 0) NMV_W4S_CASE(2, 8, 2, 2, 0)
    // streamed
    NMV_W4S_CASE(4, 8, 2, 1, 1) NMV_W4S_CASE(4, 8, 4, 1, 1)
    NMV_W4S_CASE(2, 8, 2, 2, 2) NMV_W4S_CASE(2, 8, 2, 1, 2) NMV_W4S_CASE(2, 8, 2, 1, 1)
  }
#endif
#undef NMV_W4S_CASE
#undef NMV_W4S_CASE_
  return -1;
}

#ifdef NMV_W4S_STAMPS
extern "C" int nmv_dbg_w4s_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_w4s_stamps), (size_t)n * 8);
}
#endif

int w4s_launch(const W4StreamPlan& pl, const GemmParams& p, bool f16, hipStream_t s) {
  if (p.native) return f16 ? w4s_launch_t<F16, true>(pl, p, s) : w4s_launch_t<BF16, true>(pl, p, s);
  return f16 ? w4s_launch_t<F16, false>(pl, p, s) : w4s_launch_t<BF16, false>(pl, p, s);
}

}  // namespace nmv
