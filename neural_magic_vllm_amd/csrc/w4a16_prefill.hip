// W4A16 GEMM for prompt-sized calls (M >= 65 rows) on the MFMA-native tensor: the compute-bound end of the operator
// `gptq_marlin_gemm` (reference csrc/quantization/gptq_marlin/gptq_marlin.cu:396-1363; its thread-block tiles for
// prefill :1577-1731), written for gfx950 from the machine's side:
//
//   * a workgroup owns a 256-row x 128-column tile of C and walks K in 64-k stages; 8 waves (2 per SIMD), each a
//     64 x 64 sub-tile = 2 x 2 `v_mfma_f32_32x32x16` tiles (64 accumulator registers), 16 MFMAs per stage and wave
//     against 16 `ds_read_b128` -- one LDS read per MFMA, half of what the LDS array sustains beside the matrix pipe;
//   * the CODES ARE EXPANDED ONCE PER WORKGROUP, not once per wave: waves 0..3 read a stage's 4 KiB of codes (LDS-DMA
//     ring, 6 stages ahead: the only HBM stream), turn them into the model dtype WITH the group scale applied --
//     w = round((q - 8) * s), the reference's own dequantisation (gptq_marlin.cu:267-278: exact (q - 8), then one
//     rounded multiply): here cvt_f32_ubyte, one exact fp32 fma, one RNE pack -- and write a [column][k] image that
//     every wave reads as plain MFMA B operands.  23 VALU instructions per 8 weights, amortised over 256 rows: about
//     3 per MFMA, which fit the matrix pipe's shadow.  One accumulator set, no row sums, no fp32 group pass;
//   * waves 4..7 move the activations: 32 LDS-DMA pieces (8 rows x 128 B, the lane's SOURCE address carries the XOR
//     swizzle) per stage, two stages ahead, into a 3-slot ring.  A wave issues ONE stream only, because `s_waitcnt vmcnt`
//     counts in issue order: the codes come from HBM, the activations from L2, and neither may wait for the other;
//   * one workgroup barrier per stage publishes the expanded tile and the landed pieces; no vmcnt(0) anywhere in the loop.
//
// LDS: A 3 x 32 KiB, expanded B 2 x 16 KiB, codes 6 x 4 KiB, scales 6 x 256 B = 153.5 KiB -- one workgroup per CU.
// Both images are [row or column][8 slots of 16 B] with slot = chunk ^ ((row >> 1) & 7): conflict-free for the
// ds_read_b128 of a 32x32x16 operand (lanes 0..31 = rows, lanes 32..63 the next 16-byte chunk) and for the expansion's
// ds_write_b128.
//
// Modes (GemmParams::epi): 0 -> c[M, N]; 1 -> silu(gate) * up on column-interleaved gate_up weights, c[M, N/2] (gate
// and up of one output sit in the same lane: tiles j = 0 / 1 of the wave's chunk); 2 -> fp32 slabs only.  splits > 1:
// fp32 slabs, ticket, the last workgroup of the tile sums them in split order (w4a16_common.h).
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "w4a16_common.h"

namespace nmv {

#if defined(NMV_W4P_STAMPS)
// development (-DNMV_W4P_STAMPS): every wave of the first 256 workgroups of row block 0 sums the shader clocks it spends in
// each phase of the main loop: 0 DMA issue, 1 multiply (operand reads + MFMA), 2 expansion, 3 vmcnt wait, 4 barrier;
// 7 = the whole loop; slot 8 = 100 MHz clock over the loop
__device__ unsigned long long g_w4p_stamps[256 * 8 * 16];
#define W4P_T0() unsigned long long w4p_t = __builtin_amdgcn_s_memtime(), w4p_t00 = w4p_t, w4p_r0 = __builtin_amdgcn_s_memrealtime(); \
                 unsigned long long w4p_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define W4P_ACC(k)                                                 \
  do {                                                             \
    const unsigned long long n_ = __builtin_amdgcn_s_memtime();    \
    w4p_sum[k] += n_ - w4p_t;                                      \
    w4p_t = n_;                                                    \
  } while (0)
#define W4P_FLUSH()                                                                                       \
  do {                                                                                                    \
    w4p_sum[7] = __builtin_amdgcn_s_memtime() - w4p_t00;                                                   \
    if (lane == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 256) {                            \
      for (int k_ = 0; k_ < 8; ++k_) g_w4p_stamps[(blockIdx.x * 8 + wave) * 16 + k_] = w4p_sum[k_];        \
      g_w4p_stamps[(blockIdx.x * 8 + wave) * 16 + 8] = __builtin_amdgcn_s_memrealtime() - w4p_r0;          \
    }                                                                                                     \
  } while (0)
#else
#define W4P_T0()
#define W4P_ACC(k)
#define W4P_FLUSH()
#endif

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

constexpr int PF_BM = 256, PF_BN = 128, PF_BK = 64;
constexpr int PF_NTHR = 512;
constexpr int PF_NA = 3, PF_NB = 2, PF_NC = 6;           // ring depths: activations, expanded weights, codes (+ scales)
constexpr int PF_A_BYTES = PF_BM * 128, PF_B_BYTES = PF_BN * 128, PF_C_BYTES = 4096, PF_S_BYTES = 256;
constexpr int PF_A_OFF = 0;
constexpr int PF_B_OFF = PF_A_OFF + PF_NA * PF_A_BYTES;
constexpr int PF_C_OFF = PF_B_OFF + PF_NB * PF_B_BYTES;
constexpr int PF_S_OFF = PF_C_OFF + PF_NC * PF_C_BYTES;
constexpr int PF_T_OFF = PF_S_OFF + PF_NC * PF_S_BYTES;  // ticket word
constexpr int PF_LDS = PF_T_OFF + 64;
constexpr uint32_t PF_OOB = 0x7ffffff0u;                 // a voffset no buffer of ours reaches: zeros, no request
static_assert(PF_LDS <= 160 * 1024, "LDS");
static_assert(PF_NTHR * 16 <= PF_B_OFF, "the split-K reduction's scratch fits the dead activation ring");

template <int N>
__device__ __forceinline__ void pf_wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One LDS-DMA piece: 64 lanes x 16 bytes from buffer `rs` (per-lane byte offset `voff`, bounds-checked; uniform `soff`)
// to LDS bytes [lds_addr, lds_addr + 1024) in lane order; the caller waits with pf_wait_vm<>.
__device__ __forceinline__ void pf_dma16(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// the workgroup barrier of the main loop: LDS writes of this wave done, nothing said about VMEM (a __syncthreads() would
// drain the DMA rings with vmcnt(0))
__device__ __forceinline__ void pf_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <typename T>
__device__ __forceinline__ f32x16_t pf_mfma(u32x4_t a, u32x4_t b, f32x16_t c) {
  if constexpr (std::is_same<T, F16>::value)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// 8 codes of one column (native dword: nibble p = k 2p, nibble p + 4 = k 2p + 1) -> 8 weights round((q - 8) * s) in k order
template <typename T>
__device__ __forceinline__ u32x4_t pf_expand(uint32_t x, float s, float c) {
  const uint32_t e = x & 0x0f0f0f0fu;          // bytes: k0, k4, k1, k5
  const uint32_t o = (x >> 4) & 0x0f0f0f0fu;   // bytes: k2, k6, k3, k7
  const float k0 = (float)(e & 0xffu), k4 = (float)((e >> 8) & 0xffu), k1 = (float)((e >> 16) & 0xffu), k5 = (float)(e >> 24);
  const float k2 = (float)(o & 0xffu), k6 = (float)((o >> 8) & 0xffu), k3 = (float)((o >> 16) & 0xffu), k7 = (float)(o >> 24);
  u32x4_t w;
  w[0] = T::pack2(__builtin_fmaf(k0, s, c), __builtin_fmaf(k1, s, c));
  w[1] = T::pack2(__builtin_fmaf(k2, s, c), __builtin_fmaf(k3, s, c));
  w[2] = T::pack2(__builtin_fmaf(k4, s, c), __builtin_fmaf(k5, s, c));
  w[3] = T::pack2(__builtin_fmaf(k6, s, c), __builtin_fmaf(k7, s, c));
  return w;
}

}  // namespace

// grid (ceil(N / 128), splits, ceil(M / 256)), 512 threads.
// p.b: native[kstep][chunk][lane] (uint4), p.s: natural [groups, N]; p.k_per_wg: a multiple of 128.
template <typename T>
__global__ __launch_bounds__(PF_NTHR, 2) void w4a16_prefill_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // development switches (compile time, -DNMV_W4P_ABL=bits: results garbage, times valid): 1 no activation DMA, 2 no code
  // DMA, 4 no expansion, 8 no operand reads, 16 no MFMA
#ifndef NMV_W4P_ABL
#define NMV_W4P_ABL 0
#endif
  constexpr int abl = NMV_W4P_ABL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_chunks = p.N >> 6;
  const int chunk0 = blockIdx.x * 2;
  const int split = blockIdx.y;
  const int m0 = blockIdx.z * PF_BM;
  const int k_wg0 = split * p.k_per_wg;
  const int T_ST = min(p.k_per_wg, p.K - k_wg0) >> 6;     // 64-k stages of this workgroup (uniform, even)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // consumer identity: 64-row block wm, 64-column block (= chunk of the strip) wn; lane = (row / column n5, k half kh)
  const int wm = wave & 3, wn = wave >> 2;
  const int n5 = lane & 31, kh = lane >> 5;
  // operand addresses inside a stage image: row * 128 + ((2 ks + kh) ^ ((row >> 1) & 7)) * 16 = base ^ (ks << 5)
  const uint32_t frag_sw = (uint32_t)((kh ^ ((n5 >> 1) & 7)) << 4);
  const uint32_t a_frag = (uint32_t)((64 * wm + n5) * 128) + frag_sw;      // + 4096 for the second 32-row tile
  const uint32_t b_frag = (uint32_t)((64 * wn + n5) * 128) + frag_sw;      // + 4096 for the second 32-column tile

  // ---- DMA roles ----
  // waves 4..7: activations.  Piece q = (wave - 4) + 4 u (u = 0..7) = rows 8 q .. 8 q + 7 of the tile; lane = (row i >> 3,
  // slot i & 7), source chunk = slot ^ ((row >> 1) & 7) with row = 8 q + (i >> 3): (row >> 1) & 7 = (4 (q & 1) + (i >> 4)) & 7,
  // the same for every u (q + 4 keeps q & 1... 32 rows further: (row >> 1) & 7 unchanged)
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.a), 0, (int)((int64_t)p.M * p.K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint4*>(p.b), 0, (int)(((int64_t)p.K * p.N) >> 1), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.s), 0, (int)((int64_t)(p.K >> 7) * p.N * 2), 0x00020000);
  const int aq = wave & 3;
  const int a_row = 8 * aq + (lane >> 3);
  const uint32_t a_chunk = (uint32_t)((lane & 7) ^ ((a_row >> 1) & 7));
  // rows past M: the buffer's bounds return zeros only past the END of the matrix, so clamp them out explicitly
  uint32_t a_voff[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int row = m0 + a_row + 32 * u;
    a_voff[u] = row < p.M ? (uint32_t)(((int64_t)row * p.K + k_wg0) * 2) + a_chunk * 16 : PF_OOB;
  }
  const uint32_t a_lds = lds0 + PF_A_OFF + (uint32_t)(aq * 1024);       // + slot * 32 KiB + u * 4 KiB

  // waves 0..3: the codes.  Wave w moves (and later expands) native row (k-step kk = w >> 1 of the stage, chunk ch = w & 1)
  const int kk = (wave >> 1) & 1, ch = wave & 1;
  const bool chunk_ok = chunk0 + ch < n_chunks;
  const uint32_t c_voff = chunk_ok ? (uint32_t)((chunk0 + ch) * 1024 + lane * 16) : PF_OOB;
  const uint32_t c_lds = lds0 + PF_C_OFF + (uint32_t)((wave & 3) * 1024);
  // wave 0 also the stage's scale row (128 columns x 2 bytes: lanes 0..15), half a group per stage: loaded per stage
  const uint32_t s_voff = (lane < 16 && chunk0 * 64 + lane * 8 < p.N) ? (uint32_t)((chunk0 * 64 + lane * 8) * 2) : PF_OOB;
  const uint32_t s_lds = lds0 + PF_S_OFF;

  auto issue_a = [&](int t) {   // waves 4..7: stage t of the activations (8 pieces)
    if constexpr (abl & 1) return;
    const bool live = t < T_ST;
    const uint32_t slot = a_lds + (uint32_t)((t % PF_NA) * PF_A_BYTES);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      pf_dma16(rs_a, live ? a_voff[u] : PF_OOB, (uint32_t)(t * (PF_BK * 2)), slot + (uint32_t)(u * 4096));
  };
  auto issue_c = [&](int t) {   // waves 0..3: stage t of the codes (and wave 0: of the scales)
    if constexpr (abl & 2) return;
    const bool live = t < T_ST;
    const int ks = ((k_wg0 >> 5) + 2 * t + kk);
    if (wave == 0 && lane < 16)    // 16 lanes x 16 bytes: the other lanes would zero-fill the slots behind this one
      pf_dma16(rs_s, live ? s_voff : PF_OOB, (uint32_t)(((k_wg0 >> 7) + (t >> 1)) * p.N * 2),
               s_lds + (uint32_t)((t % PF_NC) * PF_S_BYTES));
    pf_dma16(rs_b, live ? c_voff : PF_OOB, (uint32_t)(ks * n_chunks * 1024), c_lds + (uint32_t)((t % PF_NC) * PF_C_BYTES));
  };

  // expansion (waves 0..3): lane = (r = column within 16, g = k octet of the 32-k step); dword j = column 64 ch + 16 j + r
  const int er = lane & 15, eg = lane >> 4;
  const uint32_t e_dst = (uint32_t)((64 * ch + er) * 128 + (((4 * kk + eg) ^ ((er >> 1) & 7)) << 4));   // + j * 2048
  auto expand_stage = [&](int t) {   // codes of stage t -> expanded image slot t % PF_NB
    if constexpr (abl & 4) return;
    const unsigned char* cs = smem + PF_C_OFF + (t % PF_NC) * PF_C_BYTES + (wave & 3) * 1024 + lane * 16;
    const uint4 x = *reinterpret_cast<const uint4*>(cs);
    const uint16_t* sc = reinterpret_cast<const uint16_t*>(smem + PF_S_OFF + (t % PF_NC) * PF_S_BYTES) + 64 * ch + er;
    unsigned char* dst = smem + PF_B_OFF + (t % PF_NB) * PF_B_BYTES + e_dst;
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float s = T::to_float(sc[16 * j]);
      const u32x4_t w = pf_expand<T>(xs[j], s, -8.0f * s);
      *reinterpret_cast<u32x4_t*>(dst + j * 2048) = w;
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: codes of stages 0 .. PF_NC - 1, activations of stages 0, 1; stage 0 expanded ----
  if (wave < 4) {
#pragma unroll
    for (int t = 0; t < PF_NC; ++t) issue_c(t);
    // stage 0 and 1 have landed when at most the pieces of the 4 younger stages are in flight
    if (wave == 0) pf_wait_vm<2 * (PF_NC - 2)>(); else pf_wait_vm<PF_NC - 2>();
  } else {
    issue_a(0);
    issue_a(1);
    pf_wait_vm<8>();
  }
  __builtin_amdgcn_s_barrier();        // codes of stages 0, 1 and activations of stage 0 are in LDS
  if (wave < 4) expand_stage(0);
  pf_barrier();

  // ---- main loop: iteration t multiplies stage t, expands stage t + 1, issues A(t + 2) and codes(t + PF_NC) ----
  // The two waves of a SIMD (w and w + 4) run OUT OF PHASE inside a stage: wave w + 4 issues its 8 DMA pieces first and
  // multiplies afterwards, wave w multiplies first and expands afterwards -- each one's MFMAs run in the shadow of the
  // other's VALU / issue work (both doing their side work first puts all 32 MFMAs of the SIMD behind it).
  auto ld_frags = [&](uint32_t a_base, uint32_t b_base, int ks, u32x4_t* af, u32x4_t* bf) {
    if constexpr (abl & 8) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = u32x4_t{a_base, b_base, (uint32_t)ks, (uint32_t)i};
        bf[i] = u32x4_t{b_base, a_base, (uint32_t)i, (uint32_t)ks};
        asm volatile("" : "+v"(af[i]), "+v"(bf[i]));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const u32x4_t*>(smem + ((a_base ^ (uint32_t)(ks << 5)) + i * 4096));
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const u32x4_t*>(smem + ((b_base ^ (uint32_t)(ks << 5)) + j * 4096));
    }
  };
  auto multiply_stage = [&](int t) {
    const uint32_t a_base = (uint32_t)(PF_A_OFF + (t % PF_NA) * PF_A_BYTES) + a_frag;
    const uint32_t b_base = (uint32_t)(PF_B_OFF + (t % PF_NB) * PF_B_BYTES) + b_frag;
    u32x4_t af[2][2], bf[2][2];
    ld_frags(a_base, b_base, 0, af[0], bf[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) ld_frags(a_base, b_base, ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
      if constexpr (abl & 16) {
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" :: "v"(af[ks & 1][i]), "v"(bf[ks & 1][i]));
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = pf_mfma<T>(af[ks & 1][i], bf[ks & 1][j], acc[i][j]);
      }
    }
  };
  W4P_T0();
  for (int t = 0; t < T_ST; ++t) {
    if (wave >= 4) {
      issue_a(t + 2);
      W4P_ACC(0);
      multiply_stage(t);
      W4P_ACC(1);
      // what the NEXT iteration reads has landed: activations of stage t + 1 (issued one iteration ago: the 8 pieces of
      // this iteration may stay in flight)
      pf_wait_vm<8>();
      W4P_ACC(3);
    } else {
      issue_c(t + PF_NC);
      W4P_ACC(0);
      multiply_stage(t);
      W4P_ACC(1);
      if (t + 1 < T_ST) expand_stage(t + 1);
      W4P_ACC(2);
      // codes and scales of stage t + 2 (issued PF_NC - 2 iterations ago) have landed
      if (wave == 0) pf_wait_vm<2 * (PF_NC - 2)>(); else pf_wait_vm<PF_NC - 2>();
      W4P_ACC(3);
    }
    pf_barrier();
    W4P_ACC(4);
  }
  W4P_FLUSH();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail's zero-filled pieces

  // ---- epilogue.  D[m][n] of a 32x32 tile: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
  const int n_wave = chunk0 * 64 + 64 * wn;            // first column of the wave's chunk
  const int row_l = m0 + 64 * wm + 4 * kh;
  if (p.splits == 1 && p.epi != 2) {
    if (p.epi) {
      // silu(gate) * up on column-interleaved gate_up weights (chunk = [gate 32 | up 32]); roundings of the two ops it
      // replaces (reference activation_kernels.cu:14-26): gate and up rounded to the model dtype, silu rounded, product rounded
      if (n_wave >= p.N) return;
      uint16_t* cp = p.c + (n_wave >> 1) + n5;
      const int ldc = p.N >> 1;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_l + 32 * i + (r & 3) + 8 * (r >> 2);
          if (row >= p.M) continue;
          const float gb = round_trip<T>(acc[i][0][r]), ub = round_trip<T>(acc[i][1][r]);
          cp[(int64_t)row * ldc] = T::from_float(round_trip<T>(gb / (1.0f + expf(-gb))) * ub);
        }
      return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n_wave + 32 * j + n5;
      if (col >= p.N) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_l + 32 * i + (r & 3) + 8 * (r >> 2);
          if (row < p.M) p.c[(int64_t)row * p.N + col] = T::from_float(acc[i][j][r]);
        }
    }
    return;
  }
  // split-K across workgroups: write-through fp32 slabs, ticket, the last workgroup of the tile sums them in split order
  // (w4a16_common.h); deferred mode (epi 2) leaves the slabs to the consumer
  const int64_t slab_bytes = (int64_t)p.splits * p.M * p.N * 4;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.slab, 0, (int)slab_bytes, 0x00020000);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n_wave + 32 * j + n5;
    if (col >= p.N) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_l + 32 * i + (r & 3) + 8 * (r >> 2);
        if (row >= p.M) continue;
        const int off = (int)((((int64_t)split * p.M + row) * p.N + col) * 4);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r]), rs, off, 0, 16);
      }
  }
  if (p.epi == 2) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int* ticket_s = reinterpret_cast<int*>(smem + PF_T_OFF);
  __syncthreads();
  const int tile = blockIdx.z * gridDim.x + blockIdx.x;
  if (tid == 0)
    *ticket_s = __hip_atomic_fetch_add(p.tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*ticket_s != p.splits - 1) return;
  if (tid == 0) __hip_atomic_store(p.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  splitk_reduce_tile<T, PF_NTHR>(p, rs, m0, PF_BM, chunk0 * 64, PF_BN, reinterpret_cast<f32x4_t*>(smem));
}

// ---------------------------------------------------------------------------------------------------------
// Host side.
static int env_p(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// Plan: 256 x 128 tiles; split-K (whole 128-k groups per slice) where the tiles alone leave most of the 256 CUs idle.
// A slice costs M * N * 8 bytes of slab traffic and a prologue: the estimate below weighs rounds of workgroups x k range
// against both.
bool w4p_make_plan(int M, int N, int K, int64_t tickets_len, bool unsplit, W4PrefillPlan* out) {
  if (!env_p("NMV_W4P", 1)) return false;
  if (M < env_p("NMV_W4P_MIN_M", 65) || N % 64 != 0 || K % 128 != 0) return false;
  if ((int64_t)M * K * 2 >= ((int64_t)1 << 31) || (((int64_t)K * N) >> 1) >= ((int64_t)1 << 31)) return false;
  W4PrefillPlan pl;
  pl.m_blocks = (M + PF_BM - 1) / PF_BM;
  pl.n_blocks = (N + PF_BN - 1) / PF_BN;
  const int groups = K / 128;
  const int64_t tiles = (int64_t)pl.m_blocks * pl.n_blocks;
  const int forced = unsplit ? 0 : env_p("NMV_W4P_SPLITS", 0);
  const int max_splits = unsplit ? 1 : env_p("NMV_W4P_MAX_SPLITS", 8);
  int best = 0;
  double best_cost = 1e30;
  for (int s = 1; s <= groups && s <= max_splits; ++s) {
    if (groups % s != 0) continue;
    if (s > 1 && tiles > tickets_len) break;
    if (forced) {
      if (s == forced) { best = s; break; }
      continue;
    }
    const int64_t rounds = (tiles * s + 255) / 256;
    // k-steps of 64 per round, + 6 stages' worth of prologue / epilogue; slabs: 8 bytes per element and slice at 3 TB/s
    // against ~0.45 us per stage
    const double cost = (double)rounds * (K / 64 / s + 6) + (s > 1 ? (double)M * N * 8.0 * s / 3e12 / 0.45e-6 : 0.0);
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  if (best == 0) return false;
  pl.splits = best;
  pl.k_per_wg = (groups / best) * 128;
  pl.lds_bytes = PF_LDS;
  *out = pl;
  return true;
}

template <typename T>
static int w4p_launch_one(const W4PrefillPlan& pl, const GemmParams& p, hipStream_t s) {
  auto kern = w4a16_prefill_kernel<T>;
  static unsigned long long optin = 0;   // per instantiation
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -2;
  if (lds_optin_needed(&optin, dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return -2;
  }
  dim3 grid(pl.n_blocks, pl.splits, pl.m_blocks), block(PF_NTHR);
  hipLaunchKernelGGL(kern, grid, block, pl.lds_bytes, s, p);
  return 0;
}

int w4p_launch(const W4PrefillPlan& pl, const GemmParams& p, bool f16, hipStream_t s) {
  if (!p.native) return -1;
  return f16 ? w4p_launch_one<F16>(pl, p, s) : w4p_launch_one<BF16>(pl, p, s);
}

}  // namespace nmv

#if defined(NMV_W4P_STAMPS)
extern "C" int w4p_dbg_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(nmv::g_w4p_stamps), (size_t)n * 8);
}
#endif
