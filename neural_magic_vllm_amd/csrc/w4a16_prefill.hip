// W4A16 GEMM for prompt-sized calls (M >= 65 rows) on the MFMA-native tensor: the compute-bound end of the operator
// `gptq_marlin_gemm` (reference csrc/quantization/gptq_marlin/gptq_marlin.cu:396-1363; its thread-block tiles for
// prefill :1577-1731), written for gfx950 from the machine's side.  What the machine said (profiles/r04_prefill_*.txt):
// on a SIMD, MFMA time and 4 clocks per VALU instruction ADD UP (no shadow across waves), an LDS-DMA piece costs the CU
// ~30 clocks of its vector-memory path whoever issues it, and `s_waitcnt vmcnt` counts in issue order.  Hence:
//
//   * a workgroup owns a 256-row x 256-column tile of C and walks K in 32-k stages; 8 waves (2 per SIMD), each a
//     64 x 128 sub-tile = 2 x 4 `v_mfma_f32_32x32x16` tiles (128 accumulator registers), 16 MFMAs per stage and wave
//     against 12 `ds_read_b128`.  The square tile halves the bytes per flop of the 256 x 128 one: 20 DMA pieces per
//     stage (16 KiB of activations + 4 KiB of codes) for 32 MFMAs per SIMD;
//   * the CODES ARE EXPANDED ONCE PER WORKGROUP, not once per wave: every wave turns 2 dwords per lane and stage
//     (1/8 of the stage's 4 KiB) into the model dtype WITH the group scale applied -- w = round((q - 8) * s), the
//     reference's own dequantisation (gptq_marlin.cu:267-278: exact (q - 8), then one rounded multiply): here
//     cvt_f32_ubyte, one exact fp32 fma, one RNE pack = 23 VALU instructions per 8 weights -- and writes a [column][k]
//     image that all waves read as plain MFMA B operands.  One accumulator set, no row sums, no fp32 group pass;
//   * all waves are alike (no loader waves: a ninth wave would put three on one SIMD and cap every wave at 168
//     registers): per stage a wave issues 2 activation pieces (16 rows x 64 B, the lane's SOURCE address carries the
//     XOR swizzle), waves 0..3 one piece of codes, wave 4 the scale row of the next group; everything 4-5 stages ahead
//     into LDS rings, one `s_waitcnt vmcnt` per stage ("what was issued three stages ago has landed"), one workgroup
//     barrier per stage; no vmcnt(0) anywhere in the loop.
//
// LDS: A 3 x 32 KiB (64-k double stages, full 128-byte lines), expanded B 2 x 16 KiB, codes 6 x 4 KiB, scales 3 x 512 B = 153.5 KiB.
// Both operand images are [row or column][4 slots of 16 B] with slot = chunk ^ ((row >> 2) & 3): conflict-free for the
// ds_read_b128 of a 32x32x16 operand (lanes 0..31 = rows, lanes 32..63 the next 16-byte chunk) and for the expansion's
// ds_write_b128.
//
// Modes (GemmParams::epi): 0 -> c[M, N]; 1 -> silu(gate) * up on column-interleaved gate_up weights, c[M, N/2] (gate
// and up of one output sit in the same lane: tiles j / j + 1 of a chunk); 2 -> fp32 slabs only.  splits > 1: fp32 slabs,
// ticket, the last workgroup of the tile sums them in split order (w4a16_common.h).
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
    if (lane == 0 && blockIdx.x < 256) {                            \
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

constexpr int PF_BM = 256, PF_BN = 256, PF_BK = 32;
constexpr int PF_NTHR = 512;
constexpr int PF_NA = 3, PF_NB = 2, PF_NC = 6, PF_NS = 3;   // ring depths: activations (64-k double stages), expanded weights, codes, scales
constexpr int PF_DA = 2, PF_DC = 5;                          // ahead: activations (double stages), codes (stages)
constexpr int PF_A_BYTES = PF_BM * 128, PF_B_BYTES = PF_BN * 64, PF_C_BYTES = 4096, PF_S_BYTES = 512;
constexpr int PF_A_OFF = 0;
constexpr int PF_B_OFF = PF_A_OFF + PF_NA * PF_A_BYTES;
constexpr int PF_C_OFF = PF_B_OFF + PF_NB * PF_B_BYTES;
constexpr int PF_S_OFF = PF_C_OFF + PF_NC * PF_C_BYTES;
constexpr int PF_T_OFF = PF_S_OFF + PF_NS * PF_S_BYTES;      // ticket word
constexpr int PF_LDS = PF_T_OFF + 64;
constexpr uint32_t PF_OOB = 0x7ffffff0u;                     // a voffset no buffer of ours reaches: zeros, no request
static_assert(PF_LDS <= 160 * 1024, "LDS");
static_assert(PF_NTHR * 16 <= PF_B_OFF, "the split-K reduction's scratch fits the dead activation ring");
static_assert(PF_NA > PF_DA && PF_NC > PF_DC, "a slot is refilled only after the stage that read it");

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

// one exact fp32 fma that the vectoriser may not pair up (v_pk_fma_f32 costs more than two v_fma_f32 beside MFMAs)
__device__ __forceinline__ float pf_fma(float q, float s, float c) {
  float r = __builtin_fmaf(q, s, c);
  asm volatile("" : "+v"(r));
  return r;
}

// 8 codes of one column (native dword: nibble p = k 2p, nibble p + 4 = k 2p + 1) -> 8 weights round((q - 8) * s) in k order
template <typename T>
__device__ __forceinline__ u32x4_t pf_expand(uint32_t x, float s, float c) {
  uint32_t e = x & 0x0f0f0f0fu;          // bytes: k0, k4, k1, k5
  uint32_t o = (x >> 4) & 0x0f0f0f0fu;   // bytes: k2, k6, k3, k7
  asm volatile("" : "+v"(e), "+v"(o));   // keep the byte form: one v_cvt_f32_ubyteN per code
  const float k0 = (float)(e & 0xffu), k4 = (float)((e >> 8) & 0xffu), k1 = (float)((e >> 16) & 0xffu), k5 = (float)(e >> 24);
  const float k2 = (float)(o & 0xffu), k6 = (float)((o >> 8) & 0xffu), k3 = (float)((o >> 16) & 0xffu), k7 = (float)(o >> 24);
  u32x4_t w;
  w[0] = T::pack2(pf_fma(k0, s, c), pf_fma(k1, s, c));
  w[1] = T::pack2(pf_fma(k2, s, c), pf_fma(k3, s, c));
  w[2] = T::pack2(pf_fma(k4, s, c), pf_fma(k5, s, c));
  w[3] = T::pack2(pf_fma(k6, s, c), pf_fma(k7, s, c));
  return w;
}

}  // namespace

// 1-D grid (w4p_grid() workgroups: row blocks in rounds of G x padded (column block, split) units), 512 threads.
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
  // XCD-aware placement.  Workgroup L of the 1-D grid runs on XCD L % 8 (round-robin dispatch), and each XCD has its own
  // 4 MiB L2: the 256 rows x K of activations a tile re-reads every stage must stay in the L2 of the XCD that runs it.
  // So an XCD works on ONE row block at a time: row blocks are taken G = min(8, largest power of two <= row blocks) at
  // a time, row block r G + (xcd % G) by the 8 / G XCDs with that residue, which deal its (column block, split) units
  // among themselves.  (Column-block-fastest order put 2-3 row blocks = 4-6 MiB of activations through every L2: the
  // activation stream ran at 3.3 TB/s from the Infinity Cache and the kernel waited for it, profiles/r04_prefill_*.txt.)
  const int n_blocks = (n_chunks + 3) >> 2, m_blocks = (p.M + PF_BM - 1) / PF_BM;
  const int G = m_blocks >= 8 ? 8 : m_blocks >= 4 ? 4 : m_blocks >= 2 ? 2 : 1, XQ = 8 / G;
  const int units = n_blocks * p.splits, units_pad = (units + XQ - 1) / XQ * XQ;
  const int L = blockIdx.x, per_round = G * units_pad;
  const int rnd = L / per_round, rem = L - rnd * per_round;
  const int xcd = rem & 7;
  const int unit = (rem >> 3) * XQ + xcd / G;
  const int m_block = rnd * G + xcd % G;
  if (m_block >= m_blocks || unit >= units) return;     // uniform: padding of the last round / of the unit count
  const int n_block = unit % n_blocks, split = unit / n_blocks;
  const int chunk0 = n_block * 4;
  const int m0 = m_block * PF_BM;
  const int k_wg0 = split * p.k_per_wg;
  const int T_ST = min(p.k_per_wg, p.K - k_wg0) >> 5;     // 32-k stages of this workgroup (uniform, a multiple of 4)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // consumer identity: 64-row block wm, 128-column block wn; lane = (row / column n5, k half kh)
  const int wm = wave & 3, wn = wave >> 2;
  const int n5 = lane & 31, kh = lane >> 5;
  // operand addresses.  Expanded weights, one 32-k stage: column * 64 + ((2 ks + kh) ^ ((column >> 2) & 3)) * 16 = base ^ (ks << 5).
  // Activations, one 64-k double stage in FULL 128-byte lines (a CU keeps a limited number of cache lines in flight: a
  // 64-byte piece of a line uses a whole slot): row * 128 + ((4 h + 2 ks + kh) ^ ((row >> 1) & 7)) * 16 =
  // base ^ (ks << 5) ^ (h << 6), h = the 32-k half the stage is.
  const uint32_t a_frag = (uint32_t)((64 * wm + n5) * 128) + (uint32_t)((kh ^ ((n5 >> 1) & 7)) << 4);    // + 4096 for the second 32-row tile
  const uint32_t b_frag = (uint32_t)((128 * wn + n5) * 64) + (uint32_t)((kh ^ ((n5 >> 2) & 3)) << 4);    // + 2048 j for the 32-column tiles

  // ---- DMA roles ----
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.a), 0, (int)((int64_t)p.M * p.K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint4*>(p.b), 0, (int)(((int64_t)p.K * p.N) >> 1), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.s), 0, (int)((int64_t)(p.K >> 7) * p.N * 2), 0x00020000);
  // activations: a double stage (64 k) is 32 pieces of 8 rows x 128 B; wave 4 + v moves rows 64 v .. 64 v + 63 of it, the
  // pieces of rows + 32 h .. + 32 h + 31 in the iteration of half h.  lane = (row i >> 3, slot i & 7); source chunk =
  // slot ^ ((row >> 1) & 7) with (row >> 1) & 7 = (4 (u & 1) + (i >> 4)) & 7 for piece u of the four.
  // Rows past M: the buffer's bounds return zeros only past the END of the matrix, so clamp them out explicitly.
  uint32_t a_voff[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = m0 + 64 * (wave & 3) + 32 * h + 8 * u + (lane >> 3);
      const uint32_t chunk = (uint32_t)((lane & 7) ^ ((4 * (u & 1) + (lane >> 4)) & 7));
      a_voff[h][u] = row < p.M ? (uint32_t)(((int64_t)row * p.K + k_wg0) * 2) + chunk * 16 : PF_OOB;
    }
  const uint32_t a_lds = lds0 + PF_A_OFF + (uint32_t)((wave & 3) * 8192);  // + slot * 32 KiB + h * 4 KiB + u * 1 KiB
  // codes and scales: waves 0..3, wave w the native row (k-step = stage, chunk chunk0 + w) = 1 KiB and the 64 scales of
  // that chunk (8 lanes x 16 bytes per group) -- the wave that loads them is the wave that expands them: its own
  // `s_waitcnt vmcnt` is all the synchronisation the codes need, and the stream from HBM never holds up the activations'
  const bool chunk_ok = chunk0 + (wave & 3) < n_chunks;
  const uint32_t c_voff = chunk_ok ? (uint32_t)((chunk0 + (wave & 3)) * 1024 + lane * 16) : PF_OOB;
  const uint32_t c_lds = lds0 + PF_C_OFF + (uint32_t)((wave & 3) * 1024);
  const uint32_t s_voff = (lane < 8 && chunk_ok) ? (uint32_t)(((chunk0 + (wave & 3)) * 64 + lane * 8) * 2) : PF_OOB;
  const uint32_t s_lds = lds0 + PF_S_OFF + (uint32_t)((wave & 3) * 128);

  auto issue_a = [&](int ta, int h) {   // waves 4..7: half h (rows) of this wave's share of double stage ta: 4 pieces
    if constexpr (abl & 1) return;
    const bool live = 2 * ta < T_ST;
    const uint32_t slot = a_lds + (uint32_t)((ta % PF_NA) * PF_A_BYTES + h * 4096);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      pf_dma16(rs_a, live ? a_voff[h][u] : PF_OOB, (uint32_t)(ta * 128), slot + (uint32_t)(u * 1024));
  };
  auto issue_c = [&](int t) {   // waves 0..3: stage t of the codes
    if constexpr (abl & 2) return;
    pf_dma16(rs_b, t < T_ST ? c_voff : PF_OOB, (uint32_t)(((k_wg0 >> 5) + t) * n_chunks * 1024),
             c_lds + (uint32_t)((t % PF_NC) * PF_C_BYTES));
  };
  auto issue_s = [&](int g) {   // waves 0..3: the chunk's scales of the workgroup's group g (8 lanes x 16 bytes: the other
    if constexpr (abl & 2) return;   // lanes would zero-fill what lies behind)
    if (lane < 8)
      pf_dma16(rs_s, 4 * g < T_ST ? s_voff : PF_OOB, (uint32_t)(((k_wg0 >> 7) + g) * p.N * 2), s_lds + (uint32_t)((g % PF_NS) * PF_S_BYTES));
  };

  // expansion: wave w (0..3) turns native row chunk w -- 4 dwords per lane -- into 4 x 16 bytes of the expanded image;
  // lane = (r = column within 16, g = k octet of the 32-k stage); dword j = column 64 w + 16 j + r
  const int er = lane & 15, eg = lane >> 4;
  const uint32_t e_src = (uint32_t)(PF_C_OFF + (wave & 3) * 1024 + lane * 16);
  const uint32_t e_col = (uint32_t)(64 * (wave & 3) + er);                                           // + 16 j
  const uint32_t e_dst = (uint32_t)(PF_B_OFF) + e_col * 64 + (uint32_t)((eg ^ ((er >> 2) & 3)) << 4);   // + 1024 j
  auto expand_stage = [&](int t) {   // codes of stage t -> expanded image slot t % PF_NB
    if constexpr (abl & 4) return;
    const uint4 x = *reinterpret_cast<const uint4*>(smem + e_src + (t % PF_NC) * PF_C_BYTES);
    const uint16_t* sc = reinterpret_cast<const uint16_t*>(smem + PF_S_OFF + ((t >> 2) % PF_NS) * PF_S_BYTES) + e_col;
    unsigned char* dst = smem + e_dst + (t % PF_NB) * PF_B_BYTES;
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sj = T::to_float(sc[16 * j]);
      *reinterpret_cast<u32x4_t*>(dst + j * 1024) = pf_expand<T>(xs[j], sj, -8.0f * sj);
    }
  };

  f32x16_t acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto ld_frags = [&](uint32_t a_base, uint32_t b_base, int ks, u32x4_t* af, u32x4_t* bf) {
    if constexpr (abl & 8) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = u32x4_t{a_base, b_base, (uint32_t)ks, (uint32_t)i};
        asm volatile("" : "+v"(af[i]));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf[j] = u32x4_t{b_base, a_base, (uint32_t)j, (uint32_t)ks};
        asm volatile("" : "+v"(bf[j]));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const u32x4_t*>(smem + ((a_base ^ (uint32_t)(ks << 5)) + i * 4096));
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const u32x4_t*>(smem + ((b_base ^ (uint32_t)(ks << 5)) + j * 2048));
    }
  };
  auto multiply_stage = [&](int t) {
    const uint32_t a_base = ((uint32_t)(PF_A_OFF + ((t >> 1) % PF_NA) * PF_A_BYTES) + a_frag) ^ (uint32_t)((t & 1) << 6);
    const uint32_t b_base = (uint32_t)(PF_B_OFF + (t % PF_NB) * PF_B_BYTES) + b_frag;
    u32x4_t af[2][2], bf[2][4];
    ld_frags(a_base, b_base, 0, af[0], bf[0]);
    ld_frags(a_base, b_base, 1, af[1], bf[1]);
    __builtin_amdgcn_sched_barrier(0);   // all 12 operand reads in flight before the first MFMA
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (abl & 16) {
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" :: "v"(af[ks][i]));
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(bf[ks][j]));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][j] = pf_mfma<T>(af[ks][i], bf[ks][j], acc[i][j]);
      }
    }
  };

  // ---- prologue: activations of stages 0 .. PF_DA - 1, codes of stages 0 .. PF_DC - 1, scales of groups 0, 1 ----
  if (wave >= 4) {
#pragma unroll
    for (int ta = 0; ta < PF_DA; ++ta) {
      issue_a(ta, 0);
      issue_a(ta, 1);
    }
  } else {
    issue_s(0);
    issue_s(1);
#pragma unroll
    for (int t = 0; t < PF_DC; ++t) issue_c(t);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wave < 4) expand_stage(0);
  pf_barrier();

  // ---- main loop: iteration t multiplies stage t.  Waves 4..7 issue a half of the activations' double stage
  // t / 2 + PF_DA after making sure that what they issued two iterations ago or earlier has landed; waves 0..3 issue
  // codes(t + PF_DC) (and, when that stage opens a group, the group's scales), wait for THEIR OWN codes(t + 1) and expand
  // stage t + 1 into the other half of the expanded ring. ----
  W4P_T0();
  for (int t4 = 0; t4 < T_ST; t4 += 4) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int t = t4 + s;
      if (wave >= 4) {
        // what this wave issued two iterations ago or earlier has landed (the 4 pieces of the previous one may stay in
        // flight): at the top of an odd iteration that completes double stage (t + 1) / 2, which the barrier publishes
        pf_wait_vm<4>();
        W4P_ACC(3);
        issue_a((t >> 1) + PF_DA, s & 1);
        W4P_ACC(0);
      } else {
        if (s == 3) issue_s((t + PF_DC) >> 2);   // stage t + PF_DC = 4 (g): first of its group
        issue_c(t + PF_DC);
        W4P_ACC(0);
        // codes(t + 1) were issued PF_DC - 1 iterations ago: the PF_DC - 1 pieces since, and a scale piece among them
        // when one of those iterations had s == 3 (always, with PF_DC - 1 = 4 iterations), may stay in flight
        pf_wait_vm<PF_DC - 1 + 1>();
        W4P_ACC(3);
        expand_stage(t + 1);
        W4P_ACC(2);
      }
      multiply_stage(t);
      W4P_ACC(1);
      pf_barrier();
      W4P_ACC(4);
    }
  }
  W4P_FLUSH();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail's zero-filled pieces

  // ---- epilogue.  D[m][n] of a 32x32 tile: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
  const int n_wave = chunk0 * 64 + 128 * wn;           // first column of the wave's two chunks
  const int row_l = m0 + 64 * wm + 4 * kh;
  if (p.splits == 1 && p.epi != 2) {
    if (p.epi) {
      // silu(gate) * up on column-interleaved gate_up weights (chunk = [gate 32 | up 32]); roundings of the two ops it
      // replaces (reference activation_kernels.cu:14-26): gate and up rounded to the model dtype, silu rounded, product rounded
      const int ldc = p.N >> 1;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (n_wave + 64 * c >= p.N) continue;
        uint16_t* cp = p.c + ((n_wave + 64 * c) >> 1) + n5;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = row_l + 32 * i + (r & 3) + 8 * (r >> 2);
            if (row >= p.M) continue;
            const float gb = round_trip<T>(acc[i][2 * c][r]), ub = round_trip<T>(acc[i][2 * c + 1][r]);
            cp[(int64_t)row * ldc] = T::from_float(round_trip<T>(gb / (1.0f + expf(-gb))) * ub);
          }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
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
  for (int j = 0; j < 4; ++j) {
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
  const int tile = m_block * n_blocks + n_block;
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
  // the kernel's XCD-aware decomposition of the 1-D grid (see its head): rounds of G row blocks x padded units
  const int G = pl.m_blocks >= 8 ? 8 : pl.m_blocks >= 4 ? 4 : pl.m_blocks >= 2 ? 2 : 1, XQ = 8 / G;
  const int units = pl.n_blocks * pl.splits, units_pad = (units + XQ - 1) / XQ * XQ;
  const int rounds = (pl.m_blocks + G - 1) / G;
  dim3 grid((unsigned)(rounds * G * units_pad)), block(PF_NTHR);
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
