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
constexpr int PF_DA = 2, PF_DC = 4;                          // ahead: activations (double stages), codes (stages)
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
  // activations: a double stage (64 k) is 32 pieces of 8 rows x 128 B; wave w moves rows 32 w .. 32 w + 31 of it, the two
  // pieces of rows + 16 h .. + 16 h + 15 in the iteration of half h.  lane = (row i >> 3, slot i & 7); source chunk =
  // slot ^ ((row >> 1) & 7) with (row >> 1) & 7 = (4 u + (i >> 4)) & 7 for piece u of the two.
  // Rows past M: the buffer's bounds return zeros only past the END of the matrix, so clamp them out explicitly.
  uint32_t a_voff[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = m0 + 32 * wave + 16 * h + 8 * u + (lane >> 3);
      const uint32_t chunk = (uint32_t)((lane & 7) ^ ((4 * u + (lane >> 4)) & 7));
      a_voff[h][u] = row < p.M ? (uint32_t)(((int64_t)row * p.K + k_wg0) * 2) + chunk * 16 : PF_OOB;
    }
  const uint32_t a_lds = lds0 + PF_A_OFF + (uint32_t)(wave * 4096);        // + slot * 32 KiB + h * 2 KiB + u * 1 KiB
  // codes: waves 0..3, wave w the native row (k-step = stage, chunk chunk0 + w) = 1 KiB
  const uint32_t c_voff = (wave < 4 && chunk0 + wave < n_chunks) ? (uint32_t)((chunk0 + wave) * 1024 + lane * 16) : PF_OOB;
  const uint32_t c_lds = lds0 + PF_C_OFF + (uint32_t)((wave & 3) * 1024);
  // scales: wave 4, lanes 0..31: 256 columns x 2 bytes of one group
  const uint32_t s_voff = (lane < 32 && chunk0 * 64 + lane * 8 < p.N) ? (uint32_t)((chunk0 * 64 + lane * 8) * 2) : PF_OOB;
  const uint32_t s_lds = lds0 + PF_S_OFF;

  auto issue_a = [&](int ta, int h) {   // half h (rows) of this wave's share of double stage ta: 2 pieces
    if constexpr (abl & 1) return;
    const bool live = 2 * ta < T_ST;
    const uint32_t slot = a_lds + (uint32_t)((ta % PF_NA) * PF_A_BYTES + h * 2048);
#pragma unroll
    for (int u = 0; u < 2; ++u)
      pf_dma16(rs_a, live ? a_voff[h][u] : PF_OOB, (uint32_t)(ta * 128), slot + (uint32_t)(u * 1024));
  };
  auto issue_c = [&](int t) {   // waves 0..3: stage t of the codes
    if constexpr (abl & 2) return;
    pf_dma16(rs_b, t < T_ST ? c_voff : PF_OOB, (uint32_t)(((k_wg0 >> 5) + t) * n_chunks * 1024),
             c_lds + (uint32_t)((t % PF_NC) * PF_C_BYTES));
  };
  auto issue_s = [&](int g) {   // wave 4: scale row of the workgroup's group g (32 lanes x 16 bytes: the other lanes would
    if constexpr (abl & 2) return;   // zero-fill what lies behind the slot)
    if (lane < 32)
      pf_dma16(rs_s, 4 * g < T_ST ? s_voff : PF_OOB, (uint32_t)(((k_wg0 >> 7) + g) * p.N * 2), s_lds + (uint32_t)((g % PF_NS) * PF_S_BYTES));
  };

  // expansion: wave w turns dwords j = 2 (w & 1), + 1 of native row chunk ce = w >> 1 into 2 x 16 bytes of the expanded image;
  // lane = (r = column within 16, g = k octet of the 32-k stage); dword j = column 64 ce + 16 j + r
  const int ce = wave >> 1, jp = wave & 1;
  const int er = lane & 15, eg = lane >> 4;
  const uint32_t e_src = (uint32_t)(PF_C_OFF + ce * 1024 + lane * 16 + 8 * jp);
  const uint32_t e_col = (uint32_t)(64 * ce + 32 * jp + er);                                        // + 16 for the second dword
  const uint32_t e_dst = (uint32_t)(PF_B_OFF) + e_col * 64 + (uint32_t)((eg ^ ((er >> 2) & 3)) << 4);   // + 1024 for the second dword

  f32x16_t acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // One stage of one wave, written out in issue order: 16 MFMAs, and in the shadow of each (a 32x32x16 MFMA holds the
  // matrix pipe for 32 clocks, the wave may issue ~6 other instructions meanwhile) a slice of everything else -- the 6
  // operand reads of the second k-step, the expansion of this wave's two dwords of stage t + 1 (2 x 23 VALU), its LDS
  // reads and writes.  On this machine side work does NOT hide behind the MFMAs of the SIMD's other wave (measured,
  // profiles/r04_prefill_v3_v6.txt: per SIMD, MFMA time and 4 clocks per other instruction add up), only behind the
  // wave's own: `sched_barrier` fences pin the order written here.
#define PF_FENCE() __builtin_amdgcn_sched_barrier(0)
  auto rd = [&](uint32_t addr) -> u32x4_t {
    if constexpr (abl & 8) {
      u32x4_t v = {addr, addr, addr, addr};
      asm volatile("" : "+v"(v));
      return v;
    } else {
      return *reinterpret_cast<const u32x4_t*>(smem + addr);
    }
  };
  auto mm = [&](f32x16_t& c, const u32x4_t& a, const u32x4_t& b) {
    if constexpr (abl & 16) asm volatile("" :: "v"(a), "v"(b));
    else c = pf_mfma<T>(a, b, c);
  };
  auto cvt8 = [&](uint32_t x, float* q) {   // native dword -> 8 codes as floats, k order
    uint32_t e = x & 0x0f0f0f0fu;          // bytes: k0, k4, k1, k5
    uint32_t o = (x >> 4) & 0x0f0f0f0fu;   // bytes: k2, k6, k3, k7
    asm volatile("" : "+v"(e), "+v"(o));   // keep the byte form: one v_cvt_f32_ubyteN per code
    q[0] = (float)(e & 0xffu); q[4] = (float)((e >> 8) & 0xffu); q[1] = (float)((e >> 16) & 0xffu); q[5] = (float)(e >> 24);
    q[2] = (float)(o & 0xffu); q[6] = (float)((o >> 8) & 0xffu); q[3] = (float)((o >> 16) & 0xffu); q[7] = (float)(o >> 24);
  };
  auto stage = [&](int t, int s) {   // s = t & 3 as a compile-time constant of the unrolled loop
    const uint32_t a0 = ((uint32_t)(PF_A_OFF + ((t >> 1) % PF_NA) * PF_A_BYTES) + a_frag) ^ (uint32_t)((t & 1) << 6);
    const uint32_t b0 = (uint32_t)(PF_B_OFF + (t % PF_NB) * PF_B_BYTES) + b_frag;
    const uint32_t a1 = a0 ^ 32u, b1 = b0 ^ 32u;
    // expansion inputs of stage t + 1 and the first k-step's operands
    uint2 x = make_uint2(0u, 0u);
    uint32_t s0b = 0, s1b = 0;
    if constexpr (!(abl & 4)) {
      x = *reinterpret_cast<const uint2*>(smem + e_src + ((t + 1) % PF_NC) * PF_C_BYTES);
      const uint16_t* sc = reinterpret_cast<const uint16_t*>(smem + PF_S_OFF + (((t + 1) >> 2) % PF_NS) * PF_S_BYTES) + e_col;
      s0b = sc[0];
      s1b = sc[16];
    }
    unsigned char* dst = smem + e_dst + ((t + 1) % PF_NB) * PF_B_BYTES;
    u32x4_t af0[2], bf0[4], af1[2], bf1[4];
    af0[0] = rd(a0); bf0[0] = rd(b0); af0[1] = rd(a0 + 4096); bf0[1] = rd(b0 + 2048); bf0[2] = rd(b0 + 4096); bf0[3] = rd(b0 + 6144);
    PF_FENCE();
    float q[8], w[8];
    float sA, cA;
    // ---- k-step 0: operand reads of k-step 1, expansion of dword 0 ----
    mm(acc[0][0], af0[0], bf0[0]); af1[0] = rd(a1); sA = T::to_float((uint16_t)s0b); cA = -8.0f * sA; PF_FENCE();
    mm(acc[1][0], af0[1], bf0[0]); af1[1] = rd(a1 + 4096); if constexpr (!(abl & 4)) cvt8(x.x, q); PF_FENCE();
    mm(acc[0][1], af0[0], bf0[1]); bf1[0] = rd(b1); PF_FENCE();
    mm(acc[1][1], af0[1], bf0[1]); bf1[1] = rd(b1 + 2048);
    if constexpr (!(abl & 4)) { w[0] = pf_fma(q[0], sA, cA); w[1] = pf_fma(q[1], sA, cA); w[2] = pf_fma(q[2], sA, cA); w[3] = pf_fma(q[3], sA, cA); }
    PF_FENCE();
    mm(acc[0][2], af0[0], bf0[2]); bf1[2] = rd(b1 + 4096);
    if constexpr (!(abl & 4)) { w[4] = pf_fma(q[4], sA, cA); w[5] = pf_fma(q[5], sA, cA); w[6] = pf_fma(q[6], sA, cA); w[7] = pf_fma(q[7], sA, cA); }
    PF_FENCE();
    mm(acc[1][2], af0[1], bf0[2]); bf1[3] = rd(b1 + 6144);
    u32x4_t wv;
    if constexpr (!(abl & 4)) { wv[0] = T::pack2(w[0], w[1]); wv[1] = T::pack2(w[2], w[3]); wv[2] = T::pack2(w[4], w[5]); wv[3] = T::pack2(w[6], w[7]); }
    PF_FENCE();
    mm(acc[0][3], af0[0], bf0[3]);
    if constexpr (!(abl & 4)) *reinterpret_cast<u32x4_t*>(dst) = wv;
    PF_FENCE();
    mm(acc[1][3], af0[1], bf0[3]); sA = T::to_float((uint16_t)s1b); cA = -8.0f * sA; PF_FENCE();
    // ---- k-step 1: expansion of dword 1 ----
    mm(acc[0][0], af1[0], bf1[0]); if constexpr (!(abl & 4)) cvt8(x.y, q); PF_FENCE();
    mm(acc[1][0], af1[1], bf1[0]); issue_a((t >> 1) + PF_DA, s & 1); PF_FENCE();
    mm(acc[0][1], af1[0], bf1[1]);
    if constexpr (!(abl & 4)) { w[0] = pf_fma(q[0], sA, cA); w[1] = pf_fma(q[1], sA, cA); w[2] = pf_fma(q[2], sA, cA); w[3] = pf_fma(q[3], sA, cA); }
    PF_FENCE();
    mm(acc[1][1], af1[1], bf1[1]);
    if constexpr (!(abl & 4)) { w[4] = pf_fma(q[4], sA, cA); w[5] = pf_fma(q[5], sA, cA); w[6] = pf_fma(q[6], sA, cA); w[7] = pf_fma(q[7], sA, cA); }
    PF_FENCE();
    mm(acc[0][2], af1[0], bf1[2]);
    if constexpr (!(abl & 4)) { wv[0] = T::pack2(w[0], w[1]); wv[1] = T::pack2(w[2], w[3]); wv[2] = T::pack2(w[4], w[5]); wv[3] = T::pack2(w[6], w[7]); }
    PF_FENCE();
    mm(acc[1][2], af1[1], bf1[2]);
    if constexpr (!(abl & 4)) *reinterpret_cast<u32x4_t*>(dst + 1024) = wv;
    PF_FENCE();
    mm(acc[0][3], af1[0], bf1[3]);
    if (wave < 4) issue_c(t + PF_DC);
    else if (wave == 4 && s == 0) issue_s((t >> 2) + 1);
    PF_FENCE();
    mm(acc[1][3], af1[1], bf1[3]); PF_FENCE();
  };
  auto expand_first = [&]() {   // stage 0, before the loop
    if constexpr (abl & 4) return;
    const uint2 x = *reinterpret_cast<const uint2*>(smem + e_src);
    const uint16_t* sc = reinterpret_cast<const uint16_t*>(smem + PF_S_OFF) + e_col;
    const float s0 = T::to_float(sc[0]), s1 = T::to_float(sc[16]);
    *reinterpret_cast<u32x4_t*>(smem + e_dst) = pf_expand<T>(x.x, s0, -8.0f * s0);
    *reinterpret_cast<u32x4_t*>(smem + e_dst + 1024) = pf_expand<T>(x.y, s1, -8.0f * s1);
  };

  // ---- prologue: activations of double stages 0 .. PF_DA - 1, codes of stages 0 .. PF_DC - 1, scale row of group 0 ----
#pragma unroll
  for (int ta = 0; ta < PF_DA; ++ta) {
    issue_a(ta, 0);
    issue_a(ta, 1);
  }
  if (wave < 4) {
#pragma unroll
    for (int t = 0; t < PF_DC; ++t) issue_c(t);
  } else if (wave == 4) {
    issue_s(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  expand_first();
  pf_barrier();

  // ---- main loop: iteration t multiplies stage t and expands stage t + 1; in the shadows of its later MFMAs a wave issues
  // its half of the activations' double stage t / 2 + PF_DA, waves 0..3 codes(t + PF_DC), wave 4 at the first stage of a
  // group the scale row of the next group.  At the bottom it makes sure that what it issued two iterations ago or earlier
  // has landed (the pieces of this and the previous iteration may stay in flight), and the barrier publishes: at the end of
  // an odd iteration the double stage (t + 1) / 2 of the activations; codes(t + 2) (issued at t - 2, expanded at t + 1);
  // the expanded stage t + 1. ----
  W4P_T0();
  for (int t4 = 0; t4 < T_ST; t4 += 4) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int t = t4 + s;
      stage(t, s);
      W4P_ACC(1);
      if (wave < 4) pf_wait_vm<6>();
      else if (wave == 4 && s < 2) pf_wait_vm<5>();
      else pf_wait_vm<4>();
      W4P_ACC(3);
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
  const int64_t slab_bytes = (int64_t)p.splits * p.M * p.N * (p.slab16 ? 2 : 4);
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
        const int64_t elem = ((int64_t)split * p.M + row) * p.N + col;
        if (p.slab16) __builtin_amdgcn_raw_buffer_store_b16(T::from_float(acc[i][j][r]), rs, (int)(elem * 2), 0, 16);
        else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r]), rs, (int)(elem * 4), 0, 16);
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
// The small tile: 128 rows x 128 columns, 64-k stages -- for calls whose 256 x 256 tiles would be too few to fill 256 CUs
// without slicing K (Llama-3-8B at 512 rows: qkv 48 tiles, o_proj and down 32; an fp32 slab costs what the large tile
// saves).  Same discipline as above: 8 alike waves (4 x 2, each 32 rows x 64 columns = 2 accumulator tiles, 8 MFMAs per
// stage), every wave expands 2 dwords per stage, moves 2 activation pieces (8 rows x 128 B: full lines), waves 0..3 a
// piece of codes, wave 4 the scale row of a group every other stage; all of it in the shadows of the wave's own MFMAs, one
// wait and one barrier per stage.  Twice the side work per MFMA of the large tile (8 other instructions per MFMA):
// expected and measured slower per flop, faster per call where it avoids the slabs.
// LDS: A 4 x 16 KiB, expanded B 2 x 16 KiB ([column][64 k], slot = chunk ^ ((column >> 1) & 7)), codes 6 x 4 KiB, scales
// 3 x 256 B = 121 KiB.
namespace {
constexpr int PS_BM = 128, PS_BN = 128, PS_BK = 64;
constexpr int PS_NA = 4, PS_NB = 2, PS_NC = 6, PS_NS = 3;
constexpr int PS_DA = 3, PS_DC = 4;
constexpr int PS_A_BYTES = PS_BM * 128, PS_B_BYTES = PS_BN * 128, PS_C_BYTES = 4096, PS_S_BYTES = 256;
constexpr int PS_A_OFF = 0;
constexpr int PS_B_OFF = PS_A_OFF + PS_NA * PS_A_BYTES;
constexpr int PS_C_OFF = PS_B_OFF + PS_NB * PS_B_BYTES;
constexpr int PS_S_OFF = PS_C_OFF + PS_NC * PS_C_BYTES;
constexpr int PS_T_OFF = PS_S_OFF + PS_NS * PS_S_BYTES;
constexpr int PS_LDS = PS_T_OFF + 64;
static_assert(PF_NTHR * 16 <= PS_B_OFF, "the split-K reduction's scratch fits the dead activation ring");
static_assert(PS_NA > PS_DA && PS_NC > PS_DC + 1, "a slot is refilled only after the stage that read it");
}  // namespace

// 1-D grid as above with 128-row blocks and 128-column blocks; p.k_per_wg: a multiple of 128.
template <typename T>
__global__ __launch_bounds__(PF_NTHR, 2) void w4a16_prefill_small_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_chunks = p.N >> 6;
  const int n_blocks = (n_chunks + 1) >> 1, m_blocks = (p.M + PS_BM - 1) / PS_BM;
  const int G = m_blocks >= 8 ? 8 : m_blocks >= 4 ? 4 : m_blocks >= 2 ? 2 : 1, XQ = 8 / G;
  const int units = n_blocks * p.splits, units_pad = (units + XQ - 1) / XQ * XQ;
  const int L = blockIdx.x, per_round = G * units_pad;
  const int rnd = L / per_round, rem = L - rnd * per_round;
  const int xcd = rem & 7;
  const int unit = (rem >> 3) * XQ + xcd / G;
  const int m_block = rnd * G + xcd % G;
  if (m_block >= m_blocks || unit >= units) return;     // uniform: padding of the last round / of the unit count
  const int n_block = unit % n_blocks, split = unit / n_blocks;
  const int chunk0 = n_block * 2;
  const int m0 = m_block * PS_BM;
  const int k_wg0 = split * p.k_per_wg;
  const int T_ST = min(p.k_per_wg, p.K - k_wg0) >> 6;     // 64-k stages of this workgroup (uniform, even)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // consumer identity: 32-row block wm, 64-column block (= chunk of the strip) wn; lane = (row / column n5, k half kh)
  const int wm = wave & 3, wn = wave >> 2;
  const int n5 = lane & 31, kh = lane >> 5;
  // both images: row * 128 + ((2 ks + kh) ^ ((row >> 1) & 7)) * 16 = base ^ (ks << 5), ks = 0..3
  const uint32_t frag_sw = (uint32_t)((kh ^ ((n5 >> 1) & 7)) << 4);
  const uint32_t a_frag = (uint32_t)((32 * wm + n5) * 128) + frag_sw;
  const uint32_t b_frag = (uint32_t)((64 * wn + n5) * 128) + frag_sw;      // + 4096 for the second 32-column tile

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.a), 0, (int)((int64_t)p.M * p.K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint4*>(p.b), 0, (int)(((int64_t)p.K * p.N) >> 1), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.s), 0, (int)((int64_t)(p.K >> 7) * p.N * 2), 0x00020000);
  // activations: a stage is 16 pieces of 8 rows x 128 B; wave w moves rows 16 w .. 16 w + 15; lane = (row i >> 3, slot i & 7),
  // source chunk = slot ^ ((row >> 1) & 7) with (row >> 1) & 7 = (4 u + (i >> 4)) & 7 for piece u
  uint32_t a_voff[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = m0 + 16 * wave + 8 * u + (lane >> 3);
    const uint32_t chunk = (uint32_t)((lane & 7) ^ ((4 * u + (lane >> 4)) & 7));
    a_voff[u] = row < p.M ? (uint32_t)(((int64_t)row * p.K + k_wg0) * 2) + chunk * 16 : PF_OOB;
  }
  const uint32_t a_lds = lds0 + PS_A_OFF + (uint32_t)(wave * 2048);
  // codes: waves 0..3, wave w the native row (k-step 2 t + (w >> 1), chunk chunk0 + (w & 1)) = 1 KiB
  const uint32_t c_voff = (wave < 4 && chunk0 + (wave & 1) < n_chunks) ? (uint32_t)((chunk0 + (wave & 1)) * 1024 + lane * 16) : PF_OOB;
  const uint32_t c_lds = lds0 + PS_C_OFF + (uint32_t)((wave & 3) * 1024);
  // scales: wave 4, lanes 0..15: 128 columns x 2 bytes of one group
  const uint32_t s_voff = (lane < 16 && chunk0 * 64 + lane * 8 < p.N) ? (uint32_t)((chunk0 * 64 + lane * 8) * 2) : PF_OOB;
  const uint32_t s_lds = lds0 + PS_S_OFF;

  auto issue_a = [&](int t) {
    const bool live = t < T_ST;
    const uint32_t slot = a_lds + (uint32_t)((t % PS_NA) * PS_A_BYTES);
#pragma unroll
    for (int u = 0; u < 2; ++u)
      pf_dma16(rs_a, live ? a_voff[u] : PF_OOB, (uint32_t)(t * 128), slot + (uint32_t)(u * 1024));
  };
  auto issue_c = [&](int t) {   // waves 0..3
    pf_dma16(rs_b, t < T_ST ? c_voff : PF_OOB, (uint32_t)(((k_wg0 >> 5) + 2 * t + ((wave >> 1) & 1)) * n_chunks * 1024),
             c_lds + (uint32_t)((t % PS_NC) * PS_C_BYTES));
  };
  auto issue_s = [&](int g) {   // wave 4: scale row of the workgroup's group g (16 lanes x 16 bytes)
    if (lane < 16)
      pf_dma16(rs_s, 2 * g < T_ST ? s_voff : PF_OOB, (uint32_t)(((k_wg0 >> 7) + g) * p.N * 2), s_lds + (uint32_t)((g % PS_NS) * PS_S_BYTES));
  };

  // expansion: wave w turns dwords j = 2 (w & 1), + 1 of the stage's native row w >> 1 (k-step kk = w >> 2, chunk ch = (w >> 1) & 1);
  // lane = (r = column within 16, g = k octet of the 32-k step); dword j = column 64 ch + 16 j + r, chunk of the row 4 kk + g
  const int kk = wave >> 2, ch = (wave >> 1) & 1, jp = wave & 1;
  const int er = lane & 15, eg = lane >> 4;
  const uint32_t e_src = (uint32_t)(PS_C_OFF + (wave >> 1) * 1024 + lane * 16 + 8 * jp);
  const uint32_t e_col = (uint32_t)(64 * ch + 32 * jp + er);                                            // + 16 for the second dword
  const uint32_t e_dst = (uint32_t)(PS_B_OFF) + e_col * 128 + (uint32_t)(((4 * kk + eg) ^ ((er >> 1) & 7)) << 4);   // + 2048

  f32x16_t acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  auto rd = [&](uint32_t addr) -> u32x4_t { return *reinterpret_cast<const u32x4_t*>(smem + addr); };
  auto cvt8 = [&](uint32_t x, float* q) {   // native dword -> 8 codes as floats, k order
    uint32_t e = x & 0x0f0f0f0fu;          // bytes: k0, k4, k1, k5
    uint32_t o = (x >> 4) & 0x0f0f0f0fu;   // bytes: k2, k6, k3, k7
    asm volatile("" : "+v"(e), "+v"(o));
    q[0] = (float)(e & 0xffu); q[4] = (float)((e >> 8) & 0xffu); q[1] = (float)((e >> 16) & 0xffu); q[5] = (float)(e >> 24);
    q[2] = (float)(o & 0xffu); q[6] = (float)((o >> 8) & 0xffu); q[3] = (float)((o >> 16) & 0xffu); q[7] = (float)(o >> 24);
  };
  auto stage = [&](int t, int s) {   // s = t & 1 as a compile-time constant of the unrolled loop
    const uint32_t a0 = (uint32_t)(PS_A_OFF + (t % PS_NA) * PS_A_BYTES) + a_frag;
    const uint32_t b0 = (uint32_t)(PS_B_OFF + (t % PS_NB) * PS_B_BYTES) + b_frag;
    const uint2 x = *reinterpret_cast<const uint2*>(smem + e_src + ((t + 1) % PS_NC) * PS_C_BYTES);
    const uint16_t* sc = reinterpret_cast<const uint16_t*>(smem + PS_S_OFF + (((t + 1) >> 1) % PS_NS) * PS_S_BYTES) + e_col;
    const uint32_t s0b = sc[0], s1b = sc[16];
    unsigned char* dst = smem + e_dst + ((t + 1) % PS_NB) * PS_B_BYTES;
    u32x4_t af[4], bf[4][2];
    af[0] = rd(a0); bf[0][0] = rd(b0); bf[0][1] = rd(b0 + 4096);
    PF_FENCE();
    float q[8], w[8], sA, cA;
    u32x4_t wv;
    // M0
    acc[0] = pf_mfma<T>(af[0], bf[0][0], acc[0]);
    af[1] = rd(a0 ^ 32u); bf[1][0] = rd(b0 ^ 32u); bf[1][1] = rd((b0 ^ 32u) + 4096);
    sA = T::to_float((uint16_t)s0b); cA = -8.0f * sA;
    PF_FENCE();
    // M1
    acc[1] = pf_mfma<T>(af[0], bf[0][1], acc[1]);
    cvt8(x.x, q);
    PF_FENCE();
    // M2
    acc[0] = pf_mfma<T>(af[1], bf[1][0], acc[0]);
    af[2] = rd(a0 ^ 64u); bf[2][0] = rd(b0 ^ 64u); bf[2][1] = rd((b0 ^ 64u) + 4096);
    w[0] = pf_fma(q[0], sA, cA); w[1] = pf_fma(q[1], sA, cA); w[2] = pf_fma(q[2], sA, cA); w[3] = pf_fma(q[3], sA, cA);
    issue_a(t + PS_DA);
    PF_FENCE();
    // M3
    acc[1] = pf_mfma<T>(af[1], bf[1][1], acc[1]);
    w[4] = pf_fma(q[4], sA, cA); w[5] = pf_fma(q[5], sA, cA); w[6] = pf_fma(q[6], sA, cA); w[7] = pf_fma(q[7], sA, cA);
    wv[0] = T::pack2(w[0], w[1]); wv[1] = T::pack2(w[2], w[3]); wv[2] = T::pack2(w[4], w[5]); wv[3] = T::pack2(w[6], w[7]);
    PF_FENCE();
    // M4
    acc[0] = pf_mfma<T>(af[2], bf[2][0], acc[0]);
    af[3] = rd(a0 ^ 96u); bf[3][0] = rd(b0 ^ 96u); bf[3][1] = rd((b0 ^ 96u) + 4096);
    *reinterpret_cast<u32x4_t*>(dst) = wv;
    sA = T::to_float((uint16_t)s1b); cA = -8.0f * sA;
    PF_FENCE();
    // M5
    acc[1] = pf_mfma<T>(af[2], bf[2][1], acc[1]);
    cvt8(x.y, q);
    PF_FENCE();
    // M6
    acc[0] = pf_mfma<T>(af[3], bf[3][0], acc[0]);
    w[0] = pf_fma(q[0], sA, cA); w[1] = pf_fma(q[1], sA, cA); w[2] = pf_fma(q[2], sA, cA); w[3] = pf_fma(q[3], sA, cA);
    w[4] = pf_fma(q[4], sA, cA); w[5] = pf_fma(q[5], sA, cA); w[6] = pf_fma(q[6], sA, cA); w[7] = pf_fma(q[7], sA, cA);
    if (wave < 4) issue_c(t + PS_DC);
    else if (wave == 4 && s == 0) issue_s((t >> 1) + 2);
    PF_FENCE();
    // M7
    acc[1] = pf_mfma<T>(af[3], bf[3][1], acc[1]);
    wv[0] = T::pack2(w[0], w[1]); wv[1] = T::pack2(w[2], w[3]); wv[2] = T::pack2(w[4], w[5]); wv[3] = T::pack2(w[6], w[7]);
    *reinterpret_cast<u32x4_t*>(dst + 2048) = wv;
    PF_FENCE();
  };

  // ---- prologue: activations of stages 0 .. PS_DA - 1, codes of stages 0 .. PS_DC - 1, scale rows of groups 0, 1 ----
#pragma unroll
  for (int t = 0; t < PS_DA; ++t) issue_a(t);
  if (wave < 4) {
#pragma unroll
    for (int t = 0; t < PS_DC; ++t) issue_c(t);
  } else if (wave == 4) {
    issue_s(0);
    issue_s(1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {   // stage 0 expanded before the loop
    const uint2 x = *reinterpret_cast<const uint2*>(smem + e_src);
    const uint16_t* sc = reinterpret_cast<const uint16_t*>(smem + PS_S_OFF) + e_col;
    const float s0 = T::to_float(sc[0]), s1 = T::to_float(sc[16]);
    *reinterpret_cast<u32x4_t*>(smem + e_dst) = pf_expand<T>(x.x, s0, -8.0f * s0);
    *reinterpret_cast<u32x4_t*>(smem + e_dst + 2048) = pf_expand<T>(x.y, s1, -8.0f * s1);
  }
  pf_barrier();

  // ---- main loop: as above; at the bottom of iteration t what the wave issued at t - 2 or earlier has landed (the pieces of
  // this and the previous iteration may stay in flight); the barrier publishes A(t + 1) (issued at t - 2), codes(t + 2)
  // (issued at t - 2, expanded at t + 1) and the expanded stage t + 1.  Wave 4 issues a scale row at even t. ----
  for (int t2 = 0; t2 < T_ST; t2 += 2) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      stage(t2 + s, s);
      if (wave < 4) pf_wait_vm<6>();
      else if (wave == 4) pf_wait_vm<5>();
      else pf_wait_vm<4>();
      pf_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail's zero-filled pieces

  // ---- epilogue.  D[m][n] of a 32x32 tile: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
  const int n_wave = chunk0 * 64 + 64 * wn;
  const int row_l = m0 + 32 * wm + 4 * kh;
  if (p.splits == 1 && p.epi != 2) {
    if (p.epi) {
      if (n_wave >= p.N) return;
      uint16_t* cp = p.c + (n_wave >> 1) + n5;
      const int ldc = p.N >> 1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_l + (r & 3) + 8 * (r >> 2);
        if (row >= p.M) continue;
        const float gb = round_trip<T>(acc[0][r]), ub = round_trip<T>(acc[1][r]);
        cp[(int64_t)row * ldc] = T::from_float(round_trip<T>(gb / (1.0f + expf(-gb))) * ub);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n_wave + 32 * j + n5;
      if (col >= p.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_l + (r & 3) + 8 * (r >> 2);
        if (row < p.M) p.c[(int64_t)row * p.N + col] = T::from_float(acc[j][r]);
      }
    }
    return;
  }
  const int64_t slab_bytes = (int64_t)p.splits * p.M * p.N * (p.slab16 ? 2 : 4);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.slab, 0, (int)slab_bytes, 0x00020000);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n_wave + 32 * j + n5;
    if (col >= p.N) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_l + (r & 3) + 8 * (r >> 2);
      if (row >= p.M) continue;
      const int64_t elem = ((int64_t)split * p.M + row) * p.N + col;
      if (p.slab16) __builtin_amdgcn_raw_buffer_store_b16(T::from_float(acc[j][r]), rs, (int)(elem * 2), 0, 16);
      else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[j][r]), rs, (int)(elem * 4), 0, 16);
    }
  }
  if (p.epi == 2) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int* ticket_s = reinterpret_cast<int*>(smem + PS_T_OFF);
  __syncthreads();
  const int tile = m_block * n_blocks + n_block;
  if (tid == 0)
    *ticket_s = __hip_atomic_fetch_add(p.tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*ticket_s != p.splits - 1) return;
  if (tid == 0) __hip_atomic_store(p.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  splitk_reduce_tile<T, PF_NTHR>(p, rs, m0, PS_BM, chunk0 * 64, PS_BN, reinterpret_cast<f32x4_t*>(smem));
}

// ---------------------------------------------------------------------------------------------------------
// Host side.
static int env_p(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// Plan: the tile (256 x 256, or 128 x 128 where the large one would leave most of the 256 CUs idle or need fp32 slabs to
// fill them) and the split-K count (whole 128-k groups per slice) by an estimate from the measured main-loop rates.
// NMV_W4P=0 turns the kernels off (native calls of more than 64 rows then take w4n_gemm_kernel's 64-row tiles);
// NMV_W4P_TILE / NMV_W4P_SPLITS force a choice (tests, tools/sweep_prefill.py).
// Against the tall kernel on a Marlin tensor (MI355X, profiles/r04_prefill_sweep.txt, Llama-3-8B): level at 65 .. 256
// rows except o_proj (+ 3-7 us), ahead from 512 rows on the wide projection (gate_up M = 512: 128 us against 165-173)
// and from 1024 on the others (M = 1024: o 52 against 66, down 127 against 152; M = 2048: qkv 112 against 144).
// w4p_wins(): the calls a holder of BOTH tensors should send here (nmv_w4_native_prefill_plan).
bool w4p_wins(int M, int N, int K) {
  if (env_p("NMV_W4P", 1) <= 0 || M < 512 || N % 64 != 0 || K % 128 != 0) return false;
  const int64_t big_tiles = (int64_t)((M + PF_BM - 1) / PF_BM) * ((N + PF_BN - 1) / PF_BN);
  return big_tiles >= 64;
}

bool w4p_make_plan(int M, int N, int K, int64_t tickets_len, bool unsplit, W4PrefillPlan* out) {
  const int mode = env_p("NMV_W4P", 1);
  if (mode <= 0) return false;
  if (M < env_p("NMV_W4P_MIN_M", 65) || N % 64 != 0 || K % 128 != 0) return false;
  if ((int64_t)M * K * 2 >= ((int64_t)1 << 31) || (((int64_t)K * N) >> 1) >= ((int64_t)1 << 31)) return false;
  const int groups = K / 128;
  const int forced = unsplit ? 0 : env_p("NMV_W4P_SPLITS", 0);
  const int max_splits = unsplit ? 1 : env_p("NMV_W4P_MAX_SPLITS", 8);
  const int tile_env = env_p("NMV_W4P_TILE", 0);   // 0: by the estimate below, 1: 128 x 128, 2: 256 x 256
  W4PrefillPlan best_pl;
  double best_cost = 1e30;
  bool found = false;
  for (int small = 0; small < 2; ++small) {
    if (tile_env && (tile_env == 1) != (small == 1)) continue;
    const int bm = small ? PS_BM : PF_BM, bn = small ? PS_BN : PF_BN;
    W4PrefillPlan pl;
    pl.small = small;
    pl.m_blocks = (M + bm - 1) / bm;
    pl.n_blocks = (N + bn - 1) / bn;
    const int64_t tiles = (int64_t)pl.m_blocks * pl.n_blocks;
    for (int s = 1; s <= groups && s <= max_splits; ++s) {
      if (groups % s != 0) continue;
      if (s > 1 && tiles > tickets_len) break;
      if (forced && s != forced) continue;
      const int64_t rounds = (tiles * s + 255) / 256;
      // in us, from the measured main-loop rates (profiles/r04_prefill_sweep.txt): 1.86 / 0.72 per 64 k of a large / small
      // tile, ~5 of prologue + epilogue; a slice's fp32 slab is written in the shadow of the next tiles but read back
      // (by the last workgroup of the tile, or by the consumer launch in deferred mode) at ~4 TB/s
      const double per64 = small ? 0.72 : 1.86;
      const double cost = (double)rounds * (K / 64 / s * per64 + 5.0) + (s > 1 ? (double)M * N * 4.0 * s / 4e6 + 2.0 : 0.0);
      if (cost < best_cost) {
        best_cost = cost;
        best_pl = pl;
        best_pl.splits = s;
        found = true;
      }
    }
  }
  if (!found) return false;
  best_pl.k_per_wg = (groups / best_pl.splits) * 128;
  best_pl.lds_bytes = best_pl.small ? PS_LDS : PF_LDS;
  *out = best_pl;
  return true;
}

template <typename T, bool SMALL>
static int w4p_launch_one(const W4PrefillPlan& pl, const GemmParams& p, hipStream_t s) {
  auto kern = SMALL ? w4a16_prefill_small_kernel<T> : w4a16_prefill_kernel<T>;
  static unsigned long long optin = 0;   // per instantiation
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -2;
  if (lds_optin_needed(&optin, dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return -2;
  }
  // the kernels' XCD-aware decomposition of the 1-D grid (see their heads): rounds of G row blocks x padded units
  const int G = pl.m_blocks >= 8 ? 8 : pl.m_blocks >= 4 ? 4 : pl.m_blocks >= 2 ? 2 : 1, XQ = 8 / G;
  const int units = pl.n_blocks * pl.splits, units_pad = (units + XQ - 1) / XQ * XQ;
  const int rounds = (pl.m_blocks + G - 1) / G;
  dim3 grid((unsigned)(rounds * G * units_pad)), block(PF_NTHR);
  hipLaunchKernelGGL(kern, grid, block, pl.lds_bytes, s, p);
  return 0;
}

int w4p_launch(const W4PrefillPlan& pl, const GemmParams& p, bool f16, hipStream_t s) {
  if (!p.native) return -1;
  if (pl.small) return f16 ? w4p_launch_one<F16, true>(pl, p, s) : w4p_launch_one<BF16, true>(pl, p, s);
  return f16 ? w4p_launch_one<F16, false>(pl, p, s) : w4p_launch_one<BF16, false>(pl, p, s);
}

}  // namespace nmv

#if defined(NMV_W4P_STAMPS)
extern "C" int w4p_dbg_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(nmv::g_w4p_stamps), (size_t)n * 8);
}
#endif
