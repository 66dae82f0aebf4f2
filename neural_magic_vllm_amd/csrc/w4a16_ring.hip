// W4A16 GEMM for decode batches of 17..64 rows on the MFMA-native weight tensor: the round-4 "ring" kernel.
//
// Behavioural reference: /root/reference/csrc/quantization/gptq_marlin/gptq_marlin.cu:396-1363 (its 4-stage
// cp.async pipeline: gptq_marlin.cuh:19-20; the same op: C[M,N] = A[M,K] . ((q - 8) * s[k/128, n])).
//
// Why another kernel: the round-3 stream kernel's 64-row stage ran its pipes one after another --
// the waves that issue the MFMAs also fetched the weights into a 4-k-step register ring, parked the next
// activation stage through registers (loads, 32 v_dot2, LDS stores) and met at a workgroup barrier per stage;
// at 251 registers there were two such waves per SIMD and nothing to overlap with.  Here the roles are split:
//
//   * 4 LOADER waves (one per SIMD) move everything by LDS-DMA (`buffer_load_dwordx4 ... lds`, no registers, no
//     LDS stores): per 128-k scale group one ring SLOT = activations [16 MT rows][128 k] (from L2, rows XOR-swizzled
//     through the per-lane source address so that the operand reads are conflict-free), the group's 4-bit codes
//     of the workgroup's 128 columns (8 KiB, non-temporal) and its scale row.  A loader keeps D groups in flight
//     behind a counted `s_waitcnt vmcnt`, publishes a landed group with one LDS add to its FULL word and then sums
//     the rows it moved (the zero-point term, below).
//   * 12 CONSUMER waves (three per SIMD) = 4 column blocks of 32 x 3 k-lanes: a wave owns 32 columns x 16 MT rows
//     and every third scale group, whole: 8 k-steps of MT / 2 `v_mfma_f32_32x32x16` each, on operands read from the
//     slot (activations = A operand: `ds_read_b128`; codes: 8 dwords of the native image, expanded in registers
//     exactly once chip-wide).  It waits for a slot by polling the FULL word, and releases it with one LDS add to
//     the slot's FREE word as soon as its last operand read has been issued (LDS serves a wave's operations in
//     order).  No workgroup barrier inside the K loop.
//   * Accumulators: columns on the lanes (D[m][n]: n = lane & 31), so a group's scale is ONE register:
//     acc += s[g, n] * acc_group, a v_pk_fma per two elements, no LDS traffic; 16 MT / 2 + 16 MT / 2 registers.
//   * Zero point as in the stream kernel: sum (128 + q) a - 136 sum a, the second term as MFMA k-steps over the
//     groups at the end (k slots = (group, hi / mid [/ lo] part of -136 * S[g][m])), S summed by the loaders.
//   * The k-lanes meet in LDS after the loop, transposed on the way: every store of the epilogue (model dtype,
//     silu(gate) * up, fp32 slabs) is a whole 16-byte piece of a row.
// What bounds it (measured, tools/debug/ring_timeline.py): instruction issue.  A SIMD retires about one wave
// instruction per 2.5-3 cycles whatever the mix, so the kernel is written to the instruction count per 128-k group.
// Every spin is bounded: a wave that gives up opens every gate (all counters are set far past any target), so the
// workgroup drains with garbage in its tile and `nmv_w4_ring_timeouts()` reports it.
#include <climits>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "w4a16_common.h"

namespace nmv {

__device__ unsigned int g_w4r_timeouts;
#if defined(NMV_W4R_STAMPS)
// development: with NMV_W4R_DBG bit 32 every wave of the first 256 workgroups records the 100 MHz clock at 8 points
__device__ unsigned long long g_w4r_stamps[256 * 16 * 16];   // 0..7 clock stamps, 8..15 summed phase times (shader clocks)
#define W4R_STAMP(i)                                                                                      \
  do {                                                                                                    \
    if ((dbg & 32) && lane == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 256)               \
      g_w4r_stamps[(blockIdx.x * 16 + wave) * 16 + (i)] = __builtin_amdgcn_s_memrealtime();                \
  } while (0)
// phase timers: W4R_T0(); ...; W4R_ACC(k) adds the shader clocks since the last W4R_T0 / W4R_ACC to sum k (0..7)
#define W4R_T0() unsigned long long w4r_t = __builtin_amdgcn_s_memtime(); unsigned long long w4r_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define W4R_ACC(k)                                                 \
  do {                                                             \
    const unsigned long long n_ = __builtin_amdgcn_s_memtime();    \
    w4r_sum[k] += n_ - w4r_t;                                      \
    w4r_t = n_;                                                    \
  } while (0)
#define W4R_FLUSH()                                                                                       \
  do {                                                                                                    \
    if ((dbg & 32) && lane == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 256)               \
      for (int k_ = 0; k_ < 8; ++k_) g_w4r_stamps[(blockIdx.x * 16 + wave) * 16 + 8 + k_] = w4r_sum[k_]; \
  } while (0)
#else
#define W4R_STAMP(i)
#define W4R_T0()
#define W4R_ACC(k)
#define W4R_FLUSH()
#endif

namespace {

constexpr int RW_CB = 4, RW_LOAD = 4;       // 32-column blocks (consumer waves per k-lane), loader waves
constexpr int ring_kl(int mt) { return mt >= 8 ? 1 : 2; }            // k-lanes: decode tiles 2, the 128-row prefill tile 1
constexpr int ring_threads(int mt) { return (ring_kl(mt) * RW_CB + RW_LOAD) * 64; }
constexpr int RW_GMAX = 32;                 // scale groups per workgroup (S image: RW_GMAX x rows floats)
constexpr int RW_WB = 8192, RW_SB = 256;    // codes / scale row of one group and 128 columns
constexpr int RW_OUT_LD = 132;              // floats per row of the epilogue image (128 + 4: 16-byte aligned, bank-shifted)
constexpr uint32_t RW_OOB = 0x7ffffff0u;    // a voffset no buffer of ours reaches: zeros, no request
constexpr uint32_t RW_SPIN_LIMIT = 1u << 18;
constexpr uint32_t RW_POISON = 0x40000000u;
constexpr int RW_NFLAGS = 64;               // full[32], free_a[8], free_w[8], sdone, ticket, pad

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// Two rings: activation slots (16 MT rows x 256 bytes, from L2: short latency) and code slots (8 KiB of 4-bit codes + the
// scale row, from HBM: long latency, so the ring is deep).  A k-lane holds one slot of each while it works on a group.
template <int MT> struct RingGeom {
  static constexpr int KL = ring_kl(MT);
  static constexpr int MP = 16 * MT;
  static constexpr int ACT = MP * 256;          // activation slot
  static constexpr int WSL = RW_WB + RW_SB;     // code slot: codes, then the scale row
  // slots: KL being read + the rest in flight (the 128-row tile: 32 KiB activation slots)
  static constexpr int RA = MT >= 8 ? 3 : 5, RW = MT >= 8 ? 4 : 8;
  static constexpr int DA = RA - KL, DW = RW - KL;   // groups a loader keeps in flight
  static_assert(DA - 1 <= 5 && DW - 1 <= 5 && 5 * (DW - 1) < 64 && 2 * MT * (DA - 1) < 64, "counted waits");
  static constexpr int W_OFF = RA * ACT;
  static constexpr int S_OFF = W_OFF + RW * WSL;
  static constexpr int F_OFF = S_OFF + RW_GMAX * MP * 4;
  static constexpr int LDS = F_OFF + RW_NFLAGS * 4;
  static_assert(KL * MP * RW_OUT_LD * 4 <= S_OFF, "the epilogue image fits the rings");
  static_assert(LDS <= 160 * 1024, "LDS");
};

template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// all but the pieces of the `rem` (uniform) most recent groups have landed, PPG pieces per group; a count the 6-bit counter
// cannot express waits for the largest whole number of groups that it can (more than asked: safe)
template <int PPG, int K>
__device__ __forceinline__ void wait_vm_k() {
  constexpr int KMAX = 63 / PPG;
  wait_vm<(K < KMAX ? K : KMAX) * PPG>();
}
template <int PPG>
__device__ __forceinline__ void wait_vm_groups(int rem) {
  if (rem <= 0) wait_vm<0>();
  else if (rem == 1) wait_vm_k<PPG, 1>();
  else if (rem == 2) wait_vm_k<PPG, 2>();
  else if (rem == 3) wait_vm_k<PPG, 3>();
  else if (rem == 4) wait_vm_k<PPG, 4>();
  else wait_vm_k<PPG, 5>();
}

// One LDS-DMA piece: 64 lanes x 16 bytes from buffer `rs` (per-lane byte offset `voff`, bounds-checked; uniform `soff`)
// to LDS bytes [lds_addr, lds_addr + 1024) in lane order.  hipcc neither counts this load nor orders LDS reads behind it:
// the caller waits with wait_vm<>.  M0 is written in the statement that uses it (nothing else in this kernel reads M0).
template <bool NT>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff, uint32_t lds_addr) {
  if constexpr (NT)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// LDS add without a wait in front of it: LDS serves the operations of a wave in issue order, so the reads issued before it
// have been performed when the add is; the "memory" clobber keeps the compiler from moving them across.
// Issued by lane 0 alone through EXEC inside the statement -- the callers run with all 64 lanes active (uniform control
// flow in whole waves) -- because an `if (lane == 0)` around it is a branch, and a branch ends the scheduling region.
__device__ __forceinline__ void lds_signal(uint32_t lds_addr) {
  const uint32_t one = 1;
  asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(lds_addr), "v"(one) : "memory");
}

// wait until *word >= need (one relaxed LDS poll per trip, the whole wave reads the same word)
__device__ __forceinline__ void spin_ge(uint32_t* word, uint32_t need, uint32_t* flags, int lane) {
  for (uint32_t it = 0;; ++it) {
    const uint32_t v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (v >= need) break;
    if (it > RW_SPIN_LIMIT) {   // give up: open every gate, the workgroup drains
      if (lane < RW_GMAX + 17) __hip_atomic_store(flags + lane, RW_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (lane == 0) atomicAdd(&g_w4r_timeouts, 1u);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum_f(float v) {   // sum over the 16 lanes of a DPP row, in every lane
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x140>(v);
  return v;
}

template <typename T>
__device__ __forceinline__ f32x16_t mfma32(u32x4_t a, uint4 b4, f32x16_t c) {
  const u32x4_t b = {b4.x, b4.y, b4.z, b4.w};
  if constexpr (std::is_same<T, F16>::value)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <typename T>
__device__ __forceinline__ f32x16_t mfma32(uint4 a, uint4 b, f32x16_t c) {
  if constexpr (std::is_same<T, F16>::value)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

}  // namespace

// grid (ceil(N / 128), splits, ceil(M / (16 MT))), 1024 threads: waves 0..11 consume, waves 12..15 load.
// p.b: native[kstep][chunk][lane] (uint4), p.s: natural [groups, N]; p.k_per_wg = 128 * (groups per workgroup) <= 4096.
// MT = 2 or 4 (32 or 64 rows: whole 32-row MFMA tiles).
template <typename T, int MT>
__global__ __launch_bounds__(ring_threads(MT), MT >= 8 ? 2 : 3) void w4a16_ring_kernel(const GemmParams p) {
  using GEO = RingGeom<MT>;
  constexpr int MP = GEO::MP, ACT = GEO::ACT, WSL = GEO::WSL, RA = GEO::RA, RW = GEO::RW, DA = GEO::DA, DW = GEO::DW;
  constexpr int MT2 = MT / 2;   // 32-row MFMA tiles
  constexpr int RW_KL = GEO::KL, RW_CONS = RW_KL * RW_CB, RW_NTHR = ring_threads(MT);
  constexpr bool STEPWISE = MT2 > 2;   // the 128-row tile reads its activation fragments k-step by k-step
  constexpr bool IS_F16 = std::is_same<T, F16>::value;
  constexpr float ZPC = W4N<T>::ZPC;
  static_assert(MT == 2 || MT == 4 || MT == 8, "whole 32-row tiles");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_chunks = p.N >> 6;
  const int chunk0 = blockIdx.x * 2;
  const int split = blockIdx.y;
  const int m0 = blockIdx.z * MP;
  const int k_wg0 = split * p.k_per_wg;
  const int G = min(p.k_per_wg, p.K - k_wg0) >> 7;     // scale groups of this workgroup (uniform, 1..RW_GMAX)
  // development switches (compile time, -DNMV_W4R_ABL=bits: results garbage, times valid): 1 no row sums, 2 no MFMA, 4 no
  // operand reads, 8 no expansion / MFMA, 16 no DMA; -DNMV_W4R_STAMPS: with NMV_W4R_DBG=32 every wave stamps the clock
#ifndef NMV_W4R_ABL
#define NMV_W4R_ABL 0
#endif
#if defined(NMV_W4R_STAMPS)
  const int dbg = NMV_W4R_ABL | (p.g_stage & 32);
#else
  constexpr int dbg = NMV_W4R_ABL;
#endif

  float* s_all = reinterpret_cast<float*>(smem + GEO::S_OFF);          // [G][MP]: -ZPC * sum of the group's activations
  uint32_t* flags = reinterpret_cast<uint32_t*>(smem + GEO::F_OFF);
  uint32_t* full = flags;                                               // [RW_GMAX]: loader arrivals per group (4 = landed)
  uint32_t* free_a = flags + RW_GMAX;                                   // [RA]: consumer releases per activation slot, cumulative
  uint32_t* free_w = flags + RW_GMAX + 8;                               // [RW]: ... per code slot
  uint32_t* sdone = flags + RW_GMAX + 16;                                // consumers that have written their last row sum
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  W4R_STAMP(0);
  if (tid < RW_NFLAGS) flags[tid] = 0;
  __syncthreads();
  W4R_STAMP(1);

  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.s), 0, (int)((int64_t)(p.K >> 7) * p.N * 2), 0x00020000);

  // consumer identity (also used by the epilogue): k-lane, 32-column block; lane = (column n5, k half kh)
  const int kl = RW_KL == 1 ? 0 : wave >> 2, cb = wave & 3;
  const int n5 = lane & 31, kh = lane >> 5;
  f32x16_t acc[MT2];
#pragma unroll
  for (int t = 0; t < MT2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  if (wave >= RW_CONS) {
    // ------------------------------------------------ loader ------------------------------------------------
    // `s_waitcnt vmcnt` counts a wave's loads in issue order, so a wave moves ONE of the two streams: waves 0, 1 the
    // activations (half of the rows each), waves 2, 3 the codes (two k-steps each; wave 2 the scale row as well).
    const int lw = wave - RW_CONS;
    W4R_T0();
    if (lw < 2) {
      const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<uint16_t*>(p.a), 0, (int)((int64_t)p.M * p.K * 2), 0x00020000);
      // piece u of this wave = rows 4 (lw + 2 u) + (lane >> 4) of the tile (4 rows x 256 bytes); the LDS image is
      // lane-linear, so lane (row, slot) fetches chunk slot ^ (row & 15) of its row: the reader XORs the same way
      constexpr int NP = MT * 2;                 // pieces per wave and group
      uint32_t va[NP];
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        const int row = 4 * (lw + 2 * u) + (lane >> 4);
        va[u] = m0 + row < p.M ? (uint32_t)((m0 + row) * p.K * 2) + (uint32_t)(((lane & 15) ^ (row & 15)) << 4) : RW_OOB;
      }
      uint32_t so_a = (uint32_t)(k_wg0 * 2);
      uint32_t sb_i = lds0;
      int slot_i = 0;
      for (int i = 0; i < G; ++i) {
        if (i >= RA) spin_ge(free_a + slot_i, (uint32_t)(RW_CB * (i / RA)), flags, lane);
        W4R_ACC(0);
        if (!(dbg & 16)) {
#pragma unroll
          for (int u = 0; u < NP; ++u) dma16<false>(rs_a, va[u], so_a, sb_i + (uint32_t)((lw + 2 * u) * 1024));
        }
        so_a += 256;
        sb_i += ACT;
        if (++slot_i == RA) { slot_i = 0; sb_i = lds0; }
        W4R_ACC(1);
        const int j = i - (DA - 1);
        if (j >= 0) {
          wait_vm<NP*(DA - 1)>();   // NP (DA - 1) pieces were issued after group j's
          W4R_ACC(2);
          if (j == 0) W4R_STAMP(2);
          lds_signal(lds0 + (uint32_t)(GEO::F_OFF + j * 4));
          W4R_ACC(3);
        }
        if (i == (G >> 1)) W4R_STAMP(3);
      }
      for (int j = max(0, G - (DA - 1)); j < G; ++j) {
        wait_vm_groups<NP>(G - 1 - j);   // groups issued after j
        lds_signal(lds0 + (uint32_t)(GEO::F_OFF + j * 4));
      }
    } else {
      const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<uint4*>(p.b), 0, (int)(((int64_t)p.K * p.N) >> 1), 0x00020000);
      // code pieces: k-steps 2 (lw - 2) and + 1 of the group, both chunks: 1 KiB contiguous each; slot image [k-step][chunk]
      const int ks0 = 2 * (lw - 2);
      uint32_t vw[2];
#pragma unroll
      for (int c = 0; c < 2; ++c)
        vw[c] = (chunk0 + c < n_chunks) ? (uint32_t)((chunk0 + c) * 1024 + lane * 16) : RW_OOB;
      const uint32_t w_row = (uint32_t)n_chunks * 1024u;                    // one k-step of all chunks
      // scale row of the group: 128 columns = 16 lanes x 16 bytes (wave 2)
      const uint32_t vs = (lane < 16 && chunk0 * 64 + lane * 8 < p.N) ? (uint32_t)(chunk0 * 128 + lane * 16) : RW_OOB;
      const bool with_s = lw == 2;
      uint32_t so_w = (uint32_t)((k_wg0 >> 5) + ks0) * w_row;
      uint32_t so_s = (uint32_t)((k_wg0 >> 7) * p.N * 2);
      uint32_t sb_i = lds0 + GEO::W_OFF;
      int slot_i = 0;
      auto landed = [&](int j) {
        if (j == 0) W4R_STAMP(2);
        lds_signal(lds0 + (uint32_t)(GEO::F_OFF + j * 4));
      };
      for (int i = 0; i < G; ++i) {
        if (i >= RW) spin_ge(free_w + slot_i, (uint32_t)(RW_CB * (i / RW)), flags, lane);
        W4R_ACC(0);
        if (!(dbg & 16)) {
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int c = 0; c < 2; ++c)
              dma16<true>(rs_w, vw[c], so_w + (uint32_t)q * w_row, sb_i + (uint32_t)(((ks0 + q) * 2 + c) * 1024));
          if (with_s) {   // under EXEC = lanes 0..15 (a uniform branch around a divergent one)
            if (lane < 16) dma16<false>(rs_s, vs, so_s, sb_i + (uint32_t)RW_WB);
          }
        }
        so_w += 4u * w_row;
        so_s += (uint32_t)(p.N * 2);
        sb_i += WSL;
        if (++slot_i == RW) { slot_i = 0; sb_i = lds0 + GEO::W_OFF; }
        W4R_ACC(1);
        const int j = i - (DW - 1);
        if (j >= 0) {
          if (with_s) wait_vm<5 * (DW - 1)>();   // 5 (4) pieces per group were issued after group j's
          else wait_vm<4 * (DW - 1)>();
          W4R_ACC(2);
          landed(j);
          W4R_ACC(3);
        }
        if (i == (G >> 1)) W4R_STAMP(3);
      }
      for (int j = max(0, G - (DW - 1)); j < G; ++j) {
        if (with_s) wait_vm_groups<5>(G - 1 - j);   // groups issued after j
        else wait_vm_groups<4>(G - 1 - j);
        landed(j);
      }
    }
    W4R_STAMP(4);
    W4R_FLUSH();
    W4R_STAMP(5);
  } else {
    // ----------------------------------------------- consumer -----------------------------------------------
    const uint32_t kmask = __builtin_amdgcn_readfirstlane(W4N<T>::MASK);
    uint32_t kmagic = W4N<T>::MAGIC;
    asm volatile("" : "+v"(kmagic));
    // operand addresses inside a slot.  A operand of k-step t (16 k), row tile mt: row 32 mt + n5, 16-byte chunk 2 t + kh,
    // XORed with row & 15.  B operand: column 32 cb + n5 = tile j = 2 (cb & 1) + (n5 >> 4) of chunk cb >> 1, k octet
    // 2 t + kh = native lane 16 (2 (t & 1) + kh) + (n5 & 15) of k-step t >> 1: dword j of that lane's vector
    // (2 t + kh) ^ rx = (2 t) ^ (kh ^ rx): with slots a multiple of 256 bytes the chunk of k-step t is ONE xor of the
    // lane's offset with 32 t (no table of eight offsets in registers)
    const int rx = n5 & 15;
    const int a_base = n5 * 256 + ((rx ^ kh) << 4);
    static_assert(ACT % 256 == 0, "the k-step xor stays inside a row");
    const int w_base = GEO::W_OFF + (((cb >> 1) * 64 + 16 * kh + rx) << 4) + 4 * (2 * (cb & 1) + (n5 >> 4));
    const int s_base = GEO::W_OFF + RW_WB + (32 * cb + n5) * 2;
    const uint32_t ones2 = W4N<T>::ONES;

    f32x16_t accg[MT2];
    uint32_t wq[8];
    uint4 w4n = make_uint4(0, 0, 0, 0);
    (void)w4n;
    // the scales of the zero-point pass after the loop (this k-lane's rounds of 8 groups, below) are fetched after the loop,
    // under the wait for the other consumers' row sums (requesting them before the loop costs four registers the loop
    // does not have: the 64-row tile sits at the 128-register limit of four waves per SIMD)
    constexpr int ZR = (RW_GMAX / 8 + RW_KL - 1) / RW_KL;
    const int zcol = chunk0 * 64 + 32 * cb + n5;
    auto ld_zs = [&](int rr, uint32_t (&z)[4]) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int gz = 8 * (kl + RW_KL * rr) + 4 * kh + e;
        const bool ok = gz < G && zcol < p.N;
        z[e] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(
            rs_s, ok ? (int)((((k_wg0 >> 7) + gz) * p.N + zcol) * 2) : (int)RW_OOB, 0, 0);
      }
    };
    int slot_a = kl, slot_w = kl;   // slots of group gi = gi % RA, gi % RW; the lane's groups are RW_KL apart
    W4R_T0();
    for (int gi = kl; gi < G; gi += RW_KL) {
      const unsigned char* sl = smem + slot_w * WSL;   // the code slot (w_base, s_base carry the ring's offset)
      if (gi == (G >> 1) + kl) W4R_STAMP(3);
      W4R_ACC(0);
      spin_ge(full + gi, RW_LOAD, flags, lane);
      W4R_ACC(1);
      if (gi == kl) W4R_STAMP(2);
      const uint32_t sc = *reinterpret_cast<const uint16_t*>(sl + s_base);
      // every operand of the group is requested at once -- 8 code dwords, 8 MT2 activation fragments (64 registers at 64
      // rows: the kernel runs three waves per SIMD, 168 registers) -- and both slots are handed back before the first MFMA:
      // the LDS latency is paid once per group, not once per k-step, and a slot is held for the length of the issue only
      if (!(dbg & 4)) {
        const int w_slot = slot_w * WSL + w_base;   // one address register, eight immediate offsets
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {   // k-steps 2 t2 (offset 0) and 2 t2 + 1 (+ 512 bytes) of 32-k step t2 (2 KiB apart)
          wq[2 * t2] = *reinterpret_cast<const uint32_t*>(smem + w_slot + t2 * 2048);
          wq[2 * t2 + 1] = *reinterpret_cast<const uint32_t*>(smem + w_slot + t2 * 2048 + 512);
        }
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) wq[t] = 0x88888888u;
      }
      lds_signal(lds0 + (uint32_t)(GEO::F_OFF + (RW_GMAX + 8 + slot_w) * 4));
      // row sums of the group (zero-point term).  Decode tiles: taken by ONE of the four column-block waves that read the
      // group, in turn, from the fragments it holds.  128-row tile: wave cb sums row tile cb of every group from a second
      // read of that tile's fragment (a run-time register index would cost more than the read)
      const bool summer = !STEPWISE && !(dbg & 1) && (((gi - kl) / RW_KL) & 3) == cb;   // uniform
      float rs[MT2];
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt) rs[mt] = 0.f;
      constexpr int NAF = STEPWISE ? 2 : 8;
      u32x4_t af[NAF][MT2];   // vector loads: as a struct of four dwords hipcc splits a fragment into b96 + b32 reads
      const int a_slot = slot_a * ACT + a_base;
      auto rd_a = [&](int t, u32x4_t (&a)[MT2]) {
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) {
          if (dbg & 4) a[mt] = u32x4_t{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
          else a[mt] = *reinterpret_cast<const u32x4_t*>(smem + ((a_slot ^ (t << 5)) + mt * 8192));
        }
      };
      if constexpr (STEPWISE) {
        rd_a(0, af[0]);
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) rd_a(t, af[t]);
        lds_signal(lds0 + (uint32_t)(GEO::F_OFF + (RW_GMAX + slot_a) * 4));
      }
      // the eight k-steps: one straight-line body (a branch per step would end the scheduling region there: measured, the
      // loop took twice as long); each step is fenced so that hipcc expands a code dword next to its MFMAs instead of
      // expanding all eight ahead (32 more live registers)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        constexpr int dummy = 0;
        (void)dummy;
        const int ti = STEPWISE ? (t & 1) : t;
        if constexpr (STEPWISE) {
          if (t + 1 < 8) rd_a(t + 1, af[(t + 1) & 1]);
          if (!(dbg & 1)) {
            const u32x4_t as = *reinterpret_cast<const u32x4_t*>(smem + ((a_slot ^ (t << 5)) + cb * 8192));
            rs[0] = T::dot2(as[0], ones2, rs[0]);
            rs[0] = T::dot2(as[1], ones2, rs[0]);
            rs[0] = T::dot2(as[2], ones2, rs[0]);
            rs[0] = T::dot2(as[3], ones2, rs[0]);
          }
          if (t == 6) lds_signal(lds0 + (uint32_t)(GEO::F_OFF + (RW_GMAX + slot_a) * 4));   // the last read is out
        }
        if (dbg & 8) {
          if (t == 0) {
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
              for (int i = 0; i < 16; ++i) accg[mt][i] = 0.f;
          }
          asm volatile("" ::"v"(af[ti][0][0]), "v"(wq[t]));
          continue;
        }
        constexpr int P0 = W4N<T>::POS;
        auto expand = [&](uint32_t x) -> uint4 {
          return make_uint4(and_or(P0 == 0 ? x : x << P0, kmask, kmagic),
                            and_or(4 >= P0 ? x >> (4 - P0) : x << (P0 - 4), kmask, kmagic),
                            and_or(x >> (8 - P0), kmask, kmagic), and_or(x >> (12 - P0), kmask, kmagic));
        };
#ifndef NMV_W4R_NO_PIPE
        // software-pipelined expansion: the NEXT k-step's code dword is expanded in the shadow of this step's MFMAs
        // (one MFMA, half of the expansion, the other MFMA, the rest: pinned for the machine scheduler)
        if (t == 0) w4n = expand(wq[0]);
        const uint4 w4 = w4n;
        if (t + 1 < 8) w4n = expand(wq[t + 1]);
#else
        const uint4 w4 = expand(wq[t]);
#endif
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) {
          if (dbg & 2) {
            asm volatile("" ::"v"(af[ti][mt][0]), "v"(af[ti][mt][3]), "v"(w4.x), "v"(w4.y), "v"(w4.z), "v"(w4.w));
            if (t == 0) {
#pragma unroll
              for (int i = 0; i < 16; ++i) accg[mt][i] = 0.f;
            }
          } else if (t == 0) {
            f32x16_t z;
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = 0.f;
            accg[mt] = mfma32<T>(af[ti][mt], w4, z);
          } else {
            accg[mt] = mfma32<T>(af[ti][mt], w4, accg[mt]);
          }
        }
#ifndef NMV_W4R_NO_PIPE
        if (t + 1 < 8 && !(dbg & 10)) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // its share of the next expansion
          __builtin_amdgcn_sched_group_barrier(0x008, MT2 - 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (!STEPWISE) {
        if (summer) {   // the fragments are still in registers: one branch per group
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) {
              rs[mt] = T::dot2(af[t][mt][0], ones2, rs[mt]);
              rs[mt] = T::dot2(af[t][mt][1], ones2, rs[mt]);
              rs[mt] = T::dot2(af[t][mt][2], ones2, rs[mt]);
              rs[mt] = T::dot2(af[t][mt][3], ones2, rs[mt]);
            }
        }
      }
      const float sf = T::to_float((uint16_t)sc);
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = fmaf(sf, accg[mt][i], acc[mt][i]);
      // lane (row, k half) + its other half; the kh = 0 lanes store
      if constexpr (STEPWISE) {
        if (!(dbg & 1)) {
          const float other = __builtin_bit_cast(
              float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, rs[0])));
          if (kh == 0) s_all[gi * MP + 32 * cb + n5] = -ZPC * (rs[0] + other);
        }
      } else if (summer) {
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) {
          const float other = __builtin_bit_cast(
              float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, rs[mt])));
          if (kh == 0) s_all[gi * MP + 32 * mt + n5] = -ZPC * (rs[mt] + other);
        }
      }
      slot_a += RW_KL;
      if (slot_a >= RA) slot_a -= RA;
      slot_w += RW_KL;
      if (slot_w >= RW) slot_w -= RW;
    }
    W4R_ACC(0);
    W4R_STAMP(4);
    W4R_FLUSH();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // my row sums are in LDS
    if (lane == 0) __hip_atomic_fetch_add(sdone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

    // ---- zero point: acc += sum_g (-ZPC S[g, m]) * s[g, n] as MFMA k-steps of 8 groups, dealt over the k-lanes (their
    //      tiles are summed below): slot 2 e + part of lane half kh = group gq + 4 kh + e, part = hi / mid half of the
    //      fp32 value (bf16: a third part in a second MFMA; fp16: the value travels as z / 16 beside 16 s) ----
    {
      constexpr int NPASS = IS_F16 ? 1 : 2;
      uint32_t zs[ZR][4];
#pragma unroll
      for (int rr = 0; rr < ZR; ++rr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) zs[rr][e] = 0u;
        if (8 * (kl + RW_KL * rr) < G) ld_zs(rr, zs[rr]);
      }
      spin_ge(sdone, RW_CONS, flags, lane);   // every consumer has written its last row sum
#pragma unroll
      for (int rr = 0; rr < ZR; ++rr) {
        const int gq = 8 * (kl + RW_KL * rr);
        if (gq >= G) break;
        uint32_t sd[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          uint32_t s16 = zs[rr][e];
          if constexpr (IS_F16) s16 = T::from_float(16.0f * T::to_float((uint16_t)s16));
          sd[e] = s16 | (s16 << 16);
        }
        const uint4 sa = make_uint4(sd[0], sd[1], sd[2], sd[3]);
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) {
          uint32_t d[NPASS][4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int gi = gq + 4 * kh + e;
            float z = gi < G ? s_all[gi * MP + 32 * mt + n5] : 0.f;
            if constexpr (IS_F16) {
              z *= 0.0625f;
              const uint16_t zh = T::from_float(z);
              d[0][e] = (uint32_t)zh | ((uint32_t)T::from_float(z - T::to_float(zh)) << 16);
            } else {
              const uint32_t zh = __float_as_uint(z) & 0xffff0000u;
              const float rem = z - __uint_as_float(zh);
              const uint16_t zm = T::from_float(rem);
              d[0][e] = (zh >> 16) | ((uint32_t)zm << 16);
              d[NPASS - 1][e] = (uint32_t)T::from_float(rem - T::to_float(zm));
            }
          }
#pragma unroll
          for (int ps = 0; ps < NPASS; ++ps)
            acc[mt] = mfma32<T>(make_uint4(d[ps][0], d[ps][1], d[ps][2], d[ps][3]), sa, acc[mt]);
        }
      }
    }
  }

  // ---- the k-lanes meet in LDS, transposed: out[kl][m][n] (row stride RW_OUT_LD floats) ----
  if (wave < RW_CONS) W4R_STAMP(5);
  __syncthreads();   // the ring is dead: loaders have drained, consumers have read their last slot
  W4R_STAMP(6);
  float* out = reinterpret_cast<float*>(smem);
  if (wave < RW_CONS) {
    // D[m][n] of a 32x32 tile: column = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
    float* o = out + kl * (MP * RW_OUT_LD) + (4 * kh) * RW_OUT_LD + 32 * cb + n5;
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[(32 * mt + (i & 3) + 8 * (i >> 2)) * RW_OUT_LD] = acc[mt][i];
  }
  __syncthreads();
  W4R_STAMP(7);

  const int n0 = chunk0 * 64;
  auto tile_sum = [&](int m, int n) -> f32x4_t {
    f32x4_t v = *reinterpret_cast<const f32x4_t*>(out + m * RW_OUT_LD + n);
#pragma unroll
    for (int q = 1; q < RW_KL; ++q) v += *reinterpret_cast<const f32x4_t*>(out + (q * MP + m) * RW_OUT_LD + n);   // k-lane order
    return v;
  };
  if (p.splits == 1 && p.epi != 2) {
    if (p.epi) {
      // silu(gate) * up on column-interleaved gate_up weights (chunk = [gate 32 | up 32]); roundings of the two ops it
      // replaces (reference activation_kernels.cu:14-26): gate and up rounded to the model dtype, silu rounded, product rounded
      for (int e = tid; e < MP * 16; e += RW_NTHR) {
        const int m = e >> 4, q = e & 15;             // q: chunk of the strip (q >> 3), four gate columns 4 (q & 7)
        const int n = (q >> 3) * 64 + 4 * (q & 7);
        if (m0 + m >= p.M || n0 + n >= p.N) continue;
        const f32x4_t gt = tile_sum(m, n), up = tile_sum(m, n + 32);
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float gb = round_trip<T>(gt[i]), ub = round_trip<T>(up[i]);
          o[i] = round_trip<T>(gb / (1.0f + expf(-gb))) * ub;
        }
        uint2 pk;
        pk.x = T::pack2(o[0], o[1]);
        pk.y = T::pack2(o[2], o[3]);
        *reinterpret_cast<uint2*>(p.c + (int64_t)(m0 + m) * (p.N >> 1) + (n0 >> 1) + (q >> 3) * 32 + 4 * (q & 7)) = pk;
      }
      return;
    }
    for (int e = tid; e < MP * 32; e += RW_NTHR) {
      const int m = e >> 5, n = 4 * (e & 31);
      if (m0 + m >= p.M || n0 + n >= p.N) continue;
      const f32x4_t v = tile_sum(m, n);
      uint2 pk;
      pk.x = T::pack2(v[0], v[1]);
      pk.y = T::pack2(v[2], v[3]);
      *reinterpret_cast<uint2*>(p.c + (int64_t)(m0 + m) * p.N + n0 + n) = pk;
    }
    return;
  }
  // split-K across workgroups: write-through fp32 slabs, ticket, the last workgroup of the tile sums them in split
  // order (w4a16_common.h); deferred mode (epi 2) leaves the slabs to the next launch of the layer
  const int64_t slab_bytes = (int64_t)p.splits * p.M * p.N * 4;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.slab, 0, (int)slab_bytes, 0x00020000);
  for (int e = tid; e < MP * 32; e += RW_NTHR) {
    const int m = e >> 5, n = 4 * (e & 31);
    if (m0 + m >= p.M || n0 + n >= p.N) continue;
    const f32x4_t v = tile_sum(m, n);
    const int off = (int)((((int64_t)split * p.M + m0 + m) * p.N + n0 + n) * 4);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, off, 0, 16);
  }
  if (p.epi == 2) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int* ticket_s = reinterpret_cast<int*>(flags + RW_GMAX + 24);
  __syncthreads();
  const int tile = blockIdx.z * gridDim.x + blockIdx.x;
  if (tid == 0)
    *ticket_s = __hip_atomic_fetch_add(p.tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*ticket_s != p.splits - 1) return;
  if (tid == 0) __hip_atomic_store(p.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  splitk_reduce_tile<T, RW_NTHR>(p, rs, m0, MP, n0, 128, reinterpret_cast<f32x4_t*>(smem));
}

// ---------------------------------------------------------------------------------------------------------
// Host side.
static int env_r(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// Plan: 128-column strips x split-K slices of whole 128-k groups (at most RW_GMAX per workgroup), about one workgroup
// per CU; no more than NMV_W4R_MAX_SPLITS slices (each costs M * N * 8 bytes of slab traffic).
bool w4r_make_plan(int M, int N, int K, int64_t tickets_len, bool unsplit, W4RingPlan* out) {
  if (!env_r("NMV_W4R", 1)) return false;
  if (M < env_r("NMV_W4R_MIN_M", 17) || M > env_r("NMV_W4R_MAX_M", 1 << 20) || N % 64 != 0 || K % 128 != 0) return false;
  W4RingPlan pl;
  // 17..32 rows: one 32-row tile; 33..64: two; from 65 rows on: row blocks of the 128-row prefill tile
  pl.mt = env_r("NMV_W4R_MT", M <= 32 ? 2 : M <= 64 ? 4 : 8);
  if (pl.mt != 2 && pl.mt != 4 && pl.mt != 8) return false;
  const int mp = 16 * pl.mt;
  pl.m_blocks = (M + mp - 1) / mp;
  pl.n_blocks = (N / 64 + 1) / 2;
  const int groups = K / 128;
  const int base = pl.n_blocks * pl.m_blocks;
  // measured (tools/sweep_ring.py, profiles/r04_ring_sweep.txt).  17..32 rows (one 32-row MFMA tile): the ring wins on
  // every Llama-3-8B projection (gate_up 18.5 us against 25.8 at M = 32, o_proj 6.6 against 7.6 as the step issues it).
  // 33..64 rows: it wins where one k range per workgroup fills the chip (gate_up: 224 strips, 28.8 us against 32.1 at
  // M = 64) and ties on the narrow projections, whose launches are prologue and epilogue: those keep the stream kernel.
  if (M > 32 && M <= 64 && base < env_r("NMV_W4R_MIN_WGS", 160)) return false;
  // 65 rows and more (row blocks of the 128-row tile): parity-green (tests/test_gpu_w4_ring.py) but not faster than the
  // tall kernel on the Marlin tensor (profiles/r04_ring_sweep.txt: gate_up M = 512 221 us against 156, qkv 50 against 52):
  // off unless asked for
  if (M > 64 && !env_r("NMV_W4R_PREFILL", 0)) return false;
  const int target = env_r("NMV_W4R_WGS", 256);
  // prompt-sized calls: a slab costs M * N * 8 bytes of traffic, as much as the codes from M = K / 16 on: at most two
  const int max_splits = unsplit ? 1 : env_r("NMV_W4R_MAX_SPLITS", M > 64 ? (base < 128 ? 2 : 1) : 8);
  const int forced = unsplit ? 0 : env_r("NMV_W4R_SPLITS", 0);
  int best = 0, best_dist = INT_MAX;
  for (int s = 1; s <= groups && s <= max_splits; ++s) {
    if (groups % s != 0 || groups / s > RW_GMAX) continue;
    if (s > 1 && (int64_t)base > tickets_len) break;
    if (forced) {
      if (s == forced) { best = s; break; }
      continue;
    }
    const int dist = std::abs(base * s - target);
    if (dist < best_dist) { best_dist = dist; best = s; }
  }
  if (best == 0) return false;
  pl.splits = best;
  pl.k_per_wg = (groups / best) * 128;
  pl.lds_bytes = pl.mt == 8 ? RingGeom<8>::LDS : pl.mt == 4 ? RingGeom<4>::LDS : RingGeom<2>::LDS;
  *out = pl;
  return true;
}

// the > 64 KiB dynamic-LDS opt-in is per device and per kernel: one bit per device ordinal
bool lds_optin_needed(unsigned long long* mask, int device) {
  if (device < 0 || device >= 64) return true;
  const unsigned long long bit = 1ull << device;
  if (__atomic_load_n(mask, __ATOMIC_RELAXED) & bit) return false;
  __atomic_fetch_or(mask, bit, __ATOMIC_RELAXED);
  return true;
}

template <typename T, int MT>
static int w4r_launch_one(const W4RingPlan& pl, const GemmParams& p, hipStream_t s) {
  auto kern = w4a16_ring_kernel<T, MT>;
  static unsigned long long optin = 0;   // per instantiation
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -2;
  if (lds_optin_needed(&optin, dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return -2;
  }
  dim3 grid(pl.n_blocks, pl.splits, pl.m_blocks), block(ring_threads(MT));
  hipLaunchKernelGGL(kern, grid, block, pl.lds_bytes, s, p);
  return 0;
}

int w4r_launch(const W4RingPlan& pl, const GemmParams& p0, bool f16, hipStream_t s) {
  if (!p0.native) return -1;
  GemmParams p = p0;
  p.g_stage = env_r("NMV_W4R_DBG", 0);
  switch (pl.mt) {
    case 2: return f16 ? w4r_launch_one<F16, 2>(pl, p, s) : w4r_launch_one<BF16, 2>(pl, p, s);
    case 4: return f16 ? w4r_launch_one<F16, 4>(pl, p, s) : w4r_launch_one<BF16, 4>(pl, p, s);
    case 8: return f16 ? w4r_launch_one<F16, 8>(pl, p, s) : w4r_launch_one<BF16, 8>(pl, p, s);
    default: return -1;
  }
}

}  // namespace nmv

#if defined(NMV_W4R_STAMPS)
extern "C" int w4r_dbg_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(nmv::g_w4r_stamps), (size_t)n * 8);
}
#endif

// the opt-in bookkeeping above, for tests/test_capi_symbols.py (host only: no HIP call)
extern "C" int w4r_dbg_lds_optin(unsigned long long* mask, int device) { return nmv::lds_optin_needed(mask, device) ? 1 : 0; }

// workgroups that gave up on a ring slot since the library was loaded (0 in a healthy process); synchronises the device
extern "C" int nmv_w4_ring_timeouts(void) {
  unsigned int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(nmv::g_w4r_timeouts), sizeof(v)) != hipSuccess) return -1;
  return (int)v;
}
