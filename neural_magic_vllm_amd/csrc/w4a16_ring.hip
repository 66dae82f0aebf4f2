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
//     of the workgroup's 128 columns (8 KiB, non-temporal) and its scale row.  A loader keeps D = R - 2 groups in
//     flight behind a counted `s_waitcnt vmcnt`, sums the rows it has landed (the zero-point term, below) and
//     publishes the slot by adding to its FULL word in LDS.
//   * 8 CONSUMER waves (two per SIMD) = 2 column chunks x 2 tile pairs x 2 k-lanes: a wave owns 32 columns
//     (16-column tiles h and h + 2 of its chunk: a gate tile and its up tile under the silu epilogue) x 16 MT rows
//     and, of every group, the two 32-k steps of its k-lane: 2 MT x 2 MFMA 16x16x32 per step on operands read
//     from the slot (activations = A operand: `ds_read_b128`; codes: the lane's 16-byte native vector, expanded in
//     registers exactly once chip-wide).  It waits for a slot by polling the FULL word, and releases it with one
//     LDS add to the slot's FREE word as soon as its last operand read has been issued (LDS serves a wave's
//     operations in order).  No workgroup barrier inside the K loop.
//   * Accumulators: columns on the lanes (D[m][n]: n = lane & 15), so a group's scale is ONE register per tile:
//     acc += s[g, n] * acc_group is a v_fmac per element with no LDS traffic; 32 + 32 accumulator registers.
//   * Zero point as in the stream kernel: sum (128 + q) a - 136 sum a, the second term as one more MFMA k-step
//     per 16 groups at the end (k slots = (group, hi / mid [/ lo] part of -136 * S[g][m])), S summed by the loaders.
//   * The two k-lanes meet in LDS after the loop, transposed on the way: every store of the epilogue (model dtype,
//     silu(gate) * up, fp32 slabs) is a whole 16-byte piece of a row.
// Every spin is bounded: a wave that gives up opens every gate (all counters are set far past any target), so the
// workgroup drains with garbage in its tile and `nmv_w4_ring_timeouts()` reports it.
#include <climits>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "w4a16_common.h"

namespace nmv {

__device__ unsigned int g_w4r_timeouts;

namespace {

constexpr int RW_CONS = 8, RW_LOAD = 4, RW_NTHR = (RW_CONS + RW_LOAD) * 64;
constexpr int RW_GMAX = 32;                 // scale groups per workgroup (S image: RW_GMAX x rows floats)
constexpr int RW_WB = 8192, RW_SB = 256;    // codes / scale row of one group and 128 columns
constexpr int RW_OUT_LD = 132;              // floats per row of the epilogue image (128 + 4: 16-byte aligned, bank-shifted)
constexpr uint32_t RW_OOB = 0x7ffffff0u;    // a voffset no buffer of ours reaches: zeros, no request
constexpr uint32_t RW_SPIN_LIMIT = 1u << 18;
constexpr uint32_t RW_POISON = 0x40000000u;
constexpr int RW_NFLAGS = 64;               // full[32], free[8], ticket, pad

template <int MT> struct RingGeom {
  static constexpr int MP = 16 * MT;
  static constexpr int ACT = MP * 256;
  static constexpr int SLOT = ACT + RW_WB + RW_SB;
  static constexpr int R = MT >= 4 ? 6 : MT == 3 ? 7 : 8;
  static constexpr int D = R - 2;
  static constexpr int S_OFF = R * SLOT;
  static constexpr int F_OFF = S_OFF + RW_GMAX * MP * 4;
  static constexpr int LDS = F_OFF + RW_NFLAGS * 4;
};

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One LDS-DMA piece: 64 lanes x 16 bytes from buffer `rs` (per-lane byte offset `voff`, bounds-checked; uniform `soff`)
// to LDS bytes [lds_addr, lds_addr + 1024) in lane order.  hipcc neither counts this load nor orders LDS reads behind it:
// the caller waits with wait_vm<>.  M0 is saved and restored inside the statement.
template <bool NT>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff, uint32_t lds_addr) {
  uint32_t keep;
  if constexpr (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen nt lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// LDS add without a wait in front of it: LDS serves the operations of a wave in issue order, so the reads issued before it
// have been performed when the add is; the "memory" clobber keeps the compiler from moving them across.
__device__ __forceinline__ void lds_signal(uint32_t lds_addr) {
  const uint32_t one = 1;
  asm volatile("ds_add_u32 %0, %1" ::"v"(lds_addr), "v"(one) : "memory");
}

// wait until *word >= need (one relaxed LDS poll per trip, the whole wave reads the same word)
__device__ __forceinline__ void spin_ge(uint32_t* word, uint32_t need, uint32_t* flags, int lane) {
  for (uint32_t it = 0;; ++it) {
    const uint32_t v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (v >= need) break;
    if (it > RW_SPIN_LIMIT) {   // give up: open every gate, the workgroup drains
      if (lane < RW_GMAX + 8) __hip_atomic_store(flags + lane, RW_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
__device__ __forceinline__ f32x4_t mfma16(uint4 a, uint4 b, f32x4_t c) {
  if constexpr (std::is_same<T, F16>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

}  // namespace

// grid (ceil(N / 128), splits, ceil(M / (16 MT))), 768 threads: waves 0..7 consume, waves 8..11 load.
// p.b: native[kstep][chunk][lane] (uint4), p.s: natural [groups, N]; p.k_per_wg = 128 * (groups per workgroup) <= 4096.
template <typename T, int MT>
__global__ __launch_bounds__(RW_NTHR, 3) void w4a16_ring_kernel(const GemmParams p) {
  using GEO = RingGeom<MT>;
  constexpr int MP = GEO::MP, ACT = GEO::ACT, SLOT = GEO::SLOT, R = GEO::R, D = GEO::D;
  constexpr int PPG = MT + 2;   // DMA pieces per loader wave and group (+ the scale piece, for one of the four)
  constexpr bool IS_F16 = std::is_same<T, F16>::value;
  constexpr float ZPC = W4N<T>::ZPC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int n_chunks = p.N >> 6;
  const int chunk0 = blockIdx.x * 2;
  const int split = blockIdx.y;
  const int m0 = blockIdx.z * MP;
  const int k_wg0 = split * p.k_per_wg;
  const int G = min(p.k_per_wg, p.K - k_wg0) >> 7;     // scale groups of this workgroup (uniform, 1..RW_GMAX)

  float* s_all = reinterpret_cast<float*>(smem + GEO::S_OFF);          // [G][MP]: -ZPC * sum of the group's activations
  uint32_t* flags = reinterpret_cast<uint32_t*>(smem + GEO::F_OFF);
  uint32_t* full = flags;                                               // [RW_GMAX]: loader arrivals per group (4 = landed)
  uint32_t* freec = flags + RW_GMAX;                                    // [R]: consumer releases per slot, cumulative
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  if (tid < RW_NFLAGS) flags[tid] = 0;
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.s), 0, (int)((int64_t)(p.K >> 7) * p.N * 2), 0x00020000);

  // consumer identity (also used by the epilogue)
  const int kl = wave >> 2, ch = (wave >> 1) & 1, h = wave & 1;
  f32x4_t acc[2][MT];
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[jj][t] = zero4;

  if (wave >= RW_CONS) {
    // ------------------------------------------------ loader ------------------------------------------------
    const int lw = wave - RW_CONS;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(p.a), 0, (int)((int64_t)p.M * p.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(p.b), 0, (int)(((int64_t)p.K * p.N) >> 1), 0x00020000);
    // activation piece u of this wave = rows 4 lw + 16 u + (lane >> 4) of the tile (4 rows x 256 bytes); the LDS image is
    // lane-linear, so lane (row, slot) fetches chunk slot ^ (row & 15) of its row: the reader XORs the same way
    const int rl = 4 * lw + (lane >> 4);                                  // row & 15
    uint32_t va[MT];
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      const int row = m0 + rl + 16 * u;
      va[u] = row < p.M ? (uint32_t)(row * p.K * 2) + (uint32_t)(((lane & 15) ^ rl) << 4) : RW_OOB;
    }
    // code pieces q = lw (k-step lw >> 1, chunk lw & 1) and q + 4 (k-step + 2): 1 KiB contiguous each
    const int chq = lw & 1;
    const uint32_t vw = (chunk0 + chq < n_chunks) ? (uint32_t)((chunk0 + chq) * 1024 + lane * 16) : RW_OOB;
    const uint32_t w_row = (uint32_t)n_chunks * 1024u;                    // one k-step of all chunks
    // scale row of the group: 128 columns = 16 lanes x 16 bytes
    const uint32_t vs = (lane < 16 && chunk0 * 64 + lane * 8 < p.N) ? (uint32_t)(chunk0 * 128 + lane * 16) : RW_OOB;
    const uint32_t ones2 = W4N<T>::ONES;

    auto issue = [&](int i) {
      const uint32_t sb = lds0 + (uint32_t)((i % R) * SLOT);
      const uint32_t so_a = (uint32_t)((k_wg0 + i * 128) * 2);
#pragma unroll
      for (int u = 0; u < MT; ++u) dma16<false>(rs_a, va[u], so_a, sb + (uint32_t)((lw + 4 * u) * 1024));
      const uint32_t so_w = (uint32_t)((k_wg0 >> 5) + i * 4 + (lw >> 1)) * w_row;
      dma16<true>(rs_w, vw, so_w, sb + (uint32_t)(ACT + lw * 1024));
      dma16<true>(rs_w, vw, so_w + 2u * w_row, sb + (uint32_t)(ACT + (lw + 4) * 1024));
      if ((i & 3) == lw) {
        if (lane < 16) dma16<false>(rs_s, vs, (uint32_t)(((k_wg0 >> 7) + i) * p.N * 2), sb + (uint32_t)(ACT + RW_WB));
      }
    };
    // group j has landed (the caller waited): sum my rows, publish
    auto publish = [&](int j) {
      const unsigned char* sl = smem + (j % R) * SLOT;
#pragma unroll
      for (int u = 0; u < MT; ++u) {
        const uint4 v = *reinterpret_cast<const uint4*>(sl + (lw + 4 * u) * 1024 + lane * 16);
        float s = T::dot2(v.x, ones2, 0.f);
        s = T::dot2(v.y, ones2, s);
        s = T::dot2(v.z, ones2, s);
        s = T::dot2(v.w, ones2, s);
        s = row16_sum_f(s);
        if ((lane & 15) == 0) s_all[j * MP + rl + 16 * u] = -ZPC * s;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_fetch_add(full + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    for (int i = 0; i < G; ++i) {
      if (i >= R) spin_ge(freec + (i % R), (uint32_t)(RW_CONS * (i / R)), flags, lane);
      issue(i);
      const int j = i - (D - 1);
      if (j >= 0) {
        wait_vm<PPG*(D - 1)>();   // at least PPG (D - 1) pieces were issued after group j's
        publish(j);
      }
    }
    for (int j = max(0, G - (D - 1)); j < G; ++j) {
      const int rem = G - 1 - j;  // groups issued after j
      if (rem >= 5) wait_vm<PPG * 5>();
      else if (rem == 4) wait_vm<PPG * 4>();
      else if (rem == 3) wait_vm<PPG * 3>();
      else if (rem == 2) wait_vm<PPG * 2>();
      else if (rem == 1) wait_vm<PPG>();
      else wait_vm<0>();
      publish(j);
    }
  } else {
    // ----------------------------------------------- consumer -----------------------------------------------
    const uint32_t kmask = __builtin_amdgcn_readfirstlane(W4N<T>::MASK);
    uint32_t kmagic = W4N<T>::MAGIC;
    asm volatile("" : "+v"(kmagic));
    // operand addresses inside a slot, step s = k-step 2 kl + s of the group
    int a_off[2], w_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int ks = 2 * kl + s;
      a_off[s] = r * 256 + (((4 * ks + g) ^ r) << 4);
      w_off[s] = ACT + (ks * 2 + ch) * 1024 + lane * 16;
    }
    const int s_off = ACT + RW_WB + (ch * 64 + 16 * h + r) * 2;   // tile h; tile h + 2 is 64 bytes further

    auto rd_step = [&](const unsigned char* sl, int s, uint4 (&af)[MT], uint4& wv) {
      wv = *reinterpret_cast<const uint4*>(sl + w_off[s]);
#pragma unroll
      for (int t = 0; t < MT; ++t) af[t] = *reinterpret_cast<const uint4*>(sl + a_off[s] + t * 4096);
    };
    auto expand = [&](uint32_t x) -> uint4 {
      constexpr int P0 = W4N<T>::POS;
      return make_uint4(and_or(P0 == 0 ? x : x << P0, kmask, kmagic),
                        and_or(4 >= P0 ? x >> (4 - P0) : x << (P0 - 4), kmask, kmagic),
                        and_or(x >> (8 - P0), kmask, kmagic), and_or(x >> (12 - P0), kmask, kmagic));
    };
    f32x4_t accg[2][MT];
    auto compute = [&](const uint4 (&af)[MT], const uint4& wv, bool first) {
      const uint32_t x0 = h ? wv.y : wv.x, x1 = h ? wv.w : wv.z;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const uint4 w4 = expand(jj ? x1 : x0);
#pragma unroll
        for (int t = 0; t < MT; ++t) accg[jj][t] = mfma16<T>(af[t], w4, first ? zero4 : accg[jj][t]);
      }
    };

    uint4 afA[MT], afB[MT], wvA, wvB;
    uint32_t sc0, sc1;
    spin_ge(full + 0, RW_LOAD, flags, lane);
    {
      const unsigned char* sl = smem;
      sc0 = *reinterpret_cast<const uint16_t*>(sl + s_off);
      sc1 = *reinterpret_cast<const uint16_t*>(sl + s_off + 64);
      rd_step(sl, 0, afA, wvA);
    }
    for (int gi = 0; gi < G; ++gi) {
      const int slot = gi % R;
      const unsigned char* sl = smem + slot * SLOT;
      rd_step(sl, 1, afB, wvB);
      // every operand read of this group has been issued: hand the slot back
      if (lane == 0) lds_signal(lds0 + (uint32_t)(GEO::F_OFF + (RW_GMAX + slot) * 4));
      compute(afA, wvA, true);
      const float sf0 = T::to_float((uint16_t)sc0), sf1 = T::to_float((uint16_t)sc1);
      if (gi + 1 < G) {
        spin_ge(full + gi + 1, RW_LOAD, flags, lane);
        const unsigned char* sn = smem + ((gi + 1) % R) * SLOT;
        sc0 = *reinterpret_cast<const uint16_t*>(sn + s_off);
        sc1 = *reinterpret_cast<const uint16_t*>(sn + s_off + 64);
        rd_step(sn, 0, afA, wvA);
      }
      compute(afB, wvB, false);
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          acc[0][t][reg] = fmaf(sf0, accg[0][t][reg], acc[0][t][reg]);
          acc[1][t][reg] = fmaf(sf1, accg[1][t][reg], acc[1][t][reg]);
        }
    }

    // ---- zero point (k-lane 0 only: S is the sum over the whole group): acc += sum_g (-ZPC S[g, m]) * s[g, n] as MFMA
    //      k-steps of 16 groups: slot 2 e + part of lane group g4 = group gq + 4 g4 + e, part = hi / mid half of the
    //      fp32 value (bf16: a third part in a second MFMA; fp16: the value travels as z / 16 beside 16 s) ----
    if (kl == 0) {
      constexpr int NPASS = IS_F16 ? 1 : 2;
      for (int gq = 0; gq < G; gq += 16) {
        uint32_t sraw[2][4];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int col = (chunk0 + ch) * 64 + 16 * (h + 2 * jj) + r;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int gi = gq + 4 * g + e;
            const bool ok = gi < G && col < p.N;
            sraw[jj][e] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(
                rs_s, ok ? (int)((((k_wg0 >> 7) + gi) * p.N + col) * 2) : (int)RW_OOB, 0, 0);
          }
        }
        uint4 zb[NPASS][MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          uint32_t d[NPASS][4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int gi = gq + 4 * g + e;
            float z = gi < G ? s_all[gi * MP + t * 16 + r] : 0.f;
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
          for (int ps = 0; ps < NPASS; ++ps) zb[ps][t] = make_uint4(d[ps][0], d[ps][1], d[ps][2], d[ps][3]);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          uint32_t sd[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            uint32_t s16 = sraw[jj][e];
            if constexpr (IS_F16) s16 = T::from_float(16.0f * T::to_float((uint16_t)s16));
            sd[e] = s16 | (s16 << 16);
          }
          const uint4 sa = make_uint4(sd[0], sd[1], sd[2], sd[3]);
#pragma unroll
          for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[jj][t] = mfma16<T>(zb[ps][t], sa, acc[jj][t]);
        }
      }
    }
  }

  // ---- the k-lanes meet in LDS, transposed: out[kl][m][n] (row stride RW_OUT_LD floats) ----
  __syncthreads();   // the ring is dead: loaders have drained, consumers have read their last slot
  float* out = reinterpret_cast<float*>(smem);
  if (wave < RW_CONS) {
    float* o = out + kl * (MP * RW_OUT_LD) + (4 * g) * RW_OUT_LD + ch * 64 + 16 * h + r;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) o[(16 * t + reg) * RW_OUT_LD + 32 * jj] = acc[jj][t][reg];
  }
  __syncthreads();

  const int n0 = chunk0 * 64;
  auto tile_sum = [&](int m, int n) -> f32x4_t {
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(out + m * RW_OUT_LD + n);
    const f32x4_t b = *reinterpret_cast<const f32x4_t*>(out + MP * RW_OUT_LD + m * RW_OUT_LD + n);
    return a + b;   // k-lane 0 + k-lane 1
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
  int* ticket_s = reinterpret_cast<int*>(flags + RW_GMAX + 16);
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
  if (M < env_r("NMV_W4R_MIN_M", 33) || M > env_r("NMV_W4R_MAX_M", 64) || N % 64 != 0 || K % 128 != 0) return false;
  W4RingPlan pl;
  pl.mt = env_r("NMV_W4R_MT", M <= 32 ? 2 : M <= 48 ? 3 : 4);
  if (pl.mt < 2 || pl.mt > 4) return false;
  const int mp = 16 * pl.mt;
  pl.m_blocks = (M + mp - 1) / mp;
  pl.n_blocks = (N / 64 + 1) / 2;
  const int groups = K / 128;
  const int base = pl.n_blocks * pl.m_blocks;
  const int target = env_r("NMV_W4R_WGS", 256);
  const int max_splits = unsplit ? 1 : env_r("NMV_W4R_MAX_SPLITS", 8);
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
  pl.lds_bytes = pl.mt == 4 ? RingGeom<4>::LDS : pl.mt == 3 ? RingGeom<3>::LDS : RingGeom<2>::LDS;
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
  dim3 grid(pl.n_blocks, pl.splits, pl.m_blocks), block(RW_NTHR);
  hipLaunchKernelGGL(kern, grid, block, pl.lds_bytes, s, p);
  return 0;
}

int w4r_launch(const W4RingPlan& pl, const GemmParams& p, bool f16, hipStream_t s) {
  if (!p.native) return -1;
  switch (pl.mt) {
    case 2: return f16 ? w4r_launch_one<F16, 2>(pl, p, s) : w4r_launch_one<BF16, 2>(pl, p, s);
    case 3: return f16 ? w4r_launch_one<F16, 3>(pl, p, s) : w4r_launch_one<BF16, 3>(pl, p, s);
    case 4: return f16 ? w4r_launch_one<F16, 4>(pl, p, s) : w4r_launch_one<BF16, 4>(pl, p, s);
    default: return -1;
  }
}

}  // namespace nmv

// workgroups that gave up on a ring slot since the library was loaded (0 in a healthy process); synchronises the device
extern "C" int nmv_w4_ring_timeouts(void) {
  unsigned int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(nmv::g_w4r_timeouts), sizeof(v)) != hipSuccess) return -1;
  return (int)v;
}
