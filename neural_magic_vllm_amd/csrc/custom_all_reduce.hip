// One-shot and two-shot P2P all-reduce over HIP IPC for the tensor-parallel decode step: the gfx950
// analogue of the reference's csrc/custom_all_reduce.cuh:130-250 (cross_device_reduce_1stage / _2stage,
// dispatch rule :442-451), which the reference compiles out on ROCm (torch_bindings.cpp:261).  Decode
// all-reduces a [B, hidden] activation (8 KB .. 1 MB) twice per layer; at that size a ring collective is
// pure latency, while xGMI lets every GPU read every peer's buffer directly (7 links x ~150 GB/s per GPU).
//
// Protocol (all ranks launch the same grid; block b of every rank owns slice b of the message):
//   1. block b copies slice b of its input into its rank's IPC-mapped staging buffer (two buffers,
//      alternating per call, so that no trailing barrier is needed: a rank can only reach call n+2
//      after every peer has passed the flag wait of call n+1, i.e. has finished reading call n);
//   2. system-scope release, then lanes 0..W-1 store the block's call number into slot
//      [b][my_rank] of every rank's flag array and spin (bounded) until slots [b][0..W-1] of the
//      own array carry it; system-scope acquire;
//   3. one-shot: every lane sums slice b of all W staging buffers in rank order (the same order on
//      every rank => bit-identical results everywhere, deterministic) in fp32 and writes the output;
//      two-shot: rank r sums only sub-slice r of slice b into its `tmp` buffer (reduce-scatter), the
//      ranks meet a second time (flags[1]), then every rank copies the W reduced sub-slices
//      (all-gather): each rank pulls 2 x the message over xGMI instead of W x.
// Call numbers live in device memory and are advanced by the kernel itself, so a captured launch
// can be replayed.  The spin is bounded in time (default 2 s): a lost peer sets an error word and the
// call writes NaN -- it neither hangs the GPU nor returns a sum of stale buffers.
// Staging and flags are allocated uncached / fine-grained when the runtime allows it.
#include "common.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <array>
#include <type_traits>
#include <unordered_map>
#include <vector>

// the fused residual-add + RMSNorm below reproduces glue_kernels.hip's roundings: no contraction
#pragma clang fp contract(off)

namespace nmv {

constexpr int AR_MAX_RANKS = 8;
// Every call launches the same number of blocks, whatever the message size: all per-block call
// counters then advance in lockstep, which is what makes "two alternating staging buffers" safe (a
// block count that varied with the size would let the buffer parity of different blocks drift apart
// and a later call's block overwrite a region some peer is still reading for the previous call).
constexpr int AR_MAX_BLOCKS = 16;
constexpr int AR_THREADS = 512;
constexpr uint64_t AR_SPIN_TICKS = 200ull * 1000 * 1000;  // s_memrealtime runs at 100 MHz: give up after 2 s

struct ArComm {
  // [phase][block][source rank] = call number; phase 1 is the second rendezvous of the two-shot form,
  // phase 2 the closing one of the registered-buffer calls (peers read the caller's own tensor there, which
  // must not be overwritten before every peer is done: end_sync, custom_all_reduce.cuh:167-196)
  uint32_t flags[3][AR_MAX_BLOCKS][AR_MAX_RANKS];
  uint32_t seq[AR_MAX_BLOCKS];                  // this rank's call number per block
  uint32_t error;                               // set when a spin ran out
  uint32_t pad[47];
};
static_assert(sizeof(ArComm) % 256 == 0, "staging stays 256-byte aligned");

struct ArPeers {
  ArComm* comm[AR_MAX_RANKS];
  uint8_t* data[AR_MAX_RANKS];  // staging: 2 x max_bytes each (alternating per call)
  uint8_t* tmp[AR_MAX_RANKS];   // two-shot: the owner's reduced slices, 2 x max_bytes each
};

struct ArState {
  int rank, world;
  int64_t max_bytes;
  void* base;        // one allocation: ArComm + 2 staging buffers + 2 reduced-slice buffers
  size_t alloc_bytes;
  hipIpcMemHandle_t handle;
  void* peer_base[AR_MAX_RANKS];
  ArPeers peers;
  bool opened;
  int force_algo;        // 0 = the reference's size rule, 1 = one-shot, 2 = two-shot
  uint64_t spin_ticks;   // bounded flag wait, in 10 ns ticks
};

// Rendezvous `phase` of call `seq` for block b: everything this block has stored so far becomes
// visible system-wide, lanes 0..W-1 signal every rank and wait (bounded) for every rank.  Returns
// false when a wait ran out: the error word is set and *dead (LDS, zeroed at kernel start) says so
// to the whole block -- the caller then poisons its output instead of reducing stale buffers.
__device__ __forceinline__ bool ar_rendezvous(const ArPeers& peers, ArComm* mine, int rank, int world, int b,
                                              uint32_t seq, int phase, uint64_t spin_ticks, int* dead) {
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x < world) {
    const int q = threadIdx.x;
    __hip_atomic_store(&peers.comm[q]->flags[phase][b][rank], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // fail fast: once a wait of this communicator has run out, no later rendezvous waits again -- a step holds ~65
    // collectives, and a lost peer must cost one bound, not 65 (the error word is never cleared: the communicator is rebuilt)
    if (__hip_atomic_load(&mine->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) *dead = 1;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (*dead == 0 &&
           (int32_t)(__hip_atomic_load(&mine->flags[phase][b][q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > spin_ticks) {
        __hip_atomic_store(&mine->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        *dead = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  // system-scope acquire on every lane: drop whatever the caches hold of the peers' buffers
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  return *dead == 0;
}

template <typename T>
__device__ __forceinline__ uint4 ar_nan16() {  // eight quiet NaNs of the model dtype
  const uint32_t n = std::is_same<T, F16>::value ? 0x7E007E00u : 0x7FC07FC0u;
  return make_uint4(n, n, n, n);
}

// fp32 sum of vector v of every rank's staging buffer, rank order, rounded to the model dtype
template <typename T>
__device__ __forceinline__ uint4 ar_sum_vec(const ArPeers& peers, int world, int64_t parity_off, int64_t v) {
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int q = 0; q < world; ++q) {
    const uint4 x = reinterpret_cast<const uint4*>(peers.data[q] + parity_off)[v];
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[2 * j] += lo_f<T>(xs[j]);
      acc[2 * j + 1] += hi_f<T>(xs[j]);
    }
  }
  uint4 o = make_uint4(T::pack2(acc[0], acc[1]), T::pack2(acc[2], acc[3]), T::pack2(acc[4], acc[5]),
                       T::pack2(acc[6], acc[7]));
  // the rounding must survive when the caller keeps computing on the value (glue_kernels.hip: pin())
  asm volatile("" : "+v"(o.x), "+v"(o.y), "+v"(o.z), "+v"(o.w));
  return o;
}

// this block's slice [v0, v1) of the local input -> this rank's staging buffer.  slab != null: the
// input is still the fp32 split-K slabs of the row-parallel GEMM (nmv_gptq_marlin_gemm_partial), summed
// in split order from +0 and rounded to the model dtype, as that GEMM's own last pass would
template <typename T>
__device__ __forceinline__ void ar_stage_slice(uint4* stage, const uint16_t* inp, const float* slab, int splits,
                                               int64_t slab_stride, int64_t v0, int64_t v1) {
  if (slab != nullptr) {
    for (int64_t v = v0 + threadIdx.x; v < v1; v += AR_THREADS) {
      f32x4_t lo4 = {0.f, 0.f, 0.f, 0.f}, hi4 = {0.f, 0.f, 0.f, 0.f};
      for (int sp = 0; sp < splits; ++sp) {
        lo4 += *reinterpret_cast<const f32x4_t*>(slab + sp * slab_stride + v * 8);
        hi4 += *reinterpret_cast<const f32x4_t*>(slab + sp * slab_stride + v * 8 + 4);
      }
      stage[v] = make_uint4(T::pack2(lo4[0], lo4[1]), T::pack2(lo4[2], lo4[3]), T::pack2(hi4[0], hi4[1]),
                            T::pack2(hi4[2], hi4[3]));
    }
  } else {
    const uint4* src = reinterpret_cast<const uint4*>(inp);
    for (int64_t v = v0 + threadIdx.x; v < v1; v += AR_THREADS) stage[v] = src[v];
  }
}

// MODE 0: one-shot all-reduce (every rank sums slice b of all W staging buffers: W x message read per
//         rank) -- the analogue of cross_device_reduce_1stage, custom_all_reduce.cuh:130-196.
// MODE 1: all-gather: slice b of every rank's buffer is copied to out[q * n_vec + slice] (small per-rank
//         records, e.g. the per-shard argmax of the sampler).
// MODE 2: two-shot all-reduce (cross_device_reduce_2stage, :203-250): rank r sums only sub-slice r of
//         slice b (reduce-scatter into its own `tmp` buffer), second rendezvous, every rank copies the
//         W reduced sub-slices (all-gather): 2 x message read per rank instead of W x.  The owner sums in
//         the same rank order and rounds once, so the result has the bits of the one-shot form.
template <typename T, int MODE>
__global__ __launch_bounds__(AR_THREADS) void p2p_all_reduce_kernel(ArPeers peers, int rank, int world,
                                                                    const uint16_t* __restrict__ inp,
                                                                    uint16_t* __restrict__ out,
                                                                    int64_t n_vec /* 16-byte vectors */,
                                                                    int64_t buf_bytes, uint64_t spin_ticks,
                                                                    const float* __restrict__ slab = nullptr,
                                                                    int splits = 0, int64_t slab_stride = 0) {
  __shared__ int dead;
  if (threadIdx.x == 0) dead = 0;
  const int b = blockIdx.x;
  ArComm* mine = peers.comm[rank];
  const uint32_t seq = mine->seq[b] + 1;  // only this block touches seq[b]
  const int64_t parity_off = (seq & 1) ? buf_bytes : 0;
  const int64_t per = (n_vec + gridDim.x - 1) / gridDim.x;
  const int64_t v0 = min((int64_t)b * per, n_vec), v1 = min(v0 + per, n_vec);
  uint4* dst = reinterpret_cast<uint4*>(out);
  // 1. my slice -> my staging buffer
  ar_stage_slice<T>(reinterpret_cast<uint4*>(peers.data[rank] + parity_off), inp, slab, splits, slab_stride, v0, v1);
  // 2. signal every rank (myself included), wait for every rank
  bool ok = ar_rendezvous(peers, mine, rank, world, b, seq, 0, spin_ticks, &dead);
  if constexpr (MODE == 1) {
    if (ok)
      for (int q = 0; q < world; ++q)
        for (int64_t v = v0 + threadIdx.x; v < v1; v += AR_THREADS)
          dst[q * n_vec + v] = reinterpret_cast<const uint4*>(peers.data[q] + parity_off)[v];
    else
      for (int q = 0; q < world; ++q)
        for (int64_t v = v0 + threadIdx.x; v < v1; v += AR_THREADS) dst[q * n_vec + v] = make_uint4(~0u, ~0u, ~0u, ~0u);
    if (threadIdx.x == 0) mine->seq[b] = seq;
    return;
  }
  if constexpr (MODE == 0) {
    // 3. sum slice b of every rank's buffer, rank order, fp32
    if (ok)
      for (int64_t v = v0 + threadIdx.x; v < v1; v += AR_THREADS) dst[v] = ar_sum_vec<T>(peers, world, parity_off, v);
  } else {
    const int64_t sub = (v1 - v0 + world - 1) / world;
    if (ok) {
      // 3a. reduce-scatter: my sub-slice of slice b -> my tmp buffer
      uint4* tmine = reinterpret_cast<uint4*>(peers.tmp[rank] + parity_off);
      const int64_t s0 = min(v0 + rank * sub, v1), s1 = min(s0 + sub, v1);
      for (int64_t v = s0 + threadIdx.x; v < s1; v += AR_THREADS) tmine[v] = ar_sum_vec<T>(peers, world, parity_off, v);
      ok = ar_rendezvous(peers, mine, rank, world, b, seq, 1, spin_ticks, &dead);
    }
    if (ok)  // 3b. all-gather of the reduced sub-slices
      for (int64_t v = v0 + threadIdx.x; v < v1; v += AR_THREADS)
        dst[v] = reinterpret_cast<const uint4*>(peers.tmp[(v - v0) / sub] + parity_off)[v];
  }
  if (!ok)  // a peer never arrived: NaN, not a sum of stale buffers
    for (int64_t v = v0 + threadIdx.x; v < v1; v += AR_THREADS) dst[v] = ar_nan16<T>();
  if (threadIdx.x == 0) mine->seq[b] = seq;
}

// ---------------------------------------------------------------------------------------------
// all-reduce + residual-add + RMSNorm in one launch (tensor-parallel decode: o_proj / down_proj
// -> all-reduce -> fused_add_rms_norm is three launches per projection otherwise).  Same protocol;
// the slices are whole rows (block b owns rows [b R, (b+1) R)), so after the rendezvous a block holds
// complete reduced rows and can finish them: sum over ranks (fp32, rank order) rounded to the model
// dtype = what nmv_ar_all_reduce writes; + residual, rounded, stored back; variance with the per-lane
// accumulation order of rms_norm_reg_kernel's 256-lane form (replayed through LDS, as its wide form
// does); x * rsqrt rounded, times weight.  Bit-identical to the three launches.  TWO: the two-shot
// form -- the block's rows are reduced sub-slice by sub-slice by their owner ranks first.
template <typename T>
__device__ __forceinline__ float ar_rnd(float f) {
  uint32_t b = T::from_float(f);
  asm volatile("" : "+v"(b));  // keep the rounding (see glue_kernels.hip: pin())
  return T::to_float((uint16_t)b);
}

template <typename T, bool TWO>
__global__ __launch_bounds__(AR_THREADS) void p2p_all_reduce_norm_kernel(
    ArPeers peers, int rank, int world, const uint16_t* __restrict__ inp, const float* __restrict__ slab,
    int splits, int64_t slab_stride, uint16_t* __restrict__ residual, const uint16_t* __restrict__ weight,
    uint16_t* __restrict__ out, float epsilon, int rows, int hidden, int64_t buf_bytes, uint64_t spin_ticks) {
  __shared__ float fold[2 * AR_THREADS][4];
  __shared__ float red[AR_THREADS / 64];
  __shared__ int dead;
  if (threadIdx.x == 0) dead = 0;
  const int b = blockIdx.x;
  ArComm* mine = peers.comm[rank];
  const uint32_t seq = mine->seq[b] + 1;
  const int64_t parity_off = (seq & 1) ? buf_bytes : 0;
  const int hv = hidden / 8;                         // 16-byte vectors per row (<= 2 * AR_THREADS)
  const int per = (rows + gridDim.x - 1) / gridDim.x;
  const int r0 = min(b * per, rows), r1 = min(r0 + per, rows);
  const int64_t v0 = (int64_t)r0 * hv, v1 = (int64_t)r1 * hv;
  // 1. my rows -> my staging buffer
  ar_stage_slice<T>(reinterpret_cast<uint4*>(peers.data[rank] + parity_off), inp, slab, splits, slab_stride, v0, v1);
  // 2. rendezvous (+ reduce-scatter and second rendezvous in the two-shot form)
  bool ok = ar_rendezvous(peers, mine, rank, world, b, seq, 0, spin_ticks, &dead);
  const int64_t sub = TWO ? max((v1 - v0 + world - 1) / world, (int64_t)1) : 1;
  if constexpr (TWO) {
    if (ok) {
      uint4* tmine = reinterpret_cast<uint4*>(peers.tmp[rank] + parity_off);
      const int64_t s0 = min(v0 + rank * sub, v1), s1 = min(s0 + sub, v1);
      for (int64_t v = s0 + threadIdx.x; v < s1; v += AR_THREADS) tmine[v] = ar_sum_vec<T>(peers, world, parity_off, v);
      ok = ar_rendezvous(peers, mine, rank, world, b, seq, 1, spin_ticks, &dead);
    }
  }
  // 3. finish my rows
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = r0; r < r1; ++r) {
    uint32_t z[2][4];
    float terms[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = threadIdx.x + i * AR_THREADS;
      const bool in = v < hv;
      const int64_t e = (int64_t)r * hv + (in ? v : 0);
      // the all-reduce output of this vector, rounded to the model dtype
      uint4 s4;
      if (!ok) s4 = ar_nan16<T>();
      else if constexpr (TWO) s4 = reinterpret_cast<const uint4*>(peers.tmp[(e - v0) / sub] + parity_off)[e];
      else s4 = ar_sum_vec<T>(peers, world, parity_off, e);
      const uint32_t ss[4] = {s4.x, s4.y, s4.z, s4.w};
      const uint4 rs4 = ld16(residual + e * 8);
      const uint32_t rs[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // z = x + residual rounded (layernorm_kernels.cu:271-274)
        const float lo = ar_rnd<T>(lo_f<T>(ss[j]) + lo_f<T>(rs[j]));
        const float hi = ar_rnd<T>(hi_f<T>(ss[j]) + hi_f<T>(rs[j]));
        z[i][j] = T::pack2(lo, hi);
        terms[i][j] = in ? lo * lo + hi * hi : 0.f;
      }
      if (in) st16(residual + e * 8, make_uint4(z[i][0], z[i][1], z[i][2], z[i][3]));
    }
    // variance in the 256-lane form's order: lane t adds the four addends of vector t, t + 256, ...
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) fold[threadIdx.x + i * AR_THREADS][j] = terms[i][j];
    __syncthreads();
    float var = 0.f;
    if (threadIdx.x < 256) {
      for (int qv = threadIdx.x; qv < hv; qv += 256)
#pragma unroll
        for (int j = 0; j < 4; ++j) var += fold[qv][j];
    }
    var = wave_sum(var);
    if (lane == 0) red[wave] = var;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < AR_THREADS / 64; ++w) tot += red[w];
    const float sc = rsqrtf(tot / hidden + epsilon);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = threadIdx.x + i * AR_THREADS;
      if (v >= hv) continue;
      const uint4 w4 = ld16(weight + v * 8);
      const uint32_t ws[4] = {w4.x, w4.y, w4.z, w4.w};
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // ((scalar_t)(x * s_variance)) * weight  (layernorm_kernels.cu:41-42)
        const float lo = ar_rnd<T>(lo_f<T>(z[i][j]) * sc) * lo_f<T>(ws[j]);
        const float hi = ar_rnd<T>(hi_f<T>(z[i][j]) * sc) * hi_f<T>(ws[j]);
        o[j] = T::pack2(lo, hi);
      }
      st16(out + ((int64_t)r * hv + v) * 8, make_uint4(o[0], o[1], o[2], o[3]));
    }
  }
  if (threadIdx.x == 0) mine->seq[b] = seq;
}

// ---------------------------------------------------------------------------------------------
// Registered-buffer form: the `_C_custom_ar` protocol of the reference (torch_bindings.cpp:262-294,
// custom_all_reduce.cuh:253-470, vllm/distributed/device_communicators/custom_all_reduce.py).  The caller owns
// every byte: `meta` (this rank's flag block, followed by the two-shot scratch) and the input tensors, whose
// IPC handles the ranks exchange (register_buffer / register_graph_buffers); peers sum the caller's tensors in
// place -- no staging copy -- so every call ends with a closing rendezvous.  RegPtrs is one entry of the
// caller's `rank_data` tensor: the address of one registered buffer on every rank.
struct RegPtrs {
  const void* p[AR_MAX_RANKS];
};

template <typename T>
__device__ __forceinline__ uint4 reg_sum_vec(const RegPtrs* __restrict__ ptrs, int world, int64_t v) {
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int q = 0; q < world; ++q) {
    const uint4 x = reinterpret_cast<const uint4*>(ptrs->p[q])[v];
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[2 * j] += lo_f<T>(xs[j]);
      acc[2 * j + 1] += hi_f<T>(xs[j]);
    }
  }
  return make_uint4(T::pack2(acc[0], acc[1]), T::pack2(acc[2], acc[3]), T::pack2(acc[4], acc[5]),
                    T::pack2(acc[6], acc[7]));
}
template <>
__device__ __forceinline__ uint4 reg_sum_vec<float>(const RegPtrs* __restrict__ ptrs, int world, int64_t v) {
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  for (int q = 0; q < world; ++q) acc += reinterpret_cast<const f32x4_t*>(ptrs->p[q])[v];
  return __builtin_bit_cast(uint4, acc);
}

template <typename T, bool TWO>
__global__ __launch_bounds__(AR_THREADS) void reg_all_reduce_kernel(ArPeers peers, const RegPtrs* __restrict__ ptrs,
                                                                    int rank, int world, void* __restrict__ out,
                                                                    int64_t n_vec, uint64_t spin_ticks) {
  __shared__ int dead;
  if (threadIdx.x == 0) dead = 0;
  const int b = blockIdx.x;
  ArComm* mine = peers.comm[rank];
  const uint32_t seq = mine->seq[b] + 1;
  const int64_t per = (n_vec + gridDim.x - 1) / gridDim.x;
  const int64_t v0 = min((int64_t)b * per, n_vec), v1 = min(v0 + per, n_vec);
  uint4* dst = reinterpret_cast<uint4*>(out);
  // opening rendezvous: every rank has reached this call, so its input (written earlier in its stream) is complete
  bool ok = ar_rendezvous(peers, mine, rank, world, b, seq, 0, spin_ticks, &dead);
  if constexpr (!TWO) {
    if (ok)
      for (int64_t v = v0 + threadIdx.x; v < v1; v += AR_THREADS) dst[v] = reg_sum_vec<T>(ptrs, world, v);
  } else {
    const int64_t sub = max((v1 - v0 + world - 1) / world, (int64_t)1);
    if (ok) {
      uint4* tmine = reinterpret_cast<uint4*>(peers.tmp[rank]);
      const int64_t s0 = min(v0 + rank * sub, v1), s1 = min(s0 + sub, v1);
      for (int64_t v = s0 + threadIdx.x; v < s1; v += AR_THREADS) tmine[v] = reg_sum_vec<T>(ptrs, world, v);
      ok = ar_rendezvous(peers, mine, rank, world, b, seq, 1, spin_ticks, &dead);
    }
    if (ok)
      for (int64_t v = v0 + threadIdx.x; v < v1; v += AR_THREADS)
        dst[v] = reinterpret_cast<const uint4*>(peers.tmp[(v - v0) / sub])[v];
  }
  if (!ok) {
    const uint4 nan = std::is_same<T, float>::value ? make_uint4(0x7FC00000u, 0x7FC00000u, 0x7FC00000u, 0x7FC00000u)
                                                    : ar_nan16<typename std::conditional<std::is_same<T, float>::value, BF16, T>::type>();
    for (int64_t v = v0 + threadIdx.x; v < v1; v += AR_THREADS) dst[v] = nan;
  } else {
    // closing rendezvous: inputs and scratch may be rewritten only after every peer has read them
    ar_rendezvous(peers, mine, rank, world, b, seq, 2, spin_ticks, &dead);
  }
  if (threadIdx.x == 0) mine->seq[b] = seq;
}

using IpcKey = std::array<uint8_t, sizeof(hipIpcMemHandle_t)>;

struct RegState {
  int rank, world, full_link;
  ArPeers peers;                               // comm[q] = rank q's meta, tmp[q] = the scratch behind it
  RegPtrs* rd_next;                            // next free entry of the caller's rank_data tensor (device)
  RegPtrs* rd_end;
  std::unordered_map<const void*, RegPtrs*> buffers;   // registered input -> its entry
  std::vector<void*> graph_unreg;              // inputs seen during graph capture, registered afterwards
  std::map<IpcKey, char*> opened;              // IPC handle -> mapped base (one mapping per allocation)
  uint64_t spin_ticks;
  int force_algo;
};

static bool reg_open(RegState* st, const void* handle, char** base) {
  IpcKey key;
  std::memcpy(key.data(), handle, key.size());
  auto it = st->opened.find(key);
  if (it == st->opened.end()) {
    void* ptr = nullptr;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof(h));
    const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      ::nmv::set_error("hipIpcOpenMemHandle: %s", hipGetErrorString(e));
      return false;
    }
    it = st->opened.emplace(key, (char*)ptr).first;
  }
  *base = it->second;
  return true;
}

// the reference's dispatch rule (custom_all_reduce.cuh:442-451): one-shot for two ranks and for small
// messages (< 512 KB up to 4 ranks, < 256 KB up to 8), two-shot beyond
static inline bool ar_two_shot(const ArState* st, int64_t bytes) {
  if (st->force_algo) return st->force_algo == 2;
  if (st->world == 2) return false;
  if ((st->world <= 4 && bytes < 512 * 1024) || (st->world <= 8 && bytes < 256 * 1024)) return false;
  return true;
}

}  // namespace nmv

using namespace nmv;

// bound of a flag wait: 2 s, or NMV_CUSTOM_AR_TIMEOUT_MS (rehearsals whose ranks time-share one GPU need more)
static uint64_t ar_default_spin_ticks() {
  const char* e = getenv("NMV_CUSTOM_AR_TIMEOUT_MS");
  const long long ms = e ? atoll(e) : 0;
  return ms > 0 ? (uint64_t)ms * 100000ull : AR_SPIN_TICKS;   // s_memrealtime: 100 MHz
}

#define AR_HIP(call)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      ::nmv::set_error("custom_all_reduce: %s failed: %s", #call, hipGetErrorString(e_));   \
      return NMV_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

extern "C" int nmv_ar_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

/* allocate this rank's comm block + staging (2 x max_bytes) + reduced-slice buffers (2 x max_bytes) on the
 * current device and export it */
extern "C" int nmv_ar_create(void** state_out, int rank, int world, int64_t max_bytes, void* handle_out) {
  NMV_CHECK(world >= 2 && world <= AR_MAX_RANKS && rank >= 0 && rank < world, "custom_all_reduce: bad rank / world");
  NMV_CHECK(max_bytes > 0 && max_bytes % 256 == 0, "custom_all_reduce: max_bytes must be a multiple of 256");
  ArState* st = new ArState();
  std::memset(st, 0, sizeof(ArState));
  st->rank = rank; st->world = world; st->max_bytes = max_bytes;
  st->spin_ticks = ar_default_spin_ticks();
  st->alloc_bytes = sizeof(ArComm) + 4 * (size_t)max_bytes;
  hipError_t e = hipExtMallocWithFlags(&st->base, st->alloc_bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipMalloc(&st->base, st->alloc_bytes);
  }
  if (e != hipSuccess) {
    delete st;
    ::nmv::set_error("custom_all_reduce: allocation failed: %s", hipGetErrorString(e));
    return NMV_ERR_HIP;
  }
  AR_HIP(hipMemset(st->base, 0, sizeof(ArComm)));
  AR_HIP(hipDeviceSynchronize());
  AR_HIP(hipIpcGetMemHandle(&st->handle, st->base));
  std::memcpy(handle_out, &st->handle, sizeof(hipIpcMemHandle_t));
  *state_out = st;
  return NMV_OK;
}

/* handles: world x nmv_ar_handle_bytes() bytes, rank order (the own slot is ignored) */
extern "C" int nmv_ar_open(void* state, const void* handles) {
  ArState* st = (ArState*)state;
  NMV_CHECK(st != nullptr && !st->opened, "custom_all_reduce: bad state");
  for (int q = 0; q < st->world; ++q) {
    if (q == st->rank) {
      st->peer_base[q] = st->base;
    } else {
      hipIpcMemHandle_t h;
      std::memcpy(&h, (const uint8_t*)handles + (size_t)q * sizeof(h), sizeof(h));
      AR_HIP(hipIpcOpenMemHandle(&st->peer_base[q], h, hipIpcMemLazyEnablePeerAccess));
      // touch the mapping through the runtime first: a copy that cannot reach the peer's memory
      // returns an error here instead of a fault inside the first kernel
      uint32_t probe[16];
      AR_HIP(hipMemcpy(probe, st->peer_base[q], sizeof(probe), hipMemcpyDeviceToHost));
    }
    st->peers.comm[q] = (ArComm*)st->peer_base[q];
    st->peers.data[q] = (uint8_t*)st->peer_base[q] + sizeof(ArComm);
    st->peers.tmp[q] = st->peers.data[q] + 2 * (size_t)st->max_bytes;
  }
  st->opened = true;
  return NMV_OK;
}

/* algo: 0 = the reference's size rule (custom_all_reduce.cuh:442-451), 1 = always one-shot, 2 = always
 * two-shot (tests, measurements).  Every rank must use the same setting. */
extern "C" int nmv_ar_set_algo(void* state, int algo) {
  ArState* st = (ArState*)state;
  NMV_CHECK(st != nullptr && algo >= 0 && algo <= 2, "custom_all_reduce: algo must be 0, 1 or 2");
  st->force_algo = algo;
  return NMV_OK;
}

/* how long a flag wait may last before the call gives up, sets the error word and writes NaN (default 2 s) */
extern "C" int nmv_ar_set_timeout_ms(void* state, int64_t ms) {
  ArState* st = (ArState*)state;
  NMV_CHECK(st != nullptr && ms > 0, "custom_all_reduce: bad timeout");
  st->spin_ticks = (uint64_t)ms * 100000ull;  // s_memrealtime: 100 MHz
  return NMV_OK;
}

/* 1 when the call would take the two-shot form (reduce-scatter + all-gather), 0 for one-shot */
extern "C" int nmv_ar_is_two_shot(void* state, int64_t bytes) {
  ArState* st = (ArState*)state;
  return (st != nullptr && ar_two_shot(st, bytes)) ? 1 : 0;
}

template <typename T>
static void ar_launch(ArState* st, const void* inp, void* out, int64_t n_vec, const float* slab, int splits,
                      int64_t slab_stride, hipStream_t s) {
  if (ar_two_shot(st, n_vec * 16))
    hipLaunchKernelGGL((p2p_all_reduce_kernel<T, 2>), dim3(AR_MAX_BLOCKS), dim3(AR_THREADS), 0, s, st->peers,
                       st->rank, st->world, (const uint16_t*)inp, (uint16_t*)out, n_vec, st->max_bytes,
                       st->spin_ticks, slab, splits, slab_stride);
  else
    hipLaunchKernelGGL((p2p_all_reduce_kernel<T, 0>), dim3(AR_MAX_BLOCKS), dim3(AR_THREADS), 0, s, st->peers,
                       st->rank, st->world, (const uint16_t*)inp, (uint16_t*)out, n_vec, st->max_bytes,
                       st->spin_ticks, slab, splits, slab_stride);
}

extern "C" int nmv_ar_all_reduce(void* state, const void* inp, void* out, int64_t numel, nmv_dtype_t dtype,
                                 void* stream) {
  ArState* st = (ArState*)state;
  NMV_CHECK(st != nullptr && st->opened, "custom_all_reduce: not initialised");
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16, "custom_all_reduce: fp16 / bf16 only");
  const int64_t bytes = numel * 2;
  NMV_CHECK(bytes > 0 && bytes % 16 == 0 && bytes <= st->max_bytes,
            "custom_all_reduce: message must be a multiple of 16 bytes and <= %lld", (long long)st->max_bytes);
  NMV_CHECK((((uintptr_t)inp | (uintptr_t)out) & 15) == 0, "custom_all_reduce: 16-byte aligned tensors");
  if (dtype == NMV_F16) ar_launch<F16>(st, inp, out, bytes / 16, nullptr, 0, 0, (hipStream_t)stream);
  else ar_launch<BF16>(st, inp, out, bytes / 16, nullptr, 0, 0, (hipStream_t)stream);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

/* all-reduce whose local input is the sum of `splits` fp32 slabs [splits, numel] (deferred split-K of
 * the row-parallel projection): bit-identical to nmv_gptq_marlin_gemm followed by nmv_ar_all_reduce */
extern "C" int nmv_ar_all_reduce_partial(void* state, const float* slab, int splits, void* out,
                                         int64_t numel, nmv_dtype_t dtype, void* stream) {
  ArState* st = (ArState*)state;
  NMV_CHECK(st != nullptr && st->opened, "custom_all_reduce: not initialised");
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16, "custom_all_reduce: fp16 / bf16 only");
  NMV_CHECK(slab != nullptr && splits >= 1, "custom_all_reduce: bad slabs");
  const int64_t bytes = numel * 2;
  NMV_CHECK(bytes > 0 && bytes % 16 == 0 && bytes <= st->max_bytes,
            "custom_all_reduce: message must be a multiple of 16 bytes and <= %lld", (long long)st->max_bytes);
  NMV_CHECK((((uintptr_t)slab | (uintptr_t)out) & 15) == 0, "custom_all_reduce: 16-byte aligned tensors");
  if (dtype == NMV_F16) ar_launch<F16>(st, nullptr, out, bytes / 16, slab, splits, numel, (hipStream_t)stream);
  else ar_launch<BF16>(st, nullptr, out, bytes / 16, slab, splits, numel, (hipStream_t)stream);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

/* all-reduce (input: inp [rows, hidden] in the model dtype, or -- inp == NULL -- the fp32 slabs
 * [splits, rows, hidden]) + fused_add_rms_norm in one launch: residual += all_reduce(x) (in place),
 * out = rms_norm(residual) * weight.  hidden % 8 == 0, hidden <= 8192.  Bit-identical to
 * nmv_ar_all_reduce(_partial) + nmv_fused_add_rms_norm. */
extern "C" int nmv_ar_all_reduce_add_rms_norm(void* state, const void* inp, const float* slab, int splits,
                                              void* residual, const void* weight, void* out,
                                              float epsilon, int rows, int hidden, nmv_dtype_t dtype,
                                              void* stream) {
  ArState* st = (ArState*)state;
  NMV_CHECK(st != nullptr && st->opened, "custom_all_reduce: not initialised");
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16, "custom_all_reduce: fp16 / bf16 only");
  NMV_CHECK((inp != nullptr) != (slab != nullptr) && (slab == nullptr || splits >= 1),
            "all_reduce_add_rms_norm: exactly one of inp / slab");
  NMV_CHECK(rows > 0 && hidden % 8 == 0 && hidden <= 8192 && (int64_t)rows * hidden * 2 <= st->max_bytes,
            "all_reduce_add_rms_norm: hidden must be a multiple of 8, <= 8192; message <= %lld bytes",
            (long long)st->max_bytes);
  NMV_CHECK((((uintptr_t)inp | (uintptr_t)slab | (uintptr_t)residual | (uintptr_t)weight | (uintptr_t)out) & 15) == 0,
            "all_reduce_add_rms_norm: 16-byte aligned tensors");
  hipStream_t s = (hipStream_t)stream;
  const int64_t stride = (int64_t)rows * hidden;
  const bool two = ar_two_shot(st, stride * 2);
#define NMV_AR_NORM(T_, TWO_)                                                                                   \
  hipLaunchKernelGGL((p2p_all_reduce_norm_kernel<T_, TWO_>), dim3(AR_MAX_BLOCKS), dim3(AR_THREADS), 0, s,       \
                     st->peers, st->rank, st->world, (const uint16_t*)inp, slab, splits, stride,                \
                     (uint16_t*)residual, (const uint16_t*)weight, (uint16_t*)out, epsilon, rows, hidden,       \
                     st->max_bytes, st->spin_ticks)
  if (dtype == NMV_F16) { if (two) NMV_AR_NORM(F16, true); else NMV_AR_NORM(F16, false); }
  else { if (two) NMV_AR_NORM(BF16, true); else NMV_AR_NORM(BF16, false); }
#undef NMV_AR_NORM
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

/* out[q * bytes_per_rank ...] = rank q's inp (bytes_per_rank % 16 == 0, <= max_bytes) */
extern "C" int nmv_ar_all_gather(void* state, const void* inp, void* out, int64_t bytes_per_rank,
                                 void* stream) {
  ArState* st = (ArState*)state;
  NMV_CHECK(st != nullptr && st->opened, "custom_all_reduce: not initialised");
  NMV_CHECK(bytes_per_rank > 0 && bytes_per_rank % 16 == 0 && bytes_per_rank <= st->max_bytes,
            "custom all_gather: record must be a multiple of 16 bytes and <= %lld", (long long)st->max_bytes);
  NMV_CHECK((((uintptr_t)inp | (uintptr_t)out) & 15) == 0, "custom all_gather: 16-byte aligned tensors");
  const int64_t n_vec = bytes_per_rank / 16;
  hipLaunchKernelGGL((p2p_all_reduce_kernel<BF16, 1>), dim3(AR_MAX_BLOCKS), dim3(AR_THREADS), 0,
                     (hipStream_t)stream, st->peers, st->rank, st->world, (const uint16_t*)inp,
                     (uint16_t*)out, n_vec, st->max_bytes, st->spin_ticks);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

/* 1 when a bounded flag wait ran out on this rank since creation (synchronises the device): the call
 * that timed out wrote NaN, the communicator's call counters are out of step with its peers' and it must
 * not be used again */
extern "C" int nmv_ar_error(void* state) {
  ArState* st = (ArState*)state;
  if (st == nullptr) return 1;
  uint32_t err = 1;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpy(&err, &((ArComm*)st->base)->error, 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  return (int)err;
}

extern "C" int nmv_ar_destroy(void* state) {
  ArState* st = (ArState*)state;
  if (st == nullptr) return NMV_OK;
  (void)hipDeviceSynchronize();
  if (st->opened)
    for (int q = 0; q < st->world; ++q)
      if (q != st->rank && st->peer_base[q]) (void)hipIpcCloseMemHandle(st->peer_base[q]);
  if (st->base) (void)hipFree(st->base);
  delete st;
  return NMV_OK;
}

/* ------------------------------------------------------------------------------------------------
 * `_C_custom_ar` (torch_bindings.cpp:262-294): the registered-buffer protocol.  The Python side
 * (neural_magic_vllm_amd/_torch_bindings.py) binds the reference's op names to these entry points. */
extern "C" int64_t nmv_car_meta_size(void) { return (int64_t)sizeof(ArComm); }

/* The flag block of the registered-buffer protocol as an UNCACHED allocation of the current device (fine-grained, as the
 * staging communicator's: the peers' flag stores and this rank's polls then never sit in an L2 that another device cannot
 * see), zero-filled, with its IPC handle (nmv_ar_handle_bytes() bytes).  The reference keeps `meta` in an ordinary torch
 * tensor (custom_all_reduce.py:118-127): the Python class wraps this pointer in one.  Falls back to hipMalloc. */
extern "C" int nmv_car_meta_alloc(int64_t nbytes, void** ptr_out, void* handle_out) {
  NMV_CHECK(nbytes > 0 && ptr_out != nullptr && handle_out != nullptr, "car_meta_alloc: bad arguments");
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, (size_t)nbytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipMalloc(&p, (size_t)nbytes);
  }
  if (e != hipSuccess) {
    ::nmv::set_error("car_meta_alloc: allocation failed: %s", hipGetErrorString(e));
    return NMV_ERR_HIP;
  }
  AR_HIP(hipMemset(p, 0, (size_t)nbytes));
  AR_HIP(hipDeviceSynchronize());
  hipIpcMemHandle_t h;
  AR_HIP(hipIpcGetMemHandle(&h, p));
  std::memcpy(handle_out, &h, sizeof(h));
  *ptr_out = p;
  return NMV_OK;
}
extern "C" int nmv_car_meta_free(void* ptr) {
  if (ptr == nullptr) return NMV_OK;
  (void)hipDeviceSynchronize();
  AR_HIP(hipFree(ptr));
  return NMV_OK;
}

/* meta: this rank's tensor of nmv_car_meta_size() + max_size bytes (zeroed); rank_data: device scratch for the
 * pointer tables; handles: world x nmv_ar_handle_bytes() (IPC handles of every rank's meta allocation), offsets:
 * byte offset of meta inside that allocation.  custom_all_reduce.cu:12-34 */
extern "C" int nmv_car_init(void** state_out, void* meta, void* rank_data, int64_t rank_data_bytes,
                            const void* handles, const int64_t* offsets, int world, int rank, int full_link) {
  NMV_CHECK(world <= 8, "world size > 8 is not supported");
  NMV_CHECK(world % 2 == 0, "Odd num gpus is not supported for now");
  NMV_CHECK(rank >= 0 && rank < world, "invalid rank passed in");
  NMV_CHECK(meta != nullptr && rank_data != nullptr && rank_data_bytes >= (int64_t)sizeof(RegPtrs),
            "init_custom_ar: meta / rank_data missing");
  RegState* st = new RegState();
  st->rank = rank; st->world = world; st->full_link = full_link;
  st->spin_ticks = ar_default_spin_ticks();
  st->force_algo = 0;
  st->rd_next = (RegPtrs*)rank_data;
  st->rd_end = st->rd_next + rank_data_bytes / (int64_t)sizeof(RegPtrs);
  std::memset(&st->peers, 0, sizeof(st->peers));
  for (int q = 0; q < world; ++q) {
    char* base = (char*)meta;
    if (q != rank) {
      if (!reg_open(st, (const uint8_t*)handles + (size_t)q * sizeof(hipIpcMemHandle_t), &base)) {
        delete st;
        ::nmv::append_error(" (init_custom_ar: rank %d's meta buffer)", q);
        return NMV_ERR_HIP;
      }
      base += offsets[q];
    }
    st->peers.comm[q] = (ArComm*)base;
    st->peers.tmp[q] = (uint8_t*)base + sizeof(ArComm);
  }
  *state_out = st;
  return NMV_OK;
}

/* custom_all_reduce.cuh:337-355 */
extern "C" int nmv_car_register_buffer(void* state, const void* self, const void* handles, const int64_t* offsets) {
  RegState* st = (RegState*)state;
  NMV_CHECK(st != nullptr, "register_buffer: bad handle");
  NMV_CHECK(st->rd_next + 1 <= st->rd_end, "Rank data buffer is overflowed by 1");
  RegPtrs d;
  std::memset(&d, 0, sizeof(d));
  for (int q = 0; q < st->world; ++q) {
    if (q == st->rank) { d.p[q] = self; continue; }
    char* base;
    NMV_CHECK(reg_open(st, (const uint8_t*)handles + (size_t)q * sizeof(hipIpcMemHandle_t), &base),
              "register_buffer: cannot map rank %d's buffer", q);
    d.p[q] = base + offsets[q];
  }
  AR_HIP(hipMemcpy(st->rd_next, &d, sizeof(d), hipMemcpyHostToDevice));
  st->buffers[self] = st->rd_next++;
  return NMV_OK;
}

/* all_reduce_reg: inp is a registered buffer, or -- while the stream is capturing -- a graph-private tensor
 * that is recorded and registered after the capture (custom_all_reduce.cuh:402-432).  elem_size 2 (fp16 / bf16
 * by dtype) or 4 (fp32). */
extern "C" int nmv_car_all_reduce(void* state, const void* inp, void* out, int64_t numel, nmv_dtype_t dtype,
                                  void* stream) {
  RegState* st = (RegState*)state;
  NMV_CHECK(st != nullptr, "all_reduce: bad handle");
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16 || dtype == NMV_F32,
            "custom allreduce only supports float32, float16 and bfloat16");
  const int64_t bytes = numel * (dtype == NMV_F32 ? 4 : 2);
  NMV_CHECK(bytes > 0 && bytes % 16 == 0, "custom allreduce currently requires input length to be multiple of %d",
            dtype == NMV_F32 ? 4 : 8);
  NMV_CHECK((((uintptr_t)inp | (uintptr_t)out) & 15) == 0, "custom allreduce: 16-byte aligned tensors");
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  AR_HIP(hipStreamIsCapturing(s, &cap));
  const RegPtrs* ptrs;
  if (cap == hipStreamCaptureStatusActive) {
    NMV_CHECK(st->rd_next + st->graph_unreg.size() + 1 <= st->rd_end, "Rank data buffer is overflowed by 1");
    ptrs = st->rd_next + st->graph_unreg.size();      // filled in by register_graph_buffers
    st->graph_unreg.push_back(const_cast<void*>(inp));
  } else {
    auto it = st->buffers.find(inp);
    NMV_CHECK(it != st->buffers.end(), "buffer address %llu is not registered!", (unsigned long long)(uintptr_t)inp);
    ptrs = it->second;
  }
  const int64_t n_vec = bytes / 16;
  bool two;
  if (st->force_algo) two = st->force_algo == 2;
  else if (st->world == 2) two = false;
  else if (st->full_link) two = !((st->world <= 4 && bytes < 512 * 1024) || (st->world <= 8 && bytes < 256 * 1024));
  else two = false;   // the reference launches nothing here (:442-451); should_custom_ar never lets such a call in
#define NMV_REG_AR(T_)                                                                                          \
  do {                                                                                                          \
    if (two)                                                                                                    \
      hipLaunchKernelGGL((reg_all_reduce_kernel<T_, true>), dim3(AR_MAX_BLOCKS), dim3(AR_THREADS), 0, s,        \
                         st->peers, ptrs, st->rank, st->world, out, n_vec, st->spin_ticks);                     \
    else                                                                                                        \
      hipLaunchKernelGGL((reg_all_reduce_kernel<T_, false>), dim3(AR_MAX_BLOCKS), dim3(AR_THREADS), 0, s,       \
                         st->peers, ptrs, st->rank, st->world, out, n_vec, st->spin_ticks);                     \
  } while (0)
  if (dtype == NMV_F16) NMV_REG_AR(F16);
  else if (dtype == NMV_BF16) NMV_REG_AR(BF16);
  else NMV_REG_AR(float);
#undef NMV_REG_AR
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_car_graph_buffer_count(void* state) {
  RegState* st = (RegState*)state;
  return st ? (int)st->graph_unreg.size() : 0;
}

/* handles_out: count x nmv_ar_handle_bytes(), offsets_out: count entries (custom_all_reduce.cuh:308-328: the
 * handle of the allocation's base address + the tensor's offset in it) */
extern "C" int nmv_car_get_graph_buffer_ipc_meta(void* state, void* handles_out, int64_t* offsets_out) {
  RegState* st = (RegState*)state;
  NMV_CHECK(st != nullptr, "get_graph_buffer_ipc_meta: bad handle");
  for (size_t i = 0; i < st->graph_unreg.size(); ++i) {
    void* ptr = st->graph_unreg[i];
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    AR_HIP(hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr));
    hipIpcMemHandle_t h;
    AR_HIP(hipIpcGetMemHandle(&h, (void*)base));
    std::memcpy((uint8_t*)handles_out + i * sizeof(h), &h, sizeof(h));
    offsets_out[i] = (int64_t)((char*)ptr - (char*)base);
  }
  return NMV_OK;
}

/* handles: [world][count] handles, offsets: [world][count] (custom_all_reduce.cuh:364-391) */
extern "C" int nmv_car_register_graph_buffers(void* state, const void* handles, const int64_t* offsets) {
  RegState* st = (RegState*)state;
  NMV_CHECK(st != nullptr, "register_graph_buffers: bad handle");
  const size_t n = st->graph_unreg.size();
  NMV_CHECK(st->rd_next + n <= st->rd_end, "Rank data buffer is overflowed by %d", (int)n);
  if (n == 0) return NMV_OK;
  std::vector<RegPtrs> rd(n);
  for (size_t i = 0; i < n; ++i) {
    std::memset(&rd[i], 0, sizeof(RegPtrs));
    for (int q = 0; q < st->world; ++q) {
      if (q == st->rank) { rd[i].p[q] = st->graph_unreg[i]; continue; }
      char* base;
      NMV_CHECK(reg_open(st, (const uint8_t*)handles + ((size_t)q * n + i) * sizeof(hipIpcMemHandle_t), &base),
                "register_graph_buffers: cannot map rank %d's buffer %d", q, (int)i);
      rd[i].p[q] = base + offsets[(size_t)q * n + i];
    }
  }
  AR_HIP(hipMemcpy(st->rd_next, rd.data(), sizeof(RegPtrs) * n, hipMemcpyHostToDevice));
  st->rd_next += n;
  st->graph_unreg.clear();
  return NMV_OK;
}

extern "C" int nmv_car_set_algo(void* state, int algo) {
  RegState* st = (RegState*)state;
  NMV_CHECK(st != nullptr && algo >= 0 && algo <= 2, "custom_all_reduce: algo must be 0, 1 or 2");
  st->force_algo = algo;
  return NMV_OK;
}

/* 1 when a flag wait ran out on this rank (synchronises the device) */
extern "C" int nmv_car_error(void* state) {
  RegState* st = (RegState*)state;
  if (st == nullptr) return 1;
  uint32_t err = 1;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpy(&err, &st->peers.comm[st->rank]->error, 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  return (int)err;
}

extern "C" int nmv_car_dispose(void* state) {
  RegState* st = (RegState*)state;
  if (st == nullptr) return NMV_OK;
  (void)hipDeviceSynchronize();
  for (auto& kv : st->opened) (void)hipIpcCloseMemHandle(kv.second);
  delete st;
  return NMV_OK;
}
