// One-shot P2P all-reduce over HIP IPC for the tensor-parallel decode step: the gfx950 analogue of
// the reference's csrc/custom_all_reduce.cuh:130-250 (cross_device_reduce_1stage), which the
// reference compiles out on ROCm (torch_bindings.cpp:261).  Decode all-reduces a [B, hidden]
// activation (8 KB .. 1 MB) twice per layer; at that size a ring collective is pure latency, while
// xGMI lets every GPU read every peer's buffer directly (7 links x ~150 GB/s per GPU).
//
// Protocol (all ranks launch the same grid; block b of every rank owns slice b of the message):
//   1. block b copies slice b of its input into its rank's IPC-mapped staging buffer (two buffers,
//      alternating per call, so that no trailing barrier is needed: a rank can only reach call n+2
//      after every peer has passed the flag wait of call n+1, i.e. has finished reading call n);
//   2. system-scope release, then lanes 0..W-1 store the block's call number into slot
//      [b][my_rank] of every rank's flag array and spin (bounded) until slots [b][0..W-1] of the
//      own array carry it;
//   3. system-scope acquire; every lane sums slice b of all W staging buffers in rank order (the
//      same order on every rank => bit-identical results everywhere, deterministic) in fp32 and
//      writes the output.
// Call numbers live in device memory and are advanced by the kernel itself, so a captured launch
// can be replayed.  The spin is bounded in time (2 s): a lost peer sets an error word instead of
// hanging the GPU.
// Staging and flags are allocated uncached / fine-grained when the runtime allows it.
#include "common.h"

#include <cstring>
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
  uint32_t flags[AR_MAX_BLOCKS][AR_MAX_RANKS];  // [block][source rank] = call number
  uint32_t seq[AR_MAX_BLOCKS];                  // this rank's call number per block
  uint32_t error;                               // set when a spin ran out
  uint32_t pad[63];
};

struct ArPeers {
  ArComm* comm[AR_MAX_RANKS];
  uint8_t* data[AR_MAX_RANKS];  // staging: 2 x max_bytes each
};

struct ArState {
  int rank, world;
  int64_t max_bytes;
  void* base;        // one allocation: ArComm + 2 staging buffers
  size_t alloc_bytes;
  hipIpcMemHandle_t handle;
  void* peer_base[AR_MAX_RANKS];
  ArPeers peers;
  bool opened;
};

// GATHER: same protocol, but slice b of every rank's buffer is copied to out[q * n_vec + slice]
// instead of summed (all-gather of small per-rank records, e.g. the per-shard argmax of the sampler)
template <typename T, bool GATHER = false>
__global__ __launch_bounds__(AR_THREADS) void one_shot_all_reduce_kernel(ArPeers peers, int rank, int world,
                                                                         const uint16_t* __restrict__ inp,
                                                                         uint16_t* __restrict__ out,
                                                                         int64_t n_vec /* 16-byte vectors */,
                                                                         int64_t buf_bytes,
                                                                         const float* __restrict__ slab = nullptr,
                                                                         int splits = 0, int64_t slab_stride = 0) {
  const int b = blockIdx.x;
  ArComm* mine = peers.comm[rank];
  const uint32_t seq = mine->seq[b] + 1;  // only this block touches seq[b]
  const int64_t parity_off = (seq & 1) ? buf_bytes : 0;
  const int64_t per = (n_vec + gridDim.x - 1) / gridDim.x;
  const int64_t v0 = (int64_t)b * per, v1 = min(v0 + per, n_vec);
  // 1. my slice -> my staging buffer
  uint4* stage = reinterpret_cast<uint4*>(peers.data[rank] + parity_off);
  const uint4* src = reinterpret_cast<const uint4*>(inp);
  if (slab != nullptr) {
    // the input is still the fp32 split-K slabs of the row-parallel GEMM (nmv_gptq_marlin_gemm_partial):
    // summed in split order from +0 and rounded to the model dtype, as that GEMM's own last pass would
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
    for (int64_t v = v0 + threadIdx.x; v < v1; v += AR_THREADS) stage[v] = src[v];
  }
  __threadfence_system();
  __syncthreads();
  // 2. signal every rank (myself included), wait for every rank
  if (threadIdx.x < world) {
    const int q = threadIdx.x;
    __hip_atomic_store(&peers.comm[q]->flags[b][rank], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while ((int32_t)(__hip_atomic_load(&mine->flags[b][q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > AR_SPIN_TICKS) {
        __hip_atomic_store(&mine->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  // system-scope acquire on every lane: drop whatever the caches hold of the peers' buffers
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  // 3. sum slice b of every rank's buffer, rank order, fp32
  uint4* dst = reinterpret_cast<uint4*>(out);
  if constexpr (GATHER) {
    for (int q = 0; q < world; ++q)
      for (int64_t v = v0 + threadIdx.x; v < v1; v += AR_THREADS)
        dst[q * n_vec + v] = reinterpret_cast<const uint4*>(peers.data[q] + parity_off)[v];
    if (threadIdx.x == 0) mine->seq[b] = seq;
    return;
  }
  for (int64_t v = v0 + threadIdx.x; v < v1; v += AR_THREADS) {
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
    dst[v] = make_uint4(T::pack2(acc[0], acc[1]), T::pack2(acc[2], acc[3]), T::pack2(acc[4], acc[5]),
                        T::pack2(acc[6], acc[7]));
  }
  if (threadIdx.x == 0) mine->seq[b] = seq;
}

// ---------------------------------------------------------------------------------------------
// all-reduce + residual-add + RMSNorm in one launch (tensor-parallel decode: o_proj / down_proj
// -> all-reduce -> fused_add_rms_norm is three launches per projection otherwise).  Same protocol;
// the slices are whole rows (block b owns rows [b R, (b+1) R)), so after the handshake a block holds
// complete reduced rows and can finish them: sum over ranks (fp32, rank order) rounded to the model
// dtype = what nmv_ar_all_reduce writes; + residual, rounded, stored back; variance with the per-lane
// accumulation order of rms_norm_reg_kernel's 256-lane form (replayed through LDS, as its wide form
// does); x * rsqrt rounded, times weight.  Bit-identical to the three launches.
template <typename T>
__device__ __forceinline__ float ar_rnd(float f) {
  uint32_t b = T::from_float(f);
  asm volatile("" : "+v"(b));  // keep the rounding (see glue_kernels.hip: pin())
  return T::to_float((uint16_t)b);
}

template <typename T>
__global__ __launch_bounds__(AR_THREADS) void one_shot_all_reduce_norm_kernel(
    ArPeers peers, int rank, int world, const uint16_t* __restrict__ inp, const float* __restrict__ slab,
    int splits, int64_t slab_stride, uint16_t* __restrict__ residual, const uint16_t* __restrict__ weight,
    uint16_t* __restrict__ out, float epsilon, int rows, int hidden, int64_t buf_bytes) {
  __shared__ float fold[2 * AR_THREADS][4];
  __shared__ float red[AR_THREADS / 64];
  const int b = blockIdx.x;
  ArComm* mine = peers.comm[rank];
  const uint32_t seq = mine->seq[b] + 1;
  const int64_t parity_off = (seq & 1) ? buf_bytes : 0;
  const int hv = hidden / 8;                         // 16-byte vectors per row (<= 2 * AR_THREADS)
  const int per = (rows + gridDim.x - 1) / gridDim.x;
  const int r0 = b * per, r1 = min(r0 + per, rows);
  uint4* stage = reinterpret_cast<uint4*>(peers.data[rank] + parity_off);
  // 1. my rows -> my staging buffer
  for (int r = r0; r < r1; ++r)
    for (int v = threadIdx.x; v < hv; v += AR_THREADS) {
      const int64_t e = (int64_t)r * hv + v;
      if (slab != nullptr) {
        f32x4_t lo4 = {0.f, 0.f, 0.f, 0.f}, hi4 = {0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < splits; ++sp) {
          lo4 += *reinterpret_cast<const f32x4_t*>(slab + sp * slab_stride + e * 8);
          hi4 += *reinterpret_cast<const f32x4_t*>(slab + sp * slab_stride + e * 8 + 4);
        }
        stage[e] = make_uint4(T::pack2(lo4[0], lo4[1]), T::pack2(lo4[2], lo4[3]), T::pack2(hi4[0], hi4[1]),
                              T::pack2(hi4[2], hi4[3]));
      } else {
        stage[e] = reinterpret_cast<const uint4*>(inp)[e];
      }
    }
  __threadfence_system();
  __syncthreads();
  // 2. handshake
  if (threadIdx.x < world) {
    const int q = threadIdx.x;
    __hip_atomic_store(&peers.comm[q]->flags[b][rank], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while ((int32_t)(__hip_atomic_load(&mine->flags[b][q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > AR_SPIN_TICKS) {
        __hip_atomic_store(&mine->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  // 3. finish my rows
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = r0; r < r1; ++r) {
    uint32_t z[2][4];
    float terms[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = threadIdx.x + i * AR_THREADS;
      const bool ok = v < hv;
      const int64_t e = (int64_t)r * hv + (ok ? v : 0);
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int q = 0; q < world; ++q) {
        const uint4 x = reinterpret_cast<const uint4*>(peers.data[q] + parity_off)[e];
        const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[2 * j] += lo_f<T>(xs[j]);
          acc[2 * j + 1] += hi_f<T>(xs[j]);
        }
      }
      const uint4 rs4 = ld16(residual + e * 8);
      const uint32_t rs[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // all-reduce output rounded to the model dtype, then z = x + residual rounded (layernorm_kernels.cu:271-274)
        const float lo = ar_rnd<T>(ar_rnd<T>(acc[2 * j]) + lo_f<T>(rs[j]));
        const float hi = ar_rnd<T>(ar_rnd<T>(acc[2 * j + 1]) + hi_f<T>(rs[j]));
        z[i][j] = T::pack2(lo, hi);
        terms[i][j] = ok ? lo * lo + hi * hi : 0.f;
      }
      if (ok) st16(residual + e * 8, make_uint4(z[i][0], z[i][1], z[i][2], z[i][3]));
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

}  // namespace nmv

using namespace nmv;

#define AR_HIP(call)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      ::nmv::set_error("custom_all_reduce: %s failed: %s", #call, hipGetErrorString(e_));   \
      return NMV_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

extern "C" int nmv_ar_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

/* allocate this rank's comm block + staging (2 x max_bytes) on the current device and export it */
extern "C" int nmv_ar_create(void** state_out, int rank, int world, int64_t max_bytes, void* handle_out) {
  NMV_CHECK(world >= 2 && world <= AR_MAX_RANKS && rank >= 0 && rank < world, "custom_all_reduce: bad rank / world");
  NMV_CHECK(max_bytes > 0 && max_bytes % 16 == 0, "custom_all_reduce: max_bytes must be a multiple of 16");
  ArState* st = new ArState();
  std::memset(st, 0, sizeof(ArState));
  st->rank = rank; st->world = world; st->max_bytes = max_bytes;
  st->alloc_bytes = sizeof(ArComm) + 2 * (size_t)max_bytes;
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
  }
  st->opened = true;
  return NMV_OK;
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
  const int64_t n_vec = bytes / 16;
  const int blocks = AR_MAX_BLOCKS;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == NMV_F16)
    hipLaunchKernelGGL((one_shot_all_reduce_kernel<F16>), dim3(blocks), dim3(AR_THREADS), 0, s, st->peers,
                       st->rank, st->world, (const uint16_t*)inp, (uint16_t*)out, n_vec, st->max_bytes);
  else
    hipLaunchKernelGGL((one_shot_all_reduce_kernel<BF16>), dim3(blocks), dim3(AR_THREADS), 0, s, st->peers,
                       st->rank, st->world, (const uint16_t*)inp, (uint16_t*)out, n_vec, st->max_bytes);
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
  const int64_t n_vec = bytes / 16;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == NMV_F16)
    hipLaunchKernelGGL((one_shot_all_reduce_kernel<F16>), dim3(AR_MAX_BLOCKS), dim3(AR_THREADS), 0, s, st->peers,
                       st->rank, st->world, (const uint16_t*)nullptr, (uint16_t*)out, n_vec, st->max_bytes, slab,
                       splits, numel);
  else
    hipLaunchKernelGGL((one_shot_all_reduce_kernel<BF16>), dim3(AR_MAX_BLOCKS), dim3(AR_THREADS), 0, s, st->peers,
                       st->rank, st->world, (const uint16_t*)nullptr, (uint16_t*)out, n_vec, st->max_bytes, slab,
                       splits, numel);
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
  if (dtype == NMV_F16)
    hipLaunchKernelGGL((one_shot_all_reduce_norm_kernel<F16>), dim3(AR_MAX_BLOCKS), dim3(AR_THREADS), 0, s,
                       st->peers, st->rank, st->world, (const uint16_t*)inp, slab, splits, stride,
                       (uint16_t*)residual, (const uint16_t*)weight, (uint16_t*)out, epsilon, rows, hidden,
                       st->max_bytes);
  else
    hipLaunchKernelGGL((one_shot_all_reduce_norm_kernel<BF16>), dim3(AR_MAX_BLOCKS), dim3(AR_THREADS), 0, s,
                       st->peers, st->rank, st->world, (const uint16_t*)inp, slab, splits, stride,
                       (uint16_t*)residual, (const uint16_t*)weight, (uint16_t*)out, epsilon, rows, hidden,
                       st->max_bytes);
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
  const int blocks = AR_MAX_BLOCKS;
  hipLaunchKernelGGL((one_shot_all_reduce_kernel<BF16, true>), dim3(blocks), dim3(AR_THREADS), 0,
                     (hipStream_t)stream, st->peers, st->rank, st->world, (const uint16_t*)inp,
                     (uint16_t*)out, n_vec, st->max_bytes);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

/* 1 when a bounded spin ran out on this rank since creation (synchronises the device) */
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
