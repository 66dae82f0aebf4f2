// Stand-alone probe (no torch): what a weight stream of the W4A16 decode GEMM can reach on MI355X as a
// function of access pattern, loads in flight per wave, waves per CU and interleaved vector work.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/debug/stream_probe.hip -o gpurun_out/stream_probe
// Run:   gpurun_out/stream_probe            (prints one line per configuration: us, TB/s)
//
// The tensor is the gate_up Marlin tensor of Llama-3-8B: K/16 = 256 rows of N*8 = 229,376 bytes (58.7 MB); a
// "chunk" is 512 bytes of a row (64 columns x 16 k).  Patterns:
//   0 marlin : wave = (chunk, k part); per k-step lanes 0..31 (by blk) read 512 B of row 2ks, lanes 32.. of row 2ks+1
//   1 band   : same bytes per wave and k-step, but the two 512-B pieces are adjacent (1 KiB contiguous), rows of
//              N*16 bytes per k-step (the "native" layout's band order)
//   2 linear : wave w reads KiB number (step * n_waves + w): the chip-wide stream is one sweep front to back
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int K = 4096, N = 28672;
constexpr int ROWS = K / 16;                 // k-tile rows
constexpr size_t ROW_BYTES = (size_t)N * 8;  // 229376
constexpr size_t TENSOR = ROWS * ROW_BYTES;  // 58.7 MB

// PAT pattern, DEPTH 16-byte loads in flight per lane, VW dummy vector ops per load, NT non-temporal
template <int PAT, int DEPTH, int VW, int NT>
__global__ void probe(const uint4* __restrict__ w, unsigned* __restrict__ sink, int ksteps_per_wave, int kparts) {
  const int lane = threadIdx.x & 63;
  const int wave_in_wg = threadIdx.x >> 6;
  const int waves_per_wg = blockDim.x >> 6;
  const int gw = blockIdx.x * waves_per_wg + wave_in_wg;   // global wave
  const int n_waves = gridDim.x * waves_per_wg;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(w), 0, (int)TENSOR, 0x00020000);
  unsigned voff, step_bytes;
  unsigned soff0;
  if (PAT == 0) {
    const int chunk = gw / kparts, kp = gw % kparts;
    const int blk = (lane >> 3) & 1, sub = (lane & 7) * 4 + (lane >> 4);   // as the kernels map lanes to vectors
    voff = (unsigned)(chunk * 512 + sub * 16 + blk * ROW_BYTES);
    soff0 = (unsigned)((size_t)kp * ksteps_per_wave * 2 * ROW_BYTES);
    step_bytes = (unsigned)(2 * ROW_BYTES);
  } else if (PAT == 1) {
    const int chunk = gw / kparts, kp = gw % kparts;
    voff = (unsigned)(chunk * 1024 + lane * 16);
    soff0 = (unsigned)((size_t)kp * ksteps_per_wave * 2 * ROW_BYTES);
    step_bytes = (unsigned)(2 * ROW_BYTES);
  } else {
    voff = (unsigned)(lane * 16);
    soff0 = (unsigned)((size_t)gw * 1024);
    step_bytes = (unsigned)((size_t)n_waves * 1024);
  }
  u32x4 q[DEPTH];
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < DEPTH; ++i)
    q[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff0 + i * step_bytes, NT ? 2 : 0);
  for (int s = 0; s < ksteps_per_wave; s += DEPTH) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      u32x4 v = q[i];
      const int nxt = s + i + DEPTH;
      // past the end: an offset that fails the bounds check (no request)
      q[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, nxt < ksteps_per_wave ? voff : 0x7ffffff0u,
                                                   soff0 + (unsigned)nxt * step_bytes, NT ? 2 : 0);
      unsigned x = v.x ^ v.y ^ v.z ^ v.w;
#pragma unroll
      for (int j = 0; j < VW; ++j) x = __builtin_amdgcn_alignbit(x, x, 7) + (acc | 0x41804180u);
      acc += x;
    }
  }
  if (acc == 0x12345678u) sink[gw] = acc;   // keeps the loads alive, (almost) never stores
}

struct Cfg { int pat, depth, vw, nt, waves_per_wg, kparts; };

template <int PAT, int DEPTH, int VW, int NT>
float run(const uint4* w, int ncopy, unsigned* sink, int waves_per_wg, int kparts, hipStream_t st) {
  // work decomposition: N/64 chunks x kparts waves, each 128/kparts k-steps (pattern 2: the same wave count)
  const int n_waves = (N / 64) * kparts;
  const int ksteps = (K / 32) / kparts;
  if (ksteps % DEPTH != 0) return -1.f;
  const int grid = n_waves / waves_per_wg;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t stride = TENSOR / 16;
  for (int i = 0; i < 2; ++i)
    hipLaunchKernelGGL((probe<PAT, DEPTH, VW, NT>), dim3(grid), dim3(waves_per_wg * 64), 0, st, w + (i % ncopy) * stride, sink, ksteps, kparts);
  CK(hipStreamSynchronize(st));
  const int iters = 20;
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((probe<PAT, DEPTH, VW, NT>), dim3(grid), dim3(waves_per_wg * 64), 0, st, w + ((i + 2) % ncopy) * stride, sink, ksteps, kparts);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

int main() {
  const int ncopy = 11;   // 646 MB: past the 256 MiB Infinity Cache
  uint4* w;
  unsigned* sink;
  CK(hipMalloc(&w, TENSOR * ncopy));
  CK(hipMalloc(&sink, 1 << 20));
  CK(hipMemset(w, 0x5a, TENSOR * ncopy));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  printf("pattern depth vw nt waves/wg kparts |   us    TB/s\n");
#define RUN(P_, D_, V_, NT_, WPW, KP)                                                          \
  {                                                                                            \
    float us = run<P_, D_, V_, NT_>(w, ncopy, sink, WPW, KP, st);                              \
    if (us > 0) printf("%7d %5d %2d %2d %8d %6d | %6.2f  %5.2f\n", P_, D_, V_, NT_, WPW, KP, us, TENSOR / us * 1e-6); \
    fflush(stdout);                                                                            \
  }
  // 1. depth x occupancy, no work
  for (int wpw : {4, 8, 16}) {
    for (int kp : {4, 8, 16}) {
      RUN(0, 4, 0, 0, wpw, kp) RUN(0, 8, 0, 0, wpw, kp) RUN(0, 16, 0, 0, wpw, kp)
      RUN(1, 4, 0, 0, wpw, kp) RUN(1, 8, 0, 0, wpw, kp) RUN(1, 16, 0, 0, wpw, kp)
      RUN(2, 4, 0, 0, wpw, kp) RUN(2, 8, 0, 0, wpw, kp) RUN(2, 16, 0, 0, wpw, kp)
    }
  }
  // 2. non-temporal
  RUN(0, 8, 0, 1, 8, 8) RUN(0, 16, 0, 1, 8, 8) RUN(2, 8, 0, 1, 8, 8) RUN(2, 16, 0, 1, 8, 8)
  RUN(0, 8, 0, 1, 4, 4) RUN(0, 16, 0, 1, 4, 4)
  // 3. with vector work per load (the real kernel: ~50 ops per 16-byte load at M <= 16)
  for (int kp : {4, 8, 16}) {
    RUN(0, 8, 16, 0, 8, kp) RUN(0, 8, 32, 0, 8, kp) RUN(0, 8, 48, 0, 8, kp) RUN(0, 8, 64, 0, 8, kp)
    RUN(0, 16, 32, 0, 8, kp) RUN(0, 16, 48, 0, 8, kp) RUN(0, 16, 64, 0, 8, kp)
    RUN(2, 8, 48, 0, 8, kp) RUN(2, 16, 48, 0, 8, kp)
  }
  RUN(0, 8, 48, 0, 16, 16) RUN(0, 8, 48, 0, 16, 8) RUN(0, 8, 48, 0, 4, 4) RUN(0, 16, 48, 0, 4, 4)
  return 0;
}
