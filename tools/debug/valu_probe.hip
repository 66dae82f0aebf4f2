// Stand-alone probe: cycles per vector instruction per SIMD on MI355X for the opcodes the W4A16 expansion uses,
// at 1 / 2 / 4 / 8 waves per SIMD (one workgroup per CU is launched so that every SIMD holds exactly that many).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/debug/valu_probe.hip -o build/probe/valu_probe
// Output: cycles per instruction per SIMD = s_memtime cycles of the loop / (instructions issued by all waves of a SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int REP = 64;      // instructions per loop body (independent: 8 chains of 8)
constexpr int ITERS = 2000;

// OP: 0 alignbit, 1 and_or, 2 lshrrev, 3 mov_dpp row_ror:8, 4 fma_f32, 5 perm_b32, 6 bfe_u32, 7 add_u32, 8 lshl_or,
//     9 cvt_pk_bf16_f32, 10 mfma 16x16x32 bf16 (chain of 4 independent), 11 mix: 2 alignbit + 2 and_or + ... as in the kernel
template <int OP>
__global__ void probe(unsigned long long* out, unsigned seed) {
  unsigned v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1) + threadIdx.x;
  unsigned m = seed | 0x00780078u, s = (threadIdx.x & 7) + 1;
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(i + threadIdx.x); b8[i] = (__bf16)1.0f; }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int rp = 0; rp < REP / 8; ++rp) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (OP == 0) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(v[i]) : "v"(s));
        if constexpr (OP == 1) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[i]) : "s"(0x00780078u), "v"(m));
        if constexpr (OP == 2) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(v[i]) : "v"(s));
        if constexpr (OP == 3) asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(v[i]));
        if constexpr (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
        if constexpr (OP == 5) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(s));
        if constexpr (OP == 6) asm volatile("v_bfe_u32 %0, %0, %1, 4" : "+v"(v[i]) : "v"(s));
        if constexpr (OP == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(s));
        if constexpr (OP == 8) asm volatile("v_lshl_or_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(s), "v"(m));
        if constexpr (OP == 9) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "+v"(v[i]) : "v"(f[i]), "v"(f[(i + 1) & 7]));
        if constexpr (OP == 10) {
          if (i < 4) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
        }
        if constexpr (OP == 11) {   // the expansion of one tile: 2 dpp + 4 alignbit + 4 and_or, then one MFMA
          if (i == 0) {
            unsigned e, o, w0, w1, w2, w3;
            asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xc" : "=v"(e) : "v"(v[rp]));
            asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0x3" : "=v"(o) : "v"(v[rp]));
            asm volatile("v_alignbit_b32 %0, %1, %1, %2" : "=v"(w0) : "v"(e), "v"(s));
            asm volatile("v_alignbit_b32 %0, %1, %1, %2" : "=v"(w1) : "v"(e), "v"(m));
            asm volatile("v_alignbit_b32 %0, %1, %1, %2" : "=v"(w2) : "v"(o), "v"(s));
            asm volatile("v_alignbit_b32 %0, %1, %1, %2" : "=v"(w3) : "v"(o), "v"(m));
            asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(w0) : "s"(0x00780078u), "v"(m));
            asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(w1) : "s"(0x00780078u), "v"(m));
            asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(w2) : "s"(0x00780078u), "v"(m));
            asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(w3) : "s"(0x00780078u), "v"(m));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 wv = {w0, w1, w2, w3};
            acc[rp & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv), b8, acc[rp & 3], 0, 0, 0);
          }
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += v[i] + (unsigned)f[i];
  r += (unsigned)(acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3]);
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2] = t1 - t0;
  if (r == 0x12345) out[1] = r;
}

template <int OP>
void run(const char* name, int per_iter, unsigned long long* dout) {
  for (int wps : {1, 2, 4, 8}) {
    const int threads = wps * 4 * 64;
    if (threads > 1024) {   // 8 waves per SIMD = two 1024-thread workgroups per CU
      hipLaunchKernelGGL((probe<OP>), dim3(512), dim3(1024), 0, 0, dout, 12345u);
    } else {
      hipLaunchKernelGGL((probe<OP>), dim3(256), dim3(threads), 0, 0, dout, 12345u);
    }
    CK(hipDeviceSynchronize());
    unsigned long long h[64];
    CK(hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost));
    double cyc = 0;
    const int nw = threads > 1024 ? 16 : threads / 64;
    for (int i = 0; i < nw; ++i) cyc += (double)h[2 * i];
    cyc /= nw;
    // s_memtime counts at 100 MHz reference? report both raw ticks and per-instruction
    const double insts_per_simd = (double)ITERS * per_iter * wps;
    printf("%-12s waves/SIMD %d: %10.0f ticks  -> %6.3f ticks per instruction per SIMD\n", name, wps, cyc, cyc / insts_per_simd);
  }
}

int main() {
  unsigned long long* dout;
  CK(hipMalloc(&dout, 1 << 20));
  run<4>("fma_f32", REP, dout);
  run<0>("alignbit", REP, dout);
  run<1>("and_or", REP, dout);
  run<2>("lshrrev", REP, dout);
  run<3>("mov_dpp", REP, dout);
  run<5>("perm_b32", REP, dout);
  run<6>("bfe_u32", REP, dout);
  run<7>("add_u32", REP, dout);
  run<8>("lshl_or", REP, dout);
  run<9>("cvt_pk_bf16", REP, dout);
  run<10>("mfma16x16x32", REP / 2, dout);
  run<11>("tile(10+mfma)", REP / 8 * 11, dout);
  return 0;
}
