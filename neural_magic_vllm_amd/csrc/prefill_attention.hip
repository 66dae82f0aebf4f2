// Prompt (prefill) attention for gfx950: varlen, causal, GQA flash-attention forward on MFMA.
//
// Behavioural reference: the prompt branch of ROCmFlashAttentionImpl.forward
// (/root/reference/vllm/attention/backends/rocm_flash_attn.py:349-430): causal attention of every
// prompt over its own freshly computed K/V ([tokens, heads, head_size] views of the qkv GEMM
// output), softmax scale `scale`, queries of a GQA group sharing one KV head.  The reference
// dispatches this to Triton / CK flash-attention or torch SDPA; this is the hand-written HIP
// replacement.  fp32 softmax and accumulation, P rounded to the model
// dtype before P.V (flash-attention convention).
//
// Decomposition: one 256-thread workgroup per (64-query tile, head, prompt); wave w owns 16 query
// rows.  Everything is computed TRANSPOSED so that no register shuffle is ever needed:
//   S^T[key][q] = K . Q^T   -- MFMA 16x16x32 with K rows as the A operand (a lane loads 16 bytes =
//                              8 head dims of one key straight from global memory) and the wave's
//                              Q fragment (loaded once) as B.  A lane (q = l&15, g = l>>4) ends
//                              up with keys 16 t + 4 g + i of ITS query row: row max / sum are a
//                              per-lane loop plus two xor-shuffles (lanes q, q+16, q+32, q+48).
//   O^T[d][q]  += V^T . P^T -- the probabilities of a lane, packed in pairs, ARE the B operand of
//                              this MFMA (the k-slot order {4g..4g+3, 16+4g..16+4g+3} per 32 keys is
//                              just a permutation of the contraction index); V^T comes from an LDS
//                              copy of the 64-key V tile read with that same key order.
// The rescale factor alpha, the running max and the row sum live per lane (its query row).
#include "common.h"

namespace nmv {

constexpr int FA_QT = 64;   // queries per workgroup
constexpr int FA_KT = 64;   // keys per iteration

template <typename T> struct FaMfma;
template <> struct FaMfma<BF16> {
  static __device__ __forceinline__ f32x4_t run(uint4 a, uint4 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct FaMfma<F16> {
  static __device__ __forceinline__ f32x4_t run(uint4 a, uint4 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef v4s_t __attribute__((address_space(3))) lds_v4s_t;

struct FaParams {
  const uint16_t* q;   // [tokens, H, D], token stride q_stride (elements)
  const uint16_t* k;   // [tokens, KVH, D], token stride kv_stride
  const uint16_t* v;
  uint16_t* out;       // [tokens, H, D], token stride o_stride
  const int* cu_seqlens;  // [num_seqs + 1] token offsets of the prompts
  int64_t q_stride, kv_stride, o_stride;
  int num_heads, num_kv_heads;
  float scale;
  const float* alibi_slopes;  // [H] or null: bias slope * (key - query) on the scaled logits
  int window;                 // sliding window (0 = none): a query sees the keys less than `window` positions back
};

// logit of (query position qp, key position key) after the softmax scale: ALiBi bias, causal / length mask
// (-inf) and the sliding window, which the reference masks with -10000 rather than -inf because a whole tile of
// a row may lie outside the window (prefix_prefill.py:130-144, :201-204; alibi :552-557)
__device__ __forceinline__ float fa_mask(float x, int key, int qp, int L, float slope, int window) {
  x += slope * (float)(key - qp);
  if (window > 0 && qp - key >= window) x = -10000.f;
  if (key > qp || key >= L) x = -INFINITY;
  return x;
}

// QU = 16-query sub-tiles per wave: 1 (64 queries per workgroup; short prompts, more workgroups) or
// 2 (128 queries: every K / V operand read from LDS feeds two MFMAs).
template <typename T, int D, int QU>
__global__ __launch_bounds__(256) void prefill_attention_kernel(const FaParams p) {
  constexpr int QT = FA_QT * QU;      // queries per workgroup
  static_assert(D % 32 == 0 && D <= 256, "head size");
  constexpr int DC = D / 32;          // 32-wide head-dim chunks (MFMA k-steps of Q.K^T)
  constexpr int DT = D / 16;          // 16-wide head-dim tiles of the output
  constexpr int VS = D + 8;           // LDS row stride of a K / V tile (elements): odd multiple of 16 B
                                      // -> conflict-free ds_read_b128 rows and transposed reads
  constexpr int TILE = FA_KT * VS;    // elements of one tile image
  constexpr int PIECES = FA_KT * D / 8;             // 16-byte pieces of a tile
  constexpr int PPT = (PIECES + 255) / 256;         // pieces per thread
  static_assert(PIECES % 256 == 0, "whole pieces per thread");
  // K and V tiles, double buffered: tile kt+1 is fetched into registers while tile kt is consumed
  __shared__ __attribute__((aligned(16))) uint16_t kv_s[2 * 2 * TILE];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int seq = blockIdx.z, head = blockIdx.y;
  const int kv_head = head / (p.num_heads / p.num_kv_heads);
  const int tok0 = p.cu_seqlens[seq];
  const int L = p.cu_seqlens[seq + 1] - tok0;
  const int qt0 = blockIdx.x * QT;
  if (qt0 >= L) return;  // uniform
  const int tr_off = (4 * g + (r >> 2)) * VS + 4 * (r & 3);  // this lane's address role in a tr read
  int q_row[QU];                                   // this lane's queries (prompt-relative)
#pragma unroll
  for (int u = 0; u < QU; ++u) q_row[u] = qt0 + (wave * QU + u) * 16 + r;

  // ---- Q fragments of the wave (B operand of S^T = K . Q^T): 8 head dims per lane and chunk ----
  uint4 qf[QU][DC];
#pragma unroll
  for (int u = 0; u < QU; ++u) {
    const uint16_t* qp = p.q + (int64_t)(tok0 + min(q_row[u], L - 1)) * p.q_stride + (int64_t)head * D + g * 8;
#pragma unroll
    for (int c = 0; c < DC; ++c) qf[u][c] = ld16(qp + c * 32);
  }
  f32x4_t o[QU][DT];
  float m_run[QU], l_run[QU];             // per lane = per query row (l: this lane's share)
#pragma unroll
  for (int u = 0; u < QU; ++u) {
    m_run[u] = -INFINITY;
    l_run[u] = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t) o[u][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  const uint16_t* kbase = p.k + (int64_t)tok0 * p.kv_stride + (int64_t)kv_head * D;
  const uint16_t* vbase = p.v + (int64_t)tok0 * p.kv_stride + (int64_t)kv_head * D;
  const int last_q = min(qt0 + QT, L) - 1;         // causal: keys 0 .. last_q
  const int n_kt = last_q / FA_KT + 1;
  // sliding window: the tiles wholly before the window of the workgroup's first query are skipped
  const int kt_lo = p.window > 0 ? max(qt0 - p.window + 1, 0) / FA_KT : 0;
  const float sc = p.scale;
  const float slope = p.alibi_slopes ? p.alibi_slopes[head] : 0.f;
  const bool every_tile = p.window > 0 || slope != 0.f;   // uniform: masks / bias touch every tile

  // tile loader: thread t owns pieces t, t+256, ... of the K and of the V tile (row = piece / (D/8))
  uint4 kreg[PPT], vreg[PPT];
  auto fetch = [&](int kt) {
    const int k0 = kt * FA_KT;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int id = threadIdx.x + i * 256;
      const int row = id / (D / 8), c8 = id % (D / 8);
      const int64_t off = (int64_t)min(k0 + row, L - 1) * p.kv_stride + c8 * 8;   // rows past L: clamped,
      kreg[i] = ld16(kbase + off);                                                // masked through P = 0
      vreg[i] = ld16(vbase + off);
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int id = threadIdx.x + i * 256;
      const int row = id / (D / 8), c8 = id % (D / 8);
      *reinterpret_cast<uint4*>(&kv_s[(buf * 2 + 0) * TILE + row * VS + c8 * 8]) = kreg[i];
      *reinterpret_cast<uint4*>(&kv_s[(buf * 2 + 1) * TILE + row * VS + c8 * 8]) = vreg[i];
    }
  };
  fetch(kt_lo);
  park(kt_lo & 1);
  __syncthreads();

  for (int kt = kt_lo; kt < n_kt; ++kt) {
    const int k0 = kt * FA_KT;
    const int buf = kt & 1;
    const uint16_t* k_s = &kv_s[(buf * 2 + 0) * TILE];
    const uint16_t* v_s = &kv_s[(buf * 2 + 1) * TILE];
    if (kt + 1 < n_kt) fetch(kt + 1);   // uniform; lands in registers while this tile is consumed
    // ---- S^T = K . Q^T for the 4 key tiles: K rows from LDS (shared by the 4 waves) ----
    f32x4_t s[QU][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int u = 0; u < QU; ++u) s[u][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const uint16_t* kp = k_s + (t * 16 + r) * VS + g * 8;
#pragma unroll
      for (int c = 0; c < DC; ++c) {
        const uint4 kf = *reinterpret_cast<const uint4*>(kp + c * 32);
#pragma unroll
        for (int u = 0; u < QU; ++u) s[u][t] = FaMfma<T>::run(kf, qf[u][c], s[u][t]);
      }
    }
    // ---- scale, causal / length mask, online softmax (row = this lane's query) ----
    const bool diag = every_tile || k0 + FA_KT - 1 > qt0;   // uniform: else only tiles that reach past the first query
    uint32_t pp[QU][8];                              // packed P: [t][pair]
#pragma unroll
    for (int u = 0; u < QU; ++u) {
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int key = k0 + t * 16 + 4 * g + i;
          float x = s[u][t][i] * sc;
          if (diag) x = fa_mask(x, key, q_row[u], L, slope, p.window);
          s[u][t][i] = x;
          mx = fmaxf(mx, x);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[u], mx);        // finite: the first tile holds a key at or before every query
      const float alpha = __expf(m_run[u] - m_new);
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          e[i] = __expf(s[u][t][i] - m_new);
          e[i] = T::to_float(T::from_float(e[i]));  // P in the model dtype, and the sum of the same
          psum += e[i];
        }
        pp[u][2 * t] = T::pack2(e[0], e[1]);
        pp[u][2 * t + 1] = T::pack2(e[2], e[3]);
      }
      l_run[u] = l_run[u] * alpha + psum;
      m_run[u] = m_new;
#pragma unroll
      for (int t = 0; t < DT; ++t) o[u][t] *= alpha;
    }
    // ---- O^T += V^T . P^T : two 32-key steps, DT output tiles ----
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        // A operand: V^T row d = 16 t + r, keys 32 st + {4g..4g+3, 16+4g..16+4g+3}: two hardware
        // transposed reads (ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of the
        // row-major V tile comes back column-major; lane 4q+p supplies row q, columns 4p..4p+3)
        const uint16_t* vp = &v_s[(32 * st) * VS + 16 * t + tr_off];
        const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)vp);
        const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(vp + 16 * VS));
        const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
        const uint4 vf = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
#pragma unroll
        for (int u = 0; u < QU; ++u)
          o[u][t] = FaMfma<T>::run(vf, make_uint4(pp[u][4 * st], pp[u][4 * st + 1], pp[u][4 * st + 2], pp[u][4 * st + 3]), o[u][t]);
      }
    }
    if (kt + 1 < n_kt) park(buf ^ 1);   // the other buffer's readers finished before the last barrier
    __syncthreads();
  }
  // ---- normalise and store: lane (q = r, g) holds d = 16 t + 4 g + i of its rows ----
#pragma unroll
  for (int u = 0; u < QU; ++u) {
    float l = l_run[u];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (q_row[u] >= L) continue;
    const float inv = 1.f / l;
    uint16_t* op = p.out + (int64_t)(tok0 + q_row[u]) * p.o_stride + (int64_t)head * D + 4 * g;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      uint2 pk;
      pk.x = T::pack2(o[u][t][0] * inv, o[u][t][1] * inv);
      pk.y = T::pack2(o[u][t][2] * inv, o[u][t][3] * inv);
      *reinterpret_cast<uint2*>(op + 16 * t) = pk;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Prefix-enabled prefill (chunked prefill / prefix caching): the new tokens of a prompt attend to
// the cached context AND to themselves (causal).  Behavioural reference:
// PagedAttention.forward_prefix (/root/reference/vllm/attention/ops/paged_attn.py:184-216) ->
// context_attention_fwd (ops/prefix_prefill.py, Triton).  The backend writes the new tokens' K/V
// into the paged cache before attention (rocm_flash_attn.py:318-331), so with an unquantised cache
// every key -- context and new -- is read from the cache: K [NB, KVH, D/8, BS, 8] gives the 16-byte
// MFMA operand of a key directly (8 head dims), V [NB, KVH, D, BS] gives the V^T operand of a head
// dim as two 8-byte runs of 4 consecutive tokens; no LDS at all.  Same transposed formulation as
// the kernel above.
struct PfxParams {
  const uint16_t* q;    // [new tokens, H, D]
  uint16_t* out;
  const uint16_t* kc;   // key cache
  const uint16_t* vc;   // value cache
  const int* block_tables;   // [num_seqs, max_blocks]
  const int* q_start;        // [num_seqs + 1] offsets of the new tokens
  const int* seq_lens;       // [num_seqs] context + new
  const int* ctx_lens;       // [num_seqs]
  int64_t q_stride, o_stride, kv_block_stride, kv_head_stride;  // elements
  int max_blocks, block_size, num_heads, num_kv_heads;
  float scale;
  const float* alibi_slopes;  // [H] or null
  int window;                 // sliding window, 0 = none
};

template <typename T, int D>
__global__ __launch_bounds__(256) void prefix_attention_kernel(const PfxParams p) {
  constexpr int DC = D / 32, DT = D / 16;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int seq = blockIdx.z, head = blockIdx.y;
  const int kv_head = head / (p.num_heads / p.num_kv_heads);
  const int tok0 = p.q_start[seq];
  const int q_len = p.q_start[seq + 1] - tok0;
  const int L = p.seq_lens[seq];
  const int ctx = p.ctx_lens[seq];
  const int qt0 = blockIdx.x * FA_QT;
  if (qt0 >= q_len) return;  // uniform
  const int q_row = qt0 + wave * 16 + r;          // new-token index of this lane's query
  const int q_pos = ctx + q_row;                  // its position in the sequence
  const int* bt = p.block_tables + (int64_t)seq * p.max_blocks;
  const int bs = p.block_size;
  const int64_t head_off = (int64_t)kv_head * p.kv_head_stride;

  uint4 qf[DC];
  {
    const uint16_t* qp = p.q + (int64_t)(tok0 + min(q_row, q_len - 1)) * p.q_stride + (int64_t)head * D + g * 8;
#pragma unroll
    for (int c = 0; c < DC; ++c) qf[c] = ld16(qp + c * 32);
  }
  f32x4_t o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t) o[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const int last_pos = ctx + min(qt0 + FA_QT, q_len) - 1;   // causal: keys 0 .. last_pos (< L)
  const int n_kt = last_pos / FA_KT + 1;
  const int kt_lo = p.window > 0 ? max(ctx + qt0 - p.window + 1, 0) / FA_KT : 0;
  const float sc = p.scale;
  const float slope = p.alibi_slopes ? p.alibi_slopes[head] : 0.f;
  const bool every_tile = p.window > 0 || slope != 0.f;   // uniform

  for (int kt = kt_lo; kt < n_kt; ++kt) {
    const int k0 = kt * FA_KT;
    // ---- S^T = K . Q^T, keys gathered through the block table ----
    f32x4_t s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      s[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const int key = min(k0 + t * 16 + r, L - 1);
      const uint16_t* kp = p.kc + (int64_t)bt[key / bs] * p.kv_block_stride + head_off + (key % bs) * 8;
#pragma unroll
      for (int c = 0; c < DC; ++c) s[t] = FaMfma<T>::run(ld16(kp + (int64_t)(c * 4 + g) * bs * 8), qf[c], s[t]);
    }
    const bool diag = every_tile || k0 + FA_KT - 1 > ctx + qt0;   // uniform
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int key = k0 + t * 16 + 4 * g + i;
        float x = s[t][i] * sc;
        if (diag) x = fa_mask(x, key, q_pos, L, slope, p.window);
        s[t][i] = x;
        mx = fmaxf(mx, x);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
    uint32_t pp[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        e[i] = T::to_float(T::from_float(__expf(s[t][i] - m_new)));
        psum += e[i];
      }
      pp[2 * t] = T::pack2(e[0], e[1]);
      pp[2 * t + 1] = T::pack2(e[2], e[3]);
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] *= alpha;
    // ---- O^T += V^T . P^T: V^T row d = 16 t + r, keys 32 st + {4g..4g+3, 16+4g..16+4g+3} ----
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const uint4 pb = make_uint4(pp[4 * st], pp[4 * st + 1], pp[4 * st + 2], pp[4 * st + 3]);
      const int ka = k0 + 32 * st + 4 * g, kb = ka + 16;
      // tokens past the end of the sequence may be uninitialised cache (NaN x 0 = NaN): zero them
      const int va = min(max(L - ka, 0), 4), vb = min(max(L - kb, 0), 4);
      const uint2 ma = make_uint2(va >= 2 ? 0xffffffffu : (va == 1 ? 0xffffu : 0u),
                                  va >= 4 ? 0xffffffffu : (va == 3 ? 0xffffu : 0u));
      const uint2 mb = make_uint2(vb >= 2 ? 0xffffffffu : (vb == 1 ? 0xffffu : 0u),
                                  vb >= 4 ? 0xffffffffu : (vb == 3 ? 0xffffu : 0u));
      const int kac = min(ka, L - 1) & ~3, kbc = min(kb, L - 1) & ~3;
      const uint16_t* vpa = p.vc + (int64_t)bt[kac / bs] * p.kv_block_stride + head_off + (kac % bs);
      const uint16_t* vpb = p.vc + (int64_t)bt[kbc / bs] * p.kv_block_stride + head_off + (kbc % bs);
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const uint2 xa = ld8(vpa + (int64_t)(16 * t + r) * bs);
        const uint2 xb = ld8(vpb + (int64_t)(16 * t + r) * bs);
        o[t] = FaMfma<T>::run(make_uint4(xa.x & ma.x, xa.y & ma.y, xb.x & mb.x, xb.y & mb.y), pb, o[t]);
      }
    }
  }
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  if (q_row >= q_len) return;
  const float inv = 1.f / l_run;
  uint16_t* op = p.out + (int64_t)(tok0 + q_row) * p.o_stride + (int64_t)head * D + 4 * g;
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    uint2 pk;
    pk.x = T::pack2(o[t][0] * inv, o[t][1] * inv);
    pk.y = T::pack2(o[t][2] * inv, o[t][3] * inv);
    *reinterpret_cast<uint2*>(op + 16 * t) = pk;
  }
}

}  // namespace nmv

using namespace nmv;

extern "C" int nmv_prefill_attention_supported(int head_size) { return head_size == 64 || head_size == 128; }

extern "C" int nmv_prefill_attention(void* out, const void* q, const void* k, const void* v,
                                     const int32_t* cu_seqlens, int num_seqs, int max_seq_len,
                                     int num_heads, int num_kv_heads, int head_size, float scale,
                                     int64_t q_stride, int64_t kv_stride, int64_t o_stride,
                                     const float* alibi_slopes, int sliding_window,
                                     nmv_dtype_t dtype, void* stream) {
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16, "prefill_attention: fp16 / bf16 only");
  NMV_CHECK(head_size == 64 || head_size == 128, "prefill_attention: head size %d not built (64, 128)", head_size);
  NMV_CHECK(num_kv_heads > 0 && num_heads % num_kv_heads == 0, "prefill_attention: heads %% kv_heads != 0");
  NMV_CHECK(q_stride % 8 == 0 && kv_stride % 8 == 0 && o_stride % 4 == 0,
            "prefill_attention: token strides must keep 16-byte (q, k, v) / 8-byte (out) alignment");
  if (num_seqs <= 0 || max_seq_len <= 0) return NMV_OK;
  FaParams p{(const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, (uint16_t*)out, cu_seqlens,
             q_stride, kv_stride, o_stride, num_heads, num_kv_heads, scale, alibi_slopes,
             sliding_window > 0 ? sliding_window : 0};
  hipStream_t s = (hipStream_t)stream;
  // QU = 2 (128-query workgroups, each K / V operand feeding two MFMAs) was measured slower at every
  // size (L = 8192: 2.09 ms vs 1.44 ms: twice the registers, more masked work on the diagonal)
  const int qu = 1;
  dim3 grid((max_seq_len + FA_QT * qu - 1) / (FA_QT * qu), num_heads, num_seqs), block(256);
#define NMV_FA_LAUNCH(T_, D_) hipLaunchKernelGGL((prefill_attention_kernel<T_, D_, 1>), grid, block, 0, s, p);
  if (dtype == NMV_BF16) {
    if (head_size == 64) NMV_FA_LAUNCH(BF16, 64) else NMV_FA_LAUNCH(BF16, 128)
  } else {
    if (head_size == 64) NMV_FA_LAUNCH(F16, 64) else NMV_FA_LAUNCH(F16, 128)
  }
#undef NMV_FA_LAUNCH
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_prefix_prefill_attention(void* out, const void* q, const void* key_cache,
                                            const void* value_cache, const int32_t* block_tables,
                                            const int32_t* query_start_loc, const int32_t* seq_lens,
                                            const int32_t* context_lens, int num_seqs,
                                            int max_query_len, int max_blocks_per_seq, int block_size,
                                            int num_heads, int num_kv_heads, int head_size, float scale,
                                            int64_t q_stride, int64_t o_stride, int64_t kv_block_stride,
                                            int64_t kv_head_stride, const float* alibi_slopes,
                                            int sliding_window, nmv_dtype_t dtype, void* stream) {
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16, "prefix_prefill_attention: fp16 / bf16 only (kv cache dtype auto)");
  NMV_CHECK(head_size == 64 || head_size == 128, "prefix_prefill_attention: head size %d not built (64, 128)", head_size);
  NMV_CHECK(block_size == 8 || block_size == 16 || block_size == 32, "prefix_prefill_attention: block size %d", block_size);
  NMV_CHECK(num_kv_heads > 0 && num_heads % num_kv_heads == 0, "prefix_prefill_attention: heads %% kv_heads != 0");
  NMV_CHECK(q_stride % 8 == 0 && o_stride % 4 == 0, "prefix_prefill_attention: token strides break alignment");
  if (num_seqs <= 0 || max_query_len <= 0) return NMV_OK;
  PfxParams p{(const uint16_t*)q, (uint16_t*)out, (const uint16_t*)key_cache, (const uint16_t*)value_cache,
              block_tables, query_start_loc, seq_lens, context_lens, q_stride, o_stride, kv_block_stride,
              kv_head_stride, max_blocks_per_seq, block_size, num_heads, num_kv_heads, scale, alibi_slopes,
              sliding_window > 0 ? sliding_window : 0};
  dim3 grid((max_query_len + FA_QT - 1) / FA_QT, num_heads, num_seqs), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == NMV_BF16) {
    if (head_size == 64) hipLaunchKernelGGL((prefix_attention_kernel<BF16, 64>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((prefix_attention_kernel<BF16, 128>), grid, block, 0, s, p);
  } else {
    if (head_size == 64) hipLaunchKernelGGL((prefix_attention_kernel<F16, 64>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((prefix_attention_kernel<F16, 128>), grid, block, 0, s, p);
  }
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}
