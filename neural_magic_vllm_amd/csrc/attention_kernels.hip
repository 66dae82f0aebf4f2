// Paged attention (single-query decode) for gfx950 / wave64.
//
// Behavioural reference: /root/reference/csrc/attention/attention_kernels.cu
//   (paged_attention_kernel :86-496, v2 reduce :564-669, launchers :690-990).
// Same inputs, same cache layouts (K [NB,kvh,D/x,BS,x], V [NB,kvh,D,BS]), same outputs
// (incl. exp_sums / max_logits / tmp_out of v2), same 1/(sum+1e-6) normaliser.
//
// The decomposition is NOT the reference's.  The reference runs one 128-thread block per
// *query* head and derives THREAD_GROUP_SIZE = WARP_SIZE / BLOCK_SIZE; with GQA that re-reads
// every KV head num_queries_per_kv times.  Here one 256-thread workgroup (4 waves) serves a
// GROUP of HG query heads that share one KV head, so K/V bytes cross HBM once per group:
//
//   * a wave walks 64-token windows; lane t owns token t of the window for the QK^T pass: it
//     streams that token's head vector as HEAD_SIZE/x 16-byte chunks straight into VGPRs
//     (16 tokens x 16 B = 256 contiguous bytes per chunk row of a block) and dots it against
//     the HG queries, which are broadcast from LDS (ds_read_b128 of one address = no conflict);
//     packed v_dot2c_f32_bf16 / v_dot2_f32_f16 do two MACs per lane-op with fp32 accumulate.
//     No cross-lane traffic is needed for a logit.
//   * softmax is online (flash-decoding): per window one wave-wide max (64-lane xor-shuffle
//     reduction), the running row-sum stays per lane and is reduced once at the end.
//   * for P.V the lanes re-map onto the V layout: a wave-load covers 64 consecutive 16-byte
//     chunks of one (block, head) = 1 KiB contiguous; probabilities go through a 128-byte/head
//     LDS strip (written by the token-owning lanes, read back as broadcast 16-byte vectors).
//   * the 4 waves are merged through LDS at the end (max / sum / fp32 accumulators).
//
// There is no logits[max_seq_len] buffer, so v1 has no LDS-imposed context limit.
// HBM-bound by construction: algorithmic bytes = 2 * L * D * sizeof(cache_t) per (seq, kv head).
#include "common.h"
#include "cache_write.h"
#include "fp32_path.h"

namespace nmv {

constexpr int PA_WIN = 64;  // tokens per wave-iteration
constexpr int PA_PARTITION = 512;

template <int HEAD_SIZE, int BLOCK_SIZE, int HG, bool FP8>
struct PAGeom {
  static constexpr int CB = FP8 ? 1 : 2;            // cache element bytes
  static constexpr int EPC = 16 / CB;               // elements per 16-byte K chunk (= x)
  static constexpr int NKC = HEAD_SIZE / EPC;       // K chunks per token
  static constexpr int PPC = EPC / 2;               // packed pairs per K chunk
  static constexpr int TPCV = EPC < BLOCK_SIZE ? EPC : BLOCK_SIZE;  // tokens per V chunk
  static constexpr int VB = TPCV * CB;              // V chunk bytes (16, or 8 for fp8 + BS 8)
  static constexpr int CPR = BLOCK_SIZE / TPCV;     // V chunks per row of a block
  static constexpr int RPL = WAVE / CPR;            // V rows covered by one wave-load
  static constexpr int NVL = (HEAD_SIZE + RPL - 1) / RPL;  // wave-loads per block
  static constexpr int WB = PA_WIN / BLOCK_SIZE;    // blocks per window
#ifndef NMV_PA_KG
#define NMV_PA_KG 8
#endif
#ifndef NMV_PA_VG
#define NMV_PA_VG 4
#endif
  static constexpr int KG = NKC < NMV_PA_KG ? NKC : NMV_PA_KG;  // K chunks in flight per lane
  // register budget: two waves per SIMD (256 VGPRs each) unless the fp32 accumulators alone take 48 of them --
  // without the hint the 4-wave forms may use 512 and the block-sparse one did (414 registers, one wave per SIMD)
  static constexpr int MIN_WAVES = (HG * NVL >= 48) ? 1 : 2;
  static_assert(HEAD_SIZE % EPC == 0, "head size must be a multiple of x");
  static_assert(PA_WIN % BLOCK_SIZE == 0, "block size must divide the window");
};

// fp8 x4 (one dword) -> two packed T pairs
template <typename T>
__device__ __forceinline__ void fp8x4_to_pairs(uint32_t w, uint32_t& p0, uint32_t& p1) {
  f32x2_t a = fp8x2_to_f32<false>(w);
  f32x2_t b = fp8x2_to_f32<true>(w);
  p0 = T::pack2(a.x, a.y);
  p1 = T::pack2(b.x, b.y);
}

// Fused decode prologue (nmv_paged_attention_*_rope_partial; not in the reference): the qkv
// projection's output is still its fp32 split-K slabs [splits][num_seqs][row_elems].  The workgroup
// sums and rounds the query heads of its group and this kv head's new key / value (what the GEMM
// would have stored), applies neox rotary embedding to q and k exactly as rotary_embedding does,
// keeps q in LDS, and -- if the new token falls into this workgroup's token range -- stores k / v
// into the paged cache before it walks the cache (its own stores are visible to it after the
// barrier; workgroups that share the kv head store identical bytes).  One launch less per layer
// than rope + cache write followed by attention.  The same prologue also accepts the finished qkv row
// in the model dtype (W8A8 / unquantised projections).  slab == qkv == nullptr: the plain kernel.
struct PAFused {
  const float* slab;            // fp32 split-K slabs of the qkv projection, or
  const uint16_t* qkv;          // the finished qkv row [num_seqs, row_elems] in the model dtype (slab == nullptr)
  int splits;
  int64_t slab_stride;          // elements between slabs = num_seqs * row_elems
  int row_elems;                // (num_heads + 2 num_kv_heads) * head_size
  const int64_t* positions;     // [num_seqs] position of the new token (= seq_len - 1)
  const uint16_t* cos_sin;      // [max_pos, head_size] model dtype: cos | sin
  const int64_t* slot_mapping;  // [num_seqs] cache slot of the new token (< 0: padding)
};

// Block-sparse attention (attention_kernels.cu:209-251, 385-393): the context is cut into blocks of `block_size`
// tokens; a head attends to a block when it is one of the last `local_blocks` blocks of the sequence ("local")
// or when (block id + head offset) is a multiple of `vert_stride` ("remote"); the head offset slides with the
// query head (head_sliding_step >= 0) or with the kv head (< 0).  vert_stride <= 1: dense.  Every other token
// is left out of the softmax (the reference stores -FLT_MAX for it and skips its V block).
struct PASparse {
  int tp_rank, local_blocks, vert_stride, block_size, head_sliding_step;
};

// NW waves per workgroup: 4 when the grid fills the chip, 8 when it does not (few sequences):
// the same KV run is then split over twice the waves and the per-wave chain of dependent windows
// halves (B=1, L=530: 14.7 -> 10.9 us; at B=64 the 4-wave form is 15 % faster).
template <typename T, bool FP8, int HEAD_SIZE, int BLOCK_SIZE, int HG, int NW, bool SPARSE = false>
__global__ __launch_bounds__(NW * WAVE, (PAGeom<HEAD_SIZE, BLOCK_SIZE, HG, FP8>::MIN_WAVES)) void paged_attention_kernel(
    float* __restrict__ exp_sums,    // [num_seqs, num_heads, max_num_partitions] (partitioned only)
    float* __restrict__ max_logits,  // same
    uint16_t* __restrict__ out,      // [num_seqs, num_heads, (max_num_partitions,) head_size]
    const uint16_t* __restrict__ q,  // [num_seqs, num_heads, head_size], row stride q_stride
    const uint8_t* k_cache, const uint8_t* v_cache,  // no restrict: the fused prologue stores the new token
    int num_heads,
    int num_kv_heads, float scale, const int* __restrict__ block_tables,
    const int* __restrict__ seq_lens, int max_num_blocks_per_seq,
    const float* __restrict__ alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
    int64_t kv_head_stride, float kv_scale, int partition_size /* 0 = not partitioned */,
    const PAFused f, const PASparse sp) {
  using G = PAGeom<HEAD_SIZE, BLOCK_SIZE, HG, FP8>;
  const int seq_idx = blockIdx.y;
  const int part_idx = blockIdx.z;
  const int max_num_partitions = gridDim.z;
  const int seq_len = seq_lens[seq_idx];
  const int start_tok = partition_size ? part_idx * partition_size : 0;
  if (start_tok >= seq_len) return;  // uniform for the whole workgroup
  const int end_tok = partition_size ? min(start_tok + partition_size, seq_len) : seq_len;

  const int head0 = blockIdx.x * HG;
  const int q_per_kv = num_heads / num_kv_heads;
  const int kv_head = head0 / q_per_kv;
  const int lane = threadIdx.x & 63;
  // wave id, made provably wave-uniform
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  __shared__ __attribute__((aligned(16))) uint32_t q_s[HG * HEAD_SIZE / 2];
  __shared__ __attribute__((aligned(16))) uint32_t p_s[NW][HG][PA_WIN / 2];
  __shared__ float red_m[NW][HG];
  __shared__ float red_l[NW][HG];
  __shared__ float out_red[NW][HG][HEAD_SIZE];

  // ---- queries of the head group -> LDS (HG*HEAD_SIZE contiguous elements) ----
  if (f.slab == nullptr && f.qkv == nullptr) {
    const uint32_t* q_ptr =
        reinterpret_cast<const uint32_t*>(q + (int64_t)seq_idx * q_stride + (int64_t)head0 * HEAD_SIZE);
    for (int i = threadIdx.x; i < HG * HEAD_SIZE / 2; i += (NW * WAVE)) q_s[i] = q_ptr[i];
  } else {
    constexpr int EMBED = HEAD_SIZE / 2, QUADS = EMBED / 4;
    const int64_t pos = f.positions[seq_idx];
    const int64_t slot = f.slot_mapping[seq_idx];
    // the new token is cached by the workgroup(s) whose token range holds it
    const bool cached = slot >= 0 && pos >= start_tok && pos < end_tok;
    const int64_t blk_idx = cached ? slot / BLOCK_SIZE : 0, blk_off = cached ? slot % BLOCK_SIZE : 0;
    const float* row = f.slab + (int64_t)seq_idx * f.row_elems;
    const uint16_t* row16 = f.qkv + (int64_t)seq_idx * f.row_elems;
    const uint16_t* cos_ptr = f.cos_sin + pos * HEAD_SIZE;
    const uint16_t* sin_ptr = cos_ptr + EMBED;
    // sums of the split-K slabs at `col` and (PAIR) at `col + EMBED`: one loop, both loads of a slab in flight
    // together (two calls = two dependent round trips per rotary pair; r03)
    auto slab_sum4 = [&](int col, float (&o)[4], float (&o2)[4], auto pair) {
      constexpr bool PAIR = decltype(pair)::value;
      if (f.slab == nullptr) {  // uniform: finished row in the model dtype
        const uint2 w = *reinterpret_cast<const uint2*>(row16 + col);
        o[0] = lo_f<T>(w.x), o[1] = hi_f<T>(w.x), o[2] = lo_f<T>(w.y), o[3] = hi_f<T>(w.y);
        if constexpr (PAIR) {
          const uint2 w2 = *reinterpret_cast<const uint2*>(row16 + col + EMBED);
          o2[0] = lo_f<T>(w2.x), o2[1] = hi_f<T>(w2.x), o2[2] = lo_f<T>(w2.y), o2[3] = hi_f<T>(w2.y);
        }
        return;
      }
      f32x4_t acc4 = {0.f, 0.f, 0.f, 0.f}, acc4b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int sp = 0; sp < f.splits; ++sp) {
        acc4 += *reinterpret_cast<const f32x4_t*>(row + sp * f.slab_stride + col);
        if constexpr (PAIR) acc4b += *reinterpret_cast<const f32x4_t*>(row + sp * f.slab_stride + col + EMBED);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[i] = round_trip<T>(acc4[i]);  // the GEMM's output rounding
        if constexpr (PAIR) o2[i] = round_trip<T>(acc4b[i]);
      }
    };
    constexpr int N_Q = HG * QUADS, N_K = QUADS, N_V = HEAD_SIZE / 4;
    for (int it = threadIdx.x; it < N_Q + N_K + N_V; it += (NW * WAVE)) {
      if (it < N_Q + N_K) {
        const bool is_k = it >= N_Q;
        const int h = is_k ? 0 : it / QUADS;
        const int d0 = ((is_k ? it - N_Q : it) % QUADS) * 4;
        const int col = (is_k ? num_heads + kv_head : head0 + h) * HEAD_SIZE + d0;
        float x[4], y[4];
        const uint2 cw = *reinterpret_cast<const uint2*>(cos_ptr + d0);
        const uint2 sw = *reinterpret_cast<const uint2*>(sin_ptr + d0);
        slab_sum4(col, x, y, std::true_type{});
        const uint32_t cs[2] = {cw.x, cw.y}, sn[2] = {sw.x, sw.y};
        uint16_t xo[4], yo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float c = (i & 1) ? hi_f<T>(cs[i >> 1]) : lo_f<T>(cs[i >> 1]);
          const float sv = (i & 1) ? hi_f<T>(sn[i >> 1]) : lo_f<T>(sn[i >> 1]);
          // pos_encoding_kernels.cu:10-37 with every product rounded to the model dtype
          xo[i] = T::from_float(round_trip<T>(x[i] * c) - round_trip<T>(y[i] * sv));
          yo[i] = T::from_float(round_trip<T>(y[i] * c) + round_trip<T>(x[i] * sv));
        }
        if (!is_k) {
          uint32_t* qx = q_s + (h * HEAD_SIZE + d0) / 2;
          qx[0] = xo[0] | ((uint32_t)xo[1] << 16);
          qx[1] = xo[2] | ((uint32_t)xo[3] << 16);
          qx[EMBED / 2] = yo[0] | ((uint32_t)yo[1] << 16);
          qx[EMBED / 2 + 1] = yo[2] | ((uint32_t)yo[3] << 16);
        } else if (cached) {
          const int64_t hb = blk_idx * num_kv_heads + kv_head;
          void* kc = const_cast<uint8_t*>(k_cache);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            cache_store_k<T, FP8>(kc, hb, HEAD_SIZE, BLOCK_SIZE, blk_off, d0 + i, xo[i], kv_scale);
            cache_store_k<T, FP8>(kc, hb, HEAD_SIZE, BLOCK_SIZE, blk_off, d0 + EMBED + i, yo[i], kv_scale);
          }
        }
      } else {
        const int e0 = (it - N_Q - N_K) * 4;
        float v[4];
        slab_sum4((num_heads + num_kv_heads + kv_head) * HEAD_SIZE + e0, v, v, std::false_type{});
        if (cached) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int64_t tgt = ((blk_idx * num_kv_heads + kv_head) * HEAD_SIZE + e0 + i) * BLOCK_SIZE + blk_off;
            const uint16_t vb = T::from_float(v[i]);
            if constexpr (!FP8)
              reinterpret_cast<uint16_t*>(const_cast<uint8_t*>(v_cache))[tgt] = vb;
            else
              const_cast<uint8_t*>(v_cache)[tgt] = f32_to_fp8(T::to_float(vb) / kv_scale);
          }
        }
      }
    }
  }
  __syncthreads();

  float slope[HG];
#pragma unroll
  for (int h = 0; h < HG; ++h) slope[h] = alibi_slopes ? alibi_slopes[head0 + h] : 0.f;

  float m_run[HG], l_lane[HG], acc[HG][G::NVL];
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    m_run[h] = -INFINITY;
    l_lane[h] = 0.f;
#pragma unroll
    for (int i = 0; i < G::NVL; ++i) acc[h][i] = 0.f;
  }

  const int* block_table = block_tables + (int64_t)seq_idx * max_num_blocks_per_seq;
  // Each wave takes a CONTIGUOUS run of KV blocks, the runs differing by at most one block
  // (start_tok is block aligned: partitions are multiples of the block size).  Dealing 64-token
  // windows round-robin instead leaves 9 windows as 3+2+2+2: the first wave then sets the time of
  // the workgroup, +25..50 % just past every multiple of 256 tokens.
  // block-sparse with sparsity blocks of whole windows (the usual 64 tokens): the runs are dealt in windows, so that
  // a window lies inside ONE sparsity block and "does head h attend to it" is a per-window scalar -- no per-lane
  // integer divisions, and unattended windows are skipped before their K / V are loaded (below)
  bool sp_u = false;
  int sp_off[HG], sp_last = 0;
  if constexpr (SPARSE) {
    sp_u = sp.block_size % PA_WIN == 0;
    sp_last = (seq_len - 1) / sp.block_size - sp.local_blocks;
#pragma unroll
    for (int h = 0; h < HG; ++h)
      sp_off[h] = sp.head_sliding_step >= 0 ? (sp.tp_rank * num_heads + head0 + h) * sp.head_sliding_step + 1
                                            : (sp.tp_rank * num_kv_heads + kv_head) * (-sp.head_sliding_step) + 1;
  }
  const int unit = sp_u ? PA_WIN : BLOCK_SIZE;
  const int n_blk = (end_tok - start_tok + unit - 1) / unit;
  const int blk_lo = (n_blk / NW) * wave + min(wave, n_blk % NW);
  const int blk_cnt = n_blk / NW + (wave < n_blk % NW ? 1 : 0);
  const int w_tok0 = start_tok + blk_lo * unit;                             // this wave's tokens:
  const int w_tok1 = min(w_tok0 + blk_cnt * unit, end_tok);                 // [w_tok0, w_tok1)
  const float qk_scale = FP8 ? scale * kv_scale : scale;
  const int64_t head_off_bytes = (int64_t)kv_head * kv_head_stride * G::CB;
  const int64_t block_stride_bytes = kv_block_stride * G::CB;

  // V-phase lane geometry
  const int cpr_idx = lane % G::CPR;
  const int row_l = lane / G::CPR;

  auto phys_of = [&](int wstart) -> int {
    const int tok = min(wstart + lane, w_tok1 - 1);
    return block_table[tok / BLOCK_SIZE];
  };
  int phys = w_tok0 < w_tok1 ? phys_of(w_tok0) : 0;

  for (int wstart = w_tok0; wstart < w_tok1; wstart += PA_WIN) {
    const int tok = wstart + lane;
    const bool valid = tok < w_tok1;
    const int tok_c = valid ? tok : w_tok1 - 1;
    const int boff = tok_c % BLOCK_SIZE;
    // block-table lookup of the NEXT window issued before this window's loads are consumed
    const int phys_next = (wstart + PA_WIN < w_tok1) ? phys_of(wstart + PA_WIN) : 0;

    // block-sparse: a window none of whose tokens any head of the group attends to is skipped before its K / V are
    // loaded (the reference skips masked blocks the same way, attention_kernels.cu:209-251).  Exact: such a window
    // would contribute p = 0 and rescale by alpha = 1.
    bool att_u[HG];   // sp_u: per-window, per-head scalars
    if constexpr (SPARSE) {
      bool any;
      if (sp_u) {
        const int kb_u = __builtin_amdgcn_readfirstlane(wstart / sp.block_size);
        const bool local_u = kb_u > sp_last;
        any = false;
#pragma unroll
        for (int h = 0; h < HG; ++h) {
          att_u[h] = local_u || (kb_u + sp_off[h]) % sp.vert_stride == 0;
          any = any || att_u[h];
        }
      } else {
        const int kb_w = (tok_c / BLOCK_SIZE) * BLOCK_SIZE / sp.block_size;
        bool lane_any = kb_w > sp_last;
#pragma unroll
        for (int h = 0; h < HG; ++h) lane_any = lane_any || (kb_w + sp_off[h]) % sp.vert_stride == 0;
        any = __ballot(valid && lane_any) != 0;
      }
      if (!any) {   // wave-uniform
        phys = phys_next;
        continue;
      }
    }

    // ================= Q.K^T : lane = token =================
    const uint8_t* kp = k_cache + (int64_t)phys * block_stride_bytes + head_off_bytes + boff * 16;
    float s[HG];
#pragma unroll
    for (int h = 0; h < HG; ++h) s[h] = 0.f;
#pragma unroll
    for (int c0 = 0; c0 < G::NKC; c0 += G::KG) {
      uint4 kc[G::KG];
#pragma unroll
      for (int j = 0; j < G::KG; ++j)
        if (c0 + j < G::NKC) kc[j] = ld16(kp + (int64_t)(c0 + j) * (BLOCK_SIZE * 16));
#pragma unroll
      for (int j = 0; j < G::KG; ++j) {
        if (c0 + j >= G::NKC) continue;
        if constexpr (!FP8) {
#pragma unroll
          for (int h = 0; h < HG; ++h) {
            const uint4 qv = *reinterpret_cast<const uint4*>(&q_s[h * (HEAD_SIZE / 2) + (c0 + j) * 4]);
            float a = s[h];
            a = T::dot2(kc[j].x, qv.x, a);
            a = T::dot2(kc[j].y, qv.y, a);
            a = T::dot2(kc[j].z, qv.z, a);
            a = T::dot2(kc[j].w, qv.w, a);
            s[h] = a;
          }
        } else {
          uint32_t kpair[8];
          fp8x4_to_pairs<T>(kc[j].x, kpair[0], kpair[1]);
          fp8x4_to_pairs<T>(kc[j].y, kpair[2], kpair[3]);
          fp8x4_to_pairs<T>(kc[j].z, kpair[4], kpair[5]);
          fp8x4_to_pairs<T>(kc[j].w, kpair[6], kpair[7]);
#pragma unroll
          for (int h = 0; h < HG; ++h) {
            const uint4 q0 = *reinterpret_cast<const uint4*>(&q_s[h * (HEAD_SIZE / 2) + (c0 + j) * 8]);
            const uint4 q1 = *reinterpret_cast<const uint4*>(&q_s[h * (HEAD_SIZE / 2) + (c0 + j) * 8 + 4]);
            float a = s[h];
            a = T::dot2(kpair[0], q0.x, a);
            a = T::dot2(kpair[1], q0.y, a);
            a = T::dot2(kpair[2], q0.z, a);
            a = T::dot2(kpair[3], q0.w, a);
            a = T::dot2(kpair[4], q1.x, a);
            a = T::dot2(kpair[5], q1.y, a);
            a = T::dot2(kpair[6], q1.z, a);
            a = T::dot2(kpair[7], q1.w, a);
            s[h] = a;
          }
        }
        // block-sparse form: left alone, the LDS reads of the queries of ALL chunks are issued first (256 VGPRs, then
        // spills); the empty asm pins the dot products of a pair of chunks before the next pair's reads
#ifndef NMV_PA_PIN
#define NMV_PA_PIN 0
#endif
        if constexpr (SPARSE || NMV_PA_PIN) {
          if ((j & 1) == 1) {
#pragma unroll
            for (int h = 0; h < HG; ++h) asm volatile("" : "+v"(s[h]));
          }
        }
      }
    }

    // the first V group of the window does not depend on the probabilities: its loads are issued here and land
    // under the softmax (max reduction, exp, LDS strip) instead of after it (measured: -1..2 % at B = 64,
    // -4..6 % at B = 8; 172 VGPRs instead of 148, still two waves per SIMD)
    constexpr int VG = G::NVL < NMV_PA_VG ? G::NVL : NMV_PA_VG;  // V wave-loads in flight per lane
    auto load_vgroup = [&](int b, int i0, uint32_t (&vraw)[VG][4]) {
      const int physb = __builtin_amdgcn_readlane(phys, b * BLOCK_SIZE);
      const uint8_t* vp = v_cache + (int64_t)physb * block_stride_bytes + head_off_bytes + (int64_t)cpr_idx * G::VB;
#pragma unroll
      for (int ii = 0; ii < VG; ++ii) {
        if (i0 + ii >= G::NVL) continue;
        const int row = (i0 + ii) * G::RPL + row_l;
        const int rowc = (HEAD_SIZE % G::RPL == 0) ? row : min(row, HEAD_SIZE - 1);
        const uint8_t* a = vp + (int64_t)rowc * (BLOCK_SIZE * G::CB);
        if constexpr (G::VB == 16) {
          const uint4 t = ld16(a);
          vraw[ii][0] = t.x; vraw[ii][1] = t.y; vraw[ii][2] = t.z; vraw[ii][3] = t.w;
        } else {
          const uint2 t = ld8(a);
          vraw[ii][0] = t.x; vraw[ii][1] = t.y; vraw[ii][2] = 0; vraw[ii][3] = 0;
        }
      }
    };
    uint32_t vpre[VG][4];
    load_vgroup(0, 0, vpre);

    // ================= online softmax =================
    // block-sparse: the sparsity block of this lane's token, by KV block as the reference computes it
    // (SPARSE is a template parameter: as a run-time flag the integer divisions below cost the dense kernel
    // 35 % at B = 64 and 4x at B = 1)
    int kb = 0;
    bool kb_local = false;
    if constexpr (SPARSE) {
      if (!sp_u) {
        kb = (tok_c / BLOCK_SIZE) * BLOCK_SIZE / sp.block_size;
        kb_local = kb > sp_last;
      }
    }
#pragma unroll
    for (int h = 0; h < HG; ++h) {
      float sv = s[h] * qk_scale;
      sv += (slope[h] != 0.f) ? slope[h] * (float)(tok - seq_len + 1) : 0.f;
      bool attend = valid;
      if constexpr (SPARSE)
        attend = valid && (sp_u ? att_u[h] : (kb_local || (kb + sp_off[h]) % sp.vert_stride == 0));
      sv = attend ? sv : -INFINITY;
      const float m_new = fmaxf(m_run[h], wave_max(sv));
      // exp(-inf) = 0 on the first window; a window in which this head attends to nothing (block-sparse) leaves
      // m_new at -inf: nothing has been accumulated yet, 0 keeps it so (and avoids exp(-inf + inf))
      const float alpha = m_new == -INFINITY ? 0.f : __expf(m_run[h] - m_new);
      const float p = attend ? __expf(sv - m_new) : 0.f;
      l_lane[h] = l_lane[h] * alpha + p;
      m_run[h] = m_new;
#pragma unroll
      for (int i = 0; i < G::NVL; ++i) acc[h][i] *= alpha;
      // probabilities are rounded to the cache's compute dtype before P.V, as the reference
      // does (attention_kernels.cu:398-400 from_float(logits_vec, ...))
      reinterpret_cast<uint16_t*>(&p_s[wave][h][0])[lane] = T::from_float(p);
    }
    // p_s is private to this wave: LDS ops of one wave execute in order, only the compiler
    // has to be kept from reordering the stores above past the loads below.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ================= P.V : lanes re-mapped onto V's [row][token] layout =================
    const int win_tokens = min(w_tok1 - wstart, PA_WIN);
    const int nb = (win_tokens + BLOCK_SIZE - 1) / BLOCK_SIZE;
#pragma unroll
    for (int b = 0; b < G::WB; ++b) {
      if (b >= nb) break;  // wave-uniform
      // tokens of this lane's chunk: t0 .. t0+TPCV-1 (window-relative)
      const int t0 = b * BLOCK_SIZE + cpr_idx * G::TPCV;
      const bool partial = (b + 1) * BLOCK_SIZE > win_tokens;  // wave-uniform
      constexpr int NP = G::TPCV / 2;  // packed pairs per V chunk
      uint32_t pp[HG][NP];
#pragma unroll
      for (int h = 0; h < HG; ++h) {
#pragma unroll
        for (int j = 0; j < NP; j += 4) {
          const uint4 t = *reinterpret_cast<const uint4*>(&p_s[wave][h][t0 / 2 + j]);
          pp[h][j] = t.x; pp[h][j + 1] = t.y; pp[h][j + 2] = t.z; pp[h][j + 3] = t.w;
        }
      }
#pragma unroll
      for (int i0 = 0; i0 < G::NVL; i0 += VG) {
        uint32_t vraw[VG][4];
        if (b == 0 && i0 == 0) {   // static after unrolling
#pragma unroll
          for (int ii = 0; ii < VG; ++ii)
#pragma unroll
            for (int e = 0; e < 4; ++e) vraw[ii][e] = vpre[ii][e];
        } else {
          load_vgroup(b, i0, vraw);
        }
#pragma unroll
        for (int ii = 0; ii < VG; ++ii) {
          if (i0 + ii >= G::NVL) continue;
          uint32_t vpair[NP];
          if constexpr (!FP8) {
#pragma unroll
            for (int j = 0; j < NP; ++j) vpair[j] = vraw[ii][j];
          } else {
#pragma unroll
            for (int j = 0; j < NP / 2; ++j)
              fp8x4_to_pairs<T>(vraw[ii][j], vpair[2 * j], vpair[2 * j + 1]);
          }
          if (partial) {
            // zero V of out-of-context tokens: the cache may hold NaN garbage there
            // (attention_kernels.cu:424-434)
#pragma unroll
            for (int j = 0; j < NP; ++j) {
              const int t = t0 + 2 * j;
              vpair[j] = (t >= win_tokens) ? 0u
                                           : ((t + 1 >= win_tokens) ? (vpair[j] & 0xffffu) : vpair[j]);
            }
          }
#pragma unroll
          for (int h = 0; h < HG; ++h) {
            float a = acc[h][i0 + ii];
#pragma unroll
            for (int j = 0; j < NP; ++j) a = T::dot2(vpair[j], pp[h][j], a);
            acc[h][i0 + ii] = a;
          }
        }
      }
    }
    phys = phys_next;
  }

  // ================= merge =================
  // (1) lanes that share a V row
#pragma unroll
  for (int h = 0; h < HG; ++h)
#pragma unroll
    for (int i = 0; i < G::NVL; ++i) {
      float a = acc[h][i];
#pragma unroll
      for (int m = 1; m < G::CPR; m <<= 1) a += __shfl_xor(a, m, 64);
      acc[h][i] = a;
    }
  // (2) per-wave row sums
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    const float lw = wave_sum(l_lane[h]);
    if (lane == 0) {
      red_m[wave][h] = m_run[h];
      red_l[wave][h] = lw;
    }
  }
  __syncthreads();
  // (3) rescale to the workgroup max and park the accumulators in LDS
  const float kvs = FP8 ? kv_scale : 1.f;
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    float mg = red_m[0][h];
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) mg = fmaxf(mg, red_m[ww][h]);
    // wave without work: exp(-inf) = 0; mg = -inf (block-sparse: no token of this partition attended): 0
    const float f = mg == -INFINITY ? 0.f : __expf(m_run[h] - mg) * kvs;
#pragma unroll
    for (int i = 0; i < G::NVL; ++i) {
      const int row = i * G::RPL + row_l;
      if (cpr_idx == 0 && row < HEAD_SIZE) out_red[wave][h][row] = acc[h][i] * f;
    }
  }
  __syncthreads();
  // (4) sum the waves, normalise, store
  for (int idx = threadIdx.x; idx < HG * HEAD_SIZE; idx += (NW * WAVE)) {
    const int h = idx / HEAD_SIZE;
    const int d = idx % HEAD_SIZE;
    float mg = red_m[0][h];
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) mg = fmaxf(mg, red_m[ww][h]);
    float lg = 0.f, o = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) {
      lg += mg == -INFINITY ? 0.f : red_l[ww][h] * __expf(red_m[ww][h] - mg);
      o += out_red[ww][h][d];
    }
    const float inv = __fdividef(1.f, lg + 1e-6f);  // attention_kernels.cu:342
    const int head = head0 + h;
    if (partition_size) {
      const int64_t pidx = ((int64_t)seq_idx * num_heads + head) * max_num_partitions + part_idx;
      out[pidx * HEAD_SIZE + d] = T::from_float(o * inv);
      if (d == 0) {
        exp_sums[pidx] = lg;
        max_logits[pidx] = mg;
      }
    } else {
      out[((int64_t)seq_idx * num_heads + head) * HEAD_SIZE + d] = T::from_float(o * inv);
    }
  }
}

// v2 second pass: merge the partitions of one (seq, head).  Mirrors attention_kernels.cu:564-669.
// One wave per (seq, head): partitions are few (max_seq_len/512) and the work is tiny.
template <typename T, int HEAD_SIZE>
__global__ __launch_bounds__(WAVE) void paged_attention_v2_reduce_kernel(
    uint16_t* __restrict__ out, const float* __restrict__ exp_sums,
    const float* __restrict__ max_logits, const uint16_t* __restrict__ tmp_out,
    const int* __restrict__ seq_lens, int max_num_partitions) {
  const int num_heads = gridDim.x;
  const int head_idx = blockIdx.x;
  const int seq_idx = blockIdx.y;
  const int seq_len = seq_lens[seq_idx];
  const int num_partitions = (seq_len + PA_PARTITION - 1) / PA_PARTITION;
  const int lane = threadIdx.x;
  const int64_t base = ((int64_t)seq_idx * num_heads + head_idx) * max_num_partitions;
  uint16_t* out_ptr = out + ((int64_t)seq_idx * num_heads + head_idx) * HEAD_SIZE;
  const uint16_t* tmp_ptr = tmp_out + base * HEAD_SIZE;
  if (num_partitions <= 1) {
    if (num_partitions == 1)
      for (int i = lane; i < HEAD_SIZE; i += WAVE) out_ptr[i] = tmp_ptr[i];
    return;
  }
  extern __shared__ float w_s[];  // [num_partitions] rescaled exp sums
  float mx = -FLT_MAX;
  for (int i = lane; i < num_partitions; i += WAVE) mx = fmaxf(mx, max_logits[base + i]);
  mx = wave_max(mx);
  float gs = 0.f;
  for (int i = lane; i < num_partitions; i += WAVE) {
    const float r = exp_sums[base + i] * expf(max_logits[base + i] - mx);
    w_s[i] = r;
    gs += r;
  }
  gs = wave_sum(gs);
  __syncthreads();
  const float inv = __fdividef(1.f, gs + 1e-6f);
  for (int i = lane; i < HEAD_SIZE; i += WAVE) {
    float a = 0.f;
    for (int j = 0; j < num_partitions; ++j)
      a += T::to_float(tmp_ptr[(int64_t)j * HEAD_SIZE + i]) * w_s[j] * inv;
    out_ptr[i] = T::from_float(a);
  }
}

struct PAArgs {
  float* exp_sums; float* max_logits; void* out; void* tmp_out;
  const void* query; const void* key_cache; const void* value_cache;
  int num_seqs, num_heads, head_size, num_kv_heads; float scale;
  const int32_t* block_tables; const int32_t* seq_lens;
  int block_size, max_seq_len, max_num_blocks_per_seq;
  const float* alibi_slopes; int64_t q_stride, kv_block_stride, kv_head_stride;
  float kv_scale; bool partitioned; hipStream_t stream;
  PAFused fused = {nullptr, nullptr, 0, 0, 0, nullptr, nullptr, nullptr};
  PASparse sparse = {0, 0, 1, 64, 0};
};

template <typename T, bool FP8, int HEAD_SIZE, int BLOCK_SIZE, int HG>
static void launch_pa(const PAArgs& a) {
  const int parts = a.partitioned ? (a.max_seq_len + PA_PARTITION - 1) / PA_PARTITION : 1;
  dim3 grid(a.num_heads / HG, a.num_seqs, parts);
  // no more workgroups than CUs: 8 waves each
  bool wide = (int64_t)grid.x * grid.y * grid.z <= 256;
  if (const char* e = getenv("NMV_PA_NW")) wide = atoi(e) == 8;   // experiments
#define NMV_PA_LAUNCH(NW_) NMV_PA_LAUNCH_S(NW_, false)
#define NMV_PA_LAUNCH_S(NW_, SP_)                                                                 \
  hipLaunchKernelGGL((paged_attention_kernel<T, FP8, HEAD_SIZE, BLOCK_SIZE, HG, NW_, SP_>), grid,  \
                     dim3(NW_ * WAVE), 0, a.stream, a.exp_sums, a.max_logits,                      \
                     (uint16_t*)(a.partitioned ? a.tmp_out : a.out), (const uint16_t*)a.query,     \
                     (const uint8_t*)a.key_cache, (const uint8_t*)a.value_cache, a.num_heads,      \
                     a.num_kv_heads, a.scale, a.block_tables, a.seq_lens,                          \
                     a.max_num_blocks_per_seq, a.alibi_slopes, a.q_stride, a.kv_block_stride,      \
                     a.kv_head_stride, a.kv_scale, a.partitioned ? PA_PARTITION : 0, a.fused, a.sparse)
  if (a.sparse.vert_stride > 1) {   // block-sparse: one form (4 waves), no fused prologue
    NMV_PA_LAUNCH_S(4, true);
  } else if (wide) {
    NMV_PA_LAUNCH(8);
  } else {
    NMV_PA_LAUNCH(4);
  }
#undef NMV_PA_LAUNCH
#undef NMV_PA_LAUNCH_S
  if (a.partitioned) {
    dim3 rgrid(a.num_heads, a.num_seqs);
    hipLaunchKernelGGL((paged_attention_v2_reduce_kernel<T, HEAD_SIZE>), rgrid, dim3(WAVE),
                       parts * sizeof(float), a.stream, (uint16_t*)a.out, a.exp_sums,
                       a.max_logits, (const uint16_t*)a.tmp_out, a.seq_lens, parts);
  }
}

template <typename T, bool FP8, int HEAD_SIZE, int BLOCK_SIZE>
static int dispatch_hg(const PAArgs& a) {
  const int g = a.num_heads / a.num_kv_heads;
  if (g % 8 == 0) launch_pa<T, FP8, HEAD_SIZE, BLOCK_SIZE, 8>(a);
  else if (g % 4 == 0) launch_pa<T, FP8, HEAD_SIZE, BLOCK_SIZE, 4>(a);
  else if (g % 2 == 0) launch_pa<T, FP8, HEAD_SIZE, BLOCK_SIZE, 2>(a);
  else launch_pa<T, FP8, HEAD_SIZE, BLOCK_SIZE, 1>(a);
  return 0;
}

template <typename T, bool FP8, int HEAD_SIZE>
static int dispatch_bs(const PAArgs& a) {
  switch (a.block_size) {
    case 8: return dispatch_hg<T, FP8, HEAD_SIZE, 8>(a);
    case 16: return dispatch_hg<T, FP8, HEAD_SIZE, 16>(a);
    case 32: return dispatch_hg<T, FP8, HEAD_SIZE, 32>(a);
    default: return -1;
  }
}

template <int HEAD_SIZE>
int pa_dispatch_head(const PAArgs& a, nmv_dtype_t dtype, nmv_kv_dtype_t kv) {
  if (dtype == NMV_F16) return kv == NMV_KV_AUTO ? dispatch_bs<F16, false, HEAD_SIZE>(a)
                                                  : dispatch_bs<F16, true, HEAD_SIZE>(a);
  return kv == NMV_KV_AUTO ? dispatch_bs<BF16, false, HEAD_SIZE>(a)
                           : dispatch_bs<BF16, true, HEAD_SIZE>(a);
}

#ifdef NMV_PA_HEAD_SIZE
// one translation unit per head size keeps the build parallel
template int pa_dispatch_head<NMV_PA_HEAD_SIZE>(const PAArgs&, nmv_dtype_t, nmv_kv_dtype_t);
#endif

}  // namespace nmv

#ifndef NMV_PA_HEAD_SIZE
// ------------------------------- C ABI (compiled once) -------------------------------
using namespace nmv;
namespace nmv {
extern template int pa_dispatch_head<64>(const PAArgs&, nmv_dtype_t, nmv_kv_dtype_t);
extern template int pa_dispatch_head<80>(const PAArgs&, nmv_dtype_t, nmv_kv_dtype_t);
extern template int pa_dispatch_head<96>(const PAArgs&, nmv_dtype_t, nmv_kv_dtype_t);
extern template int pa_dispatch_head<112>(const PAArgs&, nmv_dtype_t, nmv_kv_dtype_t);
extern template int pa_dispatch_head<128>(const PAArgs&, nmv_dtype_t, nmv_kv_dtype_t);
extern template int pa_dispatch_head<192>(const PAArgs&, nmv_dtype_t, nmv_kv_dtype_t);
extern template int pa_dispatch_head<256>(const PAArgs&, nmv_dtype_t, nmv_kv_dtype_t);
}

static int pa_entry(PAArgs& a, nmv_dtype_t dtype, nmv_kv_dtype_t kv_dtype, const char* name) {
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16 || dtype == NMV_F32, "%s: unsupported data type %d", name, (int)dtype);
  NMV_CHECK(kv_dtype == NMV_KV_AUTO || kv_dtype == NMV_KV_FP8_E4M3,
            "%s: unsupported kv cache dtype %d", name, (int)kv_dtype);
  NMV_CHECK(dtype != NMV_F32 || (a.fused.slab == nullptr && a.fused.qkv == nullptr),
            "%s: the fused decode prologue exists for 16-bit models only", name);
  NMV_CHECK(a.num_kv_heads > 0 && a.num_heads % a.num_kv_heads == 0,
            "%s: num_heads %d not divisible by num_kv_heads %d", name, a.num_heads, a.num_kv_heads);
  NMV_CHECK(a.block_size == 8 || a.block_size == 16 || a.block_size == 32,
            "%s: Unsupported block size: %d", name, a.block_size);
  NMV_CHECK(dtype == NMV_F32 || a.q_stride % 2 == 0, "%s: query row stride must be even", name);
  NMV_CHECK((int64_t)a.max_num_blocks_per_seq * a.block_size >= a.max_seq_len,
            "%s: block_tables hold %d blocks of %d tokens per sequence, max_seq_len is %d", name,
            a.max_num_blocks_per_seq, a.block_size, a.max_seq_len);
  if (a.num_seqs == 0) return NMV_OK;
  if (dtype == NMV_F32) {   // float models: fp32_path.hip
    F32AttnArgs f{a.exp_sums, a.max_logits, (float*)a.out, (float*)a.tmp_out, (const float*)a.query, a.key_cache,
                  a.value_cache, a.num_seqs, a.num_heads, a.head_size, a.num_kv_heads, a.scale, a.block_tables,
                  a.seq_lens, a.block_size, a.max_seq_len, a.max_num_blocks_per_seq, a.alibi_slopes, a.q_stride,
                  a.kv_block_stride, a.kv_head_stride, a.kv_scale, a.partitioned, a.stream,
                  F32Sparse{a.sparse.tp_rank, a.sparse.local_blocks, a.sparse.vert_stride, a.sparse.block_size,
                            a.sparse.head_sliding_step}};
    const int frc = f32_paged_attention(f, kv_dtype == NMV_KV_FP8_E4M3);
    if (frc != NMV_OK) return frc;
    NMV_LAUNCH_CHECK();
    return NMV_OK;
  }
  int rc;
  switch (a.head_size) {
    case 64: rc = pa_dispatch_head<64>(a, dtype, kv_dtype); break;
    case 80: rc = pa_dispatch_head<80>(a, dtype, kv_dtype); break;
    case 96: rc = pa_dispatch_head<96>(a, dtype, kv_dtype); break;
    case 112: rc = pa_dispatch_head<112>(a, dtype, kv_dtype); break;
    case 128: rc = pa_dispatch_head<128>(a, dtype, kv_dtype); break;
    case 192: rc = pa_dispatch_head<192>(a, dtype, kv_dtype); break;
    case 256: rc = pa_dispatch_head<256>(a, dtype, kv_dtype); break;
    default:
      set_error("%s: Unsupported head size: %d", name, a.head_size);
      return NMV_ERR_INVALID;
  }
  if (rc != 0) {
    set_error("%s: unsupported configuration", name);
    return NMV_ERR_INVALID;
  }
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

extern "C" int nmv_paged_attention_v1(void* out, const void* query, const void* key_cache,
                                      const void* value_cache, int num_seqs, int num_heads,
                                      int head_size, int num_kv_heads, float scale,
                                      const int32_t* block_tables, const int32_t* seq_lens,
                                      int block_size, int max_seq_len, int max_num_blocks_per_seq,
                                      const float* alibi_slopes, int64_t q_stride,
                                      int64_t kv_block_stride, int64_t kv_head_stride,
                                      nmv_dtype_t dtype, nmv_kv_dtype_t kv_dtype, float kv_scale,
                                      int tp_rank, int blocksparse_local_blocks, int blocksparse_vert_stride,
                                      int blocksparse_block_size, int blocksparse_head_sliding_step,
                                      void* stream) {
  PAArgs a{nullptr, nullptr, out, nullptr, query, key_cache, value_cache, num_seqs, num_heads,
           head_size, num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
           max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride, kv_head_stride,
           kv_scale, false, (hipStream_t)stream};
  NMV_CHECK(blocksparse_vert_stride <= 1 || blocksparse_block_size > 0, "paged_attention_v1: blocksparse_block_size");
  a.sparse = PASparse{tp_rank, blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
                      blocksparse_head_sliding_step};
  return pa_entry(a, dtype, kv_dtype, "paged_attention_v1");
}

extern "C" int nmv_paged_attention_v2(void* out, float* exp_sums, float* max_logits,
                                      void* tmp_out, const void* query, const void* key_cache,
                                      const void* value_cache, int num_seqs, int num_heads,
                                      int head_size, int num_kv_heads, float scale,
                                      const int32_t* block_tables, const int32_t* seq_lens,
                                      int block_size, int max_seq_len, int max_num_blocks_per_seq,
                                      const float* alibi_slopes, int64_t q_stride,
                                      int64_t kv_block_stride, int64_t kv_head_stride,
                                      nmv_dtype_t dtype, nmv_kv_dtype_t kv_dtype, float kv_scale,
                                      int tp_rank, int blocksparse_local_blocks, int blocksparse_vert_stride,
                                      int blocksparse_block_size, int blocksparse_head_sliding_step,
                                      void* stream) {
  NMV_CHECK(exp_sums && max_logits && tmp_out, "paged_attention_v2: null partition buffers");
  NMV_CHECK(blocksparse_vert_stride <= 1 || blocksparse_block_size > 0, "paged_attention_v2: blocksparse_block_size");
  PAArgs a{exp_sums, max_logits, out, tmp_out, query, key_cache, value_cache, num_seqs,
           num_heads, head_size, num_kv_heads, scale, block_tables, seq_lens, block_size,
           max_seq_len, max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
           kv_head_stride, kv_scale, true, (hipStream_t)stream};
  a.sparse = PASparse{tp_rank, blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
                      blocksparse_head_sliding_step};
  return pa_entry(a, dtype, kv_dtype, "paged_attention_v2");
}

/* paged attention whose query -- and the new token's key / value -- are still the fp32 split-K slabs
 * of the qkv projection (nmv_gptq_marlin_gemm_partial), or with splits == 0 the finished qkv row
 * [num_seqs, (heads + 2 kv_heads) * head_size] in the model dtype passed as `slab`: sum + round, neox rotary embedding
 * (rot_dim == head_size, cos_sin_cache [max_pos, head_size]), k / v of the new token stored at
 * slot_mapping[seq] of the paged cache, then the v1 / v2 kernel -- rotary_embedding +
 * reshape_and_cache + paged_attention in one launch, bit-identical to them. */
static int pa_rope_partial(PAArgs& a, const float* slab, int splits, const int64_t* positions,
                           const void* cos_sin_cache, const int64_t* slot_mapping, nmv_dtype_t dtype,
                           nmv_kv_dtype_t kv_dtype, const char* name) {
  // splits == 0: `slab` is the finished qkv row in the model dtype, not fp32 slabs
  NMV_CHECK(slab != nullptr && splits >= 0 && positions && cos_sin_cache && slot_mapping,
            "%s: null fused-prologue argument", name);
  NMV_CHECK(a.alibi_slopes == nullptr, "%s: ALiBi models do not use rotary embedding", name);
  NMV_CHECK(((uintptr_t)slab & 15) == 0 && (a.head_size * (splits ? 4 : 2)) % 8 == 0, "%s: misaligned qkv", name);
  a.fused = PAFused{splits ? slab : nullptr, splits ? nullptr : (const uint16_t*)slab, splits,
                    (int64_t)a.num_seqs * (a.num_heads + 2 * a.num_kv_heads) * a.head_size,
                    (a.num_heads + 2 * a.num_kv_heads) * a.head_size, positions,
                    (const uint16_t*)cos_sin_cache, slot_mapping};
  return pa_entry(a, dtype, kv_dtype, name);
}

extern "C" int nmv_paged_attention_v1_rope_partial(
    void* out, const float* slab, int splits, const int64_t* positions, const void* cos_sin_cache,
    const int64_t* slot_mapping, void* key_cache, void* value_cache, int num_seqs, int num_heads,
    int head_size, int num_kv_heads, float scale, const int32_t* block_tables, const int32_t* seq_lens,
    int block_size, int max_seq_len, int max_num_blocks_per_seq, int64_t kv_block_stride,
    int64_t kv_head_stride, nmv_dtype_t dtype, nmv_kv_dtype_t kv_dtype, float kv_scale, void* stream) {
  PAArgs a{nullptr, nullptr, out, nullptr, nullptr, key_cache, value_cache, num_seqs, num_heads,
           head_size, num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
           max_num_blocks_per_seq, nullptr, 0, kv_block_stride, kv_head_stride, kv_scale, false,
           (hipStream_t)stream};
  return pa_rope_partial(a, slab, splits, positions, cos_sin_cache, slot_mapping, dtype, kv_dtype,
                         "paged_attention_v1_rope_partial");
}

extern "C" int nmv_paged_attention_v2_rope_partial(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const float* slab, int splits,
    const int64_t* positions, const void* cos_sin_cache, const int64_t* slot_mapping, void* key_cache,
    void* value_cache, int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int block_size, int max_seq_len,
    int max_num_blocks_per_seq, int64_t kv_block_stride, int64_t kv_head_stride, nmv_dtype_t dtype,
    nmv_kv_dtype_t kv_dtype, float kv_scale, void* stream) {
  NMV_CHECK(exp_sums && max_logits && tmp_out, "paged_attention_v2: null partition buffers");
  PAArgs a{exp_sums, max_logits, out, tmp_out, nullptr, key_cache, value_cache, num_seqs, num_heads,
           head_size, num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
           max_num_blocks_per_seq, nullptr, 0, kv_block_stride, kv_head_stride, kv_scale, true,
           (hipStream_t)stream};
  return pa_rope_partial(a, slab, splits, positions, cos_sin_cache, slot_mapping, dtype, kv_dtype,
                         "paged_attention_v2_rope_partial");
}
#endif
