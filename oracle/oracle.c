/*
 * oracle.c -- CPU restatement of the reference algorithms on the hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under neural_magic_vllm_amd/ may import, link or call this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker / reported CPU baseline.  Plain C + OpenMP, fp32 arithmetic, no GPU.
 *
 * Parity pinning: every function is checked in tests/ (-m "not gpu") against golden vectors that
 * tools/make_golden.py generated from the reference itself (its csrc/cpu kernels compiled into
 * oracle/_ref, and its Python quantization utilities imported from /root/reference).
 *
 * Each function cites the reference lines it restates (paths relative to /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_F16 0
#define ORC_BF16 1
#define ORC_KV_AUTO 0
#define ORC_KV_FP8 1

/* ------------------------------------------------------------------ scalar conversions */
static inline float bf16_to_f(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline uint16_t f_to_bf16(float f) { /* round to nearest even, NaN preserved */
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static inline float f16_to_f(uint16_t h) {
  uint32_t s = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff, u;
  if (e == 0) {
    if (m == 0) u = s;
    else { /* subnormal */
      int sh = 0;
      while (!(m & 0x400)) { m <<= 1; ++sh; }
      m &= 0x3ff;
      u = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
    }
  } else if (e == 31) u = s | 0x7f800000u | (m << 13);
  else u = s | ((e + 112) << 23) | (m << 13);
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline uint16_t f_to_f16(float f) { /* RNE */
  uint32_t u;
  memcpy(&u, &f, 4);
  uint32_t s = (u >> 16) & 0x8000;
  int32_t e = (int32_t)((u >> 23) & 0xff) - 127 + 15;
  uint32_t m = u & 0x7fffff;
  if (((u >> 23) & 0xff) == 0xff) return (uint16_t)(s | 0x7c00 | (m ? 0x200 : 0));
  if (e >= 31) return (uint16_t)(s | 0x7c00);
  if (e <= 0) {
    if (e < -10) return (uint16_t)s;
    m |= 0x800000;
    uint32_t shift = (uint32_t)(14 - e);
    uint32_t hm = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1))) ++hm;
    return (uint16_t)(s | hm);
  }
  uint32_t hm = m >> 13, rem = m & 0x1fff;
  uint16_t h = (uint16_t)(s | ((uint32_t)e << 10) | hm);
  if (rem > 0x1000 || (rem == 0x1000 && (hm & 1))) ++h;
  return h;
}
static inline float h_to_f(uint16_t v, int dt) { return dt == ORC_BF16 ? bf16_to_f(v) : f16_to_f(v); }
static inline uint16_t f_to_h(float f, int dt) { return dt == ORC_BF16 ? f_to_bf16(f) : f_to_f16(f); }
static inline float rnd_h(float f, int dt) { return h_to_f(f_to_h(f, dt), dt); }

/* OCP fp8 e4m3fn: 1-4-3, bias 7, no inf, NaN = S.1111.111, max 448 */
static inline float fp8_to_f(uint8_t b) {
  uint32_t s = b >> 7, e = (b >> 3) & 0xf, m = b & 7;
  float v;
  if (e == 0xf && m == 7) return NAN;
  if (e == 0) v = ldexpf((float)m, -9);
  else v = ldexpf((float)(8 + m), (int)e - 10);
  return s ? -v : v;
}
static inline uint8_t f_to_fp8(float f) { /* RNE, saturating to +-448 (torch .to(float8_e4m3fn) after clamp) */
  if (f != f) return 0x7f;
  uint8_t s = signbit(f) ? 0x80 : 0;
  float a = fabsf(f);
  if (a >= 448.f) return s | 0x7e;
  if (a < ldexpf(1.f, -10)) return s; /* below half the smallest subnormal */
  int e;
  float fr = frexpf(a, &e); /* a = fr * 2^e, fr in [0.5,1) */
  int ee = e - 1;           /* a = (2 fr) * 2^ee, 2fr in [1,2) */
  if (ee < -6) {            /* subnormal: units of 2^-9 */
    float q = a * 512.f;
    float rq = nearbyintf(q);
    int m = (int)rq;
    if (m >= 8) return s | 0x08;
    return s | (uint8_t)m;
  }
  float q = (2.f * fr - 1.f) * 8.f; /* mantissa in [0,8) */
  float rq = nearbyintf(q);
  int m = (int)rq;
  if (m == 8) { m = 0; ++ee; }
  if (ee > 8 || (ee == 8 && m == 7)) return s | 0x7e;
  return s | (uint8_t)(((ee + 7) << 3) | m);
}

/* exported for tests */
void orc_fp8_decode(const uint8_t* in, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = fp8_to_f(in[i]);
}
void orc_fp8_encode(const float* in, uint8_t* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = f_to_fp8(in[i]);
}
void orc_half_decode(const uint16_t* in, float* out, int64_t n, int dt) {
  for (int64_t i = 0; i < n; ++i) out[i] = h_to_f(in[i], dt);
}
void orc_half_encode(const float* in, uint16_t* out, int64_t n, int dt) {
  for (int64_t i = 0; i < n; ++i) out[i] = f_to_h(in[i], dt);
}

/* ------------------------------------------------------------------ KV cache
 * reshape_and_cache: csrc/cache_kernels.cu:152-204 (index arithmetic :176-191);
 * CPU twin csrc/cpu/cache.cpp:35-84.  Caches are raw bytes; elem = 2 B (auto) or 1 B (fp8). */
void orc_reshape_and_cache(const uint16_t* key, const uint16_t* value, void* key_cache,
                           void* value_cache, const int64_t* slot_mapping, int num_tokens,
                           int num_heads, int head_size, int block_size, int64_t key_stride,
                           int64_t value_stride, int dt, int kv_dt, float kv_scale) {
  const int x = kv_dt == ORC_KV_AUTO ? 8 : 16;
  for (int t = 0; t < num_tokens; ++t) {
    const int64_t slot = slot_mapping[t];
    if (slot < 0) continue;
    const int64_t blk = slot / block_size, off = slot % block_size;
    for (int i = 0; i < num_heads * head_size; ++i) {
      const int h = i / head_size, d = i % head_size;
      const int64_t kidx = blk * num_heads * (head_size / x) * block_size * x +
                           (int64_t)h * (head_size / x) * block_size * x +
                           (int64_t)(d / x) * block_size * x + off * x + d % x;
      const int64_t vidx = blk * num_heads * head_size * block_size +
                           (int64_t)h * head_size * block_size + (int64_t)d * block_size + off;
      const uint16_t kk = key[t * key_stride + i], vv = value[t * value_stride + i];
      if (kv_dt == ORC_KV_AUTO) {
        ((uint16_t*)key_cache)[kidx] = kk;
        ((uint16_t*)value_cache)[vidx] = vv;
      } else {
        ((uint8_t*)key_cache)[kidx] = f_to_fp8(h_to_f(kk, dt) / kv_scale);
        ((uint8_t*)value_cache)[vidx] = f_to_fp8(h_to_f(vv, dt) / kv_scale);
      }
    }
  }
}

/* copy_blocks: csrc/cache_kernels.cu:68-94 (one layer) */
void orc_copy_blocks(uint8_t* key_cache, uint8_t* value_cache, const int64_t* block_mapping,
                     int num_pairs, int64_t bytes_per_block) {
  for (int p = 0; p < num_pairs; ++p) {
    const int64_t s = block_mapping[2 * p] * bytes_per_block, d = block_mapping[2 * p + 1] * bytes_per_block;
    memmove(key_cache + d, key_cache + s, bytes_per_block);
    memmove(value_cache + d, value_cache + s, bytes_per_block);
  }
}

/* ------------------------------------------------------------------ paged attention
 * Restates paged_attention_kernel, csrc/attention/attention_kernels.cu:86-496 (CPU twin
 * csrc/cpu/attention.cpp:221-341): logits = scale*q.k (+alibi), softmax with 1/(sum+1e-6)
 * (:342), out = P.V in fp32.  partition_size == 0: v1.  partition_size > 0: v2 first pass --
 * writes exp_sums/max_logits/tmp_out per partition (:350-361, :480-495). */
static inline float cache_at(const void* cache, int64_t idx, int dt, int kv_dt, float kv_scale) {
  if (kv_dt == ORC_KV_AUTO) return h_to_f(((const uint16_t*)cache)[idx], dt);
  return fp8_to_f(((const uint8_t*)cache)[idx]) * kv_scale;
}

void orc_paged_attention(uint16_t* out, float* exp_sums, float* max_logits, uint16_t* tmp_out,
                         const uint16_t* q, const void* k_cache, const void* v_cache, int num_seqs,
                         int num_heads, int head_size, int num_kv_heads, float scale,
                         const int32_t* block_tables, const int32_t* seq_lens, int block_size,
                         int max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
                         int64_t kv_block_stride, int64_t kv_head_stride, int dt, int kv_dt,
                         float kv_scale, int partition_size, int max_num_partitions) {
  const int x = kv_dt == ORC_KV_AUTO ? 8 : 16;
  const int qpk = num_heads / num_kv_heads;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int s = 0; s < num_seqs; ++s) {
    for (int h = 0; h < num_heads; ++h) {
      const int seq_len = seq_lens[s];
      const int kvh = h / qpk;
      const int nparts = partition_size ? (seq_len + partition_size - 1) / partition_size : 1;
      const int32_t* bt = block_tables + (int64_t)s * max_num_blocks_per_seq;
      float* qf = (float*)malloc(sizeof(float) * head_size);
      for (int d = 0; d < head_size; ++d) qf[d] = h_to_f(q[s * q_stride + (int64_t)h * head_size + d], dt);
      const float slope = alibi_slopes ? alibi_slopes[h] : 0.f;
      for (int part = 0; part < nparts; ++part) {
        const int t0 = partition_size ? part * partition_size : 0;
        const int t1 = partition_size ? (t0 + partition_size < seq_len ? t0 + partition_size : seq_len) : seq_len;
        const int nt = t1 - t0;
        if (nt <= 0) continue;
        float* logits = (float*)malloc(sizeof(float) * nt);
        float mx = -INFINITY;
        for (int t = t0; t < t1; ++t) {
          const int64_t pb = bt[t / block_size];
          const int off = t % block_size;
          float acc = 0.f;
          for (int d = 0; d < head_size; ++d) {
            const int64_t idx = pb * kv_block_stride + (int64_t)kvh * kv_head_stride +
                                (int64_t)(d / x) * block_size * x + (int64_t)off * x + d % x;
            acc += qf[d] * cache_at(k_cache, idx, dt, kv_dt, kv_scale);
          }
          float l = scale * acc;
          if (slope != 0.f) l += slope * (float)(t - seq_len + 1);
          logits[t - t0] = l;
          if (l > mx) mx = l;
        }
        float sum = 0.f;
        for (int i = 0; i < nt; ++i) { logits[i] = expf(logits[i] - mx); sum += logits[i]; }
        const float inv = 1.f / (sum + 1e-6f);
        uint16_t* o = partition_size
                          ? tmp_out + (((int64_t)s * num_heads + h) * max_num_partitions + part) * head_size
                          : out + ((int64_t)s * num_heads + h) * head_size;
        for (int d = 0; d < head_size; ++d) {
          float acc = 0.f;
          for (int t = t0; t < t1; ++t) {
            const int64_t pb = bt[t / block_size];
            const int64_t idx = pb * kv_block_stride + (int64_t)kvh * kv_head_stride +
                                (int64_t)d * block_size + t % block_size;
            acc += logits[t - t0] * inv * cache_at(v_cache, idx, dt, kv_dt, kv_scale);
          }
          o[d] = f_to_h(acc, dt);
        }
        if (partition_size) {
          const int64_t pi = ((int64_t)s * num_heads + h) * max_num_partitions + part;
          exp_sums[pi] = sum;
          max_logits[pi] = mx;
        }
        free(logits);
      }
      /* v2 reduce: csrc/attention/attention_kernels.cu:564-669 */
      if (partition_size && nparts >= 1) {
        const int64_t base = ((int64_t)s * num_heads + h) * max_num_partitions;
        uint16_t* o = out + ((int64_t)s * num_heads + h) * head_size;
        if (nparts == 1) {
          memcpy(o, tmp_out + base * head_size, sizeof(uint16_t) * head_size);
        } else {
          float mx = -INFINITY, gs = 0.f;
          for (int p = 0; p < nparts; ++p) if (max_logits[base + p] > mx) mx = max_logits[base + p];
          float* w = (float*)malloc(sizeof(float) * nparts);
          for (int p = 0; p < nparts; ++p) { w[p] = exp_sums[base + p] * expf(max_logits[base + p] - mx); gs += w[p]; }
          const float inv = 1.f / (gs + 1e-6f);
          for (int d = 0; d < head_size; ++d) {
            float acc = 0.f;
            for (int p = 0; p < nparts; ++p) acc += h_to_f(tmp_out[(base + p) * head_size + d], dt) * w[p] * inv;
            o[d] = f_to_h(acc, dt);
          }
          free(w);
        }
      }
      free(qf);
    }
  }
}

/* ------------------------------------------------------------------ Marlin layout
 * gptq_marlin_repack: csrc/quantization/gptq_marlin/gptq_marlin_repack.cu:44-265, i.e. the
 * composition of gptq unpack (quant_utils.py:125-146), marlin_permute_weights
 * (marlin_utils.py:25-37) with get_perms (marlin_perms.py:16-43) and the packing loop
 * (marlin_utils.py:40-57).  Written here directly as the permutation, not via index lists. */
static inline void marlin_src(int bits, int col_in_row, int size_n, int pz, int* n, int* k_in) {
  const int pack = 32 / bits, wpc = 1024 / pack;
  const int chunk = col_in_row / wpc, rr = col_in_row % wpc;
  int i, j, blk;
  if (bits == 4) { i = rr >> 2; j = rr & 3; blk = (pz >> 1) & 1; *k_in = 2 * (i & 3) + ((pz & 1) ? 8 : 0) + (pz >> 2); }
  else { i = rr >> 3; j = (rr >> 1) & 3; blk = rr & 1; *k_in = 2 * (i & 3) + ((pz & 1) ? 8 : 0) + (pz >> 1); }
  *n = chunk * 64 + j * 16 + blk * 8 + (i >> 2);
  (void)size_n;
}

void orc_gptq_marlin_repack(const uint32_t* qw, const int32_t* perm, uint32_t* out, int size_k,
                            int size_n, int bits) {
  const int pack = 32 / bits;
  const int64_t row_words = (int64_t)size_n * 16 / pack;
#pragma omp parallel for
  for (int kt = 0; kt < size_k / 16; ++kt)
    for (int64_t col = 0; col < row_words; ++col) {
      uint32_t res = 0;
      for (int pz = 0; pz < pack; ++pz) {
        int n, k_in;
        marlin_src(bits, (int)col, size_n, pz, &n, &k_in);
        const int k = kt * 16 + k_in;
        const int ks = perm ? perm[k] : k;
        const uint32_t w = qw[(int64_t)(ks / pack) * size_n + n];
        res |= ((w >> (bits * (ks % pack))) & ((1u << bits) - 1)) << (bits * pz);
      }
      out[kt * row_words + col] = res;
    }
}

/* inverse: Marlin tensor -> integer codes q[k][n] (uint8) */
void orc_marlin_unpack(const uint32_t* mw, uint8_t* q, int size_k, int size_n, int bits) {
  const int pack = 32 / bits;
  const int64_t row_words = (int64_t)size_n * 16 / pack;
#pragma omp parallel for
  for (int kt = 0; kt < size_k / 16; ++kt)
    for (int64_t col = 0; col < row_words; ++col) {
      const uint32_t w = mw[kt * row_words + col];
      for (int pz = 0; pz < pack; ++pz) {
        int n, k_in;
        marlin_src(bits, (int)col, size_n, pz, &n, &k_in);
        q[(int64_t)(kt * 16 + k_in) * size_n + n] = (uint8_t)((w >> (bits * pz)) & ((1u << bits) - 1));
      }
    }
}

/* undo marlin_permute_scales (gptq_marlin.py:47-56): returns s[g][n] in natural column order */
static void unpermute_scales(const uint16_t* sp, float* s, int num_groups, int size_n, int grouped, int dt) {
  for (int g = 0; g < num_groups; ++g)
    for (int n = 0; n < size_n; ++n) {
      int pos;
      if (grouped) { const int c = n % 64; pos = (n / 64) * 64 + (c % 8) * 8 + c / 8; }
      else { const int c = n % 32; pos = (n / 32) * 32 + ((c % 8) / 2) * 8 + 2 * (c / 8) + c % 2; }
      s[(int64_t)g * size_n + n] = h_to_f(sp[(int64_t)g * size_n + pos], dt);
    }
}

/* gptq_marlin_gemm: csrc/quantization/gptq_marlin/gptq_marlin.cu:1735-1868.
 * Semantics per tests/kernels/test_marlin_gemm.py:126-179 and quant_utils.py:39-106:
 * w[k][n] = half((q - 2^(bits-1)) * s[group(k)][n]);  c = half( sum_k a'[m][k] * w[k][n] ) with
 * a' = a[:, perm] under act-order and group(k) = g_idx[k] (sorted) or k / group_size. */
void orc_gptq_marlin_gemm(uint16_t* c, const uint16_t* a, const uint32_t* b_q_weight,
                          const uint16_t* b_scales, const int32_t* g_idx, const int32_t* perm,
                          int bits, int size_m, int size_n, int size_k, int num_groups, int dt) {
  uint8_t* q = (uint8_t*)malloc((size_t)size_k * size_n);
  float* s = (float*)malloc(sizeof(float) * (size_t)num_groups * size_n);
  float* w = (float*)malloc(sizeof(float) * (size_t)size_k * size_n);
  orc_marlin_unpack(b_q_weight, q, size_k, size_n, bits);
  unpermute_scales(b_scales, s, num_groups, size_n, num_groups > 1, dt);
  const int zp = 1 << (bits - 1);
  const int gsz = size_k / num_groups;
#pragma omp parallel for
  for (int k = 0; k < size_k; ++k) {
    const int g = g_idx ? g_idx[k] : k / gsz;
    for (int n = 0; n < size_n; ++n)
      w[(int64_t)k * size_n + n] = rnd_h((float)((int)q[(int64_t)k * size_n + n] - zp) * s[(int64_t)g * size_n + n], dt);
  }
#pragma omp parallel for
  for (int m = 0; m < size_m; ++m) {
    float* acc = (float*)calloc(size_n, sizeof(float));
    for (int k = 0; k < size_k; ++k) {
      const float av = h_to_f(a[(int64_t)m * size_k + (perm ? perm[k] : k)], dt);
      const float* wr = w + (int64_t)k * size_n;
      for (int n = 0; n < size_n; ++n) acc[n] += av * wr[n];
    }
    for (int n = 0; n < size_n; ++n) c[(int64_t)m * size_n + n] = f_to_h(acc[n], dt);
    free(acc);
  }
  free(q); free(s); free(w);
}

/* ------------------------------------------------------------------ glue ops
 * rms_norm / fused_add_rms_norm: csrc/layernorm_kernels.cu:22-44, 201-290 */
void orc_rms_norm(uint16_t* out, uint16_t* input, uint16_t* residual, const uint16_t* weight,
                  float eps, int num_tokens, int hidden, int dt) {
  for (int t = 0; t < num_tokens; ++t) {
    float var = 0.f;
    const int64_t row = (int64_t)t * hidden;
    float* z = (float*)malloc(sizeof(float) * hidden);
    for (int i = 0; i < hidden; ++i) {
      float x = h_to_f(input[row + i], dt);
      if (residual) { x = rnd_h(x + h_to_f(residual[row + i], dt), dt); residual[row + i] = f_to_h(x, dt); }
      z[i] = x;
      var += x * x;
    }
    const float sc = 1.0f / sqrtf(var / hidden + eps);
    for (int i = 0; i < hidden; ++i)
      out[row + i] = f_to_h(rnd_h(z[i] * sc, dt) * h_to_f(weight[i], dt), dt);
    free(z);
  }
}

/* rotary_embedding: csrc/pos_encoding_kernels.cu:10-93 */
void orc_rotary_embedding(const int64_t* positions, uint16_t* query, uint16_t* key, int num_tokens,
                          int num_heads, int num_kv_heads, int head_size, int rot_dim,
                          int64_t query_stride, int64_t key_stride, const uint16_t* cos_sin_cache,
                          int is_neox, const int64_t* offsets, int dt) {
  const int embed = rot_dim / 2;
  for (int t = 0; t < num_tokens; ++t) {
    const int64_t pos = positions[t] + (offsets ? offsets[t] : 0);
    const uint16_t* cp = cos_sin_cache + pos * rot_dim;
    const uint16_t* sp = cp + embed;
    for (int pass = 0; pass < 2; ++pass) {
      uint16_t* base = pass == 0 ? query + t * query_stride : key + t * key_stride;
      const int nh = pass == 0 ? num_heads : num_kv_heads;
      for (int i = 0; i < nh * embed; ++i) {
        uint16_t* arr = base + (int64_t)(i / embed) * head_size;
        const int ro = i % embed;
        const int xi = is_neox ? ro : 2 * ro, yi = is_neox ? embed + ro : 2 * ro + 1;
        const float c = h_to_f(cp[is_neox ? xi : xi / 2], dt), s = h_to_f(sp[is_neox ? xi : xi / 2], dt);
        const float x = h_to_f(arr[xi], dt), y = h_to_f(arr[yi], dt);
        arr[xi] = f_to_h(rnd_h(x * c, dt) - rnd_h(y * s, dt), dt);
        arr[yi] = f_to_h(rnd_h(y * c, dt) + rnd_h(x * s, dt), dt);
      }
    }
  }
}

/* silu_and_mul / gelu_and_mul / gelu_tanh_and_mul: csrc/activation_kernels.cu:14-61 */
void orc_act_and_mul(uint16_t* out, const uint16_t* input, int num_tokens, int d, int act, int dt) {
  for (int t = 0; t < num_tokens; ++t)
    for (int i = 0; i < d; ++i) {
      const float f = h_to_f(input[(int64_t)t * 2 * d + i], dt), y = h_to_f(input[(int64_t)t * 2 * d + d + i], dt);
      float a;
      if (act == 0) a = f / (1.0f + expf(-f));
      else if (act == 1) a = f * 0.5f * (1.0f + erff(f * 0.70710678118654752440f));
      else a = 0.5f * f * (1.0f + tanhf(0.79788456080286535588f * (f + 0.044715f * f * f * f)));
      out[(int64_t)t * d + i] = f_to_h(rnd_h(a, dt) * y, dt);
    }
}

/* ------------------------------------------------------------------ W8A8
 * scaled_int8_quant: csrc/quantization/compressed_tensors/int8_quant_kernels.cu:6-75 and the
 * reference test's checker tests/kernels/test_int8_quant.py:33-38,64-65. */
static inline int8_t f_to_i8_rn(float x) {
  float d = nearbyintf(x);
  if (d < -128.f) d = -128.f;
  if (d > 127.f) d = 127.f;
  return (int8_t)d;
}
void orc_scaled_int8_quant(int8_t* out, const uint16_t* input, float* scale, int num_tokens,
                           int hidden, int dynamic, int dt) {
  for (int t = 0; t < num_tokens; ++t) {
    const int64_t row = (int64_t)t * hidden;
    if (dynamic) {
      float amax = 0.f;
      for (int i = 0; i < hidden; ++i) { const float v = fabsf(h_to_f(input[row + i], dt)); if (v > amax) amax = v; }
      scale[t] = amax / 127.0f;
      const float mul = 127.0f / amax;
      for (int i = 0; i < hidden; ++i) out[row + i] = f_to_i8_rn(h_to_f(input[row + i], dt) * mul);
    } else {
      for (int i = 0; i < hidden; ++i) out[row + i] = f_to_i8_rn(h_to_f(input[row + i], dt) / scale[0]);
    }
  }
}

/* scaled_fp8_quant: csrc/quantization/fp8/common.cu:22-127 (x * (1/scale), clamp +-448, e4m3fn;
 * dynamic scale = absmax / 448) */
void orc_scaled_fp8_quant(uint8_t* out, const uint16_t* input, float* scale, int64_t n, int dynamic, int dt) {
  if (dynamic) {
    float amax = 0.f;
    for (int64_t i = 0; i < n; ++i) { const float v = fabsf(h_to_f(input[i], dt)); if (v > amax) amax = v; }
    scale[0] = amax / 448.0f;
  }
  const float inv = 1.0f / scale[0];
  for (int64_t i = 0; i < n; ++i) {
    float x = h_to_f(input[i], dt) * inv;
    x = fmaxf(-448.f, fminf(x, 448.f));
    out[i] = f_to_fp8(x);
  }
}

/* cutlass_scaled_mm: csrc/quantization/cutlass_w8a8/scaled_mm_entry.cu:48-100, epilogue
 * scaled_mm_c2x.cu:88-140; checker tests/kernels/test_cutlass.py:36-48.
 * a [M,K] row-major, bt [N,K] row-major (= column-major b).  is_fp8: bytes are e4m3fn. */
void orc_scaled_mm(uint16_t* out, const uint8_t* a, const uint8_t* bt, const float* a_scales,
                   const float* b_scales, const uint16_t* bias, int M, int N, int K, int a_per_row,
                   int b_per_col, int is_fp8, int dt) {
  float* af = NULL; float* bf = NULL;
  if (is_fp8) {
    af = (float*)malloc(sizeof(float) * (size_t)M * K);
    bf = (float*)malloc(sizeof(float) * (size_t)N * K);
    for (int64_t i = 0; i < (int64_t)M * K; ++i) af[i] = fp8_to_f(a[i]);
    for (int64_t i = 0; i < (int64_t)N * K; ++i) bf[i] = fp8_to_f(bt[i]);
  }
#pragma omp parallel for collapse(2)
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      float accf;
      if (is_fp8) {
        double acc = 0.0;
        for (int k = 0; k < K; ++k) acc += (double)af[(int64_t)m * K + k] * (double)bf[(int64_t)n * K + k];
        accf = (float)acc;
      } else {
        int64_t acc = 0;
        for (int k = 0; k < K; ++k) acc += (int)(int8_t)a[(int64_t)m * K + k] * (int)(int8_t)bt[(int64_t)n * K + k];
        accf = (float)acc;
      }
      /* epilogue in fp32, one rounding to the output type: without bias multiplies(a_scales,
       * multiplies(b_scales, acc)) (scaled_mm_c2x.cu:117-131); with bias the outer node is
       * cutlass::multiply_add(a_scales, tmp, bias) (:157-171), a fused multiply-add in device code */
      const float tmp = b_scales[b_per_col ? n : 0] * accf;
      const float as = a_scales[a_per_row ? m : 0];
      const float o = bias ? fmaf(as, tmp, h_to_f(bias[n], dt)) : as * tmp;
      out[(int64_t)m * N + n] = f_to_h(o, dt);
    }
  free(af); free(bf);
}
