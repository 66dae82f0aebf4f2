// Greedy sampling of a decode step and the on-device advance of the batch state, in two launches.
// Behavioural reference: the greedy branch of the reference sampler is torch.argmax over the
// logits (vllm/model_executor/layers/sampler.py:_greedy_sample); position / slot bookkeeping is
// host code in the reference (worker/model_runner.py:572-580: slot = block_table[pos // bs] * bs +
// pos % bs).  Here both run on the device so that a captured decode step can be replayed back to
// back: torch.argmax on [64, 128256] bf16 costs ~48 us and the five tiny torch kernels of the
// state update ~35 us per step; these two launches cost ~8 us.  HBM-bound: one pass over the
// logits with 16-byte loads, (value, index) pairs reduced per wave, per workgroup, per row.
// Ties resolve to the lowest index (what torch.argmax returns on this platform); NaN is not
// treated specially (the logits of a healthy model have none).
#include "common.h"

namespace nmv {

constexpr int AM_THREADS = 256;
constexpr int AM_SPLITS = 16;  // workgroups per row

__device__ __forceinline__ void am_better(float& v, int& i, float ov, int oi) {
  if (ov > v || (ov == v && oi < i)) {
    v = ov;
    i = oi;
  }
}

template <typename T>
__global__ __launch_bounds__(AM_THREADS) void argmax_partial_kernel(const uint16_t* __restrict__ logits,
                                                                    int64_t row_stride, int V,
                                                                    float* __restrict__ pval,
                                                                    int* __restrict__ pidx) {
  __shared__ float sv[AM_THREADS / 64];
  __shared__ int si[AM_THREADS / 64];
  const int row = blockIdx.y, split = blockIdx.x;
  const uint16_t* src = logits + (int64_t)row * row_stride;
  // slice boundaries in multiples of 8 elements (one 16-byte vector)
  const int nvec = (V + 7) / 8;
  const int per = (nvec + AM_SPLITS - 1) / AM_SPLITS;
  const int v0 = split * per, v1 = min(v0 + per, nvec);
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
  float best = -INFINITY;
  int best_i = 0x7fffffff;
  for (int v = v0 + threadIdx.x; v < v1; v += AM_THREADS) {
    const int e0 = v * 8;
    if (aligned && e0 + 8 <= V) {
      const uint4 x = ld16(src + e0);
      const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = lo_f<T>(xs[j]), hi = hi_f<T>(xs[j]);
        if (lo > best) best = lo, best_i = e0 + 2 * j;       // increasing index: strict > keeps
        if (hi > best) best = hi, best_i = e0 + 2 * j + 1;   // the first occurrence
      }
    } else {
      for (int e = e0; e < min(e0 + 8, V); ++e) {
        const float f = T::to_float(src[e]);
        if (f > best) best = f, best_i = e;
      }
    }
  }
  // a lane's vectors are not contiguous, so ties across lanes are settled on the index
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off);
    const int oi = __shfl_xor(best_i, off);
    am_better(best, best_i, ov, oi);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sv[wave] = best, si[wave] = best_i;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < AM_THREADS / 64; ++w) am_better(best, best_i, sv[w], si[w]);
    pval[row * AM_SPLITS + split] = best;
    pidx[row * AM_SPLITS + split] = best_i;
  }
}

// one wave per row: merge up to 64 candidates (value pval[row * row_stride + c * cand_stride], index
// likewise from pidx), then either publish the winner as a (value, index + idx_offset) record
// (pair_val / pair_idx: the per-shard result of a vocab-parallel lm_head) or publish the token and
// (optionally) advance the sequence: input_ids = token, positions += 1, seq_lens += 1,
// slot = block_table[pos / bs] * bs + pos % bs
__global__ __launch_bounds__(64) void argmax_final_kernel(const float* __restrict__ pval,
                                                          const int* __restrict__ pidx, int n_cand,
                                                          int64_t cand_stride, int64_t row_stride,
                                                          float* __restrict__ pair_val,
                                                          int* __restrict__ pair_idx, int idx_offset,
                                                          int64_t* __restrict__ next_tokens,
                                                          int64_t* input_ids, int64_t* positions,
                                                          int* seq_lens, int64_t* slot_mapping,
                                                          const int* __restrict__ block_tables,
                                                          int max_blocks_per_seq, int block_size) {
  const int row = blockIdx.x, lane = threadIdx.x;
  float best = lane < n_cand ? pval[row * row_stride + lane * cand_stride] : -INFINITY;
  int best_i = lane < n_cand ? pidx[row * row_stride + lane * cand_stride] : 0x7fffffff;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off);
    const int oi = __shfl_xor(best_i, off);
    am_better(best, best_i, ov, oi);
  }
  if (lane != 0) return;
  if (pair_val != nullptr) {
    pair_val[row] = best;
    pair_idx[row] = best_i + idx_offset;
    return;
  }
  next_tokens[row] = best_i;
  if (positions == nullptr) return;
  input_ids[row] = best_i;
  const int64_t pos = positions[row] + 1;
  positions[row] = pos;
  seq_lens[row] += 1;
  // a sequence that has just filled its last block has no next slot: the block table ends there (its
  // row would be read one entry past the end); -1 = "skip" for reshape_and_cache (cache_kernels.cu:164)
  const int64_t bi = pos / block_size;
  if (bi >= max_blocks_per_seq) {
    slot_mapping[row] = -1;
    return;
  }
  const int64_t blk = block_tables[(int64_t)row * max_blocks_per_seq + bi];
  slot_mapping[row] = blk * block_size + pos % block_size;
}

}  // namespace nmv

using namespace nmv;

extern "C" int64_t nmv_greedy_sample_scratch_bytes(int num_seqs) {
  return (int64_t)std::max(num_seqs, 0) * AM_SPLITS * 8;
}

extern "C" int nmv_greedy_sample_advance(int64_t* next_tokens, const void* logits, int64_t row_stride,
                                         int num_seqs, int vocab_size, nmv_dtype_t dtype,
                                         void* scratch, int64_t scratch_bytes, int64_t* input_ids,
                                         int64_t* positions, int* seq_lens, int64_t* slot_mapping,
                                         const int* block_tables, int max_blocks_per_seq,
                                         int block_size, void* stream) {
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16, "greedy_sample: unsupported dtype %d", (int)dtype);
  NMV_CHECK(vocab_size > 0 && row_stride >= vocab_size, "greedy_sample: bad vocab_size / row stride");
  NMV_CHECK(scratch != nullptr && scratch_bytes >= nmv_greedy_sample_scratch_bytes(num_seqs),
            "greedy_sample: scratch too small");
  NMV_CHECK(positions == nullptr || (input_ids && seq_lens && slot_mapping && block_tables &&
                                     max_blocks_per_seq > 0 && block_size > 0),
            "greedy_sample: the state advance needs every state tensor");
  if (num_seqs == 0) return NMV_OK;
  float* pval = reinterpret_cast<float*>(scratch);
  int* pidx = reinterpret_cast<int*>(pval + (int64_t)num_seqs * AM_SPLITS);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(AM_SPLITS, num_seqs);
  if (dtype == NMV_F16)
    hipLaunchKernelGGL((argmax_partial_kernel<F16>), grid, dim3(AM_THREADS), 0, s,
                       (const uint16_t*)logits, row_stride, vocab_size, pval, pidx);
  else
    hipLaunchKernelGGL((argmax_partial_kernel<BF16>), grid, dim3(AM_THREADS), 0, s,
                       (const uint16_t*)logits, row_stride, vocab_size, pval, pidx);
  hipLaunchKernelGGL(argmax_final_kernel, dim3(num_seqs), dim3(64), 0, s, pval, pidx, AM_SPLITS,
                     (int64_t)1, (int64_t)AM_SPLITS, (float*)nullptr, (int*)nullptr, 0, next_tokens,
                     input_ids, positions, seq_lens, slot_mapping, block_tables, max_blocks_per_seq,
                     block_size);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

/* Vocab-parallel greedy sampling, shard side: the argmax of this rank's logits [num_seqs, local
 * vocab] as a record = float value[padded] followed by int32 global_index[padded] (padded =
 * nmv_greedy_record_elems(num_seqs), a multiple of 4 so that the record is a multiple of 16 bytes);
 * entries past num_seqs are left untouched.  The records of all ranks are then all-gathered
 * (nmv_ar_all_gather) and nmv_greedy_sample_finish picks the winner. */
extern "C" int nmv_greedy_record_elems(int num_seqs) { return (std::max(num_seqs, 1) + 3) / 4 * 4; }

extern "C" int nmv_greedy_sample_shard(void* record, const void* logits, int64_t row_stride, int num_seqs,
                                       int local_vocab, int index_offset, nmv_dtype_t dtype,
                                       void* scratch, int64_t scratch_bytes, void* stream) {
  NMV_CHECK(dtype == NMV_F16 || dtype == NMV_BF16, "greedy_sample: unsupported dtype %d", (int)dtype);
  NMV_CHECK(local_vocab > 0 && row_stride >= local_vocab, "greedy_sample: bad vocab_size / row stride");
  NMV_CHECK(scratch != nullptr && scratch_bytes >= nmv_greedy_sample_scratch_bytes(num_seqs),
            "greedy_sample: scratch too small");
  if (num_seqs == 0) return NMV_OK;
  float* pval = reinterpret_cast<float*>(scratch);
  int* pidx = reinterpret_cast<int*>(pval + (int64_t)num_seqs * AM_SPLITS);
  const int padded = nmv_greedy_record_elems(num_seqs);
  float* rec_val = reinterpret_cast<float*>(record);
  int* rec_idx = reinterpret_cast<int*>(rec_val + padded);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(AM_SPLITS, num_seqs);
  if (dtype == NMV_F16)
    hipLaunchKernelGGL((argmax_partial_kernel<F16>), grid, dim3(AM_THREADS), 0, s,
                       (const uint16_t*)logits, row_stride, local_vocab, pval, pidx);
  else
    hipLaunchKernelGGL((argmax_partial_kernel<BF16>), grid, dim3(AM_THREADS), 0, s,
                       (const uint16_t*)logits, row_stride, local_vocab, pval, pidx);
  hipLaunchKernelGGL(argmax_final_kernel, dim3(num_seqs), dim3(64), 0, s, pval, pidx, AM_SPLITS,
                     (int64_t)1, (int64_t)AM_SPLITS, rec_val, rec_idx, index_offset,
                     (int64_t*)nullptr, (int64_t*)nullptr, (int64_t*)nullptr, (int*)nullptr,
                     (int64_t*)nullptr, (const int*)nullptr, 0, 0);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}

/* gathered: world records back to back (rank order); the winner per row (ties -> lowest global
 * index, i.e. torch.argmax of the gathered logits) goes to next_tokens and, with the state tensors,
 * advances the decode batch as nmv_greedy_sample_advance does. */
extern "C" int nmv_greedy_sample_finish(int64_t* next_tokens, const void* gathered, int world,
                                        int num_seqs, int64_t* input_ids, int64_t* positions,
                                        int* seq_lens, int64_t* slot_mapping, const int* block_tables,
                                        int max_blocks_per_seq, int block_size, void* stream) {
  NMV_CHECK(world >= 1 && world <= 64, "greedy_sample_finish: world must be 1..64");
  NMV_CHECK(positions == nullptr || (input_ids && seq_lens && slot_mapping && block_tables &&
                                     max_blocks_per_seq > 0 && block_size > 0),
            "greedy_sample: the state advance needs every state tensor");
  if (num_seqs == 0) return NMV_OK;
  const int padded = nmv_greedy_record_elems(num_seqs);
  const float* val = reinterpret_cast<const float*>(gathered);
  const int* idx = reinterpret_cast<const int*>(val + padded);
  hipLaunchKernelGGL(argmax_final_kernel, dim3(num_seqs), dim3(64), 0, (hipStream_t)stream, val, idx, world,
                     (int64_t)2 * padded, (int64_t)1, (float*)nullptr, (int*)nullptr, 0, next_tokens,
                     input_ids, positions, seq_lens, slot_mapping, block_tables, max_blocks_per_seq,
                     block_size);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}


/* ------------------------------------------------------------------------------------------------
 * Infinity-Cache prefetch (MI355X-first, not an op of nm-vllm 0.5.1).  At small batches a decoder layer streams
 * its weights once and leaves HBM idle most of the time (60 us per layer for 112.6 MB at B = 1: 35 % of 8 TB/s), while
 * each of its GEMM launches pays an HBM round trip before its first MFMA.  The 256 MiB Infinity Cache holds two layers'
 * codes: this kernel, launched on a SIDE stream while layer L computes, reads layer L + 1's weight tensors once (plain
 * loads, default cache policy, nothing stored), so that layer L + 1's GEMMs find them on-die -- measured on the GEMMs
 * alone (tools/bench_gemm.py, NMV_BENCH_NCOPY=1): 38.3 -> 31.5 us per layer at M = 1.  A hint only: results never depend
 * on it. */
__global__ __launch_bounds__(256) void prefetch_l3_kernel(const uint4* __restrict__ p, int64_t n_vec) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  uint32_t keep = 0;
  // 8 independent 16-byte loads in flight per lane
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 7 * stride < n_vec; i += 8 * stride) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) keep ^= v[u].x ^ v[u].w;
  }
  for (; i < n_vec; i += stride) keep ^= p[i].x;
  asm volatile("" ::"v"(keep));   // the loads must happen; their values go nowhere
}

extern "C" int nmv_prefetch_l3(const void* ptr, int64_t bytes, int workgroups, void* stream) {
  NMV_CHECK(bytes >= 0 && workgroups >= 0, "prefetch_l3: negative size");
  NMV_CHECK(((uintptr_t)ptr & 15) == 0, "prefetch_l3: the pointer must be 16-byte aligned");
  const int64_t n_vec = bytes / 16;
  if (ptr == nullptr || n_vec == 0) return NMV_OK;
  const int wgs = workgroups > 0 ? workgroups : 64;
  hipLaunchKernelGGL(prefetch_l3_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, (const uint4*)ptr, n_vec);
  NMV_LAUNCH_CHECK();
  return NMV_OK;
}
