// BM25 document side on the GPU: per-document token counting and TF weighting — what fastembed's
// Bm25._term_frequency does for SparseTextEmbedding("Qdrant/bm25").embed(texts)
// (reference call sites: src/voitta/services/sparse_embedding.py:25,49;
//  scripts/build_sparse_vectors.py:124,170; SURVEY.md a6 [EXT]):
//   for each distinct stem of the document, in a dict keyed by abs(murmur3(stem)):
//       tf = count * (k + 1) / (count + k * (1 - b + b * doc_len / avg_len))      (Python floats)
//   k = 1.2, b = 0.75, avg_len = 256.0 by default, doc_len = number of stemmed tokens.
// Input is the stream of hashed stems in text order (tokenise / stop-words / stemming / murmur3
// is host string work: bm25_text.cpp). Output per document: distinct ids ascending (a sparse
// vector is a set; Qdrant sorts by index on ingestion) with tf in f64 — the exact value the
// reference hands to Qdrant — and optionally rounded to f32, the precision Qdrant stores.
// Deviation: two different stems of one document whose 31-bit hashes collide are merged here;
// the reference keeps only the later stem's weight (p ~ 1e-6 per document).
//
// One wave per document, all-pairs counting out of LDS (documents are <= ~130 tokens for the
// reference's 512-character chunks); integer work plus one f64 divide per term.
// HBM traffic: 4 B in + 16 B out per token, negligible next to the encoder.

#include "engine_internal.h"

namespace vr {

constexpr int kBm25LdsTokens = 1024;  // per wave; longer documents run out of global memory

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ __launch_bounds__(256) void bm25_tf_kernel(const int64_t* __restrict__ off,
                                                      const int32_t* __restrict__ ids, int64_t n_docs,
                                                      double k1, double b, double avg_len,
                                                      int32_t* __restrict__ marks_g,
                                                      int32_t* __restrict__ out_cnt,
                                                      int32_t* __restrict__ out_idx,
                                                      double* __restrict__ out_val64,
                                                      float* __restrict__ out_val32) {
  __shared__ int32_t s_ids[4][kBm25LdsTokens];
  __shared__ int32_t s_marks[4][kBm25LdsTokens];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t doc = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  if (doc >= n_docs) return;
  const int64_t begin = off[doc];
  const int len = static_cast<int>(off[doc + 1] - begin);
  const bool in_lds = len <= kBm25LdsTokens;
  const int32_t* t = ids + begin;
  int32_t* marks = marks_g + begin;
  if (in_lds) {
    for (int i = lane; i < len; i += 64) s_ids[wave][i] = ids[begin + i];
    t = s_ids[wave];
    marks = s_marks[wave];
  }
  // LDS (or, for long documents, global) writes of this wave must be visible to its other lanes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();

  // pass 1: marks[i] = occurrences of ids[i] when i is its first occurrence, else 0
  int n_first = 0;
  for (int i = lane; i < len; i += 64) {
    const int32_t id = t[i];
    int cnt = 0;
    bool first = true;
    for (int j = 0; j < len; ++j) {
      const bool eq = t[j] == id;
      cnt += eq;
      first = first && !(eq && j < i);
    }
    marks[i] = first ? cnt : 0;
    n_first += first;
  }
  if (in_lds) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  } else {
    __threadfence();  // marks went to global memory: make this wave's later loads see them
  }

  // pass 2: output position = number of distinct ids below mine
  const double dl = static_cast<double>(len);
  const double norm = (1.0 - b) + (b * dl) / avg_len;
  for (int i = lane; i < len; i += 64) {
    const int c = marks[i];
    if (c == 0) continue;
    const int32_t id = t[i];
    int pos = 0;
    for (int j = 0; j < len; ++j) pos += (marks[j] > 0) && (t[j] < id);
    const double dc = static_cast<double>(c);
    const double tf = (dc * (k1 + 1.0)) / (dc + k1 * norm);
    out_idx[begin + pos] = id;
    if (out_val64) out_val64[begin + pos] = tf;
    if (out_val32) out_val32[begin + pos] = static_cast<float>(tf);
  }
  n_first = wave_sum_i32(n_first);
  if (lane == 0) out_cnt[doc] = n_first;
}

int bm25_tf(vr_engine* e, const int64_t* tok_off_dev, const int32_t* tok_ids_dev, int64_t n_docs,
            int64_t n_tokens, double k, double b, double avg_len, int32_t* out_cnt_dev,
            int32_t* out_idx_dev, double* out_val64_dev, float* out_val32_dev) {
  if (n_docs <= 0) return 0;
  VR_TRY(e->bm_marks.grow(std::max<int64_t>(n_tokens, 1), 0, e->stream));
  hipLaunchKernelGGL(bm25_tf_kernel, dim3(static_cast<unsigned>((n_docs + 3) / 4)), dim3(256), 0,
                     e->stream, tok_off_dev, tok_ids_dev, n_docs, k, b, avg_len, e->bm_marks.p,
                     out_cnt_dev, out_idx_dev, out_val64_dev, out_val32_dev);
  VR_HIP(hipGetLastError());
  return 0;
}

}  // namespace vr

using namespace vr;

extern "C" int vr_bm25_tf(vr_engine* e, const int64_t* tok_off, const int32_t* tok_ids, int64_t n_docs,
                          int mem, double k, double b, double avg_len, int32_t* out_cnt,
                          int32_t* out_idx, double* out_val) {
  VR_CHECK(e != nullptr, "null engine");
  VR_HIP(hipSetDevice(e->device));
  VR_CHECK(n_docs >= 0 && (n_docs == 0 || (tok_off && out_cnt)), "bad arguments");
  VR_CHECK(mem == VR_MEM_HOST || mem == VR_MEM_DEVICE, "bad mem %d", mem);
  VR_CHECK(avg_len > 0.0, "avg_len must be positive");
  if (n_docs == 0) return 0;
  std::lock_guard<std::mutex> lock(e->wmu);
  if (mem == VR_MEM_DEVICE) {
    int64_t n_tokens = 0;
    VR_HIP(hipMemcpyAsync(&n_tokens, tok_off + n_docs, sizeof(int64_t), hipMemcpyDeviceToHost, e->stream));
    VR_HIP(hipStreamSynchronize(e->stream));
    return bm25_tf(e, tok_off, tok_ids, n_docs, n_tokens, k, b, avg_len, out_cnt, out_idx, out_val, nullptr);
  }
  VR_CHECK(tok_off[0] == 0, "token offsets must start at 0");
  const int64_t n_tokens = tok_off[n_docs];
  const int64_t cap = std::max<int64_t>(n_tokens, 1);
  VR_TRY(e->stage_off.grow(n_docs + 1, 0, e->stream));
  VR_TRY(e->stage_idx.grow(cap, 0, e->stream));
  VR_TRY(e->stage_i32a.grow(n_docs, 0, e->stream));
  VR_TRY(e->stage_i32b.grow(cap, 0, e->stream));
  VR_TRY(e->stage_f64.grow(cap, 0, e->stream));
  VR_HIP(hipMemcpyAsync(e->stage_off.p, tok_off, sizeof(int64_t) * static_cast<size_t>(n_docs + 1),
                        hipMemcpyHostToDevice, e->stream));
  if (n_tokens > 0)
    VR_HIP(hipMemcpyAsync(e->stage_idx.p, tok_ids, sizeof(int32_t) * static_cast<size_t>(n_tokens),
                          hipMemcpyHostToDevice, e->stream));
  VR_TRY(bm25_tf(e, e->stage_off.p, e->stage_idx.p, n_docs, n_tokens, k, b, avg_len, e->stage_i32a.p,
                 e->stage_i32b.p, e->stage_f64.p, nullptr));
  VR_HIP(hipMemcpyAsync(out_cnt, e->stage_i32a.p, sizeof(int32_t) * static_cast<size_t>(n_docs),
                        hipMemcpyDeviceToHost, e->stream));
  if (n_tokens > 0) {
    VR_HIP(hipMemcpyAsync(out_idx, e->stage_i32b.p, sizeof(int32_t) * static_cast<size_t>(n_tokens),
                          hipMemcpyDeviceToHost, e->stream));
    VR_HIP(hipMemcpyAsync(out_val, e->stage_f64.p, sizeof(double) * static_cast<size_t>(n_tokens),
                          hipMemcpyDeviceToHost, e->stream));
  }
  VR_HIP(hipStreamSynchronize(e->stream));
  return 0;
}
