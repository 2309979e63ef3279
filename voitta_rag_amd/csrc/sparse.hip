// Sparse (BM25) side of the store. Replaces what the Qdrant server does for the "bm25" sparse
// vector configured with Modifier.IDF (reference: src/voitta/services/vector_store.py:95-99) on
//   upsert  (vector_store.py:291-298,313; scripts/build_sparse_vectors.py:176-207)
//   query   (vector_store.py:647-656)
//
// HBM layout: SELL-64 — every upsert batch is cut into slices of 64 consecutive rows, one row
// per wavefront lane. Inside a slice the (token id, tf weight) entries of a row are sorted by
// token id and stored in chunks of four: entry j of lane l sits at off + (j/4)*256 + l*4 + j%4,
// so a wave reads 1 KiB of ids per instruction and each lane walks its own row in ascending
// token-id order — the summation order the oracle restates. Padding ids are -1.
//
// Scoring is a brute-force scan (HBM bound; algorithmic bytes = 4 B per stored id, values are
// touched only on a hit): score(d) = sum over shared terms, ascending id, of (q_t*idf_t)*d_t with
// every multiply and add rounded to f32 separately; idf_t = ln(1 + (N - df_t + 0.5)/(df_t + 0.5))
// with the argument formed in f32 and ln taken in f64 then rounded (SURVEY.md a13 [EXT]).
// Document frequencies live in an open-addressing hash table in HBM (integer atomics only, so
// the table content does not depend on arrival order).

#include "engine_internal.h"
#include "sparse_device.h"
#include "topk_device.h"

#include <algorithm>

namespace vr {

// ---- document-frequency table ---------------------------------------------------------------

__device__ __forceinline__ void df_add(int32_t* keys, int32_t* cnt, int64_t cap, int32_t id,
                                       int32_t delta, int32_t* distinct) {
  uint64_t h = df_hash(id) & (cap - 1);
  for (int64_t probe = 0; probe < cap; ++probe) {
    int32_t cur = keys[h];
    if (cur == id) break;
    if (cur == -1) {
      int32_t prev = atomicCAS(&keys[h], -1, id);
      if (prev == -1) {
        atomicAdd(distinct, 1);
        break;
      }
      if (prev == id) break;
    }
    h = (h + 1) & (cap - 1);
  }
  atomicAdd(&cnt[h], delta);
}

__global__ void df_init_kernel(int32_t* keys, int32_t* cnt, int64_t cap) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < cap) {
    keys[i] = -1;
    cnt[i] = 0;
  }
}

__global__ void df_rehash_kernel(const int32_t* okeys, const int32_t* ocnt, int64_t ocap,
                                 int32_t* keys, int32_t* cnt, int64_t cap, int32_t* distinct) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < ocap && okeys[i] != -1) df_add(keys, cnt, cap, okeys[i], ocnt[i], distinct);
}

static int df_ensure(vr_engine* e, int64_t incoming) {
  if (e->df_cap == 0) {
    e->df_cap = 1 << 20;
    VR_TRY(e->df_keys.grow(e->df_cap, 0, e->stream));
    VR_TRY(e->df_cnt.grow(e->df_cap, 0, e->stream));
    VR_HIP(hipMalloc(reinterpret_cast<void**>(&e->df_distinct), sizeof(int32_t)));
    VR_HIP(hipMemsetAsync(e->df_distinct, 0, sizeof(int32_t), e->stream));
    hipLaunchKernelGGL(df_init_kernel, dim3(static_cast<unsigned>((e->df_cap + 255) / 256)),
                       dim3(256), 0, e->stream, e->df_keys.p, e->df_cnt.p, e->df_cap);
    e->df_bound = 0;
  }
  if ((e->df_bound + incoming) * 2 > e->df_cap) {
    // the bound counts every nnz since the last exact read; refresh it before growing
    int32_t exact = 0;
    VR_HIP(hipMemcpyAsync(&exact, e->df_distinct, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    VR_HIP(hipStreamSynchronize(e->stream));
    e->df_bound = exact;
    if ((e->df_bound + incoming) * 2 > e->df_cap) {
      int64_t ncap = e->df_cap;
      while ((e->df_bound + incoming) * 2 > ncap) ncap *= 2;
      DevArray<int32_t> nk, nc;
      VR_TRY(nk.grow(ncap, 0, e->stream));
      VR_TRY(nc.grow(ncap, 0, e->stream));
      hipLaunchKernelGGL(df_init_kernel, dim3(static_cast<unsigned>((ncap + 255) / 256)), dim3(256),
                         0, e->stream, nk.p, nc.p, ncap);
      VR_HIP(hipMemsetAsync(e->df_distinct, 0, sizeof(int32_t), e->stream));
      hipLaunchKernelGGL(df_rehash_kernel, dim3(static_cast<unsigned>((e->df_cap + 255) / 256)),
                         dim3(256), 0, e->stream, e->df_keys.p, e->df_cnt.p, e->df_cap, nk.p, nc.p,
                         ncap, e->df_distinct);
      VR_HIP(hipStreamSynchronize(e->stream));
      e->df_keys.release();
      e->df_cnt.release();
      e->df_keys = nk;
      e->df_cnt = nc;
      e->df_cap = ncap;
    }
  }
  e->df_bound += incoming;
  return 0;
}

// ---- SELL-64 build ----------------------------------------------------------------------------

// One wave per slice; lane l copies row (row_base + l) of the batch into its SELL column, padding
// with -1. The row's entries are idx/val[begin[r] .. begin[r] + count) where count is cnt[r] when
// cnt is given (padded layout written by bm25_tf_kernel) and begin[r+1] - begin[r] otherwise
// (plain CSR). Rows must already be sorted by token id (vr_upsert sorts host input; the device
// BM25 kernel emits sorted rows).
__global__ __launch_bounds__(64) void sell_build_kernel(const SliceDesc* __restrict__ slices,
                                                        int64_t slice0, int64_t batch_first_row,
                                                        const int64_t* __restrict__ begin,
                                                        const int32_t* __restrict__ cnt,
                                                        const int32_t* __restrict__ idx,
                                                        const float* __restrict__ val,
                                                        int32_t* __restrict__ sidx,
                                                        float* __restrict__ sval,
                                                        int32_t* __restrict__ row_slice) {
  const SliceDesc d = slices[slice0 + blockIdx.x];
  const int lane = threadIdx.x;
  int64_t b0 = 0;
  int len = 0;
  if (lane < d.nrows) {
    int64_t local = d.row_base + lane - batch_first_row;
    b0 = begin[local];
    len = cnt ? cnt[local] : static_cast<int>(begin[local + 1] - b0);
    row_slice[d.row_base + lane] = static_cast<int32_t>(slice0 + blockIdx.x);
  }
  for (int j = 0; j < d.width; ++j) {
    int64_t dst = d.off + static_cast<int64_t>(j >> 2) * 256 + lane * 4 + (j & 3);
    bool has = j < len;
    sidx[dst] = has ? idx[b0 + j] : -1;
    sval[dst] = has ? val[b0 + j] : 0.0f;
  }
}

// +1 document frequency for every real entry of the freshly built slices. Term frequencies are
// Zipfian: sent straight to the global table, the atomics of a batch pile up on a few hundred hot
// counters (0.78 ms per 2048-chunk batch). Each block first counts its 4096 entries in an LDS hash
// and then sends one global update per DISTINCT term it saw.
// (round 3: 512 entries per block instead of 4096. The 90k entries of a 2200-row batch were 22 blocks, each ending in
// up to 32 DEPENDENT global hash-table updates per thread — 1.07 ms of every 47-ms index step on a chip with 256 CUs idle;
// what a block's LDS table still merges are a row's neighbours' common terms, the parallelism is worth more)
constexpr int kDfChunk = 512;   // entries per block
constexpr int kDfSlots = 1024;  // LDS hash slots (load factor <= 0.5)

__global__ __launch_bounds__(256) void df_update_region_kernel(const int32_t* __restrict__ sidx, int64_t begin,
                                                               int64_t end, int32_t* keys, int32_t* cnt,
                                                               int64_t cap, int32_t* distinct, int sign) {
  __shared__ int32_t lk[kDfSlots];
  __shared__ int32_t lc[kDfSlots];
  for (int i = threadIdx.x; i < kDfSlots; i += 256) {
    lk[i] = -1;
    lc[i] = 0;
  }
  __syncthreads();
  const int64_t b0 = begin + static_cast<int64_t>(blockIdx.x) * kDfChunk;
  const int64_t b1 = b0 + kDfChunk < end ? b0 + kDfChunk : end;
  for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) {
    const int32_t id = sidx[i];
    if (id < 0) continue;
    uint32_t h = static_cast<uint32_t>(df_hash(id)) & (kDfSlots - 1);
    while (true) {  // terminates: at most kDfChunk distinct ids in kDfSlots slots
      const int32_t cur = lk[h];
      if (cur == id) break;
      if (cur == -1) {
        const int32_t prev = atomicCAS(&lk[h], -1, id);
        if (prev == -1 || prev == id) break;
      }
      h = (h + 1) & (kDfSlots - 1);
    }
    atomicAdd(&lc[h], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kDfSlots; i += 256)
    if (lk[i] != -1) df_add(keys, cnt, cap, lk[i], sign * lc[i], distinct);
}

int sparse_append(vr_engine* e, int64_t n, int64_t first_row, const int32_t* cnt_host,
                  const int64_t* begin_dev, const int32_t* cnt_dev, const int32_t* idx_dev,
                  const float* val_dev, bool account) {
  if (n <= 0) return 0;
  int64_t nnz = 0;
  for (int64_t r = 0; r < n; ++r) {
    VR_CHECK(cnt_host[r] >= 0, "negative sparse row length");
    nnz += cnt_host[r];
  }
  if (account) VR_TRY(df_ensure(e, nnz));

  const int64_t slice0 = static_cast<int64_t>(e->slices_host.size());
  int64_t used = e->sp_used;
  for (int64_t r = 0; r < n; r += 64) {
    SliceDesc d;
    d.off = used;
    d.row_base = static_cast<int32_t>(first_row + r);
    d.nrows = static_cast<int32_t>(std::min<int64_t>(64, n - r));
    int32_t w = 0;
    for (int i = 0; i < d.nrows; ++i) w = std::max(w, cnt_host[r + i]);
    d.width = (w + 3) / 4 * 4;
    d.pad = 0;
    used += static_cast<int64_t>(d.width) * 64;
    e->slices_host.push_back(d);
  }
  const int64_t n_new = static_cast<int64_t>(e->slices_host.size()) - slice0;
  VR_TRY(e->sp_idx.grow(std::max<int64_t>(used, 1), e->sp_used, e->stream));
  VR_TRY(e->sp_val.grow(std::max<int64_t>(used, 1), e->sp_used, e->stream));
  VR_TRY(e->slices.grow(static_cast<int64_t>(e->slices_host.size()), slice0, e->stream));
  VR_HIP(hipMemcpyAsync(e->slices.p + slice0, e->slices_host.data() + slice0,
                        sizeof(SliceDesc) * static_cast<size_t>(n_new), hipMemcpyHostToDevice,
                        e->stream));
  VR_HIP(hipStreamSynchronize(e->stream));  // slices_host may reallocate on the next append
  hipLaunchKernelGGL(sell_build_kernel, dim3(static_cast<unsigned>(n_new)), dim3(64), 0, e->stream,
                     e->slices.p, slice0, first_row, begin_dev, cnt_dev, idx_dev, val_dev, e->sp_idx.p,
                     e->sp_val.p, e->row_slice.p);
  const int64_t region = used - e->sp_used;
  if (region > 0 && account)
    hipLaunchKernelGGL(df_update_region_kernel, dim3(static_cast<unsigned>((region + kDfChunk - 1) / kDfChunk)),
                       dim3(256), 0, e->stream, e->sp_idx.p, e->sp_used, used, e->df_keys.p,
                       e->df_cnt.p, e->df_cap, e->df_distinct, 1);
  VR_HIP(hipGetLastError());
  e->sp_used = used;
  e->n_slices_dev = static_cast<int64_t>(e->slices_host.size());
  if (account) e->n_sparse_points += n;
  return inv_append(e, slice0, n_new, first_row, n, nnz);  // the same rows, by term (invert.hip)
}

// ---- query ------------------------------------------------------------------------------------

constexpr int kQHash = 4096;   // LDS hash slots for the query terms (4 x kMaxQueryTerms)
constexpr int kSparseWaves = 16;  // 1024-thread blocks: 2 per CU = every wave slot busy (the scan
                                  // is a latency-bound gather; it wants all the loads in flight)

// One kernel per sparse query. Prologue (every block, redundantly): read the <= kMaxQueryTerms sorted query
// terms straight from the pinned host scratch, weight them — q_t * idf_t with idf from the
// document-frequency table, or as given — and build an LDS hash. Body: SELL-64 scan, one row per
// lane, ascending-id accumulation. FUSED keeps the k best (score, row) keys per wave
// (topk_device.h) and writes one 64-entry list per block for merge_lists_kernel; otherwise a score
// per row is written (-inf = shares no term / filtered).
template <bool FUSED>
__global__ __launch_bounds__(kSparseWaves * 64) void sparse_scores_kernel(
    const SliceDesc* __restrict__ slices, int64_t n_slices, const int32_t* __restrict__ sidx,
    const float* __restrict__ sval, const int32_t* __restrict__ q_idx, const float* __restrict__ q_val,
    int nnz, int weights_given, const int32_t* __restrict__ df_keys, const int32_t* __restrict__ df_cnt,
    int64_t df_cap, float n_points, const uint8_t* __restrict__ mask, float* __restrict__ scores, int k,
    uint64_t* __restrict__ cand) {
  __shared__ int32_t hk[kQHash];
  __shared__ float hv[kQHash];
  __shared__ int32_t t_id[kMaxQueryTerms];
  __shared__ float t_w[kMaxQueryTerms];
  __shared__ uint64_t lists[kSparseWaves * kListLen];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  uint64_t* list = lists + wave * kListLen;
  lists[threadIdx.x] = 0ull;  // blockDim.x == kSparseWaves * kListLen
  for (int i = threadIdx.x; i < kQHash; i += kSparseWaves * 64) hk[i] = -1;
  if (static_cast<int>(threadIdx.x) < nnz) {
    const int32_t id = q_idx[threadIdx.x];
    const float w = sparse_query_weight(q_val[threadIdx.x], id, weights_given, df_keys, df_cnt, df_cap, n_points);
    t_id[threadIdx.x] = id;
    t_w[threadIdx.x] = w;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // a handful of terms (<= kMaxQueryTerms); serial insert keeps the table deterministic
    for (int t = 0; t < nnz; ++t) {
      const int32_t id = t_id[t];
      uint32_t h = df_hash(id) & (kQHash - 1);
      while (hk[h] != -1 && hk[h] != id) h = (h + 1) & (kQHash - 1);
      hk[h] = id;
      hv[h] = t_w[t];
    }
  }
  __syncthreads();

  const int64_t wave_stride = static_cast<int64_t>(gridDim.x) * kSparseWaves;
  for (int64_t s = static_cast<int64_t>(blockIdx.x) * kSparseWaves + wave; s < n_slices; s += wave_stride) {
    const SliceDesc d = slices[s];
    float acc = 0.0f;
    bool hit = false;
    const int4* ip = reinterpret_cast<const int4*>(sidx + d.off) + lane;
    for (int c = 0; c < d.width / 4; ++c) {
      int4 ids = ip[c * 64];
      int32_t id4[4] = {ids.x, ids.y, ids.z, ids.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int32_t id = id4[u];
        if (id < 0) continue;
        uint32_t h = df_hash(id) & (kQHash - 1);
        int32_t cur = hk[h];
        while (cur != -1 && cur != id) {
          h = (h + 1) & (kQHash - 1);
          cur = hk[h];
        }
        if (cur == id) {
          float v = sval[d.off + static_cast<int64_t>(c) * 256 + lane * 4 + u];
          acc = __fadd_rn(acc, __fmul_rn(hv[h], v));
          hit = true;
        }
      }
    }
    const bool in_slice = lane < d.nrows;
    const int64_t row = d.row_base + lane;
    const bool ok = in_slice && hit && mask[in_slice ? row : 0];
    if (FUSED) {
      wave_offer(list, k, ok ? topk_make_key(acc, row) : 0ull, 0, ok, lane);
    } else if (in_slice) {
      scores[row] = ok ? acc : -__builtin_inff();
    }
  }
  if (FUSED) {  // fold the block's lists into one: cand[block][64], descending, zero padded
    block_merge_lists(lists, kListLen, kSparseWaves, wave, lane);
    if (wave == 0) cand[static_cast<int64_t>(blockIdx.x) * kListLen + lane] = list[lane];
  }
}

// fused_k == 0: scores of every row into e->sp_scores; fused_k > 0: top fused_k keys to out_keys_dev
static int sparse_run(vr_engine* e, const int32_t* q_idx_host, const float* q_val_host, int nnz,
                      const uint8_t* mask_dev, bool weights_given, int fused_k, uint64_t* out_keys_dev) {
  VR_CHECK(nnz >= 1 && nnz <= kMaxQueryTerms, "sparse query with %d terms (1..%d supported)", nnz,
           kMaxQueryTerms);
  // ascending token id, duplicates merged by keeping the first (Qdrant sorts sparse vectors by
  // index on ingestion [EXT]); the hash lookup itself is order independent.
  std::vector<std::pair<int32_t, float>> q(static_cast<size_t>(nnz));
  for (int i = 0; i < nnz; ++i) q[static_cast<size_t>(i)] = {q_idx_host[i], q_val_host[i]};
  std::stable_sort(q.begin(), q.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  q.erase(std::unique(q.begin(), q.end(), [](const auto& a, const auto& b) { return a.first == b.first; }),
          q.end());
  nnz = static_cast<int>(q.size());
  // the kernel reads the terms from the pinned scratch over PCIe: no copy, no extra launch
  int32_t* hid = pin_host<int32_t>(e, kPinSparseIds);
  float* hval = pin_host<float>(e, kPinSparseVals);
  for (int i = 0; i < nnz; ++i) {
    hid[i] = q[static_cast<size_t>(i)].first;
    hval[i] = q[static_cast<size_t>(i)].second;
  }
  const int32_t* did = pin_dev<int32_t>(e, kPinSparseIds);
  const float* dval = pin_dev<float>(e, kPinSparseVals);
  const float n_points = static_cast<float>(e->n_sparse_points);
  if (fused_k && inv_usable(e, nnz))  // a few terms: their postings only
    return inv_scan_topk(e, hid, hval, nnz, weights_given, n_points, mask_dev, fused_k, out_keys_dev);
  if (fused_k) {
    int64_t blocks = std::min<int64_t>((e->n_slices_dev + kSparseWaves - 1) / kSparseWaves, kScanBlocks);
    if (blocks < 1) blocks = 1;
    VR_TRY(e->sp_cand.grow(blocks * kListLen, 0, e->stream));  // own buffer: may overlap a dense search
    // algorithmic bytes: every stored id once (4 B) and one mask byte per row
    prof_begin(e, VR_PROF_SPARSE_SCAN, 4.0 * static_cast<double>(e->sp_used) + static_cast<double>(e->n_rows));
    hipLaunchKernelGGL((sparse_scores_kernel<true>), dim3(static_cast<unsigned>(blocks)), dim3(kSparseWaves * 64),
                       0, e->stream, e->slices.p, e->n_slices_dev, e->sp_idx.p, e->sp_val.p, did, dval, nnz,
                       weights_given ? 1 : 0, e->df_keys.p, e->df_cnt.p, e->df_cap, n_points, mask_dev,
                       static_cast<float*>(nullptr), fused_k, e->sp_cand.p);
    prof_end(e);
    VR_HIP(hipGetLastError());
    return topk_merge_lists(e, e->sp_cand.p, static_cast<int>(blocks), 1, fused_k, out_keys_dev);
  }
  VR_TRY(e->sp_scores.grow(e->cap_rows, 0, e->stream));
  VR_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(e->sp_scores.p), 0xFF800000u,
                           static_cast<size_t>(e->n_rows), e->stream));
  if (e->n_slices_dev > 0) {
    int64_t blocks = std::min<int64_t>((e->n_slices_dev + kSparseWaves - 1) / kSparseWaves, 1024);
    // algorithmic bytes: every stored id once (4 B), one mask byte and one score per row
    prof_begin(e, VR_PROF_SPARSE_SCAN, 4.0 * static_cast<double>(e->sp_used) + 5.0 * static_cast<double>(e->n_rows));
    hipLaunchKernelGGL((sparse_scores_kernel<false>), dim3(static_cast<unsigned>(blocks)), dim3(kSparseWaves * 64),
                       0, e->stream, e->slices.p, e->n_slices_dev, e->sp_idx.p, e->sp_val.p, did, dval, nnz,
                       weights_given ? 1 : 0, e->df_keys.p, e->df_cnt.p, e->df_cap, n_points, mask_dev,
                       e->sp_scores.p, 0, static_cast<uint64_t*>(nullptr));
    prof_end(e);
  }
  VR_HIP(hipGetLastError());
  return 0;
}

int sparse_scores(vr_engine* e, const int32_t* q_idx_host, const float* q_val_host, int nnz,
                  const uint8_t* mask_dev, bool weights_given) {
  return sparse_run(e, q_idx_host, q_val_host, nnz, mask_dev, weights_given, 0, nullptr);
}

int sparse_scan_topk(vr_engine* e, const int32_t* q_idx_host, const float* q_val_host, int nnz, int k,
                     const uint8_t* mask_dev, bool weights_given, uint64_t* out_keys_dev) {
  VR_CHECK(k >= 1 && k <= kListLen, "fused selection serves k <= %d", kListLen);
  return sparse_run(e, q_idx_host, q_val_host, nnz, mask_dev, weights_given, k, out_keys_dev);
}

// ---- delete -----------------------------------------------------------------------------------

// rows are unique (host de-duplicates). counters[0] += rows that were live, counters[1] += of
// those, rows that carried a sparse vector.
__global__ void delete_rows_kernel(const int64_t* __restrict__ rows, int64_t n, int64_t n_rows,
                                   uint8_t* __restrict__ live, const int32_t* __restrict__ row_slice,
                                   const SliceDesc* __restrict__ slices,
                                   const int32_t* __restrict__ sidx, int32_t* keys, int32_t* cnt,
                                   int64_t cap, int32_t* distinct, int32_t* counters) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t row = rows[i];
  if (row < 0 || row >= n_rows || !live[row]) return;
  live[row] = 0;
  atomicAdd(&counters[0], 1);
  int32_t s = row_slice[row];
  if (s < 0) return;
  atomicAdd(&counters[1], 1);
  const SliceDesc d = slices[s];
  int lane = static_cast<int>(row - d.row_base);
  for (int j = 0; j < d.width; ++j) {
    int32_t id = sidx[d.off + static_cast<int64_t>(j >> 2) * 256 + lane * 4 + (j & 3)];
    if (id >= 0) df_add(keys, cnt, cap, id, -1, distinct);
  }
}

int sparse_delete_rows(vr_engine* e, const int64_t* rows_dev, int64_t n, int64_t* n_deleted,
                       int64_t* n_sparse_deleted) {
  *n_deleted = 0;
  *n_sparse_deleted = 0;
  if (n <= 0) return 0;
  VR_TRY(df_ensure(e, 0));
  VR_TRY(e->stage_i32a.grow(2, 0, e->stream));
  VR_HIP(hipMemsetAsync(e->stage_i32a.p, 0, 2 * sizeof(int32_t), e->stream));
  hipLaunchKernelGGL(delete_rows_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                     e->stream, rows_dev, n, e->n_rows, e->live.p, e->row_slice.p, e->slices.p,
                     e->sp_idx.p, e->df_keys.p, e->df_cnt.p, e->df_cap, e->df_distinct,
                     e->stage_i32a.p);
  VR_HIP(hipGetLastError());
  int32_t c[2] = {0, 0};
  VR_HIP(hipMemcpyAsync(c, e->stage_i32a.p, sizeof(c), hipMemcpyDeviceToHost, e->stream));
  VR_HIP(hipStreamSynchronize(e->stream));
  *n_deleted = c[0];
  *n_sparse_deleted = c[1];
  return 0;
}

// ---- document frequencies across shards (sharded.py; SURVEY.md §8e: the df deltas of every upsert / delete batch
// are summed over the shards at index time, so that a query finds the collection-wide statistic — Qdrant's
// Modifier.IDF scope, vector_store.py:95-99 — in its own engine's table and needs no exchange of its own) ----------

// one thread per (row, entry slot): the term ids of the listed rows in a fixed-stride layout, -1 where the row has no
// such entry, is dead, or carries no sparse vector; counter += live rows with a sparse vector
__global__ void sparse_row_ids_kernel(const int64_t* __restrict__ rows, int64_t n, int stride, int64_t n_rows,
                                      const uint8_t* __restrict__ live, const int32_t* __restrict__ row_slice,
                                      const SliceDesc* __restrict__ slices, const int32_t* __restrict__ sidx,
                                      int32_t* __restrict__ out, unsigned long long* counter) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n * stride) return;
  const int64_t i = t / stride;
  const int j = static_cast<int>(t - i * stride);
  const int64_t row = rows[i];
  int32_t id = -1;
  if (row >= 0 && row < n_rows && live[row]) {
    const int32_t s = row_slice[row];
    if (s >= 0) {
      const SliceDesc d = slices[s];
      const int lane = static_cast<int>(row - d.row_base);
      if (j < d.width) id = sidx[d.off + static_cast<int64_t>(j >> 2) * 256 + lane * 4 + (j & 3)];
      if (j == 0) atomicAdd(counter, 1ull);
    }
  }
  out[t] = id;
}

int sparse_max_width(const vr_engine* e) {
  int w = 0;
  for (const SliceDesc& d : e->slices_host) w = std::max(w, d.width);
  return w;
}

int sparse_row_ids(vr_engine* e, const int64_t* rows_dev, int64_t n, int stride, int32_t* out_dev, int64_t* n_points_host) {
  *n_points_host = 0;
  if (n <= 0 || stride <= 0) return 0;
  VR_TRY(e->stage_i64b.grow(1, 0, e->stream));
  unsigned long long* counter = reinterpret_cast<unsigned long long*>(e->stage_i64b.p);
  VR_HIP(hipMemsetAsync(counter, 0, sizeof(unsigned long long), e->stream));
  const int64_t total = n * stride;
  hipLaunchKernelGGL(sparse_row_ids_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, e->stream,
                     rows_dev, n, stride, e->n_rows, e->live.p, e->row_slice.p, e->slices.p, e->sp_idx.p, out_dev, counter);
  VR_HIP(hipGetLastError());
  unsigned long long c = 0;
  VR_HIP(hipMemcpyAsync(&c, counter, sizeof(c), hipMemcpyDeviceToHost, e->stream));
  VR_HIP(hipStreamSynchronize(e->stream));
  *n_points_host = static_cast<int64_t>(c);
  return 0;
}

int sparse_df_apply(vr_engine* e, const int32_t* ids_dev, int64_t n, int sign) {
  if (n <= 0) return 0;
  VR_TRY(df_ensure(e, sign > 0 ? n : 0));
  hipLaunchKernelGGL(df_update_region_kernel, dim3(static_cast<unsigned>((n + kDfChunk - 1) / kDfChunk)), dim3(256), 0,
                     e->stream, ids_dev, static_cast<int64_t>(0), n, e->df_keys.p, e->df_cnt.p, e->df_cap, e->df_distinct,
                     sign > 0 ? 1 : -1);
  VR_HIP(hipGetLastError());
  return 0;
}

// ---- statistics -------------------------------------------------------------------------------

__global__ void df_lookup_kernel(const int32_t* __restrict__ ids, int n,
                                 const int32_t* __restrict__ keys, const int32_t* __restrict__ cnt,
                                 int64_t cap, int32_t* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = cap ? df_get(keys, cnt, cap, ids[i]) : 0;
}

int sparse_lookup_df(vr_engine* e, const int32_t* ids_host, int n, int32_t* out_df_host) {
  if (n <= 0) return 0;
  VR_TRY(e->stage_i32a.grow(n, 0, e->stream));
  VR_TRY(e->stage_i32b.grow(n, 0, e->stream));
  VR_HIP(hipMemcpyAsync(e->stage_i32a.p, ids_host, sizeof(int32_t) * static_cast<size_t>(n),
                        hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(df_lookup_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                     e->stream, e->stage_i32a.p, n, e->df_keys.p, e->df_cnt.p, e->df_cap,
                     e->stage_i32b.p);
  VR_HIP(hipGetLastError());
  VR_HIP(hipMemcpyAsync(out_df_host, e->stage_i32b.p, sizeof(int32_t) * static_cast<size_t>(n),
                        hipMemcpyDeviceToHost, e->stream));
  VR_HIP(hipStreamSynchronize(e->stream));
  return 0;
}

}  // namespace vr
