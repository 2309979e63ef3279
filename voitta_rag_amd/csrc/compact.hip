// vr_compact: rebuild the index without its tombstoned rows. The reference
// deletes points in place (watcher deletes, re-index = delete + insert, orphan purge: reference
// services/indexing.py:281-288,696-721,886-901, watcher.py:149-171) and Qdrant's optimiser reclaims
// the space in the background; here deletes are tombstones and this call is the reclaim step.
// Surviving rows keep their relative order, so ranking ties (lower row first) are unchanged; the
// document-frequency table and the sparse point count already exclude deleted rows (vr_delete_rows
// maintains them), so scores are bit-identical before and after.

#include "engine_internal.h"

#include <algorithm>
#include <numeric>

namespace vr {

namespace {

// one wave per surviving row: copy its D floats between the two tiled images
__global__ __launch_bounds__(256) void gather_dense_kernel(const float* __restrict__ src, const int32_t* __restrict__ old_of_new,
                                                           int64_t n_new, int dim, int kblocks, float* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const int64_t r_new = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (r_new >= n_new) return;
  const int64_t r_old = old_of_new[r_new];
  const int64_t to = r_old / kTileRows, tn = r_new / kTileRows;
  const int ro = static_cast<int>(r_old % kTileRows), rn = static_cast<int>(r_new % kTileRows);
  for (int k = lane; k < dim; k += 64) {
    const int kb = k / kTileK, kk = k % kTileK;
    const int64_t in_blk = static_cast<int64_t>(kb) * 256 + kk / 4;  // 256 floats per 1-KiB block
    dst[tn * kblocks * 256 + in_blk + tile_pos(kk % 4, rn) * 4] = src[to * kblocks * 256 + in_blk + tile_pos(kk % 4, ro) * 4];
  }
}

template <class T>
__global__ void gather_column_kernel(const T* __restrict__ src, const int32_t* __restrict__ old_of_new, int64_t n_new,
                                     T* __restrict__ dst) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n_new) dst[i] = src[old_of_new[i]];
}

// entries of a surviving row inside its old SELL slice
__global__ void count_entries_kernel(const int32_t* __restrict__ old_of_new, int64_t n_new,
                                     const int32_t* __restrict__ row_slice, const SliceDesc* __restrict__ slices,
                                     const int32_t* __restrict__ sidx, int32_t* __restrict__ cnt) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_new) return;
  const int32_t r = old_of_new[i];
  const int32_t s = row_slice[r];
  int c = 0;
  if (s >= 0) {
    const SliceDesc d = slices[s];
    const int lane = r - d.row_base;
    while (c < d.width && sidx[d.off + static_cast<int64_t>(c >> 2) * 256 + lane * 4 + (c & 3)] >= 0) ++c;
  }
  cnt[i] = c;
}

__global__ void extract_entries_kernel(const int32_t* __restrict__ old_of_new, int64_t n_new,
                                       const int32_t* __restrict__ row_slice, const SliceDesc* __restrict__ slices,
                                       const int32_t* __restrict__ sidx, const float* __restrict__ sval,
                                       const int64_t* __restrict__ begin, int32_t* __restrict__ idx,
                                       float* __restrict__ val, uint8_t* __restrict__ has_sparse) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_new) return;
  const int32_t r = old_of_new[i];
  const int32_t s = row_slice[r];
  has_sparse[i] = s >= 0;
  if (s < 0) return;
  const SliceDesc d = slices[s];
  const int lane = r - d.row_base;
  const int n = static_cast<int>(begin[i + 1] - begin[i]);
  for (int j = 0; j < n; ++j) {
    const int64_t at = d.off + static_cast<int64_t>(j >> 2) * 256 + lane * 4 + (j & 3);
    idx[begin[i] + j] = sidx[at];
    val[begin[i] + j] = sval[at];
  }
}

// rows that never had a sparse vector keep row_slice = -1 (deleting them must not touch the point count)
__global__ void restore_no_sparse_kernel(const uint8_t* __restrict__ has_sparse, int64_t n_new, int32_t* __restrict__ row_slice) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n_new && !has_sparse[i]) row_slice[i] = -1;
}

template <class T>
int gather_column(hipStream_t s, const DevArray<T>& src, DevArray<T>& dst, const int32_t* old_of_new, int64_t n_new) {
  if (n_new > 0)
    hipLaunchKernelGGL((gather_column_kernel<T>), dim3(static_cast<unsigned>((n_new + 255) / 256)), dim3(256), 0, s, src.p,
                       old_of_new, n_new, dst.p);
  return 0;
}

}  // namespace

// The compacted index is built BESIDE the live one, in a shadow engine that shares nothing with it but the
// document-frequency table (deleted rows never counted there, so it is the same before and after): searches keep
// running on the old tables the whole time. Only the final exchange of pointers and counts happens under the
// exclusive lock — the "compaction into a shadow + swap" of a store that is mutated while it serves (Qdrant's
// optimiser does the same with segments in the background). The caller holds e->wmu: no other writer exists.
int engine_compact(vr_engine* e, int64_t* new_row_of_old, int64_t* n_rows_after) {
  VR_HIP(hipStreamSynchronize(e->stream));
  const int64_t n_old = e->n_rows;
  std::vector<uint8_t> live(static_cast<size_t>(n_old));
  if (n_old) VR_HIP(hipMemcpy(live.data(), e->live.p, static_cast<size_t>(n_old), hipMemcpyDeviceToHost));
  std::vector<int32_t> old_of_new;
  old_of_new.reserve(static_cast<size_t>(e->n_live));
  for (int64_t r = 0; r < n_old; ++r) {
    const bool keep = live[static_cast<size_t>(r)] != 0;
    if (new_row_of_old) new_row_of_old[r] = keep ? static_cast<int64_t>(old_of_new.size()) : -1;
    if (keep) old_of_new.push_back(static_cast<int32_t>(r));
  }
  const int64_t n_new = static_cast<int64_t>(old_of_new.size());
  VR_CHECK(n_new == e->n_live, "live-row count %lld does not match the bitmap (%lld)", static_cast<long long>(e->n_live),
           static_cast<long long>(n_new));
  if (n_rows_after) *n_rows_after = n_new;
  if (n_new == n_old) return 0;  // nothing to reclaim
  hipStream_t s = e->stream;

  vr_engine t;  // the shadow
  t.device = e->device;
  t.dim = e->dim;
  t.kblocks = e->kblocks;
  t.prefilter = e->prefilter;
  t.prefilter8 = e->prefilter8;
  t.stream = s;  // (the master's stream: searches run on their lanes' streams)
  // temporaries of the rebuild (released by fail() on every path out)
  DevArray<int32_t> map, cnt, tmp_idx;
  DevArray<float> tmp_val;
  DevArray<int64_t> begin;
  DevArray<uint8_t> has_sparse;
  auto fail = [&](int rc) {
    (void)hipStreamSynchronize(s);
    map.release();
    cnt.release();
    tmp_idx.release();
    tmp_val.release();
    begin.release();
    has_sparse.release();
    t.corpus.release();
    t.corpus16.release();
    t.row_err.release();
    t.row_scale.release();
    t.centre.release();
    t.centre_sum.release();
    t.live.release();
    t.folder.release();
    t.index_folder.release();
    t.created.release();
    t.modified.release();
    t.row_slice.release();
    t.slices.release();
    t.sp_idx.release();
    t.sp_val.release();
    inv_release(&t);
    t.stage_i32a.release();
    t.stage_i32b.release();
    return rc;
  };
#define VR_CTRY(expr)                  \
  do {                                 \
    const int rc_ = (expr);            \
    if (rc_ != 0) return fail(rc_);    \
  } while (0)
  // HIP calls behind this point fail through fail() as well: a plain VR_HIP return would leak the shadow — a full-size copy
  // of the index (and the temporaries below, which own their memory and are released by their destructors or here)
#define VR_CHIP(call)                                                                                       \
  do {                                                                                                      \
    hipError_t err_ = (call);                                                                               \
    if (err_ != hipSuccess) {                                                                               \
      ::vr::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(err_), __FILE__, __LINE__);         \
      return fail(-1);                                                                                      \
    }                                                                                                       \
  } while (0)
  VR_CTRY(ensure_rows(&t, std::max<int64_t>(e->cap_rows, 1024)));  // same capacity: later appends find the room they had

  VR_CTRY(map.grow(std::max<int64_t>(n_new, 1), 0, s));
  if (n_new) VR_CHIP(hipMemcpyAsync(map.p, old_of_new.data(), sizeof(int32_t) * static_cast<size_t>(n_new), hipMemcpyHostToDevice, s));

  // ---- sparse rows out of the old slices, into a temporary CSR
  std::vector<int32_t> cnt_host(static_cast<size_t>(n_new));
  std::vector<int64_t> begin_host(static_cast<size_t>(n_new) + 1, 0);
  const bool any_sparse = !e->slices_host.empty() && n_new > 0;
  if (any_sparse) {
    const unsigned blocks = static_cast<unsigned>((n_new + 255) / 256);
    VR_CTRY(cnt.grow(n_new, 0, s));
    hipLaunchKernelGGL(count_entries_kernel, dim3(blocks), dim3(256), 0, s, map.p, n_new, e->row_slice.p, e->slices.p,
                       e->sp_idx.p, cnt.p);
    VR_CHIP(hipMemcpyAsync(cnt_host.data(), cnt.p, sizeof(int32_t) * static_cast<size_t>(n_new), hipMemcpyDeviceToHost, s));
    VR_CHIP(hipStreamSynchronize(s));
    for (int64_t i = 0; i < n_new; ++i) begin_host[static_cast<size_t>(i) + 1] = begin_host[static_cast<size_t>(i)] + cnt_host[static_cast<size_t>(i)];
    const int64_t nnz = begin_host.back();
    VR_CTRY(begin.grow(n_new + 1, 0, s));
    VR_CTRY(tmp_idx.grow(std::max<int64_t>(nnz, 1), 0, s));
    VR_CTRY(tmp_val.grow(std::max<int64_t>(nnz, 1), 0, s));
    VR_CTRY(has_sparse.grow(n_new, 0, s));
    VR_CHIP(hipMemcpyAsync(begin.p, begin_host.data(), sizeof(int64_t) * (static_cast<size_t>(n_new) + 1), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(extract_entries_kernel, dim3(blocks), dim3(256), 0, s, map.p, n_new, e->row_slice.p, e->slices.p,
                       e->sp_idx.p, e->sp_val.p, begin.p, tmp_idx.p, tmp_val.p, has_sparse.p);
  }

  // ---- dense image and payload columns, gathered into the shadow
  if (n_new)
    hipLaunchKernelGGL(gather_dense_kernel, dim3(static_cast<unsigned>((n_new + 3) / 4)), dim3(256), 0, s, e->corpus.p, map.p,
                       n_new, e->dim, e->kblocks, t.corpus.p);
  VR_CTRY(gather_column(s, e->folder, t.folder, map.p, n_new));
  VR_CTRY(gather_column(s, e->index_folder, t.index_folder, map.p, n_new));
  VR_CTRY(gather_column(s, e->created, t.created, map.p, n_new));
  VR_CTRY(gather_column(s, e->modified, t.modified, map.p, n_new));
  VR_CHIP(hipMemsetAsync(t.live.p, 1, static_cast<size_t>(n_new), s));  // (ensure_rows zeroed the rest)
  VR_CHIP(hipMemsetAsync(t.row_slice.p, 0xFF, sizeof(int32_t) * static_cast<size_t>(t.cap_rows), s));  // -1

  // ---- sparse index re-packed in the new row order; the df table and the point count carry over as they are
  t.n_rows = n_new;
  t.n_live = n_new;
  if (any_sparse) {
    VR_CTRY(sparse_append(&t, n_new, 0, cnt_host.data(), begin.p, nullptr, tmp_idx.p, tmp_val.p, /*account=*/false));
    hipLaunchKernelGGL(restore_no_sparse_kernel, dim3(static_cast<unsigned>((n_new + 255) / 256)), dim3(256), 0, s,
                       has_sparse.p, n_new, t.row_slice.p);
  }
  VR_CTRY(prefilter_recentre(&t));  // the shadow of the compacted rows, around THEIR column mean
  VR_CHIP(hipStreamSynchronize(s));
  VR_CHIP(hipGetLastError());
#undef VR_CTRY
#undef VR_CHIP

  {  // ---- the swap: the only moment searches wait for
    PublishLock publish(e);
    std::swap(e->corpus, t.corpus);
    std::swap(e->corpus16, t.corpus16);
    std::swap(e->row_err, t.row_err);
    std::swap(e->row_scale, t.row_scale);
    std::swap(e->centre, t.centre);
    std::swap(e->centre_norm, t.centre_norm);
    std::swap(e->centre_rows, t.centre_rows);
    std::swap(e->centre_checked_rows, t.centre_checked_rows);
    std::swap(e->live, t.live);
    std::swap(e->folder, t.folder);
    std::swap(e->index_folder, t.index_folder);
    std::swap(e->created, t.created);
    std::swap(e->modified, t.modified);
    std::swap(e->row_slice, t.row_slice);
    std::swap(e->slices, t.slices);
    std::swap(e->sp_idx, t.sp_idx);
    std::swap(e->sp_val, t.sp_val);
    e->slices_host.swap(t.slices_host);
    std::swap(e->inv_key, t.inv_key);
    std::swap(e->inv_val, t.inv_val);
    std::swap(e->inv_seg, t.inv_seg);
    std::swap(e->inv_used, t.inv_used);
    std::swap(e->n_inv_seg, t.n_inv_seg);
    std::swap(e->inv_slices, t.inv_slices);
    std::swap(e->inv_rows, t.inv_rows);
    e->n_slices_dev = t.n_slices_dev;
    e->sp_used = t.sp_used;
    e->cap_rows = t.cap_rows;
    e->n_rows = n_new;
    e->n_live = n_new;
    e->generation.fetch_add(1);
  }
  fail(0);  // what the shadow holds now is the old index: release it and the temporaries (no search can still be reading it)
  return 0;
}

}  // namespace vr
