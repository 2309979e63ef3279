// Search-time predicate over the numeric payload columns kept in HBM. Restates
// VectorStoreService._build_filter (reference: src/voitta/services/vector_store.py:462-530):
//   must     folder_path == folder_filter           (:476-482)
//   must     folder_path in include_folders         (:484-490)
//   must_not folder_path == each exclude_folders    (:492-499)
//   must_not index_folder == each exclude_index_folders (:501-508)
//   must     gte/lte range on source_modified_at (default) or source_created_at (:510-523);
//            a row lacking the field fails the range (SURVEY.md a14 [EXT]).
// Folder strings are dictionary ids (exact string equality in the reference == id equality),
// so the host folds the id lists into one pass/fail byte per dictionary id and the kernel is a
// single streaming pass: 1 + 4 + 4 (+ 8) bytes per row, HBM bound.

#include "engine_internal.h"

namespace vr {

__global__ void filter_mask_kernel(const uint8_t* __restrict__ live,
                                   const int32_t* __restrict__ folder,
                                   const int32_t* __restrict__ index_folder,
                                   const int64_t* __restrict__ ts, const uint8_t* __restrict__ pass_f,
                                   const uint8_t* __restrict__ pass_if, int has_lo, int has_hi,
                                   int64_t lo, int64_t hi, int64_t n, uint8_t* __restrict__ mask) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool ok = live[i] != 0;
  if (pass_f) ok = ok && pass_f[folder[i]];
  if (pass_if) ok = ok && pass_if[index_folder[i]];
  if (has_lo | has_hi) {
    int64_t t = ts[i];
    ok = ok && t != VR_TS_ABSENT;
    if (has_lo) ok = ok && t >= lo;
    if (has_hi) ok = ok && t <= hi;
  }
  mask[i] = ok ? 1 : 0;
}

int filter_build_mask(vr_engine* e, const vr_filter* f, const uint8_t** mask_out) {
  *mask_out = e->live.p;
  if (!f) return 0;
  VR_CHECK(f->struct_size == static_cast<int32_t>(sizeof(vr_filter)), "vr_filter size mismatch");
  const bool folder_active = f->n_must_folder_sets > 0 || f->n_not_folder > 0;
  const bool ifolder_active = f->n_not_index_folder > 0;
  const bool date_active = f->has_date_start || f->has_date_end;
  if (!folder_active && !ifolder_active && !date_active) return 0;
  if (e->n_rows == 0) return 0;
  VR_CHECK(f->n_must_folder_sets >= 0 && f->n_must_folder_sets <= 2, "at most 2 must-sets");

  const uint8_t* pf = nullptr;
  const uint8_t* pif = nullptr;
  if (folder_active) {
    const int64_t nf = static_cast<int64_t>(e->max_folder_id) + 1;
    std::vector<uint8_t> pass(static_cast<size_t>(nf), 1);
    for (int s = 0; s < f->n_must_folder_sets; ++s) {
      std::vector<uint8_t> in(static_cast<size_t>(nf), 0);
      for (int32_t j = f->must_folder_off[s]; j < f->must_folder_off[s + 1]; ++j) {
        int32_t id = f->must_folder_ids[j];
        if (id >= 0 && id < nf) in[static_cast<size_t>(id)] = 1;
      }
      for (int64_t i = 0; i < nf; ++i) pass[static_cast<size_t>(i)] &= in[static_cast<size_t>(i)];
    }
    for (int32_t j = 0; j < f->n_not_folder; ++j) {
      int32_t id = f->not_folder_ids[j];
      if (id >= 0 && id < nf) pass[static_cast<size_t>(id)] = 0;
    }
    VR_TRY(e->pass_folder.grow(nf, 0, e->stream));
    VR_HIP(hipMemcpyAsync(e->pass_folder.p, pass.data(), static_cast<size_t>(nf),
                          hipMemcpyHostToDevice, e->stream));
    VR_HIP(hipStreamSynchronize(e->stream));  // `pass` goes out of scope
    pf = e->pass_folder.p;
  }
  if (ifolder_active) {
    const int64_t nf = static_cast<int64_t>(e->max_index_folder_id) + 1;
    std::vector<uint8_t> pass(static_cast<size_t>(nf), 1);
    for (int32_t j = 0; j < f->n_not_index_folder; ++j) {
      int32_t id = f->not_index_folder_ids[j];
      if (id >= 0 && id < nf) pass[static_cast<size_t>(id)] = 0;
    }
    VR_TRY(e->pass_ifolder.grow(nf, 0, e->stream));
    VR_HIP(hipMemcpyAsync(e->pass_ifolder.p, pass.data(), static_cast<size_t>(nf),
                          hipMemcpyHostToDevice, e->stream));
    VR_HIP(hipStreamSynchronize(e->stream));
    pif = e->pass_ifolder.p;
  }
  VR_TRY(e->mask.grow(e->cap_rows, 0, e->stream));
  // rows past n_rows in the last 16-row tile must read as excluded
  {
    int64_t tail = e->cap_rows - e->n_rows;
    if (tail > 64) tail = 64;
    if (tail > 0)
      VR_HIP(hipMemsetAsync(e->mask.p + e->n_rows, 0, static_cast<size_t>(tail), e->stream));
  }
  const int64_t* ts = f->date_field == 1 ? e->created.p : e->modified.p;
  hipLaunchKernelGGL(filter_mask_kernel, dim3(static_cast<unsigned>((e->n_rows + 255) / 256)),
                     dim3(256), 0, e->stream, e->live.p, e->folder.p, e->index_folder.p, ts, pf, pif,
                     f->has_date_start, f->has_date_end, f->date_start, f->date_end, e->n_rows,
                     e->mask.p);
  VR_HIP(hipGetLastError());
  *mask_out = e->mask.p;
  return 0;
}

}  // namespace vr
