// extern "C" surface of libvoitta_engine.so (include/voitta_engine.h): engine lifecycle, the
// store (upsert / delete / count / read-back) and the three searches. Every entry point cites
// the reference call it stands in for in the header; this file only sequences kernels on the
// engine's stream and moves small results back to the host.

#include "engine_internal.h"

#include <algorithm>
#include <climits>
#include <thread>

#include "host_parallel.h"

namespace vr {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

__global__ void fill_i64_kernel(int64_t* p, int64_t v, int64_t n) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

int ensure_rows(vr_engine* e, int64_t need) {
  if (need <= e->cap_rows) return 0;
  int64_t ncap = e->cap_rows ? e->cap_rows : 1024;
  while (ncap < need) ncap *= 2;
  ncap = (ncap + 63) / 64 * 64;
  const int64_t keep_rows = e->n_rows;
  const int64_t keep_tiles = (keep_rows + kTileRows - 1) / kTileRows;
  VR_TRY(e->corpus.grow(ncap * e->dim, keep_tiles * kTileRows * e->dim, e->stream));
  if (e->prefilter) {
    const int per_u16 = e->prefilter8 ? 2 : 1;  // shadow elements per uint16 of storage
    VR_TRY(e->corpus16.grow(ncap * e->dim / per_u16, keep_tiles * kTileRows * e->dim / per_u16, e->stream));
    VR_TRY(e->row_err.grow(ncap, keep_tiles * kTileRows, e->stream));
    if (e->prefilter8) VR_TRY(e->row_scale.grow(ncap, keep_tiles * kTileRows, e->stream));
  }
  VR_TRY(e->live.grow(ncap, keep_rows, e->stream));
  VR_TRY(e->folder.grow(ncap, keep_rows, e->stream));
  VR_TRY(e->index_folder.grow(ncap, keep_rows, e->stream));
  VR_TRY(e->created.grow(ncap, keep_rows, e->stream));
  VR_TRY(e->modified.grow(ncap, keep_rows, e->stream));
  VR_TRY(e->row_slice.grow(ncap, keep_rows, e->stream));
  // DevArray rounds to powers of two; use the smallest capacity everywhere
  ncap = std::min({e->live.cap, e->folder.cap, e->index_folder.cap, e->created.cap,
                   e->modified.cap, e->row_slice.cap, e->corpus.cap / e->dim});
  ncap = ncap / 64 * 64;
  VR_HIP(hipMemsetAsync(e->live.p + keep_rows, 0, static_cast<size_t>(ncap - keep_rows), e->stream));
  e->cap_rows = ncap;
  return 0;
}

static int64_t decode_keys(const uint64_t* keys, int k, int64_t* rows, float* scores) {
  int64_t n = 0;
  for (int i = 0; i < k; ++i) {
    uint64_t key = keys[i];
    if (key == 0) {
      rows[i] = -1;
      scores[i] = 0.0f;
      continue;
    }
    uint32_t hi = static_cast<uint32_t>(key >> 32);
    uint32_t u = (hi & 0x80000000u) ? (hi ^ 0x80000000u) : ~hi;
    float s;
    memcpy(&s, &u, 4);
    rows[i] = static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(key & 0xFFFFFFFFu));
    scores[i] = s;
    ++n;
  }
  return n;
}

static int check_engine(vr_engine* e) {
  VR_CHECK(e != nullptr, "null engine");
  VR_HIP(hipSetDevice(e->device));
  return 0;
}

// A host query block goes into the pinned scratch and is read from there by query_image_kernel:
// no hipMemcpy on the latency path. Returns the pointer the kernels should read.
static const float* stage_query(vr_engine* e, const float* q, int nq, int mem) {
  if (mem == VR_MEM_DEVICE) return q;
  memcpy(pin_host<float>(e, kPinQuery), q, sizeof(float) * static_cast<size_t>(nq) * e->dim);
  return pin_dev<float>(e, kPinQuery);
}

// nq*k keys land in the pinned result area at kPinDenseKeys (readable after a stream sync).
// *two_stage is set when the f16 prefilter path ran: the caller must then check the candidate
// count at kPinCandCount after the sync and, if it overflowed, call again with allow_prefilter=false.
static int search_dense_block(vr_engine* e, const float* q_dev, int nq, int k, const uint8_t* mask,
                              bool allow_prefilter = true, bool* two_stage = nullptr) {
  if (two_stage) *two_stage = false;
  VR_TRY(dense_make_query_image(e, q_dev, nq));
  if (allow_prefilter && two_stage && prefilter_usable(e, nq, k)) {
    *two_stage = true;
    return prefilter_search(e, k, mask, pin_dev<uint64_t>(e, kPinDenseKeys), pin_dev<int32_t>(e, kPinCandCount));
  }
  // one or a few queries: scan and selection in one pass, results straight to pinned. A full
  // 16-query block offers 16x the candidates per tile; there the score array + select kernels win.
  if (k <= kFusedMaxK && nq <= 4)
    return dense_scan_topk(e, nq, k, mask, pin_dev<uint64_t>(e, kPinDenseKeys));
  const uint64_t* keys = nullptr;
  VR_TRY(dense_scores(e, nq, mask));
  VR_TRY(topk_select(e, e->scores.p, e->cap_rows, e->n_rows, nq, k, &keys));
  VR_HIP(hipMemcpyAsync(pin_host<uint64_t>(e, kPinDenseKeys), keys, sizeof(uint64_t) * static_cast<size_t>(nq) * k,
                        hipMemcpyDeviceToHost, e->stream));
  return 0;
}

// k keys land at kPinSparseKeys
static int search_sparse_block(vr_engine* e, const int32_t* q_idx, const float* q_val, int nnz, int k,
                               const uint8_t* mask, bool weights_given) {
  if (k <= kFusedMaxK)
    return sparse_scan_topk(e, q_idx, q_val, nnz, k, mask, weights_given, pin_dev<uint64_t>(e, kPinSparseKeys));
  const uint64_t* keys = nullptr;
  VR_TRY(sparse_scores(e, q_idx, q_val, nnz, mask, weights_given));
  VR_TRY(topk_select(e, e->sp_scores.p, e->cap_rows, e->n_rows, 1, k, &keys));
  VR_HIP(hipMemcpyAsync(pin_host<uint64_t>(e, kPinSparseKeys), keys, sizeof(uint64_t) * static_cast<size_t>(k),
                        hipMemcpyDeviceToHost, e->stream));
  return 0;
}


// ---- search lanes and the writer protocol (see vr_engine::rw) -----------------------------------------------

static void release_scratch(vr_engine* e) {
  e->upper.release();
  e->cand_rows.release();
  e->cand_keys.release();
  e->stage_dense.release();
  e->stage_len.release();
  e->stage_off.release();
  e->stage_idx.release();
  e->stage_val.release();
  e->stage_i32a.release();
  e->stage_i32b.release();
  e->stage_i64a.release();
  e->stage_i64b.release();
  e->stage_f64.release();
  e->bm_marks.release();
  e->bm_cnt.release();
  e->bm_idx.release();
  e->bm_val.release();
  e->enc_out.release();
  e->q_tiled.release();
  e->scores.release();
  e->sp_scores.release();
  e->mask.release();
  e->pass_folder.release();
  e->pass_ifolder.release();
  e->cand_a.release();
  e->cand_b.release();
  e->sp_cand.release();
  e->q_ids.release();
  e->q_w.release();
  e->bq_hat.release();
  e->bq_params.release();
  e->bq_best.release();
  e->bq_thr.release();
  e->bq_img.release();
  e->bq_cand.release();
  e->bq_cnt.release();
  e->bq_keys.release();
  e->bq_tile_ub.release();
  e->bq_pairs.release();
  e->bq_stage.release();
  e->sq_off.release();
  e->sq_ids.release();
  e->sq_val.release();
  e->sq_w.release();
  e->sq_keys.release();
  e->mg_in.release();
  e->mg_gid.release();
  e->mg_score.release();
  e->mg_cnt.release();
  if (e->pinned) (void)hipHostFree(e->pinned);
  e->pinned = nullptr;
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  if (e->ev_input) (void)hipEventDestroy(e->ev_input);
  if (e->aux_stream) (void)hipStreamDestroy(e->aux_stream);
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  e->ev_fork = e->ev_join = e->ev_input = nullptr;
  e->aux_stream = e->own_stream = nullptr;
}

// the index as the master holds it right now: pointers and counts only (call with rw held)
static void lane_view(vr_engine* L, const vr_engine* m) {
  L->n_rows = m->n_rows;
  L->n_live = m->n_live;
  L->cap_rows = m->cap_rows;
  L->corpus = m->corpus;
  L->corpus16 = m->corpus16;
  L->row_scale = m->row_scale;
  L->row_err = m->row_err;
  L->centre = m->centre;
  L->centre_norm = m->centre_norm;
  L->centre_rows = m->centre_rows;
  L->centre_checked_rows = m->centre_checked_rows;
  L->live = m->live;
  L->folder = m->folder;
  L->index_folder = m->index_folder;
  L->created = m->created;
  L->modified = m->modified;
  L->row_slice = m->row_slice;
  L->max_folder_id = m->max_folder_id;
  L->max_index_folder_id = m->max_index_folder_id;
  L->slices = m->slices;
  L->n_slices_dev = m->n_slices_dev;
  L->sp_idx = m->sp_idx;
  L->sp_val = m->sp_val;
  L->sp_used = m->sp_used;
  L->n_sparse_points = m->n_sparse_points;
  L->inv_key = m->inv_key;
  L->inv_val = m->inv_val;
  L->inv_seg = m->inv_seg;
  L->inv_used = m->inv_used;
  L->n_inv_seg = m->n_inv_seg;
  L->inv_slices = m->inv_slices;
  L->sp_has_dups = m->sp_has_dups;
  L->df_keys = m->df_keys;
  L->df_cnt = m->df_cnt;
  L->df_cap = m->df_cap;
  L->df_bound = m->df_bound;
  L->df_distinct = m->df_distinct;
  L->profiler = m->profiler;
}

static vr_engine* lane_create(vr_engine* m) {
  vr_engine* L = new vr_engine();
  L->master = m;
  L->device = m->device;
  L->dim = m->dim;
  L->kblocks = m->kblocks;
  L->prefilter = m->prefilter;
  L->prefilter8 = m->prefilter8;
  // The auxiliary stream (the sparse leg of a hybrid search, forked beside the dense scan) gets the highest stream
  // priority: streams of different priority never share a hardware queue. With equal priorities the runtime deals its
  // 4 hardware queues round-robin over ALL streams of the process, the two streams of a lane could land on one queue,
  // and the two legs then ran one after the other: hybrid p50 0.32 ms instead of 0.26 (scripts/perf_query_tail.py;
  // GPU_MAX_HW_QUEUES=8 in the environment had the same effect, but a library cannot rely on its host's environment).
  int prio_low = 0, prio_high = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
  bool ok = hipStreamCreateWithFlags(&L->own_stream, hipStreamNonBlocking) == hipSuccess &&
            hipStreamCreateWithPriority(&L->aux_stream, hipStreamNonBlocking, prio_high) == hipSuccess &&
            hipEventCreateWithFlags(&L->ev_fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&L->ev_join, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&L->ev_input, hipEventDisableTiming) == hipSuccess &&
            hipHostMalloc(&L->pinned, kPinnedBytes, hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer(&L->pinned_dev, L->pinned, 0) == hipSuccess;
  if (!ok) {
    set_error("creating a search lane failed");
    release_scratch(L);
    delete L;
    return nullptr;
  }
  L->stream = L->own_stream;
  L->pinned_bytes = kPinnedBytes;
  L->stat_last_candidates.store(-1);
  return L;
}

// A search: takes a lane (waits for one when all are busy), the shared lock, and a fresh view of the index.
// device_input: the caller's buffers were produced on the stream bound to the master (vr_set_stream); the lane's
// stream is ordered behind it.
struct SearchLane {
  vr_engine* m;
  vr_engine* L = nullptr;
  std::shared_lock<std::shared_mutex> lock;
  explicit SearchLane(vr_engine* master) : m(master) {}
  int acquire(bool device_input) {
    {
      std::unique_lock<std::mutex> g(m->lane_mu);
      while (m->lanes_free.empty() && static_cast<int>(m->lanes_all.size()) >= m->lanes_max) m->lane_cv.wait(g);
      if (!m->lanes_free.empty()) {
        L = m->lanes_free.back();
        m->lanes_free.pop_back();
      } else {
        L = lane_create(m);
        if (!L) return -1;
        m->lanes_all.push_back(L);
      }
    }
    while (m->writers_waiting.load(std::memory_order_acquire) > 0) std::this_thread::yield();
    lock = std::shared_lock<std::shared_mutex>(m->rw);
    lane_view(L, m);
    if (device_input) {
      if (hipEventRecord(L->ev_input, m->stream) != hipSuccess || hipStreamWaitEvent(L->stream, L->ev_input, 0) != hipSuccess) {
        set_error("ordering the search behind the caller's stream failed");
        return -1;
      }
    }
    return 0;
  }
  ~SearchLane() {
    if (!L) return;
    m->stat_two_stage += L->stat_two_stage.exchange(0);
    m->stat_fallback += L->stat_fallback.exchange(0);
    m->stat_batched += L->stat_batched.exchange(0);
    m->stat_batch_fallback += L->stat_batch_fallback.exchange(0);
    m->stat_batch_cands += L->stat_batch_cands.exchange(0);
    m->stat_sparse_grouped += L->stat_sparse_grouped.exchange(0);
    m->stat_sparse_group_redo += L->stat_sparse_group_redo.exchange(0);
    m->stat_sparse_group_cands += L->stat_sparse_group_cands.exchange(0);
    const int64_t lc = L->stat_last_candidates.exchange(-1);
    if (lc >= 0) m->stat_last_candidates.store(lc);
    if (lock.owns_lock()) lock.unlock();
    {
      std::lock_guard<std::mutex> g(m->lane_mu);
      m->lanes_free.push_back(L);
    }
    m->lane_cv.notify_one();
  }
};

}  // namespace vr

using namespace vr;

extern "C" {

int vr_abi_version(void) { return VR_ABI_VERSION; }

const char* vr_last_error(void) { return g_last_error.c_str(); }

int vr_engine_create(const vr_config* cfg, vr_engine** out) {
  VR_CHECK(cfg && out, "null argument");
  VR_CHECK(cfg->struct_size == static_cast<int32_t>(sizeof(vr_config)), "vr_config size mismatch");
  VR_CHECK(cfg->dim > 0 && cfg->dim % 16 == 0 && cfg->dim <= kMaxDim, "dim %d must be a multiple of 16 in 16..%d",
           cfg->dim, kMaxDim);
  int n_dev = 0;
  hipError_t err = hipGetDeviceCount(&n_dev);
  VR_CHECK(err == hipSuccess && n_dev > 0,
           "no HIP device available (%s): libvoitta_engine has no CPU fallback",
           err == hipSuccess ? "device count 0" : hipGetErrorString(err));
  VR_CHECK(cfg->device >= 0 && cfg->device < n_dev, "device %d out of range (0..%d)", cfg->device,
           n_dev - 1);
  hipDeviceProp_t prop;
  VR_HIP(hipGetDeviceProperties(&prop, cfg->device));
  VR_CHECK(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
           "device %d is %s; this library carries gfx950 (MI355X) code objects only", cfg->device,
           prop.gcnArchName);
  VR_HIP(hipSetDevice(cfg->device));
  vr_engine* e = new vr_engine();
  e->device = cfg->device;
  e->dim = cfg->dim;
  e->kblocks = cfg->dim / kTileK;
  e->prefilter = (cfg->dim % 32 == 0) && !(cfg->flags & VR_ENGINE_NO_PREFILTER);
  {
    // shadow format of the two-stage search: int8 + row scale where the MFMA tiling allows, else f16;
    // VR_PREFILTER=f16 keeps the f16 shadow (tighter bounds: fewer re-scores on corpora of near-duplicates)
    const char* mode = getenv("VR_PREFILTER");
    e->prefilter8 = e->prefilter && cfg->dim % 64 == 0 && !(mode && strcmp(mode, "f16") == 0);
  }
  // a blocking stream: it orders itself against the legacy null stream, so device buffers
  // produced by a framework on its default stream are safe to hand in without extra events
  if (hipStreamCreateWithFlags(&e->own_stream, hipStreamDefault) != hipSuccess) {
    set_error("hipStreamCreate failed");
    delete e;
    return -1;
  }
  e->stream = e->own_stream;
  if (hipStreamCreateWithFlags(&e->aux_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess) {
    set_error("creating the auxiliary stream failed");
    vr_engine_destroy(e);
    return -1;
  }
  if (hipHostMalloc(&e->pinned, kPinnedBytes, hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer(&e->pinned_dev, e->pinned, 0) != hipSuccess) {
    set_error("hipHostMalloc (mapped) failed");
    e->pinned = nullptr;
    vr_engine_destroy(e);
    return -1;
  }
  e->pinned_bytes = kPinnedBytes;
  if (const char* lanes = getenv("VR_SEARCH_LANES")) e->lanes_max = std::min(16, std::max(1, atoi(lanes)));
  int64_t init = cfg->initial_rows > 0 ? cfg->initial_rows : 1024;
  if (ensure_rows(e, init) != 0) {
    vr_engine_destroy(e);
    return -1;
  }
  *out = e;
  return 0;
}

void vr_engine_destroy(vr_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  encoder_release(e);
  prof_release(e);
  e->corpus.release();
  e->corpus16.release();
  e->row_err.release();
  e->row_scale.release();
  e->centre.release();
  e->centre_sum.release();
  for (vr_engine* L : e->lanes_all) {
    if (L->own_stream) (void)hipStreamSynchronize(L->own_stream);
    release_scratch(L);
    delete L;
  }
  e->lanes_all.clear();
  e->lanes_free.clear();
  e->live.release();
  e->folder.release();
  e->index_folder.release();
  e->created.release();
  e->modified.release();
  e->row_slice.release();
  e->slices.release();
  e->sp_idx.release();
  e->sp_val.release();
  inv_release(e);
  e->df_keys.release();
  e->df_cnt.release();
  if (e->df_distinct) (void)hipFree(e->df_distinct);
  release_scratch(e);
  delete e;
}

int vr_sync(vr_engine* e) {
  VR_TRY(check_engine(e));
  VR_HIP(hipStreamSynchronize(e->stream));
  return 0;
}

void* vr_stream(vr_engine* e) { return e ? static_cast<void*>(e->stream) : nullptr; }

int vr_set_stream(vr_engine* e, void* stream) {
  VR_TRY(check_engine(e));
  std::lock_guard<std::mutex> writer(e->wmu);
  VR_HIP(hipStreamSynchronize(e->stream));
  e->stream = stream ? static_cast<hipStream_t>(stream) : e->own_stream;
  return 0;
}

int vr_encoder_load(vr_engine* e, const vr_bert_desc* desc, const void* const* tensors,
                    int32_t n_tensors, int mem) {
  VR_TRY(check_engine(e));
  VR_CHECK(desc && tensors, "null argument");
  VR_CHECK(mem == VR_MEM_HOST || mem == VR_MEM_DEVICE, "bad mem %d", mem);
  std::lock_guard<std::mutex> writer(e->wmu);
  return encoder_load(e, desc, tensors, n_tensors, mem);
}

int vr_encode(vr_engine* e, const int32_t* ids, const int32_t* offsets, int32_t n_seq, int mem,
              float* out, int out_mem) {
  VR_TRY(check_engine(e));
  VR_CHECK(n_seq >= 0 && (n_seq == 0 || (ids && offsets && out)), "bad arguments");
  VR_CHECK((mem == VR_MEM_HOST || mem == VR_MEM_DEVICE) && (out_mem == VR_MEM_HOST || out_mem == VR_MEM_DEVICE),
           "bad mem");
  std::lock_guard<std::mutex> writer(e->wmu);
  return encoder_encode(e, ids, offsets, n_seq, mem, out, out_mem);
}

}  // extern "C"

// Body of vr_upsert, also the last stage of vr_index_batch. Caller holds e->wmu and the exclusive lock.
// sp_cnt_dev (device memory, mem == VR_MEM_DEVICE only): when given, the sparse rows are in the
// padded layout bm25_tf_kernel writes — row r = idx/val[sp_off[r] .. sp_off[r] + sp_cnt_dev[r]).
static int upsert_locked(vr_engine* e, int64_t n, int mem, const float* dense, const int64_t* sp_off,
                         const int32_t* sp_idx, const float* sp_val, const int32_t* sp_cnt_dev,
                         const int32_t* folder_id, const int32_t* index_folder_id,
                         const int64_t* created, const int64_t* modified, int64_t* out_first_row) {
  const int64_t first = e->n_rows;
  if (out_first_row) *out_first_row = first;
  if (n == 0) return 0;
  VR_CHECK(dense != nullptr, "null dense");
  VR_CHECK(first + n < (int64_t{1} << 32), "row id space exhausted");
  VR_TRY(ensure_rows(e, first + n));

  const float* x_dev = dense;
  if (mem == VR_MEM_HOST) {
    VR_TRY(e->stage_dense.grow(n * e->dim, 0, e->stream));
    VR_HIP(hipMemcpyAsync(e->stage_dense.p, dense, sizeof(float) * static_cast<size_t>(n * e->dim),
                          hipMemcpyHostToDevice, e->stream));
    x_dev = e->stage_dense.p;
  }
  VR_TRY(dense_store_rows(e, x_dev, n, first));
  // shadow of the rows just stored (no-op without prefilter); the int8 shadow is re-centred — all of it rebuilt —
  // whenever the collection has doubled since its centre was last computed
  e->n_rows = first + n;  // (prefilter_recentre works on [0, n_rows); the count is set again below)
  {
    // re-centre when the collection has doubled since a re-centring was last attempted (centre_checked_rows, not centre_rows:
    // with centring switched off or a non-finite centre the latter stays 0 and every upsert would rebuild the whole shadow)
    const int rc = (e->prefilter8 && first + n >= 1024 && first + n >= 2 * std::max(e->centre_rows, e->centre_checked_rows))
                       ? prefilter_recentre(e)
                       : prefilter_store_rows(e, n, first);
    e->n_rows = first;  // (also on failure: rows whose payload columns were never written must not become visible)
    if (rc != 0) return rc;
  }

  // payload columns are always host arrays (they come from Python metadata)
  if (folder_id) {
    for (int64_t i = 0; i < n; ++i) {
      VR_CHECK(folder_id[i] >= 0, "negative folder id");
      e->max_folder_id = std::max(e->max_folder_id, folder_id[i]);
    }
    VR_HIP(hipMemcpyAsync(e->folder.p + first, folder_id, sizeof(int32_t) * static_cast<size_t>(n),
                          hipMemcpyHostToDevice, e->stream));
  } else {
    e->max_folder_id = std::max(e->max_folder_id, 0);
    VR_HIP(hipMemsetAsync(e->folder.p + first, 0, sizeof(int32_t) * static_cast<size_t>(n), e->stream));
  }
  if (index_folder_id) {
    for (int64_t i = 0; i < n; ++i) {
      VR_CHECK(index_folder_id[i] >= 0, "negative index-folder id");
      e->max_index_folder_id = std::max(e->max_index_folder_id, index_folder_id[i]);
    }
    VR_HIP(hipMemcpyAsync(e->index_folder.p + first, index_folder_id,
                          sizeof(int32_t) * static_cast<size_t>(n), hipMemcpyHostToDevice, e->stream));
  } else {
    e->max_index_folder_id = std::max(e->max_index_folder_id, 0);
    VR_HIP(hipMemsetAsync(e->index_folder.p + first, 0, sizeof(int32_t) * static_cast<size_t>(n),
                          e->stream));
  }
  const unsigned fill_blocks = static_cast<unsigned>((n + 255) / 256);
  if (created)
    VR_HIP(hipMemcpyAsync(e->created.p + first, created, sizeof(int64_t) * static_cast<size_t>(n),
                          hipMemcpyHostToDevice, e->stream));
  else
    hipLaunchKernelGGL(fill_i64_kernel, dim3(fill_blocks), dim3(256), 0, e->stream,
                       e->created.p + first, static_cast<int64_t>(VR_TS_ABSENT), n);
  if (modified)
    VR_HIP(hipMemcpyAsync(e->modified.p + first, modified, sizeof(int64_t) * static_cast<size_t>(n),
                          hipMemcpyHostToDevice, e->stream));
  else
    hipLaunchKernelGGL(fill_i64_kernel, dim3(fill_blocks), dim3(256), 0, e->stream,
                       e->modified.p + first, static_cast<int64_t>(VR_TS_ABSENT), n);
  VR_HIP(hipMemsetAsync(e->live.p + first, 1, static_cast<size_t>(n), e->stream));
  VR_HIP(hipMemsetAsync(e->row_slice.p + first, 0xFF, sizeof(int32_t) * static_cast<size_t>(n),
                        e->stream));

  if (sp_off) {
    VR_CHECK(sp_idx && sp_val, "sparse offsets without indices/values");
    if (mem == VR_MEM_HOST) {
      const int64_t nnz = sp_off[n];
      VR_CHECK(sp_off[0] == 0 && nnz >= 0, "bad sparse offsets");
      // sort every row by token id (Qdrant sorts sparse vectors by index on ingestion [EXT])
      std::vector<int32_t> idx(sp_idx, sp_idx + nnz);
      std::vector<float> val(sp_val, sp_val + nnz);
      std::vector<std::pair<int32_t, float>> tmp;
      for (int64_t r = 0; r < n; ++r) {
        int64_t b = sp_off[r], en = sp_off[r + 1];
        VR_CHECK(en >= b, "sparse offsets must be non-decreasing");
        bool sorted = true;
        for (int64_t j = b + 1; j < en; ++j) sorted = sorted && idx[static_cast<size_t>(j - 1)] < idx[static_cast<size_t>(j)];
        for (int64_t j = b; j < en; ++j) VR_CHECK(idx[static_cast<size_t>(j)] >= 0, "negative token id");
        if (sorted) continue;
        tmp.clear();
        for (int64_t j = b; j < en; ++j) tmp.emplace_back(idx[static_cast<size_t>(j)], val[static_cast<size_t>(j)]);
        std::stable_sort(tmp.begin(), tmp.end(), [](const auto& a, const auto& c) { return a.first < c.first; });
        for (size_t j = 1; j < tmp.size(); ++j)
          if (tmp[j - 1].first == tmp[j].first) e->sp_has_dups = true;  // a term listed twice (invert.hip)
        for (int64_t j = b; j < en; ++j) {
          idx[static_cast<size_t>(j)] = tmp[static_cast<size_t>(j - b)].first;
          val[static_cast<size_t>(j)] = tmp[static_cast<size_t>(j - b)].second;
        }
      }
      VR_TRY(e->stage_off.grow(n + 1, 0, e->stream));
      VR_TRY(e->stage_idx.grow(std::max<int64_t>(nnz, 1), 0, e->stream));
      VR_TRY(e->stage_val.grow(std::max<int64_t>(nnz, 1), 0, e->stream));
      VR_HIP(hipMemcpyAsync(e->stage_off.p, sp_off, sizeof(int64_t) * static_cast<size_t>(n + 1),
                            hipMemcpyHostToDevice, e->stream));
      if (nnz > 0) {
        VR_HIP(hipMemcpyAsync(e->stage_idx.p, idx.data(), sizeof(int32_t) * static_cast<size_t>(nnz),
                              hipMemcpyHostToDevice, e->stream));
        VR_HIP(hipMemcpyAsync(e->stage_val.p, val.data(), sizeof(float) * static_cast<size_t>(nnz),
                              hipMemcpyHostToDevice, e->stream));
      }
      VR_HIP(hipStreamSynchronize(e->stream));  // idx/val vectors die at scope end
      std::vector<int32_t> cnt(static_cast<size_t>(n));
      for (int64_t r = 0; r < n; ++r) cnt[static_cast<size_t>(r)] = static_cast<int32_t>(sp_off[r + 1] - sp_off[r]);
      VR_TRY(sparse_append(e, n, first, cnt.data(), e->stage_off.p, nullptr, e->stage_idx.p, e->stage_val.p));
    } else if (sp_cnt_dev) {
      std::vector<int32_t> cnt(static_cast<size_t>(n));
      VR_HIP(hipMemcpyAsync(cnt.data(), sp_cnt_dev, sizeof(int32_t) * static_cast<size_t>(n),
                            hipMemcpyDeviceToHost, e->stream));
      VR_HIP(hipStreamSynchronize(e->stream));
      VR_TRY(sparse_append(e, n, first, cnt.data(), sp_off, sp_cnt_dev, sp_idx, sp_val));
    } else {
      std::vector<int64_t> off_host(static_cast<size_t>(n + 1));
      unsigned long long dups = 0;
      VR_TRY(inv_note_csr_dups(e, sp_off, sp_idx, n, &dups));
      VR_HIP(hipMemcpyAsync(off_host.data(), sp_off, sizeof(int64_t) * static_cast<size_t>(n + 1),
                            hipMemcpyDeviceToHost, e->stream));
      VR_HIP(hipStreamSynchronize(e->stream));
      if (dups) e->sp_has_dups = true;
      VR_CHECK(off_host[0] == 0, "sparse offsets must start at 0");
      std::vector<int32_t> cnt(static_cast<size_t>(n));
      for (int64_t r = 0; r < n; ++r) {
        VR_CHECK(off_host[static_cast<size_t>(r) + 1] >= off_host[static_cast<size_t>(r)], "sparse offsets must be non-decreasing");
        cnt[static_cast<size_t>(r)] = static_cast<int32_t>(off_host[static_cast<size_t>(r) + 1] - off_host[static_cast<size_t>(r)]);
      }
      VR_TRY(sparse_append(e, n, first, cnt.data(), sp_off, nullptr, sp_idx, sp_val));
    }
  }
  VR_HIP(hipGetLastError());
  // caller-owned host arrays must not be read after we return
  if (mem == VR_MEM_HOST || folder_id || index_folder_id || created || modified)
    VR_HIP(hipStreamSynchronize(e->stream));
  e->n_rows += n;
  e->n_live += n;
  return 0;
}

extern "C" {

int vr_upsert(vr_engine* e, int64_t n, int mem, const float* dense, const int64_t* sp_off,
              const int32_t* sp_idx, const float* sp_val, const int32_t* folder_id,
              const int32_t* index_folder_id, const int64_t* created, const int64_t* modified,
              int64_t* out_first_row) {
  VR_TRY(check_engine(e));
  VR_CHECK(n >= 0, "negative row count");
  VR_CHECK(mem == VR_MEM_HOST || mem == VR_MEM_DEVICE, "bad mem %d", mem);
  std::lock_guard<std::mutex> writer(e->wmu);
  PublishLock publish(e);  // (the append itself: a fraction of a millisecond per thousand rows)
  return upsert_locked(e, n, mem, dense, sp_off, sp_idx, sp_val, nullptr, folder_id, index_folder_id,
                       created, modified, out_first_row);
}

int vr_index_batch(vr_engine* e, int64_t n, int mem, const int32_t* wp_ids, const int32_t* wp_off,
                   const int32_t* bm_ids, const int64_t* bm_off, double k, double b, double avg_len,
                   const int32_t* folder_id, const int32_t* index_folder_id, const int64_t* created,
                   const int64_t* modified, int64_t* out_first_row) {
  VR_TRY(check_engine(e));
  VR_CHECK(n >= 0 && (n == 0 || (wp_ids && wp_off)), "bad arguments");
  VR_CHECK(mem == VR_MEM_HOST || mem == VR_MEM_DEVICE, "bad mem %d", mem);
  VR_CHECK((bm_ids == nullptr) == (bm_off == nullptr), "bm_ids and bm_off go together");
  std::lock_guard<std::mutex> writer(e->wmu);
  if (n == 0) {
    if (out_first_row) *out_first_row = e->n_rows;
    return 0;
  }
  // 1. sparse side first: it is microseconds of work and its per-document counts are the only
  //    thing the host has to wait for (slice widths); the long encode is queued behind it.
  const int64_t* bm_off_dev = nullptr;
  if (bm_off) {
    int64_t n_tokens = 0;
    const int32_t* bm_ids_dev = bm_ids;
    bm_off_dev = bm_off;
    if (mem == VR_MEM_HOST) {
      VR_CHECK(bm_off[0] == 0, "token offsets must start at 0");
      n_tokens = bm_off[n];
      VR_TRY(e->stage_off.grow(n + 1, 0, e->stream));
      VR_TRY(e->stage_idx.grow(std::max<int64_t>(n_tokens, 1), 0, e->stream));
      VR_HIP(hipMemcpyAsync(e->stage_off.p, bm_off, sizeof(int64_t) * static_cast<size_t>(n + 1),
                            hipMemcpyHostToDevice, e->stream));
      if (n_tokens > 0)
        VR_HIP(hipMemcpyAsync(e->stage_idx.p, bm_ids, sizeof(int32_t) * static_cast<size_t>(n_tokens),
                              hipMemcpyHostToDevice, e->stream));
      bm_off_dev = e->stage_off.p;
      bm_ids_dev = e->stage_idx.p;
    } else {
      VR_HIP(hipMemcpyAsync(&n_tokens, bm_off + n, sizeof(int64_t), hipMemcpyDeviceToHost, e->stream));
      VR_HIP(hipStreamSynchronize(e->stream));
    }
    const int64_t cap = std::max<int64_t>(n_tokens, 1);
    VR_TRY(e->bm_cnt.grow(n, 0, e->stream));
    VR_TRY(e->bm_idx.grow(cap, 0, e->stream));
    VR_TRY(e->bm_val.grow(cap, 0, e->stream));
    VR_TRY(bm25_tf(e, bm_off_dev, bm_ids_dev, n, n_tokens, k, b, avg_len, e->bm_cnt.p, e->bm_idx.p,
                   nullptr, e->bm_val.p));
  }
  // 2. dense encode into engine-owned rows
  VR_CHECK(encoder_hidden(e) == e->dim, "encoder width %d != store dimension %d", encoder_hidden(e), e->dim);
  VR_TRY(e->enc_out.grow(n * e->dim, 0, e->stream));
  VR_TRY(encoder_encode(e, wp_ids, wp_off, static_cast<int>(n), mem, e->enc_out.p, VR_MEM_DEVICE));
  // 3. store: the only part searches wait for (they ran beside the encode)
  PublishLock publish(e);
  return upsert_locked(e, n, VR_MEM_DEVICE, e->enc_out.p, bm_off_dev, bm_off ? e->bm_idx.p : nullptr,
                       bm_off ? e->bm_val.p : nullptr, bm_off ? e->bm_cnt.p : nullptr, folder_id,
                       index_folder_id, created, modified, out_first_row);
}

int vr_delete_rows(vr_engine* e, const int64_t* rows, int64_t n) {
  VR_TRY(check_engine(e));
  VR_CHECK(n >= 0 && (n == 0 || rows), "bad arguments");
  if (n == 0) return 0;
  std::lock_guard<std::mutex> writer(e->wmu);
  PublishLock publish(e);
  std::vector<int64_t> uniq(rows, rows + n);
  std::sort(uniq.begin(), uniq.end());
  uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
  const int64_t m = static_cast<int64_t>(uniq.size());
  VR_TRY(e->stage_i64a.grow(m, 0, e->stream));
  VR_HIP(hipMemcpyAsync(e->stage_i64a.p, uniq.data(), sizeof(int64_t) * static_cast<size_t>(m),
                        hipMemcpyHostToDevice, e->stream));
  int64_t deleted = 0, sparse_deleted = 0;
  VR_TRY(sparse_delete_rows(e, e->stage_i64a.p, m, &deleted, &sparse_deleted));
  e->n_live -= deleted;
  e->n_sparse_points -= sparse_deleted;
  return 0;
}

int vr_stats(vr_engine* e, int32_t which, int64_t* out) {
  VR_CHECK(e != nullptr && out != nullptr, "null argument");
  switch (which) {
    case VR_STAT_TWO_STAGE: *out = e->stat_two_stage.load(); break;
    case VR_STAT_FALLBACK: *out = e->stat_fallback.load(); break;
    case VR_STAT_LAST_CANDIDATES: *out = e->stat_last_candidates.load(); break;
    case VR_STAT_BATCHED: *out = e->stat_batched.load(); break;
    case VR_STAT_BATCH_FALLBACK: *out = e->stat_batch_fallback.load(); break;
    case VR_STAT_BATCH_CANDIDATES: *out = e->stat_batch_cands.load(); break;
    case VR_STAT_GENERATION: *out = e->generation.load(); break;
    case VR_STAT_SPARSE_GROUPED: *out = e->stat_sparse_grouped.load(); break;
    case VR_STAT_SPARSE_GROUP_REDO: *out = e->stat_sparse_group_redo.load(); break;
    case VR_STAT_SPARSE_GROUP_CANDIDATES: *out = e->stat_sparse_group_cands.load(); break;
    default: set_error("unknown statistic %d", which); return -1;
  }
  return 0;
}

int vr_count(vr_engine* e, int64_t* n_rows, int64_t* n_live) {
  VR_CHECK(e != nullptr, "null engine");
  std::shared_lock<std::shared_mutex> view(e->rw);
  if (n_rows) *n_rows = e->n_rows;
  if (n_live) *n_live = e->n_live;
  return 0;
}

int vr_get_dense(vr_engine* e, const int64_t* rows, int64_t n, float* out) {
  VR_TRY(check_engine(e));
  VR_CHECK(n >= 0 && (n == 0 || (rows && out)), "bad arguments");
  if (n == 0) return 0;
  std::lock_guard<std::mutex> writer(e->wmu);  // (the master's staging arrays)
  std::shared_lock<std::shared_mutex> view(e->rw);
  for (int64_t i = 0; i < n; ++i)
    VR_CHECK(rows[i] >= 0 && rows[i] < e->n_rows, "row %lld out of range", static_cast<long long>(rows[i]));
  VR_TRY(e->stage_i64a.grow(n, 0, e->stream));
  VR_TRY(e->stage_dense.grow(n * e->dim, 0, e->stream));
  VR_HIP(hipMemcpyAsync(e->stage_i64a.p, rows, sizeof(int64_t) * static_cast<size_t>(n),
                        hipMemcpyHostToDevice, e->stream));
  VR_TRY(dense_read_rows(e, e->stage_i64a.p, n, e->stage_dense.p));
  VR_HIP(hipMemcpyAsync(out, e->stage_dense.p, sizeof(float) * static_cast<size_t>(n * e->dim),
                        hipMemcpyDeviceToHost, e->stream));
  VR_HIP(hipStreamSynchronize(e->stream));
  return 0;
}

int vr_sparse_stats(vr_engine* e, const int32_t* ids, int32_t n, int32_t* out_df, int64_t* out_n_points) {
  VR_TRY(check_engine(e));
  std::lock_guard<std::mutex> writer(e->wmu);
  std::shared_lock<std::shared_mutex> view(e->rw);
  if (out_n_points) *out_n_points = e->n_sparse_points;
  if (n > 0) {
    VR_CHECK(ids && out_df, "null argument");
    if (e->df_cap == 0) {
      for (int i = 0; i < n; ++i) out_df[i] = 0;
    } else {
      VR_TRY(sparse_lookup_df(e, ids, n, out_df));
    }
  }
  return 0;
}

}  // extern "C"

// Both legs of ONE hybrid query on the latency path (`e` is a search lane, n_rows > 0): the k dense keys end up in the
// pinned result area at kPinDenseKeys, the k sparse keys (when *have_sparse) at kPinSparseKeys; returns with the stream
// drained. The two legs share nothing but the mask: the (small, latency-bound) sparse leg is forked onto the auxiliary
// stream and runs under the dense scan. The k > kFusedMaxK sparse path borrows the dense leg's selection buffers and
// stays on the main stream.
// The sparse leg of a hybrid query queued on the lane's auxiliary stream right away (behind the mask on the main
// stream): what vr_query_text can do BEFORE the question's forward pass (VR_QUERY_TEXT_AHEAD=1; measured slower than
// running it beside the dense scan, see there). hybrid_one_query(..., sparse_in_flight = true) joins it.
static int hybrid_sparse_ahead(vr_engine* e, const int32_t* q_idx, const float* q_val, int nnz, int k, const uint8_t* mask,
                               bool weights_given) {
  VR_CHECK(q_idx && q_val && nnz > 0 && k <= kFusedMaxK && e->n_slices_dev > 0, "no sparse leg to start ahead");
  VR_HIP(hipEventRecord(e->ev_fork, e->stream));  // after the mask
  VR_HIP(hipStreamWaitEvent(e->aux_stream, e->ev_fork, 0));
  hipStream_t main_stream = e->stream;
  e->stream = e->aux_stream;
  const int rc = search_sparse_block(e, q_idx, q_val, nnz, k, mask, weights_given);
  e->stream = main_stream;
  if (rc != 0) return rc;
  VR_HIP(hipEventRecord(e->ev_join, e->aux_stream));
  return 0;
}

static int hybrid_one_query(vr_engine* e, const float* q, int mem, const int32_t* q_idx, const float* q_val, int nnz, int k,
                            bool weights_given, const uint8_t* mask, bool* have_sparse_out, bool sparse_in_flight = false) {
  const float* q_dev = stage_query(e, q, 1, mem);
  bool two_stage = false;
  const bool have_sparse = nnz > 0 && e->n_slices_dev > 0;
  *have_sparse_out = have_sparse;
  const bool fork = have_sparse && k <= kFusedMaxK && !sparse_in_flight;
  if (have_sparse) VR_CHECK(q_idx && q_val, "null sparse query");
  if (fork) VR_HIP(hipEventRecord(e->ev_fork, e->stream));  // after the mask, before the dense leg
  VR_TRY(search_dense_block(e, q_dev, 1, k, mask, true, &two_stage));
  if (sparse_in_flight) {
    VR_HIP(hipStreamWaitEvent(e->stream, e->ev_join, 0));
  } else if (fork) {
    // queued after the dense leg (whose scan is already running by now), executed beside it
    VR_HIP(hipStreamWaitEvent(e->aux_stream, e->ev_fork, 0));
    hipStream_t main_stream = e->stream;
    e->stream = e->aux_stream;
    const int rc = search_sparse_block(e, q_idx, q_val, nnz, k, mask, weights_given);
    e->stream = main_stream;
    if (rc != 0) return rc;
    VR_HIP(hipEventRecord(e->ev_join, e->aux_stream));
    VR_HIP(hipStreamWaitEvent(e->stream, e->ev_join, 0));
  } else if (have_sparse) {
    VR_TRY(search_sparse_block(e, q_idx, q_val, nnz, k, mask, weights_given));
  }
  VR_HIP(hipStreamSynchronize(e->stream));
  e->stat_two_stage += two_stage;
  if (two_stage) e->stat_last_candidates.store(*pin_host<int32_t>(e, kPinCandCount));
  if (two_stage && *pin_host<int32_t>(e, kPinCandCount) > kMaxCandidates) {
    ++e->stat_fallback;
    VR_TRY(search_dense_block(e, q_dev, 1, k, mask, false));  // candidate overflow: one-stage exact scan
    VR_HIP(hipStreamSynchronize(e->stream));
  }
  return 0;
}

// Dense search of nq queries; the nq x k ranking keys ((order-preserving f32 score bits << 32) | ~row, descending,
// 0 = none) go to keys_host (host array) and/or keys_dev (device array). `e` is a search lane (SearchLane).
static int search_dense_keys_locked(vr_engine* e, const float* q, int nq, int mem, int k, const vr_filter* filter,
                                    uint64_t* keys_host, uint64_t* keys_dev, const uint8_t* mask_in = nullptr) {
  const size_t row_bytes = sizeof(uint64_t) * static_cast<size_t>(k);
  if (e->n_rows == 0) {
    if (keys_host) memset(keys_host, 0, row_bytes * static_cast<size_t>(nq));
    if (keys_dev) VR_HIP(hipMemsetAsync(keys_dev, 0, row_bytes * static_cast<size_t>(nq), e->stream));
    return 0;
  }
  const uint8_t* mask = mask_in;
  if (!mask) VR_TRY(filter_build_mask(e, filter, &mask));
  const uint64_t* pinned_keys = pin_host<uint64_t>(e, kPinDenseKeys);
  // one block of <= 16 queries through the one-/two-stage scans; its keys (in the pinned result area) go to slot `at`
  auto run_block = [&](const float* qsrc, int nb, int at) -> int {
    const float* q_dev = stage_query(e, qsrc, nb, mem);
    bool two_stage = false;
    VR_TRY(search_dense_block(e, q_dev, nb, k, mask, true, &two_stage));
    VR_HIP(hipStreamSynchronize(e->stream));
    e->stat_two_stage += two_stage;
    if (two_stage) e->stat_last_candidates.store(*pin_host<int32_t>(e, kPinCandCount));
    if (two_stage && *pin_host<int32_t>(e, kPinCandCount) > kMaxCandidates) {
      // more candidates than the re-score budget (near-duplicate corpus): one-stage exact scan
      ++e->stat_fallback;
      VR_TRY(search_dense_block(e, q_dev, nb, k, mask, false));
      VR_HIP(hipStreamSynchronize(e->stream));
    }
    if (keys_host) memcpy(keys_host + static_cast<size_t>(at) * k, pinned_keys, row_bytes * static_cast<size_t>(nb));
    if (keys_dev) {
      VR_HIP(hipMemcpyAsync(keys_dev + static_cast<size_t>(at) * k, pinned_keys, row_bytes * static_cast<size_t>(nb),
                            hipMemcpyHostToDevice, e->stream));
      VR_HIP(hipStreamSynchronize(e->stream));  // the pinned area is reused by the next block
    }
    return 0;
  };
  if (batch_usable(e, nq, k)) {
    // many queries at once: integer GEMM over the int8 shadow + exact re-score (batch.hip), 1024 queries per round
    constexpr int kRound = 1024;
    std::vector<int32_t> over(static_cast<size_t>(std::min(nq, kRound)));
    for (int q0 = 0; q0 < nq; q0 += kRound) {
      const int nb = std::min(kRound, nq - q0);
      const float* q_dev = q + static_cast<int64_t>(q0) * e->dim;
      if (mem == VR_MEM_HOST) {
        VR_TRY(e->bq_stage.grow(static_cast<int64_t>(nb) * e->dim, 0, e->stream));
        VR_HIP(hipMemcpyAsync(e->bq_stage.p, q_dev, sizeof(float) * static_cast<size_t>(nb) * e->dim, hipMemcpyHostToDevice,
                              e->stream));
        q_dev = e->bq_stage.p;
      }
      const uint64_t* round_keys = nullptr;
      const int32_t* over_dev = nullptr;
      VR_TRY(batch_search(e, q_dev, nb, k, mask, &round_keys, &over_dev));
      if (keys_host)
        VR_HIP(hipMemcpyAsync(keys_host + static_cast<size_t>(q0) * k, round_keys, row_bytes * static_cast<size_t>(nb),
                              hipMemcpyDeviceToHost, e->stream));
      if (keys_dev)
        VR_HIP(hipMemcpyAsync(keys_dev + static_cast<size_t>(q0) * k, round_keys, row_bytes * static_cast<size_t>(nb),
                              hipMemcpyDeviceToDevice, e->stream));
      VR_HIP(hipMemcpyAsync(over.data(), over_dev, sizeof(int32_t) * static_cast<size_t>(nb), hipMemcpyDeviceToHost, e->stream));
      VR_HIP(hipStreamSynchronize(e->stream));
      e->stat_batched += nb;
      for (int i = 0; i < nb; ++i) e->stat_batch_cands += std::min<int32_t>(over[static_cast<size_t>(i)], kBatchCand);
      for (int i = 0; i < nb; ++i)
        if (over[static_cast<size_t>(i)] > kBatchCand) {  // candidate budget exceeded: this query alone, through the exact scans
          ++e->stat_batch_fallback;
          VR_TRY(run_block(q + static_cast<int64_t>(q0 + i) * e->dim, 1, q0 + i));
        }
    }
    return 0;
  }
  for (int q0 = 0; q0 < nq; q0 += kQueryBlock)
    VR_TRY(run_block(q + static_cast<int64_t>(q0) * e->dim, std::min(kQueryBlock, nq - q0), q0));
  return 0;
}

extern "C" {

int vr_search_dense(vr_engine* e, const float* q, int32_t nq, int mem, int32_t k,
                    const vr_filter* filter, int64_t* rows, float* scores, int32_t* counts) {
  VR_TRY(check_engine(e));
  VR_CHECK(q && rows && scores && nq >= 1, "bad arguments");
  VR_CHECK(k >= 1 && k <= kMaxK, "k = %d not in 1..%d", k, kMaxK);
  VR_CHECK(mem == VR_MEM_HOST || mem == VR_MEM_DEVICE, "bad mem %d", mem);
  SearchLane lane(e);
  VR_TRY(lane.acquire(mem == VR_MEM_DEVICE));
  std::vector<uint64_t> keys(static_cast<size_t>(nq) * k);
  VR_TRY(search_dense_keys_locked(lane.L, q, nq, mem, k, filter, keys.data(), nullptr));
  for (int i = 0; i < nq; ++i) {
    const int64_t c = decode_keys(keys.data() + static_cast<size_t>(i) * k, k, rows + static_cast<int64_t>(i) * k,
                                  scores + static_cast<int64_t>(i) * k);
    if (counts) counts[i] = static_cast<int32_t>(c);
  }
  return 0;
}

int vr_search_dense_keys(vr_engine* e, const float* q, int32_t nq, int mem, int32_t k, const vr_filter* filter,
                         uint64_t* keys, int keys_mem) {
  VR_TRY(check_engine(e));
  VR_CHECK(q && keys && nq >= 1, "bad arguments");
  VR_CHECK(k >= 1 && k <= kMaxK, "k = %d not in 1..%d", k, kMaxK);
  VR_CHECK((mem == VR_MEM_HOST || mem == VR_MEM_DEVICE) && (keys_mem == VR_MEM_HOST || keys_mem == VR_MEM_DEVICE), "bad mem");
  SearchLane lane(e);
  VR_TRY(lane.acquire(true));  // (device output: ordered behind the caller's stream either way)
  VR_TRY(search_dense_keys_locked(lane.L, q, nq, mem, k, filter, keys_mem == VR_MEM_HOST ? keys : nullptr,
                                  keys_mem == VR_MEM_DEVICE ? keys : nullptr));
  if (keys_mem == VR_MEM_DEVICE) VR_HIP(hipStreamSynchronize(lane.L->stream));
  return 0;
}

int vr_search_sparse(vr_engine* e, const int32_t* q_idx, const float* q_val, int32_t nnz, int32_t k,
                     int32_t weights_given, const vr_filter* filter, int64_t* rows, float* scores,
                     int32_t* count) {
  VR_TRY(check_engine(e));
  VR_CHECK(rows && scores && count, "bad arguments");
  VR_CHECK(k >= 1 && k <= kMaxK, "k = %d not in 1..%d", k, kMaxK);
  SearchLane lane(e);
  VR_TRY(lane.acquire(false));
  vr_engine* m = e;
  e = lane.L;  // everything below runs on the lane: its stream, its scratch, its view of the index
  (void)m;
  *count = 0;
  for (int i = 0; i < k; ++i) {
    rows[i] = -1;
    scores[i] = 0.0f;
  }
  if (e->n_rows == 0 || e->n_slices_dev == 0 || nnz <= 0) return 0;
  VR_CHECK(q_idx && q_val, "null sparse query");
  const uint8_t* mask = nullptr;
  VR_TRY(filter_build_mask(e, filter, &mask));
  const uint64_t* host_keys = pin_host<uint64_t>(e, kPinSparseKeys);
  VR_TRY(search_sparse_block(e, q_idx, q_val, nnz, k, mask, weights_given != 0));
  VR_HIP(hipStreamSynchronize(e->stream));
  *count = static_cast<int32_t>(decode_keys(host_keys, k, rows, scores));
  return 0;
}

int vr_search_hybrid(vr_engine* e, const float* q, int mem, const int32_t* q_idx, const float* q_val,
                     int32_t nnz, int32_t limit, double sparse_weight, int32_t fusion,
                     const vr_filter* filter, int64_t* out_rows, double* out_scores,
                     int32_t* out_from_dense, int32_t* out_count) {
  VR_TRY(check_engine(e));
  VR_CHECK(q && out_rows && out_scores && out_count, "bad arguments");
  VR_CHECK(limit >= 1 && limit * 3 <= kMaxK, "limit = %d not in 1..%d", limit, kMaxK / 3);
  VR_CHECK(fusion == VR_FUSION_MINMAX || fusion == VR_FUSION_RRF, "unknown fusion %d", fusion);
  VR_CHECK(mem == VR_MEM_HOST || mem == VR_MEM_DEVICE, "bad mem %d", mem);
  SearchLane lane(e);
  VR_TRY(lane.acquire(mem == VR_MEM_DEVICE));
  e = lane.L;  // everything below runs on the lane: its stream, its scratch, its view of the index
  *out_count = 0;
  if (e->n_rows == 0) return 0;
  const int k = limit * 3;  // prefetch_limit, vector_store.py:636
  const uint8_t* mask = nullptr;
  VR_TRY(filter_build_mask(e, filter, &mask));
  bool have_sparse = false;
  VR_TRY(hybrid_one_query(e, q, mem, q_idx, q_val, nnz, k, false, mask, &have_sparse));
  int64_t d_rows[kMaxK], s_rows[kMaxK];
  float d_scores[kMaxK], s_scores[kMaxK];
  int nd = static_cast<int>(decode_keys(pin_host<uint64_t>(e, kPinDenseKeys), k, d_rows, d_scores));
  int ns = have_sparse ? static_cast<int>(decode_keys(pin_host<uint64_t>(e, kPinSparseKeys), k, s_rows, s_scores)) : 0;
  if (fusion == VR_FUSION_MINMAX)
    return fuse_minmax(d_rows, d_scores, nd, s_rows, s_scores, ns, limit, sparse_weight, 1, out_rows,
                       out_scores, out_from_dense, out_count);
  return fuse_rrf(d_rows, nd, s_rows, ns, limit, sparse_weight, out_rows, out_scores, out_from_dense,
                  out_count);
}

int vr_compact(vr_engine* e, int64_t* new_row_of_old, int64_t* n_rows_after) {
  VR_TRY(check_engine(e));
  std::lock_guard<std::mutex> writer(e->wmu);  // (engine_compact takes the exclusive lock itself, for the swap only)
  return engine_compact(e, new_row_of_old, n_rows_after);
}

int vr_save(vr_engine* e, const char* path) {
  VR_TRY(check_engine(e));
  VR_CHECK(path && *path, "null path");
  std::lock_guard<std::mutex> writer(e->wmu);
  std::shared_lock<std::shared_mutex> view(e->rw);
  return engine_save(e, path);
}

int vr_load(vr_engine* e, const char* path) {
  VR_TRY(check_engine(e));
  VR_CHECK(path && *path, "null path");
  std::lock_guard<std::mutex> writer(e->wmu);
  PublishLock publish(e);
  const int rc = engine_load(e, path);
  if (rc == 0) e->generation.fetch_add(1);
  return rc;
}

int vr_fuse_minmax(const int64_t* d_rows, const float* d_scores, int32_t nd, const int64_t* s_rows,
                   const float* s_scores, int32_t ns, int32_t limit, double sparse_weight,
                   int32_t json_scores, int64_t* out_rows, double* out_scores,
                   int32_t* out_from_dense, int32_t* out_count) {
  VR_CHECK(nd >= 0 && ns >= 0 && out_rows && out_scores && out_count, "bad arguments");
  return fuse_minmax(d_rows, d_scores, nd, s_rows, s_scores, ns, limit, sparse_weight, json_scores,
                     out_rows, out_scores, out_from_dense, out_count);
}

int vr_fuse_rrf(const int64_t* d_rows, int32_t nd, const int64_t* s_rows, int32_t ns, int32_t limit,
                int64_t* out_rows, double* out_scores, int32_t* out_from_dense, int32_t* out_count) {
  VR_CHECK(nd >= 0 && ns >= 0 && out_rows && out_scores && out_count, "bad arguments");
  return fuse_rrf(d_rows, nd, s_rows, ns, limit, 0.0, out_rows, out_scores, out_from_dense, out_count);
}

}  // extern "C"

// ---- many sparse / hybrid queries per call (BASELINE configs[4]: 1k batched hybrid queries) ------------------------

namespace {

// The sparse queries of a batch as the engine wants them: per query the terms in ascending id order, a repeated id
// keeping its first value (what sparse_run does for one query; Qdrant sorts sparse vectors by index [EXT]).
struct SparseBatch {
  std::vector<int32_t> off;    // nq + 1: ranges of the queries the batch kernel serves (others: empty range)
  std::vector<int32_t> ids;
  std::vector<float> vals;
  std::vector<int32_t> alone;  // queries it cannot serve (more than kInvMaxTerms distinct terms): one by one
};

int prepare_sparse_batch(const int64_t* q_off, const int32_t* q_idx, const float* q_val, int nq, bool batchable,
                         SparseBatch* b) {
  b->off.assign(static_cast<size_t>(nq) + 1, 0);
  std::vector<std::pair<int32_t, float>> t;
  for (int i = 0; i < nq; ++i) {
    const int64_t lo = q_off[i], hi = q_off[i + 1];
    VR_CHECK(hi >= lo && hi - lo <= kMaxQueryTerms, "sparse query %d has %lld terms (0..%d supported)", i,
             static_cast<long long>(hi - lo), kMaxQueryTerms);
    t.clear();
    for (int64_t j = lo; j < hi; ++j) t.emplace_back(q_idx[j], q_val[j]);
    std::stable_sort(t.begin(), t.end(), [](const auto& a, const auto& c) { return a.first < c.first; });
    t.erase(std::unique(t.begin(), t.end(), [](const auto& a, const auto& c) { return a.first == c.first; }), t.end());
    if (!t.empty() && (!batchable || static_cast<int>(t.size()) > kInvMaxTerms)) {
      b->alone.push_back(i);
    } else {
      for (const auto& p : t) {
        b->ids.push_back(p.first);
        b->vals.push_back(p.second);
      }
    }
    b->off[static_cast<size_t>(i) + 1] = static_cast<int32_t>(b->ids.size());
  }
  return 0;
}

// Queues the batch kernel of the prepared queries on e->stream; the nq x k keys end up in e->sq_keys (device).
int sparse_batch_launch(vr_engine* e, const SparseBatch& b, int nq, int k, bool weights_given, const uint8_t* mask,
                        bool allow_grouped = true) {
  const int64_t nt = static_cast<int64_t>(b.ids.size());
  VR_TRY(e->sq_off.grow(nq + 1, 0, e->stream));
  VR_TRY(e->sq_ids.grow(std::max<int64_t>(nt, 1), 0, e->stream));
  VR_TRY(e->sq_val.grow(std::max<int64_t>(nt, 1), 0, e->stream));
  VR_TRY(e->sq_w.grow(std::max<int64_t>(2 * nt, 1), 0, e->stream));  // weights, then the terms' document-frequency shares
  VR_TRY(e->sq_keys.grow(static_cast<int64_t>(nq) * k, 0, e->stream));
  if (nt == 0 || e->n_rows == 0 || e->n_slices_dev == 0) {
    VR_HIP(hipMemsetAsync(e->sq_keys.p, 0, sizeof(uint64_t) * static_cast<size_t>(nq) * k, e->stream));
    return 0;
  }
  VR_HIP(hipMemcpyAsync(e->sq_off.p, b.off.data(), sizeof(int32_t) * (static_cast<size_t>(nq) + 1), hipMemcpyHostToDevice, e->stream));
  VR_HIP(hipMemcpyAsync(e->sq_ids.p, b.ids.data(), sizeof(int32_t) * static_cast<size_t>(nt), hipMemcpyHostToDevice, e->stream));
  VR_HIP(hipMemcpyAsync(e->sq_val.p, b.vals.data(), sizeof(float) * static_cast<size_t>(nt), hipMemcpyHostToDevice, e->stream));
  return inv_scan_topk_batch(e, e->sq_off.p, e->sq_ids.p, e->sq_val.p, e->sq_w.p, nq, static_cast<int>(nt), weights_given,
                             static_cast<float>(e->n_sparse_points), mask, k, e->sq_keys.p, b.off.data(), b.ids.data(),
                             allow_grouped);
}

// After the stream of sparse_batch_launch has been synchronised: the grouped scan gives up the queries whose candidate
// regions overflowed (invert.hip) — those are repeated on the per-query kernels, whose answer does not depend on any
// budget, and their rows of keys_host replaced. Runs on e->stream and waits for it.
int sparse_batch_redo_if_overflowed(vr_engine* e, const SparseBatch& b, int nq, int k, bool weights_given, const uint8_t* mask,
                                    uint64_t* keys_host) {
  if (*pin_host<int32_t>(e, kPinSparseOverflow) == 0 || !e->sq_overflow_q) return 0;
  std::vector<int32_t> flagged(static_cast<size_t>(nq));
  VR_HIP(hipMemcpyAsync(flagged.data(), e->sq_overflow_q, sizeof(int32_t) * static_cast<size_t>(nq), hipMemcpyDeviceToHost, e->stream));
  VR_HIP(hipStreamSynchronize(e->stream));
  SparseBatch again;  // the same batch with the other queries' terms left out (an empty range: an empty list, at no cost)
  again.off.assign(static_cast<size_t>(nq) + 1, 0);
  int64_t n_again = 0;
  for (int i = 0; i < nq; ++i) {
    if (flagged[static_cast<size_t>(i)]) {
      again.ids.insert(again.ids.end(), b.ids.begin() + b.off[static_cast<size_t>(i)], b.ids.begin() + b.off[static_cast<size_t>(i) + 1]);
      again.vals.insert(again.vals.end(), b.vals.begin() + b.off[static_cast<size_t>(i)], b.vals.begin() + b.off[static_cast<size_t>(i) + 1]);
      ++n_again;
    }
    again.off[static_cast<size_t>(i) + 1] = static_cast<int32_t>(again.ids.size());
  }
  e->stat_sparse_group_redo.fetch_add(n_again);
  if (n_again == 0) return 0;
  VR_TRY(sparse_batch_launch(e, again, nq, k, weights_given, mask, false));
  std::vector<uint64_t> keys(static_cast<size_t>(nq) * k);
  VR_HIP(hipMemcpyAsync(keys.data(), e->sq_keys.p, sizeof(uint64_t) * keys.size(), hipMemcpyDeviceToHost, e->stream));
  VR_HIP(hipStreamSynchronize(e->stream));
  for (int i = 0; i < nq; ++i)
    if (flagged[static_cast<size_t>(i)])
      memcpy(keys_host + static_cast<size_t>(i) * k, keys.data() + static_cast<size_t>(i) * k, sizeof(uint64_t) * static_cast<size_t>(k));
  return 0;
}

// The queries the batch kernel could not take, one at a time through the single-query scans (e->stream); their keys
// replace row i of keys_host. q_off / q_idx / q_val: the caller's arrays.
int sparse_batch_stragglers(vr_engine* e, const SparseBatch& b, const int64_t* q_off, const int32_t* q_idx, const float* q_val,
                            int k, bool weights_given, const uint8_t* mask, uint64_t* keys_host) {
  for (int32_t i : b.alone) {
    const int nnz = static_cast<int>(q_off[i + 1] - q_off[i]);
    uint64_t* dst = keys_host + static_cast<size_t>(i) * k;
    if (e->n_rows == 0 || e->n_slices_dev == 0) {
      memset(dst, 0, sizeof(uint64_t) * static_cast<size_t>(k));
      continue;
    }
    VR_TRY(search_sparse_block(e, q_idx + q_off[i], q_val + q_off[i], nnz, k, mask, weights_given));
    VR_HIP(hipStreamSynchronize(e->stream));
    memcpy(dst, pin_host<uint64_t>(e, kPinSparseKeys), sizeof(uint64_t) * static_cast<size_t>(k));
  }
  return 0;
}

// nq sparse searches -> nq x k keys in keys_host. `e` is a search lane; everything runs on e->stream.
int search_sparse_keys_locked(vr_engine* e, const int64_t* q_off, const int32_t* q_idx, const float* q_val, int nq, int k,
                              bool weights_given, const uint8_t* mask, uint64_t* keys_host) {
  SparseBatch b;
  VR_TRY(prepare_sparse_batch(q_off, q_idx, q_val, nq, k <= kFusedMaxK && inv_usable(e, 1), &b));
  VR_TRY(sparse_batch_launch(e, b, nq, k, weights_given, mask));
  VR_HIP(hipMemcpyAsync(keys_host, e->sq_keys.p, sizeof(uint64_t) * static_cast<size_t>(nq) * k, hipMemcpyDeviceToHost, e->stream));
  VR_HIP(hipStreamSynchronize(e->stream));
  VR_TRY(sparse_batch_redo_if_overflowed(e, b, nq, k, weights_given, mask, keys_host));
  return sparse_batch_stragglers(e, b, q_off, q_idx, q_val, k, weights_given, mask, keys_host);
}

// Both legs of nq hybrid queries: nq x k dense keys and nq x k sparse keys (host arrays). The sparse batch is queued
// on the lane's auxiliary stream first and runs beside the dense batch (its kernels are small and latency-bound).
int hybrid_keys_locked(vr_engine* e, const float* q, int nq, int mem, const int64_t* sq_off, const int32_t* sq_idx,
                       const float* sq_val, int k, bool weights_given, const vr_filter* filter, uint64_t* dense_host,
                       uint64_t* sparse_host) {
  const size_t bytes = sizeof(uint64_t) * static_cast<size_t>(nq) * k;
  if (e->n_rows == 0) {
    memset(dense_host, 0, bytes);
    memset(sparse_host, 0, bytes);
    return 0;
  }
  const uint8_t* mask = nullptr;
  VR_TRY(filter_build_mask(e, filter, &mask));
  if (nq == 1) {  // one query: the latency path (query in the pinned area / kernel arguments, both legs side by side)
    const int nnz = sq_off ? static_cast<int>(sq_off[1] - sq_off[0]) : 0;
    bool sparse_ran = false;
    VR_TRY(hybrid_one_query(e, q, mem, nnz ? sq_idx + sq_off[0] : nullptr, nnz ? sq_val + sq_off[0] : nullptr, nnz, k,
                            weights_given, mask, &sparse_ran));
    memcpy(dense_host, pin_host<uint64_t>(e, kPinDenseKeys), bytes);
    if (sparse_ran) memcpy(sparse_host, pin_host<uint64_t>(e, kPinSparseKeys), bytes);
    else memset(sparse_host, 0, bytes);
    return 0;
  }
  const bool have_sparse = sq_off != nullptr && sq_off[nq] > sq_off[0] && e->n_slices_dev > 0;
  SparseBatch b;
  if (have_sparse) {
    VR_CHECK(sq_idx && sq_val, "null sparse queries");
    VR_TRY(prepare_sparse_batch(sq_off, sq_idx, sq_val, nq, k <= kFusedMaxK && inv_usable(e, 1), &b));
    VR_HIP(hipEventRecord(e->ev_fork, e->stream));  // after the mask
    VR_HIP(hipStreamWaitEvent(e->aux_stream, e->ev_fork, 0));
    hipStream_t main_stream = e->stream;
    e->stream = e->aux_stream;
    int rc = sparse_batch_launch(e, b, nq, k, weights_given, mask);
    if (rc == 0 && hipMemcpyAsync(sparse_host, e->sq_keys.p, bytes, hipMemcpyDeviceToHost, e->stream) != hipSuccess) {
      set_error("copying the sparse keys failed");
      rc = -1;
    }
    e->stream = main_stream;
    if (rc != 0) return rc;
  } else {
    memset(sparse_host, 0, bytes);
  }
  VR_TRY(search_dense_keys_locked(e, q, nq, mem, k, filter, dense_host, nullptr, mask));
  if (have_sparse) {
    VR_HIP(hipStreamSynchronize(e->aux_stream));
    VR_TRY(sparse_batch_redo_if_overflowed(e, b, nq, k, weights_given, mask, sparse_host));
    VR_TRY(sparse_batch_stragglers(e, b, sq_off, sq_idx, sq_val, k, weights_given, mask, sparse_host));
  }
  return 0;
}

}  // namespace

extern "C" {

int vr_search_sparse_batch(vr_engine* e, const int64_t* q_off, const int32_t* q_idx, const float* q_val, int32_t nq,
                           int32_t k, int32_t weights_given, const vr_filter* filter, int64_t* rows, float* scores,
                           int32_t* counts) {
  VR_TRY(check_engine(e));
  VR_CHECK(q_off && rows && scores && nq >= 1, "bad arguments");
  VR_CHECK(k >= 1 && k <= kMaxK, "k = %d not in 1..%d", k, kMaxK);
  VR_CHECK(q_off[nq] == q_off[0] || (q_idx && q_val), "null sparse queries");
  SearchLane lane(e);
  VR_TRY(lane.acquire(false));
  vr_engine* L = lane.L;
  std::vector<uint64_t> keys(static_cast<size_t>(nq) * k, 0ull);
  if (L->n_rows > 0 && L->n_slices_dev > 0) {
    const uint8_t* mask = nullptr;
    VR_TRY(filter_build_mask(L, filter, &mask));
    VR_TRY(search_sparse_keys_locked(L, q_off, q_idx, q_val, nq, k, weights_given != 0, mask, keys.data()));
  }
  for (int i = 0; i < nq; ++i) {
    const int64_t c = decode_keys(keys.data() + static_cast<size_t>(i) * k, k, rows + static_cast<int64_t>(i) * k,
                                  scores + static_cast<int64_t>(i) * k);
    if (counts) counts[i] = static_cast<int32_t>(c);
  }
  return 0;
}

int vr_search_hybrid_keys(vr_engine* e, const float* q, int32_t nq, int mem, const int64_t* sq_off, const int32_t* sq_idx,
                          const float* sq_val, int32_t k, int32_t weights_given, const vr_filter* filter, uint64_t* keys,
                          int keys_mem) {
  VR_TRY(check_engine(e));
  VR_CHECK(q && keys && nq >= 1, "bad arguments");
  VR_CHECK(k >= 1 && k <= kMaxK, "k = %d not in 1..%d", k, kMaxK);
  VR_CHECK((mem == VR_MEM_HOST || mem == VR_MEM_DEVICE) && (keys_mem == VR_MEM_HOST || keys_mem == VR_MEM_DEVICE), "bad mem");
  SearchLane lane(e);
  VR_TRY(lane.acquire(true));
  vr_engine* L = lane.L;
  const size_t per = static_cast<size_t>(nq) * k;
  std::vector<uint64_t> dense(per), sparse(per);
  VR_TRY(hybrid_keys_locked(L, q, nq, mem, sq_off, sq_idx, sq_val, k, weights_given != 0, filter, dense.data(), sparse.data()));
  // [query][dense list, sparse list][k]
  std::vector<uint64_t> both;
  uint64_t* dst = keys;
  if (keys_mem == VR_MEM_DEVICE) {
    both.resize(2 * per);
    dst = both.data();
  }
  for (int i = 0; i < nq; ++i) {
    memcpy(dst + (2 * static_cast<size_t>(i)) * k, dense.data() + static_cast<size_t>(i) * k, sizeof(uint64_t) * k);
    memcpy(dst + (2 * static_cast<size_t>(i) + 1) * k, sparse.data() + static_cast<size_t>(i) * k, sizeof(uint64_t) * k);
  }
  if (keys_mem == VR_MEM_DEVICE) {
    VR_HIP(hipMemcpyAsync(keys, both.data(), sizeof(uint64_t) * 2 * per, hipMemcpyHostToDevice, L->stream));
    VR_HIP(hipStreamSynchronize(L->stream));
  }
  return 0;
}

int vr_fuse_batch(const int64_t* d_rows, const float* d_scores, const int32_t* d_counts, const int64_t* s_rows,
                  const float* s_scores, const int32_t* s_counts, int32_t nq, int32_t k, int32_t limit, double sparse_weight,
                  int32_t fusion, int32_t json_scores, int64_t* out_rows, double* out_scores, int32_t* out_from_dense,
                  int32_t* out_counts) {
  return fuse_batch(d_rows, d_scores, d_counts, s_rows, s_scores, s_counts, nq, k, limit, sparse_weight, fusion, json_scores,
                    out_rows, out_scores, out_from_dense, out_counts);
}

int vr_search_hybrid_batch(vr_engine* e, const float* q, int32_t nq, int mem, const int64_t* sq_off, const int32_t* sq_idx,
                           const float* sq_val, int32_t limit, double sparse_weight, int32_t fusion, const vr_filter* filter,
                           int64_t* out_rows, double* out_scores, int32_t* out_from_dense, int32_t* out_counts) {
  VR_TRY(check_engine(e));
  VR_CHECK(q && out_rows && out_scores && out_counts && nq >= 1, "bad arguments");
  VR_CHECK(limit >= 1 && limit * 3 <= kMaxK, "limit = %d not in 1..%d", limit, kMaxK / 3);
  VR_CHECK(fusion == VR_FUSION_MINMAX || fusion == VR_FUSION_RRF, "unknown fusion %d", fusion);
  VR_CHECK(mem == VR_MEM_HOST || mem == VR_MEM_DEVICE, "bad mem %d", mem);
  const int k = 3 * limit;  // prefetch_limit, vector_store.py:636
  const size_t per = static_cast<size_t>(nq) * k;
  std::vector<uint64_t> dense(per), sparse(per);
  {
    SearchLane lane(e);
    VR_TRY(lane.acquire(mem == VR_MEM_DEVICE));
    VR_TRY(hybrid_keys_locked(lane.L, q, nq, mem, sq_off, sq_idx, sq_val, k, false, filter, dense.data(), sparse.data()));
  }
  // fusion of every query on the host threads (vector_store.py:659-697, once per query)
  std::atomic<int> failed{0};
  parallel_for(nq, 8, [&](int64_t i) {
    int64_t d_rows[kMaxK], s_rows[kMaxK];
    float d_scores[kMaxK], s_scores[kMaxK];
    const int nd = static_cast<int>(decode_keys(dense.data() + static_cast<size_t>(i) * k, k, d_rows, d_scores));
    const int ns = static_cast<int>(decode_keys(sparse.data() + static_cast<size_t>(i) * k, k, s_rows, s_scores));
    int32_t* fd = out_from_dense ? out_from_dense + i * limit : nullptr;
    const int rc = fusion == VR_FUSION_MINMAX
                       ? fuse_minmax(d_rows, d_scores, nd, s_rows, s_scores, ns, limit, sparse_weight, 1, out_rows + i * limit,
                                     out_scores + i * limit, fd, out_counts + i)
                       : fuse_rrf(d_rows, nd, s_rows, ns, limit, sparse_weight, out_rows + i * limit, out_scores + i * limit,
                                  fd, out_counts + i);
    if (rc != 0) failed.store(1);
  });
  VR_CHECK(!failed.load(), "fusion failed");
  return 0;
}

int vr_merge_keys(vr_engine* e, const uint64_t* parts, int32_t n_parts, int32_t n_lists, int32_t k, int mem, int64_t* out_ids,
                  float* out_scores, int32_t* out_counts) {
  VR_TRY(check_engine(e));
  VR_CHECK(parts && out_ids && out_scores && n_parts >= 1 && n_lists >= 1 && k >= 1, "bad arguments");
  VR_CHECK(mem == VR_MEM_HOST || mem == VR_MEM_DEVICE, "bad mem %d", mem);
  SearchLane lane(e);
  VR_TRY(lane.acquire(mem == VR_MEM_DEVICE));
  vr_engine* L = lane.L;
  const int64_t total = static_cast<int64_t>(n_parts) * n_lists * k;
  const uint64_t* parts_dev = parts;
  if (mem == VR_MEM_HOST) {
    VR_TRY(L->mg_in.grow(total, 0, L->stream));
    VR_HIP(hipMemcpyAsync(L->mg_in.p, parts, sizeof(uint64_t) * static_cast<size_t>(total), hipMemcpyHostToDevice, L->stream));
    parts_dev = L->mg_in.p;
  }
  const int64_t per = static_cast<int64_t>(n_lists) * k;
  VR_TRY(L->mg_gid.grow(per, 0, L->stream));
  VR_TRY(L->mg_score.grow(per, 0, L->stream));
  VR_TRY(L->mg_cnt.grow(n_lists, 0, L->stream));
  VR_TRY(topk_merge_parts(L, parts_dev, n_parts, n_lists, k, L->mg_gid.p, L->mg_score.p, L->mg_cnt.p));
  VR_HIP(hipMemcpyAsync(out_ids, L->mg_gid.p, sizeof(int64_t) * static_cast<size_t>(per), hipMemcpyDeviceToHost, L->stream));
  VR_HIP(hipMemcpyAsync(out_scores, L->mg_score.p, sizeof(float) * static_cast<size_t>(per), hipMemcpyDeviceToHost, L->stream));
  std::vector<int32_t> cnt(static_cast<size_t>(n_lists));
  VR_HIP(hipMemcpyAsync(cnt.data(), L->mg_cnt.p, sizeof(int32_t) * static_cast<size_t>(n_lists), hipMemcpyDeviceToHost, L->stream));
  VR_HIP(hipStreamSynchronize(L->stream));
  if (out_counts) memcpy(out_counts, cnt.data(), sizeof(int32_t) * static_cast<size_t>(n_lists));
  return 0;
}

int vr_sparse_row_ids(vr_engine* e, const int64_t* rows, int64_t n, int32_t* out, int64_t cap, int mem, int32_t* stride,
                      int64_t* n_points) {
  VR_TRY(check_engine(e));
  VR_CHECK(stride != nullptr && n >= 0, "bad arguments");
  VR_CHECK(mem == VR_MEM_HOST || mem == VR_MEM_DEVICE, "bad mem %d", mem);
  std::lock_guard<std::mutex> writer(e->wmu);  // (the master's staging arrays)
  std::shared_lock<std::shared_mutex> view(e->rw);
  const int w = sparse_max_width(e);
  *stride = w;
  if (n_points) *n_points = 0;
  if (!out || n == 0 || w == 0) return 0;
  VR_CHECK(rows != nullptr, "null rows");
  const int64_t total = n * w;
  VR_CHECK(cap >= total, "room for %lld ids, %lld needed", static_cast<long long>(cap), static_cast<long long>(total));
  VR_TRY(e->stage_i64a.grow(n, 0, e->stream));
  VR_HIP(hipMemcpyAsync(e->stage_i64a.p, rows, sizeof(int64_t) * static_cast<size_t>(n), hipMemcpyHostToDevice, e->stream));
  int32_t* out_dev = out;
  if (mem == VR_MEM_HOST) {
    VR_TRY(e->stage_i32a.grow(total, 0, e->stream));
    out_dev = e->stage_i32a.p;
  }
  int64_t pts = 0;
  VR_TRY(sparse_row_ids(e, e->stage_i64a.p, n, w, out_dev, &pts));
  if (n_points) *n_points = pts;
  if (mem == VR_MEM_HOST) {
    VR_HIP(hipMemcpyAsync(out, out_dev, sizeof(int32_t) * static_cast<size_t>(total), hipMemcpyDeviceToHost, e->stream));
    VR_HIP(hipStreamSynchronize(e->stream));
  }
  return 0;
}

int vr_df_apply(vr_engine* e, const int32_t* ids, int64_t n_ids, int mem, int64_t n_points, int32_t sign) {
  VR_TRY(check_engine(e));
  VR_CHECK(n_ids >= 0 && (n_ids == 0 || ids) && (sign == 1 || sign == -1) && n_points >= 0, "bad arguments");
  VR_CHECK(mem == VR_MEM_HOST || mem == VR_MEM_DEVICE, "bad mem %d", mem);
  std::lock_guard<std::mutex> writer(e->wmu);
  PublishLock publish(e);  // (the table may be re-hashed; searches read it)
  const int32_t* ids_dev = ids;
  if (mem == VR_MEM_HOST && n_ids > 0) {
    VR_TRY(e->stage_i32a.grow(n_ids, 0, e->stream));
    VR_HIP(hipMemcpyAsync(e->stage_i32a.p, ids, sizeof(int32_t) * static_cast<size_t>(n_ids), hipMemcpyHostToDevice, e->stream));
    ids_dev = e->stage_i32a.p;
  }
  VR_TRY(sparse_df_apply(e, ids_dev, n_ids, sign));
  VR_HIP(hipStreamSynchronize(e->stream));
  e->n_sparse_points += sign * n_points;
  return 0;
}

}  // extern "C"

// ---- a question as TEXT, one call (the MCP search tool's three calls — embed_query, sparse embed_query,
// vector_store.search: mcp_server.py:469-485 — without the trips through Python between them) -----------------------

namespace {

// a few device rows for query embeddings in flight (one per concurrent vr_query_text)
struct QueryRows {
  std::mutex mu;
  std::vector<float*> free_rows;
  ~QueryRows() {
    for (float* p : free_rows) (void)hipFree(p);
  }
};
QueryRows g_query_rows[16];  // per device (indexed by ordinal % 16; rows are kMaxDim floats, any engine fits)

float* take_query_row(int device) {
  QueryRows& q = g_query_rows[device & 15];
  {
    std::lock_guard<std::mutex> g(q.mu);
    if (!q.free_rows.empty()) {
      float* p = q.free_rows.back();
      q.free_rows.pop_back();
      return p;
    }
  }
  float* p = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&p), sizeof(float) * kMaxDim) != hipSuccess) return nullptr;
  return p;
}

void give_query_row(int device, float* p) {
  QueryRows& q = g_query_rows[device & 15];
  std::lock_guard<std::mutex> g(q.mu);
  q.free_rows.push_back(p);
}

}  // namespace

extern "C" {

int vr_query_text(vr_engine* e, const vr_wordpiece* tokenizer, const char* dense_text, int64_t dense_len,
                  const char* sparse_text, int64_t sparse_len, int32_t max_len, int32_t limit, double sparse_weight,
                  int32_t fusion, const vr_filter* filter, int64_t* out_rows, double* out_scores, int32_t* out_from_dense,
                  int32_t* out_count, int32_t* out_hybrid) {
  VR_TRY(check_engine(e));
  VR_CHECK(tokenizer && dense_text && dense_len >= 0 && out_rows && out_scores && out_count, "bad arguments");
  VR_CHECK(limit >= 1 && limit * 3 <= kMaxK, "limit = %d not in 1..%d", limit, kMaxK / 3);
  VR_CHECK(fusion == VR_FUSION_MINMAX || fusion == VR_FUSION_RRF, "unknown fusion %d", fusion);
  VR_CHECK(max_len >= 2 && max_len <= 4096, "max_len %d", max_len);
  *out_count = 0;
  if (out_hybrid) *out_hybrid = 0;
  // 1. host: WordPiece ids of the (prefixed) query, hashed BM25 stems of the raw query (Bm25.query_embed: the SET of
  //    them, every value 1.0 — SURVEY.md a7)
  std::vector<int32_t> wp(static_cast<size_t>(max_len));
  int64_t wp_off[2] = {0, 0}, needed = 0;
  {
    const char* texts[1] = {dense_text};
    const int64_t lens[1] = {dense_len};
    VR_TRY(vr_wordpiece_encode(tokenizer, texts, lens, 1, max_len, wp_off, wp.data(), max_len, &needed));
  }
  std::vector<int32_t> stems;
  if (sparse_text && sparse_len > 0) {
    stems.resize(static_cast<size_t>(sparse_len / 2 + 2));
    const char* texts[1] = {sparse_text};
    const int64_t lens[1] = {sparse_len};
    int64_t off[2] = {0, 0}, need = 0;
    int rc = vr_bm25_tokenize(texts, lens, 1, off, stems.data(), static_cast<int64_t>(stems.size()), &need);
    if (rc == -2) {
      stems.resize(static_cast<size_t>(need));
      rc = vr_bm25_tokenize(texts, lens, 1, off, stems.data(), static_cast<int64_t>(stems.size()), &need);
    }
    VR_TRY(rc);
    stems.resize(static_cast<size_t>(need));
    std::sort(stems.begin(), stems.end());
    stems.erase(std::unique(stems.begin(), stems.end()), stems.end());
    VR_CHECK(static_cast<int>(stems.size()) <= kMaxQueryTerms, "query with %zu distinct terms", stems.size());
  }
  // 2. the embedding, left in device memory (the encoder is shared with the writers: one forward pass at a time)
  VR_CHECK(encoder_hidden(e) == e->dim, "encoder width %d != store dimension %d", encoder_hidden(e), e->dim);
  float* q_dev = take_query_row(e->device);
  VR_CHECK(q_dev != nullptr, "no device memory for the query embedding");
  struct Giver {
    int device;
    float* p;
    ~Giver() { give_query_row(device, p); }
  } giver{e->device, q_dev};
  const int32_t off32[2] = {0, static_cast<int32_t>(wp_off[1])};
  const bool hybrid = !stems.empty();
  if (out_hybrid) *out_hybrid = hybrid ? 1 : 0;
  // (VR_QUERY_TEXT_LANE_FIRST=0: the forward pass, then vr_search_hybrid as a caller would — the round's earlier form, for A/B timings)
  const bool lane_first = !(getenv("VR_QUERY_TEXT_LANE_FIRST") && atoi(getenv("VR_QUERY_TEXT_LANE_FIRST")) == 0);
  if (hybrid && lane_first) {
    // 3a. hybrid: the lane is taken and the filter mask built BEFORE the forward pass (both used to follow it, in a
    //     second engine call; worth a hundredth of a millisecond), then the forward pass, then both legs on the lane. Lock order as
    //     everywhere: the writers' mutex (the encoder), then the shared lock of the lane — a writer takes the same mutex
    //     before it publishes.
    const int k = limit * 3;  // prefetch_limit, vector_store.py:636
    std::vector<float> ones(stems.size(), 1.0f);
    std::unique_lock<std::mutex> writer(e->wmu);
    SearchLane lane(e);
    VR_TRY(lane.acquire(false));
    vr_engine* L = lane.L;
    if (L->n_rows == 0) return 0;
    const uint8_t* mask = nullptr;
    VR_TRY(filter_build_mask(L, filter, &mask));
    const int nnz = static_cast<int>(stems.size());
    // VR_QUERY_TEXT_AHEAD=1: the sparse leg is queued on the auxiliary stream BEFORE the forward pass (it needs the words,
    // not the embedding) instead of beside the dense scan. Measured on bench.py's from-text section and NOT the default:
    // 0.90-0.92 ms against 0.82-0.84 — the forward pass of one question is 63 small latency-bound kernels, and a sparse kernel that
    // holds every CU beside them slows each of them (profiles/r03_experiments.md §13).
    const bool ahead = L->n_slices_dev > 0 && k <= kFusedMaxK && getenv("VR_QUERY_TEXT_AHEAD") && atoi(getenv("VR_QUERY_TEXT_AHEAD")) != 0;
    if (ahead) VR_TRY(hybrid_sparse_ahead(L, stems.data(), ones.data(), nnz, k, mask, false));
    VR_TRY(encoder_encode(e, wp.data(), off32, 1, VR_MEM_HOST, q_dev, VR_MEM_DEVICE));  // (returns with the stream drained)
    writer.unlock();
    bool have_sparse = false;
    VR_TRY(hybrid_one_query(L, q_dev, VR_MEM_DEVICE, stems.data(), ones.data(), nnz, k, false, mask, &have_sparse, ahead));
    int64_t d_rows[kMaxK], s_rows[kMaxK];
    float d_scores[kMaxK], s_scores[kMaxK];
    const int nd = static_cast<int>(decode_keys(pin_host<uint64_t>(L, kPinDenseKeys), k, d_rows, d_scores));
    const int ns = have_sparse ? static_cast<int>(decode_keys(pin_host<uint64_t>(L, kPinSparseKeys), k, s_rows, s_scores)) : 0;
    if (fusion == VR_FUSION_MINMAX)
      return fuse_minmax(d_rows, d_scores, nd, s_rows, s_scores, ns, limit, sparse_weight, 1, out_rows, out_scores, out_from_dense,
                         out_count);
    return fuse_rrf(d_rows, nd, s_rows, ns, limit, sparse_weight, out_rows, out_scores, out_from_dense, out_count);
  }
  {
    std::lock_guard<std::mutex> writer(e->wmu);
    VR_TRY(encoder_encode(e, wp.data(), off32, 1, VR_MEM_HOST, q_dev, VR_MEM_DEVICE));  // (returns with the stream drained)
  }
  if (hybrid) {
    std::vector<float> ones(stems.size(), 1.0f);
    return vr_search_hybrid(e, q_dev, VR_MEM_DEVICE, stems.data(), ones.data(), static_cast<int32_t>(stems.size()), limit, sparse_weight,
                            fusion, filter, out_rows, out_scores, out_from_dense, out_count);
  }
  // no term survived the stop-word filter: the dense-only branch of VectorStoreService.search (vector_store.py:612-617)
  std::vector<float> sc(static_cast<size_t>(limit));
  int32_t c = 0;
  VR_TRY(vr_search_dense(e, q_dev, 1, VR_MEM_DEVICE, limit, filter, out_rows, sc.data(), &c));
  for (int i = 0; i < c; ++i) {
    out_scores[i] = static_cast<double>(sc[static_cast<size_t>(i)]);
    if (out_from_dense) out_from_dense[i] = 1;
  }
  *out_count = c;
  return 0;
}

}  // extern "C"
