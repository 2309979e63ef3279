// Internal declarations shared by the translation units of libvoitta_engine.so.
// Nothing here is part of the ABI; the ABI is include/voitta_engine.h.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "voitta_engine.h"

namespace vr {

void set_error(const char* fmt, ...);

#define VR_HIP(call)                                                                        \
  do {                                                                                      \
    hipError_t _err = (call);                                                               \
    if (_err != hipSuccess) {                                                               \
      ::vr::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_err), __FILE__,    \
                      __LINE__);                                                            \
      return -1;                                                                            \
    }                                                                                       \
  } while (0)

#define VR_CHECK(cond, ...)          \
  do {                               \
    if (!(cond)) {                   \
      ::vr::set_error(__VA_ARGS__);  \
      return -1;                     \
    }                                \
  } while (0)

#define VR_TRY(expr)        \
  do {                      \
    int _rc = (expr);       \
    if (_rc != 0) return _rc; \
  } while (0)

// Growable device array. grow() keeps the first `keep` elements.
template <class T>
struct DevArray {
  T* p = nullptr;
  int64_t cap = 0;
  int grow(int64_t need, int64_t keep, hipStream_t s) {
    if (need <= cap) return 0;
    int64_t ncap = cap ? cap : 1;
    while (ncap < need) ncap *= 2;
    T* np = nullptr;
    VR_HIP(hipMalloc(reinterpret_cast<void**>(&np), sizeof(T) * static_cast<size_t>(ncap)));
    if (p && keep > 0)
      VR_HIP(hipMemcpyAsync(np, p, sizeof(T) * static_cast<size_t>(keep), hipMemcpyDeviceToDevice, s));
    if (p) {
      VR_HIP(hipStreamSynchronize(s));
      VR_HIP(hipFree(p));
    }
    p = np;
    cap = ncap;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// One SELL-64 slice of the sparse index: 64 consecutive rows, entries stored in chunks of 4
// per lane so a wave reads 1 KiB per instruction: entry (lane, j) lives at
// off + (j/4)*256 + lane*4 + (j%4). width is a multiple of 4; padding ids are -1.
struct SliceDesc {
  int64_t off;
  int32_t row_base;
  int32_t nrows;
  int32_t width;
  int32_t pad;
};

// One segment of the inverted sparse index (invert.hip): the postings of <= kInvSegRows consecutive rows, sorted
// by (term, row). A posting is a 64-bit key — bits 0..11 the row inside the segment, bits 12..42 the term id,
// bits 43.. the segment's number inside its build batch (what the sort groups by) — plus the f32 weight.
struct InvSeg {
  int64_t off;       // first posting (into inv_key / inv_val)
  int32_t count;
  int32_t row_base;
  int32_t nrows;
  int32_t pad;
};
constexpr int kBatchCand = 1024;   // candidate budget per query of the batched dense search (batch.hip)
constexpr int kInvRowBits = 12;
constexpr int kInvSegRows = 1 << kInvRowBits;
constexpr int kInvSubShift = 43;
constexpr int kInvMaxTerms = 32;   // queries with more distinct terms take the forward (SELL) scan

// Position, in float4 units, of the four floats that lane group g (= k % 4) of row r owns inside a 1-KiB block (16 rows
// x 16 k) of the tiled f32 corpus. The 64 bytes a ROW owns of a block are contiguous (r * 4 + g): the exact re-score of
// single rows (batch.hip) then fetches half a cache line per block, not four sixteen-byte pieces of four lines — and
// a wave that reads the whole block, one float4 per lane, still covers the same contiguous KiB. (Until round 3 the
// position was g * 16 + r: file layout version 2 -> 3.)
constexpr inline int tile_pos(int g, int r) { return r * 4 + g; }

constexpr int kTileRows = 16;      // corpus rows per MFMA tile (v_mfma_f32_16x16x4_f32 M)
constexpr int kTileK = 16;         // k elements per 1-KiB tile block
constexpr int kTopkSeg = 4096;     // keys per block in the first select level
constexpr int kMaxK = 1024;        // largest top-k a search may ask for (hybrid: 3 * limit, so limit <= 341)
constexpr int kQueryBlock = 16;    // queries per dense pass (MFMA N)
constexpr int kScanBlocks = 512;   // grid of the fused scan+select kernels: 2 blocks per CU, and
                                   // 512 candidate lists per query for merge_lists_kernel
constexpr int kFusedMaxK = 64;     // fused selection keeps lists of 64 (one entry per lane)
constexpr int kMaxQueryTerms = 1024;  // distinct terms of one sparse query
constexpr int kMaxDim = 1024;         // dense dimension (the pinned query block below holds 16 x kMaxDim floats)

// layout of the pinned scratch
constexpr size_t kPinSparseIds = 0;            // int32[kMaxQueryTerms]
constexpr size_t kPinSparseVals = 4096;        // float[kMaxQueryTerms]
constexpr size_t kPinQuery = 8192;             // float[kQueryBlock][dim <= kMaxDim]
constexpr size_t kPinDenseKeys = 128 * 1024;   // uint64[kQueryBlock][kMaxK]
constexpr size_t kPinSparseKeys = 256 * 1024;  // uint64[kMaxK]
constexpr size_t kPinCandCount = 272 * 1024;   // int32: candidates of the last two-stage dense search
constexpr size_t kPinSparseOverflow = kPinCandCount + 16;  // int32: a query of the grouped sparse batch overflowed its candidate buffer
constexpr size_t kPinSparseCands = kPinCandCount + 20;     // int32: candidate keys the last grouped sparse batch ranked, all queries
constexpr size_t kPinnedBytes = 1 << 20;
static_assert(kPinSparseVals >= kPinSparseIds + sizeof(int32_t) * kMaxQueryTerms, "pinned layout");
static_assert(kPinQuery >= kPinSparseVals + sizeof(float) * kMaxQueryTerms, "pinned layout");
static_assert(kPinDenseKeys >= kPinQuery + sizeof(float) * kQueryBlock * kMaxDim, "pinned layout");
static_assert(kPinSparseKeys >= kPinDenseKeys + sizeof(uint64_t) * kQueryBlock * kMaxK, "pinned layout");
static_assert(kPinCandCount >= kPinSparseKeys + sizeof(uint64_t) * kMaxK && kPinCandCount + 64 <= kPinnedBytes, "pinned layout");
// re-score budget of the two-stage dense search: candidate TILES (16 rows, one tile read each: at most
// 16384 x 48 KiB = 0.8 GB at D = 768, a quarter of the one-stage scan it replaces). The candidate-row count
// the device reports is at most kMaxCandidates, or INT32_MAX when the tile budget overflowed.
constexpr int kMaxCandTiles = 16384;
constexpr int kMaxCandidates = kMaxCandTiles * 16;

}  // namespace vr

struct vr_engine {
  int device = 0;
  int dim = 0;
  int kblocks = 0;  // dim / 16
  hipStream_t stream = nullptr;      // stream every kernel of a call is queued on
  hipStream_t own_stream = nullptr;  // created by the engine; `stream` points here unless rebound
  // hybrid search runs its sparse leg here, concurrently with the dense leg (fork/join by events)
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_fork = nullptr;
  hipEvent_t ev_join = nullptr;
  // Concurrency (callers: MCP worker threads search while the indexing thread, the watcher
  // and the event loop mutate — watcher.py:149-171, indexing.py:281-288, api/routes/folders.py:137-143):
  //   rw       searches hold it SHARED for their whole duration; a mutation holds it EXCLUSIVELY only while it
  //            publishes (appends rows, sets tombstones, swaps in a compacted index)
  //   wmu      one writer (or encoder call) at a time; they share the master's stream and scratch
  //   lanes    a search runs on a LANE: a private vr_engine that owns its stream, pinned staging area and scratch
  //            arrays and holds a VIEW (pointers + counts, refreshed under the shared lock) of the master's index —
  //            so several searches run at once, each on its own HIP stream, over the same HBM-resident tables.
  std::shared_mutex rw;
  std::mutex wmu;
  std::atomic<int> writers_waiting{0};  // searches let a waiting writer in first (shared_mutex prefers readers)
  std::mutex lane_mu;
  std::condition_variable lane_cv;
  std::vector<vr_engine*> lanes_free;
  std::vector<vr_engine*> lanes_all;
  int lanes_max = 4;
  vr_engine* master = nullptr;  // set in a lane
  hipEvent_t ev_input = nullptr;  // lane: orders its stream behind the master's bound stream (device inputs)

  int64_t n_rows = 0;
  int64_t n_live = 0;
  int64_t cap_rows = 0;  // multiple of 64

  // dense corpus, MFMA-tiled: [row/16][k/16][tile_pos(k%4, row%16)][c = (k%16)/4]
  vr::DevArray<float> corpus;
  // f16 shadow of the corpus for the two-stage exact search (prefilter.hip), present when dim % 32 == 0
  // and not disabled: [row/16][k/32][lane = (k%32)/8*16 + row%16][8 halfs], plus the exact
  // rounding residual |x - f16(x)|_2 of every row.
  bool prefilter = false;
  // int8 form of the shadow (dim % 64 == 0, unless VR_PREFILTER=f16): corpus16 then holds
  // [row/16][k/64][lane = (k%64)/16*16 + row%16][16 int8] with one scale per row (row_scale), and
  // row_err the exact residual |x - scale * int8(x)|_2. Half the bytes of the f16 shadow again.
  bool prefilter8 = false;
  vr::DevArray<float> row_scale;
  // Centre of the int8 shadow (prefilter.hip "centred shadow"): the shadow holds the quantised RESIDUALS x - centre.
  // Any fixed vector is valid (a shift of every score by centre.q cancels out of the candidate test); the column
  // mean of the stored rows makes the residuals of a real embedding collection — rows that share a common direction —
  // several times smaller than the rows, and the bound with them. Recomputed (and the whole shadow rebuilt, ~1 ms per
  // million rows) whenever the collection has doubled since the last time, after a compaction and after a load.
  vr::DevArray<float> centre;      // [dim], zeros until the collection holds kCentreMinRows rows
  vr::DevArray<float> centre_sum;  // [dim] scratch of the column sums
  float centre_norm = 0.0f;        // |centre|_2, rounded up
  int64_t centre_rows = 0;         // rows the collection held when the centre was last recomputed
  int64_t centre_checked_rows = 0; // rows it held when a re-centring was last ATTEMPTED (centring switched off, or a non-finite
                                   // centre, leaves centre_rows at 0: without this every upsert would rebuild the whole shadow)
  // (a lane counts for itself; its counts are added to the master's when it is handed back)
  std::atomic<int64_t> stat_two_stage{0};       // single-query dense searches served by the two-stage path
  std::atomic<int64_t> stat_fallback{0};        // ... of which overflowed the re-score budget and were redone one-stage
  std::atomic<int64_t> stat_batched{0};         // queries served by the batched search (batch.hip)
  std::atomic<int64_t> stat_batch_cands{0};     // ... and the rows they re-scored exactly, in total
  std::atomic<int64_t> stat_batch_fallback{0};  // ... of which overflowed their candidate budget and were redone alone
  std::atomic<int64_t> stat_last_candidates{0}; // rows re-scored by the last two-stage search
  std::atomic<int64_t> stat_sparse_grouped{0};    // sparse queries served by the grouped batch scan (invert.hip)
  std::atomic<int64_t> stat_sparse_group_cands{0}; // candidate keys its selection ranked (read when the next batch starts)
  std::atomic<int64_t> stat_sparse_group_redo{0}; // queries it gave up (a candidate region overflowed) and the per-query kernels redid
  std::atomic<int64_t> generation{0};           // bumped whenever row numbers change (vr_compact's swap, vr_load)
  vr::DevArray<uint16_t> corpus16;
  vr::DevArray<float> row_err;
  vr::DevArray<float> upper;       // [cap_rows] upper bounds of the last prefilter pass
  vr::DevArray<int32_t> cand_rows; // candidate rows (+ counter at the end)
  vr::DevArray<uint64_t> cand_keys;
  vr::DevArray<uint8_t> live;
  vr::DevArray<int32_t> folder;
  vr::DevArray<int32_t> index_folder;
  vr::DevArray<int64_t> created;
  vr::DevArray<int64_t> modified;
  vr::DevArray<int32_t> row_slice;  // slice index of the row's sparse vector, -1 = none
  int32_t max_folder_id = -1;
  int32_t max_index_folder_id = -1;

  // sparse index (SELL-64, see SliceDesc)
  std::vector<vr::SliceDesc> slices_host;
  vr::DevArray<vr::SliceDesc> slices;
  int64_t n_slices_dev = 0;
  vr::DevArray<int32_t> sp_idx;
  vr::DevArray<float> sp_val;
  int64_t sp_used = 0;
  int64_t n_sparse_points = 0;

  // inverted twin of the SELL index (invert.hip), derived data: rebuilt by vr_load and vr_compact
  vr::DevArray<uint64_t> inv_key;
  vr::DevArray<float> inv_val;
  vr::DevArray<vr::InvSeg> inv_seg;
  int64_t inv_used = 0;      // postings
  int64_t n_inv_seg = 0;
  int64_t inv_slices = 0;    // slices the inverted index covers (== n_slices_dev when usable)
  int64_t inv_rows = 0;      // rows its segments span
  unsigned long long* inv_counter = nullptr;  // [0] postings emitted by the current build, [1] duplicates seen
  bool sp_has_dups = false;  // some row lists a term twice: queries stay on the forward scan

  // document-frequency table: open addressing, key -1 = empty
  vr::DevArray<int32_t> df_keys;
  vr::DevArray<int32_t> df_cnt;
  int64_t df_cap = 0;          // power of two
  int64_t df_bound = 0;        // upper bound of distinct keys (exact count + nnz since last read)
  int32_t* df_distinct = nullptr;  // device counter

  // scratch
  vr::DevArray<float> stage_dense;    // host->device staging of upsert / query rows
  vr::DevArray<float> stage_len;      // per-row length
  vr::DevArray<int64_t> stage_off;
  vr::DevArray<int32_t> stage_idx;
  vr::DevArray<float> stage_val;
  vr::DevArray<int32_t> stage_i32a;
  vr::DevArray<int32_t> stage_i32b;
  vr::DevArray<int64_t> stage_i64a;
  vr::DevArray<int64_t> stage_i64b;
  vr::DevArray<double> stage_f64;
  vr::DevArray<int32_t> bm_marks;     // bm25_tf_kernel scratch for documents longer than its LDS
  vr::DevArray<int32_t> bm_cnt;       // vr_index_batch: per-document distinct terms
  vr::DevArray<int32_t> bm_idx;       //                 padded term ids
  vr::DevArray<float> bm_val;         //                 padded tf weights (f32, as stored)
  vr::DevArray<float> enc_out;        //                 encoder output rows
  vr::DevArray<float> q_tiled;       // 16-query image, kblocks KiB
  vr::DevArray<float> scores;         // [16][cap_rows]
  vr::DevArray<float> sp_scores;      // [cap_rows]
  vr::DevArray<uint8_t> mask;         // [cap_rows]
  vr::DevArray<uint8_t> pass_folder;  // per folder id
  vr::DevArray<uint8_t> pass_ifolder;
  vr::DevArray<uint64_t> cand_a;
  vr::DevArray<uint64_t> cand_b;
  vr::DevArray<uint64_t> sp_cand;  // per-block lists of the fused sparse scan
  vr::DevArray<int32_t> q_ids;
  vr::DevArray<float> q_w;
  // batched dense search (batch.hip): preprocessed queries, their int8 images and constants, per-slab bounds,
  // thresholds, candidate rows / counts (+ overflow flags), exact keys (+ results)
  vr::DevArray<float> bq_hat, bq_params, bq_best, bq_thr;
  vr::DevArray<int32_t> bq_img, bq_cand, bq_cnt;
  vr::DevArray<uint64_t> bq_keys;
  vr::DevArray<uint16_t> bq_tile_ub;  // f16 bits: per (16-row tile, query) the largest upper bound (rounded up)
  vr::DevArray<int32_t> bq_pairs;     // per query: the tiles whose bound reaches its threshold ([nq][kBatchCand]), then the counts [nq]
  vr::DevArray<float> bq_stage;  // host queries staged on the device
  // batched sparse search (invert.hip): the queries' terms as CSR (offsets, ascending distinct ids, raw values,
  // weights q_t * idf_t) and the nq x k result keys
  vr::DevArray<int32_t> sq_off, sq_ids;
  vr::DevArray<float> sq_val, sq_w;
  vr::DevArray<uint64_t> sq_keys;
  // ... grouped form (sparse_inv_group_kernel): the group tables, per query a candidate buffer and its fill count
  vr::DevArray<int32_t> sq_grp, sq_cnt;
  const int32_t* sq_overflow_q = nullptr;  // (into sq_cnt) per query of the last grouped batch: 1 = its candidates overflowed
  vr::DevArray<uint64_t> sq_cand;   // (+ the groups' threshold keys behind the buffers)
  vr::DevArray<int32_t> sq_bounds;  // int2 run bounds per (segment, distinct term), then per (segment, group, union term)
  vr::DevArray<float> sq_entw;      // weights per (group, union term, query of the group)
  std::vector<int32_t> sq_grp_host;
  // vr_merge_keys: the parts' keys staged on the device, merged global ids / scores / counts
  vr::DevArray<uint64_t> mg_in;
  vr::DevArray<int64_t> mg_gid;
  vr::DevArray<float> mg_score;
  vr::DevArray<int32_t> mg_cnt;
  // Pinned, device-mapped host scratch (1 MiB). Query inputs are written here by the host and
  // read by the kernels straight over PCIe, results are written here by the last kernel of a
  // search: the latency path of a query has no hipMemcpy at all.
  void* pinned = nullptr;      // host address
  void* pinned_dev = nullptr;  // the same memory as the device sees it
  size_t pinned_bytes = 0;

  void* encoder = nullptr;   // vr::Encoder (encoder.hip)
  void* profiler = nullptr;  // vr::Profiler (profile.hip)
};


namespace vr {

// A mutation publishes under the exclusive lock: searches that are already running finish first, new ones wait
// (they yield while writers_waiting > 0 — a shared_mutex prefers readers and would starve the writer).
struct PublishLock {
  vr_engine* e;
  std::unique_lock<std::shared_mutex> lock;
  explicit PublishLock(vr_engine* eng) : e(eng) {
    e->writers_waiting.fetch_add(1, std::memory_order_acq_rel);
    lock = std::unique_lock<std::shared_mutex>(e->rw);
    e->writers_waiting.fetch_sub(1, std::memory_order_acq_rel);
  }
};

template <class T>
inline T* pin_host(vr_engine* e, size_t off) {
  return reinterpret_cast<T*>(static_cast<char*>(e->pinned) + off);
}
template <class T>
inline T* pin_dev(vr_engine* e, size_t off) {
  return reinterpret_cast<T*>(static_cast<char*>(e->pinned_dev) + off);
}

// ---- api.hip
int ensure_rows(vr_engine* e, int64_t need);  // grow every per-row table to hold `need` rows

// ---- persist.hip: on-disk image of the index (vr_save / vr_load)
int engine_save(vr_engine* e, const char* path);
int engine_load(vr_engine* e, const char* path);

// ---- dense.hip
int dense_store_rows(vr_engine* e, const float* x_dev, int64_t n, int64_t first_row);
int dense_make_query_image(vr_engine* e, const float* q_dev, int nq);
int dense_scores(vr_engine* e, int nq, const uint8_t* mask_dev);
int dense_read_rows(vr_engine* e, const int64_t* rows_dev, int64_t n, float* out_dev);

// ---- prefilter.hip: f16 prefilter + exact re-score (two-stage exact search)
int prefilter_store_rows(vr_engine* e, int64_t n, int64_t first_row);
// recompute the centre of the int8 shadow from rows [0, n_rows) and rebuild the shadow of all of them
int prefilter_recentre(vr_engine* e);
bool prefilter_usable(vr_engine* e, int nq, int k);
int prefilter_search(vr_engine* e, int k, const uint8_t* mask_dev, uint64_t* out_keys_dev, int32_t* out_count_dev);

// ---- batch.hip: batched dense search (int8 MFMA GEMM + exact re-score)
bool batch_usable(vr_engine* e, int nq, int k);
int batch_search(vr_engine* e, const float* q_dev, int nq, int k, const uint8_t* mask_dev, const uint64_t** out_keys_dev,
                 const int32_t** overflow_dev);

// ---- topk.hip
// scores: [nq][stride] f32 with -inf / masked rows excluded; result keys (descending) for each
// query are left in *out_keys (device, [nq][k]).
int topk_select(vr_engine* e, const float* scores, int64_t stride, int64_t n, int nq, int k,
                const uint64_t** out_keys);
// cand: [nq][n_lists][64] keys, every list sorted descending and zero padded (written by the
// fused scan kernels); writes the k best of every query to out ([nq][k], descending; device-
// visible memory, normally the pinned result area so that no copy follows).
int topk_merge_lists(vr_engine* e, uint64_t* cand, int n_lists, int nq, int k, uint64_t* out);  // clobbers cand
// per query the k best keys of its n_reg regions cand[q][r][cap] (region r holding min(cnt[q][r], cap) keys, unsorted,
// unique) and of its spill area spill[q][spill_cap] (spill_cnt[q] keys were offered to it; spill may be null);
// *overflow_mapped (device view of a pinned word) is set to 1, and overflow_q[q] (may be null; zeroed by the caller), when a
// query's spill area overflowed or it holds more than 8192 keys in all; *total_dev (may be null) += the keys gathered
int topk_select_regions(vr_engine* e, const uint64_t* cand, int n_reg, int cap, const int32_t* cnt, const uint64_t* spill,
                        int spill_cap, const int32_t* spill_cnt, int nq, int k, uint64_t* out, int32_t* overflow_mapped,
                        int32_t* total_dev, int32_t* overflow_q);
// fused scan + selection, k <= kFusedMaxK; results to out_keys_dev as above
int dense_scan_topk(vr_engine* e, int nq, int k, const uint8_t* mask_dev, uint64_t* out_keys_dev);
int sparse_scan_topk(vr_engine* e, const int32_t* q_idx_host, const float* q_val_host, int nnz, int k,
                     const uint8_t* mask_dev, bool weights_given, uint64_t* out_keys_dev);

// ---- sparse.hip
// rows of the batch live at idx/val[begin[r] .. +count); count = cnt_dev[r] or begin[r+1]-begin[r]
int sparse_append(vr_engine* e, int64_t n, int64_t first_row, const int32_t* cnt_host,
                  const int64_t* begin_dev, const int32_t* cnt_dev, const int32_t* idx_dev,
                  const float* val_dev, bool account = true);  // account: count df and points (not when re-packing)

// ---- compact.hip: drop tombstoned rows (vr_compact)
int engine_compact(vr_engine* e, int64_t* new_row_of_old_host, int64_t* n_rows_after);

// ---- bm25.hip
int bm25_tf(vr_engine* e, const int64_t* tok_off_dev, const int32_t* tok_ids_dev, int64_t n_docs,
            int64_t n_tokens, double k, double b, double avg_len, int32_t* out_cnt_dev,
            int32_t* out_idx_dev, double* out_val64_dev, float* out_val32_dev);
// inverted index: postings of slices [slice0, slice0 + n_new) (consecutive rows from first_row; nnz = their real
// entries, or -1 to have the device count them), and the one-kernel query over it
int inv_append(vr_engine* e, int64_t slice0, int64_t n_new, int64_t first_row, int64_t n_rows, int64_t nnz);
int inv_rebuild(vr_engine* e);
// rows of a caller's device CSR batch (sorted by id) that list a term twice: the count so far arrives in *out_host
// once the stream has been synchronised
int inv_note_csr_dups(vr_engine* e, const int64_t* off_dev, const int32_t* idx_dev, int64_t n,
                      unsigned long long* out_host);
void inv_release(vr_engine* e);
bool inv_usable(const vr_engine* e, int nnz);
int inv_scan_topk(vr_engine* e, const int32_t* q_idx_host, const float* q_val_host, int nnz, bool weights_given,
                  float n_points, const uint8_t* mask_dev, int k, uint64_t* out_keys_dev);
// many queries in one launch: CSR in device memory (ascending distinct ids, <= kInvMaxTerms per query; raw values in
// q_val_dev; q_w_dev: room for 2 x n_terms floats — the weights, then the terms' df shares); nq x k keys to out_keys_dev
// q_off_host / q_ids_host: the same CSR on the host (the grouped scan builds its group tables from it; nullptr or
// allow_grouped = false: the per-query kernels). The grouped scan may give up on a batch (a candidate buffer overflowed):
// it then leaves 1 in the pinned word kPinSparseOverflow, to be read once the stream has been synchronised, and 1 per
// overflowed query in e->sq_overflow_q (device) — the caller repeats THOSE queries with allow_grouped = false.
int inv_scan_topk_batch(vr_engine* e, const int32_t* q_off_dev, const int32_t* q_ids_dev, const float* q_val_dev,
                        float* q_w_dev, int nq, int n_terms, bool weights_given, float n_points, const uint8_t* mask_dev,
                        int k, uint64_t* out_keys_dev, const int32_t* q_off_host = nullptr,
                        const int32_t* q_ids_host = nullptr, bool allow_grouped = false);
int sparse_scores(vr_engine* e, const int32_t* q_idx_host, const float* q_val_host, int nnz,
                  const uint8_t* mask_dev, bool weights_given);
// term ids of the listed rows' sparse vectors as out_dev[i * stride + j] (-1: no such entry, dead row, no sparse
// vector); *n_points_host = listed rows that are live and carry a sparse vector
int sparse_max_width(const vr_engine* e);  // widest stored sparse row (a multiple of 4)
int sparse_row_ids(vr_engine* e, const int64_t* rows_dev, int64_t n, int stride, int32_t* out_dev, int64_t* n_points_host);
// df[id] += sign for every id >= 0 of ids_dev (document frequencies of rows stored on OTHER shards)
int sparse_df_apply(vr_engine* e, const int32_t* ids_dev, int64_t n, int sign);
// topk.hip: merge n_parts result lists per (query, list) — parts_dev [n_parts][n_lists][k] keys — into global ids
// (row * n_parts + part), scores and counts, device arrays [n_lists][k] / [n_lists]
int topk_merge_parts(vr_engine* e, const uint64_t* parts_dev, int n_parts, int n_lists, int k, int64_t* gid_dev,
                     float* score_dev, int32_t* cnt_dev);
int sparse_delete_rows(vr_engine* e, const int64_t* rows_dev, int64_t n, int64_t* n_deleted,
                       int64_t* n_sparse_deleted);
int sparse_lookup_df(vr_engine* e, const int32_t* ids_host, int n, int32_t* out_df_host);

// ---- encoder.hip
int encoder_load(vr_engine* e, const vr_bert_desc* d, const void* const* tensors, int n_tensors, int mem);
int encoder_encode(vr_engine* e, const int32_t* ids, const int32_t* offsets, int n_seq, int mem,
                   float* out, int out_mem);
void encoder_release(vr_engine* e);
int encoder_hidden(vr_engine* e);  // 0 when no encoder is loaded

// ---- profile.hip: HIP-event timing of one launch (no-ops unless vr_profile(e, 1))
void prof_begin(vr_engine* e, int kernel_class, double work);
void prof_end(vr_engine* e);
bool prof_on(vr_engine* e);  // HIP-event profiling active (kernels must then not be captured into graphs)
void prof_release(vr_engine* e);

// ---- filter.hip
// returns the device mask to use for this query (live[] when no filter is active)
int filter_build_mask(vr_engine* e, const vr_filter* f, const uint8_t** mask_out);

// ---- fusion.cpp (host only)
int fuse_minmax(const int64_t* d_rows, const float* d_scores, int nd, const int64_t* s_rows,
                const float* s_scores, int ns, int limit, double sparse_weight, int json_scores,
                int64_t* out_rows, double* out_scores, int32_t* out_from_dense, int32_t* out_count);
int fuse_rrf(const int64_t* d_rows, int nd, const int64_t* s_rows, int ns, int limit,
             double sparse_weight, int64_t* out_rows, double* out_scores, int32_t* out_from_dense,
             int32_t* out_count);

int fuse_batch(const int64_t* d_rows, const float* d_scores, const int32_t* d_counts, const int64_t* s_rows,
               const float* s_scores, const int32_t* s_counts, int nq, int k, int limit, double sparse_weight, int fusion,
               int json_scores, int64_t* out_rows, double* out_scores, int32_t* out_from_dense, int32_t* out_counts);

inline uint32_t f32_order_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

}  // namespace vr
