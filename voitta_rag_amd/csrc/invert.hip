// Inverted twin of the SELL-64 sparse index: what a single sparse (or hybrid) query reads.
//
// The forward scan of sparse.hip touches every stored id of the collection (4 B each: 166 MB for a million BM25
// rows), whatever the query asks; run beside the dense int8 scan of a hybrid query it also takes HBM bandwidth from
// it. A query of a handful of terms only needs the postings of those terms. They are kept here, derived from the
// SELL slices (never persisted: vr_load and vr_compact rebuild them), in a shape that keeps the result the same
// BITS as the forward scan's (reference behaviour restated in oracle/: score = sum over shared terms, ascending
// token id, every multiply and add rounded to f32; src/voitta/services/vector_store.py:647-656 is the call this
// serves):
//
//   * rows are cut into SEGMENTS of <= 4096 consecutive rows; a segment holds its postings sorted by (term, row),
//     as 64-bit keys (segment number | term id | row inside the segment) with the f32 weights beside them. A batch
//     of an upsert becomes its own segments (one bitonic sort per batch, below); when small upserts have left four
//     times the segments the rows need, the whole collection is sorted into full segments again;
//   * the query is ONE kernel, a block per segment: the query's terms are located in the segment by 64-way
//     searches (a wave per term), then taken in ascending term order — the postings of one term name distinct rows,
//     so the block adds them to a per-row f32 accumulator in LDS without conflicts, a barrier between terms keeps
//     the summation order of the forward scan — and finally the rows that were hit (and pass the filter mask) go
//     through the same per-wave top-k lists as every fused scan; merge_lists_kernel finishes.
//
// Algorithmic bytes: 12 B per posting of the query's terms + one mask byte per hit row; at a million rows and
// five Zipf-head terms ≈ 15 MB instead of 166 MB. Queries with more than kInvMaxTerms distinct terms, k > 64,
// the all-scores entry point, and collections in which some row lists a term twice (caller-supplied vectors may;
// the order of such a pair inside the sum is the row's own, which the sort here does not keep) stay on the forward
// scan (its cost does not grow with the number of terms).

#include "engine_internal.h"
#include "sparse_device.h"
#include "topk_device.h"

#include <algorithm>
#include <cstdlib>
#include <iterator>
#include <unordered_map>
#include <utility>
#include <vector>

namespace vr {

constexpr int kInvWaves = 4;
constexpr int kInvMaxSurvivors = 1024;  // rows of a segment the pruned path scores one by one (beyond: the full sum)
constexpr int kInvGroup = 4;  // terms whose postings a block requests together,
constexpr int kInvPer = 4;    // postings of each per thread
constexpr uint64_t kInvTermMask = 0x7FFFFFFFull;

__device__ __forceinline__ int32_t inv_term(uint64_t key) {
  return static_cast<int32_t>((key >> kInvRowBits) & kInvTermMask);
}

// ---- build ------------------------------------------------------------------------------------

// One wave per slice, lane = row. Real ids form a prefix of a row's column (padding is -1), so a lane counts its
// entries, the wave reserves room for all of them with one atomic and every lane writes its run.
__global__ __launch_bounds__(64) void inv_emit_kernel(const SliceDesc* __restrict__ slices, int64_t slice0,
                                                      int64_t first_row, int seg_rows,
                                                      const int32_t* __restrict__ sidx,
                                                      const float* __restrict__ sval, uint64_t* __restrict__ keys,
                                                      float* __restrict__ vals, unsigned long long* counter) {
  const SliceDesc d = slices[slice0 + blockIdx.x];
  const int lane = threadIdx.x;
  int c = 0;
  if (lane < d.nrows)
    for (int j = 0; j < d.width; ++j)
      c += sidx[d.off + static_cast<int64_t>(j >> 2) * 256 + lane * 4 + (j & 3)] >= 0;
  int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  const int total = __shfl(incl, 63);
  if (total == 0) return;
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(counter, static_cast<unsigned long long>(total));
  base = (static_cast<unsigned long long>(__shfl(static_cast<unsigned>(base >> 32), 0)) << 32) |
         __shfl(static_cast<unsigned>(base), 0);
  int64_t pos = static_cast<int64_t>(base) + incl - c;
  const int64_t rel = d.row_base + lane - first_row;
  const uint64_t hi = (static_cast<uint64_t>(rel / seg_rows) << kInvSubShift) | static_cast<uint64_t>(rel % seg_rows);
  for (int j = 0; j < c; ++j) {
    const int64_t src = d.off + static_cast<int64_t>(j >> 2) * 256 + lane * 4 + (j & 3);
    keys[pos] = hi | (static_cast<uint64_t>(static_cast<uint32_t>(sidx[src])) << kInvRowBits);
    vals[pos] = sval[src];
    ++pos;
  }
}

// The sort of a batch's postings by key: a bitonic network over the batch padded to a power of two with keys of all
// ones. Data independent (caller-supplied term ids are not hashes), in place, and off every latency path: a batch of
// 2200 chunks (88k postings) is 15 small launches, a million rows (40M) about sixty passes over 0.8 GB. Equal
// keys (a row that lists a term twice) may change places — such collections do not use this index (sp_has_dups).
constexpr int kSortTile = 4096;     // postings a block sorts in LDS (48 KiB)
constexpr int kSortThreads = 1024;

__global__ void inv_pad_kernel(uint64_t* __restrict__ keys, float* __restrict__ vals, int64_t from, int64_t to) {
  const int64_t i = from + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < to) {
    keys[i] = ~0ull;
    vals[i] = 0.0f;
  }
}

// FULL: every stage up to the tile size (a sorted tile, ascending or descending by the tile's place in the
// network); otherwise the strides below the tile size of stage `size`.
template <bool FULL>
__global__ __launch_bounds__(kSortThreads) void bitonic_tile_kernel(uint64_t* __restrict__ keys,
                                                                    float* __restrict__ vals, int64_t size) {
  __shared__ uint64_t k[kSortTile];
  __shared__ float v[kSortTile];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kSortTile;
#pragma unroll
  for (int j = 0; j < kSortTile / kSortThreads; ++j) {
    const int i = static_cast<int>(threadIdx.x) + j * kSortThreads;
    k[i] = keys[base + i];
    v[i] = vals[base + i];
  }
  __syncthreads();
  for (int64_t sz = FULL ? 2 : size; sz <= (FULL ? static_cast<int64_t>(kSortTile) : size); sz <<= 1) {
    for (int st = static_cast<int>(sz < kSortTile ? sz : kSortTile) >> 1; st > 0; st >>= 1) {
#pragma unroll
      for (int j = 0; j < kSortTile / 2 / kSortThreads; ++j) {
        const int t = static_cast<int>(threadIdx.x) + j * kSortThreads;
        const int i = ((t & ~(st - 1)) << 1) | (t & (st - 1));
        const int l = i + st;
        const bool asc = ((base + i) & sz) == 0;
        const uint64_t a = k[i], b2 = k[l];
        if ((a > b2) == asc) {
          k[i] = b2;
          k[l] = a;
          const float f = v[i];
          v[i] = v[l];
          v[l] = f;
        }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int j = 0; j < kSortTile / kSortThreads; ++j) {
    const int i = static_cast<int>(threadIdx.x) + j * kSortThreads;
    keys[base + i] = k[i];
    vals[base + i] = v[i];
  }
}

// one compare-exchange pass of stage `size` at a stride of at least the tile size
__global__ __launch_bounds__(256) void bitonic_global_kernel(uint64_t* __restrict__ keys, float* __restrict__ vals,
                                                             int64_t half, int64_t size, int64_t stride) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (t >= half) return;
  const int64_t i = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
  const int64_t l = i + stride;
  const bool asc = (i & size) == 0;
  const uint64_t a = keys[i], b = keys[l];
  if ((a > b) == asc) {
    keys[i] = b;
    keys[l] = a;
    const float f = vals[i];
    vals[i] = vals[l];
    vals[l] = f;
  }
}

// two passes of stage `size` in one: strides `stride` and `stride / 2` (both at least the tile size). A thread owns
// the four postings that differ in those two index bits, which is all either pass touches of them.
__global__ __launch_bounds__(256) void bitonic_global2_kernel(uint64_t* __restrict__ keys, float* __restrict__ vals,
                                                              int64_t quarter, int64_t size, int64_t stride) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (t >= quarter) return;
  const int64_t lo = stride >> 1;
  const int64_t i = ((t & ~(lo - 1)) << 2) | (t & (lo - 1));
  const bool asc = (i & size) == 0;
  const int64_t at[4] = {i, i + lo, i + stride, i + stride + lo};
  uint64_t k[4];
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    k[j] = keys[at[j]];
    v[j] = vals[at[j]];
  }
  auto cx = [&](int a, int b) {
    if ((k[a] > k[b]) == asc) {
      const uint64_t tk = k[a];
      k[a] = k[b];
      k[b] = tk;
      const float tv = v[a];
      v[a] = v[b];
      v[b] = tv;
    }
  };
  cx(0, 2);
  cx(1, 3);
  cx(0, 1);
  cx(2, 3);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    keys[at[j]] = k[j];
    vals[at[j]] = v[j];
  }
}

static int inv_sort(vr_engine* e, uint64_t* keys, float* vals, int64_t n_pad) {
  const unsigned tiles = static_cast<unsigned>(n_pad / kSortTile);
  hipLaunchKernelGGL((bitonic_tile_kernel<true>), dim3(tiles), dim3(kSortThreads), 0, e->stream, keys, vals,
                     static_cast<int64_t>(0));
  for (int64_t size = 2 * kSortTile; size <= n_pad; size <<= 1) {
    int64_t stride = size >> 1;
    for (; (stride >> 1) >= kSortTile; stride >>= 2)
      hipLaunchKernelGGL(bitonic_global2_kernel, dim3(static_cast<unsigned>((n_pad / 4 + 255) / 256)), dim3(256), 0,
                         e->stream, keys, vals, n_pad / 4, size, stride);
    if (stride >= kSortTile)
      hipLaunchKernelGGL(bitonic_global_kernel, dim3(static_cast<unsigned>((n_pad / 2 + 255) / 256)), dim3(256), 0,
                         e->stream, keys, vals, n_pad / 2, size, stride);
    hipLaunchKernelGGL((bitonic_tile_kernel<false>), dim3(tiles), dim3(kSortThreads), 0, e->stream, keys, vals, size);
  }
  VR_HIP(hipGetLastError());
  return 0;
}

__device__ __forceinline__ int64_t inv_lower_bound(const uint64_t* __restrict__ keys, int64_t n, uint64_t target) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < target) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// keys: the sorted postings of one build batch; segment j of the batch = the keys with bits 43.. == j
__global__ void inv_segments_kernel(const uint64_t* __restrict__ keys, int64_t n_ent, int64_t base_off,
                                    int64_t first_row, int64_t n_rows, int seg_rows, int n_sub,
                                    InvSeg* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_sub) return;
  const int64_t lo = inv_lower_bound(keys, n_ent, static_cast<uint64_t>(j) << kInvSubShift);
  const int64_t hi = inv_lower_bound(keys, n_ent, static_cast<uint64_t>(j + 1) << kInvSubShift);
  InvSeg s;
  s.off = base_off + lo;
  s.count = static_cast<int32_t>(hi - lo);
  s.row_base = static_cast<int32_t>(first_row + static_cast<int64_t>(j) * seg_rows);
  const int64_t left = n_rows - static_cast<int64_t>(j) * seg_rows;
  s.nrows = static_cast<int32_t>(left < seg_rows ? left : seg_rows);
  s.pad = 0;
  out[j] = s;
}

// The largest |weight| among a segment's postings, kept in its descriptor (InvSeg::pad, as float bits): the query
// kernels bound what a term can add to any row of the segment by |q_t idf_t| * vmax (dynamic pruning, below).
// kVmaxSplit blocks per segment (one block walking the 90k postings of a 2200-row batch alone took 140 us of every index
// step), combined with an atomic max on the bit patterns: magnitudes are non-negative, so their bits order like the values,
// and a NaN's bits are larger than any number's — it sticks, as before. inv_segments_kernel leaves pad at 0.
constexpr int kVmaxSplit = 32;
__global__ __launch_bounds__(256) void inv_vmax_kernel(InvSeg* __restrict__ segs, const float* __restrict__ vals) {
  __shared__ unsigned red[4];
  InvSeg* seg = segs + blockIdx.x / kVmaxSplit;
  const int part = static_cast<int>(blockIdx.x) % kVmaxSplit;
  const float* vp = vals + seg->off;
  const int per = (seg->count + kVmaxSplit - 1) / kVmaxSplit;
  const int lo = part * per, hi = min(seg->count, lo + per);
  unsigned m = 0;
  for (int i = lo + static_cast<int>(threadIdx.x); i < hi; i += 256) m = max(m, __float_as_uint(fabsf(vp[i])));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = max(m, static_cast<unsigned>(__shfl_xor(static_cast<int>(m), off)));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) m = max(m, red[w]);
    if (m != 0) atomicMax(reinterpret_cast<unsigned*>(&seg->pad), m);
  }
}

// counter[1] += postings equal to their left neighbour (a row that lists a term twice)
__global__ void inv_count_dups_kernel(const uint64_t* __restrict__ keys, int64_t n, unsigned long long* counter) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= 1 && i < n && keys[i] == keys[i - 1]) atomicAdd(counter + 1, 1ull);
}

// the same question asked of a CSR batch whose rows are sorted by id (a caller's device arrays)
__global__ void inv_csr_dups_kernel(const int64_t* __restrict__ off, const int32_t* __restrict__ idx, int64_t n,
                                    unsigned long long* counter) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= n) return;
  for (int64_t j = off[r] + 1; j < off[r + 1]; ++j)
    if (idx[j] == idx[j - 1]) {
      atomicAdd(counter + 1, 1ull);
      return;
    }
}

static int inv_counters(vr_engine* e) {
  if (!e->inv_counter) {
    VR_HIP(hipMalloc(reinterpret_cast<void**>(&e->inv_counter), 2 * sizeof(unsigned long long)));
    VR_HIP(hipMemsetAsync(e->inv_counter, 0, 2 * sizeof(unsigned long long), e->stream));
  }
  return 0;
}

int inv_note_csr_dups(vr_engine* e, const int64_t* off_dev, const int32_t* idx_dev, int64_t n,
                      unsigned long long* out_host) {
  VR_TRY(inv_counters(e));
  if (n > 0)
    hipLaunchKernelGGL(inv_csr_dups_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, e->stream,
                       off_dev, idx_dev, n, e->inv_counter);
  VR_HIP(hipGetLastError());
  VR_HIP(hipMemcpyAsync(out_host, e->inv_counter + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
  return 0;  // the caller synchronises the stream before it reads *out_host
}

void inv_release(vr_engine* e) {
  e->inv_key.release();
  e->inv_val.release();
  e->inv_seg.release();
  if (e->inv_counter) (void)hipFree(e->inv_counter);
  e->inv_counter = nullptr;
  e->inv_used = 0;
  e->n_inv_seg = 0;
  e->inv_slices = 0;
  e->inv_rows = 0;
}

int inv_append(vr_engine* e, int64_t slice0, int64_t n_new, int64_t first_row, int64_t n_rows, int64_t nnz) {
  if (n_new <= 0) return 0;
  VR_CHECK(slice0 == e->inv_slices, "inverted index out of step with the slices (%lld vs %lld)",
           static_cast<long long>(e->inv_slices), static_cast<long long>(slice0));
  VR_CHECK(n_rows >= 1 && n_rows < (1ll << 31) && slice0 + n_new <= static_cast<int64_t>(e->slices_host.size()),
           "bad inverted-index batch");
  const bool rebuilding = nnz < 0;  // (then only the device knows how many entries are real)
  const int64_t room = rebuilding ? std::max<int64_t>(e->sp_used, 1)  // no assumption about the order of a file's slices
                                  : std::max<int64_t>(nnz, 1);
  const int64_t n_sub = (n_rows + kInvSegRows - 1) / kInvSegRows;
  const int seg_rows = static_cast<int>((n_rows + n_sub - 1) / n_sub);  // equal parts: no sliver of a segment at the end
  int64_t n_pad = kSortTile;
  while (n_pad < room) n_pad <<= 1;
  VR_TRY(inv_counters(e));
  VR_HIP(hipMemsetAsync(e->inv_counter, 0, sizeof(unsigned long long), e->stream));
  // emitted and sorted in place, behind the postings already there (the padding of the sort is overwritten by the
  // next batch)
  VR_TRY(e->inv_key.grow(e->inv_used + n_pad, e->inv_used, e->stream));
  VR_TRY(e->inv_val.grow(e->inv_used + n_pad, e->inv_used, e->stream));
  VR_TRY(e->inv_seg.grow(e->n_inv_seg + n_sub, e->n_inv_seg, e->stream));
  uint64_t* keys = e->inv_key.p + e->inv_used;
  float* vals = e->inv_val.p + e->inv_used;
  hipLaunchKernelGGL(inv_emit_kernel, dim3(static_cast<unsigned>(n_new)), dim3(64), 0, e->stream, e->slices.p,
                     slice0, first_row, seg_rows, e->sp_idx.p, e->sp_val.p, keys, vals, e->inv_counter);
  VR_HIP(hipGetLastError());
  if (rebuilding) {
    unsigned long long c = 0;
    VR_HIP(hipMemcpyAsync(&c, e->inv_counter, sizeof(c), hipMemcpyDeviceToHost, e->stream));
    VR_HIP(hipStreamSynchronize(e->stream));
    nnz = static_cast<int64_t>(c);
    VR_CHECK(nnz <= room, "inverted-index rebuild: %lld entries in %lld slots", static_cast<long long>(nnz),
             static_cast<long long>(room));
    n_pad = kSortTile;
    while (n_pad < nnz) n_pad <<= 1;
  }
  if (nnz > 0) {
    if (n_pad > nnz)
      hipLaunchKernelGGL(inv_pad_kernel, dim3(static_cast<unsigned>((n_pad - nnz + 255) / 256)), dim3(256), 0,
                         e->stream, keys, vals, nnz, n_pad);
    VR_TRY(inv_sort(e, keys, vals, n_pad));
    if (rebuilding)
      hipLaunchKernelGGL(inv_count_dups_kernel, dim3(static_cast<unsigned>((nnz + 255) / 256)), dim3(256), 0,
                         e->stream, keys, nnz, e->inv_counter);
  }
  hipLaunchKernelGGL(inv_segments_kernel, dim3(static_cast<unsigned>((n_sub + 255) / 256)), dim3(256), 0, e->stream,
                     keys, nnz, e->inv_used, first_row, n_rows, seg_rows, static_cast<int>(n_sub),
                     e->inv_seg.p + e->n_inv_seg);
  hipLaunchKernelGGL(inv_vmax_kernel, dim3(static_cast<unsigned>(n_sub) * kVmaxSplit), dim3(256), 0, e->stream, e->inv_seg.p + e->n_inv_seg,
                     e->inv_val.p);
  VR_HIP(hipGetLastError());
  e->inv_used += nnz;
  e->n_inv_seg += n_sub;
  e->inv_slices = slice0 + n_new;
  e->inv_rows += n_rows;
  // Many small upserts leave many small segments, and a query pays a chain of dependent loads per segment: once
  // there are four times as many as rows / 4096 needs, sort the whole collection into full segments again (the
  // interval between two such rebuilds grows with the collection, like the re-centring of the int8 shadow).
  static const int64_t rebuild_factor = getenv("VR_INV_REBUILD_FACTOR") ? std::max(2, atoi(getenv("VR_INV_REBUILD_FACTOR"))) : 4;
  if (!rebuilding && e->n_inv_seg > 64 && e->n_inv_seg > rebuild_factor * ((e->inv_rows + kInvSegRows - 1) / kInvSegRows))
    return inv_rebuild(e);
  return 0;
}

int inv_rebuild(vr_engine* e) {
  e->inv_used = 0;
  e->n_inv_seg = 0;
  e->inv_slices = 0;
  e->inv_rows = 0;
  const int64_t n = static_cast<int64_t>(e->slices_host.size());
  if (n == 0) return 0;
  int64_t first = e->slices_host.front().row_base, end = first;
  for (const SliceDesc& d : e->slices_host) {
    first = std::min<int64_t>(first, d.row_base);
    end = std::max<int64_t>(end, static_cast<int64_t>(d.row_base) + d.nrows);
  }
  if (end == first) {
    e->inv_slices = n;
    return 0;
  }
  VR_TRY(inv_append(e, 0, n, first, end - first, -1));
  unsigned long long dups = 0;
  VR_HIP(hipMemcpyAsync(&dups, e->inv_counter + 1, sizeof(dups), hipMemcpyDeviceToHost, e->stream));
  VR_HIP(hipStreamSynchronize(e->stream));
  if (dups) e->sp_has_dups = true;
  return 0;
}

// ---- query ------------------------------------------------------------------------------------

static bool inv_enabled() {
  const char* v = std::getenv("VR_SPARSE_INVERTED");  // read per call: tests compare both scans in one process
  return !(v && v[0] == '0');
}

bool inv_usable(const vr_engine* e, int nnz) {
  return inv_enabled() && !e->sp_has_dups && nnz >= 1 && nnz <= kInvMaxTerms && e->n_inv_seg > 0 &&
         e->inv_slices == e->n_slices_dev;
}

// Lower bound of term t among the keys of a segment, found by the whole wave: 64 probes per step.
__device__ __forceinline__ int inv_wave_lower_bound(const uint64_t* __restrict__ kp, int count, int32_t t, int lane) {
  int lo = 0, hi = count;  // the answer lies in [lo, hi]
  while (hi - lo > 64) {
    const int step = (hi - lo + 63) >> 6;
    const int p = lo + lane * step;
    const bool less = p < hi && inv_term(kp[p]) < t;
    const int c = __popcll(__ballot(less));  // the probes are ascending: `less` holds for a prefix of the lanes
    const int nlo = c ? lo + (c - 1) * step + 1 : lo;
    const int nhi = c < 64 ? min(hi, lo + c * step) : hi;
    lo = nlo;
    hi = nhi;
  }
  const int p = lo + lane;
  const bool less = p < hi && inv_term(kp[p]) < t;
  return lo + __popcll(__ballot(less));
}

// the query travels in the kernel arguments: scalar loads, no trip to the host's pinned memory per block
struct InvQuery {
  int32_t id[kInvMaxTerms];
  float val[kInvMaxTerms];
};

// LDS of one block of the inverted scan (single-query and batched kernels share it and the body below)
struct InvShared {
  float score[kInvSegRows];   // (pruned segments: the candidate rows' numbers, as int32, in the same bytes)
  uint8_t hit[kInvSegRows];
  int32_t t_id[kInvMaxTerms];
  float t_w[kInvMaxTerms];
  int32_t t_lo[kInvMaxTerms];
  float t_frac[kInvMaxTerms];   // share of the rows that carry the term (df / N; 1 = unknown)
  int32_t t_ord[kInvMaxTerms];  // term indices by ascending |weight|
  float t_pre[kInvMaxTerms];    // t_pre[j] = |w| of terms t_ord[0..j] summed (rounded up)
  uint64_t lists[kInvWaves * kListLen];
  uint64_t tmax[kInvWaves * 64];
  uint64_t blk_thr;
  int32_t n_cand;
};

// the forward (SELL-64) side of the index, which the pruned path of the inverted scan reads row by row
struct InvForward {
  const int32_t* row_slice;
  const SliceDesc* slices;
  const int32_t* sidx;
  const float* sval;
};

__device__ __forceinline__ float inv_key_score(uint64_t key) {
  const uint32_t hi = static_cast<uint32_t>(key >> 32);
  return __uint_as_float((hi & 0x80000000u) ? (hi ^ 0x80000000u) : ~hi);
}

// The exact score of one row for the query in sh.t_id / sh.t_w (ascending ids): the row's own entries are walked in
// ascending id order and every shared term adds fl(fl(w * v)) to the sum from +0.0 — the forward scan's arithmetic
// (sparse.hip) and the inverted accumulation's, hence their bits. *hit: the row shares a term with the query.
__device__ __forceinline__ float inv_row_score(const int32_t* t_id, const float* t_w, const InvForward& fw, int64_t row,
                                               int nnz, bool* hit) {
  float acc = 0.0f;
  *hit = false;
  const int32_t s = fw.row_slice[row];
  if (s < 0) return acc;
  const SliceDesc d = fw.slices[s];
  const int lane = static_cast<int>(row - d.row_base);
  const int32_t last = t_id[nnz - 1];
  const int n_chunks = d.width / 4;
  const int32_t* base = fw.sidx + d.off + lane * 4;
  int qp = 0;
  // the row's ids are requested a dozen 16-byte chunks at a time (their addresses depend on nothing but the slice):
  // one memory round trip for a row of up to 48 entries instead of one per chunk
  constexpr int kAhead = 12;
  for (int c0 = 0; c0 < n_chunks; c0 += kAhead) {
    int4 ids[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u)
      ids[u] = c0 + u < n_chunks ? *reinterpret_cast<const int4*>(base + static_cast<int64_t>(c0 + u) * 256) : make_int4(-1, -1, -1, -1);
    bool done = false;
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const int32_t id4[4] = {ids[u].x, ids[u].y, ids[u].z, ids[u].w};
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int32_t id = id4[v];
        if (done || id < 0 || id > last) {  // padding (real ids are a prefix of the row) or beyond the query's last term
          done = true;
          continue;
        }
        while (qp < nnz && t_id[qp] < id) ++qp;
        if (qp < nnz && t_id[qp] == id) {
          acc = __fadd_rn(acc, __fmul_rn(t_w[qp], fw.sval[d.off + static_cast<int64_t>(c0 + u) * 256 + lane * 4 + v]));
          *hit = true;
        }
      }
    }
    if (done) break;
  }
  return acc;
}

// sh.t_ord: the query's terms by ascending |weight| (ties: by index), sh.t_pre: the running sums of those weights
// (rounded up). Thread 0 sorts (<= 32 terms: insertion sort); every thread of the block calls this.
__device__ __forceinline__ void inv_order_terms(InvShared& sh, int nnz) {
  const float* t_w = sh.t_w;
  if (threadIdx.x == 0) {  // terms by ascending |weight| (<= 32 of them: insertion sort), and the running sums
    for (int i = 0; i < nnz; ++i) {
      const float a = fabsf(t_w[i]);
      int j = i;
      while (j > 0 && fabsf(t_w[sh.t_ord[j - 1]]) > a) {
        sh.t_ord[j] = sh.t_ord[j - 1];
        --j;
      }
      sh.t_ord[j] = i;
    }
    float run = 0.0f;
    for (int j = 0; j < nnz; ++j) {
      run = (run + fabsf(t_w[sh.t_ord[j]])) * 1.000001f;
      sh.t_pre[j] = run;
    }
  }
  __syncthreads();
}

// The block's share of the segments (s0, s0 + s_step, ...) for ONE query whose terms (ascending ids) and weights
// already sit in sh.t_id / sh.t_w; leaves the block's k best keys in sh.lists[0 .. kListLen) (wave 0's list).
//
// Dynamic pruning (exact; the MaxScore idea). A block walks several segments for one query and its lists fill up: once
// a wave's list holds k keys, the k-th of them is a LOWER bound theta of the final k-th best score. A term t can add
// at most |w_t| vmax to a row of a segment (vmax: the segment's largest |weight|, in its descriptor), so with the terms
// ordered by |w_t| the first j of them are NON-ESSENTIAL when (|w_(1)| + .. + |w_(j)|) vmax < theta: a row that shares
// only such terms with the query cannot reach the top k. Their postings — the long lists of the common terms, which
// every query of a Zipfian vocabulary carries — are then not read at all: only the ESSENTIAL terms' postings are walked
// (to mark rows), and every marked row is scored exactly from its own forward (SELL) entries, in ascending id order —
// the same sum, the same bits. With theta unknown (the first segments of a block, or a list that never fills) and for
// negative thresholds the full accumulation below runs. The sums are rounded up and compared strictly, so a row that
// ties with theta is never pruned.
__device__ __forceinline__ void inv_scan_segments(InvShared& sh, const InvSeg* __restrict__ segs, int n_seg, int s0,
                                                  int s_step, const uint64_t* __restrict__ keys,
                                                  const float* __restrict__ vals, int nnz,
                                                  const uint8_t* __restrict__ mask, int k, const InvForward fw,
                                                  int seed_ne = -1, uint64_t theta0 = 0ull) {
  // seed_ne >= 0: the SEED pass of a batched search — every segment takes the pruned path with exactly the terms
  // t_ord[seed_ne ..] (the few rarest ones) as essential, whatever the lists hold: the k best rows that carry one of them,
  // scored exactly, give a threshold theta0 that the main pass starts from in every block (see inv_scan_topk_batch).
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  uint64_t* list = sh.lists + wave * kListLen;
  float* score = sh.score;
  uint8_t* hit = sh.hit;
  const int32_t* t_id = sh.t_id;
  const float* t_w = sh.t_w;
  int32_t* t_lo = sh.t_lo;
  for (int s = s0; s < n_seg; s += s_step) {
    const InvSeg seg = segs[s];
    if (seg.count == 0) continue;  // block-uniform
    const uint64_t* kp = keys + seg.off;
    const float* vp = vals + seg.off;
    // theta: the best k-th key any wave of the block holds (block-uniform: the lists are only written between barriers)
    uint64_t theta_key = theta0;
#pragma unroll
    for (int w = 0; w < kInvWaves; ++w) {
      const uint64_t kth = sh.lists[w * kListLen + (k - 1)];
      theta_key = kth > theta_key ? kth : theta_key;
    }
    int n_ne = 0;  // non-essential terms: t_ord[0 .. n_ne)
    if (seed_ne >= 0) {
      n_ne = seed_ne;
    } else if (theta_key != 0ull) {
      const float theta = inv_key_score(theta_key);
      const float vmax = __int_as_float(seg.pad);
      while (n_ne < nnz && sh.t_pre[n_ne] * vmax * 1.00001f < theta) ++n_ne;  // (false for NaN / inf / theta <= 0)
    }
    if (n_ne > 0 && seed_ne < 0) {
      // worth it? The pruned path adds up the essential terms' postings and scores the survivors one by one; it pays
      // when the postings it skips are a good part of all of them: judged by the terms' document frequencies
      float ess = 0.0f, all = 0.0f;
      for (int j = 0; j < nnz; ++j) {
        const float f = sh.t_frac[sh.t_ord[j]];
        all += f;
        if (j >= n_ne) ess += f;
      }
      if (!(ess * 1.5f < all)) n_ne = 0;
    }
    if (n_ne > 0 || seed_ne >= 0) {
      // ---- pruned: add up the ESSENTIAL terms' postings only (a partial score per row: an upper bound of the row's
      // score once the non-essential terms' bound is added), then score exactly — from the forward index — the rows
      // whose bound still reaches theta. The partial sums only FILTER; the score that is ranked is the full sum in
      // ascending term order (inv_row_score), the same bits as ever.
      for (int r = threadIdx.x; r < kInvSegRows; r += kInvWaves * 64) {
        score[r] = 0.0f;
        hit[r] = 0;
      }
      if (threadIdx.x == 0) sh.n_cand = 0;
      const int n_es = nnz - n_ne;
      for (int j = wave; j < n_es; j += kInvWaves) {
        const int i = sh.t_ord[n_ne + j];
        const int lb = inv_wave_lower_bound(kp, seg.count, t_id[i], lane);
        if (lane == 0) t_lo[i] = lb;
      }
      __syncthreads();
      for (int j = 0; j < n_es; ++j) {  // one term at a time: its postings name distinct rows, the next term's may name the same
        const int i = sh.t_ord[n_ne + j];
        const int32_t t = t_id[i];
        const float w = t_w[i];
        for (int p = t_lo[i] + static_cast<int>(threadIdx.x);; p += kInvWaves * 64) {
          bool in_run = false;
          if (p < seg.count) {
            const uint64_t k2 = kp[p];
            in_run = inv_term(k2) == t;
            if (in_run) {
              const int r = static_cast<int>(k2 & (kInvSegRows - 1));
              score[r] += fabsf(w * vp[p]);  // (magnitudes: a bound that no cancellation between terms can spoil)
              hit[r] = 1;
            }
          }
          if (!__all(in_run)) break;  // a wave's chunks only move away from the term's run
        }
        __syncthreads();
      }
      // survivors: partial + (what the non-essential terms could still add) >= theta, everything rounded up
      const float theta_f = seed_ne >= 0 ? -__builtin_inff() : inv_key_score(theta_key);
      const float ne_ub = (n_ne > 0 && seed_ne < 0) ? sh.t_pre[n_ne - 1] * __int_as_float(seg.pad) * 1.00001f : 0.0f;
      uint16_t* cand = reinterpret_cast<uint16_t*>(sh.tmax);  // room for kInvMaxSurvivors rows
#pragma unroll
      for (int j = 0; j < kInvSegRows / (kInvWaves * 64); ++j) {
        const int r = static_cast<int>(threadIdx.x) + j * kInvWaves * 64;
        if (r < seg.nrows && hit[r]) {
          const float part = score[r];
          if (!(part * 1.00001f + ne_ub < theta_f)) {
            const int at = atomicAdd(&sh.n_cand, 1);
            if (at < kInvMaxSurvivors) cand[at] = static_cast<uint16_t>(r);
          }
        }
      }
      __syncthreads();
      const int n_cand = sh.n_cand;
      if (n_cand <= kInvMaxSurvivors) {  // block-uniform
        for (int c0 = 0; c0 < n_cand; c0 += kInvWaves * 64) {  // block-uniform trip count (wave_offer is wave-collective)
          const int c = c0 + static_cast<int>(threadIdx.x);
          uint64_t key = 0ull;
          if (c < n_cand) {
            const int64_t row = static_cast<int64_t>(seg.row_base) + cand[c];
            if (mask[row]) {
              bool shares = false;
              const float sc = inv_row_score(sh.t_id, sh.t_w, fw, row, nnz, &shares);
              if (shares) key = topk_make_key(sc, row);
            }
          }
          wave_offer(list, k, key, 0, key != 0ull, lane);
        }
        __syncthreads();  // the next segment reads the lists (theta) and clears the accumulators
        continue;
      }
      __syncthreads();  // too many survivors to score one by one: this segment is added up in full below
    }
    for (int r = threadIdx.x; r < kInvSegRows; r += kInvWaves * 64) {
      score[r] = 0.0f;
      hit[r] = 0;
    }
    for (int i = wave; i < nnz; i += kInvWaves) {
      const int lb = inv_wave_lower_bound(kp, seg.count, t_id[i], lane);
      if (lane == 0) t_lo[i] = lb;
    }
    __syncthreads();
    // Ascending term id: the forward scan's summation order. The postings of kInvGroup terms are requested
    // together (a thread takes postings t_lo + tid + 256 u, u < kInvPer, of each: all addresses are known), so the
    // terms that follow cost LDS work only; a term with more than 256 kInvPer postings in this segment takes the
    // general loop below. (No row lists a term twice here: sp_has_dups keeps such collections on the forward scan.)
    for (int g0 = 0; g0 < nnz; g0 += kInvGroup) {
      uint64_t key[kInvGroup][kInvPer];
      float val[kInvGroup][kInvPer];
#pragma unroll
      for (int u = 0; u < kInvGroup; ++u) {
        const int i = g0 + u;
        const int base = (i < nnz ? t_lo[i] : 0) + static_cast<int>(threadIdx.x);
#pragma unroll
        for (int v = 0; v < kInvPer; ++v) {
          const int p = base + v * kInvWaves * 64;
          const int pc = (i < nnz && p < seg.count) ? p : 0;
          key[u][v] = kp[pc];
          val[u][v] = vp[pc];
        }
      }
#pragma unroll
      for (int u = 0; u < kInvGroup; ++u) {
        const int i = g0 + u;
        if (i >= nnz) break;  // block-uniform
        const int32_t t = t_id[i];
        const float w = t_w[i];
        const int base = t_lo[i] + static_cast<int>(threadIdx.x);
        bool mine[kInvPer];
        bool general = false;
#pragma unroll
        for (int v = 0; v < kInvPer; ++v) {
          mine[v] = base + v * kInvWaves * 64 < seg.count && inv_term(key[u][v]) == t;
        }
        general = mine[kInvPer - 1] && threadIdx.x == kInvWaves * 64 - 1;  // the run may go on
        if (!__syncthreads_or(general)) {  // (the barrier also orders this term after the one before)
#pragma unroll
          for (int v = 0; v < kInvPer; ++v)
            if (mine[v]) {
              const int r = static_cast<int>(key[u][v] & (kInvSegRows - 1));
              score[r] = __fadd_rn(score[r], __fmul_rn(w, val[u][v]));
              hit[r] = 1;
            }
          continue;
        }
        for (int p = t_lo[i] + wave * 64 + lane;; p += kInvWaves * 64) {
          bool in_run = false;
          if (p < seg.count) {
            const uint64_t k2 = kp[p];
            in_run = inv_term(k2) == t;
            if (in_run) {
              const int r = static_cast<int>(k2 & (kInvSegRows - 1));
              score[r] = __fadd_rn(score[r], __fmul_rn(w, vp[p]));
              hit[r] = 1;
            }
          }
          if (!__all(in_run)) break;  // a wave's chunks only move away from the term's run
        }
      }
    }
    if (threadIdx.x == 0) sh.blk_thr = theta_key;  // (a full list's k-th key already is such a bound: no ranking then)
    __syncthreads();
    // Selection. Thousands of rows may have been hit and a list insert is serial work for its wave, so first a
    // bound: a thread takes rows tid, tid + 256, ...; the k-th largest of the 256 per-thread maxima has k keys at or
    // above it (keys are unique: they carry the row), so only keys that reach it can be among the segment's k best.
    uint64_t rkey[kInvSegRows / (kInvWaves * 64)];
    uint64_t best = 0ull;
#pragma unroll
    for (int j = 0; j < kInvSegRows / (kInvWaves * 64); ++j) {
      const int r = static_cast<int>(threadIdx.x) + j * kInvWaves * 64;
      const bool in = r < seg.nrows && hit[r];
      const int64_t row = static_cast<int64_t>(seg.row_base) + r;
      rkey[j] = (in && mask[row]) ? topk_make_key(score[r], row) : 0ull;
      best = rkey[j] > best ? rkey[j] : best;
    }
    sh.tmax[threadIdx.x] = best;
    __syncthreads();
    if (best && theta_key == 0ull) {
      int rank = 0;
      for (int j = 0; j < kInvWaves * 64; ++j) rank += sh.tmax[j] > best;
      if (rank == k - 1) sh.blk_thr = best;
    }
    __syncthreads();
    const uint64_t thr = sh.blk_thr;
#pragma unroll
    for (int j = 0; j < kInvSegRows / (kInvWaves * 64); ++j) {
      if (j * kInvWaves * 64 >= seg.nrows) break;  // block-uniform
      const bool ok = rkey[j] != 0ull && rkey[j] >= thr;
      wave_offer(list, k, rkey[j], 0, ok, lane);
    }
    __syncthreads();  // the next segment clears the accumulators
  }
  block_merge_lists(sh.lists, kListLen, kInvWaves, wave, lane);
}

__global__ __launch_bounds__(kInvWaves * 64) void sparse_inv_kernel(
    const InvSeg* __restrict__ segs, int n_seg, const uint64_t* __restrict__ keys, const float* __restrict__ vals,
    const InvQuery query, int nnz, int weights_given,
    const int32_t* __restrict__ df_keys, const int32_t* __restrict__ df_cnt, int64_t df_cap, float n_points,
    const uint8_t* __restrict__ mask, int k, uint64_t* __restrict__ cand, const InvForward fw) {
  __shared__ InvShared sh;
  sh.lists[threadIdx.x] = 0ull;  // blockDim.x == kInvWaves * kListLen
  if (static_cast<int>(threadIdx.x) < nnz) {
    const int32_t id = query.id[threadIdx.x];
    sh.t_id[threadIdx.x] = id;
    sh.t_w[threadIdx.x] = sparse_query_weight(query.val[threadIdx.x], id, weights_given, df_keys, df_cnt, df_cap, n_points);
    sh.t_frac[threadIdx.x] = sparse_term_fraction(id, weights_given, df_keys, df_cnt, df_cap, n_points);
  }
  __syncthreads();
  inv_order_terms(sh, nnz);
  inv_scan_segments(sh, segs, n_seg, blockIdx.x, gridDim.x, keys, vals, nnz, mask, k, fw);
  if (threadIdx.x < kListLen) cand[static_cast<int64_t>(blockIdx.x) * kListLen + threadIdx.x] = sh.lists[threadIdx.x];
}

// ---- many queries per launch (vr_search_sparse_batch, the sparse leg of vr_search_hybrid_batch; BASELINE
// configs[4]: 1k batched hybrid queries). The reference call this stands for is the sparse query_points of
// vector_store.py:647-656, once per query. Same body, same bits: grid (segment share, query).

// q_t * idf_t for every term of every query of the batch, once (instead of once per block)
__global__ void sparse_batch_weights_kernel(const int32_t* __restrict__ ids, const float* __restrict__ vals, int n,
                                            int weights_given, const int32_t* __restrict__ df_keys,
                                            const int32_t* __restrict__ df_cnt, int64_t df_cap, float n_points,
                                            float* __restrict__ out, float* __restrict__ frac) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = sparse_query_weight(vals[i], ids[i], weights_given, df_keys, df_cnt, df_cap, n_points);
  frac[i] = sparse_term_fraction(ids[i], weights_given, df_keys, df_cnt, df_cap, n_points);
}

// q_off[nq + 1]: the terms of query y are ids / w [q_off[y], q_off[y + 1]) — ascending, distinct, at most
// kInvMaxTerms (a query with none, or one the host serves another way, has an empty range and gets empty lists)
__global__ __launch_bounds__(kInvWaves * 64) void sparse_inv_batch_kernel(
    const InvSeg* __restrict__ segs, int n_seg, const uint64_t* __restrict__ keys, const float* __restrict__ vals,
    const int32_t* __restrict__ q_off, const int32_t* __restrict__ q_ids, const float* __restrict__ q_w,
    const float* __restrict__ q_frac, const uint8_t* __restrict__ mask, int k, uint64_t* __restrict__ cand,
    const InvForward fw, const uint64_t* __restrict__ seed_keys, const int32_t* __restrict__ need_full) {
  __shared__ InvShared sh;
  const int qy = blockIdx.y;
  const int64_t slot = static_cast<int64_t>(qy) * gridDim.x + blockIdx.x;
  // (after sparse_inv_pruned_kernel: only the shares it left — its flags — are scanned here; the others keep its lists)
  if (need_full && !need_full[slot]) return;  // block-uniform
  const int t0 = q_off[qy];
  const int nnz = min(q_off[qy + 1] - t0, kInvMaxTerms);
  sh.lists[threadIdx.x] = 0ull;
  if (static_cast<int>(threadIdx.x) < nnz) {
    sh.t_id[threadIdx.x] = q_ids[t0 + threadIdx.x];
    sh.t_w[threadIdx.x] = q_w[t0 + threadIdx.x];
    sh.t_frac[threadIdx.x] = q_frac[t0 + threadIdx.x];
  }
  __syncthreads();
  inv_order_terms(sh, nnz);
  const uint64_t theta0 = seed_keys ? seed_keys[static_cast<int64_t>(qy) * k + (k - 1)] : 0ull;
  if (nnz > 0)  // block-uniform
    inv_scan_segments(sh, segs, n_seg, blockIdx.x, gridDim.x, keys, vals, nnz, mask, k, fw, -1, theta0);
  if (threadIdx.x < kListLen) cand[slot * kListLen + threadIdx.x] = sh.lists[threadIdx.x];
}

// ---- the pruned scan as its own kernel: no block barriers, waves walk segments on their own --------------------------
//
// With a threshold known up front (the seed pass below) a query only needs the rows that carry one of its few
// ESSENTIAL terms (see inv_scan_segments), a handful per segment — and the per-segment loop above, with its barriers
// and its 4096-row scratch, is then all latency: search, wait, mark, wait, walk, wait, sixteen segments in a row.
// Here a block is four independent waves; wave w takes the segments w, w + 4, ... of the block's share and for each of
// them finds the essential terms' runs (64-way searches), appends the rows they name to ITS candidate buffer in LDS and,
// whenever that fills and at the end, scores 64 candidates at a time from the forward index (one row per lane,
// inv_row_score: the forward scan's sum, the same bits) into its own top-k list. A row named by two essential terms
// is scored twice by the same wave; the list refuses a key it already holds. 11 KiB of LDS per block: the waves of many
// blocks overlap each other's memory round trips.
//   SEED: the essential terms are the query's rarest ones (no threshold yet): the k best rows that carry one of them
//         give theta0 = the k-th of their exact scores, a lower bound of the final k-th best score.
//   main: essential terms from theta0 and the largest |weight| of the block's segments. A block that cannot prune
//         (no seed, every term essential, or so many marked rows that adding up the postings is cheaper) sets its flag in
//         `need_full` and leaves its share to sparse_inv_batch_kernel.
constexpr int kPrunedCand = 512;  // candidate rows a wave buffers
constexpr int kPrunedHash = 1024; // slots of a wave's table of essential partial sums (a segment that names more than 3/4 of
                                  // them distinct rows passes the rest on unfiltered)
constexpr int kSeedStride = 4;    // the seed pass samples every fourth segment
constexpr int kSeedRowsPerWave = 1024;  // ... and a wave scores at most this many rows

struct InvPrunedShared {
  int32_t t_id[kInvMaxTerms];
  float t_w[kInvMaxTerms];
  float t_frac[kInvMaxTerms];
  int32_t t_ord[kInvMaxTerms];
  float t_pre[kInvMaxTerms];
  uint64_t lists[kInvWaves * kListLen];
  uint32_t cand[kInvWaves][kPrunedCand];
  // per wave and segment (main pass): the rows the essential terms name with the sum of what those terms add to them
  // (magnitudes), an open-addressing table; a row survives only if that sum plus the non-essential terms' bound reaches theta
  // (kInvWaves x kPrunedHash x {uint32 row, float sum} of dynamic LDS, main pass only: the seed pass does not pay for it)
  int32_t n_ne;
  int32_t vmax_bits;
  float theta;
};

// wave_list_insert that ignores a key the list already holds
__device__ __forceinline__ void wave_list_insert_unique(uint64_t* list, int k, uint64_t key, int lane) {
  const uint64_t cur = list[lane];
  if (__ballot(cur == key)) return;  // wave-uniform
  const int pos = __popcll(__ballot(cur > key));
  if (pos >= k) return;
  const uint64_t prev = __shfl_up(cur, 1);
  if (lane == pos) list[lane] = key;
  else if (lane > pos && lane < k) list[lane] = prev;
}

extern __shared__ uint32_t inv_pruned_dyn[];

template <bool SEED>
__global__ __launch_bounds__(kInvWaves * 64) void sparse_inv_pruned_kernel(
    const InvSeg* __restrict__ segs, int n_seg, const uint64_t* __restrict__ keys, const float* __restrict__ vals,
    const int32_t* __restrict__ q_off, const int32_t* __restrict__ q_ids, const float* __restrict__ q_w,
    const float* __restrict__ q_frac, const uint8_t* __restrict__ mask, int k, uint64_t* __restrict__ cand_out,
    const InvForward fw, float n_points, const uint64_t* __restrict__ seed_keys, int32_t* __restrict__ need_full,
    unsigned long long* __restrict__ dbg) {
  __shared__ InvPrunedShared sh;
  const int qy = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t0 = q_off[qy];
  const int nnz = min(q_off[qy + 1] - t0, kInvMaxTerms);
  const int64_t slot = static_cast<int64_t>(qy) * gridDim.x + blockIdx.x;
  sh.lists[threadIdx.x] = 0ull;
  if (static_cast<int>(threadIdx.x) < nnz) {
    sh.t_id[threadIdx.x] = q_ids[t0 + threadIdx.x];
    sh.t_w[threadIdx.x] = q_w[t0 + threadIdx.x];
    sh.t_frac[threadIdx.x] = q_frac[t0 + threadIdx.x];
  }
  if (threadIdx.x == 0) {
    sh.n_ne = -1;
    sh.vmax_bits = 0;
  }
  __syncthreads();
  if (nnz == 0) {  // block-uniform: a query without terms (or one the host serves another way)
    if (threadIdx.x < kListLen) cand_out[slot * kListLen + threadIdx.x] = 0ull;
    if (!SEED && threadIdx.x == 0) need_full[slot] = 0;
    return;
  }
  {  // sh.t_ord / sh.t_pre as inv_order_terms makes them
    if (threadIdx.x == 0) {
      for (int i = 0; i < nnz; ++i) {
        const float a = fabsf(sh.t_w[i]);
        int j = i;
        while (j > 0 && fabsf(sh.t_w[sh.t_ord[j - 1]]) > a) {
          sh.t_ord[j] = sh.t_ord[j - 1];
          --j;
        }
        sh.t_ord[j] = i;
      }
      float run = 0.0f;
      for (int j = 0; j < nnz; ++j) {
        run = (run + fabsf(sh.t_w[sh.t_ord[j]])) * 1.000001f;
        sh.t_pre[j] = run;
      }
    }
  }
  // the segments this block walks: bx, bx + gx, ... of all of them — or, for the seed pass, of every kSeedStride-th
  const int seg_step = SEED ? kSeedStride : 1;
  const int n_units = (n_seg + seg_step - 1) / seg_step;
  const int share = (n_units - static_cast<int>(blockIdx.x) + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x);
  if (!SEED) {  // the largest |weight| of this block's segments (non-negative floats order like their bits; NaN bits are larger still)
    int vb = 0;
    for (int si = threadIdx.x; si < share; si += kInvWaves * 64) vb = max(vb, segs[blockIdx.x + si * gridDim.x].pad);
    if (vb) atomicMax(&sh.vmax_bits, vb);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int ne = -1;
    if (SEED) {
      // the terms of largest |weight| (the rarest) until they are expected to name a few times k rows; a query whose
      // rarest terms are common ones is not seeded
      // (the seed pass looks at every kSeedStride-th segment only: a k-th best score among a quarter of the rows is
      // still a lower bound of the final one, a little weaker, at a quarter of the cost)
      float rows = 0.0f;
      const float want = fmaxf(4.0f * k, 64.0f);
      for (int j = nnz - 1; j >= 0; --j) {
        rows += sh.t_frac[sh.t_ord[j]] * n_points / kSeedStride;
        if (rows >= want) {
          ne = j;
          break;
        }
      }
      if (ne < 0 && rows > 0.0f) ne = 0;  // all terms together name fewer rows than wanted: all of them
      // (a common "rarest" term names far more rows than a seed needs: every wave stops after kSeedRowsPerWave of them —
      // any set of real rows gives a valid bound, a smaller set a weaker one)
    } else {
      const uint64_t theta_key = seed_keys[static_cast<int64_t>(qy) * k + (k - 1)];
      if (dbg && theta_key == 0ull) atomicAdd(dbg + 1, 1ull);  // diagnostics (VR_SPARSE_DEBUG=1): no seed
      if (theta_key != 0ull) {
        const float theta = inv_key_score(theta_key);
        const float vmax = __int_as_float(sh.vmax_bits);
        int n = 0;
        while (n < nnz && sh.t_pre[n] * vmax * 1.00001f < theta) ++n;  // (false for NaN / inf / theta <= 0)
        float ess = 0.0f, all = 0.0f;
        for (int j = 0; j < nnz; ++j) {
          const float f = sh.t_frac[sh.t_ord[j]];
          all += f;
          if (j >= n) ess += f;
        }
        // (the essential terms' postings are read and summed per row; only rows whose sum can still reach theta are scored
        // one by one. Worth it while those postings are a minor part of all of them)
        // ... and the rows they name fit a wave's table: ~600 of a segment's 4096)
        if (n > 0 && ess * 3.0f < all && ess < 0.15f) ne = n;
        if (dbg) atomicAdd(dbg + (ne >= 0 ? 0 : n == 0 ? 2 : 3), 1ull);  // pruned / every term essential / too many rows to mark
        sh.theta = theta;
      }
    }
    sh.n_ne = ne;
  }
  __syncthreads();
  const int n_ne = sh.n_ne;
  if (n_ne < 0) {  // block-uniform: nothing to do here (SEED: no seed for this query; main: the full scan takes this share)
    if (threadIdx.x < kListLen) cand_out[slot * kListLen + threadIdx.x] = 0ull;
    if (!SEED && threadIdx.x == 0) need_full[slot] = 1;
    return;
  }
  if (!SEED && threadIdx.x == 0) need_full[slot] = 0;

  uint64_t* list = sh.lists + wave * kListLen;
  uint32_t* mine = sh.cand[wave];
  int n_c = 0;     // wave-uniform
  int scored = 0;  // rows this wave has scored (the seed pass stops at kSeedRowsPerWave)
  auto flush = [&]() {
    scored += n_c;
    for (int c0 = 0; c0 < n_c; c0 += 64) {
      uint64_t key = 0ull;
      if (c0 + lane < n_c) {
        const int64_t row = mine[c0 + lane];
        if (mask[row]) {
          bool shares = false;
          const float sc = inv_row_score(sh.t_id, sh.t_w, fw, row, nnz, &shares);
          if (shares) key = topk_make_key(sc, row);
        }
      }
      const uint64_t thr = list[k - 1];
      uint64_t pending = __ballot(key != 0ull && key > thr);
      while (pending) {
        const int src = __builtin_ctzll(pending);
        wave_list_insert_unique(list, k, readlane_u64(key, src), lane);
        pending &= pending - 1;
      }
    }
    n_c = 0;
  };
  for (int si = wave; si < share; si += kInvWaves) {
    if (SEED && scored + n_c >= kSeedRowsPerWave) break;  // wave-uniform
    const InvSeg seg = segs[(blockIdx.x + si * gridDim.x) * seg_step];
    if (seg.count == 0) continue;  // wave-uniform
    const uint64_t* kp = keys + seg.off;
    const float* vp = vals + seg.off;
    const float vmax = __int_as_float(seg.pad);
    uint32_t* hrow = inv_pruned_dyn + wave * kPrunedHash;
    float* hsum = reinterpret_cast<float*>(inv_pruned_dyn + kInvWaves * kPrunedHash) + wave * kPrunedHash;
    int h_used = 0;  // wave-uniform: distinct rows in the table
    if (!SEED)
      for (int j = lane; j < kPrunedHash; j += 64) hrow[j] = 0xFFFFFFFFu;
    // what the NON-essential terms can add to any row of this segment, rounded up (main pass)
    const float ne_ub = (!SEED && n_ne > 0) ? sh.t_pre[n_ne - 1] * vmax * 1.00001f : 0.0f;
    const float theta = SEED ? -__builtin_inff() : sh.theta;
    for (int e = n_ne; e < nnz; ++e) {
      const int i = sh.t_ord[e];
      const int32_t t = sh.t_id[i];
      const float aw = fabsf(sh.t_w[i]);
      const int lb = inv_wave_lower_bound(kp, seg.count, t, lane);
      // the run is walked 4 x 64 postings at a time, all four loads in flight together (most runs end inside the first
      // round; walked 64 at a time every further 64 cost a memory round trip of their own)
      bool ended = false;
      for (int p0 = lb; !ended; p0 += 256) {
        uint64_t k2[4];
        float v2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int p = p0 + u * 64 + lane;
          const int pc = p < seg.count ? p : seg.count - 1;  // (count > 0 here; the clamp keeps the loads unconditional)
          k2[u] = kp[pc];
          v2[u] = vp[pc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (ended) break;  // wave-uniform
          const int p = p0 + u * 64 + lane;
          const bool in_run = p < seg.count && inv_term(k2[u]) == t;
          const uint32_t r12 = static_cast<uint32_t>(k2[u] & (kInvSegRows - 1));
          bool direct = SEED && in_run;  // (seed pass: every named row is scored)
          if (!SEED && in_run) {
            // add |w v| to the row's entry (the postings of ONE term name distinct rows, so the lanes of this instruction
            // hold distinct rows; two of them may still want the same empty slot: compare-and-swap decides)
            if (h_used >= kPrunedHash * 3 / 4) {
              direct = true;  // table full: unfiltered
            } else {
              uint32_t h = (r12 * 2654435761u) >> 22;  // 10 bits
              for (;;) {
                const uint32_t cur = hrow[h];
                if (cur == r12) break;
                if (cur == 0xFFFFFFFFu) {
                  const uint32_t prev = atomicCAS(&hrow[h], 0xFFFFFFFFu, r12);
                  if (prev == 0xFFFFFFFFu) {
                    hsum[h] = 0.0f;
                    break;
                  }
                  if (prev == r12) break;
                }
                h = (h + 1) & (kPrunedHash - 1);
              }
              hsum[h] += aw * fabsf(v2[u]);
            }
          }
          if (!SEED) h_used += __popcll(__ballot(in_run));  // (an upper bound of the distinct rows: a row of two terms counts twice)
          const uint64_t m = __ballot(direct);
          if (m) {
            if (n_c + 64 > kPrunedCand) flush();
            if (direct) mine[n_c + __popcll(m & ((1ull << lane) - 1ull))] = static_cast<uint32_t>(seg.row_base) + r12;
            n_c += __popcll(m);
          }
          if (__ballot(in_run) != ~0ull) ended = true;  // the run ended inside (or before) these 64 postings
          if (SEED && scored + n_c >= kSeedRowsPerWave) ended = true;  // wave-uniform
        }
      }
    }
    if (!SEED) {
      // the rows whose essential sum, with everything the non-essential terms could add, still reaches theta
      for (int j0 = 0; j0 < kPrunedHash; j0 += 64) {
        const uint32_t r12 = hrow[j0 + lane];
        const bool keep = r12 != 0xFFFFFFFFu && !(hsum[j0 + lane] * 1.00001f + ne_ub < theta);
        const uint64_t m = __ballot(keep);
        if (m) {
          if (n_c + 64 > kPrunedCand) flush();
          if (keep) mine[n_c + __popcll(m & ((1ull << lane) - 1ull))] = static_cast<uint32_t>(seg.row_base) + r12;
          n_c += __popcll(m);
        }
      }
    }
  }
  flush();
  if (dbg && lane == 0) atomicAdd(dbg + (SEED ? 5 : 4), static_cast<unsigned long long>(scored));  // rows scored one by one
  block_merge_lists(sh.lists, kListLen, kInvWaves, wave, lane);
  if (threadIdx.x < kListLen) cand_out[slot * kListLen + threadIdx.x] = sh.lists[threadIdx.x];
}

// ---- many queries per launch, GROUPED: every term's postings added up, once per group of queries -----------------------
//
// The kernels above spend a block (or a wave) per (segment share, query): a thousand queries drawn from one vocabulary walk
// the same long runs of the same common terms a thousand times, and each walk is a chain of memory round trips (search,
// postings, forward rows) with one query's worth of work behind it. Here a block takes one SEGMENT and a GROUP of G
// queries (the host sorts the batch so that queries which share their commonest terms sit together):
//   locate   one launch finds, for every DISTINCT term of the batch and every segment, the term's run of postings
//            (64-way searches, four terms x {first, past-the-last} per wave in flight together; thousands of waves in
//            flight), a second one lays the bounds out per (segment, group, union term) and fills in the weights — so a
//            scan block's first loads depend on nothing but its block index;
//   scan     the runs of the group's union of terms, compacted, are ONE stream of postings: a thread takes positions
//            tid, tid + 512, ... of it, kGrpSlices loads in flight per thread and batch, two batches in flight. A posting
//            of term u adds fl(w_g * v) to the accumulator plane of every query g of the group that carries u: G planes
//            of 4096 f32 in LDS. A slice of 512 positions is taken term by term in ascending id order, and a barrier
//            separates two terms whenever a query carries both (one term's postings name distinct rows; terms of
//            different queries touch different planes), so every (query, row) sum is built in the forward scan's order
//            from +0.0: the same bits as sparse_inv_kernel and sparse.hip.
//   select   the seed pass (sparse_inv_pruned_kernel<true>) gave every query k real rows with real scores; only rows whose
//            key reaches the k-th of them can be among the final k, and those few are appended to the query's candidate
//            buffer (an atomic counter per query); select_counted_kernel ranks each buffer. A query without a full seed
//            list offers every row it shares a term with; a buffer that overflows (kGrpCandCap keys) makes the host
//            repeat the batch on the per-query kernels — the answer never depends on the threshold being good.
// Bytes: 12 B per posting of the UNION of a group's terms per segment (served by L2 / Infinity Cache: the blocks of one
// segment run together, blockIdx.x = group), against 12 B per posting, query and term before.
constexpr int kGrpThreads = 512;
constexpr int kGrpMaxU = 64;     // union terms of a group (the host closes a group before it exceeds this): one per lane
#ifndef VR_GRP_SLICES
#define VR_GRP_SLICES 4
#endif
constexpr int kGrpSlices = VR_GRP_SLICES;  // postings per thread and batch (two batches in flight). 4, not 8: 56 registers
                                           // instead of 79 let four blocks of a two-query group share a CU
constexpr int kGrpPer = 2;       // ... of which a thread adds up this many together
constexpr int kGrpHdr = 4 + 8;   // ints per group: union size, 3 spare, the (<= 8) queries' numbers (-1: none)
constexpr int kGrpEnt = 2 + 8;   // ints per union term: slot of the term among the batch's distinct terms, query mask,
                                 // per query the index of its weight in q_w (-1: none)
constexpr int kGrpSearch = 4;    // terms a wave searches together
constexpr int kGrpCandCap = 16;  // candidate keys per (query, segment), at least

// first posting of term t[i] and first posting past it, for kGrpSearch terms at once: 2 kGrpSearch independent 64-way
// searches whose probes are in flight together (the steps of inv_wave_lower_bound, same invariants)
__device__ __forceinline__ void inv_wave_runs(const uint64_t* __restrict__ kp, int count, const int32_t (&t)[kGrpSearch],
                                              int lane, int (&first)[kGrpSearch], int (&past)[kGrpSearch]) {
  constexpr int C = 2 * kGrpSearch;
  int lo[C], hi[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    lo[c] = 0;
    hi[c] = count;
  }
  for (;;) {
    int widest = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) widest = max(widest, hi[c] - lo[c]);
    if (widest <= 64) break;  // wave-uniform
    uint64_t key[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int step = (hi[c] - lo[c] + 63) >> 6;
      key[c] = kp[min(lo[c] + lane * step, count - 1)];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int step = (hi[c] - lo[c] + 63) >> 6;
      const int p = lo[c] + lane * step;
      const int32_t term = inv_term(key[c]);
      const int32_t tt = t[c % kGrpSearch];
      const bool less = p < hi[c] && (c < kGrpSearch ? term < tt : term <= tt);
      const int n = __popcll(__ballot(less));
      const int nlo = n ? lo[c] + (n - 1) * step + 1 : lo[c];
      const int nhi = n < 64 ? min(hi[c], lo[c] + n * step) : hi[c];
      lo[c] = nlo;
      hi[c] = nhi;
    }
  }
  uint64_t key[C];
#pragma unroll
  for (int c = 0; c < C; ++c) key[c] = kp[min(lo[c] + lane, count - 1)];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int p = lo[c] + lane;
    const int32_t term = inv_term(key[c]);
    const int32_t tt = t[c % kGrpSearch];
    const bool less = p < hi[c] && (c < kGrpSearch ? term < tt : term <= tt);
    const int res = lo[c] + __popcll(__ballot(less));
    if (c < kGrpSearch) first[c] = res;
    else past[c - kGrpSearch] = res;
  }
}

// bounds[seg][slot] = (first posting, postings) of the batch's distinct term `slot` in segment `seg` (relative to the segment)
__global__ __launch_bounds__(256) void sparse_inv_locate_kernel(const InvSeg* __restrict__ segs,
                                                                const uint64_t* __restrict__ keys,
                                                                const int32_t* __restrict__ slot_terms, int n_slots,
                                                                int2* __restrict__ bounds) {
  const int lane = threadIdx.x & 63;
  const int s0 = (static_cast<int>(blockIdx.x) * 4 + static_cast<int>(threadIdx.x >> 6)) * kGrpSearch;
  if (s0 >= n_slots) return;  // wave-uniform
  const InvSeg seg = segs[blockIdx.y];
  int2* out = bounds + static_cast<int64_t>(blockIdx.y) * n_slots + s0;
  int first[kGrpSearch] = {0, 0, 0, 0}, past[kGrpSearch] = {0, 0, 0, 0};
  if (seg.count > 0) {
    int32_t t[kGrpSearch];
#pragma unroll
    for (int i = 0; i < kGrpSearch; ++i) t[i] = slot_terms[min(s0 + i, n_slots - 1)];
    inv_wave_runs(keys + seg.off, seg.count, t, lane, first, past);
  }
#pragma unroll
  for (int i = 0; i < kGrpSearch; ++i)
    if (lane == i && s0 + i < n_slots) out[i] = make_int2(first[i], past[i] - first[i]);
}

// per (segment, group, union term): the bounds of its slot; and (blockIdx.y == 0) per (group, union term, query) the weight
__global__ __launch_bounds__(256) void sparse_inv_expand_kernel(const int2* __restrict__ bounds, int n_slots,
                                                                const int32_t* __restrict__ ent, int n_gu, int n_seg,
                                                                const float* __restrict__ q_w, int2* __restrict__ out,
                                                                float* __restrict__ ent_w) {
  const int gu = blockIdx.x * 256 + threadIdx.x;
  if (gu >= n_gu) return;
  const int32_t slot = ent[static_cast<int64_t>(gu) * kGrpEnt];
  for (int s = blockIdx.y; s < n_seg; s += gridDim.y)
    out[static_cast<int64_t>(s) * n_gu + gu] = slot >= 0 ? bounds[static_cast<int64_t>(s) * n_slots + slot] : make_int2(0, 0);
  if (blockIdx.y == 0) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int32_t wi = ent[static_cast<int64_t>(gu) * kGrpEnt + 2 + g];
      ent_w[static_cast<int64_t>(gu) * 8 + g] = wi >= 0 ? q_w[wi] : 0.0f;
    }
  }
}

// theta[group][g] = the threshold key of query g of the group: the k-th key of its seed list
__global__ void sparse_inv_theta_kernel(const int32_t* __restrict__ hdr, int n_groups, const uint64_t* __restrict__ seed_keys,
                                        int k, uint64_t* __restrict__ theta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_groups * 8) return;
  const int32_t q = hdr[static_cast<int64_t>(i / 8) * kGrpHdr + 4 + i % 8];
  theta[i] = q >= 0 ? seed_keys[static_cast<int64_t>(q) * k + (k - 1)] : ~0ull;
}

// "no term of the query names this row yet": an accumulator's initial bits. A NaN no arithmetic produces (results carry
// the canonical quiet NaN or an operand's payload; a caller would have to store a weight with exactly these bits).
constexpr uint32_t kGrpUntouched = 0xFFFFFFFFu;

// SAMPLE: the block takes segment blockIdx.y * seg_stride and, instead of the rows that reach a threshold, leaves the best
// key of each of its waves' 512 rows per query (8 real rows with real scores per (query, sampled segment)): the k-th best
// of a query's sample is the threshold of the full pass.
template <int G, bool SAMPLE>
__global__ __launch_bounds__(kGrpThreads) void sparse_inv_group_kernel(
    const InvSeg* __restrict__ segs, const uint64_t* __restrict__ keys, const float* __restrict__ vals,
    const int32_t* __restrict__ grp_hdr, const int32_t* __restrict__ grp_ent, const float* __restrict__ ent_w,
    const int2* __restrict__ bounds, int stride_u, const uint8_t* __restrict__ mask, const uint64_t* __restrict__ theta,
    uint64_t* __restrict__ cand, int32_t* __restrict__ cnt, int cap, int dbg_mode, int seg_stride,
    uint64_t* __restrict__ spill, int32_t* __restrict__ spill_cnt, int spill_cap, int sample_tail, int n_seg_all) {
  static_assert(G >= 1 && G <= 8, "group size");
  __shared__ __align__(16) float acc[G * kInvSegRows];
  __shared__ int32_t s_mask[kGrpMaxU];         // by compacted run: the queries that carry its term
  __shared__ int32_t s_lo[kGrpMaxU];           //                   its first posting
  __shared__ int32_t s_pre[kGrpMaxU + 1];      //                   its first position in the stream (s_pre[n_c] = the total)
  __shared__ float s_w[kGrpMaxU * G];          //                   the queries' weights
  __shared__ uint64_t s_theta[G];
  __shared__ int32_t s_q[G];
  __shared__ int32_t s_nc;
  __shared__ int32_t s_cnt[G];
  const int tid = threadIdx.x, lane = tid & 63;
  const int grp = blockIdx.x;
  // (SAMPLE: every seg_stride-th segment, then the last sample_tail segments — the rows stored last often differ from
  // the bulk, and a threshold that has not seen them lets all of them through)
  const int n_regular = static_cast<int>(gridDim.y) - sample_tail;
  const int seg_no = !SAMPLE ? static_cast<int>(blockIdx.y)
                             : static_cast<int>(blockIdx.y) < n_regular ? static_cast<int>(blockIdx.y) * seg_stride
                                                                        : n_seg_all - (static_cast<int>(gridDim.y) - static_cast<int>(blockIdx.y));
  const InvSeg seg = segs[seg_no];
  const int64_t gu0 = static_cast<int64_t>(grp) * stride_u;
  // the filter bytes of the 8 rows this thread will look at when the sums are complete (rows past the segment: 0)
  uint64_t pass8 = 0ull;
  {
    const int r0 = tid * 8;
    const uint8_t* mp = mask + seg.row_base + r0;
    if (r0 + 8 <= seg.nrows && (reinterpret_cast<uintptr_t>(mp) & 7u) == 0) {
      pass8 = *reinterpret_cast<const uint64_t*>(mp);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (r0 + j < seg.nrows) pass8 |= static_cast<uint64_t>(mp[j]) << (8 * j);
    }
  }
  if (tid < G) s_cnt[tid] = 0;
  if (tid < 64) {  // wave 0: the group's union terms, one per lane; the runs that are not empty, compacted
    const bool in = lane < stride_u;
    const int2 b = in ? bounds[(static_cast<int64_t>(seg_no) * gridDim.x + grp) * stride_u + lane] : make_int2(0, 0);
    const int32_t m = in ? grp_ent[(gu0 + lane) * kGrpEnt + 1] & ((1 << G) - 1) : 0;
    float w[G];
#pragma unroll
    for (int g = 0; g < G; ++g) w[g] = in ? ent_w[(gu0 + lane) * 8 + g] : 0.0f;
    const int len = m ? b.y : 0;
    int incl = len;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    const uint64_t live = __ballot(len > 0);
    const int ci = __popcll(live & ((1ull << lane) - 1ull));
    if (len > 0) {
      s_mask[ci] = m;
      s_lo[ci] = b.x;
      s_pre[ci] = incl - len;
#pragma unroll
      for (int g = 0; g < G; ++g) s_w[ci * G + g] = w[g];
    }
    if (lane == 63) {
      s_pre[__popcll(live)] = incl;
      s_nc = __popcll(live);
    }
  } else if (tid < 64 + G) {
    const int g = tid - 64;
    s_q[g] = grp_hdr[static_cast<int64_t>(grp) * kGrpHdr + 4 + g];
    s_theta[g] = SAMPLE ? 0ull : theta[static_cast<int64_t>(grp) * 8 + g];
  }
  {
    uint4* a4 = reinterpret_cast<uint4*>(acc);
    for (int i = tid; i < G * kInvSegRows / 4; i += kGrpThreads)
      a4[i] = make_uint4(kGrpUntouched, kGrpUntouched, kGrpUntouched, kGrpUntouched);
  }
  __syncthreads();
  const int n_c = s_nc;
  const int total = s_pre[n_c];
  if (total == 0 || (dbg_mode & 4)) return;  // block-uniform: no term of the group occurs in this segment
  const uint64_t* kp = keys + seg.off;
  const float* vp = vals + seg.off;
  // the run table once more, in registers: lane c of every wave holds run c (block-uniform reads are v_readlane then,
  // not LDS round trips)
  const int r_pre = lane <= n_c ? s_pre[lane] : 0x7FFFFFFF;
  const int r_mask = lane < n_c ? s_mask[lane] : 0;
  float r_w[G];
#pragma unroll
  for (int g = 0; g < G; ++g) r_w[g] = lane < n_c ? s_w[lane * G + g] : 0.0f;
  // lane 64 does not exist: the total is kept aside for n_c == 64
  auto pre_of = [&](int c) { return c >= 64 ? total : __builtin_amdgcn_readlane(r_pre, c); };

  // the stream of postings, kGrpSlices x 512 positions per batch; batch b + 1 is requested before batch b is added up
  constexpr int kBatch = kGrpSlices * kGrpThreads;
  uint32_t bk[2][kGrpSlices];  // (the low half of a posting's key: the row sits in its low 12 bits)
  float bv[2][kGrpSlices];
  int bc[2][kGrpSlices];  // the compacted run a position belongs to
  int pc = 0;             // this thread's walk through s_pre (positions only grow)
  int pend = s_pre[1];    // ... the end of run pc in the stream, and what turns a position of it into a posting
  int pdelta = s_lo[0] - s_pre[0];
  auto request = [&](int set, int base) {
#pragma unroll
    for (int j = 0; j < kGrpSlices; ++j) {
      const int i = base + j * kGrpThreads + tid;
      const int ic = i < total ? i : total - 1;
      while (ic >= pend) {
        ++pc;
        pend = s_pre[pc + 1];
        pdelta = s_lo[pc] - s_pre[pc];
      }
      bc[set][j] = pc;
      bk[set][j] = reinterpret_cast<const uint32_t*>(kp + (ic + pdelta))[0];
      bv[set][j] = vp[ic + pdelta];
    }
  };
  unsigned dirty = 0u;  // queries whose planes have been written since the last barrier
  int su = 0;           // block-uniform walk through the runs
  // a batch is added up kGrpPer x 512 positions at a time (a thread holds kGrpPer of them, 512 apart): the planes' sums of
  // all its rows are requested together, then the runs the positions belong to are taken in ascending term id
  auto add_up = [&](int set, int base) {
#pragma unroll
    for (int j0 = 0; j0 < kGrpSlices; j0 += kGrpPer) {
      const int s0 = base + j0 * kGrpThreads;
      if (s0 >= total) break;  // block-uniform
      const int s1 = min(s0 + kGrpPer * kGrpThreads, total);
      while (s0 >= pre_of(su + 1)) ++su;
      int r[kGrpPer], mine[kGrpPer];
      float val[kGrpPer], a[kGrpPer][G];
#pragma unroll
      for (int v = 0; v < kGrpPer; ++v) {
        const int i = s0 + v * kGrpThreads + tid;
        r[v] = static_cast<int>(bk[set][j0 + v] & (kInvSegRows - 1));
        val[v] = bv[set][j0 + v];
        mine[v] = (i < total && !(dbg_mode & 1)) ? bc[set][j0 + v] : -1;
        if ((dbg_mode & 9) && bk[set][j0 + v] == 0xFFFFFFFFu && val[v] == 1.25f) acc[0] = val[v];  // (diagnostics: keeps the loads alive)
      }
      if (dbg_mode & 8) continue;
      // every plane's sum of these rows, requested together (a read the posting may not need costs nothing but LDS
      // bandwidth; the writes below are conditional — another term's thread may own the planes this term does not touch)
#pragma unroll
      for (int v = 0; v < kGrpPer; ++v)
#pragma unroll
        for (int g = 0; g < G; ++g) a[v][g] = acc[g * kInvSegRows + r[v]];
      for (int c = su; c < n_c; ++c) {  // block-uniform: the runs these positions touch, ascending term id
        const int c_pre = pre_of(c);
        if (c_pre >= s1) break;
        const unsigned m = static_cast<unsigned>(__builtin_amdgcn_readlane(r_mask, c));
        if (c_pre >= s0) {  // (a run that began earlier has been through this: its own postings name distinct rows)
          if (m & dirty) {  // a query of this term has been added to since the last barrier: its earlier term comes first
            __syncthreads();
            dirty = 0u;
#pragma unroll
            for (int v = 0; v < kGrpPer; ++v)
              if (mine[v] >= c) {  // (the sums read above may be stale now)
#pragma unroll
                for (int g = 0; g < G; ++g) a[v][g] = acc[g * kInvSegRows + r[v]];
              }
          }
          dirty |= m;
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
          if (m & (1u << g)) {  // block-uniform
            const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r_w[g]), c));
#pragma unroll
            for (int v = 0; v < kGrpPer; ++v)
              if (mine[v] == c) {
                const float from = __float_as_uint(a[v][g]) == kGrpUntouched ? 0.0f : a[v][g];
                acc[g * kInvSegRows + r[v]] = __fadd_rn(from, __fmul_rn(w, val[v]));
              }
          }
      }
    }
  };
  request(0, 0);
  for (int base = 0; base < total; base += 2 * kBatch) {  // block-uniform
    if (base + kBatch < total) request(1, base + kBatch);
    add_up(0, base);
    if (base + kBatch >= total) break;
    if (base + 2 * kBatch < total) request(0, base + 2 * kBatch);
    add_up(1, base + kBatch);
  }
  __syncthreads();
  if (dbg_mode & 2) return;

  // The rows that reach their query's threshold key (the seed's k-th: k real rows are at or above it) go to the
  // (query, segment) region of the candidate array — this block's own, so a slot costs an LDS atomic and the block ends
  // without a global round trip (the rows' filter bytes were requested in the prologue). A thread takes 8 consecutive rows.
  {
    const int r0 = tid * 8;
    uint32_t sc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const uint4 lo4 = *reinterpret_cast<const uint4*>(acc + g * kInvSegRows + r0);
      const uint4 hi4 = *reinterpret_cast<const uint4*>(acc + g * kInvSegRows + r0 + 4);
      sc[g][0] = lo4.x, sc[g][1] = lo4.y, sc[g][2] = lo4.z, sc[g][3] = lo4.w;
      sc[g][4] = hi4.x, sc[g][5] = hi4.y, sc[g][6] = hi4.z, sc[g][7] = hi4.w;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int32_t q = s_q[g];
      if (q < 0) continue;  // block-uniform
      const uint64_t th = s_theta[g];
      const uint32_t th_hi = static_cast<uint32_t>(th >> 32);
      uint64_t* region = cand + (static_cast<int64_t>(q) * gridDim.y + blockIdx.y) * cap;
      uint64_t best = 0ull;
      if (!SAMPLE && th_hi >= 0x80000000u) {
        // a threshold >= +0.0 (BM25 scores are positive): a row reaches it iff its score's bits, read as a signed integer,
        // reach the threshold's (negative scores and the untouched pattern are negative integers) — the largest of the
        // eight settles nearly every thread in four instructions (the kernel is bound by its instruction count)
        const int m = max(max(max(static_cast<int>(sc[g][0]), static_cast<int>(sc[g][1])), max(static_cast<int>(sc[g][2]), static_cast<int>(sc[g][3]))),
                          max(max(static_cast<int>(sc[g][4]), static_cast<int>(sc[g][5])), max(static_cast<int>(sc[g][6]), static_cast<int>(sc[g][7]))));
        if (m < static_cast<int>(th_hi & 0x7FFFFFFFu)) continue;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t bits = sc[g][j];
        if (bits == kGrpUntouched || !((pass8 >> (8 * j)) & 0xFFull)) continue;
        const uint32_t u = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);  // (the key's upper half, topk_make_key)
        if (u < th_hi) continue;
        const uint64_t key = topk_make_key(__uint_as_float(bits), static_cast<int64_t>(seg.row_base) + r0 + j);
        if (key == 0ull || key < th) continue;
        if (SAMPLE) {
          best = key > best ? key : best;
        } else {
          const int slot = atomicAdd(&s_cnt[g], 1);
          if (slot < cap) {
            region[slot] = key;
          } else {  // (rare: the candidates of a query crowd into one segment) the query's spill area, a global counter
            const int at = atomicAdd(&spill_cnt[q], 1);
            if (at < spill_cap) spill[static_cast<int64_t>(q) * spill_cap + at] = key;
          }
        }
      }
      if (SAMPLE) {  // the wave's best key (cap >= 8: one slot per wave)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          const uint64_t o = __shfl_xor(best, off);
          best = o > best ? o : best;
        }
        if (lane == 0) region[tid >> 6] = best;
      }
    }
  }
  if (SAMPLE) {
    if (tid < G && s_q[tid] >= 0) cnt[static_cast<int64_t>(s_q[tid]) * gridDim.y + blockIdx.y] = kGrpThreads / 64;
    return;
  }
  __syncthreads();
  if (tid < G && s_q[tid] >= 0) cnt[static_cast<int64_t>(s_q[tid]) * gridDim.y + blockIdx.y] = s_cnt[tid];
}

// The group tables of a batch (host): queries are ordered so that those which share their commonest terms (by the
// number of queries of THIS batch that carry a term) are neighbours, then cut into groups of <= G queries whose union of
// terms stays <= kGrpMaxU. tables: [n_groups x kGrpHdr][n_groups x stride_u x kGrpEnt][n_slots distinct terms, ascending].
struct GroupLayout {
  int n_groups = 0, stride_u = 0, n_slots = 0;
  int64_t ent_off = 0, slot_off = 0;
};
static GroupLayout inv_group_queries(const int32_t* off, const int32_t* ids, int nq, int G, std::vector<int32_t>* out) {
  std::unordered_map<int32_t, int32_t> freq;
  freq.reserve(static_cast<size_t>(off[nq]) * 2 + 16);
  for (int32_t i = off[0]; i < off[nq]; ++i) ++freq[ids[i]];
  std::vector<int32_t> slots;
  slots.reserve(freq.size());
  for (const auto& kv : freq) slots.push_back(kv.first);
  std::sort(slots.begin(), slots.end());
  struct Q {
    int32_t q;
    int32_t key[6];  // (-frequency, id) of its three commonest terms
  };
  std::vector<Q> order;
  order.reserve(static_cast<size_t>(nq));
  std::vector<std::pair<int32_t, int32_t>> t;
  for (int q = 0; q < nq; ++q) {
    const int n = off[q + 1] - off[q];
    if (n <= 0) continue;
    t.clear();
    for (int j = 0; j < n; ++j) t.emplace_back(-freq[ids[off[q] + j]], ids[off[q] + j]);
    const size_t top = std::min<size_t>(3, t.size());
    std::partial_sort(t.begin(), t.begin() + static_cast<std::ptrdiff_t>(top), t.end());
    Q e{q, {0, 0, 0, 0, 0, 0}};
    for (size_t j = 0; j < top; ++j) {
      e.key[2 * j] = t[j].first;
      e.key[2 * j + 1] = t[j].second;
    }
    order.push_back(e);
  }
  std::sort(order.begin(), order.end(), [](const Q& a, const Q& b) {
    for (int j = 0; j < 6; ++j)
      if (a.key[j] != b.key[j]) return a.key[j] < b.key[j];
    return a.q < b.q;
  });
  // the groups: members and their union of terms
  std::vector<std::vector<int32_t>> groups;
  std::vector<int32_t> members, uni, merged;
  size_t widest = 0;
  for (const Q& e : order) {
    const int32_t* a = ids + off[e.q];
    const int n = off[e.q + 1] - off[e.q];
    merged.clear();
    std::set_union(uni.begin(), uni.end(), a, a + n, std::back_inserter(merged));
    if (!members.empty() && (static_cast<int>(members.size()) >= G || static_cast<int>(merged.size()) > kGrpMaxU)) {
      widest = std::max(widest, uni.size());
      groups.push_back(members);
      members.clear();
      merged.assign(a, a + n);
    }
    uni.swap(merged);
    members.push_back(e.q);
  }
  if (!members.empty()) {
    widest = std::max(widest, uni.size());
    groups.push_back(members);
  }
  GroupLayout lay;
  lay.n_groups = static_cast<int>(groups.size());
  lay.stride_u = static_cast<int>(std::min<size_t>(kGrpMaxU, (widest + 7) / 8 * 8));
  lay.n_slots = static_cast<int>(slots.size());
  lay.ent_off = static_cast<int64_t>(lay.n_groups) * kGrpHdr;
  lay.slot_off = lay.ent_off + static_cast<int64_t>(lay.n_groups) * lay.stride_u * kGrpEnt;
  out->assign(static_cast<size_t>(lay.slot_off) + slots.size(), -1);
  int32_t* hdr = out->data();
  int32_t* ent = out->data() + lay.ent_off;
  std::copy(slots.begin(), slots.end(), out->begin() + static_cast<std::ptrdiff_t>(lay.slot_off));
  struct Trip {
    int32_t term, g, widx;
  };
  std::vector<Trip> trips;
  for (size_t gi = 0; gi < groups.size(); ++gi) {
    trips.clear();
    for (size_t g = 0; g < groups[gi].size(); ++g) {
      const int q = groups[gi][g];
      for (int32_t i = off[q]; i < off[q + 1]; ++i) trips.push_back(Trip{ids[i], static_cast<int32_t>(g), i});
      hdr[gi * kGrpHdr + 4 + g] = q;
    }
    std::sort(trips.begin(), trips.end(), [](const Trip& a, const Trip& b) { return a.term != b.term ? a.term < b.term : a.g < b.g; });
    int32_t n_u = 0;
    for (size_t i = 0; i < trips.size();) {
      int32_t* row = ent + (gi * static_cast<size_t>(lay.stride_u) + static_cast<size_t>(n_u)) * kGrpEnt;
      row[0] = static_cast<int32_t>(std::lower_bound(slots.begin(), slots.end(), trips[i].term) - slots.begin());
      row[1] = 0;
      size_t j = i;
      for (; j < trips.size() && trips[j].term == trips[i].term; ++j) {
        row[1] |= 1 << trips[j].g;
        row[2 + trips[j].g] = trips[j].widx;
      }
      ++n_u;
      i = j;
    }
    hdr[gi * kGrpHdr] = n_u;
    // (unused union entries keep slot -1 and, below, mask 0)
    for (int u = n_u; u < lay.stride_u; ++u) ent[(gi * static_cast<size_t>(lay.stride_u) + static_cast<size_t>(u)) * kGrpEnt + 1] = 0;
  }
  return lay;
}

int inv_scan_topk(vr_engine* e, const int32_t* q_idx_host, const float* q_val_host, int nnz, bool weights_given,
                  float n_points, const uint8_t* mask_dev, int k, uint64_t* out_keys_dev) {
  VR_CHECK(nnz >= 1 && nnz <= kInvMaxTerms && k >= 1 && k <= kListLen, "bad inverted-scan shape");
  InvQuery query;
  for (int i = 0; i < kInvMaxTerms; ++i) {
    query.id[i] = i < nnz ? q_idx_host[i] : 0;
    query.val[i] = i < nnz ? q_val_host[i] : 0.0f;
  }
  const int blocks = static_cast<int>(std::min<int64_t>(e->n_inv_seg, kScanBlocks));
  VR_TRY(e->sp_cand.grow(static_cast<int64_t>(blocks) * kListLen, 0, e->stream));
  // work 0: the postings the query's terms own are counted nowhere on the host (the slot's time and launch count
  // are what bench.py reads for this scan)
  prof_begin(e, VR_PROF_SPARSE_SCAN, 0.0);
  hipLaunchKernelGGL(sparse_inv_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kInvWaves * 64), 0, e->stream,
                     e->inv_seg.p, static_cast<int>(e->n_inv_seg), e->inv_key.p, e->inv_val.p, query, nnz,
                     weights_given ? 1 : 0, e->df_keys.p, e->df_cnt.p, e->df_cap, n_points, mask_dev, k,
                     e->sp_cand.p, InvForward{e->row_slice.p, e->slices.p, e->sp_idx.p, e->sp_val.p});
  prof_end(e);
  VR_HIP(hipGetLastError());
  return topk_merge_lists(e, e->sp_cand.p, blocks, 1, k, out_keys_dev);
}

// The grouped scan of a batch (see sparse_inv_group_kernel). out_keys_dev: in — the seed lists (sampled == false), out — the
// nq x k result keys. *done = false: no query of the batch has terms (nothing was launched). Closes the profiler slot.
static int inv_scan_grouped(vr_engine* e, const float* q_w_dev, int nq, const uint8_t* mask_dev, int k, uint64_t* out_keys_dev,
                            const int32_t* q_off_host, const int32_t* q_ids_host, bool sampled, bool* done) {
  // (2: 34 KB of LDS and 56 registers — four blocks = 32 waves per CU, the most a CU holds; measured best: the block is a
  // chain of dependent phases and other blocks are what hides them — 2.5 ms of kernels per 1000 queries against 3.0 for
  // groups of 4 at two blocks per CU, although those read fewer postings)
  const int want_group = std::getenv("VR_SPARSE_GROUP") ? atoi(std::getenv("VR_SPARSE_GROUP")) : 2;
  const int group_size = want_group == 3 || want_group == 4 || want_group == 8 ? want_group : 2;
  const int dbg_mode = std::getenv("VR_SPARSE_GROUP_DBG") ? atoi(std::getenv("VR_SPARSE_GROUP_DBG")) : 0;  // timing experiments
  const int64_t n_seg = e->n_inv_seg;
  // keys per (query, segment) region: 32 at a million rows (250 segments), more while the segments are few (a region may
  // then hold most of a query's k best), <= 8192 per query in all
  const int fit_cap = static_cast<int>(std::max<int64_t>(kGrpCandCap, std::min<int64_t>(512, 8192 / std::max<int64_t>(n_seg, 1))));
  const int cand_cap = std::getenv("VR_SPARSE_GROUP_CAP") ? std::min(fit_cap, std::max(1, atoi(std::getenv("VR_SPARSE_GROUP_CAP")))) : fit_cap;
  const GroupLayout lay = inv_group_queries(q_off_host, q_ids_host, nq, group_size, &e->sq_grp_host);
  *done = lay.n_groups > 0;
  if (!*done) return 0;
  const int64_t n_gu = static_cast<int64_t>(lay.n_groups) * lay.stride_u;
  // the sample: every seg_stride-th segment, 8 keys (one per wave of the block) per (query, sampled segment)
  const int seg_stride = sampled ? static_cast<int>(n_seg / 16) : 1;
  constexpr int kSampleTail = 4;  // ... and the last segments (the rows stored last)
  const int64_t n_samp = sampled ? (n_seg - kSampleTail + seg_stride - 1) / seg_stride + kSampleTail : 0;
  constexpr int kSampCap = kGrpThreads / 64;
  const int64_t cand_main = static_cast<int64_t>(nq) * n_seg * cand_cap, cand_samp = static_cast<int64_t>(nq) * n_samp * kSampCap;
  // (+ the keys ranked, + a flag per query whose candidates overflowed, + the fill of its spill area)
  const int64_t cnt_main = static_cast<int64_t>(nq) * n_seg + 1 + 2 * static_cast<int64_t>(nq), cnt_samp = static_cast<int64_t>(nq) * n_samp;
  // keys per query that did not fit their (query, segment) region (VR_SPARSE_GROUP_SPILL: tests shrink it to force the redo)
  const int kSpillCap = std::getenv("VR_SPARSE_GROUP_SPILL") ? std::min(4096, std::max(1, atoi(std::getenv("VR_SPARSE_GROUP_SPILL")))) : 4096;
  VR_TRY(e->sq_grp.grow(static_cast<int64_t>(e->sq_grp_host.size()), 0, e->stream));
  VR_TRY(e->sq_cnt.grow(cnt_main + cnt_samp, 0, e->stream));
  VR_TRY(e->sq_cand.grow(cand_main + cand_samp + static_cast<int64_t>(lay.n_groups) * 8 + static_cast<int64_t>(nq) * kSpillCap, 0, e->stream));
  // run bounds per (segment, distinct term), then per (segment, group, union term); the weights per (group, union term, query)
  VR_TRY(e->sq_bounds.grow(2 * (n_seg * lay.n_slots + n_seg * n_gu), 0, e->stream));
  VR_TRY(e->sq_entw.grow(n_gu * 8, 0, e->stream));
  int2* slot_bounds = reinterpret_cast<int2*>(e->sq_bounds.p);
  int2* grp_bounds = slot_bounds + n_seg * lay.n_slots;
  uint64_t* samp_cand = e->sq_cand.p + cand_main;
  uint64_t* theta = samp_cand + cand_samp;
  uint64_t* spill = theta + static_cast<int64_t>(lay.n_groups) * 8;
  int32_t* samp_cnt = e->sq_cnt.p + cnt_main;
  int32_t* ranked = e->sq_cnt.p + static_cast<int64_t>(nq) * n_seg;
  int32_t* overflow_q = ranked + 1;
  int32_t* spill_cnt = overflow_q + nq;
  const int32_t* hdr = e->sq_grp.p;
  const int32_t* ent = e->sq_grp.p + lay.ent_off;
  VR_HIP(hipMemcpyAsync(e->sq_grp.p, e->sq_grp_host.data(), sizeof(int32_t) * e->sq_grp_host.size(), hipMemcpyHostToDevice, e->stream));
  VR_HIP(hipMemsetAsync(e->sq_cnt.p, 0, sizeof(int32_t) * static_cast<size_t>(cnt_main + cnt_samp), e->stream));
  hipLaunchKernelGGL(sparse_inv_locate_kernel,
                     dim3(static_cast<unsigned>((lay.n_slots + 4 * kGrpSearch - 1) / (4 * kGrpSearch)), static_cast<unsigned>(n_seg)),
                     dim3(256), 0, e->stream, e->inv_seg.p, e->inv_key.p, e->sq_grp.p + lay.slot_off, lay.n_slots, slot_bounds);
  hipLaunchKernelGGL(sparse_inv_expand_kernel,
                     dim3(static_cast<unsigned>((n_gu + 255) / 256), static_cast<unsigned>(std::min<int64_t>(n_seg, 64))), dim3(256), 0,
                     e->stream, slot_bounds, lay.n_slots, ent, static_cast<int>(n_gu), static_cast<int>(n_seg), q_w_dev, grp_bounds,
                     e->sq_entw.p);
  auto launch = [&](auto kernel, int64_t rows_y, uint64_t* cand, int32_t* cnt, int cap, int stride) {
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(lay.n_groups), static_cast<unsigned>(rows_y)), dim3(kGrpThreads), 0, e->stream,
                       e->inv_seg.p, e->inv_key.p, e->inv_val.p, hdr, ent, e->sq_entw.p, grp_bounds, lay.stride_u, mask_dev, theta, cand,
                       cnt, cap, dbg_mode, stride, spill, spill_cnt, kSpillCap, kSampleTail, static_cast<int>(n_seg));
  };
  if (sampled) {
    // thresholds: the sampled segments scanned in full, the best key of every 512 rows kept; the k-th best of a query's
    // sample (real rows, real scores) is a lower bound of its final k-th best key
    if (group_size == 2) launch(sparse_inv_group_kernel<2, true>, n_samp, samp_cand, samp_cnt, kSampCap, seg_stride);
    else if (group_size == 3) launch(sparse_inv_group_kernel<3, true>, n_samp, samp_cand, samp_cnt, kSampCap, seg_stride);
    else if (group_size == 4) launch(sparse_inv_group_kernel<4, true>, n_samp, samp_cand, samp_cnt, kSampCap, seg_stride);
    else launch(sparse_inv_group_kernel<8, true>, n_samp, samp_cand, samp_cnt, kSampCap, seg_stride);
    VR_TRY(topk_select_regions(e, samp_cand, static_cast<int>(n_samp), kSampCap, samp_cnt, nullptr, 0, nullptr, nq, k, out_keys_dev,
                               pin_dev<int32_t>(e, kPinSparseOverflow), nullptr, nullptr));
  }
  hipLaunchKernelGGL(sparse_inv_theta_kernel, dim3(static_cast<unsigned>((lay.n_groups * 8 + 255) / 256)), dim3(256), 0, e->stream, hdr,
                     lay.n_groups, out_keys_dev, k, theta);
  if (group_size == 2) launch(sparse_inv_group_kernel<2, false>, n_seg, e->sq_cand.p, e->sq_cnt.p, cand_cap, 1);
  else if (group_size == 3) launch(sparse_inv_group_kernel<3, false>, n_seg, e->sq_cand.p, e->sq_cnt.p, cand_cap, 1);
  else if (group_size == 4) launch(sparse_inv_group_kernel<4, false>, n_seg, e->sq_cand.p, e->sq_cnt.p, cand_cap, 1);
  else launch(sparse_inv_group_kernel<8, false>, n_seg, e->sq_cand.p, e->sq_cnt.p, cand_cap, 1);
  prof_end(e);
  VR_HIP(hipGetLastError());
  e->stat_sparse_grouped += nq;
  e->sq_overflow_q = overflow_q;
  VR_TRY(topk_select_regions(e, e->sq_cand.p, static_cast<int>(n_seg), cand_cap, e->sq_cnt.p, spill, kSpillCap, spill_cnt, nq, k,
                             out_keys_dev, pin_dev<int32_t>(e, kPinSparseOverflow), ranked, overflow_q));
  if (std::getenv("VR_SPARSE_GROUP_DEBUG") && atoi(std::getenv("VR_SPARSE_GROUP_DEBUG")) != 0) {  // diagnostics: what filled the regions
    std::vector<int32_t> h(static_cast<size_t>(cnt_main));
    std::vector<uint64_t> th(static_cast<size_t>(lay.n_groups) * 8);
    VR_HIP(hipMemcpyAsync(h.data(), e->sq_cnt.p, sizeof(int32_t) * h.size(), hipMemcpyDeviceToHost, e->stream));
    VR_HIP(hipMemcpyAsync(th.data(), theta, sizeof(uint64_t) * th.size(), hipMemcpyDeviceToHost, e->stream));
    VR_HIP(hipStreamSynchronize(e->stream));
    int64_t over_regions = 0, max_region = 0, max_total = 0, over_q = 0, no_theta = 0, n_members = 0, max_spill = 0;
    int worst_q = -1, worst_seg = -1;
    for (int q = 0; q < nq; ++q) {
      int64_t tot = 0;
      for (int64_t s2 = 0; s2 < n_seg; ++s2) {
        const int32_t c = h[static_cast<size_t>(q * n_seg + s2)];
        tot += c;
        over_regions += c > cand_cap;
        if (c > max_region) max_region = c, worst_q = q, worst_seg = static_cast<int>(s2);
      }
      max_total = std::max(max_total, tot);
      over_q += h[static_cast<size_t>(nq) * n_seg + 1 + q];
      max_spill = std::max<int64_t>(max_spill, h[static_cast<size_t>(nq) * n_seg + 1 + nq + q]);
    }
    for (size_t i = 0; i < th.size(); ++i)
      if (th[i] != ~0ull) {
        ++n_members;
        no_theta += th[i] == 0ull;
      }
    fprintf(stderr, "[sparse grouped] %d queries (%lld in groups, %lld without a threshold), %lld segments, region cap %d: largest region %lld "
            "(query %d, segment %d), %lld regions over the cap, largest query total %lld, largest spill %lld, %lld queries overflowed\n", nq,
            static_cast<long long>(n_members), static_cast<long long>(no_theta), static_cast<long long>(n_seg), cand_cap,
            static_cast<long long>(max_region), worst_q, worst_seg, static_cast<long long>(over_regions),
            static_cast<long long>(max_total), static_cast<long long>(max_spill), static_cast<long long>(over_q));
  }
  VR_HIP(hipMemcpyAsync(pin_host<int32_t>(e, kPinSparseCands), ranked, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
  return 0;
}

// nq queries in device memory (CSR as sparse_inv_batch_kernel takes it, raw values in q_val_dev) -> nq x k keys in
// out_keys_dev (device-visible). Queries with an empty term range come out as empty lists.
int inv_scan_topk_batch(vr_engine* e, const int32_t* q_off_dev, const int32_t* q_ids_dev, const float* q_val_dev,
                        float* q_w_dev, int nq, int n_terms, bool weights_given, float n_points, const uint8_t* mask_dev,
                        int k, uint64_t* out_keys_dev, const int32_t* q_off_host, const int32_t* q_ids_host,
                        bool allow_grouped) {
  VR_CHECK(nq >= 1 && k >= 1 && k <= kListLen, "bad inverted-scan shape");
  *pin_host<int32_t>(e, kPinSparseOverflow) = 0;
  e->stat_sparse_group_cands += *pin_host<int32_t>(e, kPinSparseCands);  // (of the batch before: its stream has been waited for)
  *pin_host<int32_t>(e, kPinSparseCands) = 0;
  if (n_terms > 0)
    hipLaunchKernelGGL(sparse_batch_weights_kernel, dim3(static_cast<unsigned>((n_terms + 255) / 256)), dim3(256), 0,
                       e->stream, q_ids_dev, q_val_dev, n_terms, weights_given ? 1 : 0, e->df_keys.p, e->df_cnt.p, e->df_cap,
                       n_points, q_w_dev, q_w_dev + n_terms);
  // a block walks its share of the segments for one query; enough blocks to fill the chip several times over, few
  // enough lists per query for one merge block (and a candidate array of nq x gx x 512 B)
  // (few blocks per query: a block that walks many segments scans the first of them in full and — once its lists hold
  // k keys — only the essential terms' postings of the others, see inv_scan_segments. VR_SPARSE_BATCH_BLOCKS sets the
  // blocks per query for tests.)
  int gx = static_cast<int>(std::min<int64_t>(e->n_inv_seg, std::max<int64_t>(1, (16384 + nq - 1) / nq)));
  if (const char* v = std::getenv("VR_SPARSE_BATCH_BLOCKS")) gx = static_cast<int>(std::min<int64_t>(e->n_inv_seg, std::max(1, atoi(v))));
  gx = std::min(gx, kScanBlocks);
  const InvForward fw{e->row_slice.p, e->slices.p, e->sp_idx.p, e->sp_val.p};
  const dim3 block(kInvWaves * 64);
  static const bool seeding = !(std::getenv("VR_SPARSE_SEED") && atoi(std::getenv("VR_SPARSE_SEED")) == 0);
  const bool pruned = seeding && !weights_given;  // (given weights: the engine does not know the terms' frequencies)
  const int gs = std::min(gx, 16);  // blocks per query of the seed pass (a few rows per segment: latency, not work)
  VR_TRY(e->sp_cand.grow(static_cast<int64_t>(nq) * std::max(gx, gs) * kListLen, 0, e->stream));
  prof_begin(e, VR_PROF_SPARSE_SCAN, 0.0);
  const uint64_t* seed_keys = nullptr;
  const int32_t* need_full = nullptr;
  static const bool debug = std::getenv("VR_SPARSE_DEBUG") && atoi(std::getenv("VR_SPARSE_DEBUG")) != 0;
  unsigned long long* dbg = nullptr;
  if (pruned && debug) {
    VR_HIP(hipMalloc(reinterpret_cast<void**>(&dbg), 8 * sizeof(unsigned long long)));
    VR_HIP(hipMemsetAsync(dbg, 0, 8 * sizeof(unsigned long long), e->stream));
  }
  // many queries: groups of queries share a block per segment and every run of the group's terms is read once
  // (inv_scan_grouped; VR_SPARSE_GROUPED=0 keeps the per-query kernels. The switches are read per call: tests compare the
  // paths, and force an overflow, in one process)
  // (its candidate regions are nq x segments x >= 16 keys: batches whose regions would pass 2 GB stay on the per-query kernels)
  const bool grouped = pruned && !dbg && allow_grouped && q_off_host && q_ids_host && nq >= 16 && e->n_inv_seg <= 65535 &&
                       static_cast<int64_t>(nq) * e->n_inv_seg * std::max<int64_t>(kGrpCandCap, std::min<int64_t>(512, 8192 / std::max<int64_t>(e->n_inv_seg, 1))) <= (int64_t{1} << 28) &&
                       !(std::getenv("VR_SPARSE_GROUPED") && atoi(std::getenv("VR_SPARSE_GROUPED")) == 0);
  // ... with the thresholds from a sample of the segments scanned the same way, once there are enough of them
  const bool sampled = grouped && e->n_inv_seg >= 128 && !(std::getenv("VR_SPARSE_GROUP_SAMPLE") && atoi(std::getenv("VR_SPARSE_GROUP_SAMPLE")) == 0);
  if (pruned && !sampled) {
    // 1. seed: per query the k best rows among those that carry its rarest terms, scored exactly -> out_keys_dev; the
    //    k-th of them is a lower bound of the final k-th best score (real rows, real scores)
    hipLaunchKernelGGL((sparse_inv_pruned_kernel<true>), dim3(static_cast<unsigned>(gs), static_cast<unsigned>(nq)), block, 0,
                       e->stream, e->inv_seg.p, static_cast<int>(e->n_inv_seg), e->inv_key.p, e->inv_val.p, q_off_dev, q_ids_dev, q_w_dev,
                       q_w_dev + n_terms, mask_dev, k, e->sp_cand.p, fw, n_points, static_cast<const uint64_t*>(nullptr),
                       static_cast<int32_t*>(nullptr), dbg);
    VR_TRY(topk_merge_lists(e, e->sp_cand.p, gs, nq, k, out_keys_dev));
    seed_keys = out_keys_dev;
  }
  if (grouped) {
    bool done = false;
    VR_TRY(inv_scan_grouped(e, q_w_dev, nq, mask_dev, k, out_keys_dev, q_off_host, q_ids_host, sampled, &done));
    if (done) return 0;
    VR_CHECK(!sampled, "the grouped scan found no query with terms");  // (n_terms > 0 here: cannot happen)
  }
  if (pruned) {
    // 2. the pruned scan: only the essential terms' postings, from the first segment on; blocks that cannot prune flag
    //    their share
    VR_TRY(e->stage_i32b.grow(static_cast<int64_t>(nq) * gx, 0, e->stream));
    hipLaunchKernelGGL((sparse_inv_pruned_kernel<false>), dim3(static_cast<unsigned>(gx), static_cast<unsigned>(nq)), block,
                       static_cast<size_t>(kInvWaves) * kPrunedHash * 8,
                       e->stream, e->inv_seg.p, static_cast<int>(e->n_inv_seg), e->inv_key.p, e->inv_val.p, q_off_dev, q_ids_dev, q_w_dev,
                       q_w_dev + n_terms, mask_dev, k, e->sp_cand.p, fw, n_points, seed_keys, e->stage_i32b.p, dbg);
    need_full = e->stage_i32b.p;
    if (dbg) {
      unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      VR_HIP(hipMemcpyAsync(h, dbg, sizeof(h), hipMemcpyDeviceToHost, e->stream));
      VR_HIP(hipStreamSynchronize(e->stream));
      (void)hipFree(dbg);
      fprintf(stderr, "[sparse batch] %d queries x %d shares: pruned %llu, no seed %llu, every term essential %llu, too many rows %llu; "
              "rows scored one by one: %llu by the scan, %llu by the seed pass\n", nq, gx, h[0], h[1], h[2], h[3], h[4], h[5]);
    }
  }
  // 3. the full scan (every term's postings added up per segment) of the shares that are left — all of them without a seed
  hipLaunchKernelGGL(sparse_inv_batch_kernel, dim3(static_cast<unsigned>(gx), static_cast<unsigned>(nq)), block, 0, e->stream,
                     e->inv_seg.p, static_cast<int>(e->n_inv_seg), e->inv_key.p, e->inv_val.p, q_off_dev, q_ids_dev, q_w_dev,
                     q_w_dev + n_terms, mask_dev, k, e->sp_cand.p, fw, seed_keys, need_full);
  prof_end(e);
  VR_HIP(hipGetLastError());
  return topk_merge_lists(e, e->sp_cand.p, gx, nq, k, out_keys_dev);
}

}  // namespace vr
