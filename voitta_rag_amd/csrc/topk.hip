// Exact, deterministic top-k over a score column — the selection half of what the Qdrant server
// does for client.query_points(limit=...) (reference: src/voitta/services/vector_store.py:612-617,
// 640-656). Scores become 64-bit keys (order-preserving f32 bits << 32 | ~row) so that "larger
// key" means "higher score, then lower row id" (tie rule: SURVEY.md F8) and keys are unique.
// -inf marks excluded rows (masked, tombstoned, or no shared sparse term) and maps to key 0.
//
// Selection is k rounds of block-wide extract-max over keys held in registers: level 1 turns
// each 4096-score segment into its k best keys, later levels fold up to 16384 keys per block
// until one block is left. Integer work only; HBM traffic is N*4 bytes per query, which for
// the corpus sizes of BASELINE.json (<= 4 MB per query) is served from L2 / Infinity Cache.

#include "engine_internal.h"
#include "topk_device.h"

namespace vr {

__device__ __forceinline__ uint64_t make_key(float s, int64_t row) {
  if (s == -__builtin_inff()) return 0ull;
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return (static_cast<uint64_t>(u) << 32) | static_cast<uint32_t>(0xFFFFFFFFu - static_cast<uint32_t>(row));
}

template <int THREADS, int ITEMS>
__device__ __forceinline__ void block_extract_topk(uint64_t (&keys)[ITEMS], int k,
                                                   uint64_t* __restrict__ out) {
  constexpr int WAVES = THREADS / 64;
  __shared__ uint64_t wmax[2][WAVES];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;

  uint64_t lmax = 0;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) lmax = keys[i] > lmax ? keys[i] : lmax;

  for (int r = 0; r < k; ++r) {
    uint64_t m = lmax;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      uint64_t o = __shfl_xor(m, off);
      m = o > m ? o : m;
    }
    if (lane == 0) wmax[r & 1][wave] = m;
    __syncthreads();
    uint64_t best = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      uint64_t v = wmax[r & 1][w];
      best = v > best ? v : best;
    }
    if (threadIdx.x == 0) out[r] = best;
    if (best == 0) {  // block-uniform: nothing left
      for (int j = r + 1 + threadIdx.x; j < k; j += THREADS) out[j] = 0;
      break;
    }
    if (lmax == best) {  // keys are unique: exactly one owner
      lmax = 0;
#pragma unroll
      for (int i = 0; i < ITEMS; ++i) {
        if (keys[i] == best) keys[i] = 0;
        lmax = keys[i] > lmax ? keys[i] : lmax;
      }
    }
  }
}

// level 1: scores [nq][stride] -> keys [nq][nseg][k]
template <int THREADS, int ITEMS>
__global__ __launch_bounds__(THREADS) void select_from_scores_kernel(
    const float* __restrict__ scores, int64_t stride, int64_t n, int k, uint64_t* __restrict__ out) {
  const int64_t seg = blockIdx.x;
  const int q = blockIdx.y;
  const int64_t base = seg * (THREADS * ITEMS);
  const float* s = scores + q * stride;
  uint64_t keys[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    int64_t row = base + i * THREADS + threadIdx.x;
    keys[i] = row < n ? make_key(s[row], row) : 0ull;
  }
  block_extract_topk<THREADS, ITEMS>(keys, k, out + (static_cast<int64_t>(q) * gridDim.x + seg) * k);
}

// level >= 2: keys [nq][count] -> keys [nq][nseg][k]
template <int THREADS, int ITEMS>
__global__ __launch_bounds__(THREADS) void select_from_keys_kernel(const uint64_t* __restrict__ in,
                                                                   int64_t count, int k,
                                                                   uint64_t* __restrict__ out) {
  const int64_t seg = blockIdx.x;
  const int q = blockIdx.y;
  const int64_t base = seg * (THREADS * ITEMS);
  const uint64_t* s = in + q * count;
  uint64_t keys[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    int64_t j = base + i * THREADS + threadIdx.x;
    keys[i] = j < count ? s[j] : 0ull;
  }
  block_extract_topk<THREADS, ITEMS>(keys, k, out + (static_cast<int64_t>(q) * gridDim.x + seg) * k);
}

// keys [nq][n_reg][cap]: region r of query q holds min(cnt[q][r], cap) real keys (unsorted, unique); what a full region
// could not take went to the query's spill area spill[q][spill_cap] (spill_cnt[q] keys were offered to it; may be null)
// -> keys [nq][k]. A query whose spill area overflowed, or with more than THREADS * ITEMS keys in all, sets *overflow and
// overflow_q[q].
template <int THREADS, int ITEMS>
__global__ __launch_bounds__(THREADS) void select_regions_kernel(const uint64_t* __restrict__ in, int n_reg, int cap,
                                                                 const int32_t* __restrict__ cnt,
                                                                 const uint64_t* __restrict__ spill, int spill_cap,
                                                                 const int32_t* __restrict__ spill_cnt, int k,
                                                                 uint64_t* __restrict__ out,
                                                                 int32_t* __restrict__ overflow,
                                                                 int32_t* __restrict__ total_out,
                                                                 int32_t* __restrict__ overflow_q) {
  __shared__ uint64_t gat[THREADS * ITEMS];
  __shared__ int n_gat;
  const int q = blockIdx.x;
  const int n_spill = spill_cnt ? spill_cnt[q] : 0;
  if (threadIdx.x == 0) n_gat = n_spill < spill_cap ? n_spill : spill_cap;
  __syncthreads();
  for (int j = threadIdx.x; j < n_spill && j < spill_cap; j += THREADS)
    if (j < THREADS * ITEMS) gat[j] = spill[static_cast<int64_t>(q) * spill_cap + j];
  const uint64_t* base = in + static_cast<int64_t>(q) * n_reg * cap;
  const int32_t* c = cnt + static_cast<int64_t>(q) * n_reg;
  for (int r = threadIdx.x; r < n_reg; r += THREADS) {
    const int have = c[r];
    if (have <= 0) continue;
    const int n = have < cap ? have : cap;
    const int at = atomicAdd(&n_gat, n);
    for (int j = 0; j < n; ++j)
      if (at + j < THREADS * ITEMS) gat[at + j] = base[static_cast<int64_t>(r) * cap + j];
  }
  __syncthreads();
  const int total = n_gat;
  if (threadIdx.x == 0) {
    if (total_out) atomicAdd(total_out, total);
    if (n_spill > spill_cap || total > THREADS * ITEMS) {
      *overflow = 1;
      if (overflow_q) overflow_q[q] = 1;
    }
  }
  uint64_t keys[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int j = i * THREADS + static_cast<int>(threadIdx.x);
    keys[i] = j < total && j < THREADS * ITEMS ? gat[j] : 0ull;
  }
  block_extract_topk<THREADS, ITEMS>(keys, k, out + static_cast<int64_t>(q) * k);
}

int topk_select_regions(vr_engine* e, const uint64_t* cand, int n_reg, int cap, const int32_t* cnt, const uint64_t* spill,
                        int spill_cap, const int32_t* spill_cnt, int nq, int k, uint64_t* out, int32_t* overflow_mapped,
                        int32_t* total_dev, int32_t* overflow_q) {
  VR_CHECK(k >= 1 && k <= kMaxK && nq >= 1 && cap >= 1 && n_reg >= 1, "bad region-select shape");
  hipLaunchKernelGGL((select_regions_kernel<1024, 8>), dim3(static_cast<unsigned>(nq)), dim3(1024), 0, e->stream, cand, n_reg, cap,
                     cnt, spill, spill_cap, spill_cnt, k, out, overflow_mapped, total_dev, overflow_q);
  VR_HIP(hipGetLastError());
  return 0;
}

// ---- merging sorted candidate lists ----------------------------------------------------------

constexpr int kMergeThreads = 1024;
constexpr int kMergeWaves = kMergeThreads / 64;
constexpr int kMergeMaxLists = 512;                                   // = kScanBlocks
constexpr int kGatherCap = 2048;  // candidates the fast path ranks in LDS; beyond: tournament
constexpr int kGatherBatch = 16;  // lists a wave loads at once in the gather pass

// One block per query. The lists are overwritten when the slow path runs.
__global__ __launch_bounds__(kMergeThreads) void merge_lists_kernel(uint64_t* __restrict__ cand, int n_lists,
                                                                     int k, uint64_t* __restrict__ out) {
  // 20 KiB of LDS only: this kernel must find a CU while a persistent scan of the other search leg
  // holds most of every CU's LDS (and a kernel with scratch pays ~50 us of dispatch-time setup, so
  // nothing here may spill either).
  __shared__ uint64_t gat[kGatherCap];
  __shared__ uint64_t wave_thr[kMergeWaves];
  __shared__ uint64_t heads[kMergeMaxLists];
  __shared__ uint64_t head_thr;
  __shared__ int gathered;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  uint64_t* in = cand + static_cast<int64_t>(blockIdx.x) * n_lists * kListLen;
  uint64_t* dst = out + static_cast<int64_t>(blockIdx.x) * k;

  // Fast path: two lower bounds T of the global k-th key; only keys >= T can be in the answer —
  // usually k plus a handful. Gather them, rank them.
  //   (a) the k-th key of ANY list (k keys of that list are at or above it) — good for few long lists;
  //   (b) the k-th largest list HEAD (k lists start at or above it) — with hundreds of short lists
  //       this one is close to the true k-th key.
  // Keys are unique (they carry the row) apart from the empty key 0.
  {
    uint64_t t = 0;  // one thread per list (n_lists <= kMergeMaxLists <= kMergeThreads)
    if (static_cast<int>(threadIdx.x) < n_lists) {
      heads[threadIdx.x] = in[threadIdx.x * kListLen];
      t = in[threadIdx.x * kListLen + (k - 1)];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const uint64_t o = __shfl_xor(t, off);
      t = o > t ? o : t;
    }
    if (lane == 0) wave_thr[wave] = t;
    if (threadIdx.x == 0) {
      gathered = 0;
      head_thr = 0;
    }
    __syncthreads();
    if (static_cast<int>(threadIdx.x) < n_lists) {
      const uint64_t h = heads[threadIdx.x];
      int rank = 0;
      for (int j = 0; j < n_lists; ++j) rank += heads[j] > h;
      if (rank == k - 1 && h != 0) head_thr = h;
    }
    __syncthreads();
    uint64_t thr = head_thr;
#pragma unroll
    for (int w = 0; w < kMergeWaves; ++w) thr = wave_thr[w] > thr ? wave_thr[w] : thr;
    // a wave takes lists wave, wave + 16, ...; kGatherBatch independent loads are issued together
    // (a load per iteration behind the LDS atomic serialised one global latency per list)
    for (int l0 = wave; l0 < n_lists; l0 += kMergeWaves * kGatherBatch) {
      uint64_t keys[kGatherBatch];
#pragma unroll
      for (int u = 0; u < kGatherBatch; ++u) {
        const int l = min(l0 + u * kMergeWaves, n_lists - 1);  // clamped: the load stays unconditional
        keys[u] = in[l * kListLen + lane];
      }
#pragma unroll
      for (int u = 0; u < kGatherBatch; ++u) {
        const uint64_t key = keys[u];
        const bool take = l0 + u * kMergeWaves < n_lists && lane < k && key != 0 && key >= thr;
        const uint64_t m = __ballot(take);
        if (m) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&gathered, __popcll(m));
          base = __builtin_amdgcn_readfirstlane(base);
          const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
          if (take && slot < kGatherCap) gat[slot] = key;
        }
      }
    }
    __syncthreads();
    const int c = gathered;
    if (c <= kGatherCap) {  // block-uniform
      for (int i = threadIdx.x; i < c; i += kMergeThreads) {
        const uint64_t key = gat[i];
        int rank = 0;
        for (int j = 0; j < c; ++j) rank += gat[j] > key;
        if (rank < k) dst[rank] = key;
      }
      for (int i = c + threadIdx.x; i < k; i += kMergeThreads) dst[i] = 0;
      return;
    }
  }
  // Slow path (rare): a tournament over the lists, in place in global memory, halving their number
  // per level. Rounds of kMergeWaves pairs: round j reads slots [32j, 32j+32) and writes slots
  // [16j, 16j+16), which only rounds <= j of this level have read — so one barrier between a round's
  // reads and its writes is enough, and a wave holds one merged list at a time.
  int n = n_lists;
  while (n > 1) {
    const int pairs = (n + 1) / 2;
    for (int p0 = 0; p0 < pairs; p0 += kMergeWaves) {
      const int p = p0 + wave;
      uint64_t res = 0;
      if (p < pairs) {
        const uint64_t a = in[(2 * p) * kListLen + lane];
        const uint64_t b = (2 * p + 1 < n) ? in[(2 * p + 1) * kListLen + (63 - lane)] : 0ull;
        res = merge64(a, b, lane);
      }
      __syncthreads();
      if (p < pairs) in[p * kListLen + lane] = res;
    }
    __syncthreads();  // (workgroup-scope fence included) the level's lists are visible to every wave
    n = pairs;
  }
  if (wave == 0 && lane < k) dst[lane] = in[lane];
}

int topk_merge_lists(vr_engine* e, uint64_t* cand, int n_lists, int nq, int k, uint64_t* out) {
  VR_CHECK(k >= 1 && k <= kListLen && n_lists >= 1 && n_lists <= kMergeMaxLists, "bad merge shape");
  hipLaunchKernelGGL(merge_lists_kernel, dim3(static_cast<unsigned>(nq)), dim3(kMergeThreads), 0, e->stream,
                     cand, n_lists, k, out);
  VR_HIP(hipGetLastError());
  return 0;
}

// ---- merging the result lists of several shards (vr_merge_keys; sharded.py after its all_gather) ---------------
//
// parts[p][l][0..k): shard p's keys of list l (descending, 0 = empty), rows local to the shard. One block per list:
// every key is ranked against all others — larger key first, equal keys (same score bits, same local row) by the
// lower shard — which is the single-engine order of the global ids row * n_parts + p whenever rows were dealt
// round-robin. <= 4096 keys per list; integer compares out of LDS.
constexpr int kPartsMax = 4096;

__global__ __launch_bounds__(256) void merge_parts_kernel(const uint64_t* __restrict__ parts, int n_parts, int n_lists,
                                                          int k, int64_t* __restrict__ gid, float* __restrict__ score,
                                                          int32_t* __restrict__ cnt) {
  __shared__ uint64_t keys[kPartsMax];
  __shared__ int total;
  const int l = blockIdx.x;
  const int n = n_parts * k;
  if (threadIdx.x == 0) total = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int p = i / k, j = i - p * k;
    keys[i] = parts[(static_cast<int64_t>(p) * n_lists + l) * k + j];
  }
  for (int j = threadIdx.x; j < k; j += 256) {
    gid[static_cast<int64_t>(l) * k + j] = -1;
    score[static_cast<int64_t>(l) * k + j] = 0.0f;
  }
  __syncthreads();
  int mine = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const uint64_t key = keys[i];
    if (key == 0) continue;
    ++mine;
    int rank = 0;
    for (int o = 0; o < n; ++o) {
      const uint64_t ok = keys[o];
      rank += (ok > key) || (ok == key && o < i);  // o < i among equal keys == lower shard (a shard's keys are distinct)
    }
    if (rank < k) {
      const uint32_t hi = static_cast<uint32_t>(key >> 32);
      const uint32_t u = (hi & 0x80000000u) ? (hi ^ 0x80000000u) : ~hi;
      const int64_t row = static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(key & 0xFFFFFFFFu));
      gid[static_cast<int64_t>(l) * k + rank] = row * n_parts + i / k;
      score[static_cast<int64_t>(l) * k + rank] = __uint_as_float(u);
    }
  }
  if (mine) atomicAdd(&total, mine);
  __syncthreads();
  if (threadIdx.x == 0) cnt[l] = total < k ? total : k;
}

int topk_merge_parts(vr_engine* e, const uint64_t* parts_dev, int n_parts, int n_lists, int k, int64_t* gid_dev,
                     float* score_dev, int32_t* cnt_dev) {
  VR_CHECK(n_parts >= 1 && n_lists >= 1 && k >= 1 && n_parts * k <= kPartsMax, "merge of %d parts x %d keys (at most %d keys per list)",
           n_parts, k, kPartsMax);
  hipLaunchKernelGGL(merge_parts_kernel, dim3(static_cast<unsigned>(n_lists)), dim3(256), 0, e->stream, parts_dev, n_parts,
                     n_lists, k, gid_dev, score_dev, cnt_dev);
  VR_HIP(hipGetLastError());
  return 0;
}

int topk_select(vr_engine* e, const float* scores, int64_t stride, int64_t n, int nq, int k,
                const uint64_t** out_keys) {
  VR_CHECK(k >= 1 && k <= kMaxK, "top-k of %d not in 1..%d", k, kMaxK);
  VR_CHECK(nq >= 1, "no queries");
  constexpr int T1 = 256, I1 = kTopkSeg / T1;
  constexpr int T2 = 1024, I2 = 16;
  int64_t nseg = (n + kTopkSeg - 1) / kTopkSeg;
  if (nseg < 1) nseg = 1;
  int64_t need = static_cast<int64_t>(nq) * nseg * k;
  VR_TRY(e->cand_a.grow(need, 0, e->stream));
  VR_TRY(e->cand_b.grow(need, 0, e->stream));
  hipLaunchKernelGGL((select_from_scores_kernel<T1, I1>), dim3(static_cast<unsigned>(nseg), nq),
                     dim3(T1), 0, e->stream, scores, stride, n, k, e->cand_a.p);
  uint64_t* cur = e->cand_a.p;
  uint64_t* nxt = e->cand_b.p;
  int64_t count = nseg * k;
  while (nseg > 1) {
    int64_t nseg2 = (count + T2 * I2 - 1) / (T2 * I2);
    hipLaunchKernelGGL((select_from_keys_kernel<T2, I2>), dim3(static_cast<unsigned>(nseg2), nq),
                       dim3(T2), 0, e->stream, cur, count, k, nxt);
    uint64_t* t = cur;
    cur = nxt;
    nxt = t;
    nseg = nseg2;
    count = nseg * k;
  }
  VR_HIP(hipGetLastError());
  *out_keys = cur;
  return 0;
}

}  // namespace vr
