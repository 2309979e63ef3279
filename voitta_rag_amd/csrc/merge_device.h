// The merge of sorted candidate lists (topk.hip's merge_lists_kernel) as a device function, so that the scan and re-score
// kernels of the single-query search can let their LAST workgroup do it instead of a launch of its own.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "topk_device.h"

namespace vr {

constexpr int kMergeThreads = 1024;
constexpr int kMergeWaves = kMergeThreads / 64;
constexpr int kMergeMaxLists = 512;                                   // = kScanBlocks
constexpr int kGatherCap = 2048;  // candidates the fast path ranks in LDS; beyond: tournament
constexpr int kGatherBatch = 16;  // lists a wave loads at once in the gather pass

// Key loads / stores of the merge. SC1: the lists were written by OTHER workgroups of the SAME launch (the scan and
// re-score kernels let their last workgroup merge, see finish_lists below): such data must be stored and loaded at
// system scope — past the L2 of the storing and of the loading XCD, which do not snoop each other
// (MI355X_MICROARCH.md, "who signals ... stores, all sc1 | loads, all sc1").
template <bool SC1>
__device__ __forceinline__ uint64_t merge_ld(const uint64_t* p) {
  if (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return *p;
}
template <bool SC1>
__device__ __forceinline__ void merge_st(uint64_t* p, uint64_t v) {
  if (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else *p = v;
}

// The k best of n_lists sorted 64-entry lists (`in`, overwritten when the slow path runs) -> dst[0 .. k), by one block of
// THREADS threads (at least one per list).
template <bool SC1, int THREADS>
__device__ __forceinline__ void merge_lists_block(uint64_t* __restrict__ in, int n_lists, int k, uint64_t* __restrict__ dst) {
  constexpr int WAVES = THREADS / 64;
  static_assert(THREADS >= kMergeMaxLists, "one thread per list");
  // 20 KiB of LDS only: this kernel must find a CU while a persistent scan of the other search leg
  // holds most of every CU's LDS (and a kernel with scratch pays ~50 us of dispatch-time setup, so
  // nothing here may spill either).
  __shared__ uint64_t gat[kGatherCap];
  __shared__ uint64_t wave_thr[WAVES];
  __shared__ uint64_t heads[kMergeMaxLists];
  __shared__ uint64_t head_thr;
  __shared__ int gathered;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;

  // Fast path: two lower bounds T of the global k-th key; only keys >= T can be in the answer —
  // usually k plus a handful. Gather them, rank them.
  //   (a) the k-th key of ANY list (k keys of that list are at or above it) — good for few long lists;
  //   (b) the k-th largest list HEAD (k lists start at or above it) — with hundreds of short lists
  //       this one is close to the true k-th key.
  // Keys are unique (they carry the row) apart from the empty key 0.
  {
    uint64_t t = 0;  // one thread per list (n_lists <= kMergeMaxLists <= THREADS)
    if (static_cast<int>(threadIdx.x) < n_lists) {
      heads[threadIdx.x] = merge_ld<SC1>(in + threadIdx.x * kListLen);
      t = merge_ld<SC1>(in + threadIdx.x * kListLen + (k - 1));
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
    for (int w = 0; w < WAVES; ++w) thr = wave_thr[w] > thr ? wave_thr[w] : thr;
    // a wave takes lists wave, wave + 16, ...; kGatherBatch independent loads are issued together
    // (a load per iteration behind the LDS atomic serialised one global latency per list)
    for (int l0 = wave; l0 < n_lists; l0 += WAVES * kGatherBatch) {
      uint64_t keys[kGatherBatch];
#pragma unroll
      for (int u = 0; u < kGatherBatch; ++u) {
        const int l = min(l0 + u * WAVES, n_lists - 1);  // clamped: the load stays unconditional
        keys[u] = merge_ld<SC1>(in + l * kListLen + lane);
      }
#pragma unroll
      for (int u = 0; u < kGatherBatch; ++u) {
        const uint64_t key = keys[u];
        const bool take = l0 + u * WAVES < n_lists && lane < k && key != 0 && key >= thr;
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
      for (int i = threadIdx.x; i < c; i += THREADS) {
        const uint64_t key = gat[i];
        int rank = 0;
        for (int j = 0; j < c; ++j) rank += gat[j] > key;
        if (rank < k) dst[rank] = key;
      }
      for (int i = c + threadIdx.x; i < k; i += THREADS) dst[i] = 0;
      return;
    }
  }
  // Slow path (rare): a tournament over the lists, in place in global memory, halving their number
  // per level. Rounds of WAVES pairs: round j reads slots [32j, 32j+32) and writes slots
  // [16j, 16j+16), which only rounds <= j of this level have read — so one barrier between a round's
  // reads and its writes is enough, and a wave holds one merged list at a time.
  int n = n_lists;
  while (n > 1) {
    const int pairs = (n + 1) / 2;
    for (int p0 = 0; p0 < pairs; p0 += WAVES) {
      const int p = p0 + wave;
      uint64_t res = 0;
      if (p < pairs) {
        const uint64_t a = merge_ld<SC1>(in + (2 * p) * kListLen + lane);
        const uint64_t b = (2 * p + 1 < n) ? merge_ld<SC1>(in + (2 * p + 1) * kListLen + (63 - lane)) : 0ull;
        res = merge64(a, b, lane);
      }
      __syncthreads();
      if (p < pairs) merge_st<SC1>(in + p * kListLen + lane, res);
    }
    __syncthreads();  // (workgroup-scope fence included) the level's lists are visible to every wave
    n = pairs;
  }
  if (wave == 0 && lane < k) dst[lane] = merge_ld<SC1>(in + lane);
}


// "Last workgroup finishes": every workgroup of a launch leaves its 64-entry list in lists[blockIdx.x] (system-scope
// stores), waits for its own stores, then ONE lane counts it in; the workgroup whose count came last merges all of
// them into dst[0 .. k) and zeroes the counter for the next launch. Returns true in that workgroup only (all its
// threads). `mine`: the block's list in LDS; a block of THREADS threads, gridDim.x <= kMergeMaxLists.
template <int THREADS>
__device__ __forceinline__ bool finish_lists(const uint64_t* mine, uint64_t* __restrict__ lists, int32_t* __restrict__ arrivals,
                                             int k, uint64_t* __restrict__ dst) {
  __shared__ int last_s;
  if (threadIdx.x < kListLen)
    __hip_atomic_store(lists + static_cast<int64_t>(blockIdx.x) * kListLen + threadIdx.x, mine[threadIdx.x], __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0)
    last_s = __hip_atomic_fetch_add(arrivals, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == static_cast<int>(gridDim.x) - 1;
  __syncthreads();
  if (!last_s) return false;  // block-uniform
  merge_lists_block<true, THREADS>(lists, static_cast<int>(gridDim.x), k, dst);
  if (threadIdx.x == 0) __hip_atomic_store(arrivals, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return true;
}

}  // namespace vr
