// Device helpers shared by the scan kernels: 64-bit ranking keys and a per-wave top-K list kept
// in LDS. Exact and deterministic: a key is (order-preserving f32 bits << 32) | ~row, so "larger
// key" == "higher score, then lower row id"; keys are unique; key 0 == "no candidate".
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace vr {

constexpr int kListLen = 64;  // entries per candidate list (one per lane); fused path serves k <= 64

__device__ __forceinline__ uint64_t topk_make_key(float s, int64_t row) {
  if (s == -__builtin_inff()) return 0ull;
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return (static_cast<uint64_t>(u) << 32) |
         static_cast<uint32_t>(0xFFFFFFFFu - static_cast<uint32_t>(row));
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int src_lane) {
  uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(v), src_lane);
  uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(v >> 32), src_lane);
  return (static_cast<uint64_t>(hi) << 32) | lo;
}

// list: kListLen keys in LDS, sorted descending, lane i owns slot i. `key` is wave-uniform.
// Keeps the k best: a key that would land at position >= k is dropped.
__device__ __forceinline__ void wave_list_insert(uint64_t* list, int k, uint64_t key, int lane) {
  const uint64_t cur = list[lane];
  const uint64_t greater = __ballot(cur > key);
  const int pos = __popcll(greater);
  if (pos >= k) return;  // wave-uniform
  const uint64_t prev = __shfl_up(cur, 1);  // all lanes take part
  if (lane == pos) list[lane] = key;
  else if (lane > pos && lane < k) list[lane] = prev;
}

// Offer the (non-uniform) per-lane key to the list of query `q_of_lane`, for every lane whose key
// beats that list's current k-th entry. lists: [nq][kListLen] of this wave.
__device__ __forceinline__ void wave_offer(uint64_t* lists, int k, uint64_t key, int q_of_lane,
                                           bool active, int lane) {
  uint64_t thr = active ? lists[q_of_lane * kListLen + (k - 1)] : ~0ull;
  uint64_t pending = __ballot(active && key > thr);
  while (pending) {
    const int src = __builtin_ctzll(pending);
    const uint64_t kk = readlane_u64(key, src);
    const int qs = __builtin_amdgcn_readlane(q_of_lane, src);
    wave_list_insert(lists + qs * kListLen, k, kk, lane);
    pending &= pending - 1;
  }
}

// Two descending 64-entry lists -> their 64 best, descending, one entry per lane.
// C[i] = max(A[i], B[63-i]) is a bitonic sequence holding the 64 largest of the union; six
// half-cleaner steps sort it.
__device__ __forceinline__ uint64_t merge64(uint64_t a, uint64_t b_reversed, int lane) {
  uint64_t v = a > b_reversed ? a : b_reversed;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint64_t p = __shfl_xor(v, d);
    const bool keep_max = (lane & d) == 0;
    v = keep_max ? (v > p ? v : p) : (v < p ? v : p);
  }
  return v;
}

// Fold the per-wave lists of a block (lists[w * stride .. + kListLen), w < n_waves, n_waves a
// power of two) into wave 0's list: log2(n_waves) rounds of pairwise merge64. Call from every
// thread of the block.
__device__ __forceinline__ void block_merge_lists(uint64_t* lists, int stride, int n_waves, int wave, int lane) {
  for (int s = 1; s < n_waves; s <<= 1) {
    __syncthreads();
    uint64_t merged = 0;
    const bool mine = (wave % (2 * s)) == 0;
    if (mine) merged = merge64(lists[wave * stride + lane], lists[(wave + s) * stride + (63 - lane)], lane);
    __syncthreads();
    if (mine) lists[wave * stride + lane] = merged;
  }
  __syncthreads();
}

}  // namespace vr
