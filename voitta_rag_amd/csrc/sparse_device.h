// Device helpers shared by the forward (sparse.hip) and inverted (invert.hip) sparse scans.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace vr {

__device__ __forceinline__ uint32_t df_hash(int32_t id) {
  uint32_t h = static_cast<uint32_t>(id) * 0x9E3779B1u;
  return h ^ (h >> 15);
}

__device__ __forceinline__ int32_t df_get(const int32_t* keys, const int32_t* cnt, int64_t cap,
                                          int32_t id) {
  uint64_t h = df_hash(id) & (cap - 1);
  for (int64_t probe = 0; probe < cap; ++probe) {
    int32_t cur = keys[h];
    if (cur == id) return cnt[h];
    if (cur == -1) return 0;
    h = (h + 1) & (cap - 1);
  }
  return 0;
}

// Share of the sparse points that carry term `id` (df_t / N; 1 when the statistics are not the engine's own): what the
// inverted scan's pruning uses to tell a long posting list from a short one before it has looked at either.
__device__ __forceinline__ float sparse_term_fraction(int32_t id, int weights_given, const int32_t* __restrict__ df_keys,
                                                      const int32_t* __restrict__ df_cnt, int64_t df_cap, float n_points) {
  if (weights_given || df_cap == 0 || !(n_points > 0.0f)) return 1.0f;
  const float f = static_cast<float>(df_get(df_keys, df_cnt, df_cap, id)) / n_points;
  return f < 1.0f ? f : 1.0f;
}

// Weight of one query term: q_t as given, or q_t * idf_t with idf_t = ln(1 + (N - df_t + 0.5)/(df_t + 0.5)) —
// the argument formed in f32, ln taken in f64 and rounded once (SURVEY.md a13 [EXT]). Both scans call this, so
// their weights are the same bits.
__device__ __forceinline__ float sparse_query_weight(float q, int32_t id, int weights_given,
                                                     const int32_t* __restrict__ df_keys,
                                                     const int32_t* __restrict__ df_cnt, int64_t df_cap,
                                                     float n_points) {
  if (weights_given) return q;
  const float df = static_cast<float>(df_cap ? df_get(df_keys, df_cnt, df_cap, id) : 0);
  const float num = __fadd_rn(__fadd_rn(n_points, -df), 0.5f);
  const float den = __fadd_rn(df, 0.5f);
  const float arg = __fadd_rn(1.0f, __fdiv_rn(num, den));
  // ln in f64, rounded once to f32. The empty asm hides that `a` is a widened float: otherwise
  // LLVM shrinks (float)log((double)x) to logf(x), whose last bit differs from the host's.
  double a = static_cast<double>(arg);
  asm volatile("" : "+v"(a));
  return __fmul_rn(q, static_cast<float>(log(a)));
}

}  // namespace vr
