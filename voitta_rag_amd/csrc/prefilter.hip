// Two-stage exact dense search. The answer is the one dense.hip's f32 scan gives — bit for bit —
// but the N*D*4-byte stream is replaced by an N*D*2-byte one plus a few dozen exact re-scores.
// Stands in for client.query_points(query=vec, limit=k) (reference:
// src/voitta/services/vector_store.py:612-617,640-645) on the single-query latency path.
//
// Stage 0 (insert): every stored row x (already cosine-preprocessed, f32) also gets an f16 shadow
//   h = f16(x), MFMA-tiled for v_mfma_f32_16x16x32_f16 ([row/16][k/32][lane = g*16 + r][16 B] with
//   k = 32*kb + 8*g + j), plus err_r = |x - h|_2 computed exactly at insert time.
// Stage 1 (scan, HBM bound, N*D*2 bytes): the query q is carried as (hi, lo) f16 (22 bits); the f16
//   MFMA accumulates a = sum h_k (qh_k + ql_k) in f32. By Cauchy-Schwarz
//       |x.q - h.q| <= err_r |q|_2
//   and with the query's representation error and both f32 accumulation errors bounded by constants
//   (see prefilter_scan_kernel) every row gets certain bounds  lo <= f <= up  on the score f that
//   the one-stage f32 fma chain returns. Per-wave top-k lists are kept on `lo`.
// Stage 2: T = k-th largest `lo`. A row with up < T cannot be among the k best f (k rows are
//   certainly >= T > it), so candidates = { up >= T } — k plus a few (the bounds are ~3e-4 wide).
// Stage 3: every candidate's tile is re-scored with exactly the instruction sequence of the
//   one-stage scan (v_mfma_f32_16x16x4_f32 chain over the f32 corpus), and the k best exact keys
//   are ranked. More than kMaxCandidates candidates (e.g. a corpus of exact duplicates) makes the
//   host fall back to the one-stage scan, so the result never depends on the bound being tight.
// (An int8 first stage was tried first: a quarter of the bytes, but its worst-case bound grows
//  linearly with D and left ~4000 candidates per query at 1M x 768.)

#include "engine_internal.h"
#include "topk_device.h"

#include <algorithm>

namespace vr {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using half_t = _Float16;

constexpr int kK16 = 32;  // k per f16 tile block (1 KiB = 16 rows x 32 halfs)

// ---- stage 0: f16 shadow of stored rows ------------------------------------------------------------

// one wave per row; reads the row back from the f32 tiled corpus (so it sees exactly what is stored)
__global__ __launch_bounds__(256) void shadow_rows_kernel(const float* __restrict__ corpus, int64_t first_row,
                                                          int64_t n, int dim, int kblocks,
                                                          half_t* __restrict__ corpus16,
                                                          float* __restrict__ row_err) {
  const int lane = threadIdx.x & 63;
  const int64_t row = first_row + static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= first_row + n) return;
  const int64_t tile = row / kTileRows;
  const int r = static_cast<int>(row % kTileRows);
  const int kb16n = dim / kK16;
  float e2 = 0.0f;
  for (int k = lane; k < dim; k += 64) {
    const int kb = k / kTileK, kk = k % kTileK;
    const float x = corpus[((tile * kblocks + kb) * 64 + (kk % 4) * 16 + r) * 4 + kk / 4];
    const float xc = fminf(fmaxf(x, -65504.0f), 65504.0f);
    const half_t h = static_cast<half_t>(xc);
    const float d = x - static_cast<float>(h);
    e2 += d * d;
    const int kb16 = k / kK16, g = (k % kK16) / 8, j = k % 8;
    corpus16[((tile * kb16n + kb16) * 64 + g * 16 + r) * 8 + j] = h;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) e2 += __shfl_xor(e2, off);
  // rounded up: 0.1 % on the norm plus an absolute floor cover the f32 summation of e2
  if (lane == 0) row_err[row] = sqrtf(e2) * 1.001f + 1.0e-12f;
}

int prefilter_store_rows(vr_engine* e, int64_t n, int64_t first_row) {
  if (!e->prefilter || n <= 0) return 0;
  hipLaunchKernelGGL(shadow_rows_kernel, dim3(static_cast<unsigned>((n + 3) / 4)), dim3(256), 0, e->stream,
                     e->corpus.p, first_row, n, e->dim, e->kblocks, reinterpret_cast<half_t*>(e->corpus16.p),
                     e->row_err.p);
  VR_HIP(hipGetLastError());
  return 0;
}

// ---- stage 1: query split + f16 scan -----------------------------------------------------------------

constexpr int kScan16Waves = 8;

// grid-stride over 16-row tiles; per tile D/32 loads of 1 KiB and 2 f16 MFMAs each (query hi, lo).
// Writes the upper bound of every row and keeps the k best LOWER bounds per wave.
// q_img is the f32 tiled query image (row 0 = the query, already preprocessed); every block splits
// it into the (hi, lo) f16 B-operand images [2][kb16n][lane = g*16 + col][8 halfs] (column 0 only)
// in its own LDS and computes |q| — cheaper than a separate launch.
__global__ __launch_bounds__(kScan16Waves * 64) void prefilter_scan_kernel(
    const uint4* __restrict__ corpus16, const float* __restrict__ q_img, const float* __restrict__ row_err,
    const uint8_t* __restrict__ mask, int64_t n_tiles, int kb16n, int dim, int k, float* __restrict__ upper,
    uint64_t* __restrict__ cand, int32_t* __restrict__ counter) {
  extern __shared__ uint4 q_lds[];                                        // [2][kb16n][64] = 2*kb16n KiB
  __shared__ float red[kScan16Waves];
  uint64_t* lists = reinterpret_cast<uint64_t*>(q_lds + 2 * kb16n * 64);  // [waves][kListLen]
  if (blockIdx.x == 0 && threadIdx.x == 0) *counter = 0;  // candidate counter of the collect pass
  for (int i = threadIdx.x; i < 2 * kb16n * 64; i += kScan16Waves * 64) q_lds[i] = make_uint4(0, 0, 0, 0);
  for (int i = threadIdx.x; i < kScan16Waves * kListLen; i += kScan16Waves * 64) lists[i] = 0ull;
  __syncthreads();
  float n2 = 0.0f;
  {
    half_t* qh = reinterpret_cast<half_t*>(q_lds);
    half_t* ql = qh + kb16n * 512;
    for (int kk = threadIdx.x; kk < dim; kk += kScan16Waves * 64) {
      const int kb = kk / kTileK, kr = kk % kTileK;
      const float q = q_img[(kb * 64 + (kr % 4) * 16 + 0) * 4 + kr / 4];
      n2 += q * q;
      const float qc = fminf(fmaxf(q, -65504.0f), 65504.0f);
      const half_t hi = static_cast<half_t>(qc);
      const half_t lo = static_cast<half_t>(qc - static_cast<float>(hi));
      const int at = ((kk / kK16) * 64 + ((kk % kK16) / 8) * 16 + 0) * 8 + kk % 8;
      qh[at] = hi;
      ql[at] = lo;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) n2 += __shfl_xor(n2, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n2;
  __syncthreads();
  float qn2 = 0.0f;
#pragma unroll
  for (int w = 0; w < kScan16Waves; ++w) qn2 += red[w];
  const float qn = sqrtf(qn2) * 1.0001f + 1.0e-12f;  // |q|_2, rounded up
  // certain bound on |f - a| apart from the row's own err_r * |q|:
  //   query representation: |h|_2 |q - qh - ql|_2 <= 1.001 (2^-21 |q| + 2^-24 sqrt(D))  (f16 lo: 11 more
  //     bits where normal, absolute spacing 2^-24 where subnormal)
  //   accumulation: the exact f32 chain and this f16 MFMA chain each err by at most ~2D roundings of
  //     2^-23 relative to sum |terms| <= |x||q| <= 1.001 |q|   (2^-23: safe even if the MFMA truncates)
  const float c_fixed = 1.001f * (4.8e-7f * qn + 6.0e-8f * sqrtf(static_cast<float>(dim))) +
                        static_cast<float>(dim) * 5.0e-7f * fmaxf(qn, 1.0f);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  uint64_t* mine = lists + wave * kListLen;
  const bool active = (lane & 15) == 0;  // C/D map: column (query) = lane & 15; only query 0 exists
  const int64_t wave_stride = static_cast<int64_t>(gridDim.x) * kScan16Waves;
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * kScan16Waves + wave; tile < n_tiles; tile += wave_stride) {
    const uint4* src = corpus16 + tile * kb16n * 64 + lane;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    int kb = 0;
    for (; kb + 8 <= kb16n; kb += 8) {
      uint4 a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = src[(kb + u) * 64];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint4 bh = q_lds[(kb + u) * 64 + lane];
        const uint4 bl = q_lds[(kb16n + kb + u) * 64 + lane];
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&a[u]),
                                                     *reinterpret_cast<const f16x8*>(&bh), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&a[u]),
                                                     *reinterpret_cast<const f16x8*>(&bl), acc, 0, 0, 0);
      }
    }
    for (; kb < kb16n; ++kb) {
      const uint4 a = src[kb * 64];
      const uint4 bh = q_lds[kb * 64 + lane];
      const uint4 bl = q_lds[(kb16n + kb) * 64 + lane];
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&a),
                                                   *reinterpret_cast<const f16x8*>(&bh), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&a),
                                                   *reinterpret_cast<const f16x8*>(&bl), acc, 0, 0, 0);
    }
    const int64_t row0 = tile * kTileRows + (lane >> 4) * 4;  // rows = 4*(lane >> 4) + reg
    uint64_t key[4] = {0, 0, 0, 0};
    if (active) {
      const uchar4 m = *reinterpret_cast<const uchar4*>(mask + row0);
      const float4 e4 = *reinterpret_cast<const float4*>(row_err + row0);
      const unsigned char mm[4] = {m.x, m.y, m.z, m.w};
      const float ee[4] = {e4.x, e4.y, e4.z, e4.w};
      float up[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = acc[r];
        const float err = ee[r] * qn + c_fixed + 2.0e-7f * fabsf(a);
        up[r] = mm[r] ? a + err : -__builtin_inff();
        key[r] = mm[r] ? topk_make_key(a - err, row0 + r) : 0ull;
      }
      *reinterpret_cast<float4*>(upper + row0) = make_float4(up[0], up[1], up[2], up[3]);
    }
    const uint64_t thr = active ? mine[k - 1] : ~0ull;
    if (__ballot(key[0] > thr || key[1] > thr || key[2] > thr || key[3] > thr)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) wave_offer(mine, k, key[r], 0, active, lane);
    }
  }
  block_merge_lists(lists, kListLen, kScan16Waves, wave, lane);
  if (wave == 0) cand[static_cast<int64_t>(blockIdx.x) * kListLen + lane] = lists[lane];
}

// ---- stage 2: candidates = rows whose upper bound reaches the k-th best lower bound ----------------

__device__ __forceinline__ float key_score(uint64_t key) {
  const uint32_t hi = static_cast<uint32_t>(key >> 32);
  const uint32_t u = (hi & 0x80000000u) ? (hi ^ 0x80000000u) : ~hi;
  return __uint_as_float(u);
}

// lower_keys: the k best lower-bound keys (descending, zero padded). counter[0] receives the count.
// One float4 of upper bounds per thread (upper[] is padded to a multiple of 64 rows).
__global__ __launch_bounds__(256) void collect_candidates_kernel(const float* __restrict__ upper, int64_t n,
                                                                 const uint64_t* __restrict__ lower_keys, int k,
                                                                 int32_t* __restrict__ rows, int32_t* counter) {
  const uint64_t kth = lower_keys[k - 1];
  // fewer than k rows in play: every row with a finite bound is a candidate
  const float thr = kth ? key_score(kth) : -3.0e38f;
  const int lane = threadIdx.x & 63;
  const int64_t base = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 4;
  float4 u = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
  if (base < n) u = *reinterpret_cast<const float4*>(upper + base);
  const float uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bool take = base + c < n && uu[c] >= thr;  // -inf (masked) never passes
    const uint64_t m = __ballot(take);
    if (m) {
      const int leader = __builtin_ctzll(m);
      int slot0 = 0;
      if (lane == leader) slot0 = atomicAdd(counter, __popcll(m));
      slot0 = __shfl(slot0, leader);
      const int slot = slot0 + __popcll(m & ((1ull << lane) - 1ull));
      if (take && slot < kMaxCandidates) rows[slot] = static_cast<int32_t>(base + c);
    }
  }
}

// ---- stage 3: exact re-score + final ranking ---------------------------------------------------------

// One block per candidate. All 16 waves pull the candidate's tile (and the query image) into LDS in
// one memory round trip — a single wave walking the tile paid one HBM latency per eight k-blocks —
// then wave 0 runs the same MFMA chain as dense.hip's scan_tile over it, in k order.
constexpr int kRescoreThreads = 1024;
constexpr int kRescoreChunk = 64;  // k-blocks staged at a time: 2 KiB of LDS each (tile + query)

__global__ __launch_bounds__(kRescoreThreads) void rescore_kernel(const float4* __restrict__ corpus,
                                                                  const float4* __restrict__ q_img, int kblocks,
                                                                  const int32_t* __restrict__ rows,
                                                                  const int32_t* __restrict__ counter,
                                                                  uint64_t* __restrict__ keys) {
  extern __shared__ float4 stage[];  // [2][chunk][64]
  const int chunk = min(kblocks, kRescoreChunk);
  const int count = min(*counter, kMaxCandidates);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  for (int c = blockIdx.x; c < count; c += gridDim.x) {
    const int64_t row = rows[c];
    const int64_t tile = row / kTileRows;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int kb0 = 0; kb0 < kblocks; kb0 += chunk) {
      const int nkb = min(chunk, kblocks - kb0);
      const float4* a_src = corpus + (tile * kblocks + kb0) * 64;
      const float4* b_src = q_img + kb0 * 64;
      __syncthreads();  // the previous chunk / candidate has been consumed
#pragma unroll 4
      for (int i = threadIdx.x; i < nkb * 64; i += kRescoreThreads) {
        stage[i] = a_src[i];
        stage[chunk * 64 + i] = b_src[i];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll 4
        for (int kb = 0; kb < nkb; ++kb) {
          const float4 a = stage[kb * 64 + lane];
          const float4 b = stage[(chunk + kb) * 64 + lane];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
        }
      }
    }
    const int r = static_cast<int>(row % kTileRows);
    if (wave == 0 && (lane & 15) == 0 && (lane >> 4) == r / 4) keys[c] = topk_make_key(acc[r % 4], row);
  }
}

// single block: rank the exact keys, write the k best (descending, zero padded) and the candidate
// count (so the host can detect an overflow) to the result area
__global__ __launch_bounds__(1024) void rank_candidates_kernel(const uint64_t* __restrict__ keys,
                                                               const int32_t* __restrict__ counter, int k,
                                                               uint64_t* __restrict__ out,
                                                               int32_t* __restrict__ out_count) {
  __shared__ uint64_t s[kMaxCandidates];
  const int total = *counter;
  const int c = min(total, kMaxCandidates);
  for (int i = threadIdx.x; i < c; i += 1024) s[i] = keys[i];
  for (int i = threadIdx.x; i < k; i += 1024) out[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < c; i += 1024) {
    const uint64_t key = s[i];
    int rank = 0;
    for (int j = 0; j < c; ++j) rank += s[j] > key;
    if (rank < k && key != 0) out[rank] = key;
  }
  if (threadIdx.x == 0) *out_count = total;
}

bool prefilter_usable(vr_engine* e, int nq, int k) {
  return e->prefilter && nq == 1 && k <= kFusedMaxK && e->n_rows >= 4096;
}

// q image (f32, tiled) must already be in e->q_tiled. Results: k keys at out_keys_dev and the
// candidate count at out_count_dev (both normally in the pinned result area).
int prefilter_search(vr_engine* e, int k, const uint8_t* mask_dev, uint64_t* out_keys_dev, int32_t* out_count_dev) {
  const int kb16n = e->dim / kK16;
  const int64_t n_tiles = (e->n_rows + kTileRows - 1) / kTileRows;
  VR_TRY(e->upper.grow(e->cap_rows, 0, e->stream));
  VR_TRY(e->cand_rows.grow(kMaxCandidates + 16, 0, e->stream));
  VR_TRY(e->cand_keys.grow(kMaxCandidates, 0, e->stream));
  int64_t blocks = std::min<int64_t>((n_tiles + kScan16Waves - 1) / kScan16Waves, kScanBlocks);
  VR_TRY(e->cand_a.grow(blocks * kListLen, 0, e->stream));
  VR_TRY(e->cand_b.grow(kListLen, 0, e->stream));
  int32_t* counter = e->cand_rows.p + kMaxCandidates;
  hipStream_t s = e->stream;
  const size_t lds = static_cast<size_t>(2) * kb16n * 1024 + kScan16Waves * kListLen * sizeof(uint64_t);
  if (lds > 64 * 1024)
    VR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(prefilter_scan_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  // algorithmic bytes: the f16 shadow once, one error norm, one mask byte, one upper bound per row
  prof_begin(e, VR_PROF_DENSE_SCAN, static_cast<double>(e->n_rows) * (e->dim * 2.0 + 4.0 + 1.0 + 4.0));
  hipLaunchKernelGGL(prefilter_scan_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kScan16Waves * 64), lds, s,
                     reinterpret_cast<const uint4*>(e->corpus16.p), e->q_tiled.p, e->row_err.p, mask_dev, n_tiles, kb16n,
                     e->dim, k, e->upper.p, e->cand_a.p, counter);
  prof_end(e);
  VR_TRY(topk_merge_lists(e, e->cand_a.p, static_cast<int>(blocks), 1, k, e->cand_b.p));
  const int64_t n4 = (e->n_rows + 3) / 4;
  hipLaunchKernelGGL(collect_candidates_kernel, dim3(static_cast<unsigned>((n4 + 255) / 256)), dim3(256), 0, s,
                     e->upper.p, e->n_rows, e->cand_b.p, k, e->cand_rows.p, counter);
  const size_t rescore_lds = static_cast<size_t>(2) * std::min(e->kblocks, kRescoreChunk) * 1024;
  if (rescore_lds > 64 * 1024)
    VR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rescore_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(rescore_lds)));
  hipLaunchKernelGGL(rescore_kernel, dim3(128), dim3(kRescoreThreads), rescore_lds, s,
                     reinterpret_cast<const float4*>(e->corpus.p),
                     reinterpret_cast<const float4*>(e->q_tiled.p), e->kblocks, e->cand_rows.p, counter,
                     e->cand_keys.p);
  hipLaunchKernelGGL(rank_candidates_kernel, dim3(1), dim3(1024), 0, s, e->cand_keys.p, counter, k, out_keys_dev,
                     out_count_dev);
  VR_HIP(hipGetLastError());
  return 0;
}

}  // namespace vr
