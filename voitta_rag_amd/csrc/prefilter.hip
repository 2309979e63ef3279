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
// Stage 3: every 16-row tile holding a candidate is re-scored once with exactly the instruction sequence
//   of the one-stage scan (v_mfma_f32_16x16x4_f32 chain over the f32 corpus), the exact keys of its
//   candidate rows go into per-block top-k lists, and those are merged. More than kMaxCandTiles candidate
//   tiles (e.g. a large corpus of near-duplicates) makes the host fall back to the one-stage scan, so the
//   result never depends on the bound being tight.
// int8 form of stage 0/1 (dim % 64 == 0; VR_PREFILTER=f16 keeps the f16 shadow): x ~ s_r * x8 with one
//   scale per row, x8 in [-127, 127], MFMA-tiled for v_mfma_i32_16x16x64_i8, and err_r = |x - s_r x8|_2
//   exact as before; the query is carried as TWO int8 vectors, q = a qa + b qb + rho with b = a / 254,
//   so its own residual |rho|_2 (~3e-5 |q|) is far below err_r (~7e-3 for a random unit row at D = 768).
//       x.q = s_r (a x8.qa + b x8.qb)  +  s_r x8.rho  +  (x - s_r x8).q
//   the first term is exact in int32, the others are bounded by Cauchy-Schwarz:
//       |x.q - A| <= (|x| + err_r) |rho| + err_r |q|.
//   A quarter of the f32 bytes; the bound is ~25x wider than the f16 one, which on a 1M x 768 corpus of
//   random unit rows means ~50-100 re-scores per query instead of ~12. (A first int8 attempt bounded the
//   error element-wise, growing linearly with D: ~4000 candidates. The row norm of the residual is what
//   makes it usable.)

// Centred shadow (int8 form): the shadow quantises r = x - c for one fixed vector c per engine (vr_engine::centre).
//   x.q = c.q + r.q: the first term is the same real number for every row, so it drops out of "upper bound of row i
//   >= k-th largest lower bound" — the scan works on r.q alone and never computes c.q. Only two things change:
//   err_r = |r - s_r r8|_2 (plus 2^-23 (|x| + |c|) for the rounding of x - c), and |r| <= |x| + |c| in the
//   (|r| + err_r) |rho| term, which the kernels take as |c| |rho| on top of their per-query constant. With c = 0 this
//   is the uncentred scheme. c = column mean of the stored rows: on a collection whose rows share a common
//   direction (pairwise cosine 0.7, as sentence embeddings do) the residuals are ~0.55 of the rows, candidates per
//   query fall from ~6400 to a few hundred, and the batched search stops overflowing its per-query budget
//   (scripts/perf_aniso.py).

#include "engine_internal.h"
#include "topk_device.h"

#include <algorithm>
#include <cmath>
#include <vector>

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
    const float x = corpus[((tile * kblocks + kb) * 64 + tile_pos(kk % 4, r)) * 4 + kk / 4];
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

using i32x4 = __attribute__((ext_vector_type(4))) int;
constexpr int kK8 = 64;  // k per int8 tile block (1 KiB = 16 rows x 64 bytes)

// int8 shadow: one wave per row, two passes over the stored f32 row (max |x|, then quantise)
__global__ __launch_bounds__(256) void shadow_rows8_kernel(const float* __restrict__ corpus, int64_t first_row,
                                                           int64_t n, int dim, int kblocks,
                                                           int8_t* __restrict__ corpus8,
                                                           float* __restrict__ row_scale,
                                                           float* __restrict__ row_err,
                                                           const float* __restrict__ centre, float centre_norm) {
  const int lane = threadIdx.x & 63;
  const int64_t row = first_row + static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= first_row + n) return;
  const int64_t tile = row / kTileRows;
  const int r = static_cast<int>(row % kTileRows);
  const int kb8n = dim / kK8;
  auto at = [&](int k) {  // the residual x - centre (the row itself while there is no centre yet)
    const int kb = k / kTileK, kk = k % kTileK;
    const float x = corpus[((tile * kblocks + kb) * 64 + tile_pos(kk % 4, r)) * 4 + kk / 4];
    return centre ? x - centre[k] : x;
  };
  float mx = 0.0f;
  for (int k = lane; k < dim; k += 64) mx = fmaxf(mx, fabsf(at(k)));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  const bool finite = mx <= 3.0e38f;  // false for inf and NaN rows: stored as zeros with an infinite bound
  const float s = finite ? mx / 127.0f : 0.0f;
  const float inv = finite && mx > 0.0f ? 127.0f / mx : 0.0f;
  float e2 = 0.0f;
  for (int k = lane; k < dim; k += 64) {
    const float x = at(k);
    const float t = fminf(fmaxf(rintf(x * inv), -127.0f), 127.0f);
    const float d = x - s * t;
    e2 += d * d;
    corpus8[((tile * kb8n + k / kK8) * 64 + ((k % kK8) / 16) * 16 + r) * 16 + k % 16] = static_cast<int8_t>(t);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) e2 += __shfl_xor(e2, off);
  if (lane == 0) {
    row_scale[row] = s;
    // rounded up: 0.1 % on the norm for the f32 summation, 2^-22 |x|_2 (<= mx sqrt(D)) for the
    // roundings inside d = x - s t, and an absolute floor
    // ... and 2^-23 (|x|_2 + |centre|_2) for the rounding of x - centre itself (|x|_2 <= 1.001: cosine-preprocessed)
    row_err[row] = finite ? sqrtf(e2) * 1.001f + 2.4e-7f * mx * sqrtf(static_cast<float>(dim)) + 1.0e-12f +
                                (centre ? 1.2e-7f * (1.001f + centre_norm) : 0.0f)
                          : __builtin_inff();
  }
}

int prefilter_store_rows(vr_engine* e, int64_t n, int64_t first_row) {
  if (!e->prefilter || n <= 0) return 0;
  if (e->prefilter8)
    hipLaunchKernelGGL(shadow_rows8_kernel, dim3(static_cast<unsigned>((n + 3) / 4)), dim3(256), 0, e->stream,
                       e->corpus.p, first_row, n, e->dim, e->kblocks, reinterpret_cast<int8_t*>(e->corpus16.p),
                       e->row_scale.p, e->row_err.p, e->centre_rows > 0 ? e->centre.p : static_cast<const float*>(nullptr),
                       e->centre_norm);
  else
    hipLaunchKernelGGL(shadow_rows_kernel, dim3(static_cast<unsigned>((n + 3) / 4)), dim3(256), 0, e->stream,
                       e->corpus.p, first_row, n, e->dim, e->kblocks, reinterpret_cast<half_t*>(e->corpus16.p),
                       e->row_err.p);
  VR_HIP(hipGetLastError());
  return 0;
}

// column sums of rows [0, n_rows) of the tiled f32 corpus: one block per 64 tiles, one thread per position of a tile's
// k-block image, atomics into sum[dim] (a few thousand blocks x dim x 16 adds: microseconds)
__global__ __launch_bounds__(256) void column_sum_kernel(const float* __restrict__ corpus, int64_t n_rows, int kblocks,
                                                         float* __restrict__ sum) {
  const int64_t tile0 = static_cast<int64_t>(blockIdx.x) * 64;
  const int64_t n_tiles = (n_rows + kTileRows - 1) / kTileRows;
  for (int p = threadIdx.x; p < kblocks * 256; p += 256) {
    const int kb = p >> 8, pos = (p >> 2) & 63, c = p & 3;  // pos = tile_pos(g, r) = r * 4 + g
    const int r = pos >> 2;
    float acc = 0.0f;
    for (int64_t t = tile0; t < min(tile0 + 64, n_tiles); ++t)
      if (t * kTileRows + r < n_rows) acc += corpus[(t * kblocks + kb) * 256 + p - (kb << 8)];
    atomicAdd(sum + kb * kTileK + (pos & 3) + 4 * c, acc);  // k = kTileK kb + (k % 4 = g) + 4 (k % 16 / 4 = c)
  }
}

__global__ void centre_from_sum_kernel(const float* __restrict__ sum, int dim, float inv_n, float* __restrict__ centre) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < dim) centre[k] = sum[k] * inv_n;
}

constexpr int64_t kCentreMinRows = 1024;  // below this the shadow stays uncentred (centre = 0)

int prefilter_recentre(vr_engine* e) {
  if (!e->prefilter || !e->prefilter8 || e->n_rows <= 0) return 0;
  const bool centring = !(getenv("VR_PREFILTER_CENTRE") && atoi(getenv("VR_PREFILTER_CENTRE")) == 0);  // (read per call: rare)
  if (!centring) e->centre_rows = 0;
  e->centre_checked_rows = e->n_rows;
  if (centring && e->n_rows >= kCentreMinRows) {
    static_assert(kTileK == 16, "column_sum_kernel decodes the [k/16][lane][4] image");
    VR_TRY(e->centre.grow(e->dim, 0, e->stream));
    VR_TRY(e->centre_sum.grow(e->dim, 0, e->stream));
    VR_HIP(hipMemsetAsync(e->centre_sum.p, 0, sizeof(float) * static_cast<size_t>(e->dim), e->stream));
    const int64_t n_tiles = (e->n_rows + kTileRows - 1) / kTileRows;
    hipLaunchKernelGGL(column_sum_kernel, dim3(static_cast<unsigned>((n_tiles + 63) / 64)), dim3(256), 0, e->stream,
                       e->corpus.p, e->n_rows, e->kblocks, e->centre_sum.p);
    hipLaunchKernelGGL(centre_from_sum_kernel, dim3(static_cast<unsigned>((e->dim + 255) / 256)), dim3(256), 0, e->stream,
                       e->centre_sum.p, e->dim, 1.0f / static_cast<float>(e->n_rows), e->centre.p);
    VR_HIP(hipGetLastError());
    std::vector<float> host(static_cast<size_t>(e->dim));
    VR_HIP(hipMemcpyAsync(host.data(), e->centre.p, sizeof(float) * host.size(), hipMemcpyDeviceToHost, e->stream));
    VR_HIP(hipStreamSynchronize(e->stream));
    double n2 = 0.0;
    bool finite = true;
    for (float v : host) {
      n2 += static_cast<double>(v) * v;
      finite = finite && std::isfinite(v);
    }
    if (!centring) {
      e->centre_norm = 0.0f;
      e->centre_rows = 0;
    } else if (finite) {
      e->centre_norm = static_cast<float>(std::sqrt(n2) * 1.000001 + 1e-30);
      e->centre_rows = e->n_rows;
    } else {  // a row of infinities or NaNs was stored: no centre (its own bound is infinite either way)
      e->centre_norm = 0.0f;
      e->centre_rows = 0;
    }
  }
  return prefilter_store_rows(e, e->n_rows, 0);
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
  if (blockIdx.x == 0 && threadIdx.x == 0) counter[0] = counter[1] = 0;  // candidate counters of the collect pass
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
    const int64_t row0 = tile * kTileRows + (lane >> 4) * 4;  // rows = 4*(lane >> 4) + reg
    uchar4 m = make_uchar4(0, 0, 0, 0);  // requested ahead of the tile: see prefilter_scan8_kernel
    float4 e4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
      m = *reinterpret_cast<const uchar4*>(mask + row0);
      e4 = *reinterpret_cast<const float4*>(row_err + row0);
    }
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
    uint64_t key[4] = {0, 0, 0, 0};
    if (active) {
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

// ---- stage 1, int8 form ----------------------------------------------------------------------------------

// Same walk as prefilter_scan_kernel over 1-KiB blocks of 16 rows x 64 int8; two integer MFMAs per block
// (query parts qa, qb). Every block first quantises the query into its own LDS:
//   t = max |q_i|, a = t / 127, qa = rint(q / a); r = q - a qa; b = a / 254, qb = rint(r / b); rho = r - b qb.
__global__ __launch_bounds__(kScan16Waves * 64) void prefilter_scan8_kernel(
    const uint4* __restrict__ corpus8, const float* __restrict__ q_img, const float* __restrict__ row_err,
    const float* __restrict__ row_scale, const uint8_t* __restrict__ mask, int64_t n_tiles, int kb8n, int dim, int k,
    float* __restrict__ upper, uint64_t* __restrict__ cand, int32_t* __restrict__ counter, float centre_norm) {
  extern __shared__ uint4 q_lds[];                                        // [2][kb8n][64] = 2*kb8n KiB
  __shared__ float red[3][kScan16Waves];
  uint64_t* lists = reinterpret_cast<uint64_t*>(q_lds + 2 * kb8n * 64);  // [waves][kListLen]
  if (blockIdx.x == 0 && threadIdx.x == 0) counter[0] = counter[1] = 0;  // candidate counters of the collect pass
  for (int i = threadIdx.x; i < 2 * kb8n * 64; i += kScan16Waves * 64) q_lds[i] = make_uint4(0, 0, 0, 0);
  for (int i = threadIdx.x; i < kScan16Waves * kListLen; i += kScan16Waves * 64) lists[i] = 0ull;
  auto q_at = [&](int kk) {
    const int kb = kk / kTileK, kr = kk % kTileK;
    return q_img[(kb * 64 + (kr % 4) * 16 + 0) * 4 + kr / 4];
  };
  float mx = 0.0f, n2 = 0.0f;
  for (int kk = threadIdx.x; kk < dim; kk += kScan16Waves * 64) {
    const float q = q_at(kk);
    mx = fmaxf(mx, fabsf(q));
    n2 += q * q;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    mx = fmaxf(mx, __shfl_xor(mx, off));
    n2 += __shfl_xor(n2, off);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = mx;
    red[1][threadIdx.x >> 6] = n2;
  }
  __syncthreads();  // also orders the zero fill above before the quantised bytes below
  float t = 0.0f, qn2 = 0.0f;
#pragma unroll
  for (int w = 0; w < kScan16Waves; ++w) {
    t = fmaxf(t, red[0][w]);
    qn2 += red[1][w];
  }
  const bool q_ok = t > 0.0f && t <= 3.0e38f;  // a zero, infinite or NaN query scans as zeros with infinite bounds
  const float a = q_ok ? t / 127.0f : 0.0f;
  const float inv_a = q_ok ? 127.0f / t : 0.0f;
  const float b = a / 254.0f;
  const float inv_b = a > 0.0f ? 254.0f / a : 0.0f;
  float rho2 = 0.0f;
  {
    int8_t* qa = reinterpret_cast<int8_t*>(q_lds);
    int8_t* qb = qa + kb8n * 1024;
    for (int kk = threadIdx.x; kk < dim; kk += kScan16Waves * 64) {
      const float q = q_at(kk);
      const float ta = fminf(fmaxf(rintf(q * inv_a), -127.0f), 127.0f);
      const float r = q - a * ta;
      const float tb = fminf(fmaxf(rintf(r * inv_b), -127.0f), 127.0f);
      const float rho = r - b * tb;
      rho2 += rho * rho;
      const int at = ((kk / kK8) * 64 + ((kk % kK8) / 16) * 16 + 0) * 16 + kk % 16;
      qa[at] = static_cast<int8_t>(ta);
      qb[at] = static_cast<int8_t>(tb);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) rho2 += __shfl_xor(rho2, off);
  if ((threadIdx.x & 63) == 0) red[2][threadIdx.x >> 6] = rho2;
  __syncthreads();
  float rho2_all = 0.0f;
#pragma unroll
  for (int w = 0; w < kScan16Waves; ++w) rho2_all += red[2][w];
  const float qn = sqrtf(qn2) * 1.0001f + 1.0e-12f;  // |q|_2, rounded up
  // |rho|_2 rounded up: the f32 roundings inside r and rho are each <= 2^-24 of terms no larger than |q_i|
  const float rho_n = q_ok ? sqrtf(rho2_all) * 1.001f + 3.0e-7f * qn + 1.0e-12f : __builtin_inff();
  // apart from the row's own terms: the exact f32 chain errs by at most ~2D roundings of 2^-24 relative to
  // sum |terms| <= 1.001 |q|; forming A = s (a dotA + b dotB) in f32 costs three roundings of its parts
  // (|b dotB s| <= |q| sqrt(D) / 254), charged per row below and here
  // (+ |centre| |rho|: the shadow holds residuals r = x - centre, |r| <= |x| + |centre| — see "Centred shadow" above)
  const float c_fixed = static_cast<float>(dim) * 5.0e-7f * fmaxf(qn, 1.0f) + 1.0e-7f * qn * sqrtf(static_cast<float>(dim)) +
                        centre_norm * rho_n;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  uint64_t* mine = lists + wave * kListLen;
  const bool active = (lane & 15) == 0;  // C/D map: column (query) = lane & 15; only query 0 exists
  const int64_t wave_stride = static_cast<int64_t>(gridDim.x) * kScan16Waves;
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * kScan16Waves + wave; tile < n_tiles; tile += wave_stride) {
    const uint4* src = corpus8 + tile * kb8n * 64 + lane;
    const int64_t row0 = tile * kTileRows + (lane >> 4) * 4;  // rows = 4*(lane >> 4) + reg
    // the per-row words are requested before the tile itself, so that they are there when the MFMA
    // chain ends (asked for afterwards they cost an exposed memory latency per tile)
    uchar4 m = make_uchar4(0, 0, 0, 0);
    float4 e4 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = e4;
    if (active) {
      m = *reinterpret_cast<const uchar4*>(mask + row0);
      e4 = *reinterpret_cast<const float4*>(row_err + row0);
      s4 = *reinterpret_cast<const float4*>(row_scale + row0);
    }
    i32x4 acc_a = {0, 0, 0, 0}, acc_b = {0, 0, 0, 0};
    int kb = 0;
    for (; kb + 6 <= kb8n; kb += 6) {
      uint4 v[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) v[u] = src[(kb + u) * 64];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const uint4 ba = q_lds[(kb + u) * 64 + lane];
        const uint4 bb = q_lds[(kb8n + kb + u) * 64 + lane];
        acc_a = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const i32x4*>(&v[u]),
                                                      *reinterpret_cast<const i32x4*>(&ba), acc_a, 0, 0, 0);
        acc_b = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const i32x4*>(&v[u]),
                                                      *reinterpret_cast<const i32x4*>(&bb), acc_b, 0, 0, 0);
      }
    }
    for (; kb < kb8n; ++kb) {
      const uint4 v = src[kb * 64];
      const uint4 ba = q_lds[kb * 64 + lane];
      const uint4 bb = q_lds[(kb8n + kb) * 64 + lane];
      acc_a = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const i32x4*>(&v),
                                                    *reinterpret_cast<const i32x4*>(&ba), acc_a, 0, 0, 0);
      acc_b = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const i32x4*>(&v),
                                                    *reinterpret_cast<const i32x4*>(&bb), acc_b, 0, 0, 0);
    }
    uint64_t key[4] = {0, 0, 0, 0};
    if (active) {
      const unsigned char mm[4] = {m.x, m.y, m.z, m.w};
      const float ee[4] = {e4.x, e4.y, e4.z, e4.w};
      const float ss[4] = {s4.x, s4.y, s4.z, s4.w};
      float up[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float fa = a * static_cast<float>(acc_a[r]);
        const float fb = b * static_cast<float>(acc_b[r]);
        const float score = ss[r] * (fa + fb);
        const float err = ee[r] * qn + (1.001f + ee[r]) * rho_n + c_fixed + 2.0e-6f * ss[r] * (fabsf(fa) + fabsf(fb));
        up[r] = mm[r] ? score + err : -__builtin_inff();
        key[r] = mm[r] ? topk_make_key(score - err, row0 + r) : 0ull;
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

// lower_keys: the k best lower-bound keys (descending, zero padded). Candidates are gathered PER TILE:
// tiles[slot] = tile index, masks[slot] = its 16-bit row mask; counter[0] = candidate tiles, counter[1] =
// candidate rows. The int8 bounds are wide enough that a tight cluster of stored rows can put tens of
// thousands of rows in play; they share tiles, and stage 3 costs one tile read per TILE, not per row.
// One float4 of upper bounds per thread (upper[] is padded to a multiple of 64 rows), so the four lanes
// 4j .. 4j+3 of a wave cover one tile.
__global__ __launch_bounds__(256) void collect_candidates_kernel(const float* __restrict__ upper, int64_t n,
                                                                 const uint64_t* __restrict__ lower_keys, int k,
                                                                 int32_t* __restrict__ tiles,
                                                                 int32_t* __restrict__ masks, int32_t* counter) {
  const uint64_t kth = lower_keys[k - 1];
  // fewer than k rows in play: every row with a finite bound is a candidate
  const float thr = kth ? key_score(kth) : -3.0e38f;
  const int lane = threadIdx.x & 63;
  const int64_t base = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 4;
  float4 u = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
  if (base < n) u = *reinterpret_cast<const float4*>(upper + base);
  const float uu[4] = {u.x, u.y, u.z, u.w};
  int m = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (base + c < n && uu[c] >= thr) m |= 1 << c;  // -inf (masked) never passes
  if (__ballot(m != 0) == 0) return;
  int m16 = m << (4 * (lane & 3));
  m16 |= __shfl_xor(m16, 1);
  m16 |= __shfl_xor(m16, 2);
  const bool take = (lane & 3) == 0 && m16 != 0;
  const uint64_t takers = __ballot(take);
  int rows_here = take ? __popc(m16) : 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) rows_here += __shfl_xor(rows_here, off);
  const int leader = __builtin_ctzll(takers);
  int slot0 = 0;
  if (lane == leader) {
    slot0 = atomicAdd(counter, __popcll(takers));
    atomicAdd(counter + 1, rows_here);
  }
  slot0 = __shfl(slot0, leader);
  const int slot = slot0 + __popcll(takers & ((1ull << lane) - 1ull));
  if (take && slot < kMaxCandTiles) {
    tiles[slot] = static_cast<int32_t>(base / kTileRows);
    masks[slot] = m16;
  }
}

// ---- stage 3: exact re-score + final ranking ---------------------------------------------------------

// Grid-stride over candidate tiles. All 16 waves pull the tile (and, once, the query image) into LDS in
// one memory round trip — a single wave walking the tile paid one HBM latency per eight k-blocks —
// then wave 0 runs the same MFMA chain as dense.hip's scan_tile over it, in k order, and offers the
// exact keys of the masked rows to the block's top-k list. The lists are merged by topk_merge_lists.
constexpr int kRescoreThreads = 1024;
constexpr int kRescoreBlocks = 128;
constexpr int kRescoreChunk = 64;  // k-blocks staged at a time: 2 KiB of LDS each (tile + query)

__global__ __launch_bounds__(kRescoreThreads) void rescore_kernel(const float4* __restrict__ corpus,
                                                                  const float4* __restrict__ q_img, int kblocks,
                                                                  const int32_t* __restrict__ tiles,
                                                                  const int32_t* __restrict__ masks,
                                                                  const int32_t* __restrict__ counter, int k,
                                                                  uint64_t* __restrict__ lists_out,
                                                                  int32_t* __restrict__ out_count) {
  extern __shared__ float4 stage[];  // [2][chunk][64]
  __shared__ uint64_t best[kListLen];
  const int chunk = min(kblocks, kRescoreChunk);
  const bool q_resident = kblocks <= kRescoreChunk;  // one chunk: the query image is staged once
  const int n_tiles = counter[0];
  const int count = min(n_tiles, kMaxCandTiles);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = n_tiles > kMaxCandTiles ? 0x7fffffff : counter[1];
  if (threadIdx.x < kListLen) best[threadIdx.x] = 0ull;
  bool q_staged = !q_resident;  // a resident query image travels with the block's FIRST tile: one memory round trip, not two
  const bool active = (lane & 15) == 0;  // C/D map: column (query) = lane & 15; only query 0 exists
  for (int c = blockIdx.x; c < count; c += gridDim.x) {
    const int64_t tile = tiles[c];
    const int mask = masks[c];
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int kb0 = 0; kb0 < kblocks; kb0 += chunk) {
      const int nkb = min(chunk, kblocks - kb0);
      const float4* a_src = corpus + (tile * kblocks + kb0) * 64;
      const float4* b_src = q_img + kb0 * 64;
      __syncthreads();  // the previous chunk / tile has been consumed
#pragma unroll 4
      for (int i = threadIdx.x; i < nkb * 64; i += kRescoreThreads) {
        stage[i] = a_src[i];
        if (!q_staged || !q_resident) stage[chunk * 64 + i] = b_src[i];  // (resident: nkb == kblocks, b_src == q_img)
      }
      q_staged = true;
      __syncthreads();
      if (wave == 0) {
#pragma unroll 4
        for (int kb = 0; kb < nkb; ++kb) {
          const float4 a = stage[kb * 64 + tile_pos(lane >> 4, lane & 15)];
          const float4 b = stage[(chunk + kb) * 64 + lane];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
        }
      }
    }
    if (wave == 0) {
      const int r0 = (lane >> 4) * 4;  // rows = 4*(lane >> 4) + reg
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint64_t key = active && ((mask >> (r0 + r)) & 1) ? topk_make_key(acc[r], tile * kTileRows + r0 + r) : 0ull;
        wave_offer(best, k, key, 0, active, lane);
      }
    }
  }
  __syncthreads();
  if (wave == 0) lists_out[static_cast<int64_t>(blockIdx.x) * kListLen + lane] = best[lane];
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
  VR_TRY(e->cand_rows.grow(2 * kMaxCandTiles + 16, 0, e->stream));
  int64_t blocks = std::min<int64_t>((n_tiles + kScan16Waves - 1) / kScan16Waves, kScanBlocks);
  VR_TRY(e->cand_a.grow(std::max<int64_t>(blocks, kRescoreBlocks) * kListLen, 0, e->stream));
  VR_TRY(e->cand_b.grow(kListLen, 0, e->stream));
  int32_t* counter = e->cand_rows.p + 2 * kMaxCandTiles;  // [0] candidate tiles, [1] candidate rows
  hipStream_t s = e->stream;
  if (e->prefilter8) {
    const int kb8n = e->dim / kK8;
    const size_t lds = static_cast<size_t>(2) * kb8n * 1024 + kScan16Waves * kListLen * sizeof(uint64_t);
    if (lds > 64 * 1024)
      VR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(prefilter_scan8_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    // algorithmic bytes: the int8 shadow once, one error norm, one scale, one mask byte, one upper bound per row
    prof_begin(e, VR_PROF_DENSE_SCAN, static_cast<double>(e->n_rows) * (e->dim * 1.0 + 4.0 + 4.0 + 1.0 + 4.0));
    hipLaunchKernelGGL(prefilter_scan8_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kScan16Waves * 64), lds, s,
                       reinterpret_cast<const uint4*>(e->corpus16.p), e->q_tiled.p, e->row_err.p, e->row_scale.p,
                       mask_dev, n_tiles, kb8n, e->dim, k, e->upper.p, e->cand_a.p, counter,
                       e->centre_rows > 0 ? e->centre_norm : 0.0f);
    prof_end(e);
  } else {
    const size_t lds = static_cast<size_t>(2) * kb16n * 1024 + kScan16Waves * kListLen * sizeof(uint64_t);
    if (lds > 64 * 1024)
      VR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(prefilter_scan_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    // algorithmic bytes: the f16 shadow once, one error norm, one mask byte, one upper bound per row
    prof_begin(e, VR_PROF_DENSE_SCAN, static_cast<double>(e->n_rows) * (e->dim * 2.0 + 4.0 + 1.0 + 4.0));
    hipLaunchKernelGGL(prefilter_scan_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kScan16Waves * 64), lds, s,
                       reinterpret_cast<const uint4*>(e->corpus16.p), e->q_tiled.p, e->row_err.p, mask_dev, n_tiles,
                       kb16n, e->dim, k, e->upper.p, e->cand_a.p, counter);
    prof_end(e);
  }
  VR_TRY(topk_merge_lists(e, e->cand_a.p, static_cast<int>(blocks), 1, k, e->cand_b.p));
  const int64_t n4 = (e->n_rows + 3) / 4;
  int32_t* cand_tiles = e->cand_rows.p;
  int32_t* cand_masks = e->cand_rows.p + kMaxCandTiles;
  hipLaunchKernelGGL(collect_candidates_kernel, dim3(static_cast<unsigned>((n4 + 255) / 256)), dim3(256), 0, s,
                     e->upper.p, e->n_rows, e->cand_b.p, k, cand_tiles, cand_masks, counter);
  const size_t rescore_lds = static_cast<size_t>(2) * std::min(e->kblocks, kRescoreChunk) * 1024;
  if (rescore_lds > 64 * 1024)
    VR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rescore_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(rescore_lds)));
  hipLaunchKernelGGL(rescore_kernel, dim3(kRescoreBlocks), dim3(kRescoreThreads), rescore_lds, s,
                     reinterpret_cast<const float4*>(e->corpus.p),
                     reinterpret_cast<const float4*>(e->q_tiled.p), e->kblocks, cand_tiles, cand_masks, counter, k,
                     e->cand_a.p, out_count_dev);
  VR_TRY(topk_merge_lists(e, e->cand_a.p, kRescoreBlocks, 1, k, out_keys_dev));
  VR_HIP(hipGetLastError());
  return 0;
}

}  // namespace vr
