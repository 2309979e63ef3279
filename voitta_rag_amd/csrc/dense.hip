// Dense side of the store: cosine preprocessing + MFMA-tiled layout on insert, and the exact
// f32 brute-force scorer. Replaces the Qdrant server's part of
//   VectorStoreService.store_chunks   (reference: src/voitta/services/vector_store.py:291-313)
//   client.query_points(query=vec)    (reference: src/voitta/services/vector_store.py:612-617,640-645)
//
// HBM layout of the corpus ("MFMA-tiled"): the N x D f32 matrix is cut into 16-row x 16-k blocks
// of 1 KiB. Blocks are ordered [row/16][k/16]; inside a block the float4 at lane l = g*16 + r
// (r = row%16, g = 0..3) holds x[r][16*kb + 4*c + g] for c = 0..3. One wave-wide
// global_load_dwordx4 therefore reads 1 KiB contiguous, and component c of every lane is exactly
// the A operand (A[i = l&15][k = l>>4]) of the c-th v_mfma_f32_16x16x4_f32 of that block, with k
// increasing in natural order across c and kb. Because the f32 MFMA is bit-for-bit a k-ordered
// fmaf chain, score(q, x) == fmaf(q[D-1], x[D-1], ... fmaf(q[0], x[0], 0.0f)) — the definition the
// oracle restates (the C restatement under oracle/, test infrastructure).
//
// Roofline: HBM. Algorithmic bytes per query pass = N * D * 4 (corpus) ; one pass serves up to
// 16 queries (the MFMA N dimension), the MFMA pipe needs 1 KiB per 128 cycles per SIMD
// (= 19.6 TB/s chip-wide), above what HBM delivers, so the kernel stays bandwidth-bound.

#include "engine_internal.h"
#include "topk_device.h"

#include <algorithm>
#include <cfloat>

namespace vr {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// ---- insert path ---------------------------------------------------------------------------

// Qdrant's cosine preprocessing [EXT, SURVEY.md a10]: length2 = sum of x*x accumulated
// sequentially in f32 (multiply and add rounded separately); vectors that are zero or already
// unit length (|length2 - 1| <= 1e-6) are stored untouched, everything else is divided by
// sqrt(length2). One lane walks one row so the summation order is the sequential one.
__global__ __launch_bounds__(64) void row_length_kernel(const float* __restrict__ x, int64_t n,
                                                        int dim, float* __restrict__ length) {
  int64_t row = static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x;
  if (row >= n) return;
  const float4* p = reinterpret_cast<const float4*>(x + row * dim);
  float acc = 0.0f;
  for (int k = 0; k < dim / 4; ++k) {
    float4 v = p[k];
    acc = __fadd_rn(acc, __fmul_rn(v.x, v.x));
    acc = __fadd_rn(acc, __fmul_rn(v.y, v.y));
    acc = __fadd_rn(acc, __fmul_rn(v.z, v.z));
    acc = __fadd_rn(acc, __fmul_rn(v.w, v.w));
  }
  bool keep = (acc < FLT_EPSILON) || (fabsf(__fadd_rn(acc, -1.0f)) <= 1.0e-6f);
  length[row] = keep ? 0.0f : __fsqrt_rn(acc);
}

// One wave writes one 1-KiB block (tile, kb). Rows outside [first_row, first_row + n) of a
// shared first/last tile are left alone.
__global__ __launch_bounds__(256) void tile_write_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ length,
                                                         int64_t n, int dim, int kblocks,
                                                         int64_t first_row, int64_t first_tile,
                                                         int64_t n_blocks,
                                                         float* __restrict__ corpus) {
  int lane = threadIdx.x & 63;
  int64_t blk = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (blk >= n_blocks) return;
  int64_t tile = first_tile + blk / kblocks;
  int kb = static_cast<int>(blk % kblocks);
  int r = lane & 15, g = lane >> 4;
  int64_t row = tile * kTileRows + r;
  int64_t local = row - first_row;
  if (local < 0 || local >= n) return;
  const float* src = x + local * dim + kb * kTileK + g;
  float len = length[local];
  float4 v;
  v.x = src[0];
  v.y = src[4];
  v.z = src[8];
  v.w = src[12];
  if (len > 0.0f) {
    v.x = __fdiv_rn(v.x, len);
    v.y = __fdiv_rn(v.y, len);
    v.z = __fdiv_rn(v.z, len);
    v.w = __fdiv_rn(v.w, len);
  }
  reinterpret_cast<float4*>(corpus)[(tile * kblocks + kb) * 64 + tile_pos(g, r)] = v;
}

int dense_store_rows(vr_engine* e, const float* x_dev, int64_t n, int64_t first_row) {
  if (n <= 0) return 0;
  VR_TRY(e->stage_len.grow(n, 0, e->stream));
  hipLaunchKernelGGL(row_length_kernel, dim3(static_cast<unsigned>((n + 63) / 64)), dim3(64), 0,
                     e->stream, x_dev, n, e->dim, e->stage_len.p);
  int64_t first_tile = first_row / kTileRows;
  int64_t last_tile = (first_row + n - 1) / kTileRows;
  int64_t n_blocks = (last_tile - first_tile + 1) * e->kblocks;
  hipLaunchKernelGGL(tile_write_kernel, dim3(static_cast<unsigned>((n_blocks + 3) / 4)), dim3(256),
                     0, e->stream, x_dev, e->stage_len.p, n, e->dim, e->kblocks, first_row,
                     first_tile, n_blocks, e->corpus.p);
  VR_HIP(hipGetLastError());
  return 0;
}

// The query block goes through exactly the same preprocessing and tiling as a 16-row corpus
// tile (Qdrant preprocesses the query vector with the same cosine rule [EXT]); rows >= nq are 0.
// One launch on the latency path of every query: stage the <=16 raw query rows in LDS (coalesced),
// let lane r walk row r sequentially (the same length2 order as row_length_kernel, but out of LDS
// instead of a dependent chain of global loads), then write the whole tiled image.
__global__ __launch_bounds__(256) void query_image_kernel(const float* __restrict__ q, int nq, int dim,
                                                          int kblocks, float* __restrict__ image) {
  extern __shared__ __align__(16) float q_raw[];  // [nq][dim]
  __shared__ float len[kQueryBlock];
  for (int i = threadIdx.x; i < nq * dim; i += 256) q_raw[i] = q[i];
  __syncthreads();
  if (static_cast<int>(threadIdx.x) < nq) {
    const float4* p = reinterpret_cast<const float4*>(q_raw + threadIdx.x * dim);  // dim % 16 == 0
    float acc = 0.0f;
#pragma unroll 8
    for (int k = 0; k < dim / 4; ++k) {  // wide LDS reads; the additions stay one chain in k order
      const float4 v = p[k];
      acc = __fadd_rn(acc, __fmul_rn(v.x, v.x));
      acc = __fadd_rn(acc, __fmul_rn(v.y, v.y));
      acc = __fadd_rn(acc, __fmul_rn(v.z, v.z));
      acc = __fadd_rn(acc, __fmul_rn(v.w, v.w));
    }
    const bool keep = (acc < FLT_EPSILON) || (fabsf(__fadd_rn(acc, -1.0f)) <= 1.0e-6f);
    len[threadIdx.x] = keep ? 0.0f : __fsqrt_rn(acc);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kblocks * 64; i += 256) {
    const int kb = i >> 6, lane = i & 63;
    const int r = lane & 15, g = lane >> 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nq) {
      const float* src = q_raw + r * dim + kb * kTileK + g;
      v = make_float4(src[0], src[4], src[8], src[12]);
      const float l = len[r];
      if (l > 0.0f) {
        v.x = __fdiv_rn(v.x, l);
        v.y = __fdiv_rn(v.y, l);
        v.z = __fdiv_rn(v.z, l);
        v.w = __fdiv_rn(v.w, l);
      }
    }
    reinterpret_cast<float4*>(image)[i] = v;
  }
}

int dense_make_query_image(vr_engine* e, const float* q_dev, int nq) {
  VR_CHECK(nq >= 1 && nq <= kQueryBlock, "query block of %d not in 1..16", nq);
  size_t img = static_cast<size_t>(e->kblocks) * 256;
  VR_TRY(e->q_tiled.grow(static_cast<int64_t>(img), 0, e->stream));
  const size_t lds = static_cast<size_t>(nq) * e->dim * sizeof(float);
  hipLaunchKernelGGL(query_image_kernel, dim3(1), dim3(256), lds, e->stream, q_dev, nq, e->dim, e->kblocks,
                     e->q_tiled.p);
  VR_HIP(hipGetLastError());
  return 0;
}

// ---- scorer --------------------------------------------------------------------------------

constexpr int kScoreUnroll = 8;

// scores of one 16-row tile against the <=16 queries of the LDS image: D/4 back-to-back
// v_mfma_f32_16x16x4_f32 on one accumulator = the k-ordered f32 fma chain of every (row, query).
__device__ __forceinline__ f32x4 scan_tile(const float4* __restrict__ src,
                                           const float4* __restrict__ q_lds, int kblocks, int lane) {
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  int kb = 0;
  for (; kb + kScoreUnroll <= kblocks; kb += kScoreUnroll) {
    float4 a[kScoreUnroll];
#pragma unroll
    for (int u = 0; u < kScoreUnroll; ++u) a[u] = src[(kb + u) * 64];
#pragma unroll
    for (int u = 0; u < kScoreUnroll; ++u) {
      float4 b = q_lds[(kb + u) * 64 + lane];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b.w, acc, 0, 0, 0);
    }
  }
  for (; kb < kblocks; ++kb) {
    float4 a = src[kb * 64];
    float4 b = q_lds[kb * 64 + lane];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  }
  return acc;
}

// grid-stride over 16-row tiles, one tile per wave per iteration. LDS holds the query image
// (kblocks KiB), read back as the B operand with conflict-free ds_read_b128.
__global__ __launch_bounds__(256) void dense_scores_kernel(const float4* __restrict__ corpus,
                                                           const float4* __restrict__ q_img,
                                                           const uint8_t* __restrict__ mask,
                                                           float* __restrict__ scores,
                                                           int64_t n_tiles, int kblocks,
                                                           int64_t stride, int nq) {
  extern __shared__ float4 q_lds[];
  for (int i = threadIdx.x; i < kblocks * 64; i += 256) q_lds[i] = q_img[i];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t wave_stride = static_cast<int64_t>(gridDim.x) * 4;
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * 4 + wave; tile < n_tiles;
       tile += wave_stride) {
    const f32x4 acc = scan_tile(corpus + tile * kblocks * 64 + tile_pos(lane >> 4, lane & 15), q_lds, kblocks, lane);
    // C/D map of the 16x16 MFMA: column (query) = lane & 15, rows = 4*(lane >> 4) + reg.
    const int q = lane & 15;
    if (q < nq) {
      const int64_t row0 = tile * kTileRows + (lane >> 4) * 4;
      const uchar4 m = *reinterpret_cast<const uchar4*>(mask + row0);
      const float ninf = -__builtin_inff();
      float4 out;
      out.x = m.x ? acc[0] : ninf;
      out.y = m.y ? acc[1] : ninf;
      out.z = m.z ? acc[2] : ninf;
      out.w = m.w ? acc[3] : ninf;
      *reinterpret_cast<float4*>(scores + q * stride + row0) = out;
    }
  }
}

int dense_scores(vr_engine* e, int nq, const uint8_t* mask_dev) {
  int64_t n_tiles = (e->n_rows + kTileRows - 1) / kTileRows;
  if (n_tiles == 0) return 0;
  VR_TRY(e->scores.grow(e->cap_rows * kQueryBlock, 0, e->stream));
  size_t lds = static_cast<size_t>(e->kblocks) * 1024;
  int64_t blocks = (n_tiles + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  // algorithmic bytes of one pass: the corpus rows once (N * D * 4) plus the mask and the scores
  prof_begin(e, VR_PROF_DENSE_SCAN,
             static_cast<double>(e->n_rows) * (e->dim * 4.0 + 1.0 + 4.0 * nq));
  hipLaunchKernelGGL(dense_scores_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), lds,
                     e->stream, reinterpret_cast<const float4*>(e->corpus.p),
                     reinterpret_cast<const float4*>(e->q_tiled.p), mask_dev, e->scores.p, n_tiles,
                     e->kblocks, e->cap_rows, nq);
  prof_end(e);
  VR_HIP(hipGetLastError());
  return 0;
}

// ---- fused scan + selection (k <= 64) ------------------------------------------------------------

// Same scan, but instead of writing N scores per query every wave keeps the k best keys of the
// rows it saw in an LDS list per query (topk_device.h); at the end the block's four lists are
// folded into one and written out: cand[q][block][64], sorted descending, zero padded.
// merge_lists_kernel (topk.hip) then reduces the gridDim.x lists of a query to its top k.
constexpr int kScanWaves = 8;  // 512-thread blocks: 2 per CU keep 16 waves of loads in flight

__global__ __launch_bounds__(kScanWaves * 64) void dense_scan_topk_kernel(
    const float4* __restrict__ corpus, const float4* __restrict__ q_img,
    const uint8_t* __restrict__ mask, int64_t n_tiles, int kblocks, int nq, int k,
    uint64_t* __restrict__ cand) {
  extern __shared__ float4 q_lds[];
  uint64_t* lists = reinterpret_cast<uint64_t*>(q_lds + kblocks * 64);  // [waves][nq][kListLen]
  for (int i = threadIdx.x; i < kblocks * 64; i += kScanWaves * 64) q_lds[i] = q_img[i];
  for (int i = threadIdx.x; i < kScanWaves * nq * kListLen; i += kScanWaves * 64) lists[i] = 0ull;
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  uint64_t* mine = lists + wave * nq * kListLen;
  const int q = lane & 15;
  const bool active = q < nq;
  const int64_t wave_stride = static_cast<int64_t>(gridDim.x) * kScanWaves;
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * kScanWaves + wave; tile < n_tiles;
       tile += wave_stride) {
    const f32x4 acc = scan_tile(corpus + tile * kblocks * 64 + tile_pos(lane >> 4, lane & 15), q_lds, kblocks, lane);
    const int64_t row0 = tile * kTileRows + (lane >> 4) * 4;
    const uchar4 m = *reinterpret_cast<const uchar4*>(mask + row0);
    const uint64_t k0 = m.x ? topk_make_key(acc[0], row0) : 0ull;
    const uint64_t k1 = m.y ? topk_make_key(acc[1], row0 + 1) : 0ull;
    const uint64_t k2 = m.z ? topk_make_key(acc[2], row0 + 2) : 0ull;
    const uint64_t k3 = m.w ? topk_make_key(acc[3], row0 + 3) : 0ull;
    // once the lists have warmed up almost no tile holds a candidate: one ballot decides
    const uint64_t thr = active ? mine[q * kListLen + (k - 1)] : ~0ull;
    if (__ballot(k0 > thr || k1 > thr || k2 > thr || k3 > thr)) {
      wave_offer(mine, k, k0, q, active, lane);
      wave_offer(mine, k, k1, q, active, lane);
      wave_offer(mine, k, k2, q, active, lane);
      wave_offer(mine, k, k3, q, active, lane);
    }
  }
  for (int qq = 0; qq < nq; ++qq) {  // fold the waves' lists of each query: log2(waves) merge rounds
    block_merge_lists(lists + qq * kListLen, nq * kListLen, kScanWaves, wave, lane);
    if (wave == 0)
      cand[(static_cast<int64_t>(qq) * gridDim.x + blockIdx.x) * kListLen + lane] = lists[qq * kListLen + lane];
  }
}

int dense_scan_topk(vr_engine* e, int nq, int k, const uint8_t* mask_dev, uint64_t* out_keys_dev) {
  VR_CHECK(k >= 1 && k <= kListLen, "fused selection serves k <= %d", kListLen);
  int64_t n_tiles = (e->n_rows + kTileRows - 1) / kTileRows;
  int64_t blocks = std::min<int64_t>((n_tiles + kScanWaves - 1) / kScanWaves, kScanBlocks);
  if (blocks < 1) blocks = 1;
  VR_TRY(e->cand_a.grow(static_cast<int64_t>(nq) * blocks * kListLen, 0, e->stream));
  size_t lds = static_cast<size_t>(e->kblocks) * 1024 +
               static_cast<size_t>(kScanWaves) * nq * kListLen * sizeof(uint64_t);
  VR_CHECK(lds <= 160 * 1024, "query image + lists need %zu bytes of LDS", lds);
  if (lds > 64 * 1024)  // beyond the default dynamic-LDS limit
    VR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dense_scan_topk_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  prof_begin(e, VR_PROF_DENSE_SCAN, static_cast<double>(e->n_rows) * (e->dim * 4.0 + 1.0));
  hipLaunchKernelGGL(dense_scan_topk_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kScanWaves * 64), lds,
                     e->stream, reinterpret_cast<const float4*>(e->corpus.p),
                     reinterpret_cast<const float4*>(e->q_tiled.p), mask_dev, n_tiles, e->kblocks, nq, k,
                     e->cand_a.p);
  prof_end(e);
  VR_HIP(hipGetLastError());
  return topk_merge_lists(e, e->cand_a.p, static_cast<int>(blocks), nq, k, out_keys_dev);
}

// ---- read-back -------------------------------------------------------------------------------

__global__ void read_rows_kernel(const float* __restrict__ corpus, const int64_t* __restrict__ rows,
                                 int64_t n, int dim, int kblocks, float* __restrict__ out) {
  int64_t i = blockIdx.x;
  int64_t row = rows[i];
  int64_t tile = row / kTileRows;
  int r = static_cast<int>(row % kTileRows);
  for (int k = threadIdx.x; k < dim; k += blockDim.x) {
    int kb = k / kTileK, kk = k % kTileK;
    int c = kk / 4, g = kk % 4;
    out[i * dim + k] = corpus[((tile * kblocks + kb) * 64 + tile_pos(g, r)) * 4 + c];
  }
}

int dense_read_rows(vr_engine* e, const int64_t* rows_dev, int64_t n, float* out_dev) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(read_rows_kernel, dim3(static_cast<unsigned>(n)), dim3(256), 0, e->stream,
                     e->corpus.p, rows_dev, n, e->dim, e->kblocks, out_dev);
  VR_HIP(hipGetLastError());
  return 0;
}

}  // namespace vr
