// Batched dense search: hundreds to thousands of queries against the whole corpus in one pass —
// BASELINE configs[4] ("1k batched queries, QPS + recall@10"), the batched form of
// client.query_points(query=vec, limit=k) (reference: src/voitta/services/vector_store.py:612-617,640-645;
// the caller that would batch is the MCP search tool, mcp_server.py:469-485, under load).
//
// The answer is, bit for bit, what the one-stage f32 scan (dense.hip) returns for every query; the
// work is organised as the two-stage search of prefilter.hip, with stage 1 turned into a real
// integer GEMM on the matrix cores:
//   prep     every query is cosine-preprocessed exactly as dense.hip does it (sequential f32 sum of
//            squares, the same keep rule and divisions), kept as q^ (f32) and quantised into TWO int8
//            vectors, q^ = a qa + b qb + rho (b = a / 254), tiled for v_mfma_i32_16x16x64_i8 like the
//            int8 shadow of the corpus, with its constants (a, b, |q^|, |rho|, fixed error term).
//   scan     [N rows x D] int8 shadow  x  [D x 2Q] int8 query parts: 256 rows x 128 queries per block
//            (8 waves as 2 x 4, 128 rows x 32 queries x {a, b} per wave = 128 accumulator registers),
//            128-deep K-tiles staged through two 64-KiB LDS buffers by direct-to-LDS loads (the shadow
//            and the query images are already MFMA-operand shaped 1-KiB blocks: linear LDS images,
//            conflict-free 16-byte fragment reads). The epilogue turns the two exact int32 dot products
//            into the approximate score A = s_r (a dot_a + b dot_b) and the CERTAIN bound E of
//            prefilter_scan8_kernel (residual norms by Cauchy-Schwarz; same formula, same constants).
//            Its epilogue leaves two things:
//              per (128-row slab, query): the best LOWER bound A - E  ->  T_q = k-th largest of them. Each is the
//                      lower bound of a different row, so at least k rows score >= T_q;
//              per (16-row tile, query): the best UPPER bound A + E, as an f16 rounded up.
//   revisit  rows with A + E >= T_q are the candidates of query q (a superset of its top k), and they can only sit in
//            tiles whose stored bound reaches T_q: batch_flag_kernel lists those (tile, query) pairs — about as many as
//            candidates, tens per query — and batch_pairs_kernel recomputes their 16 rows' bounds (v_dot4_i32_i8, the
//            same bound function). Until r02j the GEMM simply ran a second time for this (VR_BATCH_TWO_PASS=1 still
//            does): the revisit costs 0.25 ms where the second pass cost 2.4 (1k queries, a million rows). Storing all
//            N x Q bounds instead would be 4 GB per 1k queries at 1M rows; one f16 per tile and query is 131 MB.
//   rescore  every (query, candidate row) pair is scored with the exact k-ordered f32 fma chain (one lane per
//            pair; the MFMA chain of dense.hip is bit for bit this chain) and ranked by the usual 64-bit keys.
// A query whose candidates overflow its budget (a corpus of near-duplicates) is flagged; the host redoes
// those with the one-stage 16-query scan, so the result never depends on the bound being tight.
//
// Roofline: int8 MFMA. Algorithmic work = 2 N D Q operations (SURVEY.md §8d counts the dense top-k of Q batched
// queries as 2 N D Q); executed: 2 query parts = 2x that on the int8 pipe.

#include "engine_internal.h"
#include "topk_device.h"

#include <algorithm>
#include <cfloat>

#include <hip/hip_fp16.h>

namespace vr {

using i32x4 = __attribute__((ext_vector_type(4))) int;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kBQ = 128;          // queries per block column
constexpr int kBStage = 64 * 1024;  // bytes per stage: 16 tiles x 2 kb8 KiB of rows + 8 qfrags x {a,b} x 2 kb8 KiB
constexpr int kQParams = 8;       // floats per query: a, b, |q|, |rho|, c_fixed, (3 spare)

// ---- prep ---------------------------------------------------------------------------------------------

// one block per query. Writes q^ (natural order), the two int8 images and the constants.
__global__ __launch_bounds__(256) void batch_prep_kernel(const float* __restrict__ q, int nq, int dim, int kb8n,
                                                         float* __restrict__ qhat, int8_t* __restrict__ img_a,
                                                         int8_t* __restrict__ img_b, float* __restrict__ params,
                                                         float centre_norm) {
  extern __shared__ __align__(16) float row[];  // [dim]
  __shared__ float red[3][4];
  __shared__ float len_s;
  const int qi = blockIdx.x;
  for (int i = threadIdx.x; i < dim; i += 256) row[i] = q[static_cast<int64_t>(qi) * dim + i];
  __syncthreads();
  if (threadIdx.x == 0) {  // the sequential chain of row_length_kernel / query_image_kernel (dense.hip)
    const float4* p = reinterpret_cast<const float4*>(row);
    float acc = 0.0f;
    for (int k = 0; k < dim / 4; ++k) {
      const float4 v = p[k];
      acc = __fadd_rn(acc, __fmul_rn(v.x, v.x));
      acc = __fadd_rn(acc, __fmul_rn(v.y, v.y));
      acc = __fadd_rn(acc, __fmul_rn(v.z, v.z));
      acc = __fadd_rn(acc, __fmul_rn(v.w, v.w));
    }
    const bool keep = (acc < FLT_EPSILON) || (fabsf(__fadd_rn(acc, -1.0f)) <= 1.0e-6f);
    len_s = keep ? 0.0f : __fsqrt_rn(acc);
  }
  __syncthreads();
  const float len = len_s;
  float mx = 0.0f, n2 = 0.0f;
  for (int i = threadIdx.x; i < dim; i += 256) {
    float v = row[i];
    if (len > 0.0f) v = __fdiv_rn(v, len);
    row[i] = v;
    // q^ as the B operand of the exact re-score's v_mfma_f32_16x16x4_f32 chain: [kb][g] float4 whose component c is
    // element 16 kb + 4 c + g (what a lane of k-group g supplies for the c-th MFMA of k-block kb)
    qhat[static_cast<int64_t>(qi) * dim + (i / 16) * 16 + (i % 4) * 4 + (i % 16) / 4] = v;
    mx = fmaxf(mx, fabsf(v));
    n2 += v * v;
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
  __syncthreads();
  const float t = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  const float qn2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  // the quantisation of prefilter_scan8_kernel, term for term
  const bool q_ok = t > 0.0f && t <= 3.0e38f;
  const float a = q_ok ? t / 127.0f : 0.0f;
  const float inv_a = q_ok ? 127.0f / t : 0.0f;
  const float b = a / 254.0f;
  const float inv_b = a > 0.0f ? 254.0f / a : 0.0f;
  float rho2 = 0.0f;
  const int frag = qi >> 4, col = qi & 15;
  for (int kk = threadIdx.x; kk < dim; kk += 256) {
    const float v = row[kk];
    const float ta = fminf(fmaxf(rintf(v * inv_a), -127.0f), 127.0f);
    const float r = v - a * ta;
    const float tb = fminf(fmaxf(rintf(r * inv_b), -127.0f), 127.0f);
    const float rho = r - b * tb;
    rho2 += rho * rho;
    // [qfrag][kb8][lane = (k % 64) / 16 * 16 + query % 16][16 bytes]
    const int64_t at = ((static_cast<int64_t>(frag) * kb8n + kk / 64) * 64 + ((kk % 64) / 16) * 16 + col) * 16 + kk % 16;
    img_a[at] = static_cast<int8_t>(ta);
    img_b[at] = static_cast<int8_t>(tb);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) rho2 += __shfl_xor(rho2, off);
  if ((threadIdx.x & 63) == 0) red[2][threadIdx.x >> 6] = rho2;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float rho2_all = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    const float qn = sqrtf(qn2) * 1.0001f + 1.0e-12f;
    const float rho_n = q_ok ? sqrtf(rho2_all) * 1.001f + 3.0e-7f * qn + 1.0e-12f : __builtin_inff();
    const float c_fixed = static_cast<float>(dim) * 5.0e-7f * fmaxf(qn, 1.0f) + 1.0e-7f * qn * sqrtf(static_cast<float>(dim)) +
                          centre_norm * rho_n;  // (the centred shadow's |centre| |rho| term, as in prefilter_scan8_kernel)
    float* p = params + static_cast<int64_t>(qi) * kQParams;
    p[0] = a;
    p[1] = b;
    p[2] = qn;
    p[3] = rho_n;
    p[4] = c_fixed;
    // forming A = s (a dot_a + b dot_b) in f32 costs three roundings (2e-6 is 30x their sum) of s (|a dot_a| + |b dot_b|)
    // <= |s x8| (|a qa| + |b qb|) (Cauchy-Schwarz) <= (2.01 + 2 |centre|) (1.25 |q|): stored rows have |x| <= 1 + 1e-6,
    // the shadow holds s x8 ~ x - centre within its residual (itself at most |x - centre|), a qa is q within a sqrt(D) / 2
    // and |b qb| <= a sqrt(D) / 2 <= 0.126 |q| for D <= 1024. A constant per query instead of four operations per element.
    p[5] = 2.0e-6f * (2.01f + 2.0f * centre_norm) * 1.25f * qn;
  }
}

// ---- scan ---------------------------------------------------------------------------------------------

// The approximate score of one (row, query) from its two exact int32 dot products, and the CERTAIN bound on its
// distance from the exact score: prefilter_scan8_kernel's E = e_r |q| + (1.001 + e_r) |rho| + c_fixed + 2e-6 s_r (|fa| + |fb|),
// regrouped per query as e_r P1 + P2 + ... with P1 = |q| + |rho|, P2 = 1.001 |rho| + c_fixed (both rounded UP by three
// ulps, which covers the two fused operations that replace five). One function, so that the scan and the revisit of
// the flagged tiles (batch_pairs_kernel) compute the same bits.
__device__ __forceinline__ void batch_query_consts(const float* __restrict__ p, float& pa, float& pb, float& p1, float& p2) {
  pa = p[0];
  pb = p[1];
  p1 = (p[2] + p[3]) * 1.0000004f;
  p2 = (1.001f * p[3] + p[4] + p[5]) * 1.0000004f;  // p[5]: the rounding of forming A, bounded per QUERY (batch_prep_kernel)
}
__device__ __forceinline__ void batch_bound(float pa, float pb, float p1, float p2, float ss, float ee, int dot_a, int dot_b,
                                            float& score, float& err) {
  const float fa = pa * static_cast<float>(dot_a);
  score = ss * fmaf(pb, static_cast<float>(dot_b), fa);
  err = fmaf(ee, p1, p2);
}

__device__ __forceinline__ void glds16b(const void* src, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// Work items: id -> (row block of 256 rows, chunk of 128 queries), interleaved so that the chunks of one row block have ids
// 8 apart — one XCD under round-robin placement — and run side by side there (they share the rows through its L2), see
// `decode` below. The blocks are PERSISTENT: block b takes the items b, b + gridDim.x, ... and treats the K-steps of all
// of them as one stream through the two stage buffers — the first K-step of the next item is requested during the last
// K-step of the current one, like any other next step, so only a block's very first item pays a prologue, and the stores
// of an item's epilogue drain under the next item's main loop (one-item blocks: 4k cycles of prologue and 2-3k of drain per
// 39k-cycle item, phase stamps of round 3 — profiles/r03_batch_scan_stamps.txt).
// PASS 1: best[q][128-row slab] = max lower bound (slab = row block x wave row). PASS 2: rows with upper bound >= thr[q] -> cand[q][...].
template <int PASS>
__global__ __launch_bounds__(512) void batch_scan_kernel(
    const uint4* __restrict__ corpus8, const uint4* __restrict__ img_a, const uint4* __restrict__ img_b,
    const float* __restrict__ params, const float* __restrict__ row_err, const float* __restrict__ row_scale,
    const uint8_t* __restrict__ mask, int64_t n_tiles, int n_rb, int rb_stride, int n_qc, int nq, int kb8n,
    float* __restrict__ best, const float* __restrict__ thr, int32_t* __restrict__ cand, int32_t* __restrict__ cand_cnt,
    __half* __restrict__ tile_ub, unsigned long long* __restrict__ stamps, int n_ids) {
  __shared__ uint4 lds[2 * kBStage / 16];  // the only LDS object (direct-to-LDS loads in flight beside fragment reads)
  // diagnostics (VR_BATCH_STAMPS=1): every 61st block's thread 0 writes the shader clock at the phase boundaries of its FIRST item
  bool stamping = stamps != nullptr && blockIdx.x % 61 == 0 && threadIdx.x == 0;
  unsigned long long* my_stamps = stamps + (blockIdx.x / 61) * 16;
  int stamp_at = 0;
#define VR_STAMP() do { if (stamping) my_stamps[stamp_at++] = __builtin_readcyclecounter(); } while (0)
  VR_STAMP();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int g = lane >> 4;
  const int nk = kb8n / 2;  // 128-deep K-tiles (kb8n is even: dim % 128 == 0)
  const int64_t last_tile = n_tiles - 1;
  const int nq_pad = n_qc * kBQ;
  // ids b and b + 8 share an XCD: the n_qc query chunks of a row block get ids 8 apart. n_rb row blocks take part:
  // every rb_stride-th of the corpus (a sampled pass 1)
  auto decode = [&](int id, int& rbi_, int& qc_) {
    const int lane8 = id & 7, rest = id >> 3;
    qc_ = rest % n_qc;
    rbi_ = (rest / n_qc) * 8 + lane8;
  };
  auto next_item = [&](int id) {  // the next id of this block that names a row block of the corpus
    for (id += gridDim.x; id < n_ids; id += gridDim.x) {
      int r, q;
      decode(id, r, q);
      if (r < n_rb) return id;
    }
    return n_ids;
  };
  int id = static_cast<int>(blockIdx.x) - static_cast<int>(gridDim.x);
  id = next_item(id);
  if (id >= n_ids) return;  // block-uniform

  // staging: wave w moves the 1-KiB blocks (row tile w, w + 8) x (kb8 0, 1) of the shadow and the blocks
  // (qfrag w) x {a, b} x (kb8 0, 1) of the query images; every block is one wave-wide 16-byte-per-lane load.
  // LDS image of a stage, in 1-KiB blocks: rows  [tile 0..15][kb 0..1] = block 2 t + c;
  //                                        query [qfrag 0..7][part a, b][kb 0..1] = block 32 + 4 f + 2 p + c
  auto stage = [&](int buf, int rbi_, int qc_, int kt) {
    const int64_t t0 = static_cast<int64_t>(rbi_) * rb_stride * 16;
    const uint4* gA0 = corpus8 + std::min<int64_t>(t0 + wave, last_tile) * kb8n * 64 + lane;
    const uint4* gA1 = corpus8 + std::min<int64_t>(t0 + wave + 8, last_tile) * kb8n * 64 + lane;
    const int64_t qf = static_cast<int64_t>(qc_) * 8 + wave;  // (images are padded to whole chunks of 128 queries)
    const uint4* gBa = img_a + qf * kb8n * 64 + lane;
    const uint4* gBb = img_b + qf * kb8n * 64 + lane;
    uint4* d = lds + buf * (kBStage / 16);
    const int k0 = 2 * kt * 64;
    glds16b(gA0 + k0, d + (2 * wave) * 64);
    glds16b(gA0 + k0 + 64, d + (2 * wave + 1) * 64);
    glds16b(gA1 + k0, d + (2 * (wave + 8)) * 64);
    glds16b(gA1 + k0 + 64, d + (2 * (wave + 8) + 1) * 64);
    glds16b(gBa + k0, d + (32 + 4 * wave) * 64);
    glds16b(gBa + k0 + 64, d + (32 + 4 * wave + 1) * 64);
    glds16b(gBb + k0, d + (32 + 4 * wave + 2) * 64);
    glds16b(gBb + k0 + 64, d + (32 + 4 * wave + 3) * 64);
  };

  int rbi, qc;
  decode(id, rbi, qc);
  int step = 0;  // K-steps taken so far: step & 1 is the stage buffer of the next one
  // The K-steps of an item are taken in an order rotated by its row block: the integer sums do not care, and items that
  // share a query chunk do not all ask for the same lines of its image at the same moment.
  stage(0, rbi, qc, rbi % nk);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  VR_STAMP();
  while (true) {
    const int rot = rbi % nk;
    const int64_t tile0 = static_cast<int64_t>(rbi) * rb_stride * 16;
    const int next_id = next_item(id);
    int nrbi = 0, nqc = 0;
    if (next_id < n_ids) decode(next_id, nrbi, nqc);
    // the constants of this lane's two queries (needed by the epilogue only; requested now, so that they are there)
    float pa[2], pb[2], p1[2], p2[2], pthr[2];
    int qidx[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      qidx[f] = qc * kBQ + (2 * wn + f) * 16 + (lane & 15);
      batch_query_consts(params + static_cast<int64_t>(std::min(qidx[f], nq - 1)) * kQParams, pa[f], pb[f], p1[f], p2[f]);
      pthr[f] = PASS == 2 ? thr[std::min(qidx[f], nq - 1)] : 0.0f;
    }
    i32x4 acc[8][2][2];  // [row tile of the wave][qfrag of the wave][part a, b]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int f = 0; f < 2; ++f) acc[i][f][0] = acc[i][f][1] = i32x4{0, 0, 0, 0};

    for (int kt = 0; kt < nk; ++kt, ++step) {
      const uint4* st = lds + (step & 1) * (kBStage / 16);
      if (kt + 1 < nk) stage((step + 1) & 1, rbi, qc, (kt + 1 + rot) % nk);
      else if (next_id < n_ids) stage((step + 1) & 1, nrbi, nqc, nrbi % nk);  // the next item's first K-step
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        uint4 bf[2][2];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int p = 0; p < 2; ++p) bf[f][p] = st[(32 + 4 * (2 * wn + f) + 2 * p + c) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const uint4 af = st[(2 * (8 * wm + i) + c) * 64 + lane];
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int p = 0; p < 2; ++p)
              acc[i][f][p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const i32x4*>(&af),
                                                                   *reinterpret_cast<const i32x4*>(&bf[f][p]),
                                                                   acc[i][f][p], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      VR_STAMP();
    }
    // the stage buffer of the item's LAST K-step is free now (every wave is past the barrier); the other one may already
    // hold the next item's first K-step. The fold of the tile bounds below uses the free one.
    uint4* fold = lds + ((step - 1) & 1) * (kBStage / 16);

    // epilogue. C/D map: query = lane & 15 of the fragment, rows 4 (lane >> 4) + r of the tile.
    // The per-row words of all eight tiles are requested together (clamped addresses, so nothing depends on the
    // tail test): as a loop with a break in it they were eight round trips to memory, one after the other.
    uchar4 m8[8];
    float4 e8[8], s8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t tile = std::min<int64_t>(tile0 + 8 * wm + i, last_tile);
      const int64_t row0 = tile * kTileRows + 4 * g;
      m8[i] = *reinterpret_cast<const uchar4*>(mask + row0);
      e8[i] = *reinterpret_cast<const float4*>(row_err + row0);
      s8[i] = *reinterpret_cast<const float4*>(row_scale + row0);
    }
    float run[2] = {-__builtin_inff(), -__builtin_inff()};
    if (stamping && (m8[7].x | 1)) VR_STAMP();  // (after the per-row words have arrived)
    if (PASS == 1) {
      // two elements per instruction on the packed-f32 VALU: the lane's two query fragments side by side. A masked row (or
      // one behind the corpus's end) gets a bias of -inf into its score, so both of its bounds are -inf and no select is needed.
      using f32x2 = __attribute__((ext_vector_type(2))) float;
      const f32x2 PA = {pa[0], pa[1]}, PB = {pb[0], pb[1]}, P1 = {p1[0], p1[1]}, P2 = {p2[0], p2[1]};
      f32x2 run2 = {-__builtin_inff(), -__builtin_inff()};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool real = static_cast<int>(tile0) + 8 * wm + i <= static_cast<int>(last_tile);  // wave-uniform (tiles < 2^28)
        const unsigned char mm[4] = {m8[i].x, m8[i].y, m8[i].z, m8[i].w};
        const float ee[4] = {e8[i].x, e8[i].y, e8[i].z, e8[i].w};
        const float ss[4] = {s8[i].x, s8[i].y, s8[i].z, s8[i].w};
        f32x2 top2 = {-__builtin_inff(), -__builtin_inff()};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float bias = (real && mm[r]) ? 0.0f : -__builtin_inff();
          const f32x2 da = {static_cast<float>(acc[i][0][0][r]), static_cast<float>(acc[i][1][0][r])};
          const f32x2 db = {static_cast<float>(acc[i][0][1][r]), static_cast<float>(acc[i][1][1][r])};
          const f32x2 t = __builtin_elementwise_fma(PB, db, PA * da);
          const f32x2 score = __builtin_elementwise_fma(f32x2{ss[r], ss[r]}, t, f32x2{bias, bias});
          const f32x2 err = __builtin_elementwise_fma(f32x2{ee[r], ee[r]}, P1, P2);
          const f32x2 lo = score - err, up = score + err;
          run2 = __builtin_elementwise_max(run2, lo);
          top2 = __builtin_elementwise_max(top2, up);
        }
        // the tile's 16 rows are spread over the four lane groups: each leaves its four-row maximum in LDS as
        // [group][16 tiles][128 queries] f16 rounded up; they are folded on the way out (cross-lane maxima here — two
        // swizzles per tile and fragment — cost pass 1 a sixth of its time, and so did writing the bounds as 2-byte
        // stores from 16 lanes)
        if (tile_ub) {
          __half* dst = reinterpret_cast<__half*>(fold) + ((g * 16 + 8 * wm + i) * kBQ) + (2 * wn) * 16 + (lane & 15);
          dst[0] = __float2half_ru(top2.x);
          dst[16] = __float2half_ru(top2.y);
        }
      }
      run[0] = run2.x;
      run[1] = run2.y;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t tile = tile0 + 8 * wm + i;
        const bool real = tile <= last_tile;  // wave-uniform
        const int64_t row0 = tile * kTileRows + 4 * g;
        const unsigned char mm[4] = {m8[i].x, m8[i].y, m8[i].z, m8[i].w};
        const float ee[4] = {e8[i].x, e8[i].y, e8[i].z, e8[i].w};
        const float ss[4] = {s8[i].x, s8[i].y, s8[i].z, s8[i].w};
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float score, err;
            batch_bound(pa[f], pb[f], p1[f], p2[f], ss[r], ee[r], acc[i][f][0][r], acc[i][f][1][r], score, err);
            if (real && mm[r] && score + err >= pthr[f] && qidx[f] < nq) {
              const int slot = atomicAdd(cand_cnt + qidx[f], 1);
              if (slot < kBatchCand) cand[static_cast<int64_t>(qidx[f]) * kBatchCand + slot] = static_cast<int32_t>(row0 + r);
            }
          }
      }
    }
    VR_STAMP();
    if (PASS == 1 && tile_ub) {
      __syncthreads();
      const int row = threadIdx.x >> 5, part = threadIdx.x & 31;  // 16 tiles x 32 pieces of four queries (8 bytes)
      using h4 = __attribute__((ext_vector_type(4))) _Float16;
      h4 best4 = reinterpret_cast<const h4*>(fold)[row * 32 + part];
#pragma unroll
      for (int gg = 1; gg < 4; ++gg) {
        const h4 o = reinterpret_cast<const h4*>(fold)[(gg * 16 + row) * 32 + part];
#pragma unroll
        for (int j = 0; j < 4; ++j) best4[j] = o[j] > best4[j] ? o[j] : best4[j];
      }
      *reinterpret_cast<h4*>(tile_ub + (tile0 + row) * nq_pad + qc * kBQ + part * 4) = best4;
    }
    if (PASS == 1) {
      // this wave's 128 rows -> one value per query (the four lane groups hold different rows of the same queries)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        float v = run[f];
        v = fmaxf(v, __shfl_xor(v, 16));
        v = fmaxf(v, __shfl_xor(v, 32));
        if (g == 0 && qidx[f] < nq) best[static_cast<int64_t>(qidx[f]) * (2 * n_rb) + 2 * rbi + wm] = v;
      }
    }
    VR_STAMP();
    stamping = false;  // (the first item only)
    if (next_id >= n_ids) break;  // block-uniform
    // the next item's first K-step stages its second one into the buffer the fold above was read from
    if (PASS == 1 && tile_ub) __syncthreads();
    id = next_id;
    rbi = nrbi;
    qc = nqc;
  }
#undef VR_STAMP
}

// ---- instead of a second pass: revisit the few (tile, query) pairs that can hold a candidate ------------------
//
// Pass 1 leaves, beside the per-slab lower bounds, the largest UPPER bound of every (16-row tile, query) as an f16
// rounded up (2 B x N/16 x Q: 131 MB for 1k queries over a million rows). A row can only be a candidate of query q
// if its tile's bound reaches T_q, and there are about as many such pairs as candidates (tens per query), so the
// second integer GEMM over the whole corpus is replaced by
//   batch_flag_kernel   one compare per (tile, query) -> a list of pairs;
//   batch_pairs_kernel  a wave per pair: the 16 rows' exact int32 dot products with the query's two int8 parts on
//                       v_dot4_i32_i8 (lane = (row, 16-byte k segment), the shadow tile and the query image are read
//                       in the layout the MFMAs read them), then batch_bound() — the same bits as pass 1 — and the
//                       rows whose upper bound reaches T_q go to the query's candidate list, as pass 2 put them.

__global__ __launch_bounds__(256) void batch_flag_kernel(const __half* __restrict__ tile_ub, const float* __restrict__ thr,
                                                         int64_t n_cells, int nq, int nq_pad, int32_t* __restrict__ pairs,
                                                         int32_t* __restrict__ pair_cnt) {
  // eight cells (one tile, eight consecutive queries) per thread; hits are rare (about as many as candidates).
  // The lists are per query: one list for all put every hit through ONE atomic counter (0.8 ms for 68k hits).
  const int64_t cell0 = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 8;
  if (cell0 >= n_cells) return;
  const int q0 = static_cast<int>(cell0 % nq_pad);  // nq_pad % 128 == 0: the eight cells share the tile
  const int64_t tile = cell0 / nq_pad;
  const uint4 raw = *reinterpret_cast<const uint4*>(tile_ub + cell0);
  const float4 t0 = *reinterpret_cast<const float4*>(thr + q0), t1 = *reinterpret_cast<const float4*>(thr + q0 + 4);  // thr is padded
  const float tq[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float ub = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>((w[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu)));
    const int q = q0 + j;
    if (q < nq && ub >= tq[j]) {
      const int slot = atomicAdd(pair_cnt + q, 1);
      // (more tiles than the budget of rows: batch_final_kernel reports the query as overflowed and it is redone alone)
      if (slot < kBatchCand) pairs[static_cast<int64_t>(q) * kBatchCand + slot] = static_cast<int32_t>(tile);
    }
  }
}

__global__ __launch_bounds__(256) void batch_pairs_kernel(
    const uint4* __restrict__ corpus8, const uint4* __restrict__ img_a, const uint4* __restrict__ img_b,
    const float* __restrict__ params, const float* __restrict__ row_err, const float* __restrict__ row_scale,
    const uint8_t* __restrict__ mask, int64_t n_rows, int kb8n, const float* __restrict__ thr, int nq,
    const int32_t* __restrict__ pairs, const int32_t* __restrict__ pair_cnt, int32_t* __restrict__ cand,
    int32_t* __restrict__ cand_cnt) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, ks = lane >> 4;  // the shadow's layout: lane = (k % 64) / 16 * 16 + row % 16, 16 bytes each
  const int64_t waves = static_cast<int64_t>(gridDim.x) * 4;
  const int wpq = static_cast<int>(waves / nq > 0 ? waves / nq : 1);  // waves that share a query's list
  for (int64_t job = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6); job < static_cast<int64_t>(nq) * wpq; job += waves) {
    const int q = static_cast<int>(job / wpq);
    const int n = min(pair_cnt[q], kBatchCand);
    float pa, pb, p1, p2;
    batch_query_consts(params + static_cast<int64_t>(q) * kQParams, pa, pb, p1, p2);
    const float tq = thr[q];
    // the query images: [qfrag][kb8][lane = (k % 64) / 16 * 16 + query % 16][16 bytes]
    const uint4* qa = img_a + (static_cast<int64_t>(q >> 4) * kb8n) * 64 + ks * 16 + (q & 15);
    const uint4* qb = img_b + (static_cast<int64_t>(q >> 4) * kb8n) * 64 + ks * 16 + (q & 15);
    for (int sl = static_cast<int>(job % wpq); sl < n; sl += wpq) {
      const int64_t tile = pairs[static_cast<int64_t>(q) * kBatchCand + sl];
      const uint4* x = corpus8 + tile * kb8n * 64 + lane;
      int da = 0, db = 0;
      for (int kb = 0; kb < kb8n; ++kb) {
        const uint4 xv = x[kb * 64], av = qa[kb * 64], bv = qb[kb * 64];
        da = __builtin_amdgcn_sdot4(static_cast<int>(xv.x), static_cast<int>(av.x), da, false);
        da = __builtin_amdgcn_sdot4(static_cast<int>(xv.y), static_cast<int>(av.y), da, false);
        da = __builtin_amdgcn_sdot4(static_cast<int>(xv.z), static_cast<int>(av.z), da, false);
        da = __builtin_amdgcn_sdot4(static_cast<int>(xv.w), static_cast<int>(av.w), da, false);
        db = __builtin_amdgcn_sdot4(static_cast<int>(xv.x), static_cast<int>(bv.x), db, false);
        db = __builtin_amdgcn_sdot4(static_cast<int>(xv.y), static_cast<int>(bv.y), db, false);
        db = __builtin_amdgcn_sdot4(static_cast<int>(xv.z), static_cast<int>(bv.z), db, false);
        db = __builtin_amdgcn_sdot4(static_cast<int>(xv.w), static_cast<int>(bv.w), db, false);
      }
      da += __shfl_xor(da, 16);
      da += __shfl_xor(da, 32);
      db += __shfl_xor(db, 16);
      db += __shfl_xor(db, 32);
      const int64_t row = tile * kTileRows + r;
      if (ks == 0 && row < n_rows && mask[row]) {
        float score, err;
        batch_bound(pa, pb, p1, p2, row_scale[row], row_err[row], da, db, score, err);
        if (score + err >= tq) {
          const int slot = atomicAdd(cand_cnt + q, 1);
          if (slot < kBatchCand) cand[static_cast<int64_t>(q) * kBatchCand + slot] = static_cast<int32_t>(row);
        }
      }
    }
  }
}

// thr[q] = score of the k-th best key of query q (keys: [nq][k], descending, zero padded); fewer than k rows
// in play: every row with a finite bound is a candidate
__global__ void batch_threshold_kernel(const uint64_t* __restrict__ keys, int nq, int k, float* __restrict__ thr) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const uint64_t kth = keys[static_cast<int64_t>(q) * k + (k - 1)];
  float t = -3.0e38f;
  if (kth) {
    const uint32_t hi = static_cast<uint32_t>(kth >> 32);
    t = __uint_as_float((hi & 0x80000000u) ? (hi ^ 0x80000000u) : ~hi);
  }
  thr[q] = t;
}

// ---- exact re-score of the (query, candidate row) pairs --------------------------------------------------

// A wave per (query, 16 of its candidate rows): the rows — from sixteen different tiles — are gathered straight into the
// A operand of dense.hip's v_mfma_f32_16x16x4_f32 chain (lane g * 16 + r fetches the 16 bytes that hold elements
// 16 kb + 4 c + g, c = 0..3, of candidate r: the tiled corpus stores exactly that float4), the query's image (written
// by batch_prep_kernel in B-operand order, the same value in all sixteen columns) is the B operand, and the chain runs in
// k order from +0.0 — bit for bit the one-stage scan's score (and the fmaf chain this kernel replaced, which walked a row
// per LANE: 16 useful bytes of every 128 fetched, 768 dependent steps). keys[q][slot] (0 beyond the query's count).
__global__ __launch_bounds__(256) void batch_rescore_kernel(const float4* __restrict__ corpus,
                                                            const float4* __restrict__ qimg,
                                                            const int32_t* __restrict__ cand,
                                                            const int32_t* __restrict__ cand_cnt, int nq, int dim,
                                                            int kblocks, uint64_t* __restrict__ keys) {
  const int lane = threadIdx.x & 63;
  const int64_t group = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);  // (query, 16 slots)
  constexpr int kGroups = kBatchCand / 16;
  const int q = static_cast<int>(group / kGroups), slot0 = static_cast<int>(group % kGroups) * 16;
  if (q >= nq) return;  // wave-uniform
  uint64_t* out = keys + static_cast<int64_t>(q) * kBatchCand + slot0;
  const int cnt = min(cand_cnt[q], kBatchCand);
  if (slot0 >= cnt) {  // wave-uniform: nothing in this group
    if (lane < 16) out[lane] = 0ull;
    return;
  }
  const int r = lane & 15, g = lane >> 4;
  const bool valid = slot0 + r < cnt;
  const int64_t row = cand[static_cast<int64_t>(q) * kBatchCand + (valid ? slot0 + r : slot0)];  // (a real row either way)
  const float4* xp = corpus + (row / kTileRows) * kblocks * 64 + tile_pos(g, static_cast<int>(row % kTileRows));
  const float4* qp = qimg + static_cast<int64_t>(q) * kblocks * 4 + g;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  int kb = 0;
  for (; kb + 8 <= kblocks; kb += 8) {  // eight gathers in flight per lane
    float4 a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = xp[(kb + u) * 64];
      b[u] = qp[(kb + u) * 4];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc, 0, 0, 0);
    }
  }
  for (; kb < kblocks; ++kb) {
    const float4 a = xp[kb * 64], b = qp[kb * 4];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  }
  // C/D map: lane holds rows 4 g + reg of column (lane & 15); every column carries the same query
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int i = 4 * g + reg;
    const int64_t row_i = __shfl(static_cast<long long>(row), i);  // lane i gathered candidate i
    const bool valid_i = slot0 + i < cnt;
    if (r == 0) out[i] = valid_i ? topk_make_key(acc[reg], row_i) : 0ull;
  }
}

// one block per query: the k best of its <= kBatchCand exact keys (k rounds of block-wide extract-max),
// and the overflow flag for the host
__global__ __launch_bounds__(256) void batch_final_kernel(const uint64_t* __restrict__ keys,
                                                          const int32_t* __restrict__ cand_cnt,
                                                          const int32_t* __restrict__ pair_cnt, int k,
                                                          uint64_t* __restrict__ out, int32_t* __restrict__ overflow) {
  __shared__ uint64_t wmax[2][4];
  const int q = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t v[kBatchCand / 256];
  uint64_t lmax = 0;
#pragma unroll
  for (int i = 0; i < kBatchCand / 256; ++i) {
    v[i] = keys[static_cast<int64_t>(q) * kBatchCand + i * 256 + threadIdx.x];
    lmax = v[i] > lmax ? v[i] : lmax;
  }
  // (the host compares it with the budget; a query with more flagged tiles than the budget holds is over it as well)
  if (threadIdx.x == 0) overflow[q] = (pair_cnt && pair_cnt[q] > kBatchCand) ? kBatchCand + 1 : cand_cnt[q];
  for (int r = 0; r < k; ++r) {
    uint64_t m = lmax;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const uint64_t o = __shfl_xor(m, off);
      m = o > m ? o : m;
    }
    if (lane == 0) wmax[r & 1][wave] = m;
    __syncthreads();
    uint64_t bestk = wmax[r & 1][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) bestk = wmax[r & 1][w] > bestk ? wmax[r & 1][w] : bestk;
    if (threadIdx.x == 0) out[static_cast<int64_t>(q) * k + r] = bestk;
    if (bestk != 0 && lmax == bestk) {  // keys are unique: exactly one owner
      lmax = 0;
#pragma unroll
      for (int i = 0; i < kBatchCand / 256; ++i) {
        if (v[i] == bestk) v[i] = 0;
        lmax = v[i] > lmax ? v[i] : lmax;
      }
    }
  }
}

bool batch_usable(vr_engine* e, int nq, int k) {
  static const bool off = getenv("VR_BATCH_SEARCH") && atoi(getenv("VR_BATCH_SEARCH")) == 0;
  return !off && e->prefilter8 && e->dim % 128 == 0 && nq > kQueryBlock && k <= kFusedMaxK && e->n_rows >= 16384;
}

// q_dev: nq x D raw queries on the device. Leaves nq x k keys in out_keys_dev and one overflow flag per query
// in overflow_dev (device memory owned by the engine; the caller copies both back).
int batch_search(vr_engine* e, const float* q_dev, int nq, int k, const uint8_t* mask_dev, const uint64_t** out_keys_dev,
                 const int32_t** overflow_dev) {
  hipStream_t s = e->stream;
  const int dim = e->dim, kb8n = dim / 64;
  const int64_t n_tiles = (e->n_rows + kTileRows - 1) / kTileRows;
  const int n_rb = static_cast<int>((n_tiles + 15) / 16);
  const int n_qc = (nq + kBQ - 1) / kBQ;
  const int64_t nq_pad = static_cast<int64_t>(n_qc) * kBQ;
  VR_TRY(e->bq_hat.grow(static_cast<int64_t>(nq) * dim, 0, s));
  VR_TRY(e->bq_img.grow(2 * nq_pad * dim / 4, 0, s));  // two int8 images, counted in int32
  VR_TRY(e->bq_params.grow(static_cast<int64_t>(nq) * kQParams, 0, s));
  VR_TRY(e->bq_best.grow(static_cast<int64_t>(nq) * 2 * n_rb, 0, s));
  VR_TRY(e->bq_thr.grow(nq_pad, 0, s));  // (padded: batch_flag_kernel reads eight thresholds at a time)
  VR_TRY(e->bq_cand.grow(static_cast<int64_t>(nq) * kBatchCand, 0, s));
  VR_TRY(e->bq_cnt.grow(2 * static_cast<int64_t>(nq), 0, s));  // counts, then overflow flags
  VR_TRY(e->bq_keys.grow(static_cast<int64_t>(nq) * kBatchCand + static_cast<int64_t>(nq) * k, 0, s));
  int8_t* img_a = reinterpret_cast<int8_t*>(e->bq_img.p);
  int8_t* img_b = img_a + nq_pad * dim;
  VR_HIP(hipMemsetAsync(e->bq_img.p, 0, static_cast<size_t>(2 * nq_pad * dim), s));  // padding queries: zeros
  VR_HIP(hipMemsetAsync(e->bq_cnt.p, 0, sizeof(int32_t) * 2 * static_cast<size_t>(nq), s));
  hipLaunchKernelGGL(batch_prep_kernel, dim3(static_cast<unsigned>(nq)), dim3(256), static_cast<size_t>(dim) * sizeof(float),
                     s, q_dev, nq, dim, kb8n, e->bq_hat.p, img_a, img_b, e->bq_params.p,
                     e->centre_rows > 0 ? e->centre_norm : 0.0f);
  const unsigned grid = static_cast<unsigned>(((n_rb + 7) / 8) * n_qc * 8);
  // Pass 1 only has to produce SOME k rows' lower bounds per query, so it could run on every stride-th row block
  // (VR_BATCH_SAMPLE=stride; at least 8 k slabs, and 64, stay in the sample). Measured and NOT the default: the
  // threshold of a sample is lower, and the candidate count is steep in it — bench corpus, 1000 queries: 69
  // candidates per query with the full pass, 239 at stride 4 (scan 5.3 -> 3.5 ms, call 6.9 -> 6.7 ms: the exact
  // re-score eats the gain), 800 at stride 16 with a fifth of the queries over budget; an anisotropic corpus
  // overflows at stride 4 already (profiles/r02_gemm_experiments.md §11).
  static const int want_stride = getenv("VR_BATCH_SAMPLE") ? std::max(1, atoi(getenv("VR_BATCH_SAMPLE"))) : 1;
  const int stride = std::max(1, std::min(want_stride, 2 * n_rb / std::max(8 * k, 64)));
  const int n_rb1 = (n_rb + stride - 1) / stride;
  const unsigned grid1 = static_cast<unsigned>(((n_rb1 + 7) / 8) * n_qc * 8);
  // algorithmic work of the batched scan: 2 N D Q operations (the second query part is overhead); timed: the GEMM
  // pass, the threshold selection, and the flag + pairs kernels (or the second pass)
  prof_begin(e, VR_PROF_BATCH_SCAN, 2.0 * static_cast<double>(e->n_rows) * dim * nq);
  // VR_BATCH_TWO_PASS=1 (and a sampled pass 1) keeps the second integer GEMM; the default revisits flagged pairs
  static const bool two_pass_env = getenv("VR_BATCH_TWO_PASS") && atoi(getenv("VR_BATCH_TWO_PASS")) != 0;
  const bool two_pass = two_pass_env || stride != 1;
  const int64_t n_cells = static_cast<int64_t>(n_rb) * 16 * nq_pad;  // (tile, query) cells, padded to whole blocks
  const int64_t pair_cap = static_cast<int64_t>(nq) * kBatchCand;  // tiles listed per query, then the counts
  __half* tile_ub = nullptr;
  if (!two_pass) {
    VR_TRY(e->bq_tile_ub.grow(n_cells, 0, s));
    VR_TRY(e->bq_pairs.grow(pair_cap + nq, 0, s));
    VR_HIP(hipMemsetAsync(e->bq_pairs.p + pair_cap, 0, sizeof(int32_t) * static_cast<size_t>(nq), s));
    tile_ub = reinterpret_cast<__half*>(e->bq_tile_ub.p);
  }
  // VR_BATCH_STAMPS=1 (diagnostics): phase time stamps of a sample of blocks, printed after the call
  static const bool stamps_on = getenv("VR_BATCH_STAMPS") && atoi(getenv("VR_BATCH_STAMPS")) != 0;
  unsigned long long* stamps = nullptr;
  const size_t n_stamp_blocks = grid1 / 61 + 1;  // (room for the one-item-per-block grid)
  if (stamps_on) {
    VR_HIP(hipMalloc(reinterpret_cast<void**>(&stamps), n_stamp_blocks * 16 * sizeof(unsigned long long)));
    VR_HIP(hipMemsetAsync(stamps, 0, n_stamp_blocks * 16 * sizeof(unsigned long long), s));
  }
  // persistent blocks: one per CU (128 KiB of LDS each), ids dealt round-robin — a multiple of 8 blocks keeps an id's XCD
  static const int n_cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return std::max(8, n / 8 * 8);
  }();
  static const bool persistent = !(getenv("VR_BATCH_PERSISTENT") && atoi(getenv("VR_BATCH_PERSISTENT")) == 0);
  const unsigned blocks1 = persistent ? std::min<unsigned>(grid1, static_cast<unsigned>(n_cus)) : grid1;
  hipLaunchKernelGGL((batch_scan_kernel<1>), dim3(blocks1), dim3(512), 0, s, reinterpret_cast<const uint4*>(e->corpus16.p),
                     reinterpret_cast<const uint4*>(img_a), reinterpret_cast<const uint4*>(img_b), e->bq_params.p,
                     e->row_err.p, e->row_scale.p, mask_dev, n_tiles, n_rb1, stride, n_qc, nq, kb8n, e->bq_best.p,
                     static_cast<const float*>(nullptr), static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr),
                     tile_ub, stamps, static_cast<int>(grid1));
  if (stamps_on) {
    std::vector<unsigned long long> h(n_stamp_blocks * 16);
    VR_HIP(hipMemcpyAsync(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    VR_HIP(hipStreamSynchronize(s));
    (void)hipFree(stamps);
    double sum[16] = {0};
    int n = 0;
    const int nk = kb8n / 2;  // K-steps of the kernel (<= 8: stamps fit in 16 slots)
    for (size_t b = 0; b < n_stamp_blocks; ++b) {
      const unsigned long long* t = h.data() + b * 16;
      if (!t[0] || !t[nk + 4]) continue;
      for (int i = 1; i <= nk + 4; ++i) sum[i] += static_cast<double>(t[i] - t[i - 1]);
      ++n;
    }
    fprintf(stderr, "[batch_scan stamps] %d blocks, cycles per phase: prologue %.0f |", n, n ? sum[1] / n : 0.0);
    for (int i = 2; i <= nk + 1; ++i) fprintf(stderr, " k%d %.0f", i - 2, n ? sum[i] / n : 0.0);
    fprintf(stderr, " | row words %.0f | bounds %.0f | fold+store %.0f\n", n ? sum[nk + 2] / n : 0.0, n ? sum[nk + 3] / n : 0.0,
            n ? sum[nk + 4] / n : 0.0);
  }
  const uint64_t* kth = nullptr;
  VR_TRY(topk_select(e, e->bq_best.p, 2 * n_rb1, 2 * n_rb1, nq, k, &kth));
  hipLaunchKernelGGL(batch_threshold_kernel, dim3(static_cast<unsigned>((nq + 255) / 256)), dim3(256), 0, s, kth, nq, k,
                     e->bq_thr.p);
  if (two_pass) {
    hipLaunchKernelGGL((batch_scan_kernel<2>), dim3(grid), dim3(512), 0, s, reinterpret_cast<const uint4*>(e->corpus16.p),
                       reinterpret_cast<const uint4*>(img_a), reinterpret_cast<const uint4*>(img_b), e->bq_params.p,
                       e->row_err.p, e->row_scale.p, mask_dev, n_tiles, n_rb, 1, n_qc, nq, kb8n, static_cast<float*>(nullptr),
                       e->bq_thr.p, e->bq_cand.p, e->bq_cnt.p, static_cast<__half*>(nullptr),
                       static_cast<unsigned long long*>(nullptr), static_cast<int>(grid));
  } else {
    hipLaunchKernelGGL(batch_flag_kernel, dim3(static_cast<unsigned>((n_cells / 8 + 255) / 256)), dim3(256), 0, s, tile_ub,
                       e->bq_thr.p, n_cells, nq, static_cast<int>(nq_pad), e->bq_pairs.p, e->bq_pairs.p + pair_cap);
    hipLaunchKernelGGL(batch_pairs_kernel, dim3(2048), dim3(256), 0, s, reinterpret_cast<const uint4*>(e->corpus16.p),
                       reinterpret_cast<const uint4*>(img_a), reinterpret_cast<const uint4*>(img_b), e->bq_params.p,
                       e->row_err.p, e->row_scale.p, mask_dev, e->n_rows, kb8n, e->bq_thr.p, nq, e->bq_pairs.p,
                       e->bq_pairs.p + pair_cap, e->bq_cand.p, e->bq_cnt.p);
  }
  prof_end(e);
  uint64_t* keys = e->bq_keys.p;
  uint64_t* out = keys + static_cast<int64_t>(nq) * kBatchCand;
  hipLaunchKernelGGL(batch_rescore_kernel, dim3(static_cast<unsigned>(static_cast<int64_t>(nq) * (kBatchCand / 16) / 4)), dim3(256),
                     0, s, reinterpret_cast<const float4*>(e->corpus.p), reinterpret_cast<const float4*>(e->bq_hat.p), e->bq_cand.p,
                     e->bq_cnt.p, nq, dim,
                     e->kblocks, keys);
  hipLaunchKernelGGL(batch_final_kernel, dim3(static_cast<unsigned>(nq)), dim3(256), 0, s, keys, e->bq_cnt.p,
                     two_pass ? static_cast<const int32_t*>(nullptr) : e->bq_pairs.p + pair_cap, k, out, e->bq_cnt.p + nq);
  VR_HIP(hipGetLastError());
  *out_keys_dev = out;
  *overflow_dev = e->bq_cnt.p + nq;
  return 0;
}

}  // namespace vr
