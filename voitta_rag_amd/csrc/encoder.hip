// Dense encode on the GPU: the BERT-family forward that SentenceTransformer.encode runs for the
// reference (call sites: src/voitta/services/embedding.py:40,53,68-73,85; SURVEY.md a4 [EXT]).
//
// Layout: sequences are PACKED — activations are [T, H] over the T real tokens of a batch, with
// an offsets array (cu) marking sequence boundaries. No padding rows exist, so no FLOP or byte is
// spent on them (sentence-transformers pads to the longest sequence of each 32-batch; padding
// never changes a result because masked logits underflow to exactly 0).
//
// Kernels (all f32; the matrix products run on the f32-input MFMA, which is an exact f32 fma
// chain, so the only differences from the torch-CPU reference are summation orders):
//   embed_ln_kernel   word+position+type gather, LayerNorm           HBM bound, 2*H*4 B/token
//   gemm_f32_kernel   C = A W^T + b (+GELU | +residual)              MFMA bound (v_mfma_f32_32x32x2_f32)
//                     128x128x32 block tile, 4 waves of 64x64, register-staged double-buffered LDS
//   attention_kernel  softmax(Q K^T / sqrt(d)) V per (sequence, head) MFMA (v_mfma_f32_16x16x4_f32),
//                     online softmax, keys streamed through LDS 64 at a time
//   layernorm_kernel  LayerNorm over H                               HBM bound
//   pool_kernel       mean / CLS pooling + L2 normalise              HBM bound
// Algorithmic FLOP per token = L * (24 H^2 + 4 S H), of which the GEMMs are
// 24 H^2 L: the dominant kernel of the indexing path.

#include "engine_internal.h"

#include <map>
#include <tuple>

#include <algorithm>
#include <cmath>

namespace vr {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

struct SplitWeight {  // f16x3 mode: w * 2^s as (hi, lo) f16, [out][in]; unscale = 2^-s
  _Float16* hi = nullptr;
  _Float16* lo = nullptr;
  float unscale = 1.0f;
};

struct LayerWeights {
  float *wqkv, *bqkv, *wo, *bo, *ln1g, *ln1b, *w1, *b1, *w2, *b2, *ln2g, *ln2b;
  SplitWeight s_qkv, s_o, s_1, s_2;
  // f16 mode, LayerNorm folded into the consuming GEMM (see EPI_FOLD_*): weights scaled by the LayerNorm
  // gain in front of them, their column sums and the bias with the LayerNorm shift pushed through
  SplitWeight s_qkv_f{}, s_1_f{};  // s_qkv_f: layers >= 1 only (layer 0 reads the embedding LayerNorm's output)
  float *cs_qkv = nullptr, *c_qkv = nullptr, *cs_1 = nullptr, *c_1 = nullptr;
};

struct Encoder {
  vr_bert_desc d{};
  float *word = nullptr, *pos = nullptr, *type = nullptr, *lng = nullptr, *lnb = nullptr;
  std::vector<LayerWeights> layers;
  std::vector<float*> owned;  // every device allocation, for release
  // workspace for up to ws_tokens packed tokens
  int64_t ws_tokens = 0;
  float *x = nullptr, *qkv = nullptr, *ctx = nullptr, *tmp = nullptr, *ffn = nullptr;
  float* xs = nullptr;  // f16x3 / f16 mode: the hidden state as GEMM input ((hi, lo) or plain f16 rows)
  float* lnstat = nullptr;  // f16 mode: two arrays of per-row (mean, 1/sigma), see forward_chunk
  float* lnpart = nullptr;  // f16 mode: per (row, 64 columns) partial (sum, sum of squares) of the folded LayerNorms
  DevArray<int32_t> ids, cu;
  DevArray<float> out;
  DevArray<float> skinny_ws;  // K-slice partial sums of gemm_f16_skinny_kernel
  // hipGraphs of small forward passes (a query, a handful of sequences): ~135 launches of a few
  // microseconds of work each are launch-bound; replayed as one graph they are not
  struct GraphKey {
    int T, n_seq, max_len;
    const void *ids, *cu, *out;
    bool operator<(const GraphKey& o) const {
      return std::tie(T, n_seq, max_len, ids, cu, out) < std::tie(o.T, o.n_seq, o.max_len, o.ids, o.cu, o.out);
    }
  };
  struct GraphEntry {
    int seen = 0;
    bool bad = false;
    hipGraphExec_t exec = nullptr;
  };
  std::map<GraphKey, GraphEntry> graphs;
};

// captured graphs hold raw workspace pointers: every reallocation of a buffer they touch drops them
static void invalidate_graphs(Encoder* enc) {
  for (auto& kv : enc->graphs)
    if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  enc->graphs.clear();
}

// ---- elementwise / normalisation ---------------------------------------------------------------

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

constexpr int kMaxPairs = 8;  // H <= 1024: H/128 float2 per lane

// ---- split precision (f16x3) ---------------------------------------------------------------------
// A GEMM operand v is carried as two f16 numbers, hi = f16(v) and lo = f16(v - hi): together 22
// significant bits. The product a*w is then a_hi*w_hi + a_hi*w_lo + a_lo*w_hi (the dropped
// a_lo*w_lo term is < 2^-22 relative), three passes of the f16 MFMA whose products are exact in
// its f32 accumulator — f32-class accuracy at 16/3 of the f32-MFMA rate. Activations are O(1) so
// they are split unscaled (absolute granularity 2^-24, f16's subnormal spacing); values beyond
// f16's range are clamped. Weights are pre-scaled by a power of two per tensor (exact).
using half_t = _Float16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

__device__ __forceinline__ void split_f16(float v, half_t& hi, half_t& lo) {
  v = fminf(fmaxf(v, -65504.0f), 65504.0f);
  hi = static_cast<half_t>(v);
  lo = static_cast<half_t>(v - static_cast<float>(hi));
}

struct half2_t {
  half_t x, y;
};

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)) for the f16x3 epilogues, where erff() made the epilogue VALU-bound
// (a third of the FFN-up tile time). Abramowitz-Stegun 7.1.26: erfc(z) = t (a1 + t (a2 + ... a5 t)) e^{-z^2},
// t = 1 / (1 + p z), |error| <= 1.5e-7 — the size of an f32 ulp of erf, and far inside what the (hi, lo)
// split of the result keeps. Negative x uses 0.5 x erfc(|x| / sqrt 2) directly, so the tail has no
// cancellation. 13 instructions (one v_rcp_f32, one v_exp_f32) against ~40 for erff.
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float q = p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);  // erfc(z)
  return x >= 0.0f ? x * fmaf(-0.5f, q, 1.0f) : 0.5f * x * q;
}

// two elements at a time on the packed-f32 VALU (v_pk_fma_f32 / v_pk_mul_f32): same operations in the
// same order as gelu_fast, so the same bits
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
  const f32x2 z = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
  const f32x2 d = __builtin_elementwise_fma(f32x2{0.3275911f, 0.3275911f}, z, f32x2{1.0f, 1.0f});
  const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  f32x2 p = __builtin_elementwise_fma(f32x2{1.061405429f, 1.061405429f}, t, f32x2{-1.453152027f, -1.453152027f});
  p = __builtin_elementwise_fma(p, t, f32x2{1.421413741f, 1.421413741f});
  p = __builtin_elementwise_fma(p, t, f32x2{-0.284496736f, -0.284496736f});
  p = __builtin_elementwise_fma(p, t, f32x2{0.254829592f, 0.254829592f});
  const f32x2 a = (-1.4426950408889634f * z) * z;
  const f32x2 q = (p * t) * f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
  const f32x2 pos = x * __builtin_elementwise_fma(f32x2{-0.5f, -0.5f}, q, f32x2{1.0f, 1.0f});
  const f32x2 neg = (0.5f * x) * q;
  return f32x2{x.x >= 0.0f ? pos.x : neg.x, x.y >= 0.0f ? pos.y : neg.y};
}

// GELU of the single-pass f16 mode: its output is rounded to f16 (relative 4.9e-4), so the two
// quarter-rate transcendentals of gelu_fast (rcp, exp2: half of its issue cycles, and the GELU epilogue
// was 15 % of the FFN-up GEMM) buy nothing. erf(x / sqrt 2) = w Q(w^2) with w = clamp(x, +-3 sqrt 2), Q a
// degree-9 polynomial (near-minimax fit, |erf error| <= 3.3e-6 inside the clamp, 2.2e-5 beyond it):
// relative error of the result <= 1.4e-5 for x > 0, absolute <= 0.5 |x| 2.2e-5 for x < 0.
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 x) {
  constexpr float kClamp = 4.242640495300293f;
  const f32x2 w = {__builtin_amdgcn_fmed3f(x.x, -kClamp, kClamp), __builtin_amdgcn_fmed3f(x.y, -kClamp, kClamp)};
  const f32x2 s = w * w;
  f32x2 q = __builtin_elementwise_fma(f32x2{-5.183208029e-12f, -5.183208029e-12f}, s, f32x2{5.512241219e-10f, 5.512241219e-10f});
  q = __builtin_elementwise_fma(q, s, f32x2{-2.635556129e-08f, -2.635556129e-08f});
  q = __builtin_elementwise_fma(q, s, f32x2{7.569865943e-07f, 7.569865943e-07f});
  q = __builtin_elementwise_fma(q, s, f32x2{-1.478758622e-05f, -1.478758622e-05f});
  q = __builtin_elementwise_fma(q, s, f32x2{2.114593954e-04f, 2.114593954e-04f});
  q = __builtin_elementwise_fma(q, s, f32x2{-2.317534527e-03f, -2.317534527e-03f});
  q = __builtin_elementwise_fma(q, s, f32x2{1.985279098e-02f, 1.985279098e-02f});
  q = __builtin_elementwise_fma(q, s, f32x2{-1.329084933e-01f, -1.329084933e-01f});
  q = __builtin_elementwise_fma(q, s, f32x2{7.978681326e-01f, 7.978681326e-01f});
  const f32x2 hx = 0.5f * x;
  return __builtin_elementwise_fma(hx, w * q, hx);
}

// Four pairs at once, step by step across the pairs: the same operations per element as gelu_poly2 (same bits), but a
// dependent v_pk_fma_f32 chain issues one instruction per ~2 issue slots (the compiler pads every step with s_nop 0 and keeps
// the source order of the four chains): written pair by pair the polynomial of a 16 x 8 epilogue piece was 4 x 11 dependent
// steps; interleaved, the four chains fill each other's slots.
__device__ __forceinline__ void gelu_poly2x4(f32x2 (&x)[4]) {
  constexpr float kClamp = 4.242640495300293f;
  constexpr float kC[10] = {-5.183208029e-12f, 5.512241219e-10f, -2.635556129e-08f, 7.569865943e-07f, -1.478758622e-05f,
                            2.114593954e-04f,  -2.317534527e-03f, 1.985279098e-02f,  -1.329084933e-01f, 7.978681326e-01f};
  f32x2 w[4], s[4], q[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = f32x2{__builtin_amdgcn_fmed3f(x[i].x, -kClamp, kClamp), __builtin_amdgcn_fmed3f(x[i].y, -kClamp, kClamp)};
#pragma unroll
  for (int i = 0; i < 4; ++i) s[i] = w[i] * w[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = __builtin_elementwise_fma(f32x2{kC[0], kC[0]}, s[i], f32x2{kC[1], kC[1]});
#pragma unroll
  for (int c = 2; c < 10; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = __builtin_elementwise_fma(q[i], s[i], f32x2{kC[c], kC[c]});
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 hx = 0.5f * x[i];
    x[i] = __builtin_elementwise_fma(hx, w[i] * q[i], hx);
  }
}

// Split (hi, lo) matrices live in ONE interleaved array: row r of a [rows][K] matrix is 2K halfs,
// element k's hi at  r*2K + (k/8)*16 + k%8  and its lo 8 halfs further. A 128-byte line then holds
// 32 consecutive k of BOTH halves of one row — exactly what a 32-deep GEMM K-tile needs of that row,
// so the GEMM's global->LDS loads consume every line they touch (with separate hi and lo arrays a
// 32-deep K-tile used 64 bytes of each line and the L1 fetched every line twice).
// Callers pass hi = base and lo = base + 8; split_at() turns a column into the offset inside a row.
__host__ __device__ __forceinline__ int split_at(int k) { return ((k >> 3) << 4) + (k & 7); }

// one element of a LayerNorm output, exactly as row_layernorm computes it (same operations, same order)
__device__ __forceinline__ float ln_apply(float v, float mean, float inv, float g, float b) {
  return (v - mean) * inv * g + b;
}

// LayerNorm of one row held as float2 pairs per lane (biased variance, eps inside the sqrt).
// out_hi/out_lo (optional): the same row split for the f16x3 GEMMs.
__device__ __forceinline__ void row_layernorm(float2 (&v)[kMaxPairs], int pairs, int H,
                                              const float* __restrict__ g,
                                              const float* __restrict__ b, float eps, int lane,
                                              float* __restrict__ out, half_t* __restrict__ out_hi = nullptr,
                                              half_t* __restrict__ out_lo = nullptr,
                                              float2* __restrict__ stat = nullptr) {
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < kMaxPairs; ++i)
    if (i < pairs) s += v[i].x + v[i].y;
  const float mean = wave_sum(s) / static_cast<float>(H);
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < kMaxPairs; ++i)
    if (i < pairs) {
      float dx = v[i].x - mean, dy = v[i].y - mean;
      q += dx * dx + dy * dy;
    }
  const float var = wave_sum(q) / static_cast<float>(H);
  const float inv = 1.0f / sqrtf(var + eps);
  // stat (optional): the row's (mean, 1/sigma) — whoever holds the pre-LN row can then re-derive the
  // f32 output with ln_apply() instead of reading it back (see forward_chunk, f16 mode)
  if (stat && lane == 0) *stat = make_float2(mean, inv);
#pragma unroll
  for (int i = 0; i < kMaxPairs; ++i)
    if (i < pairs) {
      int e = (i * 64 + lane) * 2;
      float2 gg = *reinterpret_cast<const float2*>(g + e);
      float2 bb = *reinterpret_cast<const float2*>(b + e);
      float2 o;
      o.x = (v[i].x - mean) * inv * gg.x + bb.x;
      o.y = (v[i].y - mean) * inv * gg.y + bb.y;
      if (out) *reinterpret_cast<float2*>(out + e) = o;
      if (out_hi) {
        half2_t h, l;
        split_f16(o.x, h.x, l.x);
        split_f16(o.y, h.y, l.y);
        if (out_lo) {  // interleaved (hi, lo) row
          *reinterpret_cast<half2_t*>(out_hi + split_at(e)) = h;
          *reinterpret_cast<half2_t*>(out_lo + split_at(e)) = l;
        } else {       // plain f16 row (VR_PRECISION_F16)
          *reinterpret_cast<half2_t*>(out_hi + e) = h;
        }
      }
    }
}

// one wave per token: x[t] = LayerNorm(word[id] + pos[p] + type[0])
__global__ __launch_bounds__(256) void embed_ln_kernel(const int32_t* __restrict__ ids,
                                                       const int32_t* __restrict__ cu, int n_seq,
                                                       int tok_base, int T, int H, int vocab,
                                                       const float* __restrict__ word,
                                                       const float* __restrict__ pos,
                                                       const float* __restrict__ type,
                                                       const float* __restrict__ g,
                                                       const float* __restrict__ b, float eps,
                                                       float* __restrict__ x, half_t* __restrict__ x_hi,
                                                       half_t* __restrict__ x_lo, float* __restrict__ pre,
                                                       float2* __restrict__ stat) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  // position = index inside the sequence: binary search of the offsets
  int lo = 0, hi = n_seq;
  const int tg = t + tok_base;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (cu[mid] <= tg) lo = mid; else hi = mid;
  }
  const int p = tg - cu[lo];
  int id = ids[tg];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const int pairs = H / 128;
  float2 v[kMaxPairs];
#pragma unroll
  for (int i = 0; i < kMaxPairs; ++i)
    if (i < pairs) {
      int e = (i * 64 + lane) * 2;
      float2 w = *reinterpret_cast<const float2*>(word + static_cast<int64_t>(id) * H + e);
      float2 pp = *reinterpret_cast<const float2*>(pos + static_cast<int64_t>(p) * H + e);
      float2 tt = *reinterpret_cast<const float2*>(type + e);
      v[i].x = w.x + pp.x + tt.x;
      v[i].y = w.y + pp.y + tt.y;
      if (pre) *reinterpret_cast<float2*>(pre + static_cast<int64_t>(t) * H + e) = v[i];  // the pre-LN row
    }
  const int64_t o = static_cast<int64_t>(t) * H;
  row_layernorm(v, pairs, H, g, b, eps, lane, x ? x + o : nullptr, x_hi ? x_hi + (x_lo ? 2 : 1) * o : nullptr,
                x_lo ? x_lo + 2 * o : nullptr, stat ? stat + t : nullptr);
}

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ in, int T, int H,
                                                        const float* __restrict__ g,
                                                        const float* __restrict__ b, float eps,
                                                        float* __restrict__ out, half_t* __restrict__ out_hi,
                                                        half_t* __restrict__ out_lo, float2* __restrict__ stat) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const int pairs = H / 128;
  float2 v[kMaxPairs];
#pragma unroll
  for (int i = 0; i < kMaxPairs; ++i)
    if (i < pairs) v[i] = *reinterpret_cast<const float2*>(in + static_cast<int64_t>(t) * H + (i * 64 + lane) * 2);
  const int64_t o = static_cast<int64_t>(t) * H;
  row_layernorm(v, pairs, H, g, b, eps, lane, out ? out + o : nullptr, out_hi ? out_hi + (out_lo ? 2 : 1) * o : nullptr,
                out_lo ? out_lo + 2 * o : nullptr, stat ? stat + t : nullptr);
}

// f16-mode LayerNorm for H % 256 == 0: the row as float4 per lane (16-byte loads, 8-byte f16 stores —
// the generic kernel's 8-/4-byte accesses reached 4.6 TB/s on this 6-bytes-per-element pass). Writes the
// f16 row, the row's (mean, 1/sigma), and the f32 row only when `out` is given.
__global__ __launch_bounds__(256) void layernorm_f16_kernel(const float* __restrict__ in, int T, int H,
                                                            const float* __restrict__ g, const float* __restrict__ b,
                                                            float eps, float* __restrict__ out,
                                                            half_t* __restrict__ out_h, float2* __restrict__ stat) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const int quads = H / 256;  // <= 4
  const int64_t o = static_cast<int64_t>(t) * H;
  float4 v[4];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < quads) {
      v[i] = *reinterpret_cast<const float4*>(in + o + (i * 64 + lane) * 4);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  const float mean = wave_sum(s) / static_cast<float>(H);
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < quads) {
      const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  const float var = wave_sum(q) / static_cast<float>(H);
  const float inv = 1.0f / sqrtf(var + eps);
  if (stat && lane == 0) stat[t] = make_float2(mean, inv);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < quads) {
      const int e = (i * 64 + lane) * 4;
      const float4 gg = *reinterpret_cast<const float4*>(g + e);
      const float4 bb = *reinterpret_cast<const float4*>(b + e);
      float4 r;
      r.x = ln_apply(v[i].x, mean, inv, gg.x, bb.x);
      r.y = ln_apply(v[i].y, mean, inv, gg.y, bb.y);
      r.z = ln_apply(v[i].z, mean, inv, gg.z, bb.z);
      r.w = ln_apply(v[i].w, mean, inv, gg.w, bb.w);
      if (out) *reinterpret_cast<float4*>(out + o + e) = r;
      half_t h[4], l[4];
      split_f16(r.x, h[0], l[0]);
      split_f16(r.y, h[1], l[1]);
      split_f16(r.z, h[2], l[2]);
      split_f16(r.w, h[3], l[3]);
      *reinterpret_cast<uint2*>(out_h + o + e) = *reinterpret_cast<const uint2*>(h);
    }
}

// one block per sequence: mean (sum / max(count, 1e-9)) or CLS pooling, then x / max(|x|, 1e-12)
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ x,
                                                   const int32_t* __restrict__ cu, int seq0,
                                                   int tok_base, int H, int pooling, int normalize,
                                                   int compact, float* __restrict__ out) {
  __shared__ float red[4];
  const int seq = seq0 + blockIdx.x;
  // compact: x holds one row per sequence (its [CLS] row), see the last layer of forward_chunk
  const int t0 = compact ? static_cast<int>(blockIdx.x) : cu[seq] - tok_base;
  const int len = cu[seq + 1] - cu[seq];
  float vals[4];
  float ss = 0.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int c = threadIdx.x + 256 * j;
    float v = 0.0f;
    if (c < H && len > 0) {
      if (pooling == 1) {
        v = x[static_cast<int64_t>(t0) * H + c];
      } else {
        float acc = 0.0f;
        for (int t = 0; t < len; ++t) acc += x[static_cast<int64_t>(t0 + t) * H + c];
        v = acc / fmaxf(static_cast<float>(len), 1e-9f);
      }
    }
    vals[j] = v;
    ss += v * v;
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]);
  const float den = normalize ? fmaxf(norm, 1e-12f) : 1.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int c = threadIdx.x + 256 * j;
    if (c < H) out[static_cast<int64_t>(seq) * H + c] = vals[j] / den;
  }
}

// dst[i] = src[first token of sequence seq0 + i]; rows of `row_f4` float4s (an f32 row of H floats and
// an interleaved (hi, lo) f16 row of 2H halfs have the same 4H bytes). One wave per row.
__global__ __launch_bounds__(256) void gather_first_rows_kernel(const float4* __restrict__ src,
                                                                const int32_t* __restrict__ cu, int seq0,
                                                                int tok_base, int n, int T, int row_f4,
                                                                float4* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int t = min(cu[seq0 + i] - tok_base, T - 1);  // an empty sequence reads a neighbour's row; pooling ignores it
  for (int c = lane; c < row_f4; c += 64) dst[static_cast<int64_t>(i) * row_f4 + c] = src[static_cast<int64_t>(t) * row_f4 + c];
}

// dst[i] = LayerNorm output of the first token of sequence seq0 + i, re-derived from the pre-LN rows
// and their (mean, 1/sigma) — the f32 hidden state is not stored in f16 mode. One wave per row.
__global__ __launch_bounds__(256) void gather_first_rows_ln_kernel(const float* __restrict__ pre,
                                                                   const float2* __restrict__ stat,
                                                                   const float* __restrict__ g,
                                                                   const float* __restrict__ b,
                                                                   const int32_t* __restrict__ cu, int seq0,
                                                                   int tok_base, int n, int T, int H,
                                                                   float* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int t = min(cu[seq0 + i] - tok_base, T - 1);
  const float2 st = stat[t];
  for (int c = lane; c < H; c += 64)
    dst[static_cast<int64_t>(i) * H + c] = ln_apply(pre[static_cast<int64_t>(t) * H + c], st.x, st.y, g[c], b[c]);
}

// the same from f16 pre-LN rows (f16 residual stream, see EPI_RLS_*)
__global__ __launch_bounds__(256) void gather_first_rows_ln16_kernel(const half_t* __restrict__ pre,
                                                                     const float2* __restrict__ stat,
                                                                     const float* __restrict__ g,
                                                                     const float* __restrict__ b,
                                                                     const int32_t* __restrict__ cu, int seq0,
                                                                     int tok_base, int n, int T, int H,
                                                                     float* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int t = min(cu[seq0 + i] - tok_base, T - 1);
  const float2 st = stat[t];
  for (int c = lane; c < H; c += 64)
    dst[static_cast<int64_t>(i) * H + c] =
        ln_apply(static_cast<float>(pre[static_cast<int64_t>(t) * H + c]), st.x, st.y, g[c], b[c]);
}

// ---- GEMM: C[M,N] = A[M,K] W[N,K]^T + bias (+ GELU | + R) ---------------------------------------

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDT = BK + 4;  // padded LDS row (floats); rows stay 16-byte aligned

// F16: plain f16 output; RESIDUAL_LN: the residual is LayerNorm(R) re-derived from the pre-LN rows R and
// their (mean, 1/sigma) — both in the 256-tile kernel only
enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RESIDUAL = 2, EPI_BIAS_F16 = 3, EPI_BIAS_RESIDUAL_LN = 4,
       // LayerNorm folded into the GEMMs of the large-batch f16 path (forward_chunk, `fold_big`):
       //   LN(x) W^T = inv (x (g o W)^T - mean colsum(g o W)) + W b  — the consumer reads f16 PRE-LN rows and its
       //   epilogue applies the row statistics (FOLD_*: bias = c, ln_g = colsum, ln_stat = (mean, 1/sigma));
       //   the producer (RESIDUAL_LN_STATS) also stores its output rows as f16 and per-(row, 64 columns)
       //   partial (sum, sum of squares), from which ln_finalize_kernel makes the statistics.
       EPI_FOLD_F16 = 5, EPI_FOLD_GELU = 6, EPI_BIAS_RESIDUAL_LN_STATS = 7,
       // ... with an f16 RESIDUAL STREAM (gemm_f16_pp_kernel only): the pre-LayerNorm rows exist as f16 only — the
       // array the next projection reads anyway — so the producer stores 2 bytes per element instead of 6 and
       // the residual read is 2 bytes instead of 4 (the statistics still come from the f32 sums before the
       // rounding). R32_O16: residual rows still f32 (layer 0: the embedding sum), f16 out; R16_O16: f16 in and
       // out, IN PLACE (every element is read and written by the same lane); R16_O32: f16 in, f32 + f16 out
       // (the last layer of a mean-pooled model: the final LayerNorm and the pooling read f32 rows).
       EPI_RLS_R32_O16 = 8, EPI_RLS_R16_O16 = 9, EPI_RLS_R16_O32 = 10 };

// Linear tile id -> (row panel, column panel), row panels taken kGroupM at a time with the column
// index slow inside a group. The ~32 blocks an XCD runs together then cover ~8 row panels x ~4
// column panels, so both operands' working set (~4 MB per sweep of K) fits the XCD's 4-MiB L2;
// with the plain column-fastest order all of W (7-9 MB) cycles through L2 for every row panel.
constexpr int kGroupM = 8;
__device__ __forceinline__ void grouped_tile(int id, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int per_group = kGroupM * tiles_n;
  const int group = id / per_group;
  const int within = id - group * per_group;
  const int gm = min(kGroupM, tiles_m - group * kGroupM);
  tm = group * kGroupM + within % gm;
  tn = within / gm;
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A,
                                                       const float* __restrict__ W,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ R,
                                                       float* __restrict__ C, int M, int N, int K) {
  __shared__ float lds[2 * (BM + BN) * LDT];
  float* sA = lds;                 // [2][BM*LDT]
  float* sB = lds + 2 * BM * LDT;  // [2][BN*LDT]

  // XCD-aware tile order: blocks that share an XCD (blockIdx % 8) walk consecutive tiles of one
  // row panel, so the A panel and the W columns they re-read stay in that XCD's L2.
  const int tiles_n = N / BN;
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / 8;
  int tm, tn;
  grouped_tile(swz, (M + BM - 1) / BM, tiles_n, tm, tn);
  const int bm = tm * BM;
  const int bn = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // global -> register staging: thread covers rows (tid>>3) + 32p, 16 bytes at column (tid&7)*4
  const int lrow = tid >> 3;
  const int lcol = (tid & 7) * 4;
  // A rows past M are clamped to row M-1: they only feed C rows that are never stored.
  const float* w_g = W + static_cast<int64_t>(bn + lrow) * K + lcol;
  const int64_t rstep = static_cast<int64_t>(32) * K;
  const int last = M - 1;
  const float* a_g0 = A + static_cast<int64_t>(min(bm + lrow, last)) * K + lcol;
  const float* a_g1 = A + static_cast<int64_t>(min(bm + lrow + 32, last)) * K + lcol;
  const float* a_g2 = A + static_cast<int64_t>(min(bm + lrow + 64, last)) * K + lcol;
  const float* a_g3 = A + static_cast<int64_t>(min(bm + lrow + 96, last)) * K + lcol;
  float4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
#define VR_LOAD_TILE(k0)                                                                         \
  do {                                                                                           \
    ra0 = *reinterpret_cast<const float4*>(a_g0 + (k0));                                         \
    ra1 = *reinterpret_cast<const float4*>(a_g1 + (k0));                                         \
    ra2 = *reinterpret_cast<const float4*>(a_g2 + (k0));                                         \
    ra3 = *reinterpret_cast<const float4*>(a_g3 + (k0));                                         \
    rw0 = *reinterpret_cast<const float4*>(w_g + (k0));                                          \
    rw1 = *reinterpret_cast<const float4*>(w_g + rstep + (k0));                                  \
    rw2 = *reinterpret_cast<const float4*>(w_g + 2 * rstep + (k0));                              \
    rw3 = *reinterpret_cast<const float4*>(w_g + 3 * rstep + (k0));                              \
  } while (0)
#define VR_STORE_TILE(buf)                                                                       \
  do {                                                                                           \
    float* da = sA + (buf) * BM * LDT + lrow * LDT + lcol;                                       \
    float* db = sB + (buf) * BN * LDT + lrow * LDT + lcol;                                       \
    *reinterpret_cast<float4*>(da) = ra0;                                                        \
    *reinterpret_cast<float4*>(da + 32 * LDT) = ra1;                                             \
    *reinterpret_cast<float4*>(da + 64 * LDT) = ra2;                                             \
    *reinterpret_cast<float4*>(da + 96 * LDT) = ra3;                                             \
    *reinterpret_cast<float4*>(db) = rw0;                                                        \
    *reinterpret_cast<float4*>(db + 32 * LDT) = rw1;                                             \
    *reinterpret_cast<float4*>(db + 64 * LDT) = rw2;                                             \
    *reinterpret_cast<float4*>(db + 96 * LDT) = rw3;                                             \
  } while (0)

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nk = K / BK;
  VR_LOAD_TILE(0);
  VR_STORE_TILE(0);
  __syncthreads();

  // MFMA operand addressing: lane l supplies row/col (l & 31) and k-slot (l >> 5); its float4
  // read at k offset kk + 4*(l>>5) feeds four MFMAs (component c covers k = kk + 4*(l>>5) + c;
  // A and B use the same map, so every product pairs equal k).
  const int frow = lane & 31;
  const int fk = 4 * (lane >> 5);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) VR_LOAD_TILE((kt + 1) * BK);
    const float* a_base = sA + buf * BM * LDT + (wm * 64 + frow) * LDT + fk;
    const float* b_base = sB + buf * BN * LDT + (wn * 64 + frow) * LDT + fk;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 8) {
      float4 a0 = *reinterpret_cast<const float4*>(a_base + kk);
      float4 a1 = *reinterpret_cast<const float4*>(a_base + 32 * LDT + kk);
      float4 b0 = *reinterpret_cast<const float4*>(b_base + kk);
      float4 b1 = *reinterpret_cast<const float4*>(b_base + 32 * LDT + kk);
      const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
      const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c], bv[j][c], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) VR_STORE_TILE(buf ^ 1);
    __syncthreads();
  }

#undef VR_LOAD_TILE
#undef VR_STORE_TILE

  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = bn + wn * 64 + j * 32 + (lane & 31);
      const float bs = bias[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = bm + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M) {
          float v = acc[i][j][r] + bs;
          if (EPI == EPI_BIAS_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
          if (EPI == EPI_BIAS_RESIDUAL) v += R[static_cast<int64_t>(row) * N + col];
          C[static_cast<int64_t>(row) * N + col] = v;
        }
      }
    }
}

static int launch_gemm(vr_engine* e, int epi, const float* A, const float* W, const float* bias,
                       const float* R, float* C, int M, int N, int K) {
  VR_CHECK(N % BN == 0 && K % BK == 0, "GEMM shape N=%d K=%d must be multiples of %d / %d", N, K, BN, BK);
  if (M <= 0) return 0;
  hipStream_t s = e->stream;
  const int grid = ((M + BM - 1) / BM) * (N / BN);
  prof_begin(e, VR_PROF_GEMM, 2.0 * M * static_cast<double>(N) * K);
  switch (epi) {
    case EPI_BIAS:
      hipLaunchKernelGGL((gemm_f32_kernel<EPI_BIAS>), dim3(grid), dim3(256), 0, s, A, W, bias, R, C, M, N, K);
      break;
    case EPI_BIAS_GELU:
      hipLaunchKernelGGL((gemm_f32_kernel<EPI_BIAS_GELU>), dim3(grid), dim3(256), 0, s, A, W, bias, R, C, M, N, K);
      break;
    default:
      hipLaunchKernelGGL((gemm_f32_kernel<EPI_BIAS_RESIDUAL>), dim3(grid), dim3(256), 0, s, A, W, bias, R, C, M, N, K);
      break;
  }
  prof_end(e);
  VR_HIP(hipGetLastError());
  return 0;
}

// ---- GEMM, split precision: C = (A_hi + A_lo)(W_hi + W_lo)^T * unscale + bias ... ---------------------

constexpr int HBM_ = 128, HBN_ = 128, HBK_ = 64;  // block tile; 64 f16 along K = one 128-B line per row
constexpr int HLDT = HBK_ + 8;                     // padded LDS row: 72 halfs = 144 B (conflict-free b128 reads)

// Same 128x128 block / 64x64 wave decomposition as gemm_f32_kernel, on v_mfma_f32_32x32x16_f16:
// per 16-deep k-step a wave reads hi and lo fragments of 2 A tiles and 2 B tiles (8 ds_read_b128)
// and issues 2*2*3 MFMAs (hi*hi, hi*lo, lo*hi) into one f32 accumulator per tile.
// Staging moves whole 128-B lines (8 lanes x 16 B per row): the next K-tile is loaded into
// registers while the current one is multiplied out of the single 72-KiB LDS image (2 blocks/CU).
// EPI_BIAS: f32 out. EPI_BIAS_GELU: split (hi, lo) out only. EPI_BIAS_RESIDUAL: + R, f32 out.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f16x3_kernel(
    const half_t* __restrict__ Ah, const half_t* __restrict__ Al, const half_t* __restrict__ Wh,
    const half_t* __restrict__ Wl, const float* __restrict__ bias, const float* __restrict__ R,
    float* __restrict__ C, half_t* __restrict__ Ch, half_t* __restrict__ Cl, int M, int N, int K,
    float unscale) {
  __shared__ half_t lds[4 * HBM_ * HLDT];  // [Ah, Al, Wh, Wl][128][HLDT] = 72 KiB
  const int tiles_n = N / HBN_;
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / 8;
  int tm, tn;
  grouped_tile(swz, (M + HBM_ - 1) / HBM_, tiles_n, tm, tn);
  const int bm = tm * HBM_;
  const int bn = tn * HBN_;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // staging: thread t moves the 16-B piece (t & 7) of rows (t >> 3) + 32p, p = 0..3, of each array
  const int lrow = tid >> 3;
  const int lcol = (tid & 7) * 8;
  const int last = M - 1;
  // (named registers, not arrays: arrays indexed inside macros ended up in scratch memory)
  // (interleaved (hi, lo) rows of 2K halfs, see split_at: 8 halfs of k sit 16 apart)
  const int64_t a_off0 = static_cast<int64_t>(min(bm + lrow, last)) * (2 * K) + 2 * lcol;
  const int64_t a_off1 = static_cast<int64_t>(min(bm + lrow + 32, last)) * (2 * K) + 2 * lcol;
  const int64_t a_off2 = static_cast<int64_t>(min(bm + lrow + 64, last)) * (2 * K) + 2 * lcol;
  const int64_t a_off3 = static_cast<int64_t>(min(bm + lrow + 96, last)) * (2 * K) + 2 * lcol;
  const int64_t w_off0 = static_cast<int64_t>(bn + lrow) * (2 * K) + 2 * lcol;
  const int64_t w_step = static_cast<int64_t>(32) * (2 * K);
  uint4 r_ah0, r_ah1, r_ah2, r_ah3, r_al0, r_al1, r_al2, r_al3;
  uint4 r_wh0, r_wh1, r_wh2, r_wh3, r_wl0, r_wl1, r_wl2, r_wl3;
#define VR_HLOAD1(P, k0)                                                              \
  r_ah##P = *reinterpret_cast<const uint4*>(Ah + a_off##P + 2 * (k0));                \
  r_al##P = *reinterpret_cast<const uint4*>(Al + a_off##P + 2 * (k0));                \
  r_wh##P = *reinterpret_cast<const uint4*>(Wh + w_off0 + P * w_step + 2 * (k0));     \
  r_wl##P = *reinterpret_cast<const uint4*>(Wl + w_off0 + P * w_step + 2 * (k0));
#define VR_HLOAD(k0) \
  do {               \
    VR_HLOAD1(0, k0) VR_HLOAD1(1, k0) VR_HLOAD1(2, k0) VR_HLOAD1(3, k0) \
  } while (0)
#define VR_HSTORE1(P)                                                                 \
  *reinterpret_cast<uint4*>(lds + (lrow + 32 * P) * HLDT + lcol) = r_ah##P;                          \
  *reinterpret_cast<uint4*>(lds + HBM_ * HLDT + (lrow + 32 * P) * HLDT + lcol) = r_al##P;            \
  *reinterpret_cast<uint4*>(lds + 2 * HBM_ * HLDT + (lrow + 32 * P) * HLDT + lcol) = r_wh##P;        \
  *reinterpret_cast<uint4*>(lds + 3 * HBM_ * HLDT + (lrow + 32 * P) * HLDT + lcol) = r_wl##P;
#define VR_HSTORE() \
  do {              \
    VR_HSTORE1(0) VR_HSTORE1(1) VR_HSTORE1(2) VR_HSTORE1(3) \
  } while (0)

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nk = K / HBK_;
  VR_HLOAD(0);
  VR_HSTORE();
  __syncthreads();

  // 32x32x16 f16 operands: lane l supplies row/col (l & 31), k = 8*(l >> 5) + j, j = 0..7 (16 B)
  const int frow = lane & 31;
  const int fk = 8 * (lane >> 5);
  const half_t* pa = lds + (wm * 64 + frow) * HLDT + fk;
  const half_t* pw = lds + 2 * HBM_ * HLDT + (wn * 64 + frow) * HLDT + fk;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) VR_HLOAD((kt + 1) * HBK_);
#pragma unroll
    for (int kk = 0; kk < HBK_; kk += 16) {
      f16x8 ah[2], al[2], wh[2], wl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const f16x8*>(pa + i * 32 * HLDT + kk);
        al[i] = *reinterpret_cast<const f16x8*>(pa + HBM_ * HLDT + i * 32 * HLDT + kk);
        wh[i] = *reinterpret_cast<const f16x8*>(pw + i * 32 * HLDT + kk);
        wl[i] = *reinterpret_cast<const f16x8*>(pw + HBM_ * HLDT + i * 32 * HLDT + kk);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();  // every wave is done reading this K-tile
    if (kt + 1 < nk) VR_HSTORE();
    __syncthreads();
  }
#undef VR_HLOAD
#undef VR_HLOAD1
#undef VR_HSTORE
#undef VR_HSTORE1

  // Epilogue through LDS (free now: the loop ended on a barrier). In the MFMA's C layout a lane
  // owns one column and 16 scattered rows, which gives 4-byte (or, split, 2-byte) stores — they
  // cost a quarter of the kernel. Each wave parks its 64x64 tile in its own LDS region and reads
  // it back by rows: 16 lanes x 16 B cover a 256-B row segment, bias / GELU / residual / split are
  // applied on float4s, and every global access is 16 B (8 B for the f16 halves).
  constexpr int SLD = 64 + 4;  // padded row of the staging tile (floats)
  float* stage = reinterpret_cast<float*>(lds) + wave * (64 * SLD);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        stage[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SLD + j * 32 + (lane & 31)] = acc[i][j][r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();  // the region is private to this wave: no block barrier needed
  const int c4 = (lane & 15) * 4;
  const int gcol = bn + wn * 64 + c4;
  const float4 b4 = *reinterpret_cast<const float4*>(bias + gcol);
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int lr = it * 4 + (lane >> 4);
    const int grow = bm + wm * 64 + lr;
    float4 v = *reinterpret_cast<const float4*>(stage + lr * SLD + c4);
    if (grow >= M) continue;
    v.x = v.x * unscale + b4.x;
    v.y = v.y * unscale + b4.y;
    v.z = v.z * unscale + b4.z;
    v.w = v.w * unscale + b4.w;
    const int64_t o = static_cast<int64_t>(grow) * N + gcol;
    if (EPI == EPI_BIAS_GELU) {
      float g[4] = {v.x, v.y, v.z, v.w};
      half_t h[4], l[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        g[c] = gelu_fast(g[c]);
        split_f16(g[c], h[c], l[c]);
      }
      const int64_t so = static_cast<int64_t>(grow) * (2 * N) + split_at(gcol);
      *reinterpret_cast<uint2*>(Ch + so) = *reinterpret_cast<const uint2*>(h);
      *reinterpret_cast<uint2*>(Cl + so) = *reinterpret_cast<const uint2*>(l);
    } else {
      if (EPI == EPI_BIAS_RESIDUAL) {
        const float4 r4 = *reinterpret_cast<const float4*>(R + o);
        v.x += r4.x;
        v.y += r4.y;
        v.z += r4.z;
        v.w += r4.w;
      }
      *reinterpret_cast<float4*>(C + o) = v;
    }
  }
}

// ---- the same product on a 256x256 block tile -----------------------------------------------------
//
// 8 waves as 2(M) x 4(N), wave tile 128x64 = 4x2 MFMA tiles: per 16-deep k-step a wave reads 12
// fragments (8 of A, 4 of W; hi and lo) for 24 MFMAs — half the LDS bytes per MFMA of the 128x128
// kernel above, whose LDS pipe was as busy as its matrix pipe. One block per CU (128 KiB of LDS):
// two 64-KiB stages [A, W][256 rows][128 B = 32 k of hi and lo], filled by global_load_lds_dwordx4 (no
// staging registers, no ds_write pass) for K-tile t+1 while K-tile t is multiplied; one barrier per
// K-tile. A direct-to-LDS load writes wave-uniform base + lane*16, so the image is unpadded; bank
// conflicts are avoided by an XOR swizzle applied on the SOURCE side: LDS chunk p of row r holds
// the row's 16-byte chunk p ^ ((r >> 1) & 7), and fragment reads use the same permutation (every
// lane group of a ds_read_b128 then covers all 64 banks once).

constexpr int GBM = 256, GBN = 256, GBK = 32;
constexpr int kStageHalfs = 4 * 256 * GBK;  // 64 KiB
constexpr int kGroupM256 = 4;

__device__ __forceinline__ void glds16(const half_t* src, half_t* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// PASSES = 3: split operands, interleaved (hi, lo) rows of 2K halfs (split_at), 32-deep K-tiles.
// PASSES = 1: plain f16 operands (VR_PRECISION_F16), rows of K halfs, 64-deep K-tiles. Either way a
// K-tile of one row is one 128-byte line, so staging, swizzle and LDS image are the same code.
template <int EPI, int PASSES>
__global__ __launch_bounds__(512) void gemm_f16x3_256_kernel(
    const half_t* __restrict__ Ah, const half_t* __restrict__ Al, const half_t* __restrict__ Wh,
    const half_t* __restrict__ Wl, const float* __restrict__ bias, const float* __restrict__ R,
    float* __restrict__ C, half_t* __restrict__ Ch, half_t* __restrict__ Cl, int M, int N, int K,
    float unscale, const float2* __restrict__ ln_stat, const float* __restrict__ ln_g,
    const float* __restrict__ ln_b) {
  __shared__ half_t lds[2 * kStageHalfs];  // the only LDS object of the kernel (a second one makes hipcc
                                           // drain the in-flight loads before every fragment read)
  // Persistent blocks: the grid is one block per CU and every block walks tiles g, g + G, g + 2G, ...
  // Inside a wave of G tiles the blocks that share an XCD (blockIdx % 8) take consecutive tiles, and
  // consecutive tiles are grouped kGroupM256 row panels at a time with the column index slow, so the
  // A panels and W columns an XCD re-reads stay in its L2.
  const int tiles_n = (N + GBN - 1) / GBN;  // N may end inside the last column tile (rows clamped, stores guarded)
  const int tiles_m = (M + GBM - 1) / GBM;
  const int total = tiles_m * tiles_n;
  const int G = gridDim.x;
  const int local = (G % 8 == 0) ? (static_cast<int>(blockIdx.x) % 8) * (G / 8) + static_cast<int>(blockIdx.x) / 8
                                 : static_cast<int>(blockIdx.x);
  auto coords = [&](int t, int& bm_, int& bn_) {
    const int per_group = kGroupM256 * tiles_n;
    const int group = t / per_group;
    const int within = t - group * per_group;
    const int gm = min(kGroupM256, tiles_m - group * kGroupM256);
    bm_ = (group * kGroupM256 + within % gm) * GBM;
    bn_ = (within / gm) * GBN;
  };
  int tile = local;
  if (tile >= total) return;  // block-uniform
  int bm, bn;
  coords(tile, bm, bn);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // staging: a direct-to-LDS load moves 8 rows x 128 B (one line per row: 32 k of hi and lo, see
  // split_at). Wave w fills row groups w, w + 8, w + 16, w + 24 of the A image and of the W image;
  // lane -> row (lane >> 3) of the group, LDS chunk (lane & 7), source chunk (lane & 7) ^ ((row >> 1) & 7).
  const int srow = lane >> 3;
  const int schunk = ((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 8;
  const int r0 = wave * 8 + srow;  // + 64 j
  const int64_t K2 = (PASSES == 3 ? 2 : 1) * static_cast<int64_t>(K);  // halfs per operand row
  const half_t *g_a0, *g_a1, *g_a2, *g_a3, *g_w0, *g_w1, *g_w2, *g_w3;
  auto set_ptrs = [&](int bm_, int bn_) {
    g_a0 = Ah + static_cast<int64_t>(min(bm_ + r0, M - 1)) * K2 + schunk;
    g_a1 = Ah + static_cast<int64_t>(min(bm_ + r0 + 64, M - 1)) * K2 + schunk;
    g_a2 = Ah + static_cast<int64_t>(min(bm_ + r0 + 128, M - 1)) * K2 + schunk;
    g_a3 = Ah + static_cast<int64_t>(min(bm_ + r0 + 192, M - 1)) * K2 + schunk;
    g_w0 = Wh + static_cast<int64_t>(min(bn_ + r0, N - 1)) * K2 + schunk;
    g_w1 = Wh + static_cast<int64_t>(min(bn_ + r0 + 64, N - 1)) * K2 + schunk;
    g_w2 = Wh + static_cast<int64_t>(min(bn_ + r0 + 128, N - 1)) * K2 + schunk;
    g_w3 = Wh + static_cast<int64_t>(min(bn_ + r0 + 192, N - 1)) * K2 + schunk;
  };
  set_ptrs(bm, bn);
  half_t* l_dst = lds + wave * 8 * 64;  // rows of 64 halfs (128 B); the W image starts at 256 * 64
#define VR_GLDS_STAGE(buf, k0)                                          \
  do {                                                                  \
    half_t* d = l_dst + (buf) * kStageHalfs;                            \
    glds16(g_a0 + (k0), d);                                         \
    glds16(g_a1 + (k0), d + 64 * 64);                               \
    glds16(g_a2 + (k0), d + 128 * 64);                              \
    glds16(g_a3 + (k0), d + 192 * 64);                              \
    glds16(g_w0 + (k0), d + 256 * 64);                              \
    glds16(g_w1 + (k0), d + 256 * 64 + 64 * 64);                    \
    glds16(g_w2 + (k0), d + 256 * 64 + 128 * 64);                   \
    glds16(g_w3 + (k0), d + 256 * 64 + 192 * 64);                   \
  } while (0)

  f32x4 acc[8][4];

  // v_mfma_f32_16x16x32_f16 fragments: lane l supplies row (l & 15), k = 8*(l >> 4) + j of a 32-deep step
  // (the chip holds a higher clock on this shape than on 32x32x16 at the same work per cycle).
  // PASSES 3: k8-group g = l >> 4 of the 32-deep tile, hi = chunk 2g, lo = chunk 2g + 1.
  // PASSES 1: chunk 4*kk + (l >> 4) of the 64-deep tile.
  const int frow = lane & 15;
  const int fsw = (frow >> 1) & 7;
  const int fq = lane >> 4;
  const int pa = (wm * 128 + frow) * 64;
  const int pw = 256 * 64 + (wn * 64 + frow) * 64;

  constexpr int kTileHalfs = 64;               // halfs of one row per K-tile: 32 k x (hi, lo) or 64 k
  const int nk = K / (PASSES == 3 ? 32 : 64);
  int par = 0;  // stage buffer holding K-tile 0 of the current tile
  VR_GLDS_STAGE(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  while (true) {
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the next tile of this block: its first K-tile is loaded during this tile's LAST K-tile, so only the
  // first tile of a block pays a prologue, and this tile's stores drain under the next main loop
  const int next = tile + G;
  const bool has_next = next < total;
  int nbm = bm, nbn = bn;
  if (has_next) coords(next, nbm, nbn);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt == nk - 1 && has_next) set_ptrs(nbm, nbn);
    const half_t* st = lds + ((par + kt) & 1) * kStageHalfs;
    // A direct-to-LDS load holds the issuing wave for ~100+ cycles. Eight of them in a row at the top
    // of the K-tile left the matrix pipe idle for a third of it; instead one load is issued after
    // every row of MFMA tiles, where the partner wave of the SIMD fills the gap.
    half_t* nd = l_dst + ((par + kt + 1) & 1) * kStageHalfs;
    const int nk0 = kt + 1 < nk ? (kt + 1) * kTileHalfs : 0;  // K-tile 0 of the next tile (or, at the very end, a harmless re-load)
    constexpr int kSteps = PASSES == 3 ? 1 : 2;  // 32-deep MFMA steps per K-tile
#pragma unroll
    for (int kk = 0; kk < kSteps; ++kk) {
      // the four weight fragments stay resident for the step; activation fragments are read one row
      // of tiles ahead of their MFMAs
      f16x8 wh[4], wl[4], ah[2], al[2];
      const int fh = PASSES == 3 ? ((2 * fq) ^ fsw) * 8 : ((4 * kk + fq) ^ fsw) * 8;
      const int fl = ((2 * fq + 1) ^ fsw) * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        wh[j] = *reinterpret_cast<const f16x8*>(st + pw + j * 16 * 64 + fh);
        if (PASSES == 3) wl[j] = *reinterpret_cast<const f16x8*>(st + pw + j * 16 * 64 + fl);
      }
      ah[0] = *reinterpret_cast<const f16x8*>(st + pa + fh);
      if (PASSES == 3) al[0] = *reinterpret_cast<const f16x8*>(st + pa + fl);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i + 1 < 8) {
          ah[(i + 1) & 1] = *reinterpret_cast<const f16x8*>(st + pa + (i + 1) * 16 * 64 + fh);
          if (PASSES == 3) al[(i + 1) & 1] = *reinterpret_cast<const f16x8*>(st + pa + (i + 1) * 16 * 64 + fl);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i & 1], wh[j], acc[i][j], 0, 0, 0);
          if (PASSES == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i & 1], wl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i & 1], wh[j], acc[i][j], 0, 0, 0);
          }
        }
        if (kk == 0) {
          const int piece = i;  // compile-time after unrolling
          const half_t* src = piece == 0 ? g_a0 : piece == 1 ? g_a1 : piece == 2 ? g_a2 : piece == 3 ? g_a3
                            : piece == 4 ? g_w0 : piece == 5 ? g_w1 : piece == 6 ? g_w2 : g_w3;
          glds16(src + nk0, nd + (piece >> 2) * 256 * 64 + (piece & 3) * 64 * 64);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the load where it was written, between the MFMA rows
        // (measured: dropping this barrier, or s_setprio(1) around the MFMA rows, changes nothing: +-0.5 %)
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of K-tile kt+1 have landed
    __syncthreads();                                   // ... everybody's have, and K-tile kt is consumed
  }

  // Epilogue through LDS, 16 rows of the wave at a time (see gemm_f16x3_kernel): the wave parks 16x64
  // accumulators in its own region of the stage buffer that does NOT hold the next tile's first K-tile
  // and reads them back by rows for 16-byte accesses.
  // The residual rows of a tile are requested one tile AHEAD, all eight loads at once: read one by
  // one inside the store loop they cost eight exposed HBM latencies per tile.
  constexpr int SLD = 64 + 4;
  const int sbuf = ((par + nk) & 1) ^ 1;
  float* stage = reinterpret_cast<float*>(lds) + sbuf * (kStageHalfs / 2) + wave * (16 * SLD);
  const int c4 = (lane & 15) * 4;
  const int gcol = bn + wn * 64 + c4;
  const bool col_ok = gcol < N;  // N % 4 == 0: a lane's four columns are in or out together
  const float4 b4 = *reinterpret_cast<const float4*>(bias + (col_ok ? gcol : 0));
  constexpr bool kFold = EPI == EPI_FOLD_F16 || EPI == EPI_FOLD_GELU;
  constexpr bool kStats = EPI == EPI_BIAS_RESIDUAL_LN_STATS;
  constexpr bool kResidLN = EPI == EPI_BIAS_RESIDUAL_LN || kStats;
  constexpr bool kResidual = EPI == EPI_BIAS_RESIDUAL || kResidLN;
  // LayerNorm weight / bias of this lane's columns (residual re-derivation), or, folded consumers: the
  // column sums of the LayerNorm-scaled weight matrix
  float4 lg4 = make_float4(0.f, 0.f, 0.f, 0.f), lb4 = lg4;
  if (kResidLN || kFold) lg4 = *reinterpret_cast<const float4*>(ln_g + (col_ok ? gcol : 0));
  if (kResidLN) lb4 = *reinterpret_cast<const float4*>(ln_b + (col_ok ? gcol : 0));
  // The residual rows of a 16-row piece are requested one piece AHEAD, four loads at once (read one
  // by one inside the store loop they cost an exposed HBM latency each), with their (mean, 1/sigma)
  // when the residual is a LayerNorm output.
  float4 r4[2][4];
  float2 st4[2][4];
  auto fetch_residual = [&](int pc, float4 (&r)[4], float2 (&st)[4]) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int64_t rr = min(bm + wm * 128 + 16 * pc + (lane >> 4) + 4 * it, M - 1);
      if (kResidual) r[it] = *reinterpret_cast<const float4*>(R + rr * N + (col_ok ? gcol : 0));
      if (kResidLN || kFold) st[it] = ln_stat[rr];
    }
  };
  if (kResidual || kFold) fetch_residual(0, r4[0], st4[0]);
#pragma unroll
  for (int pc = 0; pc < 8; ++pc) {  // piece pc = rows 16 pc .. 16 pc + 15 of the wave's 128 = MFMA tile row pc
    const int row0 = bm + wm * 128 + 16 * pc + (lane >> 4);  // + 4 * it
    if ((kResidual || kFold) && pc + 1 < 8) fetch_residual(pc + 1, r4[(pc + 1) & 1], st4[(pc + 1) & 1]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        stage[(4 * (lane >> 4) + r) * SLD + j * 16 + (lane & 15)] = acc[pc][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    auto emit = [&](int it) {
      const int lr = it * 4 + (lane >> 4);
      const int grow = row0 + 4 * it;
      float4 v = *reinterpret_cast<const float4*>(stage + lr * SLD + c4);
      const bool ok = grow < M && col_ok;
      if (!kStats && !ok) return;  // (the statistics variant keeps every lane for its row sums)
      if (kFold) {  // inv (acc unscale - mean colsum) + c
        const float2 st = st4[pc & 1][it];
        v.x = fmaf(fmaf(v.x, unscale, -(st.x * lg4.x)), st.y, b4.x);
        v.y = fmaf(fmaf(v.y, unscale, -(st.x * lg4.y)), st.y, b4.y);
        v.z = fmaf(fmaf(v.z, unscale, -(st.x * lg4.z)), st.y, b4.z);
        v.w = fmaf(fmaf(v.w, unscale, -(st.x * lg4.w)), st.y, b4.w);
      } else {
        v.x = v.x * unscale + b4.x;
        v.y = v.y * unscale + b4.y;
        v.z = v.z * unscale + b4.z;
        v.w = v.w * unscale + b4.w;
      }
      const int64_t o = static_cast<int64_t>(grow) * N + gcol;
      if (EPI == EPI_BIAS_GELU || EPI == EPI_FOLD_GELU) {
        const f32x2 g01 = PASSES == 1 ? gelu_poly2(f32x2{v.x, v.y}) : gelu_fast2(f32x2{v.x, v.y});
        const f32x2 g23 = PASSES == 1 ? gelu_poly2(f32x2{v.z, v.w}) : gelu_fast2(f32x2{v.z, v.w});
        float g[4] = {g01.x, g01.y, g23.x, g23.y};
        half_t h[4], l[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) split_f16(g[c], h[c], l[c]);
        if (PASSES == 3) {
          const int64_t so = static_cast<int64_t>(grow) * (2 * N) + split_at(gcol);
          *reinterpret_cast<uint2*>(Ch + so) = *reinterpret_cast<const uint2*>(h);
          *reinterpret_cast<uint2*>(Cl + so) = *reinterpret_cast<const uint2*>(l);
        } else {
          *reinterpret_cast<uint2*>(Ch + o) = *reinterpret_cast<const uint2*>(h);  // plain f16 rows
        }
      } else if (EPI == EPI_BIAS_F16 || EPI == EPI_FOLD_F16) {  // Q, K, V for the f16 attention kernel
        half_t h[4] = {static_cast<half_t>(fminf(fmaxf(v.x, -65504.0f), 65504.0f)),
                       static_cast<half_t>(fminf(fmaxf(v.y, -65504.0f), 65504.0f)),
                       static_cast<half_t>(fminf(fmaxf(v.z, -65504.0f), 65504.0f)),
                       static_cast<half_t>(fminf(fmaxf(v.w, -65504.0f), 65504.0f))};
        *reinterpret_cast<uint2*>(Ch + o) = *reinterpret_cast<const uint2*>(h);
      } else {
        if (EPI == EPI_BIAS_RESIDUAL) {
          v.x += r4[pc & 1][it].x;
          v.y += r4[pc & 1][it].y;
          v.z += r4[pc & 1][it].z;
          v.w += r4[pc & 1][it].w;
        }
        if (kResidLN) {  // residual = LayerNorm(R row), exactly as the LN kernel would have stored it
          const float2 st = st4[pc & 1][it];
          v.x += ln_apply(r4[pc & 1][it].x, st.x, st.y, lg4.x, lb4.x);
          v.y += ln_apply(r4[pc & 1][it].y, st.x, st.y, lg4.y, lb4.y);
          v.z += ln_apply(r4[pc & 1][it].z, st.x, st.y, lg4.z, lb4.z);
          v.w += ln_apply(r4[pc & 1][it].w, st.x, st.y, lg4.w, lb4.w);
        }
        if (!kStats) {
          *reinterpret_cast<float4*>(C + o) = v;
        } else {
          // the pre-LayerNorm row also goes out as f16 (the next projection's operand), and this wave's 64
          // columns of it contribute a (sum, sum of squares) to the row's statistics
          float s1 = 0.0f, s2 = 0.0f;
          if (ok) {
            *reinterpret_cast<float4*>(C + o) = v;
            half_t h[4] = {static_cast<half_t>(fminf(fmaxf(v.x, -65504.0f), 65504.0f)),
                           static_cast<half_t>(fminf(fmaxf(v.y, -65504.0f), 65504.0f)),
                           static_cast<half_t>(fminf(fmaxf(v.z, -65504.0f), 65504.0f)),
                           static_cast<half_t>(fminf(fmaxf(v.w, -65504.0f), 65504.0f))};
            *reinterpret_cast<uint2*>(Ch + o) = *reinterpret_cast<const uint2*>(h);
            s1 = (v.x + v.y) + (v.z + v.w);
            s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
          }
#pragma unroll
          for (int off = 1; off < 16; off <<= 1) {  // the 16 lanes that share the row
            s1 += __shfl_xor(s1, off);
            s2 += __shfl_xor(s2, off);
          }
          const int seg = (bn + wn * 64) >> 6;
          if ((lane & 15) == 0 && grow < M && seg < (N >> 6))
            reinterpret_cast<float2*>(Cl)[static_cast<int64_t>(grow) * (N >> 6) + seg] = make_float2(s1, s2);
        }
      }
    };
#pragma unroll
    for (int it = 0; it < 4; ++it) emit(it);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();  // the rows are read before the next piece overwrites the region
  }
  if (!has_next) break;
  par = (par + nk) & 1;
  tile = next;
  bm = nbm;
  bn = nbn;
  __syncthreads();  // every wave has left the staging buffer before the next tile's loads land in it
  }  // tiles
#undef VR_GLDS_STAGE
}

// The epilogue of the f16 kernels whose MFMAs take the WEIGHT fragment as their A operand (gemm_f16_pp_kernel,
// gemm_f16_d2_kernel), straight from the accumulators of a wave's 128-token x 64-feature tile at token row
// `row0`, feature `col0`: acc[i][j][r] of lane (t = lane & 15, g = lane >> 4) is token row 16 i + t, feature 4 g + r of
// weight fragment j — four CONSECUTIVE features of one token — and the W image was staged with its rows permuted so
// that fragments 2p and 2p + 1 hold features 32 p + 8 g + {0..3} and {4..7}: a lane owns 8 consecutive features per
// fragment pair = one 16-byte f16 store (or two 16-byte f32 stores), 64 contiguous bytes per token and instruction
// across the four lane groups. No transpose through LDS (it cost 128 ds_write_b32 + 32 ds_read_b128 per wave and
// tile, about half of the f16-output epilogue) and no LDS use at all.
// FULL: the wave's tile lies inside the matrix, nothing is predicated. This is not about the compares: a store under
// `if (ok)` sits in its own basic block, the waits the compiler places for the loads that are in flight across it
// (bias, residual rows) are then merged over both paths into vmcnt(0) — and on gfx950 vmcnt counts STORES too, so every
// store waited for the one before it: one 1 KiB store in flight per wave, the epilogue latency-bound. In straight-line
// code the counts are exact and the stores stay in flight.
// Diagnostic build only (make diag -> libvoitta_engine_diag.so, selected with VOITTA_ENGINE_LIB): VR_GEMM_DIAG in the
// environment switches parts of gemm_f16_pp_kernel off at run time, so that one box can time the kernel without
// its loads (1), its epilogue (2), its MFMAs (4), its fragment reads (8), its barriers (16), the epilogue's
// stores (32: arithmetic only) or the epilogue's arithmetic (64: stores of the raw accumulators only). Results are then
// wrong by construction; only the timings mean anything. The shipped library compiles none of this.
#ifdef VR_GEMM_DIAG_BUILD
__device__ int g_gemm_diag = 0;
#define VR_DIAG(bit) ((diag_bits & (bit)) != 0)
#else
#define VR_DIAG(bit) false
#endif
// Stamp build (make stamps -> libvoitta_engine_stamps.so; the diag build has them too): the shipped code path plus
// s_memtime stamps — the middle block of the grid, waves 0 and 4, six points of each of its first 16 tiles;
// VR_GEMM_STAMPS=n in the environment prints them for the first n launches (launch_pp; scripts/pp_stamps.sh).
#if defined(VR_GEMM_DIAG_BUILD) || defined(VR_GEMM_STAMP_BUILD)
#define VR_GEMM_HAS_STAMPS 1
__device__ long long g_pp_stamps[2][16][8];
#define VR_PP_STAMP(slot)                                                                  \
  do {                                                                                     \
    if (stamp_on && tile_seq < 16 && (wave & 3) == 0 && lane == 0)                         \
      g_pp_stamps[wave >> 2][tile_seq][slot] = __builtin_amdgcn_s_memtime();                \
  } while (0)
#else
#define VR_PP_STAMP(slot) \
  do {                    \
  } while (0)
#endif

constexpr int kWaveStatHalfs = 128 * 4;  // 128 float2 per wave (direct_epilogue's row statistics), in halfs
constexpr int kTileConstHalfs = (3 * 256 * 4 + 256 * 8) / 2;  // a tile's bias / gain / shift vectors and row statistics

// Output rows are written once and read by the NEXT kernel, long after they have left the caches: non-temporal stores
// (they do not push the weight panels and activation rows the other tiles still need out of L2). Measured +2.7 % on the
// GEMMs and +2 % on the attention kernel that runs between them (profiles/r02_gemm_experiments.md §5);
// -DVR_GEMM_PLAIN_STORES builds the comparison.
template <typename T>
__device__ __forceinline__ void out_store(T* p, const T& v) {
#ifdef VR_GEMM_PLAIN_STORES
  *p = v;
#else
  __builtin_nontemporal_store(v, p);
#endif
}

#ifdef VR_GEMM_NO_XPOSE  // (make noxpose: the A/B twin that stores straight from the accumulators, as round 2 did)
constexpr bool kXposeStores = false;
#else
constexpr bool kXposeStores = true;
#endif

template <int EPI, bool FULL, bool XPOSE = false>
__device__ __forceinline__ void direct_epilogue(f32x4 (&acc)[8][4], int row0, int col0, int lane,
                                                const float* __restrict__ bias, const float* __restrict__ R,
                                                float* __restrict__ C, half_t* __restrict__ Ch, half_t* __restrict__ Cl,
                                                int M, int N, float unscale, const float2* __restrict__ ln_stat,
                                                const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                float2* wave_stat, const float* tile_const = nullptr, int trow0 = 0,
                                                int tcol0 = 0, half_t* xpose = nullptr) {
  asm volatile("" : "+v"(lane));  // (lane-derived offsets are made per tile, not carried through the K loop)
  const int tok = lane & 15, fg = lane >> 4;
#ifdef VR_GEMM_DIAG_BUILD
  const int diag_bits = __builtin_amdgcn_readfirstlane(g_gemm_diag);
#endif
  const int fbase = col0 + 8 * fg;  // + 32 p (+ 4 q): this lane's features
  // FULL tiles address everything as (wave-uniform pointer) + (one 32-bit lane offset): the uniform part — row
  // 16 pc of the tile, column col0 + 32 p2 — lives in scalar registers, where per-lane 64-bit addresses of eight pieces
  // did not fit next to the accumulators. (A chunk's matrices stay below 4 GiB: kMaxChunkTokens.)
  const uint32_t loff = static_cast<uint32_t>(tok * N + 8 * fg);
  auto upiece = [&](auto* base, int pc, int col) { return base + (static_cast<int64_t>(row0 + 16 * pc) * N + col); };
  // XPOSE (gemm_f16_pp_kernel; xpose = 4 KiB of the stage buffer the finished K loop has left free, per wave): the f16 rows go out
  // through a lane exchange. Straight from the accumulators a store instruction has lane (tok, fg) write 16 bytes of token
  // `tok`: the four lanes of every quad address four different rows, and the store path takes one request per (quad, 64-byte
  // segment) — 64 per instruction; a lone block needs 4.5 us to get its 128 KiB tile out, 256 blocks together no longer
  // (scripts/probe_store_burst.hip, profiles/r03_experiments.md §5: the burst is bound per CU by requests, not by HBM).
  // With the 16 x 4 block of 16-byte pieces transposed — lane L gets token L >> 2, piece L & 3 — a quad writes 64 contiguous
  // bytes: 16 requests per instruction, the same bytes in 1.35 us. The exchange is one ds_write_b128 + one ds_read_b128 in
  // a 1-KiB slot only this wave touches (LDS executes a wave's instructions in order: no barrier, no wait in between);
  // 64-byte rows with piece p of token t at p ^ (t >> 2): both sides conflict-free. The store of a piece is issued one piece
  // later, so its LDS round trip runs under the next piece's arithmetic.
  const int xw = tok * 32 + ((fg ^ (tok >> 2)) & 3) * 8;                          // write side, halfs
  const int xtr = lane >> 2, xtc = lane & 3;
  const int xr = xtr * 32 + ((xtc ^ (xtr >> 2)) & 3) * 8;                         // read side
  const uint32_t xloff = static_cast<uint32_t>(xtr * N + 8 * xtc);
  f16x8 xh = {};
  int xpend_pc = -1, xpend_p2 = 0;  // (compile-time after unrolling)
  auto xflush = [&]() {
    if (FULL) {
      out_store(reinterpret_cast<f16x8*>(upiece(Ch, xpend_pc, col0 + 32 * xpend_p2) + xloff), xh);
    } else {
      const int grow2 = row0 + 16 * xpend_pc + xtr, c2 = col0 + 32 * xpend_p2 + 8 * xtc;
      if (grow2 < M && c2 < N) out_store(reinterpret_cast<f16x8*>(Ch + static_cast<int64_t>(grow2) * N + c2), xh);
    }
  };
  constexpr bool kFold = EPI == EPI_FOLD_F16 || EPI == EPI_FOLD_GELU;
  constexpr bool kStats = EPI == EPI_BIAS_RESIDUAL_LN_STATS || EPI == EPI_RLS_R32_O16 || EPI == EPI_RLS_R16_O16 ||
                          EPI == EPI_RLS_R16_O32;
  constexpr bool kR16 = EPI == EPI_RLS_R16_O16 || EPI == EPI_RLS_R16_O32;   // residual rows are f16
  constexpr bool kNoOut32 = EPI == EPI_RLS_R32_O16 || EPI == EPI_RLS_R16_O16;  // pre-LN rows go out as f16 only
  constexpr bool kResidLN = EPI == EPI_BIAS_RESIDUAL_LN || kStats;
  constexpr bool kResidual = EPI == EPI_BIAS_RESIDUAL || kResidLN;
  constexpr bool kGelu = EPI == EPI_BIAS_GELU || EPI == EPI_FOLD_GELU;
  constexpr bool kHalfOut = kGelu || EPI == EPI_BIAS_F16 || EPI == EPI_FOLD_F16 || kNoOut32;
  // fragment pair p2 (features fbase + 32 p2 .. + 7) outside, the eight 16-token pieces inside: the column
  // vectors (bias, LayerNorm gain / shift or column sums) of one pair stay in registers, not those of all four
  // fragments (which, with the residual rows in flight, did not fit next to the 128 accumulators)
  // kFold without residual rows (the QKV and FFN-up projections): the (mean, 1/sigma) of the wave's 128 rows are the
  // only per-row loads — all eight pieces' worth up front, for both fragment pairs (fetched piece by piece, one piece
  // ahead, each of the 16 pieces waited ~1 us for 8 bytes)
  // tile_const (FULL tiles of gemm_f16_pp_kernel): the tile's 256 bias / gain / shift values and its 256 rows' statistics
  // were fetched into LDS while the main loop ran — [bias 256][gain or column sums 256][shift 256][float2 stat 256];
  // trow0 / tcol0 = this wave's first row / column inside the tile
  const bool in_lds = FULL && tile_const != nullptr;
  const float2* lds_stat = reinterpret_cast<const float2*>(tile_const + 768) + trow0;
  constexpr bool kStatUpfront = (EPI == EPI_FOLD_F16 || EPI == EPI_FOLD_GELU);
  float2 st8[kStatUpfront ? 8 : 1] = {};
  if (kStatUpfront && !in_lds) {
#pragma unroll
    for (int pc = 0; pc < 8; ++pc)
      st8[pc] = FULL ? (ln_stat + (row0 + 16 * pc))[static_cast<uint32_t>(tok)] : ln_stat[min(row0 + 16 * pc + tok, M - 1)];
  }
#pragma unroll
  for (int p2 = 0; p2 < 2; ++p2) {
    const int c0 = fbase + 32 * p2;
    const bool col_ok = FULL || c0 < N;  // N % 8 == 0: a lane's eight features are in or out together
    const int cs = col_ok ? c0 : 0;
    float4 b4[2], lg4[2] = {}, lb4[2] = {};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t lc = static_cast<uint32_t>(8 * fg + 4 * q);
      const int uc = col0 + 32 * p2;
      if (in_lds) {
        const float* tc = tile_const + tcol0 + 32 * p2 + lc;
        b4[q] = *reinterpret_cast<const float4*>(tc);
        if (kResidLN || kFold) lg4[q] = *reinterpret_cast<const float4*>(tc + 256);
        if (kResidLN) lb4[q] = *reinterpret_cast<const float4*>(tc + 512);
        continue;
      }
      b4[q] = FULL ? *reinterpret_cast<const float4*>(bias + uc + lc) : *reinterpret_cast<const float4*>(bias + cs + 4 * q);
      if (kResidLN || kFold)
        lg4[q] = FULL ? *reinterpret_cast<const float4*>(ln_g + uc + lc) : *reinterpret_cast<const float4*>(ln_g + cs + 4 * q);
      if (kResidLN)
        lb4[q] = FULL ? *reinterpret_cast<const float4*>(ln_b + uc + lc) : *reinterpret_cast<const float4*>(ln_b + cs + 4 * q);
    }
    // residual rows (and row statistics) of a piece are requested TWO pieces ahead (one piece of arithmetic is ~600 cycles,
    // a global load 1-2 thousand: one ahead left every piece waiting)
    // (EPI_RLS_R16_O16 under gemm_f16_pp_kernel comes here for edge tiles only — whole tiles go to rls16_tile_epilogue — and
    // one piece ahead keeps this path inside the kernel's register budget)
    constexpr int kAhead = (!FULL && EPI == EPI_RLS_R16_O16) ? 1 : 2;
    float4 r4[kAhead + 1][2] = {};
    float2 st2[kAhead + 1] = {};
    auto fetch_residual = [&](int pc, float4 (&r)[2], float2& st) {
      if (FULL) {
        if (kResidual && kR16) {
          r[0] = *reinterpret_cast<const float4*>(upiece(reinterpret_cast<const half_t*>(R), pc, col0 + 32 * p2) + loff);
        } else if (kResidual) {
          r[0] = *reinterpret_cast<const float4*>(upiece(R, pc, col0 + 32 * p2) + loff);
          r[1] = *reinterpret_cast<const float4*>(upiece(R, pc, col0 + 32 * p2 + 4) + loff);
        }
        if (kResidLN || kFold) st = in_lds ? lds_stat[16 * pc + tok] : (ln_stat + (row0 + 16 * pc))[static_cast<uint32_t>(tok)];
        return;
      }
      const int64_t rr = min(row0 + 16 * pc + tok, M - 1);
      if (kResidual && kR16) {  // eight f16 residual values: carried in r[0]'s 16 bytes
        r[0] = *reinterpret_cast<const float4*>(reinterpret_cast<const half_t*>(R) + rr * N + cs);
      } else if (kResidual) {
        r[0] = *reinterpret_cast<const float4*>(R + rr * N + cs);
        r[1] = *reinterpret_cast<const float4*>(R + rr * N + cs + 4);
      }
      if (kResidLN || kFold) st = ln_stat[rr];
    };
    if ((kResidual || kFold) && !kStatUpfront) {
#pragma unroll
      for (int pc = 0; pc < kAhead; ++pc) fetch_residual(pc, r4[pc], st2[pc]);
    }
    // (tile constants in LDS: a piece's row statistics are read one piece ahead — every piece is a basic block of its own,
    // so a read at the point of use cannot be moved up by the compiler and each piece began with an exposed LDS round trip)
    float2 st_ahead = (kStatUpfront && in_lds) ? lds_stat[tok] : float2{};
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) {  // piece pc = token rows 16 pc .. 16 pc + 15 of the wave's 128
      const int grow = row0 + 16 * pc + tok;
      if ((kResidual || kFold) && !kStatUpfront && pc + kAhead < 8)
        fetch_residual(pc + kAhead, r4[(pc + kAhead) % (kAhead + 1)], st2[(pc + kAhead) % (kAhead + 1)]);
      const bool ok = FULL || (grow < M && col_ok);
      const float2 st_lds = st_ahead;
      if (kStatUpfront && in_lds && pc + 1 < 8) {
        st_ahead = lds_stat[16 * (pc + 1) + tok];
        __builtin_amdgcn_sched_barrier(0);  // (left alone, the scheduler sinks the read to the end of the piece)
      }
      const float2 st = kStatUpfront ? (in_lds ? st_lds : st8[pc]) : st2[pc % (kAhead + 1)];
      float v[2][4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float bb[4] = {b4[q].x, b4[q].y, b4[q].z, b4[q].w};
        const float gg[4] = {lg4[q].x, lg4[q].y, lg4[q].z, lg4[q].w};
        const float lb[4] = {lb4[q].x, lb4[q].y, lb4[q].z, lb4[q].w};
        const float4 rq = r4[pc % (kAhead + 1)][kR16 ? 0 : q];
        float rr4[4] = {rq.x, rq.y, rq.z, rq.w};
        if (kR16) {
          const f16x8 rh = *reinterpret_cast<const f16x8*>(&rq);
#pragma unroll
          for (int r = 0; r < 4; ++r) rr4[r] = static_cast<float>(rh[4 * q + r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[pc][2 * p2 + q][r];
          if (VR_DIAG(64)) {
            v[q][r] = x;
            continue;
          }
          if (kFold)  // inv (acc unscale - mean colsum) + c
            x = fmaf(fmaf(x, unscale, -(st.x * gg[r])), st.y, bb[r]);
          else
            x = x * unscale + bb[r];
          if (EPI == EPI_BIAS_RESIDUAL) x += rr4[r];
          if (kResidLN) x += ln_apply(rr4[r], st.x, st.y, gg[r], lb[r]);  // residual = LayerNorm(R row)
          v[q][r] = x;
        }
      }
      if (kGelu && !VR_DIAG(64)) {
        f32x2 g[4] = {f32x2{v[0][0], v[0][1]}, f32x2{v[0][2], v[0][3]}, f32x2{v[1][0], v[1][1]}, f32x2{v[1][2], v[1][3]}};
        gelu_poly2x4(g);
#pragma unroll
        for (int q = 0; q < 2; ++q)
          v[q][0] = g[2 * q].x, v[q][1] = g[2 * q].y, v[q][2] = g[2 * q + 1].x, v[q][3] = g[2 * q + 1].y;
      }
      if (kHalfOut || kStats) {  // f16 row: 8 consecutive features, one 16-byte store
        f16x8 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          h[r] = static_cast<half_t>(fminf(fmaxf(v[0][r], -65504.0f), 65504.0f));
          h[4 + r] = static_cast<half_t>(fminf(fmaxf(v[1][r], -65504.0f), 65504.0f));
        }
        if (VR_DIAG(32)) {
          if (h[0] == static_cast<half_t>(123.0f) && h[7] == static_cast<half_t>(77.0f)) Ch[0] = h[3];  // (keeps the arithmetic alive)
        } else if (XPOSE) {
          half_t* slot = xpose + ((8 * p2 + pc) & 3) * 512;
          *reinterpret_cast<f16x8*>(slot + xw) = h;
          asm volatile("" ::: "memory");
          if (xpend_pc >= 0) xflush();
          xh = *reinterpret_cast<const f16x8*>(slot + xr);
          asm volatile("" ::: "memory");
          xpend_pc = pc, xpend_p2 = p2;
        } else if (FULL)
          out_store(reinterpret_cast<f16x8*>(upiece(Ch, pc, col0 + 32 * p2) + loff), h);
        else if (ok)
          out_store(reinterpret_cast<f16x8*>(Ch + static_cast<int64_t>(grow) * N + c0), h);
      }
      if (!kHalfOut && FULL) {  // f32 row
        *reinterpret_cast<float4*>(upiece(C, pc, col0 + 32 * p2) + loff) = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
        *reinterpret_cast<float4*>(upiece(C, pc, col0 + 32 * p2 + 4) + loff) = make_float4(v[1][0], v[1][1], v[1][2], v[1][3]);
      } else if (!kHalfOut && ok) {
        *reinterpret_cast<float4*>(C + static_cast<int64_t>(grow) * N + c0) = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
        *reinterpret_cast<float4*>(C + static_cast<int64_t>(grow) * N + c0 + 4) = make_float4(v[1][0], v[1][1], v[1][2], v[1][3]);
      }
      if (kStats) {
        // this wave's 64 columns of every pre-LayerNorm row contribute a (sum, sum of squares) to the row's statistics:
        // 8 values per lane and fragment pair, the other 24 of the pair in the three other lane groups. The first
        // pair's sums wait in `wave_stat` (128 float2 of LDS that only this wave touches) for the second's — sixteen
        // registers per lane less than carrying them, which the branch-free form of this code did not have.
        float a1 = 0.0f, a2 = 0.0f;
        if (col_ok) {
          a1 = ((v[0][0] + v[0][1]) + (v[0][2] + v[0][3])) + ((v[1][0] + v[1][1]) + (v[1][2] + v[1][3]));
          a2 = ((v[0][0] * v[0][0] + v[0][1] * v[0][1]) + (v[0][2] * v[0][2] + v[0][3] * v[0][3])) +
               ((v[1][0] * v[1][0] + v[1][1] * v[1][1]) + (v[1][2] * v[1][2] + v[1][3] * v[1][3]));
        }
        a1 += __shfl_xor(a1, 16);
        a2 += __shfl_xor(a2, 16);
        a1 += __shfl_xor(a1, 32);
        a2 += __shfl_xor(a2, 32);
        if (p2 == 0) {
          if (fg == 0) wave_stat[16 * pc + tok] = make_float2(a1, a2);
        } else {
          const float2 first = wave_stat[16 * pc + tok];
          a1 += first.x;
          a2 += first.y;
          const int seg = col0 >> 6;
          if (FULL) {
            if (fg == 0)
              (reinterpret_cast<float2*>(Cl) + (static_cast<int64_t>(row0 + 16 * pc) * (N >> 6) + seg))[static_cast<uint32_t>(tok * (N >> 6))] =
                  make_float2(a1, a2);
          } else if (fg == 0 && grow < M && seg < (N >> 6)) {
            reinterpret_cast<float2*>(Cl)[static_cast<int64_t>(grow) * (N >> 6) + seg] = make_float2(a1, a2);
          }
        }
      }
      // Every piece stays a basic block of its own: in one long block the SLP vectoriser pairs up the arithmetic of
      // DIFFERENT pieces (v_pk_* over values of two pieces), all eight pieces are then in flight at once and a
      // thousand registers spill. The never-taken branch below (M is positive; the compiler cannot know) ends the
      // block without putting anything that counts in vmcnt on a side path, so the wait counts stay exact.
      if (FULL && M < 0) asm volatile("s_nop 0");
    }
  }
  if (XPOSE && (kHalfOut || kStats) && xpend_pc >= 0) xflush();
}


// Whole tiles of the f16-output projections without residual rows (QKV, FFN-up: four fifths of the bytes the GEMMs write),
// gemm_f16_pp_kernel only. Same arithmetic per element as direct_epilogue (same bits); what differs is the ORDER of things.
// In-kernel stamps (make diag, VR_GEMM_DIAG=128, scripts/pp_stamps.sh) showed direct_epilogue's whole-tile path at 9-10k
// cycles per tile for the QKV variant with its arithmetic switched off as well as on, against 5.4k for the arithmetic
// alone: every 16-token piece ended in an LDS round trip (lane exchange in front of the store, see direct_epilogue) and
// began with one (row statistics) that nothing covered — one basic block per piece, lgkmcnt(0) at every block entry.
// Here four pieces form a batch: their statistics are read together, their f16 rows are written to four LDS slots as they
// are computed (no wait), then read back transposed together and stored — two exposed LDS round trips per batch of four
// instead of eight, and the next batch's statistics are requested before the read-back.
template <int EPI>
__device__ __forceinline__ void f16_tile_epilogue(f32x4 (&acc)[8][4], int row0, int col0, int lane, half_t* __restrict__ Ch,
                                                  int M, int N, float unscale, const float* tile_const, int trow0, int tcol0,
                                                  half_t* xpose) {
  constexpr bool kFold = EPI == EPI_FOLD_F16 || EPI == EPI_FOLD_GELU;
  constexpr bool kGelu = EPI == EPI_BIAS_GELU || EPI == EPI_FOLD_GELU;
  asm volatile("" : "+v"(lane));  // (the lane-derived offsets below are made per tile, not carried through the K loop)
  const int tok = lane & 15, fg = lane >> 4;
  const float2* lds_stat = reinterpret_cast<const float2*>(tile_const + 768) + trow0;
  const int xw = tok * 32 + ((fg ^ (tok >> 2)) & 3) * 8;  // lane exchange: see direct_epilogue (XPOSE)
  const int xtr = lane >> 2, xtc = lane & 3;
  const int xr = xtr * 32 + ((xtc ^ (xtr >> 2)) & 3) * 8;
  const uint32_t xloff = static_cast<uint32_t>(xtr * N + 8 * xtc);
  float2 st[4] = {};
  if (kFold) {
#pragma unroll
    for (int i = 0; i < 4; ++i) st[i] = lds_stat[16 * i + tok];
  }
#pragma unroll
  for (int p2 = 0; p2 < 2; ++p2) {
    float4 b4[2], lg4[2] = {};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float* tc = tile_const + tcol0 + 32 * p2 + 8 * fg + 4 * q;
      b4[q] = *reinterpret_cast<const float4*>(tc);
      if (kFold) lg4[q] = *reinterpret_cast<const float4*>(tc + 256);
    }
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      float2 stc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) stc[i] = st[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pc = 4 * hb + i;
        // pairs of features on the packed-f32 instructions (element for element the operations of direct_epilogue)
        f32x2 g[4];  // pair 2 q + e: features 4 q + 2 e, + 1 of the lane's eight
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const f32x2 x = {acc[pc][2 * p2 + q][2 * e], acc[pc][2 * p2 + q][2 * e + 1]};
            const f32x2 bb = e == 0 ? f32x2{b4[q].x, b4[q].y} : f32x2{b4[q].z, b4[q].w};
            if (kFold) {
              const f32x2 gg = e == 0 ? f32x2{lg4[q].x, lg4[q].y} : f32x2{lg4[q].z, lg4[q].w};
              const f32x2 mean = {stc[i].x, stc[i].x}, inv = {stc[i].y, stc[i].y}, un = {unscale, unscale};
              g[2 * q + e] = __builtin_elementwise_fma(__builtin_elementwise_fma(x, un, -(mean * gg)), inv, bb);
            } else {
              g[2 * q + e] = x * unscale + bb;
            }
          }
        if (kGelu) gelu_poly2x4(g);
        f16x8 h;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            h[4 * q + 2 * e] = static_cast<half_t>(g[2 * q + e].x);  // (saturating: MODE.FP16_OVFL, set by the kernel)
            h[4 * q + 2 * e + 1] = static_cast<half_t>(g[2 * q + e].y);
          }
        *reinterpret_cast<f16x8*>(xpose + i * 512 + xw) = h;
        // (one basic block per piece, as in direct_epilogue: in one long block the SLP vectoriser pairs arithmetic of
        // different pieces and a thousand registers spill)
        if (M < 0) asm volatile("s_nop 0");
      }
      asm volatile("" ::: "memory");
      // the next batch's statistics are requested before this one's rows are read back
      const int nb = 2 * p2 + hb + 1;  // batches 0..3: (p2, hb); the statistics depend on hb only
      if (kFold && nb < 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) st[i] = lds_stat[16 * (4 * (nb & 1) + i) + tok];
      }
      f16x8 xh[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) xh[i] = *reinterpret_cast<const f16x8*>(xpose + i * 512 + xr);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        out_store(reinterpret_cast<f16x8*>(Ch + (static_cast<int64_t>(row0 + 16 * (4 * hb + i)) * N + col0 + 32 * p2) + xloff), xh[i]);
      asm volatile("" ::: "memory");
    }
  }
}

// Four values summed over the four 16-lane rows of a wave at once: afterwards the lanes of row r hold the total of value r,
// ((r0 + r1) + (r2 + r3)) — the bits of a += __shfl_xor(a, 16); a += __shfl_xor(a, 32) — in three lane swaps and three adds
// (v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second, v_permlane32_swap the
// upper half of the first with the lower half of the second: each swap + add halves two values at a time), on the vector
// ALU instead of eight ds_bpermute round trips through the LDS queue.
__device__ __forceinline__ float rows_sum4x4(float v0, float v1, float v2, float v3) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
  const unsigned a0 = a[0], a1 = a[1];
  const float s01 = __uint_as_float(a0) + __uint_as_float(a1);  // rows: v0 (r0 + r1), v1 (r0 + r1), v0 (r2 + r3), v1 (r2 + r3)
  const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v2), __float_as_uint(v3), false, false);
  const unsigned b0 = b[0], b1 = b[1];
  const float s23 = __uint_as_float(b0) + __uint_as_float(b1);
  const auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(s01), __float_as_uint(s23), false, false);
  const unsigned c0 = c[0], c1 = c[1];
  return __uint_as_float(c0) + __uint_as_float(c1);  // row r: value r, (r0 + r1) + (r2 + r3)
}

// Whole tiles of the two projections that add the residual stream (attention output, FFN-down; EPI_RLS_R16_O16: f16
// residual rows in, LayerNorm of them applied here, f16 pre-LayerNorm rows and per-row partial statistics out),
// gemm_f16_pp_kernel only. Same arithmetic per element and the same order of every sum as direct_epilogue (same bits).
// In the stamps direct_epilogue's predicated path took 25-27k cycles per tile for this variant — as long as the whole K loop
// of the attention-output projection — for ~6k cycles of arithmetic: sixteen 16-byte loads per wave whose quads address four
// different rows (64 requests per instruction, like the stores the lane exchange fixed), each waited for with the stores
// of the piece before it in the same queue, bias / gain / shift / row statistics from global memory, two ds_bpermute
// round trips per piece for the statistics. Here
//   * the residual rows arrive by direct-to-LDS loads — no registers, quad-contiguous requests (lane L: token L >> 2, piece
//     (L & 3) ^ f(token) of the 64-byte half row: the source-side form of the exchange layout), a batch of four pieces
//     AHEAD of the batch being computed, waited for with a counted vmcnt that leaves the previous batch's stores in flight;
//   * a piece reads its residual values out of its slot, and writes its f16 result rows into the same slot (LDS executes a
//     wave's instructions in order); four slots are read back transposed and stored, as in f16_tile_epilogue;
//   * the tile's constants and row statistics come from the LDS copy the K loop fetched; the sums over the four lane groups
//     are vector-ALU lane swaps (rows_sum4); the first fragment pair's sums wait in registers the dying accumulators free.
// xpose: 8 KiB per wave (two batches of four 1-KiB slots) in the stage buffer the K loop has left.
__device__ __forceinline__ void rls16_tile_epilogue(f32x4 (&acc)[8][4], int row0, int col0, int lane, const half_t* __restrict__ R16,
                                                    half_t* __restrict__ Ch, half_t* __restrict__ Cl, int M, int N, float unscale,
                                                    const float* tile_const, int trow0, int tcol0, half_t* xpose, half_t* xspare) {
  asm volatile("" : "+v"(lane));
  const int tok = lane & 15, fg = lane >> 4;
  const float2* lds_stat = reinterpret_cast<const float2*>(tile_const + 768) + trow0;
  const int xw = tok * 32 + ((fg ^ (tok >> 2)) & 3) * 8;
  const int xtr = lane >> 2, xtc = lane & 3;
  const int xr = xtr * 32 + ((xtc ^ (xtr >> 2)) & 3) * 8;
  const uint32_t xloff = static_cast<uint32_t>(xtr * N + 8 * xtc);
  // source of the direct-to-LDS loads: token xtr of the piece, 16-byte piece xtc ^ f(xtr) (it lands at position xtc)
  const uint32_t rloff = static_cast<uint32_t>(xtr * N + 8 * ((xtc ^ (xtr >> 2)) & 3));
  // Groups of two pieces, g = 0..7: fragment pair p2 = g >> 2, pieces 2 (g & 3), + 1. Five slot pairs (four in the stage
  // buffer the K loop has left, one in the LDS behind the tile constants) carry the group being computed and the four
  // behind it: a group's residual rows are requested FOUR groups (eight pieces of arithmetic) before they are read — with
  // batches of four pieces and one batch of lead every batch waited for memory (stamps: 17-19k cycles per tile against 9k of
  // vector instructions).
  auto pair_of_group = [&](int g) { return (g % 5) < 4 ? xpose + (g % 5) * 1024 : xspare; };
  auto request = [&](int g) {
    half_t* buf = pair_of_group(g);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      glds16(R16 + (static_cast<int64_t>(row0 + 16 * (2 * (g & 3) + i)) * N + col0 + 32 * (g >> 2)) + rloff, buf + i * 512);
  };
  float sk[4];  // the first fragment pair's sums, one register per group: lane row fg holds (a1, a2) of the group's first piece (fg = 0, 1) and of its second (fg = 2, 3)
#pragma unroll
  for (int g = 0; g < 5; ++g) request(g);
  float4 b4[2], lg4[2], lb4[2];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int p2 = g >> 2;
    half_t* buf = pair_of_group(g);
    asm volatile("" ::: "memory");
    // operations younger than this group's two loads (the loads of the groups behind it, the stores of the groups before
    // it): what the counted wait leaves in flight
    // (a group of the first fragment pair issues two stores, one of the second three — its statistics; groups 0-2 are followed
    // by the two loads of the group five behind)
    if (g == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // D1-D4
    else if (g == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // D2-D4, S0, D5
    else if (g == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // D3, D4, S0, D5, S1, D6
    else if (g <= 4) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");  // g = 3: D4, (S, D) x 3; g = 4: (S, D) x 3, S3
    else if (g == 5) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");  // S1, D6, S2, D7, S3, S4 (3)
    else if (g == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // S2, D7, S3, S4 (3), S5 (3)
    else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");              // S3, S4, S5, S6 (3 each but S3)
    if ((g & 3) == 0) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float* tc = tile_const + tcol0 + 32 * p2 + 8 * fg + 4 * q;
        b4[q] = *reinterpret_cast<const float4*>(tc);
        lg4[q] = *reinterpret_cast<const float4*>(tc + 256);
        lb4[q] = *reinterpret_cast<const float4*>(tc + 512);
      }
    }
    float2 stc[2];
    f16x8 rh2[2];  // the group's residual values and row statistics, read together
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      stc[i] = lds_stat[16 * (2 * (g & 3) + i) + tok];
      rh2[i] = *reinterpret_cast<const f16x8*>(buf + i * 512 + xw);
    }
    asm volatile("" ::: "memory");
    float pa1[2], pa2[2];  // the two pieces' sums over this lane's eight features
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pc = 2 * (g & 3) + i;
      const f16x8 rh = rh2[i];
      // pairs of features on the packed-f32 instructions; element for element the operations (and their order) of
      // direct_epilogue: x = acc * unscale + bias; x += ((r - mean) * inv) * gain + shift
      // (stage by stage across the four pairs: a dependent chain of packed instructions issues every other slot)
      f32x2 g2[4], t[4];  // pair k = 2 q + e: features 4 q + 2 e, + 1 of the lane's eight
      const f32x2 mean = {stc[i].x, stc[i].x}, inv = {stc[i].y, stc[i].y};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        g2[k] = f32x2{acc[pc][2 * p2 + (k >> 1)][2 * (k & 1)], acc[pc][2 * p2 + (k >> 1)][2 * (k & 1) + 1]} * unscale;
        t[k] = f32x2{static_cast<float>(rh[4 * (k >> 1) + 2 * (k & 1)]), static_cast<float>(rh[4 * (k >> 1) + 2 * (k & 1) + 1])} - mean;
      }
      auto pair_of = [](const float4& c, int e) { return e == 0 ? f32x2{c.x, c.y} : f32x2{c.z, c.w}; };
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        g2[k] = g2[k] + pair_of(b4[k >> 1], k & 1);
        t[k] = t[k] * inv;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = t[k] * pair_of(lg4[k >> 1], k & 1);
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = t[k] + pair_of(lb4[k >> 1], k & 1);
#pragma unroll
      for (int k = 0; k < 4; ++k) g2[k] = g2[k] + t[k];
      f16x8 h;
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          h[4 * q + 2 * e] = static_cast<half_t>(g2[2 * q + e].x);  // (saturating: MODE.FP16_OVFL, set by the kernel)
          h[4 * q + 2 * e + 1] = static_cast<half_t>(g2[2 * q + e].y);
        }
      float v[2][4];
#pragma unroll
      for (int q = 0; q < 2; ++q) v[q][0] = g2[2 * q].x, v[q][1] = g2[2 * q].y, v[q][2] = g2[2 * q + 1].x, v[q][3] = g2[2 * q + 1].y;
      *reinterpret_cast<f16x8*>(buf + i * 512 + xw) = h;  // (behind the read of the same 16 bytes)
      pa1[i] = ((v[0][0] + v[0][1]) + (v[0][2] + v[0][3])) + ((v[1][0] + v[1][1]) + (v[1][2] + v[1][3]));
      pa2[i] = ((v[0][0] * v[0][0] + v[0][1] * v[0][1]) + (v[0][2] * v[0][2] + v[0][3] * v[0][3])) +
               ((v[1][0] * v[1][0] + v[1][1] * v[1][1]) + (v[1][2] * v[1][2] + v[1][3] * v[1][3]));
      if (M < 0) asm volatile("s_nop 0");  // (one basic block per piece: see direct_epilogue)
    }
    {
      // sums over the four lane groups, the group's four values at once: lane row 0 / 1 ends up with (sum, sum of squares)
      // of the first piece's token `tok`, row 2 / 3 with the second piece's
      float tot = rows_sum4x4(pa1[0], pa2[0], pa1[1], pa2[1]);
      if (p2 == 0) {
        sk[g & 3] = tot;
      } else {
        tot += sk[g & 3];
        const int pc = 2 * (g & 3) + (fg >> 1);
        (reinterpret_cast<float*>(Cl) + 2 * (static_cast<int64_t>(row0 + 16 * pc) * (N >> 6) + (col0 >> 6)))[static_cast<uint32_t>(2 * tok * (N >> 6) + (fg & 1))] = tot;
      }
    }
    asm volatile("" ::: "memory");
    f16x8 xh[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) xh[i] = *reinterpret_cast<const f16x8*>(buf + i * 512 + xr);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      out_store(reinterpret_cast<f16x8*>(Ch + (static_cast<int64_t>(row0 + 16 * (2 * (g & 3) + i)) * N + col0 + 32 * p2) + xloff), xh[i]);
    asm volatile("" ::: "memory");
    if (g + 5 < 8) request(g + 5);  // (into the slot pair this group has just read back)
  }
}

// ---- the 256x256 f16 product with a ping-pong main loop ----------------------------------------------
//
// Same tile, LDS image, swizzle, persistent tile walk and epilogues as gemm_f16x3_256_kernel<EPI, 1>; what
// changes is WHEN things happen inside a 64-deep K-tile. There every wave read fragments, multiplied and
// waited for the next K-tile's loads in step with its SIMD partner (one barrier and one vmcnt(0) per K-tile):
// while both read LDS or waited, the matrix pipe of the SIMD idled (64 % MFMA issue in the main loop). Here
//   * a K-tile is four PHASES, one 64x32 quadrant of the wave's 128x64 tile each (16 MFMAs over both 32-deep
//     steps): a READ segment (the phase's fragment reads + two direct-to-LDS loads of the next K-tile) and
//     an MFMA segment, each closed by a raw s_barrier;
//   * waves 4-7 (the second wave of every SIMD) run one segment behind waves 0-3, so at any time one wave of
//     a SIMD multiplies while its partner reads and loads: the partner's LDS latency, load issue and barrier
//     wait sit under 256 cycles of MFMAs instead of beside them;
//   * the next K-tile arrives in four half-tiles issued in the order they are needed (rows of quadrant
//     rows 0-63 of A, quadrant columns 0-31 of W, columns 32-63, rows 64-127), each waited for with a COUNTED
//     vmcnt two to three phases after its issue (never vmcnt(0) in the loop) — two half-tiles stay in flight
//     across every barrier. A wait sits in the READ segment of the phase BEFORE the one that reads the data, so
//     that every wave's wait and one more barrier lie between a load and any wave's read of it.
//   * the MFMAs take the WEIGHT fragment as their A operand, which leaves every lane with four consecutive
//     output features of one token: the epilogue stores straight from the accumulators (no LDS transpose).
// Accumulation order per output element is unchanged (k ascending), so the result is bit-identical to
// gemm_f16x3_256_kernel<EPI, 1>. Needs an even number of K-tiles (K % 128 == 0) and N % 8 == 0: every supported width.
template <int EPI>
__global__ __launch_bounds__(512) void gemm_f16_pp_kernel(
    const half_t* __restrict__ Ah, const half_t* __restrict__ Wh, const float* __restrict__ bias,
    const float* __restrict__ R, float* __restrict__ C, half_t* __restrict__ Ch, half_t* __restrict__ Cl, int M,
    int N, int K, float unscale, const float2* __restrict__ ln_stat, const float* __restrict__ ln_g,
    const float* __restrict__ ln_b, int stagger_sleeps, int group_m) {
  __shared__ half_t lds[2 * kStageHalfs + 8 * kWaveStatHalfs + kTileConstHalfs + 8 * 1024];  // the only LDS object (see gemm_f16x3_256_kernel)
  const int tiles_n = (N + GBN - 1) / GBN;
  const int tiles_m = (M + GBM - 1) / GBM;
  const int total = tiles_m * tiles_n;
  const int G = gridDim.x;
  const int local = (G % 8 == 0) ? (static_cast<int>(blockIdx.x) % 8) * (G / 8) + static_cast<int>(blockIdx.x) / 8
                                 : static_cast<int>(blockIdx.x);
  auto coords = [&](int t, int& bm_, int& bn_) {
    const int per_group = group_m * tiles_n;
    const int group = t / per_group;
    const int within = t - group * per_group;
    const int gm = min(group_m, tiles_m - group * group_m);
    bm_ = (group * group_m + within % gm) * GBM;
    bn_ = (within / gm) * GBN;
  };
  int tile = local;
  if (tile >= total) return;  // block-uniform
  // MODE.FP16_OVFL: f32 -> f16 conversions saturate at +-65504 instead of overflowing to infinity — for every finite input the
  // bits of clamp-then-convert (scripts/probe_fp16_ovfl.hip: 100,000 values, none differs; an infinite input stays infinite,
  // where the clamp makes it 65504), without the v_med3_f32 per element the whole-tile epilogues spent on the clamp
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  int bm, bn;
  coords(tile, bm, bn);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  float2* wave_stat = reinterpret_cast<float2*>(lds + 2 * kStageHalfs + wave * kWaveStatHalfs);
  // the epilogue's lane exchange (direct_epilogue: xpose) uses stage buffer 1: the K loop ends on it (nk is even), the next
  // tile's K-tile 0 is in buffer 0, and nothing loads into buffer 1 before the barrier that follows the epilogue
  half_t* xpose = lds + kStageHalfs + wave * 4096;  // 8 KiB per wave
  half_t* xspare = lds + 2 * kStageHalfs + 8 * kWaveStatHalfs + kTileConstHalfs + wave * 1024;  // + 2 KiB per wave (rls16_tile_epilogue)
  // the tile's epilogue constants (direct_epilogue: tile_const), fetched by waves 0-4 at the start of the tile's K loop —
  // one direct-to-LDS load each — so that the epilogue starts on LDS reads instead of three rounds of global-load latency
  half_t* tile_const_h = lds + 2 * kStageHalfs + 8 * kWaveStatHalfs;
  constexpr bool kUsesGain = EPI == EPI_FOLD_F16 || EPI == EPI_FOLD_GELU || EPI == EPI_BIAS_RESIDUAL_LN ||
                             EPI == EPI_BIAS_RESIDUAL_LN_STATS || EPI == EPI_RLS_R32_O16 || EPI == EPI_RLS_R16_O16 ||
                             EPI == EPI_RLS_R16_O32;
  constexpr bool kUsesShift = kUsesGain && EPI != EPI_FOLD_F16 && EPI != EPI_FOLD_GELU;
  // (the branch-free form of the epilogues that also read residual rows needs ~30 registers more than this kernel
  // has left beside its staging state; those keep the predicated form and their global loads)
  constexpr bool kBranchFree = EPI != EPI_BIAS_RESIDUAL_LN && EPI != EPI_BIAS_RESIDUAL_LN_STATS && EPI != EPI_RLS_R32_O16 &&
                               EPI != EPI_RLS_R16_O16 && EPI != EPI_RLS_R16_O32;
  constexpr bool kRlsTile = kXposeStores && EPI == EPI_RLS_R16_O16;  // whole tiles: rls16_tile_epilogue
  auto issue_tile_consts = [&](int bm_, int bn_) {
    // edge tiles read them from global memory (predicated epilogue) — except a last column tile that ends on a 64-column
    // boundary (N = 384, 1152: the waves behind the edge have nothing to store, the others run the whole-tile epilogue)
    if (!(kBranchFree || kRlsTile) || bm_ + GBM > M || (bn_ + GBN > N && N % 64 != 0)) return;
    // (the lane offset is made opaque per call: left visible, the compiler keeps five per-lane 64-bit source addresses alive
    // across the whole kernel — ten registers the epilogue does not have)
    int l8 = lane * 8;  // 16 bytes per lane, in halfs
    asm volatile("" : "+v"(l8));
    const int c8 = min(2 * bn_ + l8, 2 * N - 8);  // (columns behind the edge re-read the last four: never used)
    if (wave == 0) glds16(reinterpret_cast<const half_t*>(bias) + c8, tile_const_h);
    if (wave == 1 && kUsesGain) glds16(reinterpret_cast<const half_t*>(ln_g) + c8, tile_const_h + 512);
    if (wave == 2 && kUsesShift) glds16(reinterpret_cast<const half_t*>(ln_b) + c8, tile_const_h + 1024);
    if (wave == 3 && kUsesGain) glds16(reinterpret_cast<const half_t*>(ln_stat + bm_) + l8, tile_const_h + 1536);
    if (wave == 4 && kUsesGain) glds16(reinterpret_cast<const half_t*>(ln_stat + bm_ + 128) + l8, tile_const_h + 2048);
  };
  // Phase diversity: every block's tile takes the same time, so without this all 256 CUs reach their
  // epilogues together — a burst of stores at the chip's store bandwidth with every matrix pipe idle,
  // followed by a main loop with the store path idle. Block `local` (of G) starts local / G of
  // `stagger_sleeps` x 64 cycles late, which spreads the epilogues over the tile period. The blocks with
  // the highest index — the ones that sleep longest — are also the ones that get one tile FEWER when the
  // tile count is not a multiple of G, so most of the delay falls into the last, partly filled round.
  for (int d = (static_cast<int64_t>(stagger_sleeps) * local) / G; d > 0; d -= 16) __builtin_amdgcn_s_sleep(16);

  // staging: one direct-to-LDS load moves 8 rows x 128 B; lane -> row (lane >> 3) of the 8, LDS chunk
  // (lane & 7), source chunk (lane & 7) ^ ((row >> 1) & 7). Wave w stages A rows 8w + {0, 128} (the rows
  // of quadrant rows 0-63 of both wave rows: "A lo"), 8w + {64, 192} ("A hi"), and W rows
  // 64 (w >> 2) + 8 (w & 3) + {0, 128} (quadrant columns 0-31 of all four wave columns: "W lo"), + {32, 160}
  // ("W hi"): every wave issues two loads per half-tile, so one vmcnt count serves all waves.
  // (the per-lane staging and fragment offsets are derived again after every epilogue — derive_lane_state — so that they
  // do not hold registers while the epilogue runs: it is the kernel's register peak)
  int schunk, ra, rwp, pa, pw, fk0, fk1;
  auto derive_lane_state = [&](int lane_) {
    const int srow = lane_ >> 3;
    schunk = ((lane_ & 7) ^ ((4 * (wave & 1) + (lane_ >> 4)) & 7)) * 8;
    ra = wave * 8 + srow;
    // image rows 64 (w >> 2) + 8 (w & 3) + srow (+ 0, 128, 32, 160): q = (w >> 1) & 1, r = 8 (w & 1) + srow of the
    // permutation above, i.e. weight row 64 (w >> 2) + 16 (w & 1) + 8 (srow >> 2) + 4 ((w >> 1) & 1) + (srow & 3)
    rwp = 64 * (wave >> 2) + 16 * (wave & 1) + 8 * (srow >> 2) + 4 * ((wave >> 1) & 1) + (srow & 3);
    // fragments (v_mfma_f32_16x16x32_f16): lane l supplies row (l & 15), k = 8 (l >> 4) + j of a 32-deep step;
    // step kk of the K-tile is chunk 4 kk + (l >> 4) of the row, stored at chunk ^ ((row >> 1) & 7)
    const int frow = lane_ & 15;
    const int fsw = (frow >> 1) & 7;
    const int fq = lane_ >> 4;
    pa = ((wave >> 2) * 128 + frow) * 64;
    pw = 256 * 64 + ((wave & 3) * 64 + frow) * 64;
    fk0 = (fq ^ fsw) * 8, fk1 = ((4 + fq) ^ fsw) * 8;
  };
  derive_lane_state(lane);
  const half_t *g_a0, *g_a1, *g_a2, *g_a3, *g_w0, *g_w1, *g_w2, *g_w3;
  auto set_ptrs = [&](int bm_, int bn_) {
    g_a0 = Ah + static_cast<int64_t>(min(bm_ + ra, M - 1)) * K + schunk;        // lo
    g_a1 = Ah + static_cast<int64_t>(min(bm_ + ra + 128, M - 1)) * K + schunk;  // lo
    g_a2 = Ah + static_cast<int64_t>(min(bm_ + ra + 64, M - 1)) * K + schunk;   // hi
    g_a3 = Ah + static_cast<int64_t>(min(bm_ + ra + 192, M - 1)) * K + schunk;  // hi
    // W rows are staged PERMUTED inside every block of 32: image row 32 P + 16 q + r holds weight row
    // 32 P + 8 (r >> 2) + 4 q + (r & 3) — see the epilogue: a lane then owns 8 consecutive output features
    g_w0 = Wh + static_cast<int64_t>(min(bn_ + rwp, N - 1)) * K + schunk;        // lo
    g_w1 = Wh + static_cast<int64_t>(min(bn_ + rwp + 128, N - 1)) * K + schunk;  // lo
    g_w2 = Wh + static_cast<int64_t>(min(bn_ + rwp + 32, N - 1)) * K + schunk;   // hi
    g_w3 = Wh + static_cast<int64_t>(min(bn_ + rwp + 160, N - 1)) * K + schunk;  // hi
  };
  set_ptrs(bm, bn);
  // wave-uniform LDS destinations (halfs, inside a stage buffer)
  const int da0 = wave * 8 * 64, da1 = da0 + 128 * 64, da2 = da0 + 64 * 64, da3 = da0 + 192 * 64;
  const int dw0 = 256 * 64 + (64 * (wave >> 2) + 8 * (wave & 3)) * 64, dw1 = dw0 + 128 * 64, dw2 = dw0 + 32 * 64,
            dw3 = dw0 + 160 * 64;

  f32x4 acc[8][4];

  const int nk = K / 64;  // even
#ifdef VR_GEMM_DIAG_BUILD
  const int diag_bits = __builtin_amdgcn_readfirstlane(g_gemm_diag);
#endif
#ifdef VR_GEMM_HAS_STAMPS
  const bool stamp_on = static_cast<int>(blockIdx.x) == G / 2;
  int tile_seq = 0;
#endif
#define VR_PP_BARRIER()                                      \
  do {                                                       \
    __builtin_amdgcn_sched_barrier(0);                       \
    if (!VR_DIAG(16)) __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);                       \
  } while (0)
#define VR_PP_HARD_BARRIER()                 \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    __builtin_amdgcn_s_barrier();            \
    __builtin_amdgcn_sched_barrier(0);       \
  } while (0)
#define VR_PP_VMCNT4() asm volatile("s_waitcnt vmcnt(4)" ::: "memory")
  // Epilogues that issue a known number of stores and no global loads (whole tiles of the branch-free f16-output variants:
  // constants and row statistics come from LDS): the last two half-tiles of the next tile's K-tile 0 (W hi, A hi, requested
  // in the last two phases of the K loop) are NOT waited for before the epilogue — their latency runs under its arithmetic —
  // but by counted waits in the first two phases of the next K loop that let exactly the younger operations (A hi where
  // it applies, the kEpiStores stores, the phase's own loads) stay in flight. Every other epilogue drains vmcnt first.
  constexpr bool kTileEpilogue = kXposeStores && (EPI == EPI_FOLD_F16 || EPI == EPI_FOLD_GELU || EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU);
  constexpr int kEpiStores = (kXposeStores && (EPI == EPI_FOLD_F16 || EPI == EPI_FOLD_GELU || EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU)) ? 16 : -1;
  bool counted_tail = false;  // block-uniform: the previous tile's epilogue left W hi / A hi to the counted waits
#define VR_PP_VMCNT_TAIL() asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kEpiStores > 0 ? kEpiStores + 4 : 0) : "memory")

  // prologue: K-tile 0 of the first tile, all eight pieces, into buffer 0
  glds16(g_a0, lds + da0);
  glds16(g_a1, lds + da1);
  glds16(g_w0, lds + dw0);
  glds16(g_w1, lds + dw1);
  glds16(g_w2, lds + dw2);
  glds16(g_w3, lds + dw3);
  glds16(g_a2, lds + da2);
  glds16(g_a3, lds + da3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  VR_PP_BARRIER();
  if (wm == 1) VR_PP_BARRIER();  // waves 4-7 run one segment behind from here on

  while (true) {
    const int next = tile + G;
    const bool has_next = next < total;
    int nbm = bm, nbn = bn;
    VR_PP_STAMP(0);

    // one K-tile out of stage buffer B; the next K-tile (or K-tile 0 of the next tile, or — at the very
    // end — a harmless re-load) goes into buffer B ^ 1
    // FIRST (a tile's K-tile 0): the first MFMA into each accumulator takes a zero C operand instead of the accumulator, so
    // the accumulators are never cleared (128 v_mov per wave and tile, all eight waves at once, nothing under them)
    auto ktile = [&](auto bsel, int kt, auto first_sel) {
      constexpr int B = decltype(bsel)::value;
      constexpr bool FIRST = decltype(first_sel)::value;
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      const half_t* st = lds + B * kStageHalfs;
      half_t* nd = lds + (B ^ 1) * kStageHalfs;
      const bool lastk = kt == nk - 1;
      if (lastk && has_next) set_ptrs(nbm, nbn);
      const int koff = lastk ? 0 : (kt + 1) * 64;
      f16x8 af[4][2], bf[2][2];
#ifdef VR_GEMM_DIAG_BUILD
      for (int i = 0; i < 4; ++i) af[i][0] = af[i][1] = f16x8{1, 1, 1, 1, 1, 1, 1, 1};
      for (int j = 0; j < 2; ++j) bf[j][0] = bf[j][1] = f16x8{1, 1, 1, 1, 1, 1, 1, 1};
#endif
      // ---- phase 0: quadrant rows 0-63 x columns 0-31 ------------------------------------------------
      if (kt == 0) issue_tile_consts(bm, bn);  // (older than every load the counted waits below reason about)
      if (!VR_DIAG(1)) glds16(g_a0 + koff, nd + da0);
      if (!VR_DIAG(1)) glds16(g_a1 + koff, nd + da1);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (!VR_DIAG(8)) bf[j][0] = *reinterpret_cast<const f16x8*>(st + pw + j * 16 * 64 + fk0);
        if (!VR_DIAG(8)) bf[j][1] = *reinterpret_cast<const f16x8*>(st + pw + j * 16 * 64 + fk1);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!VR_DIAG(8)) af[i][0] = *reinterpret_cast<const f16x8*>(st + pa + i * 16 * 64 + fk0);
        if (!VR_DIAG(8)) af[i][1] = *reinterpret_cast<const f16x8*>(st + pa + i * 16 * 64 + fk1);
      }
      if (kt != 0) VR_PP_VMCNT4();  // W hi of this K-tile has landed (K-tile 0: waited for whole, or by the counted tail)
      else if (kEpiStores > 0 && counted_tail) VR_PP_VMCNT_TAIL();  // younger: A hi, the stores, this phase's two loads
      VR_PP_BARRIER();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (!VR_DIAG(4)) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j][kk], af[i][kk], (FIRST && kk == 0) ? zero4 : acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      VR_PP_BARRIER();
      // ---- phase 1: rows 0-63 x columns 32-63 ---------------------------------------------------------
      if (!VR_DIAG(1)) glds16(g_w0 + koff, nd + dw0);
      if (!VR_DIAG(1)) glds16(g_w1 + koff, nd + dw1);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (!VR_DIAG(8)) bf[j][0] = *reinterpret_cast<const f16x8*>(st + pw + (2 + j) * 16 * 64 + fk0);
        if (!VR_DIAG(8)) bf[j][1] = *reinterpret_cast<const f16x8*>(st + pw + (2 + j) * 16 * 64 + fk1);
      }
      if (kt != 0) VR_PP_VMCNT4();  // A hi of this K-tile has landed
      else if (kEpiStores > 0 && counted_tail) VR_PP_VMCNT_TAIL();  // younger: the stores, the loads of phases 0 and 1
      VR_PP_BARRIER();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (!VR_DIAG(4)) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j][kk], af[i][kk], (FIRST && kk == 0) ? zero4 : acc[i][2 + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      VR_PP_BARRIER();
      // ---- phase 2: rows 64-127 x columns 32-63 -------------------------------------------------------
      if (!VR_DIAG(1)) glds16(g_w2 + koff, nd + dw2);
      if (!VR_DIAG(1)) glds16(g_w3 + koff, nd + dw3);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!VR_DIAG(8)) af[i][0] = *reinterpret_cast<const f16x8*>(st + pa + (4 + i) * 16 * 64 + fk0);
        if (!VR_DIAG(8)) af[i][1] = *reinterpret_cast<const f16x8*>(st + pa + (4 + i) * 16 * 64 + fk1);
      }
      VR_PP_BARRIER();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (!VR_DIAG(4)) acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j][kk], af[i][kk], (FIRST && kk == 0) ? zero4 : acc[4 + i][2 + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      VR_PP_BARRIER();
      // ---- phase 3: rows 64-127 x columns 0-31 (W fragments read again: holding them costs 16 VGPRs) ----
      if (!VR_DIAG(1)) glds16(g_a2 + koff, nd + da2);
      if (!VR_DIAG(1)) glds16(g_a3 + koff, nd + da3);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (!VR_DIAG(8)) bf[j][0] = *reinterpret_cast<const f16x8*>(st + pw + j * 16 * 64 + fk0);
        if (!VR_DIAG(8)) bf[j][1] = *reinterpret_cast<const f16x8*>(st + pw + j * 16 * 64 + fk1);
      }
      VR_PP_VMCNT4();  // A lo and W lo of the next K-tile have landed; W hi and A hi stay in flight
      VR_PP_BARRIER();
      if (kt == 0) VR_PP_STAMP(1);
      if (kt == 1) VR_PP_STAMP(6);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (!VR_DIAG(4)) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j][kk], af[i][kk], (FIRST && kk == 0) ? zero4 : acc[4 + i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      VR_PP_BARRIER();
    };
    ktile(std::integral_constant<int, 0>{}, 0, std::true_type{});
    // (the next tile's coordinates — three integer divisions — are worked out here, under the K loop, not between the
    // epilogue and the first K-tile where every wave of the block would wait for them)
    if (has_next) coords(next, nbm, nbn);
    ktile(std::integral_constant<int, 1>{}, 1, std::false_type{});
#pragma clang loop unroll(disable)
    for (int kt = 2; kt < nk; kt += 2) {
      ktile(std::integral_constant<int, 0>{}, kt, std::false_type{});
      ktile(std::integral_constant<int, 1>{}, kt + 1, std::false_type{});
    }

    // waves 0-3 are a segment ahead: they wait here for waves 4-7's last MFMA segment, so that all eight waves
    // run the epilogue TOGETHER (one half after the other costs 13 %: each wave's epilogue is bound by the latency of
    // its own loads, and the two waves of a SIMD hide each other's — profiles/r02_gemm_experiments.md §4)
    VR_PP_STAMP(2);
    if (wm == 0) VR_PP_BARRIER();
    // Epilogue, straight from the accumulators (direct_epilogue).
    // W hi / A hi of the next tile's K-tile 0 may still be in flight: waited for HERE, before this wave's
    // stores queue up behind them (vmcnt retires in order: a counted wait in the next main loop would otherwise
    // wait for the stores too).
    // whole tile, or a last column tile that ends on a 64-column boundary: every wave's 64 columns are in or out as a whole
    const bool whole_tile = bm + GBM <= M && (bn + GBN <= N || N % 64 == 0);  // block-uniform
    const bool no_columns = bn + wn * 64 >= N;                                // wave-uniform: nothing to store
    counted_tail = kEpiStores > 0 && whole_tile && !VR_DIAG(2) && !VR_DIAG(32);
    if (!counted_tail) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VR_PP_STAMP(3);
    if (!VR_DIAG(2)) {
    if (whole_tile && no_columns)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the counted waits of the next K loop assume this wave's stores)
    else if (kTileEpilogue && whole_tile && !VR_DIAG(32) && !VR_DIAG(64))
      f16_tile_epilogue<EPI>(acc, bm + wm * 128, bn + wn * 64, lane, Ch, M, N, unscale, reinterpret_cast<const float*>(tile_const_h),
                             wm * 128, wn * 64, xpose);
    else if (kRlsTile && whole_tile && !VR_DIAG(32) && !VR_DIAG(64))
      rls16_tile_epilogue(acc, bm + wm * 128, bn + wn * 64, lane, reinterpret_cast<const half_t*>(R), Ch, Cl, M, N, unscale,
                          reinterpret_cast<const float*>(tile_const_h), wm * 128, wn * 64, xpose, xspare);
    else if (kBranchFree && whole_tile)
      direct_epilogue<EPI, true, kXposeStores>(acc, bm + wm * 128, bn + wn * 64, lane, bias, R, C, Ch, Cl, M, N, unscale, ln_stat, ln_g, ln_b, wave_stat,
                                 reinterpret_cast<const float*>(tile_const_h), wm * 128, wn * 64, xpose);
    else
      direct_epilogue<EPI, false, kXposeStores && !kRlsTile>(acc, bm + wm * 128, bn + wn * 64, lane, bias, R, C, Ch, Cl, M, N, unscale, ln_stat, ln_g, ln_b, wave_stat,
                                  nullptr, 0, 0, xpose);
    } else if (acc[0][0][0] == 12345.678f && acc[7][3][3] == 1.0f) {  // (diagnostic) keep the accumulators alive
      float t = 0.0f;
      for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      C[0] = t;
    }
    VR_PP_STAMP(4);
#ifdef VR_GEMM_HAS_STAMPS
    ++tile_seq;
#endif
    if (!has_next) break;
    tile = next;
    bm = nbm;
    bn = nbn;
    // the eight staging pointers (16 registers) are dead through the epilogue: they are derived again here, from tile
    // coordinates the compiler cannot match with the ones the last K-tile used (else it keeps the old values alive)
    asm volatile("" : "+s"(bm), "+s"(bn));
    {
      int lane_again = lane;
      asm volatile("" : "+v"(lane_again));
      derive_lane_state(lane_again);
    }
    set_ptrs(bm, bn);
    VR_PP_BARRIER();
    if (wm == 1) VR_PP_BARRIER();  // waves 4-7 fall one segment behind again
  }  // tiles
#undef VR_PP_BARRIER
#undef VR_PP_VMCNT4
#undef VR_PP_VMCNT_TAIL
}

// ---- the f16 product as TWO INDEPENDENT 256x128 tiles per CU ----------------------------------------------
//
// gemm_f16_pp_kernel owns a CU alone (128 KiB of LDS, 8 waves): while its eight waves run the epilogue — at
// K = 768 the 128 KiB an output tile writes are a third of the tile's time, bound by the CU's store path —
// the matrix pipes idle, and while they multiply the store path idles. Here a block is FOUR waves (one per
// SIMD) on a 256-token x 128-feature tile with 72 KiB of LDS, so two blocks share a CU: they are dispatched
// independently and drift apart, and while one stores its tile the other has the matrix pipes to itself.
// Inside the K loop the two blocks' waves of a SIMD interleave the way the ping-pong halves do: one reads
// fragments or waits at its block's barrier while the other multiplies.
//   * K advances in steps of 32 through a ring of THREE stage buffers (A 256 x 32 + W 128 x 32 halfs = 24 KiB):
//     step s multiplies out of stage s % 3 while the loads of steps s + 1 and s + 2 are in flight — one
//     counted vmcnt(6) and one barrier per step (every wave issues 6 direct-to-LDS loads per step);
//   * rows are 64 bytes in LDS; 16-byte chunk c of row r sits at chunk c ^ ((r >> 1) & 3) (source-side swizzle of
//     the direct-to-LDS loads), which makes the ds_read_b128 fragment reads conflict-free;
//   * same wave tile (128 x 64), fragment roles, W row permutation and accumulation order (k ascending) as
//     gemm_f16_pp_kernel: the result is bit-identical to it, and the epilogue is the same code (direct_epilogue).
// One tile per block, no persistence: consecutive blocks of an XCD (blockIdx % 8) take the column tiles of one
// 256-row panel one after the other, so a panel of A is read from HBM once and then from that XCD's L2.
constexpr int D2M = 256, D2N = 128, D2K = 32;
constexpr int kD2StageHalfs = (D2M + D2N) * D2K;  // 24 KiB

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f16_d2_kernel(
    const half_t* __restrict__ Ah, const half_t* __restrict__ Wh, const float* __restrict__ bias,
    const float* __restrict__ R, float* __restrict__ C, half_t* __restrict__ Ch, half_t* __restrict__ Cl, int M,
    int N, int K, float unscale, const float2* __restrict__ ln_stat, const float* __restrict__ ln_g,
    const float* __restrict__ ln_b) {
  __shared__ half_t lds[3 * kD2StageHalfs + 4 * kWaveStatHalfs];  // the only LDS object (see gemm_f16x3_256_kernel)
  const int tiles_n = (N + D2N - 1) / D2N;
  const int tiles_m = (M + D2M - 1) / D2M;
  const int xcd = static_cast<int>(blockIdx.x) & 7, seq = static_cast<int>(blockIdx.x) >> 3;
  const int panel = (seq / tiles_n) * 8 + xcd;
  if (panel >= tiles_m) return;  // block-uniform (the grid is padded to a multiple of 8 panels)
  const int bm = panel * D2M, bn = (seq % tiles_n) * D2N;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  float2* wave_stat = reinterpret_cast<float2*>(lds + 3 * kD2StageHalfs + wave * kWaveStatHalfs);

  // staging: one direct-to-LDS load moves 16 rows x 64 B; lane -> row (lane >> 2) of the 16, LDS chunk (lane & 3),
  // source chunk (lane & 3) ^ ((row >> 1) & 3). Wave w stages A rows 16 (w + 4 t) + .. (t = 0..3) and W image rows
  // 16 (w + 4 t) + .. (t = 0, 1); image row 32 P + 16 q + r holds weight row 32 P + 8 (r >> 2) + 4 q + (r & 3)
  // (direct_epilogue: a lane then owns 8 consecutive output features).
  const int srow = lane >> 2;
  const int schunk = ((lane & 3) ^ ((srow >> 1) & 3)) * 8;
  const half_t* g_a[4];
  const half_t* g_w[2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
    g_a[t] = Ah + static_cast<int64_t>(min(bm + 16 * (wave + 4 * t) + srow, M - 1)) * K + schunk;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int img = 16 * (wave + 4 * t);  // first image row of the instruction: P = img >> 5, q = (img >> 4) & 1
    const int wrow = (img & ~31) + 8 * (srow >> 2) + 4 * ((img >> 4) & 1) + (srow & 3);
    g_w[t] = Wh + static_cast<int64_t>(min(bn + wrow, N - 1)) * K + schunk;
  }
  const int da = wave * 16 * D2K;               // + 4 t * 16 * D2K
  const int dw = D2M * D2K + wave * 16 * D2K;   // + 4 t * 16 * D2K

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragments (v_mfma_f32_16x16x32_f16): lane l supplies row (l & 15), k = 8 (l >> 4) + j of the 32-deep step
  const int frow = lane & 15;
  const int fk = ((lane >> 4) ^ ((frow >> 1) & 3)) * 8;
  const int pa = (wm * 128 + frow) * D2K + fk;
  const int pw = D2M * D2K + (wn * 64 + frow) * D2K + fk;
  const int nk = K / D2K;
#ifdef VR_GEMM_DIAG_BUILD
  const int diag_bits = __builtin_amdgcn_readfirstlane(g_gemm_diag);
#endif

#define VR_D2_BARRIER()                   \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)
  auto issue = [&](half_t* stage, int k0) {
#pragma unroll
    for (int t = 0; t < 4; ++t) glds16(g_a[t] + k0, stage + da + t * 64 * D2K);
#pragma unroll
    for (int t = 0; t < 2; ++t) glds16(g_w[t] + k0, stage + dw + t * 64 * D2K);
  };
  issue(lds, 0);
  if (nk > 1) issue(lds + kD2StageHalfs, D2K);

  auto step = [&](auto bsel, int s) {
    constexpr int B = decltype(bsel)::value;
    const half_t* st = lds + B * kD2StageHalfs;
    half_t* nd = lds + ((B + 2) % 3) * kD2StageHalfs;
    // the loads of step s have landed (those of step s + 1, if any, stay in flight) ...
    if (s + 1 < nk)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VR_D2_BARRIER();  // ... every wave's; and every wave has finished with stage (s - 1) % 3 = (s + 2) % 3
    if (s + 2 < nk && !VR_DIAG(1)) issue(nd, (s + 2) * D2K);
    f16x8 af[4], bf[4];  // token rows in two halves of 64: holding all eight fragments did not fit beside acc
#ifdef VR_GEMM_DIAG_BUILD
    for (int i = 0; i < 4; ++i) af[i] = bf[i] = f16x8{1, 1, 1, 1, 1, 1, 1, 1};
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (!VR_DIAG(8)) bf[j] = *reinterpret_cast<const f16x8*>(st + pw + j * 16 * D2K);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (!VR_DIAG(8)) af[i] = *reinterpret_cast<const f16x8*>(st + pa + (4 * h + i) * 16 * D2K);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!VR_DIAG(4)) acc[4 * h + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[4 * h + i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  };
  int s = 0;
#pragma clang loop unroll(disable)
  for (; s + 3 <= nk; s += 3) {
    step(std::integral_constant<int, 0>{}, s);
    step(std::integral_constant<int, 1>{}, s + 1);
    step(std::integral_constant<int, 2>{}, s + 2);
  }
  if (s < nk) step(std::integral_constant<int, 0>{}, s);
  if (s + 1 < nk) step(std::integral_constant<int, 1>{}, s + 1);
#undef VR_D2_BARRIER
  if (VR_DIAG(2)) {
    if (acc[0][0][0] == 12345.678f && acc[7][3][3] == 1.0f) C[0] = acc[3][1][2];  // (diagnostic) keep the accumulators alive
    return;
  }
  // (EPI_RLS_R32_O16 — layer 0 only — holds two pieces of f32 residual rows: four registers too many without branches)
  if (EPI != EPI_RLS_R32_O16 && bm + D2M <= M && bn + D2N <= N)  // block-uniform
    direct_epilogue<EPI, true>(acc, bm + wm * 128, bn + wn * 64, lane, bias, R, C, Ch, Cl, M, N, unscale, ln_stat, ln_g, ln_b, wave_stat);
  else
    direct_epilogue<EPI, false>(acc, bm + wm * 128, bn + wn * 64, lane, bias, R, C, Ch, Cl, M, N, unscale, ln_stat, ln_g, ln_b, wave_stat);
}

// ---- skinny product for M <= 256 rows (one query, a handful of sequences) ---------------------------
//
// With a few rows the 256x256 kernel leaves the chip empty (N / 256 blocks, each walking all of K:
// 1.7 ms for a 12-token bge-base query). Here the work is cut along N (64 columns per block) AND along
// K (slices of kSkinnyK), so 50-200 blocks stream disjoint slabs of the weight matrix — the only real
// traffic — and write f32 partial sums; a second small kernel adds the slices and applies the
// epilogue. D[n][m] = sum_k W[n][k] A[m][k] on v_mfma_f32_16x16x32_f16: W rows are the MFMA rows, the
// activation rows (64 per block) its columns; both operands are 16-byte global loads (k contiguous).
// f16 mode only (plain f16 rows).
constexpr int kSkinnyM = 64;
// K slice per block: the largest of these that divides K (192 for 384 / 768 / 1536 / 3072, 256 for 1024 / 4096)
static int skinny_slice(int K) {
  for (int c : {192, 256, 128, 64, 32})
    if (K % c == 0) return c;
  return 0;
}

__global__ __launch_bounds__(256) void gemm_f16_skinny_kernel(const half_t* __restrict__ A, const half_t* __restrict__ W,
                                                              float* __restrict__ part, int M, int N, int K,
                                                              int kslice) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 64 + wave * 16;  // this wave's 16 output columns
  const int k0 = blockIdx.y * kslice;
  const int row = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.z * kSkinnyM;  // 64 activation rows per block
  if (n0 >= N) return;
  const half_t* wp = W + static_cast<int64_t>(min(n0 + row, N - 1)) * K + k0 + 8 * g;
  const half_t* ap[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) ap[mb] = A + static_cast<int64_t>(min(m0 + mb * 16 + row, M - 1)) * K + k0 + 8 * g;
  f32x4 acc[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int mblocks = (min(M - m0, kSkinnyM) + 15) / 16;
  for (int k = 0; k < kslice; k += 32) {
    const f16x8 wf = *reinterpret_cast<const f16x8*>(wp + k);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
      if (mb < mblocks) {
        const f16x8 af = *reinterpret_cast<const f16x8*>(ap[mb] + k);
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, af, acc[mb], 0, 0, 0);
      }
  }
  // C layout: column (lane & 15) = activation row m, rows 4g + r = output columns n0 + 4g + r
  float* dst = part + static_cast<int64_t>(blockIdx.y) * M * N;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int m = m0 + mb * 16 + row;
    if (mb < mblocks && m < M && n0 + 4 * g < N)
      *reinterpret_cast<float4*>(dst + static_cast<int64_t>(m) * N + n0 + 4 * g) =
          make_float4(acc[mb][0], acc[mb][1], acc[mb][2], acc[mb][3]);
  }
}

// the epilogue of the 256-tile kernel (same formulas) for the float4 of raw sums v at row m, column n
template <int EPI>
__device__ __forceinline__ void skinny_apply(float4 v, int m, int n, int N, float unscale,
                                             const float* __restrict__ bias, const float* __restrict__ R,
                                             const float2* __restrict__ ln_stat, const float* __restrict__ ln_g,
                                             const float* __restrict__ ln_b, float* __restrict__ C,
                                             half_t* __restrict__ Ch) {
  const int64_t o = static_cast<int64_t>(m) * N + n;
  const float4 b4 = *reinterpret_cast<const float4*>(bias + n);
  v.x = v.x * unscale + b4.x;
  v.y = v.y * unscale + b4.y;
  v.z = v.z * unscale + b4.z;
  v.w = v.w * unscale + b4.w;
  if (EPI == EPI_BIAS_GELU) {
    const f32x2 g01 = gelu_poly2(f32x2{v.x, v.y}), g23 = gelu_poly2(f32x2{v.z, v.w});  // f16 mode only
    v = make_float4(g01.x, g01.y, g23.x, g23.y);
  }
  if (EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_F16) {
    half_t h[4] = {static_cast<half_t>(fminf(fmaxf(v.x, -65504.0f), 65504.0f)),
                   static_cast<half_t>(fminf(fmaxf(v.y, -65504.0f), 65504.0f)),
                   static_cast<half_t>(fminf(fmaxf(v.z, -65504.0f), 65504.0f)),
                   static_cast<half_t>(fminf(fmaxf(v.w, -65504.0f), 65504.0f))};
    *reinterpret_cast<uint2*>(Ch + o) = *reinterpret_cast<const uint2*>(h);
    return;
  }
  if (EPI == EPI_BIAS_RESIDUAL || EPI == EPI_BIAS_RESIDUAL_LN) {
    const float4 r = *reinterpret_cast<const float4*>(R + o);
    if (EPI == EPI_BIAS_RESIDUAL_LN) {
      const float2 st = ln_stat[m];
      const float4 lg = *reinterpret_cast<const float4*>(ln_g + n), lb = *reinterpret_cast<const float4*>(ln_b + n);
      v.x += ln_apply(r.x, st.x, st.y, lg.x, lb.x);
      v.y += ln_apply(r.y, st.x, st.y, lg.y, lb.y);
      v.z += ln_apply(r.z, st.x, st.y, lg.z, lb.z);
      v.w += ln_apply(r.w, st.x, st.y, lg.w, lb.w);
    } else {
      v.x += r.x;
      v.y += r.y;
      v.z += r.z;
      v.w += r.w;
    }
  }
  *reinterpret_cast<float4*>(C + o) = v;
}

// sum of the K slices + epilogue, one float4 per thread (second launch of the split-K skinny GEMM)
template <int EPI>
__global__ void skinny_epilogue_kernel(const float* __restrict__ part, int slices, int M, int N, float unscale,
                                       const float* __restrict__ bias, const float* __restrict__ R,
                                       const float2* __restrict__ ln_stat, const float* __restrict__ ln_g,
                                       const float* __restrict__ ln_b, float* __restrict__ C,
                                       half_t* __restrict__ Ch) {
  const int64_t i4 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t total4 = static_cast<int64_t>(M) * N / 4;
  if (i4 >= total4) return;
  const int64_t o = i4 * 4;
  const int m = static_cast<int>(o / N), n = static_cast<int>(o % N);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < slices; ++s) {
    const float4 p = *reinterpret_cast<const float4*>(part + static_cast<int64_t>(s) * M * N + o);
    v.x += p.x;
    v.y += p.y;
    v.z += p.z;
    v.w += p.w;
  }
  skinny_apply<EPI>(v, m, n, N, unscale, bias, R, ln_stat, ln_g, ln_b, C, Ch);
}

// One launch per skinny GEMM when K % 128 == 0 (every width of the supported models): a block owns 16
// output columns and (up to) 64 activation rows; its NW = 8 (4 when K % 256 != 0) waves each take 1/NW of K
// — so the weight slab of the block (16 x K halfs) is streamed by all of them at once, in 1-3 memory round
// trips per wave — and meet in LDS, where the shares are summed in wave order and the epilogue is applied. N/16 blocks (48-192 for the base
// model) keep the weight stream wide without any cross-block reduction: a second launch costs ~4 us
// here and a device-scope fence per block costs more, an LDS barrier costs nothing.
template <int EPI, int UNR, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_f16_skinny1_kernel(const half_t* __restrict__ A, const half_t* __restrict__ W,
                                                               int M, int N, int K, float unscale,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ R,
                                                               const float2* __restrict__ ln_stat,
                                                               const float* __restrict__ ln_g,
                                                               const float* __restrict__ ln_b, float* __restrict__ C,
                                                               half_t* __restrict__ Ch) {
  __shared__ float red[NW][4][16][17];  // [wave][m block][output column][activation row (+1 pad)]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 16;
  const int m0 = blockIdx.y * kSkinnyM;
  const int row = lane & 15, g = lane >> 4;
  const int kq = K / NW, k0 = wave * kq;  // this wave's share of K
  const half_t* wp = W + static_cast<int64_t>(n0 + row) * K + k0 + 8 * g;
  const half_t* ap[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) ap[mb] = A + static_cast<int64_t>(min(m0 + mb * 16 + row, M - 1)) * K + k0 + 8 * g;
  f32x4 acc[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int mblocks = (min(M - m0, kSkinnyM) + 15) / 16;
  // UNR 32-deep steps at a time: all their loads are issued before the first MFMA (a dependent load per
  // step would expose one memory latency per step, and a query's forward pass is nothing but latency)
  for (int k = 0; k < kq; k += 32 * UNR) {
    f16x8 wf[UNR], af[UNR][4];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      wf[u] = *reinterpret_cast<const f16x8*>(wp + k + 32 * u);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
        if (mb < mblocks) af[u][mb] = *reinterpret_cast<const f16x8*>(ap[mb] + k + 32 * u);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
        if (mb < mblocks) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[u], af[u][mb], acc[mb], 0, 0, 0);
  }
  // C layout: column (lane & 15) = activation row, rows 4g + r = output columns n0 + 4g + r
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][mb][4 * g + r][row] = acc[mb][r];
  __syncthreads();
  if (threadIdx.x >= 256) return;
  const int ml = threadIdx.x >> 2, n4 = (threadIdx.x & 3) * 4;  // 64 rows x 4 float4 of columns
  const int m = m0 + ml;
  if (m >= M) return;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float sum = red[0][ml >> 4][n4 + j][ml & 15];
#pragma unroll
    for (int w = 1; w < NW; ++w) sum += red[w][ml >> 4][n4 + j][ml & 15];  // fixed order
    v[j] = sum;
  }
  skinny_apply<EPI>(make_float4(v[0], v[1], v[2], v[3]), m, n0 + n4, N, unscale, bias, R, ln_stat, ln_g, ln_b, C, Ch);
}

// weights: w * scale -> (hi, lo); scale is a power of two chosen from max|w| of the tensor
__global__ void absmax_kernel(const float* __restrict__ w, int64_t n, unsigned int* __restrict__ out) {
  float m = 0.0f;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    m = fmaxf(m, fabsf(w[i]));
  m = fmaxf(m, __shfl_xor(m, 32));
  m = fmaxf(m, __shfl_xor(m, 16));
  m = fmaxf(m, __shfl_xor(m, 8));
  m = fmaxf(m, __shfl_xor(m, 4));
  m = fmaxf(m, __shfl_xor(m, 2));
  m = fmaxf(m, __shfl_xor(m, 1));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));  // non-negative floats order as uints
}

__global__ void split_weights_kernel(const float* __restrict__ w, int64_t n, int K, float scale,
                                     half_t* __restrict__ hi, half_t* __restrict__ lo) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) {
    half_t h, l;
    split_f16(w[i] * scale, h, l);
    if (lo) {
      const int64_t o = (i / K) * (2 * K) + split_at(static_cast<int>(i % K));  // interleaved layout, see split_at
      hi[o] = h;
      lo[o] = l;
    } else {
      hi[i] = h;  // plain f16 rows (VR_PRECISION_F16)
    }
  }
}

template <int PASSES>
static void launch_256(int epi, int grid, hipStream_t s, const half_t* Ah, const half_t* Al, const half_t* Wh,
                       const half_t* Wl, const float* bias, const float* R, float* C, half_t* Ch, half_t* Cl, int M,
                       int N, int K, float unscale, const float2* ln_stat, const float* ln_g, const float* ln_b) {
#define VR_LAUNCH_256(E)                                                                                         \
  hipLaunchKernelGGL((gemm_f16x3_256_kernel<E, PASSES>), dim3(grid), dim3(512), 0, s, Ah, Al, Wh, Wl, bias, R, C, \
                     Ch, Cl, M, N, K, unscale, ln_stat, ln_g, ln_b)
  switch (epi) {
    case EPI_BIAS: VR_LAUNCH_256(EPI_BIAS); break;
    case EPI_BIAS_GELU: VR_LAUNCH_256(EPI_BIAS_GELU); break;
    case EPI_BIAS_F16: VR_LAUNCH_256(EPI_BIAS_F16); break;
    case EPI_BIAS_RESIDUAL_LN: VR_LAUNCH_256(EPI_BIAS_RESIDUAL_LN); break;
    case EPI_FOLD_F16: if constexpr (PASSES == 1) VR_LAUNCH_256(EPI_FOLD_F16); break;
    case EPI_FOLD_GELU: if constexpr (PASSES == 1) VR_LAUNCH_256(EPI_FOLD_GELU); break;
    case EPI_BIAS_RESIDUAL_LN_STATS: if constexpr (PASSES == 1) VR_LAUNCH_256(EPI_BIAS_RESIDUAL_LN_STATS); break;
    default: VR_LAUNCH_256(EPI_BIAS_RESIDUAL); break;
  }
#undef VR_LAUNCH_256
}

// the ping-pong kernel serves the f16 mode when K is whole pairs of 64-deep K-tiles and a lane's 8 features are in
// or out of N together; VR_GEMM_PP=0 keeps the one-barrier-per-K-tile loop (A/B runs)
static bool pp_usable(int N, int K) {
  static const bool pp_on = !(getenv("VR_GEMM_PP") && atoi(getenv("VR_GEMM_PP")) == 0);
  return pp_on && K % 128 == 0 && N % 8 == 0;
}

// gemm_f16_d2_kernel: any K that is a multiple of 32 (the ping-pong kernel needs 128). OFF unless VR_GEMM_D2=1:
// measured on bge-base, 2200 chunks (profiles/r02_gemm_experiments.md) it runs at 750 TFLOP/s against the ping-pong
// kernel's 930 — its epilogues do overlap the other block's main loop, but 32-deep stages of a 256x128 tile pull
// 1.5x the bytes through L2 -> LDS in 64-byte row segments, and that path (14.5 TB/s here) is what bounds it.
static bool d2_usable(int N, int K) {
  // VR_GEMM_D2=1: every projection; 2: only those whose width leaves the 256-wide kernel half a column tile (N % 256 != 0)
  static const int d2_mode = getenv("VR_GEMM_D2") ? atoi(getenv("VR_GEMM_D2")) : 0;
  const bool on = d2_mode == 1 || (d2_mode == 2 && N % 256 != 0);
  return on && K % D2K == 0 && N % 8 == 0;
}

// the kernels whose epilogue is direct_epilogue (all EPI_* variants, f16 residual stream included)
static bool direct_usable(int N, int K) { return d2_usable(N, K) || pp_usable(N, K); }

static void launch_d2(int epi, hipStream_t s, const half_t* Ah, const half_t* Wh, const float* bias, const float* R,
                      float* C, half_t* Ch, half_t* Cl, int M, int N, int K, float unscale, const float2* ln_stat,
                      const float* ln_g, const float* ln_b) {
  const int tiles_m = (M + D2M - 1) / D2M, tiles_n = (N + D2N - 1) / D2N;
  const int grid = (tiles_m + 7) / 8 * 8 * tiles_n;
#ifdef VR_GEMM_DIAG_BUILD
  static bool diag_set = false;
  if (!diag_set) {
    const int v = getenv("VR_GEMM_DIAG") ? atoi(getenv("VR_GEMM_DIAG")) : 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_diag), &v, sizeof(int));
    diag_set = true;
  }
#endif
#define VR_LAUNCH_D2(E)                                                                                       \
  hipLaunchKernelGGL((gemm_f16_d2_kernel<E>), dim3(grid), dim3(256), 0, s, Ah, Wh, bias, R, C, Ch, Cl, M, N, K, \
                     unscale, ln_stat, ln_g, ln_b)
  switch (epi) {
    case EPI_BIAS: VR_LAUNCH_D2(EPI_BIAS); break;
    case EPI_BIAS_GELU: VR_LAUNCH_D2(EPI_BIAS_GELU); break;
    case EPI_BIAS_F16: VR_LAUNCH_D2(EPI_BIAS_F16); break;
    case EPI_BIAS_RESIDUAL_LN: VR_LAUNCH_D2(EPI_BIAS_RESIDUAL_LN); break;
    case EPI_FOLD_F16: VR_LAUNCH_D2(EPI_FOLD_F16); break;
    case EPI_FOLD_GELU: VR_LAUNCH_D2(EPI_FOLD_GELU); break;
    case EPI_BIAS_RESIDUAL_LN_STATS: VR_LAUNCH_D2(EPI_BIAS_RESIDUAL_LN_STATS); break;
    case EPI_RLS_R32_O16: VR_LAUNCH_D2(EPI_RLS_R32_O16); break;
    case EPI_RLS_R16_O16: VR_LAUNCH_D2(EPI_RLS_R16_O16); break;
    case EPI_RLS_R16_O32: VR_LAUNCH_D2(EPI_RLS_R16_O32); break;
    default: VR_LAUNCH_D2(EPI_BIAS_RESIDUAL); break;
  }
#undef VR_LAUNCH_D2
}

static void launch_pp(int epi, int grid, hipStream_t s, const half_t* Ah, const half_t* Wh, const float* bias,
                      const float* R, float* C, half_t* Ch, half_t* Cl, int M, int N, int K, float unscale,
                      const float2* ln_stat, const float* ln_g, const float* ln_b) {
  // start-time spread of the persistent blocks, as a fraction of one tile's time (estimated: ~2400 cycles per
  // K-tile + ~12000 of epilogue, in units of 64 cycles); only when a block walks several tiles
  static const float stagger_frac = getenv("VR_GEMM_STAGGER") ? static_cast<float>(atof(getenv("VR_GEMM_STAGGER"))) : 0.0f;
  // row panels per group of the tile walk (column index slow inside a group): experiment switch VR_GEMM_GROUPM
  static const int group_m = getenv("VR_GEMM_GROUPM") ? std::max(1, atoi(getenv("VR_GEMM_GROUPM"))) : kGroupM256;
  const int tiles_total = ((M + GBM - 1) / GBM) * ((N + GBN - 1) / GBN);
  const int stagger = tiles_total >= 3 * grid ? static_cast<int>(stagger_frac * ((K / 64) * 2400.0f + 12000.0f) / 64.0f) : 0;
#ifdef VR_GEMM_DIAG_BUILD
  static bool diag_set = false;
  if (!diag_set) {
    const int v = getenv("VR_GEMM_DIAG") ? atoi(getenv("VR_GEMM_DIAG")) : 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_diag), &v, sizeof(int));
    diag_set = true;
  }
#endif
#define VR_LAUNCH_PP(E)                                                                                        \
  hipLaunchKernelGGL((gemm_f16_pp_kernel<E>), dim3(grid), dim3(512), 0, s, Ah, Wh, bias, R, C, Ch, Cl, M, N, K, \
                     unscale, ln_stat, ln_g, ln_b, stagger, group_m)
  switch (epi) {
    case EPI_BIAS: VR_LAUNCH_PP(EPI_BIAS); break;
    case EPI_BIAS_GELU: VR_LAUNCH_PP(EPI_BIAS_GELU); break;
    case EPI_BIAS_F16: VR_LAUNCH_PP(EPI_BIAS_F16); break;
    case EPI_BIAS_RESIDUAL_LN: VR_LAUNCH_PP(EPI_BIAS_RESIDUAL_LN); break;
    case EPI_FOLD_F16: VR_LAUNCH_PP(EPI_FOLD_F16); break;
    case EPI_FOLD_GELU: VR_LAUNCH_PP(EPI_FOLD_GELU); break;
    case EPI_BIAS_RESIDUAL_LN_STATS: VR_LAUNCH_PP(EPI_BIAS_RESIDUAL_LN_STATS); break;
    case EPI_RLS_R32_O16: VR_LAUNCH_PP(EPI_RLS_R32_O16); break;
    case EPI_RLS_R16_O16: VR_LAUNCH_PP(EPI_RLS_R16_O16); break;
    case EPI_RLS_R16_O32: VR_LAUNCH_PP(EPI_RLS_R16_O32); break;
    default: VR_LAUNCH_PP(EPI_BIAS_RESIDUAL); break;
  }
#undef VR_LAUNCH_PP
#ifdef VR_GEMM_HAS_STAMPS
  static int stamps_left = getenv("VR_GEMM_STAMPS") ? atoi(getenv("VR_GEMM_STAMPS")) : 0;
  if (stamps_left > 0) {
    --stamps_left;
    (void)hipStreamSynchronize(s);
    static long long h[2][16][8];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pp_stamps), sizeof(h));
    fprintf(stderr, "[pp stamps] epi %d M %d N %d K %d: per tile, cycles: main loop (to first new-load wait, to second) | align+wait | epilogue | to next tile\n",
            epi, M, N, K);
    for (int w = 0; w < 2; ++w)
      for (int t = 0; t + 1 < 12; ++t)
        fprintf(stderr, "  wave %d tile %2d: main %6lld (first %5lld, second %5lld) | %5lld | %6lld | %5lld\n", 4 * w, t, h[w][t][2] - h[w][t][0],
                h[w][t][1] - h[w][t][0], h[w][t][6] - h[w][t][0], h[w][t][3] - h[w][t][2], h[w][t][4] - h[w][t][3], h[w][t + 1][0] - h[w][t][4]);
  }
#endif
}

// The skinny GEMM of a handful of tokens (M <= 16) with the LayerNorm IN FRONT of it folded in: the
// activation operand is built from the pre-LayerNorm f32 rows instead of being read as f16 rows that a
// LayerNorm launch wrote a few microseconds earlier. Every block recomputes the (mean, 1/sigma) of the
// <= 16 rows (two passes over values it holds in registers, cross-wave sums in LDS in wave order, so every
// block gets the same bits), normalises its waves' shares of the columns and feeds the MFMA; block 0 also
// stores the statistics, which the residual epilogue of the NEXT projection needs (ln_apply re-derives
// the LayerNorm output from the pre-LN row). K = hidden size: each wave holds STEPS * 8 values per row.
// Two launches fewer per layer for a single query (7 -> 5): its forward pass is launch latency, not work.
template <int EPI, int STEPS, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_f16_skinny_ln_kernel(
    const float* __restrict__ pre, const float* __restrict__ ln_g, const float* __restrict__ ln_b, float eps,
    float2* __restrict__ stat_out, const half_t* __restrict__ W, int M, int N, int K, float unscale,
    const float* __restrict__ bias, half_t* __restrict__ Ch) {
  __shared__ float red[NW][16][17];  // [wave][output column][activation row (+1 pad)]
  __shared__ float part[2][NW][16];  // row sums per wave: [pass][wave][row]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 16;
  const int row = lane & 15, g = lane >> 4;
  const int kq = K / NW, k0 = wave * kq;
  const float* xp = pre + static_cast<int64_t>(min(row, M - 1)) * K + k0 + 8 * g;
  const half_t* wp = W + static_cast<int64_t>(n0 + row) * K + k0 + 8 * g;
  float x[STEPS][8];
  f16x8 wf[STEPS];
  float sum = 0.0f;
#pragma unroll
  for (int u = 0; u < STEPS; ++u) {
    wf[u] = *reinterpret_cast<const f16x8*>(wp + 32 * u);
    const float4 a = *reinterpret_cast<const float4*>(xp + 32 * u), c = *reinterpret_cast<const float4*>(xp + 32 * u + 4);
    x[u][0] = a.x, x[u][1] = a.y, x[u][2] = a.z, x[u][3] = a.w, x[u][4] = c.x, x[u][5] = c.y, x[u][6] = c.z, x[u][7] = c.w;
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += x[u][j];
  }
  // a row's values of this wave sit in the four lanes row, row + 16, row + 32, row + 48
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  if (g == 0) part[0][wave][row] = sum;
  __syncthreads();
  float total = part[0][0][row];
#pragma unroll
  for (int w = 1; w < NW; ++w) total += part[0][w][row];
  const float mean = total / static_cast<float>(K);
  float sq = 0.0f;
#pragma unroll
  for (int u = 0; u < STEPS; ++u)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = x[u][j] - mean;
      sq += d * d;
    }
  sq += __shfl_xor(sq, 16);
  sq += __shfl_xor(sq, 32);
  if (g == 0) part[1][wave][row] = sq;
  __syncthreads();
  float total_sq = part[1][0][row];
#pragma unroll
  for (int w = 1; w < NW; ++w) total_sq += part[1][w][row];
  const float inv = 1.0f / sqrtf(total_sq / static_cast<float>(K) + eps);
  if (blockIdx.x == 0 && wave == 0 && g == 0 && row < M) stat_out[row] = make_float2(mean, inv);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < STEPS; ++u) {
    const float4 g0 = *reinterpret_cast<const float4*>(ln_g + k0 + 8 * g + 32 * u), g1 = *reinterpret_cast<const float4*>(ln_g + k0 + 8 * g + 32 * u + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(ln_b + k0 + 8 * g + 32 * u), b1 = *reinterpret_cast<const float4*>(ln_b + k0 + 8 * g + 32 * u + 4);
    const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    f16x8 af;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      half_t h, l;
      split_f16(ln_apply(x[u][j], mean, inv, gg[j], bb[j]), h, l);  // as layernorm_f16_kernel rounds its output
      af[j] = h;
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[u], af, acc, 0, 0, 0);
  }
  // C layout: column (lane & 15) = activation row, rows 4g + r = output columns n0 + 4g + r
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][4 * g + r][row] = acc[r];
  __syncthreads();
  if (threadIdx.x >= 64) return;
  const int m = threadIdx.x >> 2, n4 = (threadIdx.x & 3) * 4;  // 16 rows x 4 float4 of columns
  if (m >= M) return;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float t = red[0][n4 + j][m];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += red[w][n4 + j][m];  // fixed order
    v[j] = t;
  }
  skinny_apply<EPI>(make_float4(v[0], v[1], v[2], v[3]), m, n0 + n4, N, unscale, bias, nullptr, nullptr, nullptr, nullptr,
                    nullptr, Ch);
}

// K = hidden size H: 8 waves x 3 steps (768), 8 x 4 (1024), 4 x 3 (384); anything else keeps the LayerNorm launch
static bool skinny_ln_supported(int M, int N, int K) { return M <= 16 && N % 16 == 0 && (K == 768 || K == 1024 || K == 384); }

static int launch_skinny_ln(vr_engine* e, int epi, const float* pre, const float* ln_g, const float* ln_b, float eps,
                            float2* stat_out, const half_t* W, float unscale, const float* bias, half_t* Ch, int M,
                            int N, int K) {
  hipStream_t s = e->stream;
  prof_begin(e, VR_PROF_GEMM, 2.0 * M * static_cast<double>(N) * K);
  const dim3 grid(static_cast<unsigned>(N / 16));
#define VR_SKINNY_LN(E, ST, NWV)                                                                                   \
  hipLaunchKernelGGL((gemm_f16_skinny_ln_kernel<E, ST, NWV>), grid, dim3(NWV * 64), 0, s, pre, ln_g, ln_b, eps,    \
                     stat_out, W, M, N, K, unscale, bias, Ch)
  if (epi == EPI_BIAS_GELU) {
    if (K == 768) VR_SKINNY_LN(EPI_BIAS_GELU, 3, 8);
    else if (K == 1024) VR_SKINNY_LN(EPI_BIAS_GELU, 4, 8);
    else VR_SKINNY_LN(EPI_BIAS_GELU, 3, 4);
  } else {
    if (K == 768) VR_SKINNY_LN(EPI_BIAS_F16, 3, 8);
    else if (K == 1024) VR_SKINNY_LN(EPI_BIAS_F16, 4, 8);
    else VR_SKINNY_LN(EPI_BIAS_F16, 3, 4);
  }
#undef VR_SKINNY_LN
  prof_end(e);
  VR_HIP(hipGetLastError());
  return 0;
}

// passes = 3: operands are interleaved (hi, lo) rows (Al = Ah + 8, Wl = Wh + 8); passes = 1: plain f16
// rows, Al / Wl / Cl unused.
static int launch_gemm_f16x3(vr_engine* e, int epi, const half_t* Ah, const half_t* Al, const half_t* Wh,
                             const half_t* Wl, float unscale, const float* bias, const float* R, float* C,
                             half_t* Ch, half_t* Cl, int M, int N, int K, int passes = 3,
                             const float2* ln_stat = nullptr, const float* ln_g = nullptr, const float* ln_b = nullptr) {
  VR_CHECK(N % 4 == 0 && K % HBK_ == 0, "GEMM shape N=%d K=%d: N must be a multiple of 4, K of %d", N, K, HBK_);
  if (M <= 0) return 0;
  hipStream_t s = e->stream;
  static const int force_tile = getenv("VR_GEMM_TILE") ? atoi(getenv("VR_GEMM_TILE")) : 0;  // 128: A/B runs
  prof_begin(e, VR_PROF_GEMM, 2.0 * M * static_cast<double>(N) * K);
  // one-launch skinny kernel: 8 waves per block when an eighth of K is whole 32-deep MFMA steps, else 4;
  // UNR = steps whose loads are issued together (the largest of 6, 4, 3 dividing the wave's step count)
  const int sk_waves = K % 256 == 0 ? 8 : K % 128 == 0 ? 4 : 0;
  const int sk_steps = sk_waves ? K / (32 * sk_waves) : 0;
  const int sk_unr = sk_steps == 0 ? 0 : sk_steps % 6 == 0 ? 6 : sk_steps % 4 == 0 ? 4 : sk_steps % 3 == 0 ? 3 : 0;
  if (passes == 1 && M <= 4 * kSkinnyM && sk_unr > 0 && N % 16 == 0) {
    const dim3 sg(static_cast<unsigned>(N / 16), static_cast<unsigned>((M + kSkinnyM - 1) / kSkinnyM));
#define VR_SKINNY1_LAUNCH(E, U, W)                                                                                    \
  hipLaunchKernelGGL((gemm_f16_skinny1_kernel<E, U, W>), sg, dim3(W * 64), 0, s, Ah, Wh, M, N, K, unscale, bias, R,  \
                     ln_stat, ln_g, ln_b, C, Ch)
#define VR_SKINNY1(E)                                                  \
  do {                                                                 \
    if (sk_waves == 8) {                                               \
      if (sk_unr == 6) VR_SKINNY1_LAUNCH(E, 6, 8);                     \
      else if (sk_unr == 4) VR_SKINNY1_LAUNCH(E, 4, 8);                \
      else VR_SKINNY1_LAUNCH(E, 3, 8);                                 \
    } else {                                                           \
      if (sk_unr == 6) VR_SKINNY1_LAUNCH(E, 6, 4);                     \
      else if (sk_unr == 4) VR_SKINNY1_LAUNCH(E, 4, 4);                \
      else VR_SKINNY1_LAUNCH(E, 3, 4);                                 \
    }                                                                  \
  } while (0)
    switch (epi) {
      case EPI_BIAS: VR_SKINNY1(EPI_BIAS); break;
      case EPI_BIAS_GELU: VR_SKINNY1(EPI_BIAS_GELU); break;
      case EPI_BIAS_F16: VR_SKINNY1(EPI_BIAS_F16); break;
      case EPI_BIAS_RESIDUAL_LN: VR_SKINNY1(EPI_BIAS_RESIDUAL_LN); break;
      default: VR_SKINNY1(EPI_BIAS_RESIDUAL); break;
    }
#undef VR_SKINNY1
#undef VR_SKINNY1_LAUNCH
    prof_end(e);
    VR_HIP(hipGetLastError());
    return 0;
  }
  if (passes == 1 && M <= 4 * kSkinnyM && skinny_slice(K) > 0 && N % 64 == 0) {  // K % 128 != 0: split-K, two launches
    Encoder* enc = static_cast<Encoder*>(e->encoder);
    const int kslice = skinny_slice(K), slices = K / kslice;
    {
      const float* before = enc->skinny_ws.p;
      VR_TRY(enc->skinny_ws.grow(static_cast<int64_t>(slices) * M * N, 0, s));
      if (enc->skinny_ws.p != before) invalidate_graphs(enc);  // (never during a capture: a shape runs eagerly first)
    }
    hipLaunchKernelGGL(gemm_f16_skinny_kernel, dim3(static_cast<unsigned>(N / 64), static_cast<unsigned>(slices), static_cast<unsigned>((M + kSkinnyM - 1) / kSkinnyM)),
                       dim3(256), 0, s, Ah, Wh, enc->skinny_ws.p, M, N, K, kslice);
    const unsigned eb = static_cast<unsigned>((static_cast<int64_t>(M) * N / 4 + 255) / 256);
#define VR_SKINNY_EPI(E)                                                                                          \
  hipLaunchKernelGGL((skinny_epilogue_kernel<E>), dim3(eb), dim3(256), 0, s, enc->skinny_ws.p, slices, M, N, unscale, \
                     bias, R, ln_stat, ln_g, ln_b, C, Ch)
    switch (epi) {
      case EPI_BIAS: VR_SKINNY_EPI(EPI_BIAS); break;
      case EPI_BIAS_GELU: VR_SKINNY_EPI(EPI_BIAS_GELU); break;
      case EPI_BIAS_F16: VR_SKINNY_EPI(EPI_BIAS_F16); break;
      case EPI_BIAS_RESIDUAL_LN: VR_SKINNY_EPI(EPI_BIAS_RESIDUAL_LN); break;
      default: VR_SKINNY_EPI(EPI_BIAS_RESIDUAL); break;
    }
#undef VR_SKINNY_EPI
    prof_end(e);
    VR_HIP(hipGetLastError());
    return 0;
  }
  if (passes == 1 || (N % GBN == 0 && K % GBK == 0 && M >= GBM && force_tile != 128)) {
    static int n_cu = 0;  // persistent grid: one block per CU (the kernel uses 128 KiB of the CU's LDS)
    if (n_cu == 0) {
      hipDeviceProp_t prop;
      VR_HIP(hipGetDeviceProperties(&prop, e->device));
      n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount / 8 * 8 : 256;
      if (n_cu == 0) n_cu = prop.multiProcessorCount;
    }
    const int tiles = ((M + GBM - 1) / GBM) * ((N + GBN - 1) / GBN);
    const int grid256 = std::min(tiles, n_cu);
    if (passes == 1 && d2_usable(N, K))  // f16 mode: two independent half-width tiles per CU (gemm_f16_d2_kernel)
      launch_d2(epi, s, Ah, Wh, bias, R, C, Ch, Cl, M, N, K, unscale, ln_stat, ln_g, ln_b);
    else if (passes == 1 && pp_usable(N, K))  // the ping-pong main loop (gemm_f16_pp_kernel)
      launch_pp(epi, grid256, s, Ah, Wh, bias, R, C, Ch, Cl, M, N, K, unscale, ln_stat, ln_g, ln_b);
    else if (passes == 1)
      launch_256<1>(epi, grid256, s, Ah, Al, Wh, Wl, bias, R, C, Ch, Cl, M, N, K, unscale, ln_stat, ln_g, ln_b);
    else
      launch_256<3>(epi, grid256, s, Ah, Al, Wh, Wl, bias, R, C, Ch, Cl, M, N, K, unscale, ln_stat, ln_g, ln_b);
    prof_end(e);
    VR_HIP(hipGetLastError());
    return 0;
  }
  VR_CHECK(N % HBN_ == 0, "GEMM shape N=%d must be a multiple of %d", N, HBN_);
  const int grid = ((M + HBM_ - 1) / HBM_) * (N / HBN_);
  switch (epi) {
    case EPI_BIAS:
      hipLaunchKernelGGL((gemm_f16x3_kernel<EPI_BIAS>), dim3(grid), dim3(256), 0, s, Ah, Al, Wh, Wl, bias, R, C, Ch,
                         Cl, M, N, K, unscale);
      break;
    case EPI_BIAS_GELU:
      hipLaunchKernelGGL((gemm_f16x3_kernel<EPI_BIAS_GELU>), dim3(grid), dim3(256), 0, s, Ah, Al, Wh, Wl, bias, R,
                         C, Ch, Cl, M, N, K, unscale);
      break;
    default:
      hipLaunchKernelGGL((gemm_f16x3_kernel<EPI_BIAS_RESIDUAL>), dim3(grid), dim3(256), 0, s, Ah, Al, Wh, Wl, bias,
                         R, C, Ch, Cl, M, N, K, unscale);
      break;
  }
  prof_end(e);
  VR_HIP(hipGetLastError());
  return 0;
}

// ---- attention -----------------------------------------------------------------------------------

// One block = 64 queries of one (sequence, head); wave w owns queries 16w..16w+15.
// S^T = K Q^T is computed with keys on the MFMA rows, so a lane (q = lane & 15, g = lane >> 4)
// ends up holding the logits of query q against keys 4g..4g+3 of each 16-key tile: exactly the B
// operand layout of the following O^T += V^T P^T product — no transpose, no LDS round trip for P.
template <int DH>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv,
                                                        const int32_t* __restrict__ cu, int seq0,
                                                        int tok_base, int H, int qblocks, float scale,
                                                        float* __restrict__ ctx, half_t* __restrict__ ctx_hi,
                                                        half_t* __restrict__ ctx_lo) {
  constexpr int LDK = DH + 4;
  constexpr int NS = DH / 16;  // 16-wide d blocks
  __shared__ float sK[64 * LDK];
  __shared__ float sV[64 * LDK];
  const int seq = seq0 + blockIdx.x / qblocks;
  const int qb = blockIdx.x % qblocks;
  const int head = blockIdx.y;
  const int t0 = cu[seq] - tok_base;
  const int len = cu[seq + 1] - cu[seq];
  if (qb * 64 >= len) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int qi = lane & 15, g = lane >> 4;
  const int q_tok = qb * 64 + wave * 16 + qi;
  const bool q_valid = q_tok < len;
  const int64_t row3 = 3 * static_cast<int64_t>(H);

  float4 qf[NS];
  {
    const float* qp = qkv + (t0 + (q_valid ? q_tok : len - 1)) * row3 + head * DH + 4 * g;
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = *reinterpret_cast<const float4*>(qp + 16 * s);
  }
  f32x4 o[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) o[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -__builtin_inff();
  float l = 0.0f;

  for (int kt = 0; kt < len; kt += 64) {
    __syncthreads();
    for (int idx = tid; idx < 64 * (DH / 4); idx += 256) {
      const int key = idx / (DH / 4), c4 = idx % (DH / 4);
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (kt + key < len) {
        const float* p = qkv + (t0 + kt + key) * row3 + H + head * DH + c4 * 4;
        kv = *reinterpret_cast<const float4*>(p);
        vv = *reinterpret_cast<const float4*>(p + H);
      }
      *reinterpret_cast<float4*>(sK + key * LDK + c4 * 4) = kv;
      *reinterpret_cast<float4*>(sV + key * LDK + c4 * 4) = vv;
    }
    __syncthreads();

    f32x4 st[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const float4 kf = *reinterpret_cast<const float4*>(sK + (t * 16 + qi) * LDK + 16 * s + 4 * g);
        st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[s].x, st[t], 0, 0, 0);
        st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[s].y, st[t], 0, 0, 0);
        st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[s].z, st[t], 0, 0, 0);
        st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[s].w, st[t], 0, 0, 0);
      }
    }
    float mx = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt + t * 16 + 4 * g + r;
        const float v = key < len ? st[t][r] * scale : -__builtin_inff();
        st[t][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m, mx);  // finite: key kt < len is always valid
    const float alpha = expf(m - m_new);
    float psum = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = expf(st[t][r] - m_new);
        st[t][r] = p;
        psum += p;
      }
    l = l * alpha + psum;
    m = m_new;
#pragma unroll
    for (int s = 0; s < NS; ++s) o[s] *= alpha;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* vrow = sV + (t * 16 + 4 * g + r) * LDK + qi;
#pragma unroll
        for (int s = 0; s < NS; ++s)
          o[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[16 * s], st[t][r], o[s], 0, 0, 0);
      }
  }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  if (q_valid) {
    const float inv = 1.0f / l;
    const int64_t off = static_cast<int64_t>(t0 + q_tok) * H + head * DH + 4 * g;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float4 v = make_float4(o[s][0] * inv, o[s][1] * inv, o[s][2] * inv, o[s][3] * inv);
      if (ctx_hi) {  // f16x3 mode: the only consumer is the output-projection GEMM
        half_t h[4], lo[4];
        split_f16(v.x, h[0], lo[0]);
        split_f16(v.y, h[1], lo[1]);
        split_f16(v.z, h[2], lo[2]);
        split_f16(v.w, h[3], lo[3]);
        if (ctx_lo) {
          const int64_t so = static_cast<int64_t>(t0 + q_tok) * (2 * H) + split_at(head * DH + 4 * g + 16 * s);
          *reinterpret_cast<uint2*>(ctx_hi + so) = *reinterpret_cast<const uint2*>(h);
          *reinterpret_cast<uint2*>(ctx_lo + so) = *reinterpret_cast<const uint2*>(lo);
        } else {  // plain f16 rows (VR_PRECISION_F16)
          *reinterpret_cast<uint2*>(ctx_hi + off + 16 * s) = *reinterpret_cast<const uint2*>(h);
        }
      } else {
        *reinterpret_cast<float4*>(ctx + off + 16 * s) = v;
      }
    }
  }
}

// The same attention on the f16 matrix pipe (VR_PRECISION_F16): Q (pre-scaled by 1/sqrt(d_h)), K, V and
// the probabilities P are rounded to f16, every product accumulates in f32, the softmax stays f32.
// v_mfma_f32_16x16x32_f16 has the C layout of the 16x16x4 instruction, so the structure above
// carries over: S^T = K Q^T leaves lane (q, g) with the logits of keys 16t + 4g + r — and the B
// operand of O^T += V^T P^T wants 8 k-slots per lane, so a 32-key step takes its slots in the order
// (g, tile parity, r): slot 8g + 4b + r = key 32u + 16b + 4g + r. V is staged TRANSPOSED in that
// order (sVt[d][slot]), so its A fragments are plain 16-byte reads. 16 MFMAs of 16 cycles per 64
// keys against 128 of 32 cycles in the f32 kernel: the kernel is bound by the softmax's VALU work.
// NW waves = 16 * NW queries per block.
template <int DH, int NW>
__global__ __launch_bounds__(NW * 64) void attention_f16_kernel(const half_t* __restrict__ qkv,
                                                            const int32_t* __restrict__ cu, int seq0,
                                                            int tok_base, int H, int qblocks, float scale,
                                                            half_t* __restrict__ ctx_h) {
  constexpr int LDK = DH + 8;   // halfs per staged K row
  constexpr int LDV = 64 + 8;   // halfs per staged V^T row (64 key slots)
  constexpr int NS = DH / 16;   // 16-wide d blocks of the output
  constexpr int NKB = DH / 32;  // 32-deep MFMA steps over d
  __shared__ half_t sK[64 * LDK];
  __shared__ half_t sVt[DH * LDV];
  __shared__ half_t sV[64 * LDK];  // V rows as loaded; transposed into sVt by the wave that wrote them
  const int seq = seq0 + blockIdx.x / qblocks;
  const int qb = blockIdx.x % qblocks;
  const int head = blockIdx.y;
  const int t0 = cu[seq] - tok_base;
  const int len = cu[seq + 1] - cu[seq];
  if (qb * (NW * 16) >= len) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int qi = lane & 15, g = lane >> 4;
  const int q_tok = qb * (NW * 16) + wave * 16 + qi;
  const bool q_valid = q_tok < len;
  const int64_t row3 = 3 * static_cast<int64_t>(H);

  f16x8 qf[NKB];
  {
    const half_t* qp = qkv + (t0 + (q_valid ? q_tok : len - 1)) * row3 + head * DH + 8 * g;
#pragma unroll
    for (int u = 0; u < NKB; ++u) qf[u] = *reinterpret_cast<const f16x8*>(qp + 32 * u);
  }
  const float scale2 = scale * 1.4426950408889634f;  // logits in the log2 domain: the softmax uses v_exp_f32 directly
  f32x4 o[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) o[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -__builtin_inff();
  float l = 0.0f;

  for (int kt = 0; kt < len; kt += 64) {
    __syncthreads();
    // K: 16-byte rows. V: transposed on the way in, in two conflict-free steps — the wave writes its 8
    // (DH = 64) or 16 (DH = 32) key rows to sV as loaded, then lane d reads column d of 8 of those rows
    // (64 lanes = 128 contiguous bytes per row) and writes them as two 8-byte pieces of sVt[d]. (Scattering
    // the eight halfs of a loaded row straight into eight sVt rows put the lanes of a store 8 rows apart on
    // two banks: 70 % of the kernel's LDS cycles were bank conflicts.)
    for (int idx = tid; idx < 64 * (DH / 8); idx += NW * 64) {
      const int key = idx / (DH / 8), c8 = idx % (DH / 8);
      uint4 kv = make_uint4(0, 0, 0, 0), vv = kv;
      if (kt + key < len) {
        const half_t* p = qkv + (t0 + kt + key) * row3 + H + head * DH + c8 * 8;
        kv = *reinterpret_cast<const uint4*>(p);
        vv = *reinterpret_cast<const uint4*>(p + H);
      }
      *reinterpret_cast<uint4*>(sK + key * LDK + c8 * 8) = kv;
      *reinterpret_cast<uint4*>(sV + key * LDK + c8 * 8) = vv;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      // the wave's keys this iteration: kb .. kb + 512 / DH - 1; lane -> (d, group of 8 keys)
      const int kb = (idx - lane) / (DH / 8);
      const int d = lane % DH, k8 = kb + 8 * (lane / DH);
      half_t h[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = sV[(k8 + j) * LDK + d];
      // keys k8 .. k8 + 3 -> slots sa .. sa + 3, keys k8 + 4 .. k8 + 7 -> slots sa + 8 .. sa + 11
      // (key = 32u + 16b + 4g' + r  ->  slot 32u + 8g' + 4b + r, and k8 is a multiple of 8)
      const int sa = (k8 & 32) + ((k8 >> 2) & 3) * 8 + ((k8 >> 4) & 1) * 4;
      *reinterpret_cast<uint2*>(sVt + d * LDV + sa) = *reinterpret_cast<const uint2*>(h);
      *reinterpret_cast<uint2*>(sVt + d * LDV + sa + 8) = *reinterpret_cast<const uint2*>(h + 4);
    }
    __syncthreads();

    f32x4 st[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < NKB; ++u) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(sK + (t * 16 + qi) * LDK + 32 * u + 8 * g);
        st[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[u], st[t], 0, 0, 0);
      }
    }
    float mx = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt + t * 16 + 4 * g + r;
        const float v = key < len ? st[t][r] * scale2 : -__builtin_inff();
        st[t][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m, mx);  // finite: key kt < len is always valid
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    float psum = 0.0f;
    f16x8 pb[2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(st[t][r] - m_new);
        psum += p;
        pb[t >> 1][(t & 1) * 4 + r] = static_cast<half_t>(p);
      }
    l = l * alpha + psum;
    m = m_new;
#pragma unroll
    for (int s = 0; s < NS; ++s) o[s] *= alpha;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const f16x8 vf = *reinterpret_cast<const f16x8*>(sVt + (16 * s + qi) * LDV + 32 * u + 8 * g);
        o[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pb[u], o[s], 0, 0, 0);
      }
  }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  if (q_valid) {
    const float inv = 1.0f / l;
    const int64_t off = static_cast<int64_t>(t0 + q_tok) * H + head * DH + 4 * g;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      half_t h[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = static_cast<half_t>(fminf(fmaxf(o[s][r] * inv, -65504.0f), 65504.0f));
      *reinterpret_cast<uint2*>(ctx_h + off + 16 * s) = *reinterpret_cast<const uint2*>(h);
    }
  }
}

// ---- attention, one block per (sequence, head) -------------------------------------------------------------
//
// attention_f16_kernel above stages 64 keys at a time, once per 64-query block: for the ~118-token chunks of the
// indexing path that is four staging phases per (sequence, head), each with its global-load latency exposed and its
// V tile transposed through LDS by hand — the kernel runs at half of what its 1.6 GB per launch cost at the HBM rate.
// Here the block stages the sequence's K and V rows ONCE, all loads of up to 128 keys in flight together, V stays
// row-major in LDS and its transposed MFMA operand comes out of ds_read_b64_tr_b16 (lane 4q + p of a 16-lane group
// addresses row q, columns 4p..4p+3 of a 4 x 16 block; lane i receives column i); the four waves then walk the
// sequence's 16-query tiles (wave w: tiles w, w + 4, ...) against keys already in LDS. Same arithmetic as
// attention_f16_kernel (logits on the f16 MFMA, online softmax in the log2 domain over 64-key tiles, P as f16).
// `q_limit`: only queries below it are computed (the last layer of a CLS-pooled model needs token 0 only).
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int DH, int NW>
__global__ __launch_bounds__(NW * 64) void attention_seq_kernel(const half_t* __restrict__ qkv, const int32_t* __restrict__ cu,
                                                            int seq0, int tok_base, int H, float scale, int q_limit,
                                                            int lds_keys, half_t* __restrict__ ctx_h) {
  constexpr int LDK = DH;       // halfs per staged row: unpadded, 16-byte chunks XOR-swizzled by the row instead
  constexpr int CH = DH / 8;    // 16-byte chunks per row
  // chunk c of K row r sits at c ^ kswz(r): the 16 rows x one chunk of a ds_read_b128 fragment read then cover all 64
  // banks once; chunk c of V row r at c ^ vswz(r): the 8 rows x 32 bytes of a transposed read (per 32-lane half) do
  auto kswz = [](int r) { return DH == 64 ? (r >> 1) & 7 : (r >> 1) & 3; };
  auto vswz = [](int r) { return DH == 64 ? ((r >> 1) & 3) << 1 : ((r >> 2) & 1) << 1; };
  constexpr int NS = DH / 16;   // 16-wide d blocks of the output
  constexpr int NKB = DH / 32;  // 32-deep MFMA steps over d
  extern __shared__ __attribute__((aligned(16))) half_t att_lds[];
  half_t* sK = att_lds;
  half_t* sV = att_lds + static_cast<size_t>(lds_keys) * LDK;
  // heads fastest: the twelve 128-byte slices of a token's Q/K/V row are then fetched by blocks that run side by side —
  // one DRAM page opened once — instead of 2200 blocks apart (sequence-fastest order: 3.2 TB/s of 128-byte reads)
  const int n_heads = H / DH;
  const int seq = seq0 + static_cast<int>(blockIdx.x) / n_heads;
  const int head = static_cast<int>(blockIdx.x) % n_heads;
  const int t0 = cu[seq] - tok_base;
  const int len = cu[seq + 1] - cu[seq];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 15, g = lane >> 4;
  const int64_t row3 = 3 * static_cast<int64_t>(H);
  // keys are staged and multiplied in groups of 16 (one MFMA tile): a 140-token sequence costs 9 groups, not 3 x 4
  const int keys_pad = min((len + 15) & ~15, lds_keys);  // (len <= lds_keys: the host sizes the LDS by the longest sequence)

  const int q_end = min(len, q_limit);
  // Q rows go straight to registers; a wave's first tile is requested before the K/V loads, every further one while
  // the tile before it is computed
  auto load_q = [&](int qt, f16x8 (&q)[NKB]) {
    const int tok = min(qt * 16 + qi, len - 1);
    const half_t* qp = qkv + (t0 + tok) * row3 + head * DH + 8 * g;
#pragma unroll
    for (int u = 0; u < NKB; ++u) q[u] = *reinterpret_cast<const f16x8*>(qp + 32 * u);
  };
  f16x8 qf[NKB];
  if (wave * 16 < q_end) load_q(wave, qf);
  // stage K and V rows 0 .. keys_pad - 1 (zeros behind the sequence's end), up to 128 keys' loads in flight at once
  const int total = keys_pad * CH;
  constexpr int NT = NW * 64;
  constexpr int NST = 20 / NW;  // 16-byte loads of K and of V per thread and round: 160 keys of 64 dims in one round
  for (int base = 0; base < total; base += NST * NT) {
    uint4 kv[NST], vv[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int idx = base + i * NT + tid;
      const int key = idx / CH, c8 = idx % CH;
      kv[i] = vv[i] = make_uint4(0, 0, 0, 0);
      if (idx < total && key < len) {
        const half_t* p = qkv + (t0 + key) * row3 + H + head * DH + c8 * 8;
        kv[i] = *reinterpret_cast<const uint4*>(p);
        vv[i] = *reinterpret_cast<const uint4*>(p + H);
      }
    }
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int idx = base + i * NT + tid;
      const int key = idx / CH, c8 = idx % CH;
      if (idx < total) {
        *reinterpret_cast<uint4*>(sK + key * LDK + (c8 ^ kswz(key)) * 8) = kv[i];
        *reinterpret_cast<uint4*>(sV + key * LDK + (c8 ^ vswz(key)) * 8) = vv[i];
      }
    }
  }
  __syncthreads();

  const float scale2 = scale * 1.4426950408889634f;  // logits in the log2 domain: the softmax uses v_exp_f32 directly
  // transposed V reads: lane 4q + p of its 16-lane group -> row (key) q, columns 4p .. 4p + 3 of the block
  // (block rows are key0 + q with key0 a multiple of 4: the swizzle of the row depends on 4 g + q alone)
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
  const int tr_row = tr_q * LDK, tr_sw = vswz(4 * g + tr_q), tr_in = 4 * (tr_p & 1);
  const int k_sw = kswz(qi);  // rows kt + 16 t + qi
  for (int qt = wave; qt * 16 < q_end; qt += NW) {  // wave-uniform: EXEC stays full (the transposed reads need it)
    const int q_tok = qt * 16 + qi;
    const bool q_valid = q_tok < len;
    f16x8 qn[NKB];
    const bool more = (qt + NW) * 16 < q_end;
    if (more) load_q(qt + NW, qn);
    f32x4 o[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) o[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -__builtin_inff();
    float l = 0.0f;
    for (int kt = 0; kt < keys_pad; kt += 64) {
      const int nt = min(4, (keys_pad - kt) >> 4);  // 16-key groups in this tile (block-uniform)
      f32x4 st[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < nt) {
#pragma unroll
          for (int u = 0; u < NKB; ++u) {
            const f16x8 kf = *reinterpret_cast<const f16x8*>(sK + (kt + t * 16 + qi) * LDK + ((4 * u + g) ^ k_sw) * 8);
            st[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[u], st[t], 0, 0, 0);
          }
        }
      }
      float mx = -__builtin_inff();
      if (kt + 64 <= len) {  // every key of the tile exists: no masks
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            st[t][r] *= scale2;
            mx = fmaxf(mx, st[t][r]);
          }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt + t * 16 + 4 * g + r;
            const float v = key < len ? st[t][r] * scale2 : -__builtin_inff();
            st[t][r] = v;
            mx = fmaxf(mx, v);
          }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m, mx);  // finite: key kt < len is always valid
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      float psum = 0.0f;
      f16x8 pb[2];  // pb[u][j]: key kt + 32 u + 16 (j >> 2) + 4 g + (j & 3)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(st[t][r] - m_new);
          psum += p;
          pb[t >> 1][(t & 1) * 4 + r] = static_cast<half_t>(p);
        }
      l = l * alpha + psum;
      m = m_new;
#pragma unroll
      for (int s = 0; s < NS; ++s) o[s] *= alpha;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          // V^T fragment: d = 16 s + qi, k index 8 g + j <-> the key of pb[u][j]: two 4-key blocks, 16 keys apart
          // (a group behind the staged keys contributes nothing: its P values are zero, its V rows are not read)
          if (2 * u >= nt) continue;
          const half_t* vb = sV + (kt + 32 * u + 4 * g) * LDK + tr_row + ((2 * s + (tr_p >> 1)) ^ tr_sw) * 8 + tr_in;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb));
          s16x4 hi = s16x4{0, 0, 0, 0};
          if (2 * u + 1 < nt) hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + 16 * LDK));
          f16x8 vf;
          *reinterpret_cast<s16x4*>(&vf) = lo;
          *(reinterpret_cast<s16x4*>(&vf) + 1) = hi;
          o[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pb[u], o[s], 0, 0, 0);
        }
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (q_valid) {
      const float inv = 1.0f / l;
      const int64_t off = static_cast<int64_t>(t0 + q_tok) * H + head * DH + 4 * g;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        half_t h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = static_cast<half_t>(fminf(fmaxf(o[s][r] * inv, -65504.0f), 65504.0f));
        *reinterpret_cast<uint2*>(ctx_h + off + 16 * s) = *reinterpret_cast<const uint2*>(h);
      }
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < NKB; ++u) qf[u] = qn[u];
    }
  }
}

// ---- host side -----------------------------------------------------------------------------------

static int dev_alloc_copy(vr_engine* e, Encoder* enc, const void* src, size_t n_floats, int mem, float** out) {
  float* p = nullptr;
  VR_HIP(hipMalloc(reinterpret_cast<void**>(&p), n_floats * sizeof(float)));
  enc->owned.push_back(p);
  if (src)
    VR_HIP(hipMemcpyAsync(p, src, n_floats * sizeof(float),
                          mem == VR_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, e->stream));
  *out = p;
  return 0;
}

// w (device, n floats) -> scaled (hi, lo) f16 pair; scale = 2^s puts max|w| into [1024, 2048)
// (mean, 1/sigma) of every row from the per-64-column partial sums of EPI_BIAS_RESIDUAL_LN_STATS
__global__ void ln_finalize_kernel(const float2* __restrict__ part, int T, int segs, int H, float eps,
                                   float2* __restrict__ stat) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  float s1 = 0.0f, s2 = 0.0f;
  for (int i = 0; i < segs; ++i) {  // fixed order
    const float2 p = part[static_cast<int64_t>(t) * segs + i];
    s1 += p.x;
    s2 += p.y;
  }
  const float mean = s1 / static_cast<float>(H);
  const float var = fmaxf(s2 / static_cast<float>(H) - mean * mean, 0.0f);
  stat[t] = make_float2(mean, 1.0f / sqrtf(var + eps));
}

// out[n][k] = w[n][k] * g[k]
__global__ void scale_cols_kernel(const float* __restrict__ w, const float* __restrict__ g, int64_t n, int K,
                                  float* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = w[i] * g[i % K];
}

// one wave per output column n: colsum[n] = unscale * sum_k f16 weight[n][k] (the weights the MFMA really
// multiplies by), c[n] = bias[n] + sum_k w[n][k] b[k] (f32 weights, f64 accumulation)
__global__ __launch_bounds__(256) void fold_vectors_kernel(const half_t* __restrict__ wq, float unscale,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           const float* __restrict__ bias, int N, int K,
                                                           float* __restrict__ colsum, float* __restrict__ c) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  double sq = 0.0, sc = 0.0;
  for (int k = lane; k < K; k += 64) {
    sq += static_cast<double>(static_cast<float>(wq[static_cast<int64_t>(n) * K + k]));
    sc += static_cast<double>(w[static_cast<int64_t>(n) * K + k]) * static_cast<double>(b[k]);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    sq += __shfl_xor(sq, off);
    sc += __shfl_xor(sc, off);
  }
  if (lane == 0) {
    colsum[n] = static_cast<float>(sq * static_cast<double>(unscale));
    c[n] = static_cast<float>(static_cast<double>(bias[n]) + sc);
  }
}

static int make_split(vr_engine* e, Encoder* enc, const float* w_dev, size_t n, int K, bool plain, SplitWeight* out) {
  float* scratch = nullptr;
  VR_TRY(dev_alloc_copy(e, enc, nullptr, 1, 0, &scratch));
  VR_HIP(hipMemsetAsync(scratch, 0, sizeof(float), e->stream));
  const unsigned blocks = static_cast<unsigned>(std::min<size_t>(1024, (n + 255) / 256));
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, e->stream, w_dev, static_cast<int64_t>(n),
                     reinterpret_cast<unsigned int*>(scratch));
  float m = 0.0f;
  VR_HIP(hipMemcpyAsync(&m, scratch, sizeof(float), hipMemcpyDeviceToHost, e->stream));
  VR_HIP(hipStreamSynchronize(e->stream));
  VR_CHECK(std::isfinite(m), "weight tensor holds a non-finite value");
  int s = 0;
  if (m > 0.0f) {
    int ex = 0;
    (void)std::frexp(m, &ex);  // m = f * 2^ex, f in [0.5, 1)
    s = 11 - ex;
  }
  const float scale = std::ldexp(1.0f, s);
  out->unscale = std::ldexp(1.0f, -s);
  float* both = nullptr;  // one interleaved array of 2n halfs (= n floats), see split_at; plain: n halfs
  VR_TRY(dev_alloc_copy(e, enc, nullptr, plain ? (n + 1) / 2 : n, 0, &both));
  out->hi = reinterpret_cast<_Float16*>(both);
  out->lo = plain ? nullptr : out->hi + 8;
  hipLaunchKernelGGL(split_weights_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, e->stream,
                     w_dev, static_cast<int64_t>(n), K, scale, out->hi, out->lo);
  VR_HIP(hipGetLastError());
  return 0;
}

void encoder_release(vr_engine* e) {
  Encoder* enc = static_cast<Encoder*>(e->encoder);
  if (!enc) return;
  for (float* p : enc->owned) (void)hipFree(p);
  enc->ids.release();
  enc->cu.release();
  enc->skinny_ws.release();
  invalidate_graphs(enc);
  enc->out.release();
  delete enc;
  e->encoder = nullptr;
}

int encoder_hidden(vr_engine* e) {
  Encoder* enc = static_cast<Encoder*>(e->encoder);
  return enc ? enc->d.hidden : 0;
}

int encoder_load(vr_engine* e, const vr_bert_desc* d, const void* const* t, int n_tensors, int mem) {
  VR_CHECK(d->struct_size == static_cast<int32_t>(sizeof(vr_bert_desc)), "vr_bert_desc size mismatch");
  const int H = d->hidden, I = d->intermediate, L = d->layers;
  VR_CHECK(L >= 1 && H >= 128 && H % 128 == 0 && H <= 1024, "hidden %d must be a multiple of 128 in 128..1024", H);
  VR_CHECK(I % 128 == 0 && I % BK == 0, "intermediate %d must be a multiple of 128", I);
  VR_CHECK(d->heads >= 1 && H % d->heads == 0 && (H / d->heads == 32 || H / d->heads == 64),
           "head size %d unsupported (32 or 64)", d->heads ? H / d->heads : 0);
  VR_CHECK(n_tensors == 5 + 16 * L, "expected %d tensors, got %d", 5 + 16 * L, n_tensors);
  VR_CHECK(d->pooling == 0 || d->pooling == 1, "pooling must be 0 (mean) or 1 (cls)");
  VR_CHECK(d->precision == VR_PRECISION_F32 || d->precision == VR_PRECISION_F16X3 || d->precision == VR_PRECISION_F16,
           "unknown precision %d", d->precision);
  for (int i = 0; i < n_tensors; ++i) VR_CHECK(t[i] != nullptr, "tensor %d is null", i);
  encoder_release(e);
  Encoder* enc = new Encoder();
  e->encoder = enc;
  enc->d = *d;
  const size_t HH = static_cast<size_t>(H) * H;
  VR_TRY(dev_alloc_copy(e, enc, t[0], static_cast<size_t>(d->vocab) * H, mem, &enc->word));
  VR_TRY(dev_alloc_copy(e, enc, t[1], static_cast<size_t>(d->max_pos) * H, mem, &enc->pos));
  VR_TRY(dev_alloc_copy(e, enc, t[2], static_cast<size_t>(d->type_vocab) * H, mem, &enc->type));
  VR_TRY(dev_alloc_copy(e, enc, t[3], H, mem, &enc->lng));
  VR_TRY(dev_alloc_copy(e, enc, t[4], H, mem, &enc->lnb));
  const hipMemcpyKind kind = mem == VR_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  for (int l = 0; l < L; ++l) {
    const void* const* w = t + 5 + 16 * l;  // q_w q_b k_w k_b v_w v_b o_w o_b ln1_g ln1_b i_w i_b f_w f_b ln2_g ln2_b
    LayerWeights lw{};
    VR_TRY(dev_alloc_copy(e, enc, nullptr, 3 * HH, mem, &lw.wqkv));
    VR_TRY(dev_alloc_copy(e, enc, nullptr, 3 * static_cast<size_t>(H), mem, &lw.bqkv));
    for (int p = 0; p < 3; ++p) {  // fused QKV projection: rows [0,H) = query, [H,2H) = key, [2H,3H) = value
      VR_HIP(hipMemcpyAsync(lw.wqkv + p * HH, w[2 * p], HH * sizeof(float), kind, e->stream));
      VR_HIP(hipMemcpyAsync(lw.bqkv + p * H, w[2 * p + 1], H * sizeof(float), kind, e->stream));
    }
    VR_TRY(dev_alloc_copy(e, enc, w[6], HH, mem, &lw.wo));
    VR_TRY(dev_alloc_copy(e, enc, w[7], H, mem, &lw.bo));
    VR_TRY(dev_alloc_copy(e, enc, w[8], H, mem, &lw.ln1g));
    VR_TRY(dev_alloc_copy(e, enc, w[9], H, mem, &lw.ln1b));
    VR_TRY(dev_alloc_copy(e, enc, w[10], static_cast<size_t>(I) * H, mem, &lw.w1));
    VR_TRY(dev_alloc_copy(e, enc, w[11], I, mem, &lw.b1));
    VR_TRY(dev_alloc_copy(e, enc, w[12], static_cast<size_t>(I) * H, mem, &lw.w2));
    VR_TRY(dev_alloc_copy(e, enc, w[13], H, mem, &lw.b2));
    VR_TRY(dev_alloc_copy(e, enc, w[14], H, mem, &lw.ln2g));
    VR_TRY(dev_alloc_copy(e, enc, w[15], H, mem, &lw.ln2b));
    if (d->precision != VR_PRECISION_F32) {
      const bool plain = d->precision == VR_PRECISION_F16;
      VR_TRY(make_split(e, enc, lw.wqkv, 3 * HH, H, plain, &lw.s_qkv));
      VR_TRY(make_split(e, enc, lw.wo, HH, H, plain, &lw.s_o));
      VR_TRY(make_split(e, enc, lw.w1, static_cast<size_t>(I) * H, H, plain, &lw.s_1));
      VR_TRY(make_split(e, enc, lw.w2, static_cast<size_t>(I) * H, I, plain, &lw.s_2));
      if (plain && H % 64 == 0) {  // operands of the folded-LayerNorm GEMMs
        auto fold = [&](const float* w_dev, int N, const float* g, const float* b, const float* bias, SplitWeight* sw,
                        float** colsum, float** c) -> int {
          const size_t n = static_cast<size_t>(N) * H;
          float* scaled = nullptr;
          VR_HIP(hipMalloc(reinterpret_cast<void**>(&scaled), n * sizeof(float)));
          hipLaunchKernelGGL(scale_cols_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, e->stream, w_dev,
                             g, static_cast<int64_t>(n), H, scaled);
          int rc = make_split(e, enc, scaled, n, H, true, sw);
          if (rc == 0) rc = dev_alloc_copy(e, enc, nullptr, N, 0, colsum);
          if (rc == 0) rc = dev_alloc_copy(e, enc, nullptr, N, 0, c);
          if (rc == 0)
            hipLaunchKernelGGL(fold_vectors_kernel, dim3(static_cast<unsigned>((N + 3) / 4)), dim3(256), 0, e->stream, sw->hi,
                               sw->unscale, w_dev, b, bias, N, H, *colsum, *c);
          (void)hipStreamSynchronize(e->stream);
          (void)hipFree(scaled);
          return rc;
        };
        VR_TRY(fold(lw.w1, I, lw.ln1g, lw.ln1b, lw.b1, &lw.s_1_f, &lw.cs_1, &lw.c_1));
        if (!enc->layers.empty()) {
          const LayerWeights& prev = enc->layers.back();
          VR_TRY(fold(lw.wqkv, 3 * H, prev.ln2g, prev.ln2b, lw.bqkv, &lw.s_qkv_f, &lw.cs_qkv, &lw.c_qkv));
        }
      }
    }
    enc->layers.push_back(lw);
  }
  VR_HIP(hipStreamSynchronize(e->stream));  // host sources may be freed by the caller now
  return 0;
}

static int ensure_workspace(vr_engine* e, Encoder* enc, int64_t tokens) {
  if (tokens <= enc->ws_tokens) return 0;
  // Headroom: batches of the same chunk count differ by a per cent or two in tokens, and a workspace that fits the
  // largest batch SO FAR exactly is freed and reallocated (gigabytes, with a device synchronisation) every time a
  // slightly longer one arrives — 200 ms per step at the bge-large shape. A sixteenth more, in whole 4096-token units.
  if (tokens > 4096) tokens = (tokens + tokens / 16 + 4095) / 4096 * 4096;
  VR_HIP(hipStreamSynchronize(e->stream));
  invalidate_graphs(enc);
  for (float** p : {&enc->x, &enc->qkv, &enc->ctx, &enc->tmp, &enc->ffn, &enc->xs, &enc->lnstat, &enc->lnpart}) {
    if (*p) {
      enc->owned.erase(std::remove(enc->owned.begin(), enc->owned.end(), *p), enc->owned.end());
      (void)hipFree(*p);
      *p = nullptr;
    }
  }
  const size_t H = static_cast<size_t>(enc->d.hidden), I = static_cast<size_t>(enc->d.intermediate);
  const size_t T = static_cast<size_t>(tokens);
  VR_TRY(dev_alloc_copy(e, enc, nullptr, T * H, 0, &enc->x));
  VR_TRY(dev_alloc_copy(e, enc, nullptr, T * 3 * H, 0, &enc->qkv));
  VR_TRY(dev_alloc_copy(e, enc, nullptr, T * H, 0, &enc->ctx));
  VR_TRY(dev_alloc_copy(e, enc, nullptr, T * H, 0, &enc->tmp));
  VR_TRY(dev_alloc_copy(e, enc, nullptr, T * I, 0, &enc->ffn));
  if (enc->d.precision != VR_PRECISION_F32) VR_TRY(dev_alloc_copy(e, enc, nullptr, T * H, 0, &enc->xs));
  if (enc->d.precision == VR_PRECISION_F16) VR_TRY(dev_alloc_copy(e, enc, nullptr, T * 4, 0, &enc->lnstat));
  if (enc->d.precision == VR_PRECISION_F16) VR_TRY(dev_alloc_copy(e, enc, nullptr, T * ((H + 63) / 64) * 2, 0, &enc->lnpart));
  enc->ws_tokens = tokens;
  return 0;
}

// forward of sequences [seq0, seq1) whose tokens are ids_dev[tok_base .. tok_base + T)
static int forward_chunk(vr_engine* e, Encoder* enc, const int32_t* ids_dev, const int32_t* cu_dev,
                         int n_seq_total, int seq0, int seq1, int tok_base, int T, int max_len,
                         double attn_flop, float* out_dev) {
  const vr_bert_desc& d = enc->d;
  const int H = d.hidden, I = d.intermediate, nh = d.heads, dh = H / nh;
  hipStream_t s = e->stream;
  const unsigned row_blocks = static_cast<unsigned>((T + 3) / 4);
  // f16x3 mode: every GEMM input also exists as (hi, lo) f16. The split hidden state lives in xs;
  // the context and the FFN intermediate are ONLY needed split, so they reuse the f32 buffers.
  // f16 mode: the same, with plain f16 rows and no lo half (the lo pointers are null).
  const bool split = d.precision != VR_PRECISION_F32;
  const bool plain = d.precision == VR_PRECISION_F16;
  const int passes = plain ? 1 : 3;
  half_t* xh = split ? reinterpret_cast<half_t*>(enc->xs) : nullptr;
  half_t* xl = split && !plain ? xh + 8 : nullptr;  // interleaved (hi, lo) layout, see split_at
  half_t* ch = split ? reinterpret_cast<half_t*>(enc->ctx) : nullptr;
  half_t* cl = split && !plain ? ch + 8 : nullptr;
  half_t* fh = split ? reinterpret_cast<half_t*>(enc->ffn) : nullptr;
  half_t* fl = split && !plain ? fh + 8 : nullptr;
  // f16 mode does not store the f32 hidden state at all. A LayerNorm writes its f16 output (the next
  // GEMM's input) and the row's (mean, 1/sigma); the only other reader of the hidden state is the
  // residual add of the next projection's epilogue, which already reads 4 bytes per element — it
  // reads the PRE-LN row instead and re-derives the LayerNorm output (ln_apply: same operations,
  // same bits). That takes 4 of the 10 bytes per element out of every LayerNorm pass. The pre-LN
  // rows alternate between enc->tmp and enc->x; `cur` describes the current hidden state.
  struct Hidden {
    const float* pre;
    const float2* stat;
    const float* g;
    const float* b;
  };
  float2* stat_a = reinterpret_cast<float2*>(enc->lnstat);
  float2* stat_b = plain ? stat_a + T : nullptr;
  Hidden cur{enc->x, stat_b, enc->lng, enc->lnb};
  bool cur16 = false;  // f16 residual stream: the current pre-LN rows are the f16 rows in xh (cur.pre is stale)
  const float* final_x = enc->x;  // the f32 rows pooling reads
  const bool lnfuse = plain && !enc->layers.empty();  // (a model without layers pools the embedding LayerNorm's output)
  // a handful of tokens (one query): the two LayerNorm launches of a layer are folded into the projections
  // that consume them (gemm_f16_skinny_ln_kernel); VR_ENCODE_FOLD_LN=0 keeps them apart
  static const bool fold_ln_enabled = !(getenv("VR_ENCODE_FOLD_LN") && atoi(getenv("VR_ENCODE_FOLD_LN")) == 0);
  const bool fold_ln = lnfuse && fold_ln_enabled && skinny_ln_supported(T, 3 * H, H) && I % 16 == 0;
  // large batches (the 256-tile GEMM): the LayerNorm passes disappear into the GEMMs on both sides of them —
  // the producing epilogue also stores f16 pre-LN rows and per-(row, 64 columns) partial sums, a tiny
  // kernel turns those into (mean, 1/sigma), and the consuming GEMM multiplies the pre-LN rows by the
  // gain-scaled weights and applies the statistics in its epilogue (EPI_FOLD_*). VR_ENCODE_FOLD_GEMM=0: off.
  static const bool fold_big_enabled = !(getenv("VR_ENCODE_FOLD_GEMM") && atoi(getenv("VR_ENCODE_FOLD_GEMM")) == 0);
  const bool fold_big = lnfuse && fold_big_enabled && T > 4 * kSkinnyM && H % 64 == 0 && enc->layers[0].cs_1 != nullptr;
  // ... and with an f16 residual stream (EPI_RLS_*): the pre-LayerNorm rows live in xh only. VR_ENCODE_RES16=0: f32 rows
  static const bool res16_enabled = !(getenv("VR_ENCODE_RES16") && atoi(getenv("VR_ENCODE_RES16")) == 0);
  const bool res16 = fold_big && res16_enabled && direct_usable(H, H) && direct_usable(H, I);
  float2* part = reinterpret_cast<float2*>(enc->lnpart);
  const int segs = H / 64;
  const unsigned fin_blocks = static_cast<unsigned>((T + 255) / 256);
  hipLaunchKernelGGL(embed_ln_kernel, dim3(row_blocks), dim3(256), 0, s, ids_dev, cu_dev, n_seq_total,
                     tok_base, T, H, d.vocab, enc->word, enc->pos, enc->type, enc->lng, enc->lnb, d.eps,
                     lnfuse ? nullptr : enc->x, xh, xl, lnfuse ? enc->x : nullptr, lnfuse ? stat_b : nullptr);
  const int qblocks = (max_len + 63) / 64;
  const float scale = 1.0f / sqrtf(static_cast<float>(dh));
  // CLS pooling reads one row per sequence, so everything after the LAST layer's attention is needed
  // for those rows only: their context rows (and residual rows) are gathered into compact [n_seq, *]
  // matrices carved out of the qkv buffer (free once attention has run), and the output projection,
  // both LayerNorms and the FFN run on n_seq rows instead of T. Every row's arithmetic is unchanged
  // (a GEMM row does not depend on its neighbours), so the embeddings are bit-identical; for
  // bge-base this removes 10/12 of the last layer's GEMM work. The Q projection of the other
  // tokens is still computed (it is one GEMM with K and V).
  const int n_seq = seq1 - seq0;
  const bool cls_tail = d.pooling == VR_POOL_CLS && !enc->layers.empty() &&
                        static_cast<int64_t>(n_seq) * (4 * H + I) <= static_cast<int64_t>(T) * 3 * H;
  float* xc = enc->qkv;                                       // [n_seq, H] hidden state of the [CLS] rows
  float* tmpc = xc + static_cast<int64_t>(n_seq) * H;         // [n_seq, H]
  float* xsc = tmpc + static_cast<int64_t>(n_seq) * H;        // [n_seq, 2H] halfs (split) = n_seq*H floats
  float* ctxc = xsc + static_cast<int64_t>(n_seq) * H;        // [n_seq, H] f32 or [n_seq, 2H] halfs
  float* ffnc = ctxc + static_cast<int64_t>(n_seq) * H;       // [n_seq, I] f32 or [n_seq, 2I] halfs
  for (size_t li = 0; li < enc->layers.size(); ++li) {
    const LayerWeights& w = enc->layers[li];
    const bool tail = cls_tail && li + 1 == enc->layers.size();
    if (plain && fold_ln && li > 0)  // the previous layer's closing LayerNorm runs inside this projection
      VR_TRY(launch_skinny_ln(e, EPI_BIAS_F16, cur.pre, cur.g, cur.b, d.eps, const_cast<float2*>(cur.stat), w.s_qkv.hi,
                              w.s_qkv.unscale, w.bqkv, reinterpret_cast<half_t*>(enc->qkv), T, 3 * H, H));
    else if (plain && fold_big && li > 0)  // xh holds the f16 PRE-LN rows the previous FFN-down epilogue stored
      VR_TRY(launch_gemm_f16x3(e, EPI_FOLD_F16, xh, nullptr, w.s_qkv_f.hi, nullptr, w.s_qkv_f.unscale, w.c_qkv, nullptr,
                               nullptr, reinterpret_cast<half_t*>(enc->qkv), nullptr, T, 3 * H, H, 1, cur.stat, w.cs_qkv,
                               nullptr));
    else if (plain)  // Q, K, V as plain f16 rows for attention_f16_kernel
      VR_TRY(launch_gemm_f16x3(e, EPI_BIAS_F16, xh, xl, w.s_qkv.hi, w.s_qkv.lo, w.s_qkv.unscale, w.bqkv, nullptr,
                               nullptr, reinterpret_cast<half_t*>(enc->qkv), nullptr, T, 3 * H, H, passes));
    else if (split)
      VR_TRY(launch_gemm_f16x3(e, EPI_BIAS, xh, xl, w.s_qkv.hi, w.s_qkv.lo, w.s_qkv.unscale, w.bqkv, nullptr,
                               enc->qkv, nullptr, nullptr, T, 3 * H, H, passes));
    else
      VR_TRY(launch_gemm(e, EPI_BIAS, enc->x, w.wqkv, w.bqkv, nullptr, enc->qkv, T, 3 * H, H));
    const int qb = tail ? 1 : qblocks;  // tail: only the query block that holds token 0 of every sequence
    dim3 agrid(static_cast<unsigned>(n_seq * qb), static_cast<unsigned>(nh));
    prof_begin(e, VR_PROF_ATTENTION, tail ? attn_flop / qblocks : attn_flop);
    static const bool attn_seq = !(getenv("VR_ATTN_SEQ") && atoi(getenv("VR_ATTN_SEQ")) == 0);
    if (plain && attn_seq && (dh == 64 || dh == 32)) {  // one block per (sequence, head): K/V staged once
      const int lds_keys = (max_len + 15) & ~15;
      const size_t lds_bytes = static_cast<size_t>(2) * lds_keys * dh * sizeof(half_t);
      static size_t lds_allowed[2] = {0, 0};  // per instantiation: raised above the 64 KiB default when a sequence needs it
      const int which = dh == 64 ? 0 : 1;
      if (lds_bytes > 65536 && lds_bytes > lds_allowed[which]) {
        if (dh == 64)
          VR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_seq_kernel<64, 4>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes)));
        else
          VR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_seq_kernel<32, 4>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes)));
        lds_allowed[which] = lds_bytes;
      }
      const dim3 sgrid(static_cast<unsigned>(n_seq) * static_cast<unsigned>(nh));
#ifdef VR_GEMM_DIAG_BUILD  // (timing experiment: VR_ATTN_QLIMIT=16 leaves the staging and one query tile per block)
      static const int q_limit_env = getenv("VR_ATTN_QLIMIT") ? atoi(getenv("VR_ATTN_QLIMIT")) : 0;
      const int q_limit = tail ? 16 : (q_limit_env > 0 ? q_limit_env : max_len);
#else
      const int q_limit = tail ? 16 : max_len;
#endif
      static const int attn_waves = getenv("VR_ATTN_WAVES") ? atoi(getenv("VR_ATTN_WAVES")) : 4;  // (experiment switch: 4 or 8)
#define VR_ATTN_SEQ(DHV, NWV)                                                                                                   \
  hipLaunchKernelGGL((attention_seq_kernel<DHV, NWV>), sgrid, dim3(NWV * 64), lds_bytes, s, reinterpret_cast<const half_t*>(enc->qkv), \
                     cu_dev, seq0, tok_base, H, scale, q_limit, lds_keys, ch)
      if (dh == 64 && attn_waves == 8) VR_ATTN_SEQ(64, 8);
      else if (dh == 64) VR_ATTN_SEQ(64, 4);
      else if (attn_waves == 8) VR_ATTN_SEQ(32, 8);
      else VR_ATTN_SEQ(32, 4);
#undef VR_ATTN_SEQ
    } else if (plain && dh == 64)  // (8 waves = 128 queries per block stage K/V once per 128-token sequence, and measured 30 % slower)
      hipLaunchKernelGGL((attention_f16_kernel<64, 4>), agrid, dim3(256), 0, s, reinterpret_cast<const half_t*>(enc->qkv),
                         cu_dev, seq0, tok_base, H, qb, scale, ch);
    else if (plain)
      hipLaunchKernelGGL((attention_f16_kernel<32, 4>), agrid, dim3(256), 0, s, reinterpret_cast<const half_t*>(enc->qkv),
                         cu_dev, seq0, tok_base, H, qb, scale, ch);
    else if (dh == 64)
      hipLaunchKernelGGL((attention_kernel<64>), agrid, dim3(256), 0, s, enc->qkv, cu_dev, seq0, tok_base, H,
                         qb, scale, enc->ctx, ch, cl);
    else
      hipLaunchKernelGGL((attention_kernel<32>), agrid, dim3(256), 0, s, enc->qkv, cu_dev, seq0, tok_base, H,
                         qb, scale, enc->ctx, ch, cl);
    prof_end(e);
    if (tail) {
      const unsigned gblocks = static_cast<unsigned>((n_seq + 3) / 4);
      const unsigned cblocks = gblocks;
      hipLaunchKernelGGL(gather_first_rows_kernel, dim3(gblocks), dim3(256), 0, s, reinterpret_cast<const float4*>(enc->ctx),
                         cu_dev, seq0, tok_base, n_seq, T, plain ? H / 8 : H / 4, reinterpret_cast<float4*>(ctxc));
      if (plain && cur16)
        hipLaunchKernelGGL(gather_first_rows_ln16_kernel, dim3(gblocks), dim3(256), 0, s, xh, cur.stat, cur.g, cur.b,
                           cu_dev, seq0, tok_base, n_seq, T, H, xc);
      else if (plain)
        hipLaunchKernelGGL(gather_first_rows_ln_kernel, dim3(gblocks), dim3(256), 0, s, cur.pre, cur.stat, cur.g, cur.b,
                           cu_dev, seq0, tok_base, n_seq, T, H, xc);
      else
        hipLaunchKernelGGL(gather_first_rows_kernel, dim3(gblocks), dim3(256), 0, s, reinterpret_cast<const float4*>(enc->x),
                           cu_dev, seq0, tok_base, n_seq, T, H / 4, reinterpret_cast<float4*>(xc));
      half_t* xch = reinterpret_cast<half_t*>(xsc);
      half_t* cch = reinterpret_cast<half_t*>(ctxc);
      half_t* fch = reinterpret_cast<half_t*>(ffnc);
      if (split)
        VR_TRY(launch_gemm_f16x3(e, EPI_BIAS_RESIDUAL, cch, plain ? nullptr : cch + 8, w.s_o.hi, w.s_o.lo, w.s_o.unscale, w.bo, xc, tmpc,
                                 nullptr, nullptr, n_seq, H, H, passes));
      else
        VR_TRY(launch_gemm(e, EPI_BIAS_RESIDUAL, ctxc, w.wo, w.bo, xc, tmpc, n_seq, H, H));
      hipLaunchKernelGGL(layernorm_kernel, dim3(cblocks), dim3(256), 0, s, tmpc, n_seq, H, w.ln1g, w.ln1b, d.eps, xc,
                         split ? xch : nullptr, split && !plain ? xch + 8 : nullptr, static_cast<float2*>(nullptr));
      if (split) {
        VR_TRY(launch_gemm_f16x3(e, EPI_BIAS_GELU, xch, plain ? nullptr : xch + 8, w.s_1.hi, w.s_1.lo, w.s_1.unscale, w.b1, nullptr, nullptr,
                                 fch, plain ? nullptr : fch + 8, n_seq, I, H, passes));
        VR_TRY(launch_gemm_f16x3(e, EPI_BIAS_RESIDUAL, fch, plain ? nullptr : fch + 8, w.s_2.hi, w.s_2.lo, w.s_2.unscale, w.b2, xc, tmpc,
                                 nullptr, nullptr, n_seq, H, I, passes));
      } else {
        VR_TRY(launch_gemm(e, EPI_BIAS_GELU, xc, w.w1, w.b1, nullptr, ffnc, n_seq, I, H));
        VR_TRY(launch_gemm(e, EPI_BIAS_RESIDUAL, ffnc, w.w2, w.b2, xc, tmpc, n_seq, H, I));
      }
      hipLaunchKernelGGL(layernorm_kernel, dim3(cblocks), dim3(256), 0, s, tmpc, n_seq, H, w.ln2g, w.ln2b, d.eps, xc,
                         static_cast<half_t*>(nullptr), static_cast<half_t*>(nullptr), static_cast<float2*>(nullptr));
      hipLaunchKernelGGL(pool_kernel, dim3(static_cast<unsigned>(n_seq)), dim3(256), 0, s, xc, cu_dev, seq0, tok_base, H,
                         d.pooling, d.normalize, 1, out_dev);
      VR_HIP(hipGetLastError());
      return 0;
    }
    if (plain) {
      // hidden state = LN(cur.pre): pre-LN rows alternate between the two buffers
      float* t1 = cur.pre == enc->x ? enc->tmp : enc->x;
      float2* s1 = cur.stat == stat_b ? stat_a : stat_b;
      const bool last = li + 1 == enc->layers.size();
      if (res16) {
        // f16 residual stream: residual rows are f32 only in layer 0 (the embedding sum), the pre-LN rows go out
        // as f16 into xh, in place where the residual came from there
        VR_TRY(launch_gemm_f16x3(e, cur16 ? EPI_RLS_R16_O16 : EPI_RLS_R32_O16, ch, nullptr, w.s_o.hi, nullptr,
                                 w.s_o.unscale, w.bo, cur16 ? reinterpret_cast<const float*>(xh) : cur.pre, nullptr, xh,
                                 reinterpret_cast<half_t*>(part), T, H, H, 1, cur.stat, cur.g, cur.b));
        hipLaunchKernelGGL(ln_finalize_kernel, dim3(fin_blocks), dim3(256), 0, s, part, T, segs, H, d.eps, s1);
      } else if (fold_big) {
        VR_TRY(launch_gemm_f16x3(e, EPI_BIAS_RESIDUAL_LN_STATS, ch, nullptr, w.s_o.hi, nullptr, w.s_o.unscale, w.bo, cur.pre,
                                 t1, xh, reinterpret_cast<half_t*>(part), T, H, H, 1, cur.stat, cur.g, cur.b));
        hipLaunchKernelGGL(ln_finalize_kernel, dim3(fin_blocks), dim3(256), 0, s, part, T, segs, H, d.eps, s1);
      } else
      VR_TRY(launch_gemm_f16x3(e, EPI_BIAS_RESIDUAL_LN, ch, nullptr, w.s_o.hi, nullptr, w.s_o.unscale, w.bo, cur.pre, t1,
                               nullptr, nullptr, T, H, H, 1, cur.stat, cur.g, cur.b));
      if (fold_ln || fold_big)
        ;  // the attention-output LayerNorm runs inside the FFN-up projection below
      else if (H % 256 == 0)
        hipLaunchKernelGGL(layernorm_f16_kernel, dim3(row_blocks), dim3(256), 0, s, t1, T, H, w.ln1g, w.ln1b, d.eps,
                           static_cast<float*>(nullptr), xh, s1);
      else
        hipLaunchKernelGGL(layernorm_kernel, dim3(row_blocks), dim3(256), 0, s, t1, T, H, w.ln1g, w.ln1b, d.eps,
                           static_cast<float*>(nullptr), xh, static_cast<half_t*>(nullptr), s1);
      const Hidden mid{t1, s1, w.ln1g, w.ln1b};
      float* t2 = const_cast<float*>(cur.pre);  // its last reader (the epilogue above) is done
      float2* s2 = const_cast<float2*>(cur.stat);
      if (fold_ln)
        VR_TRY(launch_skinny_ln(e, EPI_BIAS_GELU, t1, w.ln1g, w.ln1b, d.eps, s1, w.s_1.hi, w.s_1.unscale, w.b1, fh, T, I, H));
      else if (fold_big)
        VR_TRY(launch_gemm_f16x3(e, EPI_FOLD_GELU, xh, nullptr, w.s_1_f.hi, nullptr, w.s_1_f.unscale, w.c_1, nullptr, nullptr,
                                 fh, nullptr, T, I, H, 1, s1, w.cs_1, nullptr));
      else
      VR_TRY(launch_gemm_f16x3(e, EPI_BIAS_GELU, xh, nullptr, w.s_1.hi, nullptr, w.s_1.unscale, w.b1, nullptr, nullptr, fh,
                               nullptr, T, I, H, 1));
      if (res16)  // (the last layer of a mean-pooled model also writes f32 rows: the final LayerNorm reads them)
        VR_TRY(launch_gemm_f16x3(e, last ? EPI_RLS_R16_O32 : EPI_RLS_R16_O16, fh, nullptr, w.s_2.hi, nullptr, w.s_2.unscale,
                                 w.b2, reinterpret_cast<const float*>(xh), last ? t2 : nullptr, xh,
                                 reinterpret_cast<half_t*>(part), T, H, I, 1, mid.stat, mid.g, mid.b));
      else if (fold_big)
        VR_TRY(launch_gemm_f16x3(e, EPI_BIAS_RESIDUAL_LN_STATS, fh, nullptr, w.s_2.hi, nullptr, w.s_2.unscale, w.b2, mid.pre,
                                 t2, xh, reinterpret_cast<half_t*>(part), T, H, I, 1, mid.stat, mid.g, mid.b));
      else
      VR_TRY(launch_gemm_f16x3(e, EPI_BIAS_RESIDUAL_LN, fh, nullptr, w.s_2.hi, nullptr, w.s_2.unscale, w.b2, mid.pre, t2,
                               nullptr, nullptr, T, H, I, 1, mid.stat, mid.g, mid.b));
      // the last LayerNorm of the network also stores its f32 rows (into the free buffer): pooling reads them
      if (fold_ln && !last)
        ;  // this layer's closing LayerNorm runs inside the next layer's Q/K/V projection
      else if (fold_big && !last)  // ... its statistics only
        hipLaunchKernelGGL(ln_finalize_kernel, dim3(fin_blocks), dim3(256), 0, s, part, T, segs, H, d.eps, s2);
      else if (H % 256 == 0)
        hipLaunchKernelGGL(layernorm_f16_kernel, dim3(row_blocks), dim3(256), 0, s, t2, T, H, w.ln2g, w.ln2b, d.eps,
                           last ? t1 : static_cast<float*>(nullptr), xh, s2);
      else
        hipLaunchKernelGGL(layernorm_kernel, dim3(row_blocks), dim3(256), 0, s, t2, T, H, w.ln2g, w.ln2b, d.eps,
                           last ? t1 : static_cast<float*>(nullptr), xh, static_cast<half_t*>(nullptr), s2);
      cur16 = res16;
      cur = Hidden{t2, s2, w.ln2g, w.ln2b};
      if (last) final_x = t1;
      continue;
    }
    if (split)
      VR_TRY(launch_gemm_f16x3(e, EPI_BIAS_RESIDUAL, ch, cl, w.s_o.hi, w.s_o.lo, w.s_o.unscale, w.bo, enc->x,
                               enc->tmp, nullptr, nullptr, T, H, H, passes));
    else
      VR_TRY(launch_gemm(e, EPI_BIAS_RESIDUAL, enc->ctx, w.wo, w.bo, enc->x, enc->tmp, T, H, H));
    hipLaunchKernelGGL(layernorm_kernel, dim3(row_blocks), dim3(256), 0, s, enc->tmp, T, H, w.ln1g, w.ln1b,
                       d.eps, enc->x, xh, xl, static_cast<float2*>(nullptr));
    if (split) {
      VR_TRY(launch_gemm_f16x3(e, EPI_BIAS_GELU, xh, xl, w.s_1.hi, w.s_1.lo, w.s_1.unscale, w.b1, nullptr,
                               nullptr, fh, fl, T, I, H, passes));
      VR_TRY(launch_gemm_f16x3(e, EPI_BIAS_RESIDUAL, fh, fl, w.s_2.hi, w.s_2.lo, w.s_2.unscale, w.b2, enc->x,
                               enc->tmp, nullptr, nullptr, T, H, I, passes));
    } else {
      VR_TRY(launch_gemm(e, EPI_BIAS_GELU, enc->x, w.w1, w.b1, nullptr, enc->ffn, T, I, H));
      VR_TRY(launch_gemm(e, EPI_BIAS_RESIDUAL, enc->ffn, w.w2, w.b2, enc->x, enc->tmp, T, H, I));
    }
    hipLaunchKernelGGL(layernorm_kernel, dim3(row_blocks), dim3(256), 0, s, enc->tmp, T, H, w.ln2g, w.ln2b,
                       d.eps, enc->x, xh, xl, static_cast<float2*>(nullptr));
  }
  hipLaunchKernelGGL(pool_kernel, dim3(static_cast<unsigned>(seq1 - seq0)), dim3(256), 0, s, final_x, cu_dev,
                     seq0, tok_base, H, d.pooling, d.normalize, 0, out_dev);
  VR_HIP(hipGetLastError());
  return 0;
}

// Tokens per forward pass. Large enough that the 768-wide GEMMs launch many full rounds of 256x256
// tiles (262144 tokens x 768 columns = 3072 tiles on 256 CUs: the partly filled last round is then
// a few per cent of the launch); activations for it are ~45 KB per token (11.8 GB of the 288).
// VR_CHUNK_TOKENS overrides it for experiments.
static const int64_t kMaxChunkTokens = getenv("VR_CHUNK_TOKENS") ? atoll(getenv("VR_CHUNK_TOKENS")) : 262144;

int encoder_encode(vr_engine* e, const int32_t* ids, const int32_t* offsets, int n_seq, int mem,
                   float* out, int out_mem) {
  Encoder* enc = static_cast<Encoder*>(e->encoder);
  VR_CHECK(enc != nullptr, "no encoder loaded (vr_encoder_load)");
  if (n_seq <= 0) return 0;
  const int H = enc->d.hidden;
  // offsets are needed on the host to cut chunks
  std::vector<int32_t> cu_host(static_cast<size_t>(n_seq) + 1);
  const int32_t* ids_dev = ids;
  const int32_t* cu_dev = offsets;
  if (mem == VR_MEM_HOST) {
    memcpy(cu_host.data(), offsets, sizeof(int32_t) * cu_host.size());
  } else {
    VR_HIP(hipMemcpyAsync(cu_host.data(), offsets, sizeof(int32_t) * cu_host.size(), hipMemcpyDeviceToHost,
                          e->stream));
    VR_HIP(hipStreamSynchronize(e->stream));
  }
  VR_CHECK(cu_host[0] == 0, "offsets must start at 0");
  const int64_t T_all = cu_host[static_cast<size_t>(n_seq)];
  for (int i = 0; i < n_seq; ++i) {
    int len = cu_host[static_cast<size_t>(i) + 1] - cu_host[static_cast<size_t>(i)];
    VR_CHECK(len >= 1 && len <= enc->d.max_pos, "sequence %d has %d tokens (1..%d allowed)", i, len,
             enc->d.max_pos);
  }
  if (mem == VR_MEM_HOST) {
    VR_TRY(enc->ids.grow(T_all, 0, e->stream));
    VR_TRY(enc->cu.grow(n_seq + 1, 0, e->stream));
    VR_HIP(hipMemcpyAsync(enc->ids.p, ids, sizeof(int32_t) * static_cast<size_t>(T_all), hipMemcpyHostToDevice,
                          e->stream));
    VR_HIP(hipMemcpyAsync(enc->cu.p, offsets, sizeof(int32_t) * cu_host.size(), hipMemcpyHostToDevice,
                          e->stream));
    ids_dev = enc->ids.p;
    cu_dev = enc->cu.p;
  }
  float* out_dev = out;
  if (out_mem == VR_MEM_HOST) {
    VR_TRY(enc->out.grow(static_cast<int64_t>(n_seq) * H, 0, e->stream));
    out_dev = enc->out.p;
  }
  int seq0 = 0;
  while (seq0 < n_seq) {
    int seq1 = seq0;
    int max_len = 0;
    int64_t T = 0;
    double len2 = 0.0;
    while (seq1 < n_seq) {
      int len = cu_host[static_cast<size_t>(seq1) + 1] - cu_host[static_cast<size_t>(seq1)];
      if (seq1 > seq0 && T + len > kMaxChunkTokens) break;
      T += len;
      len2 += static_cast<double>(len) * len;
      max_len = std::max(max_len, len);
      ++seq1;
    }
    VR_TRY(ensure_workspace(e, enc, std::max<int64_t>(T, 1024)));
    // A small forward pass (one chunk, <= 1024 tokens) is launch-bound: the second time a shape is seen
    // it is captured into a hipGraph and replayed from then on. Only on the engine's own stream (a
    // caller's stream may be under capture itself) and with the HIP-event profiler off.
    static const bool graphs_on = !(getenv("VR_ENCODE_GRAPH") && atoi(getenv("VR_ENCODE_GRAPH")) == 0);
    const bool graphable = graphs_on && seq0 == 0 && seq1 == n_seq && T <= 1024 && e->stream == e->own_stream &&
                           !prof_on(e);
    bool done = false;
    if (graphable) {
      if (enc->graphs.size() > 256) invalidate_graphs(enc);  // (shapes are few in practice; start over rather than track recency)
      Encoder::GraphEntry& g = enc->graphs[Encoder::GraphKey{static_cast<int>(T), n_seq, max_len, ids_dev, cu_dev, out_dev}];
      if (!g.exec && !g.bad && g.seen >= 1) {
        if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
          const int rc = forward_chunk(e, enc, ids_dev, cu_dev, n_seq, seq0, seq1, 0, static_cast<int>(T), max_len,
                                       4.0 * H * len2, out_dev);
          hipGraph_t graph = nullptr;
          const hipError_t end = hipStreamEndCapture(e->stream, &graph);
          if (rc != 0 || end != hipSuccess || !graph ||
              hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            g.exec = nullptr;
            g.bad = true;  // this shape runs eagerly from now on
          }
          if (graph) (void)hipGraphDestroy(graph);
          (void)hipGetLastError();
        } else {
          g.bad = true;
          (void)hipGetLastError();
        }
      }
      if (g.exec) {
        VR_HIP(hipGraphLaunch(g.exec, e->stream));
        done = true;
      }
      ++g.seen;
    }
    if (!done)
      VR_TRY(forward_chunk(e, enc, ids_dev, cu_dev, n_seq, seq0, seq1, cu_host[static_cast<size_t>(seq0)],
                           static_cast<int>(T), max_len, 4.0 * H * len2, out_dev));
    seq0 = seq1;
  }
  if (out_mem == VR_MEM_HOST) {
    VR_HIP(hipMemcpyAsync(out, out_dev, sizeof(float) * static_cast<size_t>(n_seq) * H, hipMemcpyDeviceToHost,
                          e->stream));
    VR_HIP(hipStreamSynchronize(e->stream));
  } else if (mem == VR_MEM_HOST) {
    VR_HIP(hipStreamSynchronize(e->stream));
  }
  return 0;
}

}  // namespace vr
