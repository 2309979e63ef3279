/*
 * TEST INFRASTRUCTURE — CPU oracle for the numeric part of the hot path. Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this. It is never part of
 * the product path (the product has no CPU fallback).
 *
 * What it restates
 *   The reference delegates the arithmetic of this path to a Qdrant server reached over HTTP
 *   (src/voitta/services/vector_store.py:71,313,612-617,640-656); that server is not under
 *   /root/reference and is not pinned (docker-compose.yml:3 "qdrant/qdrant", no tag), so this
 *   file restates the server's published behaviour for the calls the reference makes:
 *     - cosine collection  (vector_store.py:91-94): vectors are L2-normalised on insert and
 *       the query is normalised the same way; score = dot product               [EXT]
 *     - sparse "bm25" vector with Modifier.IDF (vector_store.py:95-99):
 *       score(d) = sum_t (q_t * idf_t) * d_t, idf_t = ln(1 + (N-df+0.5)/(df+0.5)) [EXT]
 *     - query_points(limit=k): the k best by score                              [EXT]
 *   PARITY UNPINNED for the [EXT] details: the reference's tests (tests/test_api.py) hold no
 *   vector, score or ranking fixture; the oracle is pinned only by the
 *   hand-derived known answers in tests/golden/.
 *
 * Conventions fixed here (and mirrored bit-for-bit by the HIP path)
 *   - length2 is accumulated sequentially in f32, multiply and add rounded separately
 *   - a vector with length2 < FLT_EPSILON or |length2 - 1| <= 1e-6 is stored unchanged
 *   - dense score = fmaf chain over k = 0..D-1 starting from +0.0f
 *   - sparse score = ascending-token-id sum, every product and sum rounded to f32
 *   - ranking: score descending, then row ascending; -inf never ranks
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (oracle/build.py). -ffp-contract=off matters:
 * the conventions above distinguish fused from unfused operations.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void oracle_cosine_preprocess(const float* x, int64_t n, int32_t d, float* out) {
  for (int64_t r = 0; r < n; ++r) {
    const float* p = x + r * d;
    float* o = out + r * d;
    float acc = 0.0f;
    for (int32_t k = 0; k < d; ++k) {
      float sq = p[k] * p[k];
      acc = acc + sq;
    }
    int keep = (acc < FLT_EPSILON) || (fabsf(acc - 1.0f) <= 1.0e-6f);
    if (keep) {
      memcpy(o, p, sizeof(float) * (size_t)d);
    } else {
      float len = sqrtf(acc);
      for (int32_t k = 0; k < d; ++k) o[k] = p[k] / len;
    }
  }
}

/* scores[q*n + r] for pre-processed q_hat (nq x d) and x_hat (n x d) */
void oracle_dense_scores(const float* q_hat, int32_t nq, const float* x_hat, int64_t n, int32_t d,
                         float* scores) {
  for (int32_t q = 0; q < nq; ++q) {
    const float* qp = q_hat + (int64_t)q * d;
    for (int64_t r = 0; r < n; ++r) {
      const float* xp = x_hat + r * d;
      float acc = 0.0f;
      for (int32_t k = 0; k < d; ++k) acc = fmaf(xp[k], qp[k], acc);
      scores[(int64_t)q * n + r] = acc;
    }
  }
}

float oracle_idf(int64_t n_points, int32_t df) {
  float n = (float)n_points;
  float f = (float)df;
  float num = (n - f) + 0.5f;
  float den = f + 0.5f;
  float arg = 1.0f + num / den;
  return (float)log((double)arg);
}

/* CSR rows sorted by token id; query sorted by token id, unique.
 * scores[r] = -inf when row r shares no term with the query. */
void oracle_sparse_scores(const int64_t* off, const int32_t* idx, const float* val, int64_t n,
                          const int32_t* q_idx, const float* q_val, int32_t nnz,
                          const int32_t* q_df, int64_t n_points, float* scores) {
  float* w = (float*)malloc(sizeof(float) * (size_t)(nnz > 0 ? nnz : 1));
  for (int32_t t = 0; t < nnz; ++t) w[t] = q_val[t] * oracle_idf(n_points, q_df[t]);
  for (int64_t r = 0; r < n; ++r) {
    float acc = 0.0f;
    int hit = 0;
    int32_t t = 0;
    for (int64_t j = off[r]; j < off[r + 1]; ++j) {
      while (t < nnz && q_idx[t] < idx[j]) ++t;
      if (t < nnz && q_idx[t] == idx[j]) {
        float prod = w[t] * val[j];
        acc = acc + prod;
        hit = 1;
      }
    }
    scores[r] = hit ? acc : -INFINITY;
  }
  free(w);
}

typedef struct {
  float s;
  int64_t r;
} oracle_pair;

static int cmp_pair(const void* a, const void* b) {
  const oracle_pair* x = (const oracle_pair*)a;
  const oracle_pair* y = (const oracle_pair*)b;
  if (x->s > y->s) return -1;
  if (x->s < y->s) return 1;
  return (x->r > y->r) - (x->r < y->r);
}

/* mask may be NULL. Returns the number of results (<= k); rows padded with -1. */
int32_t oracle_topk(const float* scores, const uint8_t* mask, int64_t n, int32_t k, int64_t* rows,
                    float* out_scores) {
  oracle_pair* p = (oracle_pair*)malloc(sizeof(oracle_pair) * (size_t)(n > 0 ? n : 1));
  int64_t m = 0;
  for (int64_t r = 0; r < n; ++r) {
    if (mask && !mask[r]) continue;
    if (scores[r] == -INFINITY) continue;
    p[m].s = scores[r];
    p[m].r = r;
    ++m;
  }
  qsort(p, (size_t)m, sizeof(oracle_pair), cmp_pair);
  int32_t c = (int32_t)(m < k ? m : k);
  for (int32_t i = 0; i < k; ++i) {
    rows[i] = i < c ? p[i].r : -1;
    out_scores[i] = i < c ? p[i].s : 0.0f;
  }
  free(p);
  return c;
}
