// Hybrid fusion on the host — the only arithmetic of the hot path that lives in the reference
// itself. Restates VectorStoreService._hybrid_search, src/voitta/services/vector_store.py:659-697:
//   normalize():  (s - min) / (max - min), or 1.0 for every item when max - min <= 0 (:659-669)
//   union of ids (:675); missing side contributes 0.0 (:678-679)
//   final = (1 - sparse_weight) * d + sparse_weight * s in Python floats = f64 (:634,:680)
//   result object taken from the dense list when present (:682-685)
//   sort by final descending, keep `limit`, overwrite score (:689-695)
// The reference iterates a Python set, so ties in `final` come out in arbitrary order
// (SURVEY.md F8); here ties go to the lower row id.
//
// json_scores: the reference receives scores over REST as JSON; the server prints an f32 with
// the shortest decimal that round-trips and Python parses that decimal to the nearest f64
// [EXT]. With json_scores != 0 the same transport is applied before the arithmetic.
//
// <= 6*limit candidates: latency, not roofline, matters; no GPU involved.

#include <algorithm>
#include <charconv>
#include <cstdlib>
#include <unordered_map>
#include <vector>

#include "engine_internal.h"
#include "host_parallel.h"

#pragma STDC FP_CONTRACT OFF

namespace vr {

static double to_python_float(float s, int json_scores) {
  if (!json_scores) return static_cast<double>(s);
  // shortest round-tripping decimal of the f32 (what the server prints), read back as the nearest f64 (what Python's
  // float() returns). The decimal has at most 9 significant digits, so for the exponents scores have it is ONE exactly
  // representable integer times or divided by ONE exactly representable power of ten — a single correctly rounded
  // operation, i.e. strtod's own fast path; anything else goes to strtod.
  char buf[64];
  auto r = std::to_chars(buf, buf + sizeof(buf) - 1, s, std::chars_format::scientific);
  *r.ptr = '\0';
  const char* p = buf;
  const bool neg = *p == '-';
  if (neg) ++p;
  uint64_t digits = 0;
  int nd = 0, point_at = -1;
  bool plain = true;
  for (; *p && *p != 'e'; ++p) {
    if (*p == '.') {
      point_at = nd;
    } else if (*p >= '0' && *p <= '9') {
      digits = digits * 10 + static_cast<uint64_t>(*p - '0');
      ++nd;
    } else {
      plain = false;  // inf / nan
      break;
    }
  }
  if (plain && *p == 'e' && nd >= 1 && nd <= 15) {
    const int e10 = atoi(p + 1) - (point_at < 0 ? 0 : nd - point_at);  // value = digits * 10^e10
    static const double kPow10[] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                    1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    if (e10 >= -22 && e10 <= 22) {
      const double d = static_cast<double>(digits);
      const double v = e10 < 0 ? d / kPow10[-e10] : d * kPow10[e10];
      return neg ? -v : v;
    }
  }
  return std::strtod(buf, nullptr);
}

namespace {
struct Cand {
  int64_t row;
  double d = 0.0, s = 0.0;
  bool in_dense = false, in_sparse = false;
  double final_score = 0.0;
};

void normalize(const float* scores, int n, int json_scores, std::vector<double>* out) {
  out->assign(static_cast<size_t>(n), 0.0);
  if (n == 0) return;
  static thread_local std::vector<double> v;  // (scratch kept per thread: no allocation per query)
  v.resize(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) v[static_cast<size_t>(i)] = to_python_float(scores[i], json_scores);
  double mn = v[0], mx = v[0];
  for (double x : v) {
    mn = std::min(mn, x);
    mx = std::max(mx, x);
  }
  double spread = mx - mn;
  for (int i = 0; i < n; ++i)
    (*out)[static_cast<size_t>(i)] = spread > 0 ? (v[static_cast<size_t>(i)] - mn) / spread : 1.0;
}

int emit(std::vector<Cand>& all, int limit, int64_t* out_rows, double* out_scores,
         int32_t* out_from_dense, int32_t* out_count) {
  std::sort(all.begin(), all.end(), [](const Cand& a, const Cand& b) {
    if (a.final_score != b.final_score) return a.final_score > b.final_score;
    return a.row < b.row;
  });
  int n = static_cast<int>(std::min<size_t>(all.size(), static_cast<size_t>(std::max(limit, 0))));
  for (int i = 0; i < n; ++i) {
    out_rows[i] = all[static_cast<size_t>(i)].row;
    out_scores[i] = all[static_cast<size_t>(i)].final_score;
    if (out_from_dense) out_from_dense[i] = all[static_cast<size_t>(i)].in_dense ? 1 : 0;
  }
  *out_count = n;
  return 0;
}

void gather(const int64_t* d_rows, int nd, const int64_t* s_rows, int ns, std::vector<Cand>* all,
            std::vector<int>* d_pos, std::vector<int>* s_pos) {
  d_pos->resize(static_cast<size_t>(nd));
  s_pos->resize(static_cast<size_t>(ns));
  all->reserve(static_cast<size_t>(nd + ns));
  if (nd + ns <= 128) {
    // the usual case (two lists of 3 x limit ids): an open-addressing table on the stack instead of a node-based map
    // (a batched hybrid search fuses a thousand queries on the host threads: the map's allocations were a third of it)
    constexpr int kSlots = 512;
    int64_t key[kSlots];
    int16_t at[kSlots];
    for (int i = 0; i < kSlots; ++i) at[i] = -1;
    auto slot = [&](int64_t row) {
      uint32_t h = static_cast<uint32_t>((static_cast<uint64_t>(row) * 0x9E3779B97F4A7C15ull) >> 55);  // 9 bits
      while (at[h] >= 0 && key[h] != row) h = (h + 1) & (kSlots - 1);
      if (at[h] < 0) {
        key[h] = row;
        at[h] = static_cast<int16_t>(all->size());
        Cand c;
        c.row = row;
        all->push_back(c);
      }
      return static_cast<int>(at[h]);
    };
    for (int i = 0; i < nd; ++i) (*d_pos)[static_cast<size_t>(i)] = slot(d_rows[i]);
    for (int i = 0; i < ns; ++i) (*s_pos)[static_cast<size_t>(i)] = slot(s_rows[i]);
    return;
  }
  std::unordered_map<int64_t, int> at;
  auto slot = [&](int64_t row) {
    auto it = at.find(row);
    if (it != at.end()) return it->second;
    int i = static_cast<int>(all->size());
    at.emplace(row, i);
    Cand c;
    c.row = row;
    all->push_back(c);
    return i;
  };
  for (int i = 0; i < nd; ++i) (*d_pos)[static_cast<size_t>(i)] = slot(d_rows[i]);
  for (int i = 0; i < ns; ++i) (*s_pos)[static_cast<size_t>(i)] = slot(s_rows[i]);
}
}  // namespace

int fuse_minmax(const int64_t* d_rows, const float* d_scores, int nd, const int64_t* s_rows,
                const float* s_scores, int ns, int limit, double sparse_weight, int json_scores,
                int64_t* out_rows, double* out_scores, int32_t* out_from_dense, int32_t* out_count) {
  const double dense_weight = 1.0 - sparse_weight;  // vector_store.py:634
  static thread_local std::vector<double> dn, sn;
  normalize(d_scores, nd, json_scores, &dn);
  normalize(s_scores, ns, json_scores, &sn);
  static thread_local std::vector<Cand> all;
  static thread_local std::vector<int> dp, sp;
  all.clear();
  gather(d_rows, nd, s_rows, ns, &all, &dp, &sp);
  // a later duplicate of an id overwrites the earlier one, as the dict assignment at :668 does
  for (int i = 0; i < nd; ++i) {
    Cand& c = all[static_cast<size_t>(dp[static_cast<size_t>(i)])];
    c.d = dn[static_cast<size_t>(i)];
    c.in_dense = true;
  }
  for (int i = 0; i < ns; ++i) {
    Cand& c = all[static_cast<size_t>(sp[static_cast<size_t>(i)])];
    c.s = sn[static_cast<size_t>(i)];
    c.in_sparse = true;
  }
  for (Cand& c : all) {
    double d = c.in_dense ? c.d : 0.0;
    double s = c.in_sparse ? c.s : 0.0;
    double a = dense_weight * d;
    double b = sparse_weight * s;
    c.final_score = a + b;
  }
  return emit(all, limit, out_rows, out_scores, out_from_dense, out_count);
}

// Reciprocal-rank fusion as the Qdrant server implements it for prefetch+fusion queries
// [EXT]: score = sum over lists of 1 / (position + 2), position counted from 0. No reference
// code path uses it (vector_store.py:638-639 explains why); offered because north_star names it.
int fuse_rrf(const int64_t* d_rows, int nd, const int64_t* s_rows, int ns, int limit,
             double /*sparse_weight*/, int64_t* out_rows, double* out_scores,
             int32_t* out_from_dense, int32_t* out_count) {
  static thread_local std::vector<Cand> all;
  static thread_local std::vector<int> dp, sp;
  all.clear();
  gather(d_rows, nd, s_rows, ns, &all, &dp, &sp);
  for (int i = 0; i < nd; ++i) {
    Cand& c = all[static_cast<size_t>(dp[static_cast<size_t>(i)])];
    c.final_score += 1.0 / (static_cast<double>(i) + 2.0);
    c.in_dense = true;
  }
  for (int i = 0; i < ns; ++i) {
    Cand& c = all[static_cast<size_t>(sp[static_cast<size_t>(i)])];
    c.final_score += 1.0 / (static_cast<double>(i) + 2.0);
    c.in_sparse = true;
  }
  return emit(all, limit, out_rows, out_scores, out_from_dense, out_count);
}

// nq pairs of lists at once, one query per host thread at a time (the last step of a batched or sharded hybrid
// search: vector_store.py:659-697 per query). d_* / s_*: [nq][k] with counts; out_*: [nq][limit].
int fuse_batch(const int64_t* d_rows, const float* d_scores, const int32_t* d_counts, const int64_t* s_rows,
               const float* s_scores, const int32_t* s_counts, int nq, int k, int limit, double sparse_weight, int fusion,
               int json_scores, int64_t* out_rows, double* out_scores, int32_t* out_from_dense, int32_t* out_counts) {
  VR_CHECK(nq >= 0 && k >= 1 && limit >= 1 && d_rows && d_scores && d_counts && out_rows && out_scores && out_counts,
           "bad arguments");
  VR_CHECK(fusion == VR_FUSION_MINMAX || fusion == VR_FUSION_RRF, "unknown fusion %d", fusion);
  std::atomic<int> failed{0};
  parallel_for(nq, 8, [&](int64_t i) {
    const int nd = std::max(0, std::min<int>(d_counts[i], k));
    const int ns = (s_counts && s_rows && s_scores) ? std::max(0, std::min<int>(s_counts[i], k)) : 0;
    const int64_t* sr = s_rows ? s_rows + i * k : nullptr;
    const float* ss = s_scores ? s_scores + i * k : nullptr;
    int32_t* fd = out_from_dense ? out_from_dense + i * limit : nullptr;
    const int rc = fusion == VR_FUSION_MINMAX
                       ? fuse_minmax(d_rows + i * k, d_scores + i * k, nd, sr, ss, ns, limit, sparse_weight, json_scores,
                                     out_rows + i * limit, out_scores + i * limit, fd, out_counts + i)
                       : fuse_rrf(d_rows + i * k, nd, sr, ns, limit, sparse_weight, out_rows + i * limit,
                                  out_scores + i * limit, fd, out_counts + i);
    if (rc != 0) failed.store(1);
  });
  VR_CHECK(!failed.load(), "fusion failed");
  return 0;
}

}  // namespace vr
