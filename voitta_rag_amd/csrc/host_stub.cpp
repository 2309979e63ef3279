// Sanitizer build only (make asan): the few symbols of api.hip the host-only translation units need, so that
// wordpiece.cpp, bm25_text.cpp, chunking.cpp and fusion.cpp — the string / Unicode code that sees untrusted document
// text — link into libvoitta_host_asan.so without any HIP code. Never part of libvoitta_engine.so.
#include <cstdarg>
#include <cstdio>
#include <string>

#include "engine_internal.h"

namespace vr {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

}  // namespace vr

extern "C" {

int vr_abi_version(void) { return VR_ABI_VERSION; }
const char* vr_last_error(void) { return vr::g_last_error.c_str(); }

int vr_fuse_minmax(const int64_t* d_rows, const float* d_scores, int32_t nd, const int64_t* s_rows, const float* s_scores,
                   int32_t ns, int32_t limit, double sparse_weight, int32_t json_scores, int64_t* out_rows,
                   double* out_scores, int32_t* out_from_dense, int32_t* out_count) {
  return vr::fuse_minmax(d_rows, d_scores, nd, s_rows, s_scores, ns, limit, sparse_weight, json_scores, out_rows, out_scores,
                         out_from_dense, out_count);
}

int vr_fuse_rrf(const int64_t* d_rows, int32_t nd, const int64_t* s_rows, int32_t ns, int32_t limit, int64_t* out_rows,
                double* out_scores, int32_t* out_from_dense, int32_t* out_count) {
  return vr::fuse_rrf(d_rows, nd, s_rows, ns, limit, 0.0, out_rows, out_scores, out_from_dense, out_count);
}

int vr_fuse_batch(const int64_t* d_rows, const float* d_scores, const int32_t* d_counts, const int64_t* s_rows,
                  const float* s_scores, const int32_t* s_counts, int32_t nq, int32_t k, int32_t limit, double sparse_weight,
                  int32_t fusion, int32_t json_scores, int64_t* out_rows, double* out_scores, int32_t* out_from_dense,
                  int32_t* out_counts) {
  return vr::fuse_batch(d_rows, d_scores, d_counts, s_rows, s_scores, s_counts, nq, k, limit, sparse_weight, fusion,
                        json_scores, out_rows, out_scores, out_from_dense, out_counts);
}

}  // extern "C"
