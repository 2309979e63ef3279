// In-engine kernel timing with HIP events recorded on the stream the kernels are launched on
// (torch.cuda.Event would only see torch's current stream). bench.py turns it on for the timed
// region and divides the algorithmic work of each kernel class by the summed event time to get
// the `roofline.achieved` figure; the same averages must agree with rocprofv3 --kernel-trace.
#include "engine_internal.h"

#include <cmath>

namespace vr {

struct ProfRecord {
  hipEvent_t a, b;
  int cls;
  double work;
};

struct Profiler {
  std::mutex mu;  // searches record from their lanes (any thread), writers from the master
  bool on = false;
  std::vector<ProfRecord> recs;
  std::vector<hipEvent_t> pool;
};

static thread_local int64_t tl_open = -1;  // the record this thread opened with prof_begin

static Profiler* prof(vr_engine* e) { return static_cast<Profiler*>(e->profiler); }

bool prof_on(vr_engine* e) {
  Profiler* p = prof(e);
  return p && p->on;
}

void prof_begin(vr_engine* e, int cls, double work) {
  Profiler* p = prof(e);
  tl_open = -1;
  if (!p || !p->on) return;
  std::lock_guard<std::mutex> lock(p->mu);
  ProfRecord r{};
  for (hipEvent_t* ev : {&r.a, &r.b}) {
    if (!p->pool.empty()) {
      *ev = p->pool.back();
      p->pool.pop_back();
    } else if (hipEventCreate(ev) != hipSuccess) {
      return;
    }
  }
  r.cls = cls;
  r.work = work;
  (void)hipEventRecord(r.a, e->stream);
  p->recs.push_back(r);
  tl_open = static_cast<int64_t>(p->recs.size()) - 1;
}

void prof_end(vr_engine* e) {
  Profiler* p = prof(e);
  if (!p || !p->on || tl_open < 0) return;
  std::lock_guard<std::mutex> lock(p->mu);
  if (tl_open < static_cast<int64_t>(p->recs.size())) (void)hipEventRecord(p->recs[static_cast<size_t>(tl_open)].b, e->stream);
  tl_open = -1;
}

void prof_release(vr_engine* e) {
  Profiler* p = prof(e);
  if (!p) return;
  for (ProfRecord& r : p->recs) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  for (hipEvent_t ev : p->pool) (void)hipEventDestroy(ev);
  delete p;
  e->profiler = nullptr;
}

}  // namespace vr

using namespace vr;

// host twin of query_weights_kernel's idf (sparse.hip): argument formed in f32, ln in f64, one
// rounding to f32 — so a sharded caller weights query terms exactly as a single engine would
extern "C" float vr_idf(int64_t n_points, int32_t df) {
  volatile float n = static_cast<float>(n_points);
  volatile float f = static_cast<float>(df);
  volatile float num = (n - f) + 0.5f;
  volatile float den = f + 0.5f;
  volatile float arg = 1.0f + num / den;
  volatile double a = static_cast<double>(arg);
  return static_cast<float>(std::log(a));
}

extern "C" int vr_profile(vr_engine* e, int enable) {
  VR_CHECK(e != nullptr, "null engine");
  std::lock_guard<std::mutex> lock(e->wmu);
  if (!e->profiler) e->profiler = new Profiler();
  Profiler* p = prof(e);
  std::lock_guard<std::mutex> plock(p->mu);
  if (enable) {
    VR_HIP(hipStreamSynchronize(e->stream));
    for (ProfRecord& r : p->recs) {
      p->pool.push_back(r.a);
      p->pool.push_back(r.b);
    }
    p->recs.clear();
  }
  p->on = enable != 0;
  return 0;
}

extern "C" int vr_profile_read(vr_engine* e, int kernel_class, double* total_ms, int64_t* launches,
                               double* total_work) {
  VR_CHECK(e != nullptr && total_ms && launches && total_work, "null argument");
  std::lock_guard<std::mutex> lock(e->wmu);
  *total_ms = 0.0;
  *launches = 0;
  *total_work = 0.0;
  Profiler* p = prof(e);
  if (!p) return 0;
  VR_HIP(hipSetDevice(e->device));
  VR_HIP(hipStreamSynchronize(e->stream));
  std::lock_guard<std::mutex> plock(p->mu);
  for (const ProfRecord& r : p->recs) {
    if (r.cls != kernel_class) continue;
    float ms = 0.0f;
    VR_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    *total_ms += ms;
    *total_work += r.work;
    ++*launches;
  }
  return 0;
}
