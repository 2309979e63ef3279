// Host-side parallel loop for the string stages (WordPiece, BM25 tokeniser): one text is independent
// of the next, and at 35k chunks/s on the GPU a single host thread tokenising (~33k texts/s) would be
// the bottleneck of the indexing pipeline. VOITTA_HOST_THREADS overrides the thread count (default:
// the hardware concurrency, at most 16 — the CPU share of one GPU).
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

namespace vr {

inline int host_threads() {
  static const int n = [] {
    if (const char* s = getenv("VOITTA_HOST_THREADS")) return std::max(1, atoi(s));
    const unsigned hc = std::thread::hardware_concurrency();
    return static_cast<int>(std::min(16u, std::max(1u, hc)));
  }();
  return n;
}

// fn(i) for i in [0, n): blocks of `grain` indices are handed out dynamically
template <class F>
void parallel_for(int64_t n, int64_t grain, F fn) {
  const int threads = static_cast<int>(std::min<int64_t>(host_threads(), (n + grain - 1) / grain));
  if (threads <= 1) {
    for (int64_t i = 0; i < n; ++i) fn(i);
    return;
  }
  std::atomic<int64_t> next{0};
  auto work = [&] {
    for (;;) {
      const int64_t b = next.fetch_add(grain);
      if (b >= n) return;
      const int64_t e = std::min(n, b + grain);
      for (int64_t i = b; i < e; ++i) fn(i);
    }
  };
  std::vector<std::thread> pool;
  pool.reserve(static_cast<size_t>(threads - 1));
  for (int t = 1; t < threads; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
}

}  // namespace vr
