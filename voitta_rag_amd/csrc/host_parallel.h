// Host-side parallel loop for the string stages (WordPiece, BM25 tokeniser) and the per-query fusion of a batched
// search: one item is independent of the next, and at 35k chunks/s on the GPU a single host thread tokenising
// (~33k texts/s) would be the bottleneck of the indexing pipeline. VOITTA_HOST_THREADS overrides the thread count
// (default: the hardware concurrency, at most 16 — the CPU share of one GPU).
//
// The workers are a process-wide pool created on first use and parked on a condition variable between loops:
// starting fifteen threads per call cost half a millisecond, which a 1000-query fusion (1 ms of work) cannot afford.
// One loop runs on the pool at a time; a caller that finds it busy (another thread's loop) runs its own loop inline.
#pragma once

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <pthread.h>
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

namespace detail {

struct HostPool {
  std::mutex owner;  // one parallel loop at a time
  std::mutex mu;
  std::condition_variable wake, done;
  std::function<void()> job;  // what every participating worker runs (it pulls blocks of indices itself)
  uint64_t generation = 0;
  int wanted = 0;   // workers that should run the current job
  int started = 0;  // ... that have picked it up
  int running = 0;  // ... that are still inside it
  std::vector<std::thread> workers;

  void worker_loop() {
    uint64_t seen = 0;
    for (;;) {
      std::function<void()> fn;
      {
        std::unique_lock<std::mutex> g(mu);
        wake.wait(g, [&] { return generation != seen && started < wanted; });
        seen = generation;
        ++started;
        fn = job;
      }
      fn();
      {
        std::lock_guard<std::mutex> g(mu);
        if (--running == 0) done.notify_all();
      }
    }
  }

  // runs fn on `extra` pool workers and on the caller; returns when all of them have finished
  void run(int extra, const std::function<void()>& fn) {
    {
      std::lock_guard<std::mutex> g(mu);
      while (static_cast<int>(workers.size()) < extra) {
        workers.emplace_back([this] { worker_loop(); });
        workers.back().detach();  // parked for the life of the process
      }
      job = fn;
      wanted = extra;
      started = 0;
      running = extra;
      ++generation;
    }
    wake.notify_all();
    fn();
    std::unique_lock<std::mutex> g(mu);
    done.wait(g, [&] { return running == 0; });
    wanted = 0;  // late wake-ups of this generation find nothing to start
  }
};

inline HostPool*& host_pool_slot() {
  static HostPool* pool = nullptr;
  return pool;
}

inline HostPool& host_pool() {
  static std::once_flag once;
  std::call_once(once, [] {
    host_pool_slot() = new HostPool();  // never destroyed: its threads may outlive static destruction
    // a forked child inherits the pool object but none of its threads: it starts over with an empty one
    pthread_atfork(nullptr, nullptr, [] { host_pool_slot() = new HostPool(); });
  });
  return *host_pool_slot();
}

}  // namespace detail

// fn(i) for i in [0, n): blocks of `grain` indices are handed out dynamically
template <class F>
void parallel_for(int64_t n, int64_t grain, F fn) {
  const int threads = static_cast<int>(std::min<int64_t>(host_threads(), (n + grain - 1) / grain));
  detail::HostPool& pool = detail::host_pool();
  std::unique_lock<std::mutex> own(pool.owner, std::try_to_lock);
  if (threads <= 1 || !own.owns_lock()) {
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
  pool.run(threads - 1, work);
}

}  // namespace vr
