// host_parallel.h — the host threads of the library's few host-side bulk passes (look-ahead precompute, relabelling,
// assembling the results of a large batch).  Threads are created per call: the passes are milliseconds long and rare.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <exception>
#include <system_error>
#include <thread>
#include <vector>

namespace wfst {

// threads for a pass over `work_items` items: all cores up to 32 (WFST_HOST_THREADS overrides), one for small inputs
inline unsigned host_threads(uint64_t work_items) {
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  if (const char* e = std::getenv("WFST_HOST_THREADS")) n = (unsigned)std::max(1, std::atoi(e));
  n = std::min(n, 32u);
  if (work_items < (1u << 16) && !std::getenv("WFST_HOST_THREADS")) n = 1;  // (tests force threads on small inputs)
  return n;
}

// body(thread, begin, end) over [0, n_items) in chunks handed out by an atomic counter; the first exception is rethrown
template <class F>
void parallel_chunks(unsigned n_thr, uint64_t n_items, uint64_t chunk, F&& body) {
  if (n_thr <= 1 || n_items <= chunk) {
    if (n_items) body(0u, (uint64_t)0, n_items);
    return;
  }
  std::atomic<uint64_t> next{0};
  std::vector<std::exception_ptr> errs(n_thr);
  std::vector<std::thread> pool;
  auto worker = [&](unsigned t) {
    try {
      for (;;) {
        const uint64_t b = next.fetch_add(chunk, std::memory_order_relaxed);
        if (b >= n_items) break;
        body(t, b, std::min(n_items, b + chunk));
      }
    } catch (...) {
      errs[t] = std::current_exception();
    }
  };
  pool.reserve(n_thr);
  unsigned started = 0;
  for (; started + 1 < n_thr; ++started) {
    // a thread that cannot be created (EAGAIN under a process limit) must not leave joinable threads behind: the caller's
    // thread takes the chunks nobody else will (it always works as the last "thread" anyway)
    try {
      pool.emplace_back(worker, started);
    } catch (const std::system_error&) {
      break;
    }
  }
  worker(n_thr - 1);
  for (auto& th : pool) th.join();
  for (auto& e : errs)
    if (e) std::rethrow_exception(e);
}

}  // namespace wfst
