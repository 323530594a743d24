// TEST INFRASTRUCTURE (CPU baseline of bench.py): the thread harness of the *_bench_small_sv_timed entry points of ref_driver.cpp (the
// unmodified reference) and manta_oracle.cpp (the restatement).  Mirrors the worker pool of GenerateSVCandidates.cpp:232-266 -- n threads,
// each with its own aligner set, pulling loci from one counter -- but keeps what is not the reference's work out of the clock: the threads
// are started (and, on request, pinned one per CPU of the process' affinity mask) BEFORE the clock starts, and a run is long enough per
// thread (loci_per_thread, the batch's loci are taken round robin) for the start-up and the tail to vanish.
#pragma once
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <thread>
#include <vector>

namespace bench_harness {

/// runs work(locus_index) on n_threads threads until `target` loci are done or `max_seconds` have passed (whichever comes first; a locus
/// that was started is finished).  Returns the wall seconds from "all threads ready" to "last thread done"; *done = loci processed.
template <typename MakeWorker>
double run(const int n_threads, const uint64_t target, const double max_seconds, const bool pin, const int n_loci, MakeWorker makeWorker, uint64_t* done)
{
  std::atomic<uint64_t> next(0), finished(0);
  std::atomic<int>      ready(0);
  std::atomic<bool>     go(false);
  std::vector<int>      cpus;
  if (pin) {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0)
      for (int c = 0; c < CPU_SETSIZE; ++c)
        if (CPU_ISSET(c, &set)) cpus.push_back(c);
  }
  std::chrono::steady_clock::time_point t0;
  auto body = [&](const int t) {
    if (!cpus.empty()) {
      cpu_set_t one;
      CPU_ZERO(&one);
      CPU_SET(cpus[size_t(t) % cpus.size()], &one);
      (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
    }
    auto work = makeWorker();  // per-thread state (the aligner objects hold scratch: one set per thread, as the reference's workers do)
    ready.fetch_add(1);
    while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
    const auto deadline = t0 + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(max_seconds));
    while (true) {
      const uint64_t i = next.fetch_add(1);
      if (i >= target) break;
      if (max_seconds > 0 && std::chrono::steady_clock::now() > deadline) break;
      work(int(i % uint64_t(n_loci)));
      finished.fetch_add(1);
    }
  };
  // (every worker is a thread of its own: the caller's thread is never pinned and only keeps the clock)
  std::vector<std::thread> pool;
  for (int t = 0; t < n_threads; ++t) pool.emplace_back(body, t);
  while (ready.load() < n_threads) std::this_thread::yield();
  t0 = std::chrono::steady_clock::now();
  go.store(true, std::memory_order_release);
  for (auto& th : pool) th.join();
  const auto t1 = std::chrono::steady_clock::now();
  if (done) *done = finished.load();
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // namespace bench_harness
