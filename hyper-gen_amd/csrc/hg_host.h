// Host-only seams between hg_formats.cpp (no HIP) and the hg_api*.hip translation units.
#pragma once
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/hypergen.h"

// enlarge `buf` to at least `need` bytes keeping its first `keep` bytes; false = out of memory
typedef bool (*hg_grow_fn)(uint8_t *&buf, size_t &cap, size_t need, size_t keep, void *user);

hg_status hg_read_fastx_impl(const char *path, uint32_t mode, uint8_t **pbuf, size_t *pcap, size_t *n_bps,
                             hg_grow_fn grow, void *user);

// hg_formats.cpp: bases [b0, b1) of one genome into their place in its hg_pack2 blob (b0 % 64 == 0; b1 % 64 == 0 or
// b1 == n_bps; `out` must not overlap `seq`) -- the unit of work of the host threads that pack a host-fed batch
void hg_pack2_piece(const uint8_t *seq, size_t n_bps, uint32_t norm_mode, uint8_t *out, size_t b0, size_t b1);

// hg_formats.cpp: NUMA node of the page that holds p (-1 unknown)
int hg_numa_node_of(const void *p);

// A few worker threads that live for one call: run(n, fn) executes fn(i) for i in [0, n) on all of them (the caller's
// thread takes part) and returns when every index is done.
class CallPool {
 public:
  // node >= 0: the workers bind themselves to that NUMA node's CPUs (the caller's own thread is left where it is)
  explicit CallPool(unsigned threads, int node = -1) {
    try {
      for (unsigned t = 1; t < threads; ++t)
        th_.emplace_back([this, node, threads] {
          (void)hg_bind_thread_to_numa_node(node, threads);
          worker();
        });
    } catch (...) {  // no more threads to be had: the pool works with the ones it got (the caller's thread at least)
    }
  }
  ~CallPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  template <class F>
  void run(size_t n, F &&fn) {
    if (!n) return;
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = [&fn](size_t i) { fn(i); };
      n_ = n, next_ = 0, done_ = 0, ++gen_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return done_ == n_; });
    fn_ = nullptr;
  }

 private:
  void work() {
    for (;;) {
      size_t i;
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (next_ >= n_) return;
        i = next_++;
      }
      fn_(i);
      std::lock_guard<std::mutex> lk(mu_);
      if (++done_ == n_) cv_done_.notify_all();
    }
  }
  void worker() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
      }
      work();
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, cv_done_;
  std::function<void(size_t)> fn_;
  size_t n_ = 0, next_ = 0, done_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

