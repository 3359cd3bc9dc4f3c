// fq_pool.h -- a small persistent worker pool for the host phases.
// The host side of a call is a few dozen short parallel passes over per-read arrays; forking and joining a set of std::threads for
// each cost 1-2 ms per pass on the hosts this runs on (containerised: thread creation is a heavyweight operation there), i.e. tens of
// milliseconds per call.  Workers here are created once, sleep on a condition variable between passes, and inherit the CPU
// affinity the creating thread has at that time (a call pins itself to its NUMA node before its first pass).
#pragma once
#include <condition_variable>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

class FqWorkPool {
 public:
  FqWorkPool() = default;
  FqWorkPool(const FqWorkPool &) = delete;
  FqWorkPool &operator=(const FqWorkPool &) = delete;
  ~FqWorkPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_go_.notify_all();
    for (auto &t : th_) t.join();
  }
  // f(t) for every t in [0, T), the caller taking t = 0; returns when all are done.  One pass at a time per pool.
  template <class F>
  void run(int T, F &&f) {
    if (T <= 1) { f(0); return; }
    // a pass started from inside a pass of the same pool (by the calling thread's chunk or by a worker) runs on its own thread, in
    // order: the pool's workers are all part of the outer pass, and waiting for them here would never end
    if (in_pass() == this) { for (int t = 0; t < T; ++t) f(t); return; }
    std::lock_guard<std::mutex> one(run_mu_);
    struct Mark { FqWorkPool *&slot, *prev; Mark(FqWorkPool *&s, FqWorkPool *p) : slot(s), prev(s) { slot = p; } ~Mark() { slot = prev; } } mark(in_pass(), this);
    {
      std::unique_lock<std::mutex> lk(mu_);
      while ((int)th_.size() < T - 1) { const int idx = (int)th_.size(); th_.emplace_back([this, idx] { worker(idx); }); }
      call_ = [](void *ctx, int t) { (*static_cast<typename std::remove_reference<F>::type *>(ctx))(t); };
      ctx_ = &f; T_ = T;
      remaining_ = (int)th_.size();
      ++gen_;
    }
    cv_go_.notify_all();
    f(0);
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return remaining_ == 0; });
  }

 private:
  static FqWorkPool *&in_pass() { static thread_local FqWorkPool *p = nullptr; return p; }   // the pool whose pass this thread is running a chunk of
  void worker(int idx) {
    unsigned long seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(mu_);
      cv_go_.wait(lk, [&] { return stop_ || gen_ != seen; });
      if (stop_) return;
      seen = gen_;
      void (*call)(void *, int) = call_;
      void *ctx = ctx_;
      const int T = T_;
      lk.unlock();
      in_pass() = this;
      if (idx + 1 < T) call(ctx, idx + 1);
      in_pass() = nullptr;
      lk.lock();
      if (--remaining_ == 0) cv_done_.notify_one();
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_, run_mu_;
  std::condition_variable cv_go_, cv_done_;
  void (*call_)(void *, int) = nullptr;
  void *ctx_ = nullptr;
  int T_ = 0, remaining_ = 0;
  unsigned long gen_ = 0;
  bool stop_ = false;
};
