// svo_hip_pool.h -- the small thread pool under FrontendLockstep (svo_hip_lockstep.h): host phases of many camera streams,
// item i always on thread i % size().  Plain C++ threads, no device call: built and raced by itself under
// ThreadSanitizer in tests/san/.
#pragma once

#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace svo_hip {

// A fixed set of threads that run `fn(item)` for item = 0 .. n-1 and return when all are done; the calling thread takes
// part.  Waiting threads spin briefly (a phase follows the last within microseconds), then yield, then sleep.
class WorkerPool {
 public:
  // n_threads >= 1 counts the caller: n_threads - 1 threads are started.  pin: every thread of the pool -- the calling
  // thread included, for good -- is bound to a CPU of its own out of the process' affinity mask, one hardware thread per
  // core first; pools made one after the other (one per lock-step group) take consecutive CPUs.
  explicit WorkerPool(int n_threads, bool pin = false);
  ~WorkerPool();
  WorkerPool(const WorkerPool&) = delete;
  WorkerPool& operator=(const WorkerPool&) = delete;
  int size() const { return static_cast<int>(threads_.size()) + 1; }
  // exceptions thrown by fn are collected; the first one is rethrown here once every item has been handled or skipped.
  // Item i always goes to thread i % size(): a stream's data stays in the caches of the core that touched it last.
  void run(int n_items, const std::function<void(int)>& fn);

 private:
  void worker(int tid, int cpu);
  void work_off(int tid);
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::atomic<unsigned long> generation_{ 0 };
  std::atomic<int> pending_{ 0 }, n_items_{ 0 };
  std::atomic<int> sleepers_{ 0 };
  std::atomic<bool> stop_{ false };
  int spin_limit_ = 20000, yield_limit_ = 400;
  const std::function<void(int)>* fn_ = nullptr;
  std::mutex err_mu_;
  std::exception_ptr error_;
};


// ONE WorkerPool for several lock-step groups, one phase at a time: a group's thread takes the whole pool for its phase (every
// stream of the phase on a thread of its own instead of two or three streams per thread of a pool a third the size) and gives
// it back before it talks to the device or waits for it.  Unlike SharedPool below nothing runs side by side inside the pool:
// item i is always on thread i % size(), whoever the caller -- the callers take turns as thread 0.
class ExclusivePool {
 public:
  explicit ExclusivePool(int n_threads) : pool_(n_threads) {}
  int size() const { return pool_.size(); }
  void run(int n_items, const std::function<void(int)>& fn)
  {
    struct Turn {
      std::atomic_flag& f;
      explicit Turn(std::atomic_flag& flag) : f(flag) { while (f.test_and_set(std::memory_order_acquire)) std::this_thread::yield(); }
      ~Turn() { f.clear(std::memory_order_release); }
    } turn(busy_);
    pool_.run(n_items, fn);
  }

 private:
  WorkerPool pool_;
  std::atomic_flag busy_ = ATOMIC_FLAG_INIT;
};

// One set of worker threads for SEVERAL lock-step groups.  A group's thread is busy with its context's serial work and
// waits for the device a third of the time; with a pool of its own its workers idle through all of that.  Here every
// group's phases draw on the same workers: a phase of 8 items finds up to all of them free and takes a stream's time
// instead of two or three, and a worker that would idle through one group's device wait serves another group's phase.
// Item i of a group prefers worker (i + seed) % n_workers -- a stream's data stays in one core's caches as long as that
// worker is free -- and is taken by whoever is idle otherwise.  run() may be called from several threads at once (one
// per group); the caller works on its own items too.
class SharedPool {
 public:
  explicit SharedPool(int n_workers);
  ~SharedPool();
  SharedPool(const SharedPool&) = delete;
  SharedPool& operator=(const SharedPool&) = delete;
  int workers() const { return static_cast<int>(threads_.size()); }
  void run(int n_items, const std::function<void(int)>& fn, int seed);

 private:
  static constexpr int kSlots = 16, kMaxItems = 256;
  struct Job {
    std::atomic<int> state{ 0 };        // 0 free, 1 live
    std::atomic<int> readers{ 0 };      // workers looking at the job right now
    std::atomic<int> pending{ 0 };      // items not finished
    std::atomic<unsigned long long> taken[kMaxItems / 64];
    const std::function<void(int)>* fn = nullptr;
    int n_items = 0, seed = 0;
    std::mutex err_mu;
    std::exception_ptr error;
  };
  bool work_on(Job& job, int worker);   // true if it ran an item
  void worker(int id);
  Job jobs_[kSlots];
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::atomic<unsigned long> epoch_{ 0 };
  std::atomic<int> sleepers_{ 0 };
  std::atomic<bool> stop_{ false };
  int spin_limit_ = 20000, yield_limit_ = 400;
};

}  // namespace svo_hip
