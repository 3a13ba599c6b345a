// svo_hip_pool.cpp -- WorkerPool (svo_hip_pool.h).
#include "svo_hip_pool.h"

#include <pthread.h>
#include <sched.h>

#include <cstdio>
#include <cstdlib>

#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#define SVOH_CPU_RELAX() _mm_pause()
#else
#define SVOH_CPU_RELAX() do { } while (0)
#endif

namespace svo_hip {

// ---- WorkerPool --------------------------------------------------------------------------------------------------
namespace {
// the CPUs this process may run on, one hardware thread per core first (so that a pool smaller than the mask does not
// put two busy threads on the two hardware threads of one core)
// Read ONCE, at the first pinned pool of the process: a pinned pool binds its calling thread to one CPU for good, so the mask
// of that thread is a single CPU when it builds its next pool (svoh_mini_frontend: a new engine per lap on the group's
// thread) -- every later pool would put all its workers on that one CPU.
std::vector<int> read_allowed_cpus()
{
  std::vector<int> cpus;
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof set, &set) != 0) return cpus;
  std::vector<int> first, rest;
  std::vector<std::pair<int, int>> seen_cores;   // (package, core)
  for (int c = 0; c < CPU_SETSIZE; ++c) {
    if (!CPU_ISSET(c, &set)) continue;
    int core = c, pkg = 0;
    {
      char path[128];
      snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/core_id", c);
      if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &core) != 1) core = c; fclose(f); }
      snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/physical_package_id", c);
      if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &pkg) != 1) pkg = 0; fclose(f); }
    }
    const std::pair<int, int> key(pkg, core);
    bool dup = false;
    for (const auto& k : seen_cores) dup = dup || k == key;
    if (dup) rest.push_back(c); else { seen_cores.push_back(key); first.push_back(c); }
  }
  cpus = first;
  cpus.insert(cpus.end(), rest.begin(), rest.end());
  return cpus;
}

const std::vector<int>& allowed_cpus()
{
  static const std::vector<int> cpus = read_allowed_cpus();   // thread-safe (magic static)
  return cpus;
}
std::atomic<unsigned> g_next_cpu_slot{ 0 };
void bind_to(int cpu)
{
  if (cpu < 0) return;
  cpu_set_t set;
  CPU_ZERO(&set);
  CPU_SET(cpu, &set);
  (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
}
}  // namespace

WorkerPool::WorkerPool(int n_threads, bool pin)
{
  if (n_threads < 1) n_threads = 1;
  // how long an idle worker spins (pause instructions), then yields, before it sleeps: SVOH_LOCKSTEP_SPIN / _YIELD
  if (const char* e = getenv("SVOH_LOCKSTEP_SPIN")) spin_limit_ = atoi(e);
  if (const char* e = getenv("SVOH_LOCKSTEP_YIELD")) yield_limit_ = atoi(e);
  std::vector<int> cpus;
  if (pin) cpus = allowed_cpus();
  const unsigned slot0 = cpus.empty() ? 0u : g_next_cpu_slot.fetch_add(static_cast<unsigned>(n_threads));
  auto cpu_of = [&](int tid) { return cpus.empty() ? -1 : cpus[(slot0 + static_cast<unsigned>(tid)) % cpus.size()]; };
  bind_to(cpu_of(0));
  for (int i = 1; i < n_threads; ++i) {
    try { threads_.emplace_back(&WorkerPool::worker, this, i, cpu_of(i)); }
    catch (...) { break; }   // a thread that cannot be started (pid limit of a container): the pool is smaller, nothing else
  }
}

WorkerPool::~WorkerPool()
{
  stop_.store(true);
  generation_.fetch_add(1);
  { std::lock_guard<std::mutex> lock(mu_); cv_.notify_all(); }
  for (std::thread& t : threads_) t.join();
}

void WorkerPool::work_off(int tid)
{
  const int n = n_items_.load(std::memory_order_acquire), step = size();
  for (int i = tid; i < n; i += step) {
    try { (*fn_)(i); }
    catch (...) { std::lock_guard<std::mutex> lock(err_mu_); if (!error_) error_ = std::current_exception(); }
  }
}

void WorkerPool::worker(int tid, int cpu)
{
  bind_to(cpu);
  unsigned long seen = 0;
  for (;;) {
    // the next phase usually follows within microseconds: spin, then yield, then sleep
    int spins = 0;
    while (generation_.load(std::memory_order_acquire) == seen) {
      if (spins < spin_limit_) { SVOH_CPU_RELAX(); ++spins; }
      else if (spins < spin_limit_ + yield_limit_) { std::this_thread::yield(); ++spins; }
      else {
        sleepers_.fetch_add(1);
        {
          std::unique_lock<std::mutex> lock(mu_);
          cv_.wait(lock, [&] { return generation_.load() != seen || stop_.load(); });
        }
        sleepers_.fetch_sub(1);
      }
    }
    if (stop_.load()) return;
    seen = generation_.load(std::memory_order_acquire);
    work_off(tid);
    // every thread reports back, with or without items of its own: run() does not return -- and the next run() does not
    // publish its items -- before all of them have, so that no thread can ever be a run behind
    pending_.fetch_sub(1, std::memory_order_acq_rel);
  }
}

void WorkerPool::run(int n_items, const std::function<void(int)>& fn)
{
  if (n_items <= 0) return;
  if (threads_.empty() || n_items == 1) { for (int i = 0; i < n_items; ++i) fn(i); return; }
  { std::lock_guard<std::mutex> lock(err_mu_); error_ = nullptr; }
  // (every thread of the pool has reported back from the run before)
  fn_ = &fn;
  n_items_.store(n_items, std::memory_order_release);
  pending_.store(static_cast<int>(threads_.size()), std::memory_order_release);
  generation_.fetch_add(1);
  if (sleepers_.load() > 0) { std::lock_guard<std::mutex> lock(mu_); cv_.notify_all(); }
  work_off(0);
  while (pending_.load(std::memory_order_acquire) != 0) SVOH_CPU_RELAX();
  std::exception_ptr e;
  { std::lock_guard<std::mutex> lock(err_mu_); e = error_; error_ = nullptr; }
  if (e) std::rethrow_exception(e);
}

// ---- SharedPool ---------------------------------------------------------------------------------------------------
SharedPool::SharedPool(int n_workers)
{
  if (const char* e = getenv("SVOH_LOCKSTEP_SPIN")) spin_limit_ = atoi(e);
  if (const char* e = getenv("SVOH_LOCKSTEP_YIELD")) yield_limit_ = atoi(e);
  for (Job& j : jobs_) for (auto& t : j.taken) t.store(0);
  for (int i = 0; i < n_workers; ++i) {
    try { threads_.emplace_back(&SharedPool::worker, this, i); }
    catch (...) { break; }
  }
}

SharedPool::~SharedPool()
{
  stop_.store(true);
  epoch_.fetch_add(1);
  { std::lock_guard<std::mutex> lock(mu_); cv_.notify_all(); }
  for (std::thread& t : threads_) t.join();
}

// Take items of `job` until none is left to take: first the ones that prefer this worker, then anybody's.
bool SharedPool::work_on(Job& job, int worker)
{
  const int n = job.n_items, W = static_cast<int>(threads_.size());
  bool ran = false;
  auto claim = [&](int i) {
    const unsigned long long bit = 1ull << (i & 63);
    return (job.taken[i >> 6].fetch_or(bit, std::memory_order_acq_rel) & bit) == 0;
  };
  auto exec = [&](int i) {
    try { (*job.fn)(i); }
    catch (...) { std::lock_guard<std::mutex> lock(job.err_mu); if (!job.error) job.error = std::current_exception(); }
    job.pending.fetch_sub(1, std::memory_order_acq_rel);
    ran = true;
  };
  if (worker >= 0) {
    int first = (worker - job.seed) % W;
    if (first < 0) first += W;
    for (int i = first; i < n; i += W) if (claim(i)) exec(i);
  }
  for (int i = 0; i < n; ++i) {
    if (job.taken[i >> 6].load(std::memory_order_acquire) & (1ull << (i & 63))) continue;
    if (claim(i)) exec(i);
  }
  return ran;
}

void SharedPool::worker(int id)
{
  int idle = 0;
  unsigned long seen = epoch_.load();
  for (;;) {
    if (stop_.load()) return;
    bool ran = false;
    for (Job& j : jobs_) {
      if (j.state.load(std::memory_order_acquire) != 1) continue;
      j.readers.fetch_add(1, std::memory_order_acq_rel);
      if (j.state.load(std::memory_order_acquire) == 1 && j.pending.load(std::memory_order_acquire) > 0) ran = work_on(j, id) || ran;
      j.readers.fetch_sub(1, std::memory_order_acq_rel);
    }
    if (ran) { idle = 0; continue; }
    if (idle < spin_limit_) { SVOH_CPU_RELAX(); ++idle; }
    else if (idle < spin_limit_ + yield_limit_) { std::this_thread::yield(); ++idle; }
    else {
      sleepers_.fetch_add(1);
      {
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [&] { return epoch_.load() != seen || stop_.load(); });
      }
      sleepers_.fetch_sub(1);
      idle = 0;
    }
    seen = epoch_.load();
  }
}

void SharedPool::run(int n_items, const std::function<void(int)>& fn, int seed)
{
  if (n_items <= 0) return;
  if (threads_.empty() || n_items == 1 || n_items > kMaxItems) { for (int i = 0; i < n_items; ++i) fn(i); return; }
  // a free slot (there are more slots than groups; a caller that finds none runs its items by itself)
  Job* job = nullptr;
  for (Job& j : jobs_) {
    int expect = 0;
    if (j.state.load(std::memory_order_relaxed) == 0 && j.readers.load(std::memory_order_acquire) == 0 && j.state.compare_exchange_strong(expect, 2)) { job = &j; break; }
  }
  if (!job) { for (int i = 0; i < n_items; ++i) fn(i); return; }
  // (state 2: claimed by this caller, invisible to the workers, and no worker is still inside from its last life)
  while (job->readers.load(std::memory_order_acquire) != 0) SVOH_CPU_RELAX();
  job->fn = &fn; job->n_items = n_items; job->seed = seed; job->error = nullptr;
  for (auto& t : job->taken) t.store(0, std::memory_order_relaxed);
  job->pending.store(n_items, std::memory_order_release);
  job->state.store(1, std::memory_order_release);
  epoch_.fetch_add(1);
  if (sleepers_.load() > 0) { std::lock_guard<std::mutex> lock(mu_); cv_.notify_all(); }
  work_on(*job, -1);
  while (job->pending.load(std::memory_order_acquire) != 0) SVOH_CPU_RELAX();
  std::exception_ptr e = job->error;
  job->error = nullptr;
  job->state.store(0, std::memory_order_release);   // (workers that still look at it find nothing to take; the next owner waits for them)
  if (e) std::rethrow_exception(e);
}

}  // namespace svo_hip
