// pool_race.cpp -- WorkerPool (host/svo_hip_pool.h) under ThreadSanitizer: the host phases of the lock-step front end are
// back-to-back runs of a few microseconds each, with items that write neighbouring slots of shared arrays, pools of
// several groups side by side (pools of their own, a SharedPool, an ExclusivePool taken in turns), phases with fewer items than threads, exceptions out of items, idle threads that have
// gone to sleep.  No device call.  Prints "ok" and exits 0; ThreadSanitizer makes the exit code non-zero on a report.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../../svo_pro_universal_amd/host/svo_hip_pool.h"

using svo_hip::WorkerPool;

static void one_group(int n_threads, int n_rounds, long* checksum)
{
  WorkerPool pool(n_threads);
  std::vector<long> slots(67, 0), sums(67, 0);
  long total = 0;
  for (int r = 0; r < n_rounds; ++r) {
    const int n = 1 + (r * 7) % 67;   // 1 .. 67 items: fewer than threads, equal, many more
    // phase 1: every item writes its own slot (plain stores: the pool's hand-over must order them for phase 2)
    pool.run(n, [&](int i) { slots[(size_t)i] = (long)r * 1000 + i; });
    // phase 2: every item reads its neighbours' slots of phase 1
    pool.run(n, [&](int i) { sums[(size_t)i] = slots[(size_t)i] + slots[(size_t)((i + 1) % n)]; });
    for (int i = 0; i < n; ++i) total += sums[(size_t)i];   // the caller reads what the workers wrote
    if (r % 97 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(3));   // long enough for the workers to fall asleep
    if (r % 53 == 0) {   // an item throws: run() rethrows once all items are done, the pool stays usable
      bool caught = false;
      try { pool.run(n, [&](int i) { if (i == n / 2) throw std::runtime_error("item"); slots[(size_t)i] = -1; }); }
      catch (const std::runtime_error&) { caught = true; }
      if (!caught) { fprintf(stderr, "exception lost\n"); abort(); }
    }
  }
  *checksum = total;
}

// the same phases by several callers at once on ONE SharedPool (one caller per lock-step group)
static void shared_group(svo_hip::SharedPool* pool, int seed, int n_rounds, long* checksum)
{
  std::vector<long> slots(67, 0), sums(67, 0);
  long total = 0;
  for (int r = 0; r < n_rounds; ++r) {
    const int n = 1 + (r * 7) % 67;
    pool->run(n, [&](int i) { slots[(size_t)i] = (long)r * 1000 + i; }, seed);
    pool->run(n, [&](int i) { sums[(size_t)i] = slots[(size_t)i] + slots[(size_t)((i + 1) % n)]; }, seed);
    for (int i = 0; i < n; ++i) total += sums[(size_t)i];
    if (r % 97 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(3));
    if (r % 53 == 0) {
      bool caught = false;
      try { pool->run(n, [&](int i) { if (i == n / 2) throw std::runtime_error("item"); slots[(size_t)i] = -1; }, seed); }
      catch (const std::runtime_error&) { caught = true; }
      if (!caught) { fprintf(stderr, "exception lost\n"); abort(); }
    }
  }
  *checksum = total;
}

// ... and by several callers that take ONE ExclusivePool in turns, a phase at a time
static void exclusive_group(svo_hip::ExclusivePool* pool, int n_rounds, long* checksum)
{
  std::vector<long> slots(67, 0), sums(67, 0);
  long total = 0;
  for (int r = 0; r < n_rounds; ++r) {
    const int n = 1 + (r * 7) % 67;
    pool->run(n, [&](int i) { slots[(size_t)i] = (long)r * 1000 + i; });
    pool->run(n, [&](int i) { sums[(size_t)i] = slots[(size_t)i] + slots[(size_t)((i + 1) % n)]; });
    for (int i = 0; i < n; ++i) total += sums[(size_t)i];
    if (r % 97 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(3));
    if (r % 53 == 0) {
      bool caught = false;
      try { pool->run(n, [&](int i) { if (i == n / 2) throw std::runtime_error("item"); slots[(size_t)i] = -1; }); }
      catch (const std::runtime_error&) { caught = true; }
      if (!caught) { fprintf(stderr, "exception lost\n"); abort(); }
    }
  }
  *checksum = total;
}

int main(int argc, char** argv)
{
  const int n_rounds = argc > 1 ? atoi(argv[1]) : 3000;
  if (getenv("SVOH_LOCKSTEP_SPIN") == nullptr) setenv("SVOH_LOCKSTEP_SPIN", "300", 1);   // idle threads reach the sleeping path often
  long a = 0, b = 0, c = 0, d = 0;
  one_group(1, 200, &d);   // a pool of the caller alone
  std::thread g1(one_group, 4, n_rounds, &a), g2(one_group, 3, n_rounds, &b);   // two groups side by side, as the tool runs them
  one_group(6, n_rounds, &c);
  g1.join(); g2.join();
  long e = 0, f = 0, g = 0;
  {
    svo_hip::SharedPool shared(5);
    std::thread s1(shared_group, &shared, 0, n_rounds, &e), s2(shared_group, &shared, 8, n_rounds, &f);
    shared_group(&shared, 16, n_rounds, &g);
    s1.join(); s2.join();
  }
  long x1 = 0, x2 = 0, x3 = 0;
  {
    svo_hip::ExclusivePool excl(5);
    std::thread t1(exclusive_group, &excl, n_rounds, &x1), t2(exclusive_group, &excl, n_rounds, &x2);
    exclusive_group(&excl, n_rounds, &x3);
    t1.join(); t2.join();
  }
  // the same arithmetic without threads
  auto expect = [](int n_rounds_) {
    long total = 0;
    for (int r = 0; r < n_rounds_; ++r) {
      const int n = 1 + (r * 7) % 67;
      for (int i = 0; i < n; ++i) total += ((long)r * 1000 + i) + ((long)r * 1000 + (i + 1) % n);
    }
    return total;
  };
  // (after a throwing round the slots of that round hold -1 or the old value, but phase 1 of the next round rewrites every slot it reads)
  if (a != expect(n_rounds) || b != expect(n_rounds) || c != expect(n_rounds) || d != expect(200)) { fprintf(stderr, "checksum mismatch\n"); return 1; }
  if (e != expect(n_rounds) || f != expect(n_rounds) || g != expect(n_rounds)) { fprintf(stderr, "checksum mismatch (shared pool)\n"); return 1; }
  if (x1 != expect(n_rounds) || x2 != expect(n_rounds) || x3 != expect(n_rounds)) { fprintf(stderr, "checksum mismatch (exclusive pool)\n"); return 1; }
  printf("ok\n");
  return 0;
}
