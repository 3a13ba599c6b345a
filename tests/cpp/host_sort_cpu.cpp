// reprojector_utils::sortCandidatesByReprojStats of the host mirror (one 128-bit comparison per pair of candidates,
// svo_hip_host.cpp) against the reference's call -- std::sort of the candidates themselves with the three-field lambda
// (reprojector.cpp:545-556) -- on lists made to hurt: few distinct types and counts (long runs of ties, whose order is
// whatever introsort leaves and must be the same), scores that are equal, negative, -0.0 / +0.0, denormal, infinite, and in
// some lists NaN.  No GPU call.  Prints "ok <lists> <candidates>" or the first difference.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <limits>
#include <memory>
#include <vector>
#include "../../svo_pro_universal_amd/host/svo_hip_host.h"
using namespace svo_hip;
int main()
{
  uint64_t st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  auto frame = std::make_shared<Frame>();
  const double specials[] = { 0.0, -0.0, 1.0, -1.0, 5e-324, -5e-324, std::numeric_limits<double>::infinity(), -std::numeric_limits<double>::infinity(),
                              20.0, 20.0, 37.0, 1e300, -1e300 };
  size_t total = 0;
  int lists = 0;
  for (int it = 0; it < 400; ++it) {
    const size_t n = it < 5 ? (size_t)it : 1 + rnd() % 1500;
    const bool with_nan = it % 7 == 3;
    const int n_types = 1 + (int)(rnd() % 4), n_counts = 1 + (int)(rnd() % 5);
    std::vector<reprojector::Candidate> a(n);
    for (size_t i = 0; i < n; ++i) {
      a[i].ref_frame = frame; a[i].ref_index = i;
      a[i].type = (uint8_t)(rnd() % n_types);
      a[i].n_reproj = (rnd() % 97 == 0) ? (rnd() % 2 ? 2147483647 - (int)(rnd() % 2) : -2147483647 - 1 + (int)(rnd() % 2)) : (int)(rnd() % n_counts) - 2;   // (the extremes set, not added: signed overflow is undefined -- found by tests/san)
      const uint64_t pick = rnd() % 10;
      a[i].score = pick < 4 ? specials[rnd() % (sizeof specials / sizeof specials[0])] : pick < 8 ? (double)(rnd() % 40) : (double)(int64_t)rnd() * 1e-12;
      if (with_nan && rnd() % 11 == 0) a[i].score = std::numeric_limits<double>::quiet_NaN();
    }
    std::vector<reprojector::Candidate> b = a;
    reprojector_utils::sortCandidatesByReprojStats(a);
    std::sort(b.begin(), b.end(), [](const reprojector::Candidate& lhs, const reprojector::Candidate& rhs) {
      return lhs.type > rhs.type || (lhs.type == rhs.type && lhs.n_reproj > rhs.n_reproj) ||
             (lhs.type == rhs.type && lhs.n_reproj == rhs.n_reproj && lhs.score > rhs.score);
    });
    for (size_t i = 0; i < n; ++i)
      if (a[i].ref_index != b[i].ref_index) { printf("list %d (n %zu, nan %d): position %zu holds candidate %zu, the reference's call leaves %zu\n", it, n, (int)with_nan, i, a[i].ref_index, b[i].ref_index); return 1; }
    total += n; ++lists;
  }
  printf("ok %d %zu\n", lists, total);
  return 0;
}
