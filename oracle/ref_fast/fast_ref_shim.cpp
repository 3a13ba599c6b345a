// fast_ref_shim.cpp -- extern "C" face of the REFERENCE's own FAST code, for oracle/_ref/libfast_ref.so.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; the four translation units it is linked with are compiled from
// where they lie under /root/reference (oracle/ref_fast/Makefile), never copied:
//   src/fast_neon/src/faster_corner_10_sse.cpp:13-202   fast::fast_corner_detect_10_sse2
//   src/fast_neon/src/fast_10.cpp                       fast::fast_corner_detect_10 (images narrower than 22 pixels)
//   src/fast_neon/src/fast_10_score.cpp:21-3148         fast::fast_corner_score_10
//   src/fast_neon/src/nonmax_3x3.cpp:17-112             fast::fast_nonmax_3x3
// They include nothing but <vector> and their own fast/fast.h: no stand-in header, library or generated file is
// involved.  The three calls below are the calls of fd_utils::fastDetector
// (src/svo_direct/src/feature_detection_utils.cpp:160-177), in its order, on one image.
#include <cstddef>
#include <cstdint>
#include <vector>

#include <fast/fast.h>

extern "C" {

// corners / scores / survivors of ONE image.  xy: cap x 2 int32 (x, y), scores: cap int32, nonmax: cap int32 (indices
// into the corner list).  Returns the number of corners (which may exceed cap: then nothing was written) and the
// number of survivors in *n_nonmax.
int fast_ref_detect_score_nonmax(const uint8_t* img, int width, int height, int stride, int threshold,
                                 int32_t* xy, int32_t* scores, int32_t* nonmax, int cap, int32_t* n_nonmax)
{
  std::vector<fast::fast_xy> corners;
  fast::fast_corner_detect_10_sse2((fast::fast_byte*)img, width, height, stride, (short)threshold, corners);
  std::vector<int> sc, nm;
  fast::fast_corner_score_10((fast::fast_byte*)img, stride, corners, threshold, sc);
  fast::fast_nonmax_3x3(corners, sc, nm);
  *n_nonmax = (int32_t)nm.size();
  const int n = (int)corners.size();
  if (n > cap) return n;
  for (int i = 0; i < n; ++i) { xy[2 * i] = corners[(size_t)i].x; xy[2 * i + 1] = corners[(size_t)i].y; scores[i] = sc[(size_t)i]; }
  for (size_t i = 0; i < nm.size(); ++i) nonmax[i] = nm[i];
  return n;
}

// the plain detector of the same file set (what an image narrower than 22 pixels takes): for the SSE2-vs-plain cross check
int fast_ref_detect_plain(const uint8_t* img, int width, int height, int stride, int threshold, int32_t* xy, int cap)
{
  std::vector<fast::fast_xy> corners;
  fast::fast_corner_detect_10((fast::fast_byte*)img, width, height, stride, (short)threshold, corners);
  const int n = (int)corners.size();
  if (n > cap) return n;
  for (int i = 0; i < n; ++i) { xy[2 * i] = corners[(size_t)i].x; xy[2 * i + 1] = corners[(size_t)i].y; }
  return n;
}

}  // extern "C"
