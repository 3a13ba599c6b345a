/* svo_oracle_detector.c -- CPU restatement of the keyframe feature detector.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 * PARITY: the FAST stage (orc_fast_corner_detect_10 / _score_10 / _nonmax_3x3) is PINNED against the reference's own
 * code -- src/fast_neon compiles here without any third-party header (oracle/ref_fast/Makefile -> oracle/_ref/),
 * tests/golden/fast_ref.npz holds its outputs, tests/test_fast_ref_cpu.py compares: 80 (image, threshold) cases,
 * 378 452 corners, every list bit for bit.  Everything after it (grid step, edgelets, fillFeatures) is UNPINNED: it
 * needs OpenCV / Eigen / glog, the reference holds no golden vector for it and cannot be built here.
 *
 * Follows
 *   FastDetector::detect / FastGradDetector::detect   src/svo_direct/src/feature_detection.cpp:113-194
 *   fd_utils::fastDetector                             src/svo_direct/src/feature_detection_utils.cpp:145-195
 *   fd_utils::edgeletDetector_V2                       :313-385
 *   fd_utils::fillFeatures                             :72-143
 *   getAngleAtPixelUsingHistogram + angle_hist::*      :831-839, 947-1009
 *   fast::fast_corner_detect_10_sse2                   src/fast_neon/src/faster_corner_10_sse.cpp:13-202
 *   fast::fast_corner_score_10                         src/fast_neon/src/fast_10_score.cpp:21-3148
 *   fast::fast_nonmax_3x3                              src/fast_neon/src/nonmax_3x3.cpp:17-112
 *   OccupandyGrid2D::getCellIndex                      src/svo_common/include/svo/common/occupancy_grid_2d.h:82-94
 *
 * FAST: the generated decision trees of the reference test "at least 10 contiguous pixels of the 16-pixel
 * Bresenham circle are all brighter than c + b or all darker than c - b" (strict); the score's iteration
 * (b += min_diff until the test fails, return b - 1) converges to the largest barrier for which the pixel
 * still is a corner.  Both are restated by that definition.
 * Third-party arithmetic outside /root/reference (OpenCV 4.x, README.md:36): cv::GaussianBlur(3x3, sigma 0) on
 * 8U = [1 2 1]x[1 2 1]/16 in fixed point, rounded half up; cv::Scharr 16S = [-1 0 1] x [3 10 3], exact;
 * BORDER_REFLECT_101.  std::sort in fillFeatures is not stable: the order among equal scores is the
 * library's; here ties keep cell order (the selected SET only differs when max_n_features cuts inside a tie).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "svo_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static const int kCircle[16][2] = {  /* fast_10_score.cpp:3158-3175: offset[i] = dx + stride*dy */
  { 0, 3 }, { 1, 3 }, { 2, 2 }, { 3, 1 }, { 3, 0 }, { 3, -1 }, { 2, -2 }, { 1, -3 },
  { 0, -3 }, { -1, -3 }, { -2, -2 }, { -3, -1 }, { -3, 0 }, { -3, 1 }, { -2, 2 }, { -1, 3 },
};

/* largest barrier b for which p is a FAST-10 corner (-1 if it is not even one at b = 0) */
static int fast10_max_barrier(const uint8_t* p, int stride)
{
  int d[16];
  const int c = *p;
  for (int i = 0; i < 16; ++i) d[i] = (int)p[kCircle[i][0] + stride * kCircle[i][1]] - c;
  int best = -1000;
  for (int s = 0; s < 16; ++s) {
    int mb = 1000, md = 1000;
    for (int k = 0; k < 10; ++k) {
      const int v = d[(s + k) & 15];
      if (v < mb) mb = v;       /* brighter arc: all v > b  <=>  b < min v */
      if (-v < md) md = -v;     /* darker arc */
    }
    if (mb > best) best = mb;
    if (md > best) best = md;
  }
  return best - 1;
}

int orc_fast_corner_detect_10(const uint8_t* img, int w, int h, int stride, int barrier, int32_t* xy, int cap)
{
  int n = 0;
  if (h < 7) return 0;  /* faster_corner_10_sse.cpp:193-195 (narrow images take the plain detector: same set) */
  for (int y = 3; y < h - 3; ++y)
    for (int x = 3; x < w - 3; ++x)
      if (fast10_max_barrier(img + (size_t)y * stride + x, stride) >= barrier) {
        if (n < cap) { xy[2 * n] = x; xy[2 * n + 1] = y; }
        ++n;
      }
  return n;
}

void orc_fast_corner_score_10(const uint8_t* img, int stride, const int32_t* xy, int n, int barrier, int32_t* scores)
{
  (void)barrier;  /* known to be a corner at `barrier`: the result is >= barrier */
  for (int i = 0; i < n; ++i) scores[i] = fast10_max_barrier(img + (size_t)xy[2 * i + 1] * stride + xy[2 * i], stride);
}

/* nonmax_3x3.cpp:17-112, literally (row_start / point_above / point_below bookkeeping) */
int orc_fast_nonmax_3x3(const int32_t* xy, const int32_t* scores, int n, int32_t* nonmax)
{
  int n_out = 0;
  if (n < 1) return 0;
  const int last_row = xy[2 * (n - 1) + 1];
  int* row_start = (int*)malloc(sizeof(int) * (size_t)(last_row + 1));
  for (int i = 0; i <= last_row; ++i) row_start[i] = -1;
  int prev_row = -1;
  for (int i = 0; i < n; ++i)
    if (xy[2 * i + 1] != prev_row) { row_start[xy[2 * i + 1]] = i; prev_row = xy[2 * i + 1]; }
  int point_above = 0, point_below = 0;
  for (int i = 0; i < n; ++i) {
    const int score = scores[i];
    const int px = xy[2 * i], py = xy[2 * i + 1];
    int suppressed = 0;
    if (i > 0 && xy[2 * (i - 1)] == px - 1 && xy[2 * (i - 1) + 1] == py && scores[i - 1] >= score) continue;
    if (i < n - 1 && xy[2 * (i + 1)] == px + 1 && xy[2 * (i + 1) + 1] == py && scores[i + 1] >= score) continue;
    if (py != 0 && row_start[py - 1] != -1) {
      if (xy[2 * point_above + 1] < py - 1) point_above = row_start[py - 1];
      for (; xy[2 * point_above + 1] < py && xy[2 * point_above] < px - 1; point_above++) {}
      for (int j = point_above; xy[2 * j + 1] < py && xy[2 * j] <= px + 1; j++) {
        const int x = xy[2 * j];
        if ((x == px - 1 || x == px || x == px + 1) && scores[j] >= score) { suppressed = 1; break; }
      }
      if (suppressed) continue;
    }
    if (py != last_row && row_start[py + 1] != -1 && point_below < n) {
      if (xy[2 * point_below + 1] < py + 1) point_below = row_start[py + 1];
      for (; point_below < n && xy[2 * point_below + 1] == py + 1 && xy[2 * point_below] < px - 1; point_below++) {}
      for (int j = point_below; j < n && xy[2 * j + 1] == py + 1 && xy[2 * j] <= px + 1; j++) {
        const int x = xy[2 * j];
        if ((x == px - 1 || x == px || x == px + 1) && scores[j] >= score) { suppressed = 1; break; }
      }
      if (suppressed) continue;
    }
    nonmax[n_out++] = i;
  }
  free(row_start);
  return n_out;
}

static size_t cell_index(int x, int y, int scale, int cell_size, int n_cols)
{
  /* getCellIndex(Eigen::Vector2d(scale * x, scale * y)) */
  const double px = (double)(scale * x), py = (double)(scale * y);
  return (size_t)(floor(py / cell_size) * n_cols + floor(px / cell_size));
}

/* feature_detection_utils.cpp:145-195 */
void orc_fast_detector(const orc_pyramid* pyr, int threshold, int border, int min_level, int max_level,
                       orc_corner* corners, const uint8_t* occupancy, int cell_size, int n_cols)
{
  for (int level = min_level; level <= max_level; ++level) {
    const orc_image* im = &pyr->level[level];
    const int scale = 1 << level;
    const int cap = im->width * im->height;
    int32_t* xy = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)cap);
    const int n = orc_fast_corner_detect_10(im->data, im->width, im->height, im->pitch, threshold, xy, cap);
    int32_t* scores = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int32_t* nm = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    orc_fast_corner_score_10(im->data, im->pitch, xy, n, threshold, scores);
    const int n_nm = orc_fast_nonmax_3x3(xy, scores, n, nm);
    const int maxw = im->width - border, maxh = im->height - border;
    for (int q = 0; q < n_nm; ++q) {
      const int i = nm[q];
      const int x = xy[2 * i], y = xy[2 * i + 1];
      if (x < border || y < border || x >= maxw || y >= maxh) continue;
      const size_t k = cell_index(x, y, scale, cell_size, n_cols);
      if (occupancy[k]) continue;
      const float score = (float)scores[i];
      if (score > corners[k].score) {
        corners[k].x = x * scale; corners[k].y = y * scale; corners[k].score = score; corners[k].level = level;
        corners[k].angle = 0.0f;
      }
    }
    free(xy); free(scores); free(nm);
  }
}

static int reflect101(int i, int n)
{
  if (n == 1) return 0;
  while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
  return i;
}

/* cv::GaussianBlur(src, dst, Size(3,3), 0) on CV_8UC1, BORDER_DEFAULT */
void orc_gaussian_blur_3x3(const orc_image* src, uint8_t* dst)
{
  for (int y = 0; y < src->height; ++y)
    for (int x = 0; x < src->width; ++x) {
      int s = 0;
      for (int j = -1; j <= 1; ++j)
        for (int i = -1; i <= 1; ++i)
          s += (2 - abs(i)) * (2 - abs(j)) * (int)src->data[(size_t)reflect101(y + j, src->height) * src->pitch + reflect101(x + i, src->width)];
      dst[(size_t)y * src->width + x] = (uint8_t)((s + 8) >> 4);
    }
}

/* cv::Scharr(img, d, CV_16S, dx_order, dy_order, 1, 0, BORDER_DEFAULT); img is w x h, pitch w */
void orc_scharr_16s(const uint8_t* img, int w, int h, int x_derivative, int16_t* dst)
{
  static const int sm[3] = { 3, 10, 3 }, de[3] = { -1, 0, 1 };
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int j = -1; j <= 1; ++j)
        for (int i = -1; i <= 1; ++i) {
          const int kx = x_derivative ? de[i + 1] : sm[i + 1];
          const int ky = x_derivative ? sm[j + 1] : de[j + 1];
          s += kx * ky * (int)img[(size_t)reflect101(y + j, h) * w + reflect101(x + i, w)];
        }
      dst[(size_t)y * w + x] = (int16_t)s;
    }
}

/* feature_detection_utils.cpp:831-839, 947-1009 */
double orc_angle_at_pixel_using_histogram(const orc_image* img, int px, int py, int halfpatch_size)
{
  enum { n_bins = 36 };
  double hist[n_bins];
  const double pi2 = 2.0 * M_PI;
  for (int i = 0; i < n_bins; ++i) hist[i] = 0.0;
  for (int dy = -halfpatch_size; dy <= halfpatch_size; ++dy)
    for (int dx = -halfpatch_size; dx <= halfpatch_size; ++dx) {
      const int x = px + dx, y = py + dy;
      if (y > 0 && y < img->height - 1 && x > 0 && x < img->width - 1) {
        const double gx = (double)((int)img->data[(size_t)y * img->pitch + x + 1] - (int)img->data[(size_t)y * img->pitch + x - 1]);
        const double gy = (double)((int)img->data[(size_t)(y + 1) * img->pitch + x] - (int)img->data[(size_t)(y - 1) * img->pitch + x]);
        const double mag = sqrt(gx * gx + gy * gy);
        const double angle = atan2(gy, gx);
        size_t bin = (size_t)round(n_bins * (angle + M_PI) / pi2);
        bin = (bin < n_bins) ? bin : 0u;
        hist[bin] += mag;
      }
    }
  {
    double prev = hist[n_bins - 1];
    const double h0 = hist[0];
    for (int i = 0; i < n_bins; ++i) {
      const double tmp = hist[i];
      hist[i] = 0.25 * prev + 0.5 * hist[i] + 0.25 * ((i + 1 == n_bins) ? h0 : hist[i + 1]);
      prev = tmp;
    }
  }
  double max_v = hist[0];
  int max_bin = 0;
  for (int i = 1; i < n_bins; ++i)
    if (hist[i] > max_v) { max_v = hist[i]; max_bin = i; }
  return max_bin * 2.0 * M_PI / n_bins;
}

/* feature_detection_utils.cpp:313-385: level 1 only, coordinates doubled, level 0 assigned */
void orc_edgelet_detector_v2(const orc_pyramid* pyr, int threshold, int border, orc_corner* corners,
                             const uint8_t* occupancy, int cell_size, int n_cols)
{
  enum { level = 1, scale = 2 };
  const orc_image* im = &pyr->level[level];
  const int w = im->width, h = im->height;
  uint8_t* blur = (uint8_t*)malloc((size_t)w * h);
  int16_t* dx = (int16_t*)malloc(sizeof(int16_t) * (size_t)w * h);
  int16_t* dy = (int16_t*)malloc(sizeof(int16_t) * (size_t)w * h);
  float* score = (float*)calloc((size_t)w * h, sizeof(float));
  orc_gaussian_blur_3x3(im, blur);
  orc_scharr_16s(blur, w, h, 1, dx);
  orc_scharr_16s(blur, w, h, 0, dy);
  for (int y = border; y < h - border; ++y)
    for (int x = border; x < w - border; ++x) {
      const int gx = dx[(size_t)y * w + x], gy = dy[(size_t)y * w + x];
      const float mag = (float)sqrt((double)(gx * gx + gy * gy));   /* std::sqrt(int) is the double overload */
      score[(size_t)y * w + x] = (mag > threshold) ? mag : 0.0f;
    }
  for (int y = border; y < h - border; ++y)
    for (int x = border; x < w - border; ++x) {
      const float* p = &score[(size_t)y * w + x];
      const size_t k = cell_index(x, y, scale, cell_size, n_cols);
      if (occupancy[k]) continue;
      const float c = *p;
      if (c < threshold) continue;
      if (p[1] >= c) continue;
      if (p[-1] > c) continue;
      if (p[w] >= c) continue;
      if (p[-w] > c) continue;
      if (p[w + 1] >= c) continue;
      if (p[w - 1] > c) continue;
      if (p[-w + 1] >= c) continue;
      if (p[-w - 1] > c) continue;
      orc_corner* cc = &corners[k];
      if (c > cc->score) {
        cc->x = x * scale; cc->y = y * scale; cc->level = level - 1; cc->score = c;
        cc->angle = (float)orc_angle_at_pixel_using_histogram(im, x, y, 4);
      }
    }
  free(blur); free(dx); free(dy); free(score);
}

typedef struct { float score; int idx; } sort_item;
static int cmp_desc(const void* a, const void* b)
{
  const sort_item* x = (const sort_item*)a; const sort_item* y = (const sort_item*)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  return x->idx - y->idx;   /* ties keep cell order (std::sort leaves them in library order) */
}

/* feature_detection_utils.cpp:72-143; appends at n_old, returns the new total */
static int fill_features(const orc_corner* corners, int n_cells, int type, const uint8_t* mask, int mask_pitch,
                         double threshold, int max_n_features, int n_old, double* px, double* score, int32_t* level,
                         double* grad, uint8_t* types, uint8_t* occupancy, int cell_size, int n_cols)
{
  sort_item* items = (sort_item*)malloc(sizeof(sort_item) * (size_t)(n_cells > 0 ? n_cells : 1));
  int n = 0;
  for (int k = 0; k < n_cells; ++k) {
    const orc_corner* c = &corners[k];
    if (!((double)c->score > threshold)) continue;
    if (mask && mask[(size_t)c->y * mask_pitch + c->x] == 0) continue;
    items[n].score = c->score; items[n].idx = k; ++n;
    occupancy[cell_index(c->x, c->y, 1, cell_size, n_cols)] = 1;
  }
  qsort(items, (size_t)n, sizeof(sort_item), cmp_desc);
  const int n_new = n < max_n_features ? n : max_n_features;
  for (int i = 0; i < n_new; ++i) {
    const orc_corner* c = &corners[items[i].idx];
    const int o = n_old + i;
    px[2 * o] = c->x; px[2 * o + 1] = c->y;
    score[o] = c->score; level[o] = c->level;
    grad[2 * o] = (double)cosf(c->angle); grad[2 * o + 1] = (double)sinf(c->angle);   /* std::cos(float) */
    types[o] = (uint8_t)type;
  }
  free(items);
  return n_old + n_new;
}

/* FastDetector::detect (feature_detection.cpp:113-131) / FastGradDetector::detect (:155-194) */
int orc_detect_features(const orc_pyramid* pyr, const svoh_detector_options* opt, const uint8_t* occupancy_in,
                        const uint8_t* mask, int mask_pitch, int max_n_features, double* px, double* score,
                        int32_t* level, double* grad, uint8_t* type)
{
  const int w = pyr->level[0].width, h = pyr->level[0].height;
  const int n_cols = (int)ceil((double)w / opt->cell_size), n_rows = (int)ceil((double)h / opt->cell_size);
  const int n_cells = n_cols * n_rows;
  uint8_t* occ = (uint8_t*)malloc((size_t)n_cells);
  if (occupancy_in) memcpy(occ, occupancy_in, (size_t)n_cells); else memset(occ, 0, (size_t)n_cells);
  orc_corner* corners = (orc_corner*)malloc(sizeof(orc_corner) * (size_t)n_cells);
  for (int k = 0; k < n_cells; ++k) { corners[k].x = corners[k].y = corners[k].level = 0; corners[k].score = (float)opt->threshold_primary; corners[k].angle = 0.0f; }
  orc_fast_detector(pyr, (int)opt->threshold_primary, opt->border, opt->min_level, opt->max_level, corners, occ, opt->cell_size, n_cols);
  int n = fill_features(corners, n_cells, SVOH_FT_CORNER, mask, mask_pitch, opt->threshold_primary, max_n_features, 0, px, score,
                        level, grad, type, occ, opt->cell_size, n_cols);
  if (opt->detect_edgelets) {
    const int max_features = max_n_features - n;
    if (max_features > 0) {
      for (int k = 0; k < n_cells; ++k) { corners[k].x = corners[k].y = corners[k].level = 0; corners[k].score = (float)opt->threshold_secondary; corners[k].angle = 0.0f; }
      orc_edgelet_detector_v2(pyr, (int)opt->threshold_secondary, opt->border, corners, occ, opt->cell_size, n_cols);
      n = fill_features(corners, n_cells, SVOH_FT_EDGELET, mask, mask_pitch, opt->threshold_secondary, max_features, n, px, score,
                        level, grad, type, occ, opt->cell_size, n_cols);
    }
  }
  free(occ); free(corners);
  return n;
}
