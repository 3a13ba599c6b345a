// detector.hip -- keyframe feature detector for gfx950.
//
// Replaces FastDetector::detect / FastGradDetector::detect
//   src/svo_direct/src/feature_detection.cpp:113-194
// i.e. fd_utils::fastDetector (feature_detection_utils.cpp:145-195: fast_corner_detect_10 +
// fast_corner_score_10 + fast_nonmax_3x3 of src/fast_neon, best corner per free grid cell),
// fd_utils::edgeletDetector_V2 (:313-385: GaussianBlur 3x3 + Scharr on level 1, 8-neighbour
// non-maximum suppression, histogram angle :831-839, 947-1009) and fd_utils::fillFeatures (:72-143).
//
// The reference makes corner LISTS (detect -> score -> nonmax -> per-cell best); here every
// stage is dense and per pixel, which is what the list algorithms compute:
//   * FAST-10 score map: the largest barrier for which the pixel still is a corner (the value
//     fast_corner_score_10's iteration converges to), 0 where it is not one at the threshold;
//   * a corner survives fast_nonmax_3x3 iff no 8-neighbour that is a corner has a score >= its own;
//   * the per-cell "first strictly better in visiting order" (levels ascending, raster order
//     inside a level) is an atomicMax on (score, ~visit order) packed in 64 bits.
// Everything is integer except the edgelet magnitude (float of a correctly rounded double sqrt of
// an int) and the histogram angle (double; atan2 of the device libm).
// HBM-bound in principle (17 bytes read per pixel and level from a 0.36 MB image that sits in L2);
// at keyframe rate it is launch-latency bound.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "svoh_internal.h"

namespace svoh {

__device__ __constant__ int kCircleDx[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
__device__ __constant__ int kCircleDy[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };

// ---- FAST-10 score map (fast_10_score.cpp:21-3148 by its fixed point) ----
// score(x, y) = max over the 16 arcs of 10 contiguous circle pixels of min(p_i - c) resp. min(c - p_i), minus 1;
// stored as u8 when >= barrier (>= 1), else 0.
__device__ __forceinline__ void fast_score_body(const DevImage& im, int barrier, uint8_t* __restrict__ S)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= im.w || y >= im.h) return;
  int out = 0;
  if (x >= 3 && y >= 3 && x < im.w - 3 && y < im.h - 3) {
    const uint8_t* p = im.data + (size_t)y * im.pitch + x;
    const int c = *p;
    int d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = (int)p[kCircleDx[i] + (ptrdiff_t)im.pitch * kCircleDy[i]] - c;
    // sliding minima of window 10 = 8 + 2 over the circular sequence, for d (brighter) and -d (darker)
    int best = -1000;
#pragma unroll
    for (int sign = 0; sign < 2; ++sign) {
      int v[16], m2[16], m4[16], m8[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = sign ? -d[i] : d[i];
#pragma unroll
      for (int i = 0; i < 16; ++i) m2[i] = min(v[i], v[(i + 1) & 15]);
#pragma unroll
      for (int i = 0; i < 16; ++i) m4[i] = min(m2[i], m2[(i + 2) & 15]);
#pragma unroll
      for (int i = 0; i < 16; ++i) m8[i] = min(m4[i], m4[(i + 4) & 15]);
#pragma unroll
      for (int i = 0; i < 16; ++i) best = max(best, min(m8[i], m2[(i + 8) & 15]));
    }
    const int score = best - 1;
    out = score >= barrier ? score : 0;
  }
  S[(size_t)y * im.w + x] = (uint8_t)out;
}
__global__ __launch_bounds__(256) void fast_score_kernel(DevImage im, int barrier, uint8_t* __restrict__ S) { fast_score_body(im, barrier, S); }
// the same for frame blockIdx.z of a batch: ims[z] is the frame's level, its map starts z * map_stride bytes into S
__global__ __launch_bounds__(256) void fast_score_batch_kernel(const DevImage* __restrict__ ims, int barrier, uint8_t* __restrict__ S, size_t map_stride)
{
  fast_score_body(ims[blockIdx.z], barrier, S + (size_t)blockIdx.z * map_stride);
}

struct GridDesc {
  int cell_size, n_cols, n_rows;
  const uint8_t* occupancy;          // n_cols * n_rows
  unsigned long long* keys;          // n_cols * n_rows, atomicMax targets
};

__device__ __forceinline__ int cell_index(const GridDesc& g, int x, int y, int scale)
{
  // getCellIndex(Eigen::Vector2d(scale*x, scale*y)): floor(py / cell_size) * n_cols + floor(px / cell_size)
  const double px = (double)(scale * x), py = (double)(scale * y);
  return (int)(floor(py / g.cell_size) * g.n_cols + floor(px / g.cell_size));
}

// ---- fast_nonmax_3x3 + border + per-cell best (feature_detection_utils.cpp:176-192) ----
__device__ __forceinline__ void fast_select_body(const uint8_t* __restrict__ S, int w, int h, int level, int border,
                                                 float init_score, const GridDesc& g)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const int s = S[(size_t)y * w + x];
  if (s == 0) return;   // not a corner (corners lie in [3, w-3) x [3, h-3): the 8 neighbours exist)
  const uint8_t* p = S + (size_t)y * w + x;
  if (p[-1] >= s || p[1] >= s || p[-w - 1] >= s || p[-w] >= s || p[-w + 1] >= s || p[w - 1] >= s || p[w] >= s || p[w + 1] >= s)
    return;
  if (x < border || y < border || x >= w - border || y >= h - border) return;
  const int k = cell_index(g, x, y, 1 << level);
  if ((unsigned)k >= (unsigned)(g.n_cols * g.n_rows) || g.occupancy[k]) return;
  if (!((float)s > init_score)) return;   // score > corners.at(k).score, which starts at threshold_primary
  const unsigned order = ((unsigned)level << 28) | ((unsigned)y << 14) | (unsigned)x;   // the reference's visiting order
  const unsigned long long key = ((unsigned long long)(unsigned)s << 32) | (unsigned long long)(0xFFFFFFFFu - order);
  atomicMax(g.keys + k, key);
}
__global__ __launch_bounds__(256) void fast_select_kernel(const uint8_t* __restrict__ S, int w, int h, int level, int border,
                                                          float init_score, GridDesc g)
{
  fast_select_body(S, w, h, level, border, init_score, g);
}
// frame blockIdx.z of a batch (all frames of one size): its map, its occupancy bytes and its keys lie z strides on
__global__ __launch_bounds__(256) void fast_select_batch_kernel(const uint8_t* __restrict__ S, size_t map_stride, int w, int h, int level, int border,
                                                                float init_score, GridDesc g, size_t cell_stride)
{
  g.occupancy += (size_t)blockIdx.z * cell_stride;
  g.keys += (size_t)blockIdx.z * cell_stride;
  fast_select_body(S + (size_t)blockIdx.z * map_stride, w, h, level, border, init_score, g);
}

// ---- edgelets: GaussianBlur 3x3 + Scharr + magnitude (feature_detection_utils.cpp:326-346) ----
__device__ __forceinline__ int reflect101(int i, int n)
{
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

__device__ __forceinline__ void edge_score_body(const DevImage& im, int border, int threshold, float* __restrict__ E)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= im.w || y >= im.h) return;
  float out = 0.0f;
  if (x >= border && y >= border && x < im.w - border && y < im.h - border) {
    // 5x5 raw footprint -> 3x3 blurred ([1 2 1] x [1 2 1], (s + 8) >> 4) -> Scharr
    int raw[5][5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const uint8_t* row = im.data + (size_t)reflect101(y + j - 2, im.h) * im.pitch;
#pragma unroll
      for (int i = 0; i < 5; ++i) raw[j][i] = row[reflect101(x + i - 2, im.w)];
    }
    int b[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int s = (raw[j][i] + 2 * raw[j][i + 1] + raw[j][i + 2]) + 2 * (raw[j + 1][i] + 2 * raw[j + 1][i + 1] + raw[j + 1][i + 2]) +
                      (raw[j + 2][i] + 2 * raw[j + 2][i + 1] + raw[j + 2][i + 2]);
        b[j][i] = (s + 8) >> 4;
      }
    // The blurred image's own border reflection only matters within 1 pixel of the image edge; border >= 1
    // keeps the 3x3 blurred neighbourhood inside the image, where reflecting the raw footprint is the same thing
    // as long as border >= 2.  (Smaller borders are refused on the host.)
    const int gx = 3 * (b[0][2] - b[0][0]) + 10 * (b[1][2] - b[1][0]) + 3 * (b[2][2] - b[2][0]);
    const int gy = 3 * (b[2][0] - b[0][0]) + 10 * (b[2][1] - b[0][1]) + 3 * (b[2][2] - b[0][2]);
    const float mag = (float)sqrt((double)(gx * gx + gy * gy));
    out = (mag > (float)threshold) ? mag : 0.0f;
  }
  E[(size_t)y * im.w + x] = out;
}
__global__ __launch_bounds__(256) void edge_score_kernel(DevImage im, int border, int threshold, float* __restrict__ E) { edge_score_body(im, border, threshold, E); }
__global__ __launch_bounds__(256) void edge_score_batch_kernel(const DevImage* __restrict__ ims, int border, int threshold, float* __restrict__ E, size_t map_stride_floats)
{
  edge_score_body(ims[blockIdx.z], border, threshold, E + (size_t)blockIdx.z * map_stride_floats);
}

// 8-neighbour non-maximum suppression with the reference's asymmetric comparisons + per-cell best (:349-383)
// corner_keys != NULL: a cell that holds a corner (a non-zero key) counts as occupied too -- what the occupancy grid says
// after fd_utils::fillFeatures has marked the corners' cells (svoh_detect_cells_batch: both phases without the host in between)
__device__ __forceinline__ void edge_select_body(const float* __restrict__ E, int w, int h, int border, int threshold,
                                                 float init_score, const GridDesc& g, const unsigned long long* __restrict__ corner_keys)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x < border || y < border || x >= w - border || y >= h - border) return;
  const int k = cell_index(g, x, y, 2);
  if ((unsigned)k >= (unsigned)(g.n_cols * g.n_rows) || g.occupancy[k]) return;
  if (corner_keys && corner_keys[k] != 0ull) return;
  const float* p = E + (size_t)y * w + x;
  const float c = *p;
  if (c < (float)threshold) return;
  if (p[1] >= c || p[-1] > c || p[w] >= c || p[-w] > c || p[w + 1] >= c || p[w - 1] > c || p[-w + 1] >= c || p[-w - 1] > c) return;
  if (!(c > init_score)) return;
  const unsigned order = ((unsigned)y << 14) | (unsigned)x;
  const unsigned long long key = ((unsigned long long)__float_as_uint(c) << 32) | (unsigned long long)(0xFFFFFFFFu - order);
  atomicMax(g.keys + k, key);   // positive floats order like their bit patterns
}
__global__ __launch_bounds__(256) void edge_select_kernel(const float* __restrict__ E, int w, int h, int border, int threshold,
                                                          float init_score, GridDesc g)
{
  edge_select_body(E, w, h, border, threshold, init_score, g, nullptr);
}
__global__ __launch_bounds__(256) void edge_select_batch_kernel(const float* __restrict__ E, size_t map_stride_floats, int w, int h, int border, int threshold,
                                                                float init_score, GridDesc g, const unsigned long long* __restrict__ corner_keys, size_t cell_stride)
{
  g.occupancy += (size_t)blockIdx.z * cell_stride;
  g.keys += (size_t)blockIdx.z * cell_stride;
  edge_select_body(E + (size_t)blockIdx.z * map_stride_floats, w, h, border, threshold, init_score, g, corner_keys + (size_t)blockIdx.z * cell_stride);
}

// getAngleAtPixelUsingHistogram(img_pyr[1], (x, y), 4) for the winner of every cell (:831-839, 947-1009)
template <bool BATCH>
__device__ __forceinline__ void edge_angle_body(const DevImage& im, const unsigned long long* __restrict__ keys, int n_cells,
                                                float* __restrict__ angle)
{
  constexpr int n_bins = 36;
  __shared__ double s_hist[64][n_bins + 1];
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= n_cells) return;
  const unsigned long long key = keys[k];
  if (key == 0ull) { angle[k] = 0.0f; return; }
  const unsigned order = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
  const int px = (int)(order & 0x3FFFu), py = (int)((order >> 14) & 0x3FFFu);
  double* hist = s_hist[threadIdx.x];
  for (int i = 0; i < n_bins; ++i) hist[i] = 0.0;
  const double pi = 3.14159265358979323846, pi2 = 2.0 * 3.14159265358979323846;
  for (int dy = -4; dy <= 4; ++dy)
    for (int dx = -4; dx <= 4; ++dx) {
      const int x = px + dx, y = py + dy;
      if (y > 0 && y < im.h - 1 && x > 0 && x < im.w - 1) {
        const uint8_t* q = im.data + (size_t)y * im.pitch + x;
        const double gx = (double)((int)q[1] - (int)q[-1]);
        const double gy = (double)((int)q[im.pitch] - (int)q[-(ptrdiff_t)im.pitch]);
        const double mag = sqrt(gx * gx + gy * gy);
        const double ang = atan2(gy, gx);
        size_t bin = (size_t)round(n_bins * (ang + pi) / pi2);
        bin = (bin < (size_t)n_bins) ? bin : 0u;
        hist[bin] += mag;
      }
    }
  double prev = hist[n_bins - 1];
  const double h0 = hist[0];
  for (int i = 0; i < n_bins; ++i) {
    const double tmp = hist[i];
    hist[i] = 0.25 * prev + 0.5 * hist[i] + 0.25 * ((i + 1 == n_bins) ? h0 : hist[i + 1]);
    prev = tmp;
  }
  double max_v = hist[0];
  int max_bin = 0;
  for (int i = 1; i < n_bins; ++i)
    if (hist[i] > max_v) { max_v = hist[i]; max_bin = i; }
  angle[k] = (float)(max_bin * 2.0 * pi / n_bins);
}
__global__ __launch_bounds__(64) void edge_angle_kernel(DevImage im, const unsigned long long* __restrict__ keys, int n_cells,
                                                        float* __restrict__ angle)
{
  edge_angle_body<false>(im, keys, n_cells, angle);
}
__global__ __launch_bounds__(64) void edge_angle_batch_kernel(const DevImage* __restrict__ ims, const unsigned long long* __restrict__ keys, int n_cells,
                                                              float* __restrict__ angle, size_t cell_stride)
{
  edge_angle_body<true>(ims[blockIdx.y], keys + (size_t)blockIdx.y * cell_stride, n_cells, angle + (size_t)blockIdx.y * cell_stride);
}

// ---- host side: fd_utils::fillFeatures (feature_detection_utils.cpp:72-143) ----
struct HostCorner { int x, y, level; float score, angle; };

static int fill_features(const std::vector<HostCorner>& corners, int type, const uint8_t* mask, int mask_pitch, double threshold,
                         int max_n_features, int n_old, const svoh_detector_options& opt, int n_cols, std::vector<uint8_t>& occupancy,
                         double* px, double* score, int32_t* level, double* grad, uint8_t* types)
{
  std::vector<int> idx;
  for (size_t k = 0; k < corners.size(); ++k) {
    const HostCorner& c = corners[k];
    if (!((double)c.score > threshold)) continue;
    if (mask && mask[(size_t)c.y * mask_pitch + c.x] == 0) continue;
    idx.push_back((int)k);
    const size_t cell = (size_t)(std::floor((double)c.y / opt.cell_size) * n_cols + std::floor((double)c.x / opt.cell_size));
    occupancy[cell] = 1;
  }
  // bigger score first; equal scores keep cell order (the reference's std::sort leaves them in library order)
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return corners[a].score > corners[b].score; });
  const int n_new = std::min<int>(max_n_features, (int)idx.size());
  for (int i = 0; i < n_new; ++i) {
    const HostCorner& c = corners[idx[i]];
    const int o = n_old + i;
    px[2 * o] = c.x; px[2 * o + 1] = c.y;
    score[o] = c.score; level[o] = c.level;
    grad[2 * o] = (double)std::cos(c.angle); grad[2 * o + 1] = (double)std::sin(c.angle);   // std::cos(float)
    types[o] = (uint8_t)type;
  }
  return n_old + n_new;
}

}  // namespace svoh

using namespace svoh;

extern "C" int svoh_detect_features(svoh_ctx* ctx, svoh_frame_t frame, const svoh_detector_options* options,
                                    const uint8_t* occupancy, const uint8_t* mask, int mask_pitch, int max_n_features,
                                    double* px, double* score, int32_t* level, double* grad, uint8_t* type,
                                    int32_t* n_features)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, options && px && score && level && grad && type && n_features, "NULL argument");
  *n_features = 0;
  const svoh_detector_options& opt = *options;
  SVOH_REQUIRE(ctx, opt.cell_size >= 1 && opt.min_level >= 0 && opt.max_level >= opt.min_level && opt.max_level < SVOH_MAX_LEVELS,
               "bad detector cell size / level range");
  SVOH_REQUIRE(ctx, opt.border >= 3, "detector border must be >= 3 (FAST circle radius; the reference's default is 8)");
  SVOH_REQUIRE(ctx, opt.threshold_primary >= 1.0 && opt.threshold_primary <= 254.0, "threshold_primary out of [1, 254]");
  const Frame* f = find_frame(ctx, frame);
  if (!f) return set_error(ctx, SVOH_ERR_BAD_HANDLE, "unknown frame handle %llu", (unsigned long long)frame);
  SVOH_REQUIRE(ctx, f->n_levels > opt.max_level, "pyramid has too few levels for the detector");
  SVOH_REQUIRE(ctx, !opt.detect_edgelets || f->n_levels > 1, "the edgelet detector works on level 1");
  const int w = f->lv[0].w, h = f->lv[0].h;
  SVOH_REQUIRE(ctx, w < (1 << 14) && h < (1 << 14), "image larger than 16383 pixels a side");
  SVOH_REQUIRE(ctx, !mask || mask_pitch >= w, "mask pitch smaller than the image width");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int n_cols = (int)std::ceil((double)w / opt.cell_size), n_rows = (int)std::ceil((double)h / opt.cell_size);
  const int n_cells = n_cols * n_rows;
  if (max_n_features <= 0) return SVOH_OK;

  // device scratch: [occupancy n_cells | keys 8*n_cells | angles 4*n_cells | maps]
  size_t map_bytes = 0;
  for (int l = opt.min_level; l <= opt.max_level; ++l) map_bytes = std::max(map_bytes, (size_t)f->lv[l].w * f->lv[l].h);
  if (opt.detect_edgelets) map_bytes = std::max(map_bytes, sizeof(float) * (size_t)f->lv[1].w * f->lv[1].h);
  const size_t o_keys = ((size_t)n_cells + 63) & ~(size_t)63;
  const size_t o_angle = o_keys + sizeof(unsigned long long) * (size_t)n_cells;
  const size_t o_map = (o_angle + sizeof(float) * (size_t)n_cells + 255) & ~(size_t)255;
  SVOH_HIP_TRY(ctx, ctx->d_scratch2.reserve(o_map + map_bytes));
  SVOH_HIP_TRY(ctx, ctx->h_scratch1.reserve(o_map));
  uint8_t* d = static_cast<uint8_t*>(ctx->d_scratch2.ptr);
  uint8_t* hs = static_cast<uint8_t*>(ctx->h_scratch1.ptr);
  std::vector<uint8_t> occ((size_t)n_cells, 0);
  if (occupancy) for (int k = 0; k < n_cells; ++k) occ[k] = occupancy[k] ? 1 : 0;

  GridDesc g;
  g.cell_size = opt.cell_size; g.n_cols = n_cols; g.n_rows = n_rows;
  g.occupancy = d;
  g.keys = reinterpret_cast<unsigned long long*>(d + o_keys);
  auto upload_grid = [&]() -> int {
    memcpy(hs, occ.data(), (size_t)n_cells);
    SVOH_HIP_TRY(ctx, hipMemcpyAsync(d, hs, (size_t)n_cells, hipMemcpyHostToDevice, ctx->stream));
    SVOH_HIP_TRY(ctx, hipMemsetAsync(g.keys, 0, sizeof(unsigned long long) * (size_t)n_cells, ctx->stream));
    return SVOH_OK;
  };
  auto fetch_keys = [&](bool with_angles) -> int {
    const size_t bytes = with_angles ? o_map - o_keys : o_angle - o_keys;
    SVOH_HIP_TRY(ctx, hipMemcpyAsync(hs + o_keys, d + o_keys, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SVOH_OK;
  };
  auto grid2d = [](int iw, int ih) { return dim3((unsigned)((iw + 63) / 64), (unsigned)((ih + 3) / 4)); };

  // ---- corners: fd_utils::fastDetector ----
  {
    const int rc = upload_grid();
    if (rc != SVOH_OK) return rc;
  }
  if (ctx->timing_on()) SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_start, ctx->stream));
  for (int l = opt.min_level; l <= opt.max_level; ++l) {
    const DevImage& im = f->lv[l];
    if (im.h < 7 || im.w < 7) continue;
    uint8_t* S = d + o_map;
    hipLaunchKernelGGL(fast_score_kernel, grid2d(im.w, im.h), dim3(256), 0, ctx->stream, im, (int)opt.threshold_primary, S);
    hipLaunchKernelGGL(fast_select_kernel, grid2d(im.w, im.h), dim3(256), 0, ctx->stream, S, im.w, im.h, l, opt.border,
                       (float)opt.threshold_primary, g);
  }
  SVOH_HIP_TRY(ctx, hipGetLastError());
  {
    const int rc = fetch_keys(false);
    if (rc != SVOH_OK) return rc;
  }
  const unsigned long long* hk = reinterpret_cast<const unsigned long long*>(hs + o_keys);
  std::vector<HostCorner> corners((size_t)n_cells);
  for (int k = 0; k < n_cells; ++k) {
    HostCorner c{ 0, 0, 0, (float)opt.threshold_primary, 0.0f };
    if (hk[k]) {
      const unsigned order = 0xFFFFFFFFu - (unsigned)(hk[k] & 0xFFFFFFFFull);
      const int lv = (int)(order >> 28), y = (int)((order >> 14) & 0x3FFFu), x = (int)(order & 0x3FFFu);
      c = HostCorner{ x << lv, y << lv, lv, (float)(unsigned)(hk[k] >> 32), 0.0f };
    }
    corners[k] = c;
  }
  int n = fill_features(corners, SVOH_FT_CORNER, mask, mask_pitch, opt.threshold_primary, max_n_features, 0, opt, n_cols, occ, px, score,
                        level, grad, type);

  // ---- edgelets in the cells that are still free: fd_utils::edgeletDetector_V2 ----
  const int max_features = max_n_features - n;
  if (opt.detect_edgelets && max_features > 0) {
    const int rc = upload_grid();
    if (rc != SVOH_OK) return rc;
    const DevImage& im = f->lv[1];
    float* E = reinterpret_cast<float*>(d + o_map);
    float* d_angle = reinterpret_cast<float*>(d + o_angle);
    hipLaunchKernelGGL(edge_score_kernel, grid2d(im.w, im.h), dim3(256), 0, ctx->stream, im, opt.border, (int)opt.threshold_secondary, E);
    hipLaunchKernelGGL(edge_select_kernel, grid2d(im.w, im.h), dim3(256), 0, ctx->stream, E, im.w, im.h, opt.border,
                       (int)opt.threshold_secondary, (float)opt.threshold_secondary, g);
    hipLaunchKernelGGL(edge_angle_kernel, dim3((unsigned)((n_cells + 63) / 64)), dim3(64), 0, ctx->stream, im, g.keys, n_cells, d_angle);
    SVOH_HIP_TRY(ctx, hipGetLastError());
    const int rc2 = fetch_keys(true);
    if (rc2 != SVOH_OK) return rc2;
    const float* ha = reinterpret_cast<const float*>(hs + o_angle);
    for (int k = 0; k < n_cells; ++k) {
      HostCorner c{ 0, 0, 0, (float)opt.threshold_secondary, 0.0f };
      if (hk[k]) {
        const unsigned order = 0xFFFFFFFFu - (unsigned)(hk[k] & 0xFFFFFFFFull);
        const int y = (int)((order >> 14) & 0x3FFFu), x = (int)(order & 0x3FFFu);
        const unsigned bits = (unsigned)(hk[k] >> 32);
        float sc;
        memcpy(&sc, &bits, sizeof sc);
        c = HostCorner{ x * 2, y * 2, 0, sc, ha[k] };   // level = 1 - 1, coordinates scaled to level 0
      }
      corners[k] = c;
    }
    n = fill_features(corners, SVOH_FT_EDGELET, mask, mask_pitch, opt.threshold_secondary, max_features, n, opt, n_cols, occ, px, score,
                      level, grad, type);
  }
  if (ctx->timing_on()) SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_stop, ctx->stream));
  SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->misc_timed = ctx->timing_on(); ctx->misc_launched = true;
  *n_features = n;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

// ---- the detector for many frames in one round trip (svoh_detect_cells_batch) and its host half --------------------
namespace svoh {
static void decode_cells(const svoh_detector_options& opt, int n_cells, const uint64_t* keys, const float* angles, bool edges, std::vector<HostCorner>& corners)
{
  corners.resize((size_t)n_cells);
  for (int k = 0; k < n_cells; ++k) {
    HostCorner c{ 0, 0, 0, (float)(edges ? opt.threshold_secondary : opt.threshold_primary), 0.0f };
    if (keys[k]) {
      const unsigned order = 0xFFFFFFFFu - (unsigned)(keys[k] & 0xFFFFFFFFull);
      if (!edges) {
        const int lv = (int)(order >> 28), y = (int)((order >> 14) & 0x3FFFu), x = (int)(order & 0x3FFFu);
        c = HostCorner{ x << lv, y << lv, lv, (float)(unsigned)(keys[k] >> 32), 0.0f };
      } else {
        const int y = (int)((order >> 14) & 0x3FFFu), x = (int)(order & 0x3FFFu);
        const unsigned bits = (unsigned)(keys[k] >> 32);
        float sc;
        memcpy(&sc, &bits, sizeof sc);
        c = HostCorner{ x * 2, y * 2, 0, sc, angles[k] };   // level = 1 - 1, coordinates scaled to level 0
      }
    }
    corners[(size_t)k] = c;
  }
}
}  // namespace svoh

extern "C" int svoh_detect_fill_features(const svoh_detector_options* options, int width, int height, const uint64_t* corner_keys,
                                         const uint64_t* edge_keys, const float* edge_angles, int max_n_features, double* px,
                                         double* score, int32_t* level, double* grad, uint8_t* type, int32_t* n_features)
try {
  if (!options || !corner_keys || !px || !score || !level || !grad || !type || !n_features || width <= 0 || height <= 0 || options->cell_size < 1)
    return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "svoh_detect_fill_features: bad arguments");
  *n_features = 0;
  if (max_n_features <= 0) return SVOH_OK;
  const svoh_detector_options& opt = *options;
  const int n_cols = (int)std::ceil((double)width / opt.cell_size), n_rows = (int)std::ceil((double)height / opt.cell_size);
  const int n_cells = n_cols * n_rows;
  std::vector<uint8_t> occ((size_t)n_cells, 0);   // (fillFeatures marks the cells it fills; nobody reads them afterwards here)
  std::vector<HostCorner> corners;
  decode_cells(opt, n_cells, corner_keys, nullptr, false, corners);
  int n = fill_features(corners, SVOH_FT_CORNER, nullptr, 0, opt.threshold_primary, max_n_features, 0, opt, n_cols, occ, px, score, level, grad, type);
  const int max_features = max_n_features - n;
  if (opt.detect_edgelets && max_features > 0) {
    if (!edge_keys || !edge_angles) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "svoh_detect_fill_features: edgelets asked for without their arrays");
    decode_cells(opt, n_cells, edge_keys, edge_angles, true, corners);
    n = fill_features(corners, SVOH_FT_EDGELET, nullptr, 0, opt.threshold_secondary, max_features, n, opt, n_cols, occ, px, score, level, grad, type);
  }
  *n_features = n;
  return SVOH_OK;
} SVOH_ABI_CATCH(nullptr)

// the device half queued, nothing waited for: what comes back stands in ctx->h_detect behind ctx->ev_detect
static int enqueue_detect_cells(svoh_ctx* ctx, int n_frames, const svoh_frame_t* frames, const svoh_detector_options* options, const uint8_t* occupancy)
{
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, n_frames >= 1 && n_frames <= 4096 && frames && options, "bad arguments");
  SVOH_REQUIRE(ctx, !ctx->detect_pending.in_flight, "a detector batch is in flight: svoh_detect_cells_batch_collect first");
  const svoh_detector_options& opt = *options;
  SVOH_REQUIRE(ctx, opt.cell_size >= 1 && opt.min_level >= 0 && opt.max_level >= opt.min_level && opt.max_level < SVOH_MAX_LEVELS,
               "bad detector cell size / level range");
  SVOH_REQUIRE(ctx, opt.border >= 3, "detector border must be >= 3 (FAST circle radius; the reference's default is 8)");
  SVOH_REQUIRE(ctx, opt.threshold_primary >= 1.0 && opt.threshold_primary <= 254.0, "threshold_primary out of [1, 254]");
  const Frame* f0 = find_frame(ctx, frames[0]);
  if (!f0) return set_error(ctx, SVOH_ERR_BAD_HANDLE, "unknown frame handle %llu", (unsigned long long)frames[0]);
  const int w = f0->lv[0].w, h = f0->lv[0].h;
  SVOH_REQUIRE(ctx, w < (1 << 14) && h < (1 << 14), "image larger than 16383 pixels a side");
  SVOH_REQUIRE(ctx, f0->n_levels > opt.max_level, "pyramid has too few levels for the detector");
  SVOH_REQUIRE(ctx, !opt.detect_edgelets || f0->n_levels > 1, "the edgelet detector works on level 1");
  std::vector<const Frame*> fr((size_t)n_frames);
  for (int i = 0; i < n_frames; ++i) {
    fr[(size_t)i] = find_frame(ctx, frames[i]);
    if (!fr[(size_t)i]) return set_error(ctx, SVOH_ERR_BAD_HANDLE, "unknown frame handle %llu", (unsigned long long)frames[i]);
    SVOH_REQUIRE(ctx, fr[(size_t)i]->lv[0].w == w && fr[(size_t)i]->lv[0].h == h && fr[(size_t)i]->n_levels == f0->n_levels, "the frames of a batch must be of one size");
  }
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int n_cols = (int)std::ceil((double)w / opt.cell_size), n_rows = (int)std::ceil((double)h / opt.cell_size);
  const int n_cells = n_cols * n_rows;
  const size_t nf = (size_t)n_frames;
  // one stride for the per-cell arrays (occupancy bytes, corner keys, edge keys, angles): element k of frame z at [z * cell_stride + k]
  const size_t cell_stride = ((size_t)n_cells + 63) & ~(size_t)63;
  size_t map_bytes = 0;
  for (int l = opt.min_level; l <= opt.max_level; ++l) map_bytes = std::max(map_bytes, (size_t)f0->lv[l].w * f0->lv[l].h);
  if (opt.detect_edgelets) map_bytes = std::max(map_bytes, sizeof(float) * (size_t)f0->lv[1].w * f0->lv[1].h);
  map_bytes = (map_bytes + 255) & ~(size_t)255;
  const int n_lv = opt.max_level + 1 > 2 ? opt.max_level + 1 : 2;
  // device: [image table n_lv x n_frames | occupancy | corner keys | edge keys | angles | maps]; the first two travel up, keys and angles come back
  const size_t o_occ = (sizeof(DevImage) * (size_t)n_lv * nf + 255) & ~(size_t)255;
  const size_t o_ck = (o_occ + cell_stride * nf + 255) & ~(size_t)255;
  const size_t o_ek = o_ck + 8 * cell_stride * nf;
  const size_t o_ang = o_ek + 8 * cell_stride * nf;
  const size_t o_map = (o_ang + 4 * cell_stride * nf + 255) & ~(size_t)255;
  // (blocks of the detector's own: between the two halves the caller may make any other call of the context)
  SVOH_HIP_TRY(ctx, ctx->d_detect.reserve(o_map + map_bytes * nf));
  SVOH_HIP_TRY(ctx, ctx->h_detect.reserve(o_map));
  uint8_t* d = static_cast<uint8_t*>(ctx->d_detect.ptr);
  uint8_t* hs = static_cast<uint8_t*>(ctx->h_detect.ptr);
  DevImage* tab = reinterpret_cast<DevImage*>(hs);
  for (int l = 0; l < n_lv; ++l)
    for (int i = 0; i < n_frames; ++i) tab[(size_t)l * nf + i] = l < fr[(size_t)i]->n_levels ? fr[(size_t)i]->lv[l] : DevImage{ nullptr, 0, 0, 0, 0 };
  for (int i = 0; i < n_frames; ++i) {
    uint8_t* o = hs + o_occ + cell_stride * (size_t)i;
    if (occupancy) for (int k = 0; k < n_cells; ++k) o[k] = occupancy[(size_t)i * n_cells + k] ? 1 : 0;
    else memset(o, 0, (size_t)n_cells);
  }
  SVOH_HIP_TRY(ctx, svoh_copy_to_device(ctx, d, hs, o_ck));
  SVOH_HIP_TRY(ctx, hipMemsetAsync(d + o_ck, 0, o_ang - o_ck, ctx->stream));
  const DevImage* d_tab = reinterpret_cast<const DevImage*>(d);
  GridDesc g;
  g.cell_size = opt.cell_size; g.n_cols = n_cols; g.n_rows = n_rows;
  g.occupancy = d + o_occ;
  g.keys = reinterpret_cast<unsigned long long*>(d + o_ck);
  auto grid3d = [&](int iw, int ih) { return dim3((unsigned)((iw + 63) / 64), (unsigned)((ih + 3) / 4), (unsigned)n_frames); };
  if (ctx->timing_on()) SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_start, ctx->stream));
  for (int l = opt.min_level; l <= opt.max_level; ++l) {
    const DevImage& im = f0->lv[l];
    if (im.h < 7 || im.w < 7) continue;
    uint8_t* S = d + o_map;
    hipLaunchKernelGGL(fast_score_batch_kernel, grid3d(im.w, im.h), dim3(256), 0, ctx->stream, d_tab + (size_t)l * nf, (int)opt.threshold_primary, S, map_bytes);
    hipLaunchKernelGGL(fast_select_batch_kernel, grid3d(im.w, im.h), dim3(256), 0, ctx->stream, static_cast<const uint8_t*>(S), map_bytes, im.w, im.h, l, opt.border,
                       (float)opt.threshold_primary, g, cell_stride);
  }
  SVOH_HIP_TRY(ctx, hipGetLastError());
  if (opt.detect_edgelets) {
    const DevImage& im = f0->lv[1];
    float* E = reinterpret_cast<float*>(d + o_map);
    GridDesc ge = g;
    ge.keys = reinterpret_cast<unsigned long long*>(d + o_ek);
    hipLaunchKernelGGL(edge_score_batch_kernel, grid3d(im.w, im.h), dim3(256), 0, ctx->stream, d_tab + nf, opt.border, (int)opt.threshold_secondary, E, map_bytes / sizeof(float));
    hipLaunchKernelGGL(edge_select_batch_kernel, grid3d(im.w, im.h), dim3(256), 0, ctx->stream, static_cast<const float*>(E), map_bytes / sizeof(float), im.w, im.h, opt.border,
                       (int)opt.threshold_secondary, (float)opt.threshold_secondary, ge, static_cast<const unsigned long long*>(g.keys), cell_stride);
    hipLaunchKernelGGL(edge_angle_batch_kernel, dim3((unsigned)((n_cells + 63) / 64), (unsigned)n_frames), dim3(64), 0, ctx->stream, d_tab + nf,
                       static_cast<const unsigned long long*>(ge.keys), n_cells, reinterpret_cast<float*>(d + o_ang), cell_stride);
    SVOH_HIP_TRY(ctx, hipGetLastError());
  }
  if (ctx->timing_on()) SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_stop, ctx->stream));
  ctx->misc_timed = ctx->timing_on(); ctx->misc_launched = true;
  SVOH_HIP_TRY(ctx, svoh_copy_to_host(ctx, hs + o_ck, d + o_ck, (opt.detect_edgelets ? o_map : o_ek) - o_ck));
  if (!ctx->ev_detect) SVOH_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_detect, hipEventDisableTiming));
  SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_detect, ctx->stream));
  svoh_ctx::DetectPending& dp = ctx->detect_pending;
  dp.in_flight = true; dp.n_frames = n_frames; dp.n_cells = n_cells; dp.cell_stride = cell_stride; dp.o_ck = o_ck; dp.o_ek = o_ek; dp.o_ang = o_ang;
  dp.edgelets = opt.detect_edgelets != 0;
  return SVOH_OK;
}

// waits for the batch in flight -- for ITS results, not for what was queued behind it -- and hands them out
static int collect_detect_cells(svoh_ctx* ctx, uint64_t* corner_keys, uint64_t* edge_keys, float* edge_angles)
{
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  svoh_ctx::DetectPending& dp = ctx->detect_pending;
  SVOH_REQUIRE(ctx, dp.in_flight, "no detector batch in flight");
  SVOH_REQUIRE(ctx, corner_keys && (!dp.edgelets || (edge_keys && edge_angles)), "NULL output array (edgelets were asked for: all three)");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  SVOH_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_detect));
  dp.in_flight = false;
  const uint8_t* hs = static_cast<const uint8_t*>(ctx->h_detect.ptr);
  const size_t n_cells = (size_t)dp.n_cells;
  for (int i = 0; i < dp.n_frames; ++i) {
    memcpy(corner_keys + (size_t)i * n_cells, hs + dp.o_ck + 8 * dp.cell_stride * (size_t)i, 8 * n_cells);
    if (dp.edgelets) {
      memcpy(edge_keys + (size_t)i * n_cells, hs + dp.o_ek + 8 * dp.cell_stride * (size_t)i, 8 * n_cells);
      memcpy(edge_angles + (size_t)i * n_cells, hs + dp.o_ang + 4 * dp.cell_stride * (size_t)i, 4 * n_cells);
    }
  }
  return SVOH_OK;
}

extern "C" int svoh_detect_cells_batch_enqueue(svoh_ctx* ctx, int n_frames, const svoh_frame_t* frames, const svoh_detector_options* options, const uint8_t* occupancy)
try {
  return enqueue_detect_cells(ctx, n_frames, frames, options, occupancy);
} SVOH_ABI_CATCH(ctx)

extern "C" int svoh_detect_cells_batch_collect(svoh_ctx* ctx, uint64_t* corner_keys, uint64_t* edge_keys, float* edge_angles)
try {
  return collect_detect_cells(ctx, corner_keys, edge_keys, edge_angles);
} SVOH_ABI_CATCH(ctx)

extern "C" int svoh_detect_cells_batch(svoh_ctx* ctx, int n_frames, const svoh_frame_t* frames, const svoh_detector_options* options,
                                       const uint8_t* occupancy, uint64_t* corner_keys, uint64_t* edge_keys, float* edge_angles)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, corner_keys && options && (!options->detect_edgelets || (edge_keys && edge_angles)), "NULL output array (edgelets asked for: all three)");
  const int rc = enqueue_detect_cells(ctx, n_frames, frames, options, occupancy);
  if (rc != SVOH_OK) return rc;
  return collect_detect_cells(ctx, corner_keys, edge_keys, edge_angles);
} SVOH_ABI_CATCH(ctx)

// ---- getAngleAtPixelUsingHistogram for given pixels of given frames (the refresh of an upgraded edgelet's direction) ---------------
namespace svoh {
struct AngleJob { int32_t frame, level, x, y; };
// one lane per pixel: the dominant bin of the smoothed 36-bin histogram of the 9x9 window's gradient directions, magnitude-weighted
// (feature_detection_utils.cpp:831-839, 947-1009; the arithmetic of edge_angle_body above)
__global__ __launch_bounds__(64) void histogram_angle_bins_kernel(const DevImage* __restrict__ levels /* n_frames x SVOH_MAX_LEVELS */, const AngleJob* __restrict__ jobs,
                                                                  int n, int32_t* __restrict__ bins)
{
  constexpr int n_bins = 36;
  __shared__ double s_hist[64][n_bins + 1];
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= n) return;
  const AngleJob j = jobs[k];
  const DevImage im = levels[(size_t)j.frame * SVOH_MAX_LEVELS + j.level];
  double* hist = s_hist[threadIdx.x];
  for (int i = 0; i < n_bins; ++i) hist[i] = 0.0;
  const double pi = 3.14159265358979323846, pi2 = 2.0 * 3.14159265358979323846;
  for (int dy = -4; dy <= 4; ++dy)
    for (int dx = -4; dx <= 4; ++dx) {
      const int x = j.x + dx, y = j.y + dy;
      if (y > 0 && y < im.h - 1 && x > 0 && x < im.w - 1) {
        const uint8_t* q = im.data + (size_t)y * im.pitch + x;
        const double gx = (double)((int)q[1] - (int)q[-1]);
        const double gy = (double)((int)q[im.pitch] - (int)q[-(ptrdiff_t)im.pitch]);
        const double mag = sqrt(gx * gx + gy * gy);
        const double ang = atan2(gy, gx);
        size_t bin = (size_t)round(n_bins * (ang + pi) / pi2);
        bin = (bin < (size_t)n_bins) ? bin : 0u;
        hist[bin] += mag;
      }
    }
  double prev = hist[n_bins - 1];
  const double h0 = hist[0];
  for (int i = 0; i < n_bins; ++i) {
    const double tmp = hist[i];
    hist[i] = 0.25 * prev + 0.5 * hist[i] + 0.25 * ((i + 1 == n_bins) ? h0 : hist[i + 1]);
    prev = tmp;
  }
  double max_v = hist[0];
  int max_bin = 0;
  for (int i = 1; i < n_bins; ++i)
    if (hist[i] > max_v) { max_v = hist[i]; max_bin = i; }
  bins[k] = max_bin;
}
}  // namespace svoh

extern "C" int svoh_histogram_angle_bins(svoh_ctx* ctx, int n_frames, const svoh_frame_t* frames, int n, const int32_t* frame_idx, const int32_t* level,
                                         const int32_t* px, int32_t* bins)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, n >= 0 && n_frames >= 0 && n_frames <= (1 << 16) && n <= (1 << 24), "bad arguments");
  if (n == 0) return SVOH_OK;
  SVOH_REQUIRE(ctx, frames && frame_idx && level && px && bins && n_frames >= 1, "NULL argument");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
  const size_t o_lv = 0, o_jobs = al(sizeof(DevImage) * SVOH_MAX_LEVELS * (size_t)n_frames), o_bins = o_jobs + al(sizeof(AngleJob) * (size_t)n), total = o_bins + al(4 * (size_t)n);
  SVOH_HIP_TRY(ctx, ctx->h_scratch1.reserve(total));
  SVOH_HIP_TRY(ctx, ctx->d_scratch1.reserve(total));
  uint8_t* h = static_cast<uint8_t*>(ctx->h_scratch1.ptr);
  uint8_t* d = static_cast<uint8_t*>(ctx->d_scratch1.ptr);
  DevImage* lv = reinterpret_cast<DevImage*>(h + o_lv);
  std::vector<int> n_levels((size_t)n_frames);
  for (int f = 0; f < n_frames; ++f) {
    const Frame* fr = find_frame(ctx, frames[f]);
    if (!fr) return set_error(ctx, SVOH_ERR_BAD_HANDLE, "unknown frame handle %llu", (unsigned long long)frames[f]);
    n_levels[(size_t)f] = fr->n_levels;
    for (int l = 0; l < SVOH_MAX_LEVELS; ++l) lv[(size_t)f * SVOH_MAX_LEVELS + l] = l < fr->n_levels ? fr->lv[l] : DevImage{ nullptr, 0, 0, 0, 0 };
  }
  AngleJob* jobs = reinterpret_cast<AngleJob*>(h + o_jobs);
  for (int k = 0; k < n; ++k) {
    SVOH_REQUIRE(ctx, frame_idx[k] >= 0 && frame_idx[k] < n_frames && level[k] >= 0 && level[k] < n_levels[(size_t)frame_idx[k]], "frame index or level out of range");
    jobs[k] = AngleJob{ frame_idx[k], level[k], px[2 * k], px[2 * k + 1] };
  }
  SVOH_HIP_TRY(ctx, svoh_copy_to_device(ctx, d, h, o_bins));
  hipLaunchKernelGGL(histogram_angle_bins_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, reinterpret_cast<const DevImage*>(d + o_lv),
                     reinterpret_cast<const AngleJob*>(d + o_jobs), n, reinterpret_cast<int32_t*>(d + o_bins));
  SVOH_HIP_TRY(ctx, hipGetLastError());
  SVOH_HIP_TRY(ctx, svoh_copy_to_host(ctx, h + o_bins, d + o_bins, 4 * (size_t)n));
  SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(bins, h + o_bins, 4 * (size_t)n);
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)
