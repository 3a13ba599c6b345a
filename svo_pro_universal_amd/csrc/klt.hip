// klt.hip -- batched pyramidal KLT feature alignment for gfx950 (a-9).
//
// Replaces the per-track loop of FeatureTracker::trackFrameBundle
//   src/svo_tracker/src/feature_tracker.cpp:64-99
// i.e. feature_alignment::alignPyr2D, src/svo_direct/src/feature_alignment.cpp:761-973
// (batch form alignPyr2DVec, :732-758).
//
// One wavefront (64 lanes) owns one track through all pyramid levels and
// iterations; a 16x16 patch gives every lane 4 horizontally adjacent pixels, an
// 8x8 patch one pixel.  The template (u8 value + raw int16 central differences)
// lives in registers.  Everything the reference computes per pixel is integer:
// the 7-bit fixed-point bilinear interpolation, the residual and the products
// res*dx, res*dy; their sums are < 2^24, so the reference's float accumulators
// hold exact integers and a wave-wide integer reduction reproduces them bit for
// bit, independent of order.  The float part (2x2 inverse, update, convergence
// test) is evaluated by every lane identically, in the reference's expression
// order; this file is compiled with -ffp-contract=off so no FMA is formed.
// Memory per track-iteration: (P+1)^2 bytes of the current level (L2-resident).
#include <cstdlib>
#include <cstring>
#include <vector>

#include "svoh_internal.h"

namespace svoh {

struct KltArgs {
  const DevImage* ref_levels;   // n_tracks x SVOH_MAX_LEVELS
  DevImage cur_levels[SVOH_MAX_LEVELS];
  svoh_klt_options opt;
  int n_tracks;
  const int32_t* px_ref;        // 2 x n
  double* px_cur;               // 2 x n, in/out
  uint8_t* status;              // n
};

__device__ __forceinline__ int wave_sum_i32(int v)
{
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// One level of alignPyr2D for patch size P (16 or 8).  Returns: 0 = continue to the
// next level, 1 = return false (not converged / NaN).  `converged` and px_cur are updated.
template <int P>
__device__ __forceinline__ int klt_level(const DevImage& img_ref, const DevImage& img_cur, int level, int px_ref0_x,
                                         int px_ref0_y, int n_iter, float min_update_squared, double& pcx, double& pcy,
                                         bool& converged, int lane)
{
  constexpr int PPL = P * P / 64;  // pixels per lane: 4 (16x16) or 1 (8x8)
  const int halfpatch_size = P / 2;
  const int scale = 1 << level;
  const int width = img_ref.w, height = img_ref.h;
  const int step = img_ref.pitch;
  const float prfx = (float)px_ref0_x / (float)scale - (float)halfpatch_size;
  const float prfy = (float)px_ref0_y / (float)scale - (float)halfpatch_size;
  const int prx = (int)prfx, pry = (int)prfy;
  const float offx = prfx - (float)prx, offy = prfy - (float)pry;
  if (prx < 1 || pry < 1 || prx >= width - P - 1 || pry >= height - P - 1) return 0;  // too close to the border

  // this lane's pixels: row y, columns x0 .. x0+PPL-1
  const int y = (lane * PPL) / P;
  const int x0 = (lane * PPL) % P;
  int tmpl[PPL], gdx[PPL], gdy[PPL];
  int h00 = 0, h01 = 0, h11 = 0;
  {
    const uint8_t* it = img_ref.data + (ptrdiff_t)(pry + y) * step + prx + x0;
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      tmpl[k] = it[k];
      gdx[k] = (int)it[k + 1] - (int)it[k - 1];
      gdy[k] = (int)it[k + step] - (int)it[k - step];
      h00 += gdx[k] * gdx[k];
      h01 += gdx[k] * gdy[k];
      h11 += gdy[k] * gdy[k];
    }
  }
  const float H00 = (float)wave_sum_i32(h00), H01 = (float)wave_sum_i32(h01), H11 = (float)wave_sum_i32(h11);
  const float H10 = H01;
  // Eigen Matrix2f::inverse()
  const float det = H00 * H11 - H10 * H01;
  const float invdet = 1.0f / det;
  const float Hi00 = H11 * invdet, Hi10 = -H10 * invdet, Hi01 = -H01 * invdet, Hi11 = H00 * invdet;

  float u = (float)(pcx / scale - halfpatch_size - offx);
  float v = (float)(pcy / scale - halfpatch_size - offy);
  bool go_to_next_level = false;
  converged = false;
  const int cur_step = img_ref.pitch;  // the reference indexes the current image with the reference's step
  for (int iter = 0; iter < n_iter; ++iter) {
    if (u != u || v != v) return 1;
    go_to_next_level = false;
    const int u_r = (int)floorf(u);
    const int v_r = (int)floorf(v);
    if (u_r < 0 || v_r < 0 || u_r >= width - P || v_r >= height - P) {
      go_to_next_level = true;
      break;
    }
    const float subpix_x = u - u_r;
    const float subpix_y = v - v_r;
    const int wTL = (int)(unsigned short)((1.0f - subpix_x) * (1.0f - subpix_y) * 128);
    const int wTR = (int)(unsigned short)(subpix_x * (1.0f - subpix_y) * 128);
    const int wBL = (int)(unsigned short)((1.0f - subpix_x) * subpix_y * 128);
    const int wBR = (int)(unsigned short)(128 - wTL - wTR - wBL);
    const uint8_t* it = img_cur.data + (ptrdiff_t)(v_r + y) * cur_step + u_r + x0;
    int top[PPL + 1], bot[PPL + 1];
#pragma unroll
    for (int k = 0; k < PPL + 1; ++k) { top[k] = it[k]; bot[k] = it[k + cur_step]; }
    int j0 = 0, j1 = 0;
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      const int cur = (int)(unsigned short)((wTL * top[k] + wTR * top[k + 1] + wBL * bot[k] + wBR * bot[k + 1] + 64) >> 7);
      const int res = cur - tmpl[k];
      j0 += res * gdx[k];
      j1 += res * gdy[k];
    }
    const float Jres0 = -(float)wave_sum_i32(j0);
    const float Jres1 = -(float)wave_sum_i32(j1);
    const float up0 = (Hi00 * Jres0 + Hi01 * Jres1) * 2.0f;
    const float up1 = (Hi10 * Jres0 + Hi11 * Jres1) * 2.0f;
    u += up0;
    v += up1;
    if (up0 * up0 + up1 * up1 < min_update_squared) {
      converged = true;
      break;
    }
  }
  pcx = (double)((u + halfpatch_size + offx) * scale);
  pcy = (double)((v + halfpatch_size + offy) * scale);
  if (!converged && !go_to_next_level) return 1;
  return 0;
}

__global__ __launch_bounds__(64) void klt_track_kernel(const KltArgs a)
{
  const int t = blockIdx.x;
  if (t >= a.n_tracks) return;
  const int lane = threadIdx.x;
  const DevImage* ref = a.ref_levels + (size_t)t * SVOH_MAX_LEVELS;
  double pcx = a.px_cur[2 * t], pcy = a.px_cur[2 * t + 1];
  const int rx = a.px_ref[2 * t], ry = a.px_ref[2 * t + 1];
  bool converged = false;
  bool failed = false;
  for (int level = a.opt.max_level; level >= a.opt.min_level; --level) {
    const int P = a.opt.patch_sizes[level];
    int rc;
    if (P == 16)
      rc = klt_level<16>(ref[level], a.cur_levels[level], level, rx, ry, a.opt.max_iter, a.opt.min_update_squared, pcx,
                         pcy, converged, lane);
    else if (P == 8)
      rc = klt_level<8>(ref[level], a.cur_levels[level], level, rx, ry, a.opt.max_iter, a.opt.min_update_squared, pcx,
                        pcy, converged, lane);
    else if (P == 32)
      rc = 1;  // rejected on the host
    else
      rc = 1;
    if (rc) { failed = true; break; }
  }
  if (lane == 0) {
    a.px_cur[2 * t] = pcx;
    a.px_cur[2 * t + 1] = pcy;
    a.status[t] = (!failed && converged) ? 1 : 0;
  }
}

}  // namespace svoh

using namespace svoh;

extern "C" int svoh_klt_track_batch(svoh_ctx* ctx, const svoh_klt_options* options, int n_tracks,
                                    const svoh_frame_t* ref_frames, svoh_frame_t cur_frame, const int32_t* px_ref,
                                    double* px_cur, uint8_t* status)
{
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, options && n_tracks >= 0, "bad arguments");
  if (n_tracks == 0) return SVOH_OK;
  SVOH_REQUIRE(ctx, ref_frames && px_ref && px_cur && status, "NULL argument");
  SVOH_REQUIRE(ctx, options->max_level >= options->min_level && options->min_level >= 0 &&
                        options->max_level < SVOH_MAX_LEVELS && options->max_iter >= 1,
               "bad KLT level range / max_iter");
  for (int l = options->min_level; l <= options->max_level; ++l)
    if (options->patch_sizes[l] != 8 && options->patch_sizes[l] != 16)
      return set_error(ctx, SVOH_ERR_UNSUPPORTED, "KLT patch size %d at level %d not built (8 and 16 are)",
                       options->patch_sizes[l], l);
  const Frame* fc = find_frame(ctx, cur_frame);
  if (!fc) return set_error(ctx, SVOH_ERR_BAD_HANDLE, "unknown current frame handle");
  SVOH_REQUIRE(ctx, fc->n_levels > options->max_level, "current pyramid has too few levels");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));

  const size_t lv_bytes = sizeof(DevImage) * SVOH_MAX_LEVELS * (size_t)n_tracks;
  const size_t in_bytes = lv_bytes + sizeof(int32_t) * 2 * (size_t)n_tracks;
  const size_t io_bytes = sizeof(double) * 2 * (size_t)n_tracks + (size_t)n_tracks;
  SVOH_HIP_TRY(ctx, ctx->h_scratch0.reserve(in_bytes + io_bytes));
  SVOH_HIP_TRY(ctx, ctx->d_scratch0.reserve(in_bytes + io_bytes));
  uint8_t* h = static_cast<uint8_t*>(ctx->h_scratch0.ptr);
  uint8_t* d = static_cast<uint8_t*>(ctx->d_scratch0.ptr);
  DevImage* hl = reinterpret_cast<DevImage*>(h);
  for (int i = 0; i < n_tracks; ++i) {
    const Frame* fr = find_frame(ctx, ref_frames[i]);
    if (!fr) return set_error(ctx, SVOH_ERR_BAD_HANDLE, "track %d: unknown reference frame handle", i);
    if (fr->n_levels <= options->max_level)
      return set_error(ctx, SVOH_ERR_INVALID_ARGUMENT, "track %d: reference pyramid has too few levels", i);
    for (int l = 0; l < SVOH_MAX_LEVELS; ++l)
      hl[(size_t)i * SVOH_MAX_LEVELS + l] = l < fr->n_levels ? fr->lv[l] : DevImage{ nullptr, 0, 0, 0, 0 };
    for (int l = options->min_level; l <= options->max_level; ++l)
      if (fr->lv[l].w != fc->lv[l].w || fr->lv[l].h != fc->lv[l].h)
        return set_error(ctx, SVOH_ERR_INVALID_ARGUMENT, "track %d: reference and current level %d differ in size", i, l);
  }
  memcpy(h + lv_bytes, px_ref, sizeof(int32_t) * 2 * (size_t)n_tracks);
  memcpy(h + in_bytes, px_cur, sizeof(double) * 2 * (size_t)n_tracks);
  SVOH_HIP_TRY(ctx, hipMemcpyAsync(d, h, in_bytes + sizeof(double) * 2 * (size_t)n_tracks, hipMemcpyHostToDevice,
                                   ctx->stream));
  KltArgs args;
  args.ref_levels = reinterpret_cast<const DevImage*>(d);
  for (int l = 0; l < SVOH_MAX_LEVELS; ++l) args.cur_levels[l] = l < fc->n_levels ? fc->lv[l] : DevImage{ nullptr, 0, 0, 0, 0 };
  args.opt = *options;
  args.n_tracks = n_tracks;
  args.px_ref = reinterpret_cast<const int32_t*>(d + lv_bytes);
  args.px_cur = reinterpret_cast<double*>(d + in_bytes);
  args.status = d + in_bytes + sizeof(double) * 2 * (size_t)n_tracks;
  hipLaunchKernelGGL(klt_track_kernel, dim3(n_tracks), dim3(64), 0, ctx->stream, args);
  SVOH_HIP_TRY(ctx, hipGetLastError());
  SVOH_HIP_TRY(ctx, hipMemcpyAsync(h + in_bytes, d + in_bytes, io_bytes, hipMemcpyDeviceToHost, ctx->stream));
  SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(px_cur, h + in_bytes, sizeof(double) * 2 * (size_t)n_tracks);
  memcpy(status, h + in_bytes + sizeof(double) * 2 * (size_t)n_tracks, (size_t)n_tracks);
  return SVOH_OK;
}
