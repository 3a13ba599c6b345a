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
#include <unordered_map>
#include <vector>

#include "svoh_internal.h"

namespace svoh {

struct KltArgs {
  const DevImage* frame_levels; // n_frames x SVOH_MAX_LEVELS (table of the distinct frames of this call)
  const int32_t* ref_idx;       // n_tracks: row of frame_levels holding the template
  const int32_t* cur_idx;       // n_tracks: row of frame_levels of the current frame
  svoh_klt_options opt;
  int n_tracks;
  int n_frames;                 // rows of frame_levels (device-resident indices are range-checked in the kernel)
  const int32_t* px_ref;        // 2 x n
  double* px_cur;               // 2 x n, in/out
  uint8_t* status;              // n
  unsigned int* unit_counts;    // 4 per track: iterations 16x16, 8x8, templates 16x16, 8x8
};

__device__ __forceinline__ int wave_sum_i32(int v) { return svoh::wave_sum_i32_dpp(v); }

// One level of alignPyr2D for patch size P (16 or 8).  Returns: 0 = continue to the
// next level, 1 = return false (not converged / NaN).  `converged` and px_cur are updated.
template <int P>
__device__ __forceinline__ int klt_level(const DevImage& img_ref, const DevImage& img_cur, int level, int px_ref0_x,
                                         int px_ref0_y, int n_iter, float min_update_squared, double& pcx, double& pcy,
                                         bool& converged, int lane, int& n_iters, int& n_tmpl)
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
  ++n_tmpl;
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
    ++n_iters;
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

// blockDim.x / 64 tracks per workgroup (one wave each); waves never synchronise
__global__ __launch_bounds__(256) void klt_track_kernel(const KltArgs a)
{
  const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (t >= a.n_tracks) return;
  const int lane = threadIdx.x & 63;
  const int ri = a.ref_idx[t], ci = a.cur_idx[t];
  if ((unsigned)ri >= (unsigned)a.n_frames || (unsigned)ci >= (unsigned)a.n_frames) {
    if (lane == 0) {
      a.status[t] = 0;
      reinterpret_cast<uint4*>(a.unit_counts)[t] = make_uint4(0u, 0u, 0u, 0u);
    }
    return;
  }
  const DevImage* ref = a.frame_levels + (size_t)ri * SVOH_MAX_LEVELS;
  const DevImage* curl = a.frame_levels + (size_t)ci * SVOH_MAX_LEVELS;
  double pcx = a.px_cur[2 * t], pcy = a.px_cur[2 * t + 1];
  const int rx = a.px_ref[2 * t], ry = a.px_ref[2 * t + 1];
  bool converged = false;
  bool failed = false;
  int it16 = 0, it8 = 0, t16 = 0, t8 = 0;
  for (int level = a.opt.max_level; level >= a.opt.min_level; --level) {
    const int P = a.opt.patch_sizes[level];
    int rc;
    if (P == 16)
      rc = klt_level<16>(ref[level], curl[level], level, rx, ry, a.opt.max_iter, a.opt.min_update_squared, pcx,
                         pcy, converged, lane, it16, t16);
    else if (P == 8)
      rc = klt_level<8>(ref[level], curl[level], level, rx, ry, a.opt.max_iter, a.opt.min_update_squared, pcx,
                        pcy, converged, lane, it8, t8);
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
    reinterpret_cast<uint4*>(a.unit_counts)[t] = make_uint4((unsigned)it16, (unsigned)it8, (unsigned)t16, (unsigned)t8);
  }
}

}  // namespace svoh

using namespace svoh;

static int check_klt_options(svoh_ctx* ctx, const svoh_klt_options* options)
{
  SVOH_REQUIRE(ctx, options->max_level >= options->min_level && options->min_level >= 0 &&
                        options->max_level < SVOH_MAX_LEVELS && options->max_iter >= 1,
               "bad KLT level range / max_iter");
  for (int l = options->min_level; l <= options->max_level; ++l)
    if (options->patch_sizes[l] != 8 && options->patch_sizes[l] != 16)
      return set_error(ctx, SVOH_ERR_UNSUPPORTED, "KLT patch size %d at level %d not built (8 and 16 are)",
                       options->patch_sizes[l], l);
  return SVOH_OK;
}

// Stage what is host-resident, launch, fetch what the caller wants on the host.
// idx = [ref_idx (n) | cur_idx (n)] when host-resident; with SVOH_MEM_DEVICE the five
// per-track arrays are used where they are and nothing is copied back.
static int launch_klt(svoh_ctx* ctx, const svoh_klt_options* options, const std::vector<const Frame*>& frames,
                      int n_tracks, int mem_space, const int32_t* ref_idx, const int32_t* cur_idx,
                      const int32_t* px_ref, double* px_cur, uint8_t* status)
{
  const bool on_device = mem_space == SVOH_MEM_DEVICE;
  const size_t n = (size_t)n_tracks;
  const size_t lv_bytes = (sizeof(DevImage) * SVOH_MAX_LEVELS * frames.size() + 63) & ~(size_t)63;
  const size_t idx_bytes = on_device ? 0 : ((sizeof(int32_t) * 2 * n + 63) & ~(size_t)63);
  const size_t pxr_bytes = on_device ? 0 : ((sizeof(int32_t) * 2 * n + 63) & ~(size_t)63);
  const size_t in_bytes = lv_bytes + idx_bytes + pxr_bytes;
  const size_t io_bytes = on_device ? 0 : (sizeof(double) * 2 * n + n);
  SVOH_HIP_TRY(ctx, ctx->h_scratch0.reserve(in_bytes + io_bytes));
  SVOH_HIP_TRY(ctx, ctx->d_scratch0.reserve(in_bytes + io_bytes));
  uint8_t* h = static_cast<uint8_t*>(ctx->h_scratch0.ptr);
  uint8_t* d = static_cast<uint8_t*>(ctx->d_scratch0.ptr);
  DevImage* hl = reinterpret_cast<DevImage*>(h);
  for (size_t k = 0; k < frames.size(); ++k)
    for (int l = 0; l < SVOH_MAX_LEVELS; ++l)
      hl[k * SVOH_MAX_LEVELS + l] = l < frames[k]->n_levels ? frames[k]->lv[l] : DevImage{ nullptr, 0, 0, 0, 0 };
  if (!on_device) {
    memcpy(h + lv_bytes, ref_idx, sizeof(int32_t) * n);
    memcpy(h + lv_bytes + sizeof(int32_t) * n, cur_idx, sizeof(int32_t) * n);
    memcpy(h + lv_bytes + idx_bytes, px_ref, sizeof(int32_t) * 2 * n);
    memcpy(h + in_bytes, px_cur, sizeof(double) * 2 * n);
  }
  SVOH_HIP_TRY(ctx, hipMemcpyAsync(d, h, in_bytes + (on_device ? 0 : sizeof(double) * 2 * n), hipMemcpyHostToDevice,
                                   ctx->stream));
  KltArgs args;
  args.frame_levels = reinterpret_cast<const DevImage*>(d);
  args.opt = *options;
  args.n_tracks = n_tracks;
  args.n_frames = (int)frames.size();
  if (on_device) {
    args.ref_idx = ref_idx; args.cur_idx = cur_idx; args.px_ref = px_ref; args.px_cur = px_cur; args.status = status;
  } else {
    args.ref_idx = reinterpret_cast<const int32_t*>(d + lv_bytes);
    args.cur_idx = args.ref_idx + n_tracks;
    args.px_ref = reinterpret_cast<const int32_t*>(d + lv_bytes + idx_bytes);
    args.px_cur = reinterpret_cast<double*>(d + in_bytes);
    args.status = d + in_bytes + sizeof(double) * 2 * n;
  }
  {
    unsigned long long* dummy;
    int rc = reset_counters(ctx, &dummy);
    if (rc == SVOH_OK) rc = reserve_unit_counts(ctx, n, &args.unit_counts);
    if (rc != SVOH_OK) return rc;
  }
  SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_start, ctx->stream));
  {
    const char* e = getenv("SVOH_KLT_BLOCK");
    int block = e ? atoi(e) : 256;
    if (block != 64 && block != 128 && block != 256) block = 256;
    const int tpb = block / 64;
    hipLaunchKernelGGL(klt_track_kernel, dim3((n_tracks + tpb - 1) / tpb), dim3(block), 0, ctx->stream, args);
  }
  SVOH_HIP_TRY(ctx, hipGetLastError());
  SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_stop, ctx->stream));
  ctx->misc_timed = true;
  {
    int rc = reduce_unit_counts(ctx, n);
    if (rc != SVOH_OK) return rc;
  }
  if (on_device) return SVOH_OK;  // stream-ordered; the caller synchronises when it needs the results
  SVOH_HIP_TRY(ctx, hipMemcpyAsync(h + in_bytes, d + in_bytes, io_bytes, hipMemcpyDeviceToHost, ctx->stream));
  SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(px_cur, h + in_bytes, sizeof(double) * 2 * n);
  memcpy(status, h + in_bytes + sizeof(double) * 2 * n, n);
  return SVOH_OK;
}

extern "C" int svoh_klt_track_multi(svoh_ctx* ctx, const svoh_klt_options* options, int n_tracks,
                                    const svoh_frame_t* ref_frames, const svoh_frame_t* cur_frames,
                                    const int32_t* px_ref, double* px_cur, uint8_t* status)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, options && n_tracks >= 0, "bad arguments");
  if (n_tracks == 0) return SVOH_OK;
  SVOH_REQUIRE(ctx, ref_frames && cur_frames && px_ref && px_cur && status, "NULL argument");
  {
    const int rc = check_klt_options(ctx, options);
    if (rc != SVOH_OK) return rc;
  }
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));

  // table of the distinct frames + two indices per track
  std::unordered_map<svoh_frame_t, int32_t> table;
  std::vector<const Frame*> frames;
  std::vector<int32_t> idx(2 * (size_t)n_tracks);
  auto lookup = [&](svoh_frame_t hnd) -> int32_t {
    auto it = table.find(hnd);
    if (it != table.end()) return it->second;
    const Frame* f = find_frame(ctx, hnd);
    if (!f) return -1;
    const int32_t k = (int32_t)frames.size();
    frames.push_back(f);
    table.emplace(hnd, k);
    return k;
  };
  svoh_frame_t last_r = 0, last_c = 0;
  int32_t ir = -1, ic = -1;
  for (int i = 0; i < n_tracks; ++i) {
    if (ir < 0 || ref_frames[i] != last_r) { ir = lookup(ref_frames[i]); last_r = ref_frames[i]; }
    if (ic < 0 || cur_frames[i] != last_c) { ic = lookup(cur_frames[i]); last_c = cur_frames[i]; }
    if (ir < 0 || ic < 0) return set_error(ctx, SVOH_ERR_BAD_HANDLE, "track %d: unknown frame handle", i);
    const Frame* fr = frames[ir];
    const Frame* fc = frames[ic];
    if (fr->n_levels <= options->max_level || fc->n_levels <= options->max_level)
      return set_error(ctx, SVOH_ERR_INVALID_ARGUMENT, "track %d: pyramid has too few levels", i);
    for (int l = options->min_level; l <= options->max_level; ++l)
      if (fr->lv[l].w != fc->lv[l].w || fr->lv[l].h != fc->lv[l].h)
        return set_error(ctx, SVOH_ERR_INVALID_ARGUMENT, "track %d: reference and current level %d differ in size", i, l);
    idx[i] = ir;
    idx[(size_t)n_tracks + i] = ic;
  }
  return launch_klt(ctx, options, frames, n_tracks, SVOH_MEM_HOST, idx.data(), idx.data() + n_tracks, px_ref, px_cur,
                    status);
} SVOH_ABI_CATCH(ctx)

extern "C" int svoh_klt_track_batch(svoh_ctx* ctx, const svoh_klt_options* options, int n_tracks,
                                    const svoh_frame_t* ref_frames, svoh_frame_t cur_frame, const int32_t* px_ref,
                                    double* px_cur, uint8_t* status)
try {
  if (n_tracks <= 0) return n_tracks == 0 ? SVOH_OK : set_error(ctx, SVOH_ERR_INVALID_ARGUMENT, "negative n_tracks");
  std::vector<svoh_frame_t> cur((size_t)n_tracks, cur_frame);
  return svoh_klt_track_multi(ctx, options, n_tracks, ref_frames, cur.data(), px_ref, px_cur, status);
} SVOH_ABI_CATCH(ctx)

extern "C" int svoh_klt_track_indexed(svoh_ctx* ctx, const svoh_klt_options* options, int n_frames,
                                      const svoh_frame_t* frames, int n_tracks, const int32_t* ref_frame_idx,
                                      const int32_t* cur_frame_idx, const int32_t* px_ref, double* px_cur,
                                      uint8_t* status, int mem_space)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, options && n_tracks >= 0 && n_frames >= 1 && frames, "bad arguments");
  SVOH_REQUIRE(ctx, mem_space == SVOH_MEM_HOST || mem_space == SVOH_MEM_DEVICE, "bad mem_space");
  if (n_tracks == 0) return SVOH_OK;
  SVOH_REQUIRE(ctx, ref_frame_idx && cur_frame_idx && px_ref && px_cur && status, "NULL argument");
  {
    const int rc = check_klt_options(ctx, options);
    if (rc != SVOH_OK) return rc;
  }
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  // every frame of the table must be usable as reference and as current frame of any
  // track: enough levels, and one common size per level (alignPyr2D assumes it)
  std::vector<const Frame*> tab((size_t)n_frames);
  for (int k = 0; k < n_frames; ++k) {
    tab[k] = find_frame(ctx, frames[k]);
    if (!tab[k]) return set_error(ctx, SVOH_ERR_BAD_HANDLE, "frames[%d]: unknown frame handle", k);
    if (tab[k]->n_levels <= options->max_level)
      return set_error(ctx, SVOH_ERR_INVALID_ARGUMENT, "frames[%d]: pyramid has too few levels", k);
    for (int l = options->min_level; l <= options->max_level; ++l)
      if (tab[k]->lv[l].w != tab[0]->lv[l].w || tab[k]->lv[l].h != tab[0]->lv[l].h)
        return set_error(ctx, SVOH_ERR_INVALID_ARGUMENT, "frames[%d]: level %d differs in size from frames[0]", k, l);
  }
  if (mem_space == SVOH_MEM_HOST)
    for (int i = 0; i < n_tracks; ++i)
      if (ref_frame_idx[i] < 0 || ref_frame_idx[i] >= n_frames || cur_frame_idx[i] < 0 || cur_frame_idx[i] >= n_frames)
        return set_error(ctx, SVOH_ERR_INVALID_ARGUMENT, "track %d: frame index out of range", i);
  return launch_klt(ctx, options, tab, n_tracks, mem_space, ref_frame_idx, cur_frame_idx, px_ref, px_cur, status);
} SVOH_ABI_CATCH(ctx)
