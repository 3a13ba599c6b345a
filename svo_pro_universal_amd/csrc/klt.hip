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

// sum over the LANES lanes that share a track: the whole wave (6 DPP adds + v_readlane), or one DPP row of 16 lanes
// (4 DPP adds, no lane of another row is read: safe when other rows are switched off)
template <int LANES>
__device__ __forceinline__ int group_sum_i32(int v)
{
  if constexpr (LANES == 64) return svoh::wave_sum_i32_dpp(v);
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);  // row_mirror
  return v;
}

// ---- a track's windows in LDS (round 4) ---------------------------------------------------------------------------
// Round 3's counters (VALU busy 58 %, waves waiting 55 %, SQ_INSTS_LDS = 0) read as latency, but the arithmetic of the
// vector L1 says otherwise: every lane of a track read ITS OWN patch row, so a wave-wide load instruction touched 64
// different cache lines, and at ~1 200 line look-ups per track (96 per 16x16 iteration, 144 per template) x 400 waves per
// compute unit the L1's one line per cycle was the kernel's time (490 K of its 525 K cycles).  Now a track's search window
// of the current level -- the patch plus kKltMargin pixels on every side, the motion inside a level being a few pixels at
// most -- and its template's 18 x 18 (10 x 10) reference footprint are staged ONCE per level into LDS by loads in which
// the four lanes of a quad read the four 8-byte pieces of one row (one line look-up per row instead of one per lane and
// instruction: 48 look-ups per track and 16x16 level instead of ~330), and every iteration reads its rows from LDS.  A
// patch that leaves its window (more than kKltMargin pixels from where the level began) has the window staged again
// around it.  Same integers everywhere: positions and status stay bit-identical (tests/test_klt_matcher_gpu.py).
constexpr int kKltMargin = 4;
constexpr int kKltLdsPitch = 40;                   // bytes per staged row: 10 dwords -- 16 consecutive rows start in 16 different banks
constexpr int kKltLdsRows = 28;                    // 16 + 2 * margin + 1 rows, rounded up to the 4 rows of a staging step
constexpr int kKltLdsPerTrack = kKltLdsRows * kKltLdsPitch;
typedef __attribute__((address_space(3))) unsigned char* klt_lds_ptr;

// NDW dwords from a byte address in LDS that need not be aligned (gfx950 reads misaligned LDS words in hardware)
template <int NDW>
__device__ __forceinline__ void klt_lds_read(klt_lds_ptr p, unsigned (&out)[NDW])
{
  typedef unsigned u32_unaligned __attribute__((aligned(1)));
#pragma unroll
  for (int i = 0; i < NDW; ++i) out[i] = *reinterpret_cast<const __attribute__((address_space(3))) u32_unaligned*>(p + 4 * i);
}

// rows wy .. wy + n_rows - 1, 32 bytes from column wx on, of `img` into the track's LDS area (lane16: lane within the
// track's 16).  Row indices are clamped into the image (a clamped row is never read back); bytes beyond a row's end
// are the next row's or the slab's tail padding (never used either: kSlabTailPad >= 32).
template <int N_ROWS>
__device__ __forceinline__ void klt_stage_window(const DevImage& img, int wx, int wy, int lane16, klt_lds_ptr lds)
{
  const int piece = lane16 & 3, r0 = lane16 >> 2;
#pragma unroll
  for (int k = 0; k < (N_ROWS + 3) / 4; ++k) {
    const int row = r0 + 4 * k;
    int yy = wy + row;
    yy = yy < 0 ? 0 : (yy >= img.h ? img.h - 1 : yy);
    unsigned long long v;
    __builtin_memcpy(&v, img.data + (ptrdiff_t)yy * img.pitch + wx + 8 * piece, 8);
    if (N_ROWS % 4 == 0 || row < N_ROWS)
      *reinterpret_cast<__attribute__((address_space(3))) unsigned long long*>(lds + row * kKltLdsPitch + 8 * piece) = v;
  }
}

// One level of alignPyr2D for patch size P (16 or 8), LANES lanes per track: 64 (the wave owns one track) or 16 (four
// tracks side by side, one per DPP row; every value below is then uniform per row, not per wave, and a row whose
// track has left the loop idles while the others finish).  Returns per lane: 0 = continue to the next level,
// 1 = return false (not converged / NaN).  `converged` and px_cur are updated.  `run` = this lane's track takes part.
template <int P, int LANES>
__device__ __forceinline__ int klt_level(const DevImage& img_ref, const DevImage& img_cur, int level, int px_ref0_x,
                                         int px_ref0_y, int n_iter, float min_update_squared, double& pcx, double& pcy,
                                         bool& converged, int lane, int& n_iters, int& n_tmpl, bool run, klt_lds_ptr lds)
{
  static_assert(LANES == 16, "the LDS windows are staged by the 16 lanes of a track");
  constexpr int PPL = P * P / LANES;  // pixels per lane: 16 (16x16 patch on a row of lanes: one patch row each) or 4 (8x8)
  static_assert(PPL == 4 || PPL == 16, "a lane holds 4 or 16 consecutive pixels of one patch row");
  typedef short short2v __attribute__((ext_vector_type(2)));
  const int halfpatch_size = P / 2;
  const int scale = 1 << level;
  const int width = img_ref.w, height = img_ref.h;
  const float prfx = (float)px_ref0_x / (float)scale - (float)halfpatch_size;
  const float prfy = (float)px_ref0_y / (float)scale - (float)halfpatch_size;
  const int prx = (int)prfx, pry = (int)prfy;
  const float offx = prfx - (float)prx, offy = prfy - (float)pry;
  bool active = run && !(prx < 1 || pry < 1 || prx >= width - P - 1 || pry >= height - P - 1);  // else: too close to the border
  // this lane's pixels: row y, columns x0 .. x0+PPL-1
  const int y = (lane * PPL) / P;
  const int x0 = (lane * PPL) % P;
  // The template and its raw differences stay in registers for the level, packed: four u8 template pixels per dword,
  // two int16 differences per dword (|d| <= 255) -- 20 registers instead of 48 for a 16-pixel row, which is what lets
  // five waves per SIMD share the latency of the row loads.  Every sum below is an exact integer (v_dot2_i32_i16).
  unsigned tq[PPL / 4];
  short2v gx2[PPL / 2], gy2[PPL / 2];
  int h00 = 0, h01 = 0, h11 = 0;
  if (active) {
    // rows y-1, y, y+1 from column x0-1 on: PPL+2 pixels each, as unaligned 8-byte loads (the spare bytes stay inside
    // the row, the next row or the slab's tail padding)
    constexpr int NW = PPL == 16 ? 6 : 2;
    unsigned U[NW], M[NW], D[NW];
    // the footprint rows pry - 1 .. pry + P from column prx - 1 on, through LDS (window row r = image row pry - 1 + r)
    klt_stage_window<P + 2>(img_ref, prx - 1, pry - 1, lane, lds);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      const klt_lds_ptr it = lds + y * kKltLdsPitch + x0;   // row y of the window = image row pry + y - 1
      klt_lds_read<NW>(it, U);
      klt_lds_read<NW>(it + kKltLdsPitch, M);
      klt_lds_read<NW>(it + 2 * kKltLdsPitch, D);
    }
    __builtin_amdgcn_wave_barrier();   // the area is written again (the current window) only after every lane has read
    auto byte_of = [](const unsigned (&w)[NW], int j) { return (int)((w[j >> 2] >> (8 * (j & 3))) & 255u); };
#pragma unroll
    for (int i = 0; i < PPL / 4; ++i) tq[i] = __builtin_amdgcn_alignbyte(M[i + 1], M[i], 1);   // pixels 4i .. 4i+3 = bytes 4i+1 .. 4i+4
#pragma unroll
    for (int k = 0; k < PPL; k += 2) {
      short2v gx, gy;
      gx.x = (short)(byte_of(M, k + 2) - byte_of(M, k));      gx.y = (short)(byte_of(M, k + 3) - byte_of(M, k + 1));
      gy.x = (short)(byte_of(D, k + 1) - byte_of(U, k + 1));  gy.y = (short)(byte_of(D, k + 2) - byte_of(U, k + 2));
      gx2[k / 2] = gx; gy2[k / 2] = gy;
      h00 = __builtin_amdgcn_sdot2(gx, gx, h00, false);
      h01 = __builtin_amdgcn_sdot2(gx, gy, h01, false);
      h11 = __builtin_amdgcn_sdot2(gy, gy, h11, false);
    }
    ++n_tmpl;
  } else {
#pragma unroll
    for (int i = 0; i < PPL / 4; ++i) tq[i] = 0u;
#pragma unroll
    for (int k = 0; k < PPL / 2; ++k) { gx2[k] = short2v{ 0, 0 }; gy2[k] = short2v{ 0, 0 }; }
  }
  // residuals of two neighbouring pixels at once: (cur0, cur1) - (tmpl0, tmpl1) as a packed int16 subtraction, then one
  // v_dot2_i32_i16 per Jacobian row
  auto accumulate_pair = [&](int k, int cur0, int cur1, int& j0, int& j1) {
    const unsigned c2 = (unsigned)cur0 | ((unsigned)cur1 << 16);
    const unsigned t2 = __builtin_amdgcn_perm(0u, tq[k >> 2], (k & 2) ? 0x0c030c02u : 0x0c010c00u);
    const short2v r2 = __builtin_bit_cast(short2v, c2) - __builtin_bit_cast(short2v, t2);
    j0 = __builtin_amdgcn_sdot2(r2, gx2[k >> 1], j0, false);
    j1 = __builtin_amdgcn_sdot2(r2, gy2[k >> 1], j1, false);
  };
  int rc = 0;
  float u = 0.f, v = 0.f, Hi00 = 0.f, Hi01 = 0.f, Hi10 = 0.f, Hi11 = 0.f;
  bool go_to_next_level = false;
  const bool entered = active;
  if (active) {
    const float H00 = (float)group_sum_i32<LANES>(h00), H01 = (float)group_sum_i32<LANES>(h01), H11 = (float)group_sum_i32<LANES>(h11);
    const float H10 = H01;
    // Eigen Matrix2f::inverse()
    const float det = H00 * H11 - H10 * H01;
    const float invdet = 1.0f / det;
    Hi00 = H11 * invdet; Hi10 = -H10 * invdet; Hi01 = -H01 * invdet; Hi11 = H00 * invdet;
    u = (float)(pcx / scale - halfpatch_size - offx);
    v = (float)(pcy / scale - halfpatch_size - offy);
    converged = false;
  }
  const int cur_step = img_ref.pitch;  // the reference indexes the current image with the reference's step
  DevImage img_win = img_cur;          // ... so the window's rows are taken with that step too
  img_win.pitch = cur_step;
  int wx = 0, wy = 0;
  bool have_window = false;
  for (int iter = 0; iter < n_iter; ++iter) {
    if (__ballot(active) == 0) break;
    if (active) {
      if (u != u || v != v) { rc = 1; active = false; }
    }
    if (active) {
      go_to_next_level = false;
      const int u_r = (int)floorf(u);
      const int v_r = (int)floorf(v);
      if (u_r < 0 || v_r < 0 || u_r >= width - P || v_r >= height - P) {
        go_to_next_level = true;
        active = false;
      }
      if (active) {
        ++n_iters;
        // the patch's 17 (9) rows and columns from (u_r, v_r) on must lie inside the staged window
        if (!have_window || u_r < wx || v_r < wy || u_r - wx > 2 * kKltMargin || v_r - wy > 2 * kKltMargin) {
          wx = u_r > kKltMargin ? u_r - kKltMargin : 0;
          wy = v_r > kKltMargin ? v_r - kKltMargin : 0;
          __builtin_amdgcn_wave_barrier();
          klt_stage_window<P + 2 * kKltMargin + 1>(img_win, wx, wy, lane, lds);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          have_window = true;
        }
        const klt_lds_ptr wit = lds + (v_r - wy + y) * kKltLdsPitch + (u_r - wx) + x0;
        const float subpix_x = u - u_r;
        const float subpix_y = v - v_r;
        const int wTL = (int)(unsigned short)((1.0f - subpix_x) * (1.0f - subpix_y) * 128);
        const int wTR = (int)(unsigned short)(subpix_x * (1.0f - subpix_y) * 128);
        const int wBL = (int)(unsigned short)((1.0f - subpix_x) * subpix_y * 128);
        const int wBR = (int)(unsigned short)(128 - wTL - wTR - wBL);
        int j0 = 0, j1 = 0;
        if constexpr (PPL == 16) {
          // a whole 16-pixel row per lane: 2 x 17 pixels as three unaligned 8-byte loads each (the spare bytes stay
          // inside the row, the next row or the slab's tail padding), taps packed and summed as in the 4-pixel case
          unsigned T[6], B[6];
          klt_lds_read<6>(wit, T);
          klt_lds_read<6>(wit + kKltLdsPitch, B);
          const unsigned W = (unsigned)wTL | ((unsigned)wTR << 8) | ((unsigned)wBL << 16) | ((unsigned)wBR << 24);
#pragma unroll
          for (int dq = 0; dq < 4; ++dq) {
            const unsigned T3 = __builtin_amdgcn_alignbyte(T[dq + 1], T[dq], 3), B3 = __builtin_amdgcn_alignbyte(B[dq + 1], B[dq], 3);
            const unsigned q[4] = { __builtin_amdgcn_perm(B[dq], T[dq], 0x05040100u), __builtin_amdgcn_perm(B[dq], T[dq], 0x06050201u),
                                    __builtin_amdgcn_perm(B[dq], T[dq], 0x07060302u), __builtin_amdgcn_perm(B3, T3, 0x05040100u) };
            int cur[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) cur[r] = (int)(unsigned short)(__builtin_amdgcn_udot4(q[r], W, 64u, false) >> 7);
            accumulate_pair(4 * dq, cur[0], cur[1], j0, j1);
            accumulate_pair(4 * dq + 2, cur[2], cur[3], j0, j1);
          }
        } else {
          // the lane's 2 x 5 pixels as two unaligned 8-byte loads (the spare bytes stay inside the row, the next row or
          // the slab's tail padding); per pixel the four taps are gathered into one dword (v_perm_b32) and the 7-bit
          // fixed-point bilinear sum is ONE v_dot4_u32_u8 against the packed weights (all <= 128): the same integers
          uint2 T, B;
          {
            unsigned t2[2], b2[2];
            klt_lds_read<2>(wit, t2);
            klt_lds_read<2>(wit + kKltLdsPitch, b2);
            T = make_uint2(t2[0], t2[1]); B = make_uint2(b2[0], b2[1]);
          }
          const unsigned W = (unsigned)wTL | ((unsigned)wTR << 8) | ((unsigned)wBL << 16) | ((unsigned)wBR << 24);
          const unsigned T3 = __builtin_amdgcn_alignbyte(T.y, T.x, 3), B3 = __builtin_amdgcn_alignbyte(B.y, B.x, 3);
          const unsigned q0 = __builtin_amdgcn_perm(B.x, T.x, 0x05040100u), q1 = __builtin_amdgcn_perm(B.x, T.x, 0x06050201u);
          const unsigned q2 = __builtin_amdgcn_perm(B.x, T.x, 0x07060302u), q3 = __builtin_amdgcn_perm(B3, T3, 0x05040100u);
          const unsigned q[4] = { q0, q1, q2, q3 };
          int cur[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) cur[k] = (int)(unsigned short)(__builtin_amdgcn_udot4(q[k], W, 64u, false) >> 7);
          accumulate_pair(0, cur[0], cur[1], j0, j1);
          accumulate_pair(2, cur[2], cur[3], j0, j1);
        }
        const float Jres0 = -(float)group_sum_i32<LANES>(j0);
        const float Jres1 = -(float)group_sum_i32<LANES>(j1);
        const float up0 = (Hi00 * Jres0 + Hi01 * Jres1) * 2.0f;
        const float up1 = (Hi10 * Jres0 + Hi11 * Jres1) * 2.0f;
        u += up0;
        v += up1;
        if (up0 * up0 + up1 * up1 < min_update_squared) {
          converged = true;
          active = false;
        }
      }
    }
  }
  if (entered && rc == 0) {
    pcx = (double)((u + halfpatch_size + offx) * scale);
    pcy = (double)((v + halfpatch_size + offy) * scale);
    if (!converged && !go_to_next_level) rc = 1;
  }
  return rc;
}

__device__ __forceinline__ double readlane_f64(double x, int src_lane)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(x), src_lane);
  return __hiloint2double(hi, lo);
}

// A wave owns FOUR tracks, one per DPP row (lanes 16r .. 16r+15 hold track 4*wave + r) through all levels and
// iterations: a 16x16 patch gives a lane one row of 16 pixels, an 8x8 patch 4 pixels.  On a wave of its own a track
// spent most of an iteration's instructions on bookkeeping that is the same for 1, 4 or 16 pixels per lane (weights,
// two reductions, the 2x2 update); side by side the four tracks share it, reductions stay inside a row (4 DPP adds),
// and a row whose track has left the loop idles while the others finish.  Waves never synchronise.
__global__ __launch_bounds__(256) void klt_track_kernel(const KltArgs a)
{
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int row = lane >> 4, lane16 = lane & 15;
  const int t = wave * 4 + row;                      // this lane's track
  extern __shared__ __align__(16) unsigned char s_win[];   // one window area per track of the workgroup (4 per wave)
  const klt_lds_ptr lds = (klt_lds_ptr)s_win + ((threadIdx.x >> 6) * 4 + row) * kKltLdsPerTrack;
  const bool exists = t < a.n_tracks;
  int ri = 0, ci = 0;
  if (exists) { ri = a.ref_idx[t]; ci = a.cur_idx[t]; }
  const bool ok_idx = exists && (unsigned)ri < (unsigned)a.n_frames && (unsigned)ci < (unsigned)a.n_frames;
  double pcx = 0.0, pcy = 0.0;
  int rx = 0, ry = 0;
  if (ok_idx) { pcx = a.px_cur[2 * t]; pcy = a.px_cur[2 * t + 1]; rx = a.px_ref[2 * t]; ry = a.px_ref[2 * t + 1]; }
  bool converged = false, failed = false;
  int it16 = 0, it8 = 0, t16 = 0, t8 = 0;
  for (int level = a.opt.max_level; level >= a.opt.min_level; --level) {
    const int P = a.opt.patch_sizes[level];
    if (P == 16) {
      const bool run = ok_idx && !failed;
      if (__ballot(run) != 0) {
        const DevImage ref = a.frame_levels[(size_t)(run ? ri : 0) * SVOH_MAX_LEVELS + level];
        const DevImage cur = a.frame_levels[(size_t)(run ? ci : 0) * SVOH_MAX_LEVELS + level];
        const int rc = klt_level<16, 16>(ref, cur, level, rx, ry, a.opt.max_iter, a.opt.min_update_squared, pcx, pcy, converged,
                                         lane16, it16, t16, run, lds);
        if (run && rc) failed = true;
      }
    } else if (P == 8) {
      const bool run = ok_idx && !failed;
      if (__ballot(run) != 0) {
        const DevImage ref = a.frame_levels[(size_t)(run ? ri : 0) * SVOH_MAX_LEVELS + level];
        const DevImage cur = a.frame_levels[(size_t)(run ? ci : 0) * SVOH_MAX_LEVELS + level];
        const int rc = klt_level<8, 16>(ref, cur, level, rx, ry, a.opt.max_iter, a.opt.min_update_squared, pcx, pcy, converged,
                                        lane16, it8, t8, run, lds);
        if (run && rc) failed = true;
      }
    } else {
      if (ok_idx) failed = true;   // rejected on the host
    }
    if (__ballot(ok_idx && !failed) == 0) break;
  }
  if (exists && lane16 == 0) {
    if (ok_idx) {
      a.px_cur[2 * t] = pcx;
      a.px_cur[2 * t + 1] = pcy;
      a.status[t] = (!failed && converged) ? 1 : 0;
      reinterpret_cast<uint4*>(a.unit_counts)[t] = make_uint4((unsigned)it16, (unsigned)it8, (unsigned)t16, (unsigned)t8);
    } else {
      a.status[t] = 0;
      reinterpret_cast<uint4*>(a.unit_counts)[t] = make_uint4(0u, 0u, 0u, 0u);
    }
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
  if (ctx->timing_on()) SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_start, ctx->stream));
  {
    int block = SvohKnobs::or_default(ctx->knobs.klt_block, 256);
    if (block != 64 && block != 128 && block != 256) block = 256;
    const int tpb = block / 64 * 4;   // four tracks per wave
    hipLaunchKernelGGL(klt_track_kernel, dim3((n_tracks + tpb - 1) / tpb), dim3(block), (size_t)tpb * kKltLdsPerTrack, ctx->stream, args);
  }
  SVOH_HIP_TRY(ctx, hipGetLastError());
  if (ctx->timing_on()) SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_stop, ctx->stream));
  ctx->misc_timed = ctx->timing_on(); ctx->misc_launched = true;
  {
    int rc = reduce_unit_counts(ctx, n);
    if (rc != SVOH_OK) return rc;
  }
  if (on_device) return SVOH_OK;  // stream-ordered; the caller synchronises when it needs the results
  SVOH_HIP_TRY(ctx, svoh_copy_to_host(ctx, h + in_bytes, d + in_bytes, io_bytes));
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
