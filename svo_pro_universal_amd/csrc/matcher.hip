// matcher.hip -- batched Matcher (direct + epipolar) and depth-filter seed update
// for gfx950 (a-10 ... a-14).
//
// Replaces, for a batch of independent features / seeds:
//   warp::getWarpMatrixAffine / getBestSearchLevel / warpAffine   src/svo_direct/src/patch_warp.cpp:20-60, 97-156
//   patch_utils::createPatchFromPatchWithBorder                   src/svo_direct/include/svo/direct/patch_utils.h:18-30
//   patch_score::ZMSSD<4>                                         src/svo_direct/include/svo/direct/patch_score.h:44-285
//   feature_alignment::align1D / align2D                          src/svo_direct/src/feature_alignment.cpp:31-391
//   Matcher::findMatchDirect / findEpipolarMatchDirect / scanEpipolar* / findLocalMatch
//                                                                 src/svo_direct/src/matcher.cpp:31-505
//   depth_filter_utils::updateSeed / updateFilterVogiatzis / updateFilterGaussian / computeTau
//                                                                 src/svo_direct/src/depth_filter.cpp:367-596
//   DepthFilter::updateSeeds (synchronous branch)                 src/svo_direct/src/depth_filter.cpp:200-233
//
// One thread owns one feature / seed from warp to filter update: units are
// independent (that is the data parallelism of this path), every unit is a short,
// branchy, mostly-integer program, and the reference's float accumulators (Jres in
// align1D/2D) are order dependent, so a sequential per-unit evaluation in the
// reference's own order is what keeps results identical.  The 10x10 warped patch
// of every thread lives in LDS (100 B per thread, stride 25 dwords: conflict-free);
// gradients are recomputed from it instead of being stored.  Integer results
// (warped patch bytes, ZMSSD scores, visited pixels, result codes) are exact; this
// file is compiled with -ffp-contract=off so the float/double expressions round
// like the reference's (non fast-math) evaluation order.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "svoh_internal.h"
#include "svoh_math.h"

namespace svoh {

struct DevFrameView {
  DevImage lv[SVOH_MAX_LEVELS];
  CamModel cam;
  Rigid T_f_w;
  double seed_mu_range;
  int32_t n_levels;
  int32_t id;
  int32_t pose_result_index_plus1;   // > 0: T_f_w holds T_cam_imu until matcher_prologue_kernel has run (svoh_frame_view)
  int32_t feat_n;                    // the frame's resident columns (svoh_frame_view::features), or 0 / NULL
  const double* feat_px; const double* feat_f; const double* feat_grad; const int32_t* feat_level;
  // ... the same columns in tile order, perm[q] = the feature at place q (svoh::FeatureSet)
  const double* feat_spx; const double* feat_sf; const double* feat_sgrad; const int32_t* feat_slevel; const int32_t* feat_perm;
  // SVOH_BATCH_WHOLE_SETS: the batch's units [unit_begin, unit_begin + feat_n) are this reference frame's features 0 .. feat_n - 1
  int32_t unit_begin;
};

// SVOH_BATCH_WHOLE_SETS: the reference frame whose units hold unit / slot p = the last one that begins at or before p
__device__ __forceinline__ int whole_sets_frame_of(const DevFrameView* ref_frames, int n_ref, int p, int n = 0)
{
  // keyframes hold about the same number of seeds: the frame that would hold p if they all held n / n_ref is usually the one
  // (one load instead of a chain of log2(n_ref) dependent ones at the head of every wave); else the search
  if (n > 0) {
    const int g = (int)(((long long)p * n_ref) / n);
    if (g >= 0 && g < n_ref) { const int b = ref_frames[g].unit_begin; if (p >= b && p < b + ref_frames[g].feat_n) return g; }
  }
  int lo = 0, hi = n_ref;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ref_frames[mid].unit_begin <= p) lo = mid; else hi = mid; }
  return lo;
}

struct MatcherArgs {
  const DevFrameView* ref_frames;
  const DevFrameView* cur_frame;
  svoh_matcher_options mopt;
  svoh_depth_filter_options dopt;
  int n;
  int n_ref_frames, n_cur_frames;  // device-resident indices are range-checked in the kernels
  const int32_t* ref_frame_idx;
  const int32_t* cur_frame_idx;  // may be NULL
  const double* px;
  const double* f;
  const double* grad;
  const int32_t* level;
  uint8_t* type;
  // direct
  const double* depth;
  const double* landmark_xyz;  // 3 per feature (world) = the pixelwise warp (one lane per unit), or NULL = the affine warp
  double* px_cur;
  int32_t* result;
  double* f_cur;
  int32_t* search_level;
  double* h_inv;
  double* A_cur_ref;
  // seeds
  double* state;
  uint8_t* success;
  unsigned int* unit_counts;  // 4 per feature: warps, ZMSSD evaluations, align iterations, filter updates
  // plain epipolar matches (stereo seam)
  const Rigid* T_cur_ref;     // n_ref_frames x n_cur_frames, or NULL = from the poses
  const double* d_inv;        // 3 per feature, or NULL = d_inv_common
  double d_inv_common[3];
  double* depth_out;
  // SVOH_BATCH_WHOLE_SETS (seed batches over resident columns): ref_frame_idx / px / f / grad / level above are not read by the packed
  // kernel, which walks the reference frames' tile-ordered columns; the other geometries get them filled in by the prologue
  int whole_sets;
};

constexpr int kPwbStride = 100;
constexpr int kG8MaxUnits = 49152;   // up to here a launch uses eight lanes per unit (measured crossover)

// indices of feature i are usable (always true for host-resident batches, which the host validates)
__device__ __forceinline__ bool feature_indices_ok(const MatcherArgs& a, int i)
{
  const int ri = a.ref_frame_idx[i];
  if ((unsigned)ri >= (unsigned)a.n_ref_frames) return false;
  if (a.cur_frame_idx && (unsigned)a.cur_frame_idx[i] >= (unsigned)a.n_cur_frames) return false;
  const int lv = a.level[i];
  return lv >= 0 && lv < a.ref_frames[ri].n_levels;
}

struct MatcherState {
  unsigned char* pwb;  // this unit's 10x10 patch with border, in LDS
  int sub;             // eight lanes per unit (template G8): this lane's row of the patch; unused otherwise
  double A[4];         // A_cur_ref, col-major
  double epi_image[2];
  double epi_dir[2];   // normalised epipolar direction (the 1-D refinement's direction)
  double epi_length_pyramid;
  double h_inv;
  int search_level;
  bool reject;
  bool align_1d;
  double px_cur[2];
  Vec3 f_cur;
  int n_warp, n_zmssd, n_align_it;  // work counters
  unsigned tpl[16];    // packed geometry (G == 2): the 8x8 template as 16 dwords; unused (and optimised away) otherwise
#ifdef SVOH_SEED_STAMPS
  long long t[4], tlast;  // diagnostic builds: cycles in geometry / warp / scan / align+rest
#endif
};

#ifdef SVOH_SEED_STAMPS
#define SVOH_MSTAMP(m, k) do { const long long now_ = clock64(); (m).t[k] += now_ - (m).tlast; (m).tlast = now_; } while (0)
#else
#define SVOH_MSTAMP(m, k) do { } while (0)
#endif

__device__ __forceinline__ int patch_at(const unsigned char* pwb, int r) { return pwb[((r >> 3) + 1) * 10 + (r & 7) + 1]; }

__device__ __forceinline__ void normalize2(double& x, double& y)
{
  const double z = x * x + y * y;
  if (z > 0.0) { const double n = sqrt(z); x /= n; y /= n; }
}
__device__ __forceinline__ void normalize3(Vec3& v)
{
  const double z = v.x * v.x + v.y * v.y + v.z * v.z;
  if (z > 0.0) { const double n = sqrt(z); v.x /= n; v.y /= n; v.z /= n; }
}
__device__ __forceinline__ void mat2d_inverse(const double m[4], double r[4])
{
  const double det = m[0] * m[3] - m[1] * m[2];
  const double invdet = 1.0 / det;
  r[0] = m[3] * invdet; r[1] = -m[1] * invdet; r[2] = -m[2] * invdet; r[3] = m[0] * invdet;
}
__device__ __forceinline__ bool is_edgelet(int t)
{
  return t == SVOH_FT_EDGELET || t == SVOH_FT_EDGELET_SEED || t == SVOH_FT_EDGELET_SEED_CONVERGED;
}

// patch_warp.cpp:20-60 (pinhole cameras)
__device__ void get_warp_matrix_affine(const CamModel& cam_ref, const CamModel& cam_cur, double pxr, double pyr,
                                       const Vec3& f_ref, double depth_ref, const Rigid& T_cur_ref, int level_ref,
                                       double A[4])
{
  const int kHalfPatchSize = 5;
  const Vec3 xyz_ref = { f_ref.x * depth_ref, f_ref.y * depth_ref, f_ref.z * depth_ref };
  Vec3 du = back_project3(cam_ref, pxr + (double)kHalfPatchSize * (1 << level_ref), pyr + 0.0 * (1 << level_ref));
  Vec3 dv = back_project3(cam_ref, pxr + 0.0 * (1 << level_ref), pyr + (double)kHalfPatchSize * (1 << level_ref));
  du.x *= xyz_ref.z; du.y *= xyz_ref.z; du.z *= xyz_ref.z;
  dv.x *= xyz_ref.z; dv.y *= xyz_ref.z; dv.z *= xyz_ref.z;
  double cu, cv, duu, duv, dvu, dvv;
  project3(cam_cur, transform(T_cur_ref, xyz_ref), cu, cv);
  project3(cam_cur, transform(T_cur_ref, du), duu, duv);
  project3(cam_cur, transform(T_cur_ref, dv), dvu, dvv);
  A[0] = (duu - cu) / kHalfPatchSize;
  A[1] = (duv - cv) / kHalfPatchSize;
  A[2] = (dvu - cu) / kHalfPatchSize;
  A[3] = (dvv - cv) / kHalfPatchSize;
}

// patch_warp.cpp:97-110
__device__ __forceinline__ int get_best_search_level(const double A[4], int max_level)
{
  int search_level = 0;
  double D = A[0] * A[3] - A[1] * A[2];
  while (D > 3.0 && search_level < max_level) {
    search_level += 1;
    D *= 0.25;
  }
  return search_level;
}

// patch_warp.cpp:112-156, halfpatch_size = 5 (10x10 patch with border).
// Two passes so that the gathers have memory-level parallelism: pass 1 evaluates the
// reference's in-image test for all 100 sample positions without touching memory
// (the reference returns false at the first failing pixel and the caller discards the
// patch, so "any pixel fails" is the same outcome); pass 2 recomputes the same float
// coordinates and issues a whole row of taps before using them.
__device__ bool warp_affine(const double A_cur_ref[4], const DevImage& img_ref, double pxr, double pyr, int level_ref,
                            int search_level, unsigned char* patch)
{
  constexpr int halfpatch_size = 5;
  double Ai[4];
  mat2d_inverse(A_cur_ref, Ai);
  const float s = (float)(1 << search_level);
  const float a00 = (float)Ai[0] * s, a10 = (float)Ai[1] * s, a01 = (float)Ai[2] * s, a11 = (float)Ai[3] * s;
  if (a00 != a00) return false;
  const float prx = (float)pxr / (float)(1 << level_ref);
  const float pry = (float)pyr / (float)(1 << level_ref);
  const int stride = img_ref.pitch;
  bool inside = true;
  for (int y = -halfpatch_size; y < halfpatch_size; ++y) {
#pragma unroll
    for (int x = -halfpatch_size; x < halfpatch_size; ++x) {
      const float fx = (float)x, fy = (float)y;
      const float pxx = (a00 * fx + a01 * fy) + prx;
      const float pxy = (a10 * fx + a11 * fy) + pry;
      const int xi = (int)floorf(pxx);
      const int yi = (int)floorf(pxy);
      inside = inside && pxx == pxx && pxy == pxy && !(xi < 0 || yi < 0 || xi >= img_ref.w - 1 || yi >= img_ref.h - 1);  // no `xi + 1`: xi may be INT_MAX
    }
  }
  if (!inside) return false;
  for (int y = -halfpatch_size; y < halfpatch_size; ++y) {
    unsigned t00[10], t01[10], t10[10], t11[10];
    float sx[10], sy[10];
#pragma unroll
    for (int x = -halfpatch_size; x < halfpatch_size; ++x) {
      const int k = x + halfpatch_size;
      const float fx = (float)x, fy = (float)y;
      const float pxx = (a00 * fx + a01 * fy) + prx;
      const float pxy = (a10 * fx + a11 * fy) + pry;
      const int xi = (int)floorf(pxx);
      const int yi = (int)floorf(pxy);
      sx[k] = pxx - xi;
      sy[k] = pxy - yi;
      const uint8_t* ptr = img_ref.data + (ptrdiff_t)yi * stride + xi;
      t00[k] = ptr[0]; t10[k] = ptr[1]; t01[k] = ptr[stride]; t11[k] = ptr[stride + 1];
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const float subpix_x = sx[k], subpix_y = sy[k];
      const float w00 = (1.0f - subpix_x) * (1.0f - subpix_y);
      const float w01 = (1.0f - subpix_x) * subpix_y;
      const float w10 = subpix_x * (1.0f - subpix_y);
      const float w11 = 1.0f - w00 - w01 - w10;
      patch[(y + halfpatch_size) * 10 + k] = (unsigned char)(w00 * t00[k] + w01 * t01[k] + w10 * t10[k] + w11 * t11[k]);
    }
  }
  return true;
}

// patch_score.h:264-283 with the constructor's sums (patch_score.h:80-92)
// warp::warpPixelwise (patch_warp.cpp:158-230; Matcher::Options::use_affine_warp_ == false, matcher.cpp:67-81): every
// pixel of the 10x10 patch (search level of the current frame) is back-projected to the landmark's distance from the
// current camera, carried into the reference frame and sampled in the reference level.  One lane per unit: nothing in
// the reference clears the flag, the branch exists for completeness (svoh_match_direct_batch_pixelwise).
__device__ bool warp_pixelwise(const DevFrameView& cur_frame, const DevFrameView& ref_frame, double pxr, double pyr,
                               const Vec3& landmark, int level_ref, int level_cur, unsigned char* patch)
{
  constexpr int halfpatch_size = 5;
  const Rigid T_w_ref = inverse(ref_frame.T_f_w), T_w_cur = inverse(cur_frame.T_f_w);   // Frame::pos() = T_world_cam().getPosition()
  const double dr0 = T_w_ref.t.x - landmark.x, dr1 = T_w_ref.t.y - landmark.y, dr2 = T_w_ref.t.z - landmark.z;
  const double dc0 = T_w_cur.t.x - landmark.x, dc1 = T_w_cur.t.y - landmark.y, dc2 = T_w_cur.t.z - landmark.z;
  const double depth_ref = sqrt(dr0 * dr0 + dr1 * dr1 + dr2 * dr2);
  const double depth_cur = sqrt(dc0 * dc0 + dc1 * dc1 + dc2 * dc2);
  Vec3 xyz_ref = back_project3(ref_frame.cam, pxr, pyr);
  normalize3(xyz_ref);
  xyz_ref.x *= depth_ref; xyz_ref.y *= depth_ref; xyz_ref.z *= depth_ref;
  const Rigid T_cur_ref = mul(cur_frame.T_f_w, T_w_ref);
  double pcx, pcy;
  project3(cur_frame.cam, transform(T_cur_ref, xyz_ref), pcx, pcy);
  const double pcs0 = pcx / (1 << level_cur), pcs1 = pcy / (1 << level_cur);
  const Rigid T_ref_cur = mul(ref_frame.T_f_w, T_w_cur);
  const DevImage& img_ref = ref_frame.lv[level_ref];
  const int stride = img_ref.pitch;
  for (int y = -halfpatch_size; y < halfpatch_size; ++y) {
    for (int x = -halfpatch_size; x < halfpatch_size; ++x) {
      const double es0 = (double)x + pcs0, es1 = (double)y + pcs1;
      Vec3 e_cur = back_project3(cur_frame.cam, es0 * (1 << level_cur), es1 * (1 << level_cur));
      normalize3(e_cur);
      e_cur.x *= depth_cur; e_cur.y *= depth_cur; e_cur.z *= depth_cur;
      double er0, er1;
      project3(ref_frame.cam, transform(T_ref_cur, e_cur), er0, er1);
      er0 = er0 / (1 << level_ref); er1 = er1 / (1 << level_ref);
      const int xi = (int)floor(er0);
      const int yi = (int)floor(er1);
      if (!(er0 == er0) || !(er1 == er1) || xi < 0 || yi < 0 || xi + 1 >= img_ref.w || yi + 1 >= img_ref.h) return false;
      const float subpix_x = (float)(er0 - xi);
      const float subpix_y = (float)(er1 - yi);
      const float w00 = (1.0f - subpix_x) * (1.0f - subpix_y);
      const float w01 = (1.0f - subpix_x) * subpix_y;
      const float w10 = subpix_x * (1.0f - subpix_y);
      const float w11 = 1.0f - w00 - w01 - w10;
      const uint8_t* ptr = img_ref.data + (ptrdiff_t)yi * stride + xi;
      patch[(y + halfpatch_size) * 10 + (x + halfpatch_size)] =
          (unsigned char)(w00 * ptr[0] + w01 * ptr[stride] + w10 * ptr[1] + w11 * ptr[stride + 1]);
    }
  }
  return true;
}

__device__ int zmssd_score(const unsigned char* pwb, int sumA, int sumAA, const uint8_t* cur_patch, int stride)
{
  unsigned sumB = 0, sumBB = 0, sumAB = 0;
  unsigned cpx[64];
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    const uint8_t* p = cur_patch + (ptrdiff_t)y * stride;
#pragma unroll
    for (int x = 0; x < 8; ++x) cpx[y * 8 + x] = p[x];  // all 64 loads are independent: issue them together
  }
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    const unsigned c = cpx[r];
    sumB += c; sumBB += c * c; sumAB += c * (unsigned)patch_at(pwb, r);
  }
  const int iB = (int)sumB, iBB = (int)sumBB, iAB = (int)sumAB;
  return sumAA - 2 * iAB + iBB - (sumA * sumA - 2 * sumA * iB + iB * iB) / 64;
}

__device__ void mat3f_inverse(const float* m, float* r)
{
#define M3(i, j) m[(i) * 3 + (j)]
#define COF3(i, j) (M3(((i) + 1) % 3, ((j) + 1) % 3) * M3(((i) + 2) % 3, ((j) + 2) % 3) - M3(((i) + 1) % 3, ((j) + 2) % 3) * M3(((i) + 2) % 3, ((j) + 1) % 3))
  const float c00 = COF3(0, 0), c10 = COF3(1, 0), c20 = COF3(2, 0);
  const float det = (c00 * M3(0, 0) + c10 * M3(1, 0)) + c20 * M3(2, 0);
  const float invdet = 1.0f / det;
  r[0] = c00 * invdet; r[1] = c10 * invdet; r[2] = c20 * invdet;
  r[3] = COF3(0, 1) * invdet; r[4] = COF3(1, 1) * invdet; r[5] = COF3(2, 1) * invdet;
  r[6] = COF3(0, 2) * invdet; r[7] = COF3(1, 2) * invdet; r[8] = COF3(2, 2) * invdet;
#undef COF3
#undef M3
}

#define M4(i, j) m[(i) * 4 + (j)]
__device__ __forceinline__ float det3_helper(const float* m, int i1, int i2, int i3, int j1, int j2, int j3)
{
  return M4(i1, j1) * (M4(i2, j2) * M4(i3, j3) - M4(i2, j3) * M4(i3, j2));
}
__device__ __forceinline__ float cofactor4(const float* m, int i, int j)
{
  const int i1 = (i + 1) % 4, i2 = (i + 2) % 4, i3 = (i + 3) % 4;
  const int j1 = (j + 1) % 4, j2 = (j + 2) % 4, j3 = (j + 3) % 4;
  return det3_helper(m, i1, i2, i3, j1, j2, j3) + det3_helper(m, i2, i3, i1, j1, j2, j3) + det3_helper(m, i3, i1, i2, j1, j2, j3);
}
__device__ void mat4f_inverse(const float* m, float* r)
{
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float c = cofactor4(m, i, j);
      r[j * 4 + i] = ((i + j) & 1) ? -c : c;
    }
  const float d = ((M4(0, 0) * r[0] + M4(1, 0) * r[1]) + M4(2, 0) * r[2]) + M4(3, 0) * r[3];
#pragma unroll
  for (int k = 0; k < 16; ++k) r[k] /= d;
}
#undef M4

// feature_alignment.cpp:31-209
// ---- eight lanes per unit (small batches) -----------------------------------------------------------------------
// A launch of a few hundred or thousand units is as slow as its slowest lane, and a lone lane needs ~0.1 ms for
// warp + scan + refinement (every instruction waits for the one before).  For small batches eight neighbouring
// lanes share a unit: lane `sub` owns row `sub` of the 8x8 patch (and every eighth pixel of the 10x10 warp).
// Integer sums (ZMSSD) are combined with three exchanges; the order-dependent float sums of the refinements
// are handed from row to row, so every unit goes through exactly the additions of the one-lane code and the
// results are bit-identical.  The scalar geometry is computed by all eight lanes alike.
__device__ __forceinline__ int g8_sum(int v)
{
  v += __shfl_xor(v, 1, 8); v += __shfl_xor(v, 2, 8); v += __shfl_xor(v, 4, 8);
  return v;
}
__device__ __forceinline__ float g8_sum_exact(float v)   // only for sums whose every partial is exactly representable
{
  v += __shfl_xor(v, 1, 8); v += __shfl_xor(v, 2, 8); v += __shfl_xor(v, 4, 8);
  return v;
}
__device__ __forceinline__ bool g8_all(bool p)
{
  int v = p ? 1 : 0;
  v &= __shfl_xor(v, 1, 8); v &= __shfl_xor(v, 2, 8); v &= __shfl_xor(v, 4, 8);
  return v != 0;
}
// the LDS patch is written and read by different lanes of the same wave
__device__ __forceinline__ void g8_lds_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// warp_affine with the 100 pixels dealt out to the eight lanes (pixel k = y*10 + x to lane k % 8)
__device__ bool warp_affine_g8(const double A_cur_ref[4], const DevImage& img_ref, double pxr, double pyr, int level_ref,
                               int search_level, unsigned char* patch, int sub)
{
  constexpr int halfpatch_size = 5;
  double Ai[4];
  mat2d_inverse(A_cur_ref, Ai);
  const float s = (float)(1 << search_level);
  const float a00 = (float)Ai[0] * s, a10 = (float)Ai[1] * s, a01 = (float)Ai[2] * s, a11 = (float)Ai[3] * s;
  if (a00 != a00) return false;
  const float prx = (float)pxr / (float)(1 << level_ref);
  const float pry = (float)pyr / (float)(1 << level_ref);
  const int stride = img_ref.pitch;
  bool inside = true;
  for (int k = sub; k < 100; k += 8) {
    const float fx = (float)(k % 10 - halfpatch_size), fy = (float)(k / 10 - halfpatch_size);
    const float pxx = (a00 * fx + a01 * fy) + prx;
    const float pxy = (a10 * fx + a11 * fy) + pry;
    const int xi = (int)floorf(pxx);
    const int yi = (int)floorf(pxy);
    inside = inside && pxx == pxx && pxy == pxy && !(xi < 0 || yi < 0 || xi >= img_ref.w - 1 || yi >= img_ref.h - 1);
  }
  if (!g8_all(inside)) return false;
  // a lane's 12 or 13 samples: all tap loads first (two unaligned 2-byte loads per sample, every one of them in flight
  // together), then the arithmetic -- one round trip to memory per lane instead of one per sample
  unsigned short t0[13], t1[13];
#pragma unroll
  for (int j = 0; j < 13; ++j) {
    const int k = sub + 8 * j;
    t0[j] = 0; t1[j] = 0;
    if (k < 100) {
      const float fx = (float)(k % 10 - halfpatch_size), fy = (float)(k / 10 - halfpatch_size);
      const float pxx = (a00 * fx + a01 * fy) + prx;
      const float pxy = (a10 * fx + a11 * fy) + pry;
      const int xi = (int)floorf(pxx);
      const int yi = (int)floorf(pxy);
      const uint8_t* ptr = img_ref.data + (ptrdiff_t)yi * stride + xi;
      __builtin_memcpy(&t0[j], ptr, 2);
      __builtin_memcpy(&t1[j], ptr + stride, 2);
    }
  }
#pragma unroll
  for (int j = 0; j < 13; ++j) {
    const int k = sub + 8 * j;
    if (k < 100) {
      const float fx = (float)(k % 10 - halfpatch_size), fy = (float)(k / 10 - halfpatch_size);
      const float pxx = (a00 * fx + a01 * fy) + prx;
      const float pxy = (a10 * fx + a11 * fy) + pry;
      const int xi = (int)floorf(pxx);
      const int yi = (int)floorf(pxy);
      const float subpix_x = pxx - xi, subpix_y = pxy - yi;
      const unsigned t00 = t0[j] & 255u, t10 = t0[j] >> 8, t01 = t1[j] & 255u, t11 = t1[j] >> 8;
      const float w00 = (1.0f - subpix_x) * (1.0f - subpix_y);
      const float w01 = (1.0f - subpix_x) * subpix_y;
      const float w10 = subpix_x * (1.0f - subpix_y);
      const float w11 = 1.0f - w00 - w01 - w10;
      patch[k] = (unsigned char)(w00 * t00 + w01 * t01 + w10 * t10 + w11 * t11);
    }
  }
  g8_lds_fence();
  return true;
}

// the same with the 100 pixels dealt out to the 64 lanes of a wave (geometry 3: pixel k to lane k % 64, two samples per lane)
__device__ bool warp_affine_w64(const double A_cur_ref[4], const DevImage& img_ref, double pxr, double pyr, int level_ref,
                                int search_level, unsigned char* patch)
{
  constexpr int halfpatch_size = 5;
  const int lane = (int)(threadIdx.x & 63);
  double Ai[4];
  mat2d_inverse(A_cur_ref, Ai);
  const float s = (float)(1 << search_level);
  const float a00 = (float)Ai[0] * s, a10 = (float)Ai[1] * s, a01 = (float)Ai[2] * s, a11 = (float)Ai[3] * s;
  if (a00 != a00) return false;
  const float prx = (float)pxr / (float)(1 << level_ref);
  const float pry = (float)pyr / (float)(1 << level_ref);
  const int stride = img_ref.pitch;
  bool inside = true;
  float pxx[2], pxy[2];
  int xi[2], yi[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = lane + 64 * j;
    const float fx = (float)(k % 10 - halfpatch_size), fy = (float)(k / 10 - halfpatch_size);
    pxx[j] = (a00 * fx + a01 * fy) + prx;
    pxy[j] = (a10 * fx + a11 * fy) + pry;
    xi[j] = (int)floorf(pxx[j]);
    yi[j] = (int)floorf(pxy[j]);
    if (k < 100)
      inside = inside && pxx[j] == pxx[j] && pxy[j] == pxy[j] && !(xi[j] < 0 || yi[j] < 0 || xi[j] >= img_ref.w - 1 || yi[j] >= img_ref.h - 1);
  }
  if (!__all(inside)) return false;
  unsigned short t0[2] = { 0, 0 }, t1[2] = { 0, 0 };
#pragma unroll
  for (int j = 0; j < 2; ++j)
    if (lane + 64 * j < 100) {
      const uint8_t* ptr = img_ref.data + (ptrdiff_t)yi[j] * stride + xi[j];
      __builtin_memcpy(&t0[j], ptr, 2);
      __builtin_memcpy(&t1[j], ptr + stride, 2);
    }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = lane + 64 * j;
    if (k < 100) {
      const float subpix_x = pxx[j] - xi[j], subpix_y = pxy[j] - yi[j];
      const unsigned t00 = t0[j] & 255u, t10 = t0[j] >> 8, t01 = t1[j] & 255u, t11 = t1[j] >> 8;
      const float w00 = (1.0f - subpix_x) * (1.0f - subpix_y);
      const float w01 = (1.0f - subpix_x) * subpix_y;
      const float w10 = subpix_x * (1.0f - subpix_y);
      const float w11 = 1.0f - w00 - w01 - w10;
      patch[k] = (unsigned char)(w00 * t00 + w01 * t01 + w10 * t10 + w11 * t11);
    }
  }
  g8_lds_fence();
  return true;
}

// PatchScore constructor sums (patch_score.h:80-92), one row per lane
__device__ __forceinline__ void patch_sums_g8(const unsigned char* pwb, int sub, int& sumA, int& sumAA)
{
  int a = 0, aa = 0;
#pragma unroll
  for (int x = 0; x < 8; ++x) { const int n = patch_at(pwb, sub * 8 + x); a += n; aa += n * n; }
  sumA = g8_sum(a); sumAA = g8_sum(aa);
}

// zmssd_score, one row per lane (integer sums: any order)
__device__ int zmssd_score_g8(const unsigned char* pwb, int sumA, int sumAA, const uint8_t* cur_patch, int stride, int sub)
{
  // the lane's row of the current patch as one unaligned 8-byte load, its template row from three aligned LDS dwords;
  // the three integer sums as v_dot4_u32_u8 (same integers as the byte-by-byte loop)
  uint2 c;
  __builtin_memcpy(&c, cur_patch + (ptrdiff_t)sub * stride, 8);
  const int o = (sub + 1) * 10 + 1;                        // byte offset of the row's first pixel in the 10x10 patch
  const unsigned* w = reinterpret_cast<const unsigned*>(pwb) + (o >> 2);
  const unsigned w0 = w[0], w1 = w[1], w2 = w[2];
  const unsigned t0 = __builtin_amdgcn_alignbyte(w1, w0, (unsigned)(o & 3)), t1 = __builtin_amdgcn_alignbyte(w2, w1, (unsigned)(o & 3));
  unsigned sumB = __builtin_amdgcn_udot4(c.x, 0x01010101u, 0u, false);
  sumB = __builtin_amdgcn_udot4(c.y, 0x01010101u, sumB, false);
  unsigned sumBB = __builtin_amdgcn_udot4(c.x, c.x, 0u, false);
  sumBB = __builtin_amdgcn_udot4(c.y, c.y, sumBB, false);
  unsigned sumAB = __builtin_amdgcn_udot4(c.x, t0, 0u, false);
  sumAB = __builtin_amdgcn_udot4(c.y, t1, sumAB, false);
  const int iB = g8_sum((int)sumB), iBB = g8_sum((int)sumBB), iAB = g8_sum((int)sumAB);
  return sumAA - 2 * iAB + iBB - (sumA * sumA - 2 * sumA * iB + iB * iB) / 64;
}

// The two sub-pixel refinements are written as (set-up, one iteration).  A block-wide job queue that hands the
// iterations of finished lanes' successors to idle lanes was built on this split and measured: no gain with one
// job per lane (the slowest job still sets the pace: 0.3 % of the seeds need all 10 iterations, 4 % five or more,
// so most waves run 5-7 trips for a mean of 2.8) -- it needs several seeds per lane, i.e. a global queue.
struct AlignIter {
  float Hinv[16];      // 4x4 (2-D) or 3x3 in the first nine entries (1-D)
  float u, v, mean_diff, alpha;
  double dir0, dir1;   // 1-D only
  int n_it;
};
enum { ALIGN_CONTINUE = 0, ALIGN_CONVERGED = 1, ALIGN_STOPPED = 2, ALIGN_NAN = 3 };

// feature_alignment.cpp:31-209, the part before the iterations (H, its inverse, h_inv)
__device__ void align_1d_init(double dir0, double dir1, const unsigned char* pwb, bool affine_est_offset, bool affine_est_gain,
                              double px, double py, double* h_inv, AlignIter& it)
{
  constexpr int kPatchSize = 8, ref_step = 10;
  float H[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  for (int y = 0; y < kPatchSize; ++y) {
    const unsigned char* p = pwb + (y + 1) * ref_step + 1;
    for (int x = 0; x < kPatchSize; ++x, ++p) {
      float J[3];
      const float dx = (float)p[1] - (float)p[-1];
      const float dy = (float)p[ref_step] - (float)p[-ref_step];
      J[0] = (float)(0.5f * (dir0 * dx + dir1 * dy));
      J[1] = affine_est_offset ? 1.0f : 0.0f;
      J[2] = affine_est_gain ? -1.0f * p[0] : 0.0f;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) H[r * 3 + c] += J[r] * J[c];
    }
  }
  if (!affine_est_offset) H[4] = 1.0f;
  if (!affine_est_gain) H[8] = 1.0f;
  if (h_inv) *h_inv = 1.0 / H[0] * kPatchSize * kPatchSize;
  mat3f_inverse(H, it.Hinv);
  it.mean_diff = 0;
  it.alpha = 1.0;
  it.u = (float)px;
  it.v = (float)py;
  it.dir0 = dir0; it.dir1 = dir1;
  it.n_it = 0;
}

// one pass of the loop body of feature_alignment.cpp:103-204
__device__ int align_1d_step(const DevImage& cur_img, const unsigned char* pwb, bool affine_est_offset, bool affine_est_gain,
                             AlignIter& it)
{
  constexpr int kHalfPatchSize = 4, kPatchSize = 8, ref_step = 10;
  const float min_update_squared = (float)(0.03 * 0.03);
  const int cur_step = cur_img.pitch;
  const double dir0 = it.dir0, dir1 = it.dir1;
  float u = it.u, v = it.v;
  const int u_r = (int)floorf(u);
  const int v_r = (int)floorf(v);
  if (u_r < kHalfPatchSize || v_r < kHalfPatchSize || u_r >= cur_img.w - kHalfPatchSize || v_r >= cur_img.h - kHalfPatchSize)
    return ALIGN_STOPPED;
  if (u != u || v != v) return ALIGN_NAN;
  ++it.n_it;
  const float subpix_x = u - u_r;
  const float subpix_y = v - v_r;
  const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
  const float wTR = (float)(subpix_x * (1.0 - subpix_y));
  const float wBL = (float)((1.0 - subpix_x) * subpix_y);
  const float wBR = subpix_x * subpix_y;
  float Jres[3] = { 0, 0, 0 };
#pragma unroll 2
  for (int y = 0; y < kPatchSize; ++y) {
    const uint8_t* p = cur_img.data + (ptrdiff_t)(v_r + y - kHalfPatchSize) * cur_step + u_r - kHalfPatchSize;
    const unsigned char* rp = pwb + (y + 1) * ref_step + 1;
    for (int x = 0; x < kPatchSize; ++x, ++p, ++rp) {
      const float gdx = (float)rp[1] - (float)rp[-1];
      const float gdy = (float)rp[ref_step] - (float)rp[-ref_step];
      const float ref_dv = (float)(0.5f * (dir0 * gdx + dir1 * gdy));
      const float cur_intensity = wTL * p[0] + wTR * p[1] + wBL * p[cur_step] + wBR * p[cur_step + 1];
      const float res = cur_intensity - it.alpha * rp[0] + it.mean_diff;
      Jres[0] -= res * ref_dv;
      if (affine_est_offset) Jres[1] -= res;
      if (affine_est_gain) Jres[2] -= (-1) * res * rp[0];
    }
  }
  if (!affine_est_offset) Jres[1] = 0.0f;
  if (!affine_est_gain) Jres[2] = 0.0f;
  float update[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) update[r] = (it.Hinv[r * 3 + 0] * Jres[0] + it.Hinv[r * 3 + 1] * Jres[1]) + it.Hinv[r * 3 + 2] * Jres[2];
  it.u = (float)(u + update[0] * dir0);
  it.v = (float)(v + update[0] * dir1);
  it.mean_diff += update[1];
  it.alpha += update[2];
  return (update[0] * update[0] < min_update_squared) ? ALIGN_CONVERGED : ALIGN_CONTINUE;
}

// feature_alignment.cpp:31-209
__device__ bool align_1d(const DevImage& cur_img, double dir0, double dir1, const unsigned char* pwb, int n_iter,
                         bool affine_est_offset, bool affine_est_gain, double& px, double& py, double* h_inv,
                         int& n_it)
{
  AlignIter it;
  align_1d_init(dir0, dir1, pwb, affine_est_offset, affine_est_gain, px, py, h_inv, it);
  bool converged = false;
  for (int iter = 0; iter < n_iter; ++iter) {
    const int st = align_1d_step(cur_img, pwb, affine_est_offset, affine_est_gain, it);
    if (st == ALIGN_NAN) { n_it += it.n_it; return false; }
    if (st == ALIGN_STOPPED) break;
    if (st == ALIGN_CONVERGED) { converged = true; break; }
  }
  n_it += it.n_it;
  px = it.u;
  py = it.v;
  return converged;
}

// feature_alignment.cpp:212-391, the part before the iterations
__device__ void align_2d_init(const unsigned char* pwb, bool affine_est_offset, bool affine_est_gain, double px, double py,
                              AlignIter& it)
{
  constexpr int patch_size_ = 8, ref_step = 10;
  float H[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  for (int y = 0; y < patch_size_; ++y) {
    const unsigned char* p = pwb + (y + 1) * ref_step + 1;
    for (int x = 0; x < patch_size_; ++x, ++p) {
      float J[4];
      J[0] = (float)(0.5 * ((int)p[1] - (int)p[-1]));
      J[1] = (float)(0.5 * ((int)p[ref_step] - (int)p[-ref_step]));
      J[2] = affine_est_offset ? 1.0f : 0.0f;
      J[3] = affine_est_gain ? (float)(-1.0 * p[0]) : 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) H[r * 4 + c] += J[r] * J[c];
    }
  }
  if (!affine_est_offset) H[10] = 1.0f;
  if (!affine_est_gain) H[15] = 1.0f;
  mat4f_inverse(H, it.Hinv);
  it.mean_diff = 0;
  it.alpha = 1.0;
  it.u = (float)px;
  it.v = (float)py;
  it.dir0 = it.dir1 = 0.0;
  it.n_it = 0;
}

// one pass of the loop body of feature_alignment.cpp:300-384
__device__ int align_2d_step(const DevImage& cur_img, const unsigned char* pwb, bool affine_est_offset, bool affine_est_gain,
                             AlignIter& it)
{
  constexpr int halfpatch_size_ = 4, patch_size_ = 8, ref_step = 10;
  const float min_update_squared = (float)(0.03 * 0.03);
  const int cur_step = cur_img.pitch;
  const float u = it.u, v = it.v;
  const int u_r = (int)floorf(u);
  const int v_r = (int)floorf(v);
  if (u_r < halfpatch_size_ || v_r < halfpatch_size_ || u_r >= cur_img.w - halfpatch_size_ || v_r >= cur_img.h - halfpatch_size_)
    return ALIGN_STOPPED;
  if (u != u || v != v) return ALIGN_NAN;
  ++it.n_it;
  const float subpix_x = u - u_r;
  const float subpix_y = v - v_r;
  const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
  const float wTR = (float)(subpix_x * (1.0 - subpix_y));
  const float wBL = (float)((1.0 - subpix_x) * subpix_y);
  const float wBR = subpix_x * subpix_y;
  float Jres[4] = { 0, 0, 0, 0 };
#pragma unroll 2
  for (int y = 0; y < patch_size_; ++y) {
    const uint8_t* p = cur_img.data + (ptrdiff_t)(v_r + y - halfpatch_size_) * cur_step + u_r - halfpatch_size_;
    const unsigned char* rp = pwb + (y + 1) * ref_step + 1;
    for (int x = 0; x < patch_size_; ++x, ++p, ++rp) {
      const float ref_dx = (float)(0.5 * ((int)rp[1] - (int)rp[-1]));
      const float ref_dy = (float)(0.5 * ((int)rp[ref_step] - (int)rp[-ref_step]));
      const float search_pixel = wTL * p[0] + wTR * p[1] + wBL * p[cur_step] + wBR * p[cur_step + 1];
      const float res = search_pixel - it.alpha * rp[0] + it.mean_diff;
      Jres[0] -= res * ref_dx;
      Jres[1] -= res * ref_dy;
      if (affine_est_offset) Jres[2] -= res;
      if (affine_est_gain) Jres[3] -= (-1) * res * rp[0];
    }
  }
  if (!affine_est_offset) Jres[2] = 0.0f;
  if (!affine_est_gain) Jres[3] = 0.0f;
  float update[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    update[r] = ((it.Hinv[r * 4 + 0] * Jres[0] + it.Hinv[r * 4 + 1] * Jres[1]) + it.Hinv[r * 4 + 2] * Jres[2]) + it.Hinv[r * 4 + 3] * Jres[3];
  it.u = u + update[0];
  it.v = v + update[1];
  it.mean_diff += update[2];
  it.alpha += update[3];
  return (update[0] * update[0] + update[1] * update[1] < min_update_squared) ? ALIGN_CONVERGED : ALIGN_CONTINUE;
}

// feature_alignment.cpp:212-391
__device__ bool align_2d(const DevImage& cur_img, const unsigned char* pwb, int n_iter, bool affine_est_offset,
                         bool affine_est_gain, double& px, double& py, int& n_it)
{
  AlignIter it;
  align_2d_init(pwb, affine_est_offset, affine_est_gain, px, py, it);
  bool converged = false;
  for (int iter = 0; iter < n_iter; ++iter) {
    const int st = align_2d_step(cur_img, pwb, affine_est_offset, affine_est_gain, it);
    if (st == ALIGN_NAN) { n_it += it.n_it; return false; }
    if (st == ALIGN_STOPPED) break;
    if (st == ALIGN_CONVERGED) { converged = true; break; }
  }
  n_it += it.n_it;
  px = it.u;
  py = it.v;
  return converged;
}

// ---- the refinements with L lanes per unit (L = 8: the eight-lane geometry; L = 4: the packed geometry's jobs) ----
// acc[a] <- acc[a] - t[a][0] - ... over the 64 pixels in row order: lane `sub` owns 64 / L consecutive pixels (8 / L
// patch rows) and continues where lane sub-1 stopped, so the additions are those of the one-lane loops (a sum is the
// same chain with negated terms).
template <int NA, int L>
__device__ __forceinline__ void gl_chain_sub(float (&acc)[NA], const float (&t)[NA][64 / L], int sub)
{
  float c[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) c[a] = acc[a];
  for (int s = 0; s < L; ++s) {
    float in[NA];
    // lane i takes lane i-1's value on the VALU's DPP network (row_shr:1; one v_mov_dpp instead of a ds_bpermute round
    // trip through the LDS crossbar per stage and accumulator).  Rows are 16 lanes: the first lane of a later
    // group of a row receives the last lane of the group before it, which it never uses (sub == 0 starts from acc).
#pragma unroll
    for (int a = 0; a < NA; ++a)
      in[a] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c[a]), 0x111, 0xF, 0xF, false));
    if (sub == s) {
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        float v = (s == 0) ? acc[a] : in[a];
#pragma unroll
        for (int x = 0; x < 64 / L; ++x) v -= t[a][x];
        c[a] = v;
      }
    }
  }
#pragma unroll
  for (int a = 0; a < NA; ++a) acc[a] = __shfl(c[a], L - 1, L);
}

// nine consecutive pixels as one unaligned 12-byte load (the three spare bytes stay inside the row, the next row or
// the slab's tail padding): 2 loads per lane and iteration instead of 32 single-byte gathers, which is what the
// texture-address unit of a CU (one lane address per cycle for scattered loads) spent the refinement phase on
__device__ __forceinline__ void load_row9_u8(const uint8_t* p, unsigned (&px)[9])
{
  unsigned w[3];
  __builtin_memcpy(w, p, 12);
  px[0] = w[0] & 255u; px[1] = (w[0] >> 8) & 255u; px[2] = (w[0] >> 16) & 255u; px[3] = w[0] >> 24;
  px[4] = w[1] & 255u; px[5] = (w[1] >> 8) & 255u; px[6] = (w[1] >> 16) & 255u; px[7] = w[1] >> 24;
  px[8] = w[2] & 255u;
}

// One pass of the loop body of align_2d (feature_alignment.cpp:300-384) by the L lanes of a unit.  ALIGN_STOPPED /
// ALIGN_NAN leave every argument as it was.
template <int L>
__device__ __forceinline__ int align_2d_gl_iter(const DevImage& cur_img, const unsigned char* pwb, bool affine_est_offset,
                                                bool affine_est_gain, const float (&Hinv)[16], float& u, float& v,
                                                float& mean_diff, float& alpha, int& n_it, int sub)
{
  constexpr int halfpatch_size_ = 4, patch_size_ = 8, ref_step = 10, R = 8 / L;
  const float min_update_squared = (float)(0.03 * 0.03);
  const int cur_step = cur_img.pitch;
  const int u_r = (int)floorf(u);
  const int v_r = (int)floorf(v);
  if (u_r < halfpatch_size_ || v_r < halfpatch_size_ || u_r >= cur_img.w - halfpatch_size_ || v_r >= cur_img.h - halfpatch_size_)
    return ALIGN_STOPPED;
  if (u != u || v != v) return ALIGN_NAN;
  ++n_it;
  const float subpix_x = u - u_r;
  const float subpix_y = v - v_r;
  const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
  const float wTR = (float)(subpix_x * (1.0 - subpix_y));
  const float wBL = (float)((1.0 - subpix_x) * subpix_y);
  const float wBR = subpix_x * subpix_y;
  float t[4][8 * R];
  {
    const uint8_t* it = cur_img.data + (ptrdiff_t)(v_r + sub * R - halfpatch_size_) * cur_step + u_r - halfpatch_size_;
    unsigned rows[R + 1][9];
#pragma unroll
    for (int r = 0; r <= R; ++r) load_row9_u8(it + (ptrdiff_t)r * cur_step, rows[r]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const unsigned char* rp = pwb + (sub * R + r + 1) * ref_step + 1;
#pragma unroll
      for (int x = 0; x < patch_size_; ++x, ++rp) {
        const float ref_dx = (float)(0.5 * ((int)rp[1] - (int)rp[-1]));
        const float ref_dy = (float)(0.5 * ((int)rp[ref_step] - (int)rp[-ref_step]));
        const float search_pixel = wTL * rows[r][x] + wTR * rows[r][x + 1] + wBL * rows[r + 1][x] + wBR * rows[r + 1][x + 1];
        const float res = search_pixel - alpha * rp[0] + mean_diff;
        t[0][r * 8 + x] = res * ref_dx;
        t[1][r * 8 + x] = res * ref_dy;
        t[2][r * 8 + x] = affine_est_offset ? res : 0.0f;
        t[3][r * 8 + x] = affine_est_gain ? (-1) * res * rp[0] : 0.0f;
      }
    }
  }
  float Jres[4] = { 0, 0, 0, 0 };
  if (affine_est_gain) gl_chain_sub<4, L>(Jres, t, sub);
  else {   // the fourth sum is a sum of zeros then (and reset below): hand three accumulators from lane to lane
    float J3[3] = { 0, 0, 0 };
    float t3[3][8 * R];
#pragma unroll
    for (int a3 = 0; a3 < 3; ++a3)
#pragma unroll
      for (int x = 0; x < 8 * R; ++x) t3[a3][x] = t[a3][x];
    gl_chain_sub<3, L>(J3, t3, sub);
    Jres[0] = J3[0]; Jres[1] = J3[1]; Jres[2] = J3[2];
  }
  if (!affine_est_offset) Jres[2] = 0.0f;
  if (!affine_est_gain) Jres[3] = 0.0f;
  float update[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    update[r] = ((Hinv[r * 4 + 0] * Jres[0] + Hinv[r * 4 + 1] * Jres[1]) + Hinv[r * 4 + 2] * Jres[2]) + Hinv[r * 4 + 3] * Jres[3];
  u += update[0];
  v += update[1];
  mean_diff += update[2];
  alpha += update[3];
  return (update[0] * update[0] + update[1] * update[1] < min_update_squared) ? ALIGN_CONVERGED : ALIGN_CONTINUE;
}

// align_2d (feature_alignment.cpp:212-391).  The entries of H are sums of products of half-integers and
// integers below 2^16: every partial sum is exact in float, so the rows may be added in any order.
__device__ bool align_2d_g8(const DevImage& cur_img, const unsigned char* pwb, int n_iter, bool affine_est_offset,
                            bool affine_est_gain, double& px, double& py, int& n_it, int sub)
{
  constexpr int patch_size_ = 8, ref_step = 10;
  bool converged = false;
  float H[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  {
    const unsigned char* it = pwb + (sub + 1) * ref_step + 1;
    for (int x = 0; x < patch_size_; ++x, ++it) {
      float J[4];
      J[0] = (float)(0.5 * ((int)it[1] - (int)it[-1]));
      J[1] = (float)(0.5 * ((int)it[ref_step] - (int)it[-ref_step]));
      J[2] = affine_est_offset ? 1.0f : 0.0f;
      J[3] = affine_est_gain ? (float)(-1.0 * it[0]) : 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) H[r * 4 + c] += J[r] * J[c];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) H[k] = g8_sum_exact(H[k]);
  }
  if (!affine_est_offset) H[10] = 1.0f;
  if (!affine_est_gain) H[15] = 1.0f;
  float Hinv[16];
  mat4f_inverse(H, Hinv);
  float mean_diff = 0;
  float alpha = 1.0;
  float u = (float)px;
  float v = (float)py;
  for (int iter = 0; iter < n_iter; ++iter) {
    const int st = align_2d_gl_iter<8>(cur_img, pwb, affine_est_offset, affine_est_gain, Hinv, u, v, mean_diff, alpha, n_it, sub);
    if (st == ALIGN_NAN) return false;
    if (st == ALIGN_STOPPED) break;
    if (st == ALIGN_CONVERGED) { converged = true; break; }
  }
  px = u;
  py = v;
  return converged;
}

// one pass of the loop body of align_1d (feature_alignment.cpp:103-204) by the L lanes of a unit
template <int L>
__device__ __forceinline__ int align_1d_gl_iter(const DevImage& cur_img, double dir0, double dir1, const unsigned char* pwb,
                                                bool affine_est_offset, bool affine_est_gain, const float (&Hinv)[16], float& u,
                                                float& v, float& mean_diff, float& alpha, int& n_it, int sub)
{
  constexpr int kHalfPatchSize = 4, kPatchSize = 8, ref_step = 10, R = 8 / L;
  const float min_update_squared = (float)(0.03 * 0.03);
  const int cur_step = cur_img.pitch;
  const int u_r = (int)floorf(u);
  const int v_r = (int)floorf(v);
  if (u_r < kHalfPatchSize || v_r < kHalfPatchSize || u_r >= cur_img.w - kHalfPatchSize || v_r >= cur_img.h - kHalfPatchSize)
    return ALIGN_STOPPED;
  if (u != u || v != v) return ALIGN_NAN;
  ++n_it;
  const float subpix_x = u - u_r;
  const float subpix_y = v - v_r;
  const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
  const float wTR = (float)(subpix_x * (1.0 - subpix_y));
  const float wBL = (float)((1.0 - subpix_x) * subpix_y);
  const float wBR = subpix_x * subpix_y;
  float t[3][8 * R];
  {
    const uint8_t* it = cur_img.data + (ptrdiff_t)(v_r + sub * R - kHalfPatchSize) * cur_step + u_r - kHalfPatchSize;
    unsigned rows[R + 1][9];
#pragma unroll
    for (int r = 0; r <= R; ++r) load_row9_u8(it + (ptrdiff_t)r * cur_step, rows[r]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const unsigned char* rp = pwb + (sub * R + r + 1) * ref_step + 1;
#pragma unroll
      for (int x = 0; x < kPatchSize; ++x, ++rp) {
        const float gdx = (float)rp[1] - (float)rp[-1];
        const float gdy = (float)rp[ref_step] - (float)rp[-ref_step];
        const float ref_dv = (float)(0.5f * (dir0 * gdx + dir1 * gdy));
        const float cur_intensity = wTL * rows[r][x] + wTR * rows[r][x + 1] + wBL * rows[r + 1][x] + wBR * rows[r + 1][x + 1];
        const float res = cur_intensity - alpha * rp[0] + mean_diff;
        t[0][r * 8 + x] = res * ref_dv;
        t[1][r * 8 + x] = affine_est_offset ? res : 0.0f;
        t[2][r * 8 + x] = affine_est_gain ? (-1) * res * rp[0] : 0.0f;
      }
    }
  }
  float Jres[3] = { 0, 0, 0 };
  if (affine_est_gain) gl_chain_sub<3, L>(Jres, t, sub);
  else {
    float J2[2] = { 0, 0 };
    float t2[2][8 * R];
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
      for (int x = 0; x < 8 * R; ++x) t2[a2][x] = t[a2][x];
    gl_chain_sub<2, L>(J2, t2, sub);
    Jres[0] = J2[0]; Jres[1] = J2[1];
  }
  if (!affine_est_offset) Jres[1] = 0.0f;
  if (!affine_est_gain) Jres[2] = 0.0f;
  float update[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) update[r] = (Hinv[r * 3 + 0] * Jres[0] + Hinv[r * 3 + 1] * Jres[1]) + Hinv[r * 3 + 2] * Jres[2];
  u = (float)(u + update[0] * dir0);
  v = (float)(v + update[0] * dir1);
  mean_diff += update[1];
  alpha += update[2];
  return (update[0] * update[0] < min_update_squared) ? ALIGN_CONVERGED : ALIGN_CONTINUE;
}

// align_1d (feature_alignment.cpp:31-209): the Jacobian entries are arbitrary floats here, H goes through the chain too
__device__ bool align_1d_g8(const DevImage& cur_img, double dir0, double dir1, const unsigned char* pwb, int n_iter,
                            bool affine_est_offset, bool affine_est_gain, double& px, double& py, double* h_inv,
                            int& n_it, int sub)
{
  constexpr int kPatchSize = 8, ref_step = 10;
  bool converged = false;
  float H[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  {
    float t[9][8];
    const unsigned char* it = pwb + (sub + 1) * ref_step + 1;
#pragma unroll
    for (int x = 0; x < kPatchSize; ++x, ++it) {
      float J[3];
      const float dx = (float)it[1] - (float)it[-1];
      const float dy = (float)it[ref_step] - (float)it[-ref_step];
      J[0] = (float)(0.5f * (dir0 * dx + dir1 * dy));
      J[1] = affine_est_offset ? 1.0f : 0.0f;
      J[2] = affine_est_gain ? -1.0f * it[0] : 0.0f;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) t[r * 3 + c][x] = -(J[r] * J[c]);   // H += J J^T as H -= (-J J^T)
    }
    gl_chain_sub<9, 8>(H, t, sub);
  }
  if (!affine_est_offset) H[4] = 1.0f;
  if (!affine_est_gain) H[8] = 1.0f;
  if (h_inv) *h_inv = 1.0 / H[0] * kPatchSize * kPatchSize;
  float Hinv[16];
  {
    float Hi[9];
    mat3f_inverse(H, Hi);
#pragma unroll
    for (int k = 0; k < 9; ++k) Hinv[k] = Hi[k];
#pragma unroll
    for (int k = 9; k < 16; ++k) Hinv[k] = 0.0f;
  }
  float mean_diff = 0;
  float alpha = 1.0;
  float u = (float)px;
  float v = (float)py;
  for (int iter = 0; iter < n_iter; ++iter) {
    const int st = align_1d_gl_iter<8>(cur_img, dir0, dir1, pwb, affine_est_offset, affine_est_gain, Hinv, u, v, mean_diff, alpha, n_it, sub);
    if (st == ALIGN_NAN) return false;
    if (st == ALIGN_STOPPED) break;
    if (st == ALIGN_CONVERGED) { converged = true; break; }
  }
  px = u;
  py = v;
  return converged;
}

// H^-1 of a refinement by ONE lane, for the jobs of the packed geometry (the part of align_2d / align_1d before the
// iterations, feature_alignment.cpp:31-101 / 212-298).  H is symmetric and its entries are sums of commutative
// products, so only the upper triangle is accumulated (each entry by the additions of the one-lane code, in its order)
// and mirrored; with a constant Jacobian entry (1 for the offset, 0 for a parameter that is not estimated) the
// products are the other factor or a zero, exactly.
__device__ void refine_prepare_2d(const unsigned char* pwb, bool affine_est_offset, bool affine_est_gain, float (&Hinv)[16])
{
  constexpr int patch_size_ = 8, ref_step = 10;
  float h00 = 0, h01 = 0, h11 = 0, h02 = 0, h12 = 0, h03 = 0, h13 = 0, h23 = 0, h33 = 0;
  for (int y = 0; y < patch_size_; ++y) {
    const unsigned char* p = pwb + (y + 1) * ref_step + 1;
#pragma unroll
    for (int x = 0; x < patch_size_; ++x, ++p) {
      const float J0 = (float)(0.5 * ((int)p[1] - (int)p[-1]));
      const float J1 = (float)(0.5 * ((int)p[ref_step] - (int)p[-ref_step]));
      h00 += J0 * J0; h01 += J0 * J1; h11 += J1 * J1;
      if (affine_est_offset) { h02 += J0; h12 += J1; }
      if (affine_est_gain) {
        const float J3 = (float)(-1.0 * p[0]);
        h03 += J0 * J3; h13 += J1 * J3; h33 += J3 * J3;
        if (affine_est_offset) h23 += J3;
      }
    }
  }
  const float h22 = affine_est_offset ? (float)(patch_size_ * patch_size_) : 1.0f;
  if (!affine_est_gain) h33 = 1.0f;
  const float H[16] = { h00, h01, h02, h03, h01, h11, h12, h13, h02, h12, h22, h23, h03, h13, h23, h33 };
  mat4f_inverse(H, Hinv);
}

__device__ void refine_prepare_1d(double dir0, double dir1, const unsigned char* pwb, bool affine_est_offset, bool affine_est_gain,
                                  float (&Hinv)[16], double& h_inv)
{
  constexpr int kPatchSize = 8, ref_step = 10;
  float h00 = 0, h01 = 0, h02 = 0, h12 = 0, h22 = 0;
  for (int y = 0; y < kPatchSize; ++y) {
    const unsigned char* p = pwb + (y + 1) * ref_step + 1;
#pragma unroll
    for (int x = 0; x < kPatchSize; ++x, ++p) {
      const float dx = (float)p[1] - (float)p[-1];
      const float dy = (float)p[ref_step] - (float)p[-ref_step];
      const float J0 = (float)(0.5f * (dir0 * dx + dir1 * dy));
      h00 += J0 * J0;
      if (affine_est_offset) h01 += J0;
      if (affine_est_gain) {
        const float J2 = -1.0f * p[0];
        h02 += J0 * J2; h22 += J2 * J2;
        if (affine_est_offset) h12 += J2;
      }
    }
  }
  const float h11 = affine_est_offset ? (float)(kPatchSize * kPatchSize) : 1.0f;
  if (!affine_est_gain) h22 = 1.0f;
  const float H[9] = { h00, h01, h02, h01, h11, h12, h02, h12, h22 };
  h_inv = 1.0 / H[0] * kPatchSize * kPatchSize;
  float Hi[9];
  mat3f_inverse(H, Hi);
#pragma unroll
  for (int k = 0; k < 9; ++k) Hinv[k] = Hi[k];
#pragma unroll
  for (int k = 9; k < 16; ++k) Hinv[k] = 0.0f;
}

// ---------------------------------------------------------------------------------------------------------------
// Packed geometry (large batches).  The one-lane-per-seed kernel above spends its vector instructions on waiting
// lanes: a wave runs the epipolar scan for its longest seed, and the 1-D and the 2-D refinement loops one after
// the other, each for its slowest seed (SQ counters, profiles/r02: 568 VALU instructions per seed, of which ~300
// are the seed's own).  Here a 256-thread workgroup takes 256 seeds through three phases:
//   A  one lane per seed: geometry, affine warp (10x10 patch into LDS), epipolar scan with a packed ZMSSD
//      (template rows held as 16 dwords, current rows as unaligned 8-byte loads, v_dot4_u32_u8 sums); a seed that
//      needs the sub-pixel refinement leaves a JOB (its slot) in one of two LDS queues (2-D / 1-D);
//   E  the refinements, eight lanes per job (the row-split code of the eight-lane geometry: lane `sub` owns patch row
//      `sub`, order-dependent sums handed from row to row): every wave takes eight jobs of ONE kind per round, so no
//      wave runs both loops, a round lasts as long as the slowest of 8 (not 64) jobs, and all four SIMDs share the work;
//   F  one lane per seed again: triangulation, tau, filter update, outputs.
// Every seed goes through exactly the arithmetic of the one-lane code (same expressions, same order): results are
// bit-identical to the other two geometries (tests/test_klt_matcher_gpu.py runs all three).
#ifndef SVOH_PK_THREADS
#define SVOH_PK_THREADS 256
#endif
constexpr int kPkThreads = SVOH_PK_THREADS;

__device__ __forceinline__ unsigned udot4(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_udot4(a, b, c, false); }

// the 8x8 inner patch of a 10x10 LDS patch (4-byte aligned slot) as 16 dwords, row-major, 2 per row
__device__ __forceinline__ void pack_template(const unsigned char* pwb, unsigned (&t)[16])
{
  const unsigned* w = reinterpret_cast<const unsigned*>(pwb);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int o = (r + 1) * 10 + 1;          // byte offset of the row's first pixel: 11, 21, ...: o & 3 = 3 or 1
    const int d = o >> 2, sh = o & 3;
    const unsigned w0 = w[d], w1 = w[d + 1], w2 = w[d + 2];
    t[2 * r] = __builtin_amdgcn_alignbyte(w1, w0, sh);
    t[2 * r + 1] = __builtin_amdgcn_alignbyte(w2, w1, sh);
  }
}

// zmssd_score on packed rows: the same integer sums (patch_score.h:264-283)
__device__ __forceinline__ int zmssd_score_packed(const unsigned (&t)[16], int sumA, int sumAA, const uint8_t* cur_patch, int stride)
{
  uint2 c[8];
#pragma unroll
  for (int y = 0; y < 8; ++y) __builtin_memcpy(&c[y], cur_patch + (ptrdiff_t)y * stride, 8);   // unaligned 8-byte loads, all in flight
  unsigned sumB = 0, sumBB = 0, sumAB = 0;
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    sumB = udot4(c[y].x, 0x01010101u, sumB);  sumB = udot4(c[y].y, 0x01010101u, sumB);
    sumBB = udot4(c[y].x, c[y].x, sumBB);     sumBB = udot4(c[y].y, c[y].y, sumBB);
    sumAB = udot4(c[y].x, t[2 * y], sumAB);   sumAB = udot4(c[y].y, t[2 * y + 1], sumAB);
  }
  const int iB = (int)sumB, iBB = (int)sumBB, iAB = (int)sumAB;
  return sumAA - 2 * iAB + iBB - (sumA * sumA - 2 * sumA * iB + iB * iB) / 64;
}

// warp_affine (patch_warp.cpp:112-156) with the in-image test made on the four corner samples: the sample
// coordinates are (a00*fx + a01*fy) + prx evaluated in float, monotone in fx and in fy (every float operation is),
// so floor() takes its extremes over the 10x10 grid at the corners; a non-finite coefficient makes all four
// corner values non-finite (fx, fy != 0 there), which the test rejects like the per-sample test does.
__device__ bool warp_affine_packed(const double A_cur_ref[4], const DevImage& img_ref, double pxr, double pyr, int level_ref,
                                   int search_level, unsigned char* patch)
{
  constexpr int halfpatch_size = 5;
  double Ai[4];
  mat2d_inverse(A_cur_ref, Ai);
  const float s = (float)(1 << search_level);
  const float a00 = (float)Ai[0] * s, a10 = (float)Ai[1] * s, a01 = (float)Ai[2] * s, a11 = (float)Ai[3] * s;
  if (a00 != a00) return false;
  const float prx = (float)pxr / (float)(1 << level_ref);
  const float pry = (float)pyr / (float)(1 << level_ref);
  const int stride = img_ref.pitch;
  bool inside = true;
#pragma unroll
  for (int cy = 0; cy < 2; ++cy)
#pragma unroll
    for (int cx = 0; cx < 2; ++cx) {
      const float fx = cx ? (float)(halfpatch_size - 1) : (float)(-halfpatch_size);
      const float fy = cy ? (float)(halfpatch_size - 1) : (float)(-halfpatch_size);
      const float pxx = (a00 * fx + a01 * fy) + prx;
      const float pxy = (a10 * fx + a11 * fy) + pry;
      const int xi = (int)floorf(pxx);
      const int yi = (int)floorf(pxy);
      inside = inside && pxx == pxx && pxy == pxy && !(xi < 0 || yi < 0 || xi >= img_ref.w - 1 || yi >= img_ref.h - 1);
    }
  if (!inside) return false;
  for (int y = -halfpatch_size; y < halfpatch_size; ++y) {
    unsigned short t0[10], t1[10];
    float sx[10], sy[10];
#pragma unroll
    for (int x = -halfpatch_size; x < halfpatch_size; ++x) {
      const int k = x + halfpatch_size;
      const float fx = (float)x, fy = (float)y;
      const float pxx = (a00 * fx + a01 * fy) + prx;
      const float pxy = (a10 * fx + a11 * fy) + pry;
      const int xi = (int)floorf(pxx);
      const int yi = (int)floorf(pxy);
      sx[k] = pxx - xi;
      sy[k] = pxy - yi;
      const uint8_t* ptr = img_ref.data + (ptrdiff_t)yi * stride + xi;
      __builtin_memcpy(&t0[k], ptr, 2);            // the two taps of a row in one (unaligned) 2-byte load
      __builtin_memcpy(&t1[k], ptr + stride, 2);
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const float subpix_x = sx[k], subpix_y = sy[k];
      const unsigned t00 = t0[k] & 255u, t10 = t0[k] >> 8, t01 = t1[k] & 255u, t11 = t1[k] >> 8;
      const float w00 = (1.0f - subpix_x) * (1.0f - subpix_y);
      const float w01 = (1.0f - subpix_x) * subpix_y;
      const float w10 = subpix_x * (1.0f - subpix_y);
      const float w11 = 1.0f - w00 - w01 - w10;
      patch[(y + halfpatch_size) * 10 + k] = (unsigned char)(w00 * t00 + w01 * t01 + w10 * t10 + w11 * t11);
    }
  }
  return true;
}

__device__ __forceinline__ Rigid T_cur_ref_of(const DevFrameView& ref, const DevFrameView& cur)
{
  return mul(cur.T_f_w, inverse(ref.T_f_w));
}

// matcher.cpp:31-141
template <bool G8 = false>
__device__ int find_match_direct(MatcherState& m, const svoh_matcher_options& opt, const DevFrameView& ref_frame,
                                 const DevFrameView& cur_frame, double pxr, double pyr, const Vec3& f_ref, double gx,
                                 double gy, int level, int type, double ref_depth, double& pcx, double& pcy,
                                 const double* landmark = nullptr /* one lane per unit only: the pixelwise warp */)
{
  constexpr int kHalfPatchSize = 4, kPatchSize = 8;
  const int pxi0 = (int)pxr / (1 << level), pxi1 = (int)pyr / (1 << level);
  const int boundary = kHalfPatchSize + 2;
  if (pxi0 < boundary || pxi1 < boundary || pxi0 >= (int)(ref_frame.cam.width / (1 << level)) - boundary ||
      pxi1 >= (int)(ref_frame.cam.height / (1 << level)) - boundary)
    return SVOH_MATCH_FAIL_VISIBILITY;
  const Rigid T_cur_ref = T_cur_ref_of(ref_frame, cur_frame);
  get_warp_matrix_affine(ref_frame.cam, cur_frame.cam, pxr, pyr, f_ref, ref_depth, T_cur_ref, level, m.A);
  m.search_level = get_best_search_level(m.A, ref_frame.n_levels - 1);
  ++m.n_warp;
  bool warped;
  if constexpr (G8) warped = warp_affine_g8(m.A, ref_frame.lv[level], pxr, pyr, level, m.search_level, m.pwb, m.sub);
  else if (landmark) warped = warp_pixelwise(cur_frame, ref_frame, pxr, pyr, Vec3{ landmark[0], landmark[1], landmark[2] }, level,
                                             m.search_level, m.pwb);
  else warped = warp_affine(m.A, ref_frame.lv[level], pxr, pyr, level, m.search_level, m.pwb);
  if (!warped) return SVOH_MATCH_FAIL_WARP;
  double sx = pcx / (1 << m.search_level), sy = pcy / (1 << m.search_level);
  const double sx0 = sx, sy0 = sy;
  bool ok;
  if (is_edgelet(type)) {
    double d0 = m.A[0] * gx + m.A[2] * gy, d1 = m.A[1] * gx + m.A[3] * gy;
    normalize2(d0, d1);
    ok = G8 ? align_1d_g8(cur_frame.lv[m.search_level], d0, d1, m.pwb, opt.align_max_iter, opt.affine_est_offset != 0,
                          opt.affine_est_gain != 0, sx, sy, &m.h_inv, m.n_align_it, m.sub)
            : align_1d(cur_frame.lv[m.search_level], d0, d1, m.pwb, opt.align_max_iter, opt.affine_est_offset != 0,
                       opt.affine_est_gain != 0, sx, sy, &m.h_inv, m.n_align_it);
  } else {
    ok = G8 ? align_2d_g8(cur_frame.lv[m.search_level], m.pwb, opt.align_max_iter, opt.affine_est_offset != 0,
                          opt.affine_est_gain != 0, sx, sy, m.n_align_it, m.sub)
            : align_2d(cur_frame.lv[m.search_level], m.pwb, opt.align_max_iter, opt.affine_est_offset != 0,
                       opt.affine_est_gain != 0, sx, sy, m.n_align_it);
  }
  if (!ok) return SVOH_MATCH_FAIL_ALIGNMENT;
  const double dx = sx - sx0, dy = sy - sy0;
  if (sqrt(dx * dx + dy * dy) > opt.max_patch_diff_ratio * kPatchSize) return SVOH_MATCH_FAIL_TOO_FAR;
  pcx = sx * (1 << m.search_level);
  pcy = sy * (1 << m.search_level);
  m.px_cur[0] = pcx; m.px_cur[1] = pcy;
  m.f_cur = back_project3(cur_frame.cam, pcx, pcy);
  normalize3(m.f_cur);
  return SVOH_MATCH_SUCCESS;
}

__device__ __forceinline__ bool is_patch_within_image(const DevFrameView& frame, int px, int py, int patch_level)
{
  constexpr int kPatchSize = 8;
  return !(px < kPatchSize || py < kPatchSize || px >= ((int)(frame.cam.width / (1 << patch_level)) - kPatchSize) ||
           py >= ((int)(frame.cam.height / (1 << patch_level)) - kPatchSize));
}

// G: 0 = one lane per unit, 1 = eight lanes per unit, 2 = packed (one lane per unit in this phase, packed template)
template <int G8>
__device__ __forceinline__ bool update_zmssd(const DevFrameView& frame, int px, int py, int patch_level,
                                             const MatcherState& m, int sumA, int sumAA, int& zmssd_best)
{
  const DevImage& im = frame.lv[patch_level];
  const uint8_t* cur_patch_ptr = im.data + (ptrdiff_t)(py - 4) * im.pitch + (px - 4);
  int z;
  if constexpr (G8 == 1) z = zmssd_score_g8(m.pwb, sumA, sumAA, cur_patch_ptr, im.pitch, m.sub);
  else if constexpr (G8 == 2) z = zmssd_score_packed(m.tpl, sumA, sumAA, cur_patch_ptr, im.pitch);
  else z = zmssd_score(m.pwb, sumA, sumAA, cur_patch_ptr, im.pitch);
  if (z < zmssd_best) { zmssd_best = z; return true; }
  return false;
}

// matcher.cpp:340-413
template <int G8>
__device__ void scan_epipolar_unit_plane(MatcherState& m, const svoh_matcher_options& opt, const DevFrameView& frame,
                                         const Vec3& A, const Vec3& B, const Vec3& C, int patch_level, int sumA, int sumAA,
                                         double& bx, double& by, int& zmssd_best)
{
  size_t n_steps = (size_t)(m.epi_length_pyramid / 0.7);
  double step0 = (A.x / A.z - B.x / B.z) / n_steps, step1 = (A.y / A.z - B.y / B.z) / n_steps;
  if (n_steps > (size_t)opt.max_epi_search_steps) n_steps = (size_t)opt.max_epi_search_steps;
  const double uvC0 = C.x / C.z, uvC1 = C.y / C.z;
  double uv0 = uvC0, uv1 = uvC1;
  double best0 = uv0, best1 = uv1;
  bool forward = true;
  int last0 = 0, last1 = 0;
  if constexpr (G8 == 1) {
    // Eight steps at a time, as in the unit-sphere scan below.  A step's position is the centre plus a number of
    // SEQUENTIAL additions of the step vector (the reference's `uv += step`), restarted from the centre with the negated
    // step when the walk turns round; where it turns depends on the positions only.  Lane `sub` repeats the additions up
    // to its own step, projects that one position, and the group replays the loop's control flow over the eight pixels.
    const DevImage& im = frame.lv[patch_level];
    const int o = (m.sub + 1) * 10 + 1;   // this lane's template row (see zmssd_score_g8)
    const unsigned* w = reinterpret_cast<const unsigned*>(m.pwb) + (o >> 2);
    const unsigned w0 = w[0], w1 = w[1], w2 = w[2];
    const unsigned tr0 = __builtin_amdgcn_alignbyte(w1, w0, (unsigned)(o & 3)), tr1 = __builtin_amdgcn_alignbyte(w2, w1, (unsigned)(o & 3));
    const double half = n_steps * 0.5;
    size_t i0 = 0;
    bool stop = false;
    while (!stop && i0 < n_steps) {
      // uv0, uv1: position of step i0; mine / after: position of step i0 + sub / i0 + 8
      double mine0 = uv0, mine1 = uv1, after0 = uv0, after1 = uv1;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        after0 += step0; after1 += step1;
        if (k < m.sub) { mine0 = after0; mine1 = after1; }
      }
      int pxi0 = 0, pxi1 = 0;
      bool within = false;
      if (i0 + (size_t)m.sub < n_steps) {
        double px, py;
        const Vec3 p3 = { mine0, mine1, 1.0 };
        project3(frame.cam, p3, px, py);
        pxi0 = (int)(px / (1 << patch_level) + 0.5); pxi1 = (int)(py / (1 << patch_level) + 0.5);
        within = is_patch_within_image(frame, pxi0, pxi1, patch_level);
      }
      unsigned scored = 0;
      int turn_after = -1;      // the walk turns round after step i0 + turn_after ...
      bool turn_jump = false;   // ... by the out-of-image rule (the index jumps to the middle) rather than past the middle
      int q0[8], q1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        q0[j] = __shfl(pxi0, j, 8); q1[j] = __shfl(pxi1, j, 8);
        const bool in_j = __shfl((int)within, j, 8) != 0;
        if (stop || turn_after >= 0 || i0 + j >= n_steps) continue;
        if (q0[j] == last0 && q1[j] == last1) continue;
        last0 = q0[j]; last1 = q1[j];
        if (!in_j) {
          if (forward) { turn_after = j; turn_jump = true; }
          else stop = true;
          continue;
        }
        scored |= 1u << j;
        if (forward && (double)(i0 + j) > half) turn_after = j;
      }
      uint2 c[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (scored & (1u << j))
          __builtin_memcpy(&c[j], im.data + (ptrdiff_t)(q1[j] - 4 + m.sub) * im.pitch + (q0[j] - 4), 8);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (scored & (1u << j)) {
          unsigned sumB = __builtin_amdgcn_udot4(c[j].x, 0x01010101u, 0u, false);
          sumB = __builtin_amdgcn_udot4(c[j].y, 0x01010101u, sumB, false);
          unsigned sumBB = __builtin_amdgcn_udot4(c[j].x, c[j].x, 0u, false);
          sumBB = __builtin_amdgcn_udot4(c[j].y, c[j].y, sumBB, false);
          unsigned sumAB = __builtin_amdgcn_udot4(c[j].x, tr0, 0u, false);
          sumAB = __builtin_amdgcn_udot4(c[j].y, tr1, sumAB, false);
          const int iB = g8_sum((int)sumB), iBB = g8_sum((int)sumBB), iAB = g8_sum((int)sumAB);
          const int z = sumAA - 2 * iAB + iBB - (sumA * sumA - 2 * sumA * iB + iB * iB) / 64;
          ++m.n_zmssd;
          if (z < zmssd_best) {
            zmssd_best = z;
            best0 = __hiloint2double(__shfl(__double2hiint(mine0), j, 8), __shfl(__double2loint(mine0), j, 8));
            best1 = __hiloint2double(__shfl(__double2hiint(mine1), j, 8), __shfl(__double2loint(mine1), j, 8));
          }
        }
      if (stop) break;
      if (turn_after >= 0) {
        // `i = n_steps * 0.5; continue` or the flip after a scored step: either way the loop's own increment follows
        i0 = (turn_jump ? (size_t)half : i0 + (size_t)turn_after) + 1;
        step0 = -step0; step1 = -step1;
        uv0 = uvC0 + step0; uv1 = uvC1 + step1;
        forward = false;
      } else {
        i0 += 8;
        uv0 = after0; uv1 = after1;
      }
    }
    const Vec3 p3 = { best0, best1, 1.0 };
    project3(frame.cam, p3, bx, by);
    return;
  }
  for (size_t i = 0; i < n_steps; ++i, uv0 += step0, uv1 += step1) {
    double px, py;
    const Vec3 p3 = { uv0, uv1, 1.0 };
    project3(frame.cam, p3, px, py);
    const int pxi0 = (int)(px / (1 << patch_level) + 0.5), pxi1 = (int)(py / (1 << patch_level) + 0.5);
    if (pxi0 == last0 && pxi1 == last1) continue;
    last0 = pxi0; last1 = pxi1;
    if (!is_patch_within_image(frame, pxi0, pxi1, patch_level)) {
      if (forward) {
        i = (size_t)(n_steps * 0.5);
        step0 = -step0; step1 = -step1;
        uv0 = uvC0; uv1 = uvC1;
        forward = false;
        continue;
      } else
        break;
    }
    ++m.n_zmssd;
    if (update_zmssd<G8>(frame, pxi0, pxi1, patch_level, m, sumA, sumAA, zmssd_best)) { best0 = uv0; best1 = uv1; }
    if (forward && i > n_steps * 0.5) {
      step0 = -step0; step1 = -step1;
      uv0 = uvC0; uv1 = uvC1;
      forward = false;
    }
  }
  const Vec3 p3 = { best0, best1, 1.0 };
  project3(frame.cam, p3, bx, by);
}

// Eigen AngleAxis::toRotationMatrix() * v
__device__ Vec3 angle_axis_rotate(const Vec3& axis, double angle, const Vec3& v)
{
  double s, c;
  sincos(angle, &s, &c);   // one argument reduction for both (the values are those of sin() and cos())
  const double sa0 = s * axis.x, sa1 = s * axis.y, sa2 = s * axis.z;
  const double c0 = (1.0 - c) * axis.x, c1 = (1.0 - c) * axis.y, c2 = (1.0 - c) * axis.z;
  double R[9];
  double tmp;
  tmp = c0 * axis.y; R[1] = tmp - sa2; R[3] = tmp + sa2;
  tmp = c0 * axis.z; R[2] = tmp + sa1; R[6] = tmp - sa1;
  tmp = c1 * axis.z; R[5] = tmp - sa0; R[7] = tmp + sa0;
  R[0] = c0 * axis.x + c; R[4] = c1 * axis.y + c; R[8] = c2 * axis.z + c;
  Vec3 o;
  o.x = (R[0] * v.x + R[1] * v.y) + R[2] * v.z;
  o.y = (R[3] * v.x + R[4] * v.y) + R[5] * v.z;
  o.z = (R[6] * v.x + R[7] * v.y) + R[8] * v.z;
  return o;
}

// matcher.cpp:415-488
template <int G8>
__device__ void scan_epipolar_unit_sphere(MatcherState& m, const svoh_matcher_options& opt, const DevFrameView& frame,
                                          const Vec3& A, const Vec3& B, const Vec3& C, int patch_level, int sumA, int sumAA,
                                          double& bx, double& by, int& zmssd_best)
{
  size_t n_steps = (size_t)(m.epi_length_pyramid / 0.7);
  n_steps = n_steps > (size_t)opt.max_epi_search_steps ? (size_t)opt.max_epi_search_steps : n_steps;
  const size_t half_steps = n_steps / 2;
  Vec3 f_A = A, f_B = B, f_C = C;
  normalize3(f_A); normalize3(f_B); normalize3(f_C);
  const double step = acos((f_A.x * f_B.x + f_A.y * f_B.y) + f_A.z * f_B.z) / n_steps;
  Vec3 axis = { f_B.y * f_A.z - f_B.z * f_A.y, f_B.z * f_A.x - f_B.x * f_A.z, f_B.x * f_A.y - f_B.y * f_A.x };
  normalize3(axis);
  Vec3 f = f_C, f_best = f_C;
  int last0 = 0, last1 = 0;
  if constexpr (G8 == 1) {
    // Eight steps at a time: a step's pixel depends on its index only, not on any score, so lane `sub` computes the
    // position of step i0 + sub (the rotation and the projection are the expensive part of a step, and all eight lanes
    // used to compute the same one), the group then walks the eight positions in order with the reference's rules --
    // skip a pixel equal to the one before, jump to the second half or stop when the patch leaves the image -- which
    // fixes WHICH steps are scored before any score exists; the rows of all of them are requested together (one round
    // trip to memory per eight steps instead of one per step) and scored in step order with the same `<`.
    const DevImage& im = frame.lv[patch_level];
    const int o = (m.sub + 1) * 10 + 1;   // this lane's template row (see zmssd_score_g8)
    const unsigned* w = reinterpret_cast<const unsigned*>(m.pwb) + (o >> 2);
    const unsigned w0 = w[0], w1 = w[1], w2 = w[2];
    const unsigned tr0 = __builtin_amdgcn_alignbyte(w1, w0, (unsigned)(o & 3)), tr1 = __builtin_amdgcn_alignbyte(w2, w1, (unsigned)(o & 3));
    long long best_i = -1;
    size_t i0 = 0;
    while (i0 < n_steps) {
      const size_t i = i0 + (size_t)m.sub;
      int pxi0 = 0, pxi1 = 0;
      bool within = false;
      if (i < n_steps) {
        const double angle = i < half_steps ? i * step : (i - half_steps) * (-step);
        const Vec3 fi = angle_axis_rotate(axis, angle, f_C);
        double px, py;
        project3(frame.cam, fi, px, py);
        pxi0 = (int)(px / (1 << patch_level) + 0.5); pxi1 = (int)(py / (1 << patch_level) + 0.5);
        within = is_patch_within_image(frame, pxi0, pxi1, patch_level);
      }
      unsigned scored = 0;      // bit j: step i0 + j gets a score
      long long jump = -1;      // >= 0: the next block starts there
      bool stop = false;
      int q0[8], q1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        q0[j] = __shfl(pxi0, j, 8); q1[j] = __shfl(pxi1, j, 8);
        const bool in_j = __shfl((int)within, j, 8) != 0;
        if (stop || jump >= 0 || i0 + j >= n_steps) continue;
        if (q0[j] == last0 && q1[j] == last1) continue;
        last0 = q0[j]; last1 = q1[j];
        if (!in_j) {
          if (i0 + j < half_steps) jump = (long long)half_steps + 1;
          else stop = true;
          continue;
        }
        scored |= 1u << j;
      }
      uint2 c[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (scored & (1u << j))
          __builtin_memcpy(&c[j], im.data + (ptrdiff_t)(q1[j] - 4 + m.sub) * im.pitch + (q0[j] - 4), 8);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (scored & (1u << j)) {
          unsigned sumB = __builtin_amdgcn_udot4(c[j].x, 0x01010101u, 0u, false);
          sumB = __builtin_amdgcn_udot4(c[j].y, 0x01010101u, sumB, false);
          unsigned sumBB = __builtin_amdgcn_udot4(c[j].x, c[j].x, 0u, false);
          sumBB = __builtin_amdgcn_udot4(c[j].y, c[j].y, sumBB, false);
          unsigned sumAB = __builtin_amdgcn_udot4(c[j].x, tr0, 0u, false);
          sumAB = __builtin_amdgcn_udot4(c[j].y, tr1, sumAB, false);
          const int iB = g8_sum((int)sumB), iBB = g8_sum((int)sumBB), iAB = g8_sum((int)sumAB);
          const int z = sumAA - 2 * iAB + iBB - (sumA * sumA - 2 * sumA * iB + iB * iB) / 64;
          ++m.n_zmssd;
          if (z < zmssd_best) { zmssd_best = z; best_i = (long long)(i0 + j); }
        }
      if (stop) break;
      i0 = jump >= 0 ? (size_t)jump : i0 + 8;
    }
    if (best_i >= 0) {
      const size_t bi = (size_t)best_i;
      f_best = angle_axis_rotate(axis, bi < half_steps ? bi * step : (bi - half_steps) * (-step), f_C);
    }
    project3(frame.cam, f_best, bx, by);
    return;
  }
  for (size_t i = 0; i < n_steps; i++) {
    double angle;
    if (i < half_steps) angle = i * step;
    else angle = (i - half_steps) * (-step);
    f = angle_axis_rotate(axis, angle, f_C);
    double px, py;
    project3(frame.cam, f, px, py);
    const int pxi0 = (int)(px / (1 << patch_level) + 0.5), pxi1 = (int)(py / (1 << patch_level) + 0.5);
    if (pxi0 == last0 && pxi1 == last1) continue;
    last0 = pxi0; last1 = pxi1;
    if (!is_patch_within_image(frame, pxi0, pxi1, patch_level)) {
      if (i < half_steps) { i = half_steps; continue; }
      else break;
    }
    ++m.n_zmssd;
    if (update_zmssd<G8>(frame, pxi0, pxi1, patch_level, m, sumA, sumAA, zmssd_best)) f_best = f;
  }
  project3(frame.cam, f_best, bx, by);
}

// ---- one wave per unit: 64 steps of the epipolar scan at a time (geometry 3) -------------------------------------------
// The stereo seam searches up to 500 steps per feature (stereo_triangulation.cpp:93) and a launch is as slow as its
// slowest feature: with eight lanes per feature that is 63 dependent rounds of (positions -> rows -> scores).  A step's
// pixel depends on its index only, so a whole wave takes 64 steps per round: lane l computes the position of step
// i0 + l, the loop's control flow over the 64 pixels is three ballots (a pixel equal to its predecessor's is skipped,
// the first patch that leaves the image ends the round: jump to the second half, turn round, or stop), every scored
// lane evaluates its OWN step's ZMSSD with the packed template (eight unaligned 8-byte rows, v_dot4), and the winner is
// the smallest score, the earliest step among equals -- the sequential loop's `if (z < best)` in step order.
// Same visit order, same scores, same first minimum, same counters: bit-identical to the other geometries.
__device__ __forceinline__ int wave_min_i32(int v)
{
  const int big = 0x7fffffff;
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x141, 0xF, 0xF, false));   // row_half_mirror
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x140, 0xF, 0xF, false));   // row_mirror
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2, 3
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// what one round of 64 steps decides: which lanes are scored, where the first leaving patch is, the last pixel seen
struct W64Round { bool scored; int e_out; int n_proc; };
__device__ __forceinline__ W64Round w64_round(bool valid, bool within, int pxi0, int pxi1, int& last0, int& last1, int lane)
{
  int p0 = __shfl_up(pxi0, 1), p1 = __shfl_up(pxi1, 1);
  if (lane == 0) { p0 = last0; p1 = last1; }
  const bool dup = valid && pxi0 == p0 && pxi1 == p1;
  const unsigned long long m_valid = __ballot(valid), m_out = __ballot(valid && !dup && !within);
  W64Round r;
  r.e_out = m_out ? __ffsll((long long)m_out) - 1 : 64;
  r.scored = valid && !dup && within && lane < r.e_out;
  r.n_proc = r.e_out < 64 ? r.e_out + 1 : __popcll(m_valid);
  return r;
}

__device__ void scan_epipolar_unit_sphere_w64(MatcherState& m, const svoh_matcher_options& opt, const DevFrameView& frame,
                                              const Vec3& A, const Vec3& B, const Vec3& C, int patch_level, int sumA, int sumAA,
                                              double& bx, double& by, int& zmssd_best)
{
  size_t n_steps = (size_t)(m.epi_length_pyramid / 0.7);
  n_steps = n_steps > (size_t)opt.max_epi_search_steps ? (size_t)opt.max_epi_search_steps : n_steps;
  const size_t half_steps = n_steps / 2;
  Vec3 f_A = A, f_B = B, f_C = C;
  normalize3(f_A); normalize3(f_B); normalize3(f_C);
  const double step = acos((f_A.x * f_B.x + f_A.y * f_B.y) + f_A.z * f_B.z) / n_steps;
  Vec3 axis = { f_B.y * f_A.z - f_B.z * f_A.y, f_B.z * f_A.x - f_B.x * f_A.z, f_B.x * f_A.y - f_B.y * f_A.x };
  normalize3(axis);
  Vec3 f_best = f_C;
  const int lane = (int)(threadIdx.x & 63);
  const DevImage& im = frame.lv[patch_level];
  int last0 = 0, last1 = 0;
  long long best_i = -1;
  size_t i0 = 0;
  while (i0 < n_steps) {
    const size_t i = i0 + (size_t)lane;
    const bool valid = i < n_steps;
    int pxi0 = 0, pxi1 = 0;
    bool within = false;
    if (valid) {
      const double angle = i < half_steps ? i * step : (i - half_steps) * (-step);
      const Vec3 fi = angle_axis_rotate(axis, angle, f_C);
      double px, py;
      project3(frame.cam, fi, px, py);
      pxi0 = (int)(px / (1 << patch_level) + 0.5); pxi1 = (int)(py / (1 << patch_level) + 0.5);
      within = is_patch_within_image(frame, pxi0, pxi1, patch_level);
    }
    const W64Round r = w64_round(valid, within, pxi0, pxi1, last0, last1, lane);
    int z = 0x7fffffff;
    if (r.scored) z = zmssd_score_packed(m.tpl, sumA, sumAA, im.data + (ptrdiff_t)(pxi1 - 4) * im.pitch + (pxi0 - 4), im.pitch);
    m.n_zmssd += __popcll(__ballot(r.scored));
    const int zmin = wave_min_i32(z);
    if (zmin < zmssd_best) {
      zmssd_best = zmin;
      best_i = (long long)i0 + (__ffsll((long long)__ballot(r.scored && z == zmin)) - 1);
    }
    last0 = __builtin_amdgcn_readlane(pxi0, r.n_proc - 1); last1 = __builtin_amdgcn_readlane(pxi1, r.n_proc - 1);
    if (r.e_out < 64) {
      if (i0 + (size_t)r.e_out < half_steps) i0 = half_steps + 1;   // `i = half_steps; continue` and the loop's increment
      else break;
    } else {
      i0 += 64;
    }
  }
  if (best_i >= 0) {
    const size_t bi = (size_t)best_i;
    f_best = angle_axis_rotate(axis, bi < half_steps ? bi * step : (bi - half_steps) * (-step), f_C);
  }
  project3(frame.cam, f_best, bx, by);
}

__device__ void scan_epipolar_unit_plane_w64(MatcherState& m, const svoh_matcher_options& opt, const DevFrameView& frame,
                                             const Vec3& A, const Vec3& B, const Vec3& C, int patch_level, int sumA, int sumAA,
                                             double& bx, double& by, int& zmssd_best)
{
  size_t n_steps = (size_t)(m.epi_length_pyramid / 0.7);
  double step0 = (A.x / A.z - B.x / B.z) / n_steps, step1 = (A.y / A.z - B.y / B.z) / n_steps;
  if (n_steps > (size_t)opt.max_epi_search_steps) n_steps = (size_t)opt.max_epi_search_steps;
  const double uvC0 = C.x / C.z, uvC1 = C.y / C.z;
  double uv0 = uvC0, uv1 = uvC1;       // position of step i0
  double best0 = uv0, best1 = uv1;
  bool forward = true;
  const int lane = (int)(threadIdx.x & 63);
  const DevImage& im = frame.lv[patch_level];
  const double half = n_steps * 0.5;
  int last0 = 0, last1 = 0;
  size_t i0 = 0;
  while (i0 < n_steps) {
    // a step's position is the round's start plus SEQUENTIAL additions of the step vector (the reference's `uv += step`)
    double mine0 = uv0, mine1 = uv1;
    for (int k = 0; k < 63; ++k)
      if (k < lane) { mine0 += step0; mine1 += step1; }
    const bool valid = i0 + (size_t)lane < n_steps;
    int pxi0 = 0, pxi1 = 0;
    bool within = false;
    if (valid) {
      double px, py;
      const Vec3 p3 = { mine0, mine1, 1.0 };
      project3(frame.cam, p3, px, py);
      pxi0 = (int)(px / (1 << patch_level) + 0.5); pxi1 = (int)(py / (1 << patch_level) + 0.5);
      within = is_patch_within_image(frame, pxi0, pxi1, patch_level);
    }
    W64Round r = w64_round(valid, within, pxi0, pxi1, last0, last1, lane);
    // walking forward, the first scored step past the middle turns the walk round behind itself
    const unsigned long long m_turn = __ballot(r.scored && forward && (double)(i0 + (size_t)lane) > half);
    const int e_turn = m_turn ? __ffsll((long long)m_turn) - 1 : 64;
    if (e_turn < r.e_out) { r.scored = r.scored && lane <= e_turn; r.n_proc = e_turn + 1; }
    int z = 0x7fffffff;
    if (r.scored) z = zmssd_score_packed(m.tpl, sumA, sumAA, im.data + (ptrdiff_t)(pxi1 - 4) * im.pitch + (pxi0 - 4), im.pitch);
    m.n_zmssd += __popcll(__ballot(r.scored));
    const int zmin = wave_min_i32(z);
    if (zmin < zmssd_best) {
      zmssd_best = zmin;
      const int bl = __ffsll((long long)__ballot(r.scored && z == zmin)) - 1;
      best0 = readlane_f64(mine0, bl); best1 = readlane_f64(mine1, bl);
    }
    last0 = __builtin_amdgcn_readlane(pxi0, r.n_proc - 1); last1 = __builtin_amdgcn_readlane(pxi1, r.n_proc - 1);
    if (e_turn < r.e_out) {                 // past the middle: `step = -step; uv = uv_C`, then the loop's own increment
      i0 = i0 + (size_t)e_turn + 1;
      step0 = -step0; step1 = -step1;
      uv0 = uvC0 + step0; uv1 = uvC1 + step1;
      forward = false;
    } else if (r.e_out < 64) {
      if (!forward) break;
      i0 = (size_t)half + 1;                // `i = n_steps * 0.5; ...; continue` and the loop's increment
      step0 = -step0; step1 = -step1;
      uv0 = uvC0 + step0; uv1 = uvC1 + step1;
      forward = false;
    } else {
      i0 += 64;
      uv0 = readlane_f64(mine0, 63) + step0; uv1 = readlane_f64(mine1, 63) + step1;
    }
  }
  const Vec3 p3 = { best0, best1, 1.0 };
  project3(frame.cam, p3, bx, by);
}

// matcher.cpp:492-505
__device__ int depth_from_triangulation(const Rigid& T_search_ref, const Vec3& f_ref, const Vec3& f_cur, double& depth)
{
  const Vec3 a = rotate(T_search_ref.q, f_ref);
  const Vec3& b = f_cur;
  const double AtA[4] = { (a.x * a.x + a.y * a.y) + a.z * a.z, (b.x * a.x + b.y * a.y) + b.z * a.z,
                          (a.x * b.x + a.y * b.y) + a.z * b.z, (b.x * b.x + b.y * b.y) + b.z * b.z };
  if (AtA[0] * AtA[3] - AtA[1] * AtA[2] < 0.000001) return SVOH_MATCH_FAIL_TRIANGULATION;
  double inv[4];
  mat2d_inverse(AtA, inv);
  const double m0 = (-inv[0]) * a.x + (-inv[2]) * b.x;
  const double m1 = (-inv[0]) * a.y + (-inv[2]) * b.y;
  const double m2 = (-inv[0]) * a.z + (-inv[2]) * b.z;
  const double d0 = (m0 * T_search_ref.t.x + m1 * T_search_ref.t.y) + m2 * T_search_ref.t.z;
  depth = fabs(d0);
  return SVOH_MATCH_SUCCESS;
}

// matcher.cpp:157-241
// Matcher::findEpipolarMatchDirect (matcher.cpp:157-241) up to the sub-pixel refinement: epipolar segment, warp,
// ZMSSD scan.  Returns a final result code, or kMatchRefinePending (m.px_cur = start of the refinement at level 0,
// m.epi_dir = its 1-D direction) / kMatchTriangulatePending (no refinement wanted).
constexpr int kMatchRefinePending = -1000, kMatchTriangulatePending = -1001;
template <int G8 = 0>
__device__ int epipolar_match_search(MatcherState& m, const svoh_matcher_options& opt, const DevFrameView& ref_frame,
                                     const DevFrameView& cur_frame, const Rigid& T_cur_ref, double pxr, double pyr,
                                     const Vec3& f_ref, double gx, double gy, int level, int type,
                                     double d_estimate_inv, double d_min_inv, double d_max_inv)
{
  constexpr int ZMSSD_THRESHOLD = 2000 * 64;
  int zmssd_best = ZMSSD_THRESHOLD;
  const Vec3 Rf = rotate(T_cur_ref.q, f_ref);
  const Vec3 A = { Rf.x + T_cur_ref.t.x * d_min_inv, Rf.y + T_cur_ref.t.y * d_min_inv, Rf.z + T_cur_ref.t.z * d_min_inv };
  const Vec3 B = { Rf.x + T_cur_ref.t.x * d_max_inv, Rf.y + T_cur_ref.t.y * d_max_inv, Rf.z + T_cur_ref.t.z * d_max_inv };
  double pAx, pAy, pBx, pBy;
  project3(cur_frame.cam, A, pAx, pAy);
  project3(cur_frame.cam, B, pBx, pBy);
  m.epi_image[0] = pAx - pBx;
  m.epi_image[1] = pAy - pBy;
  get_warp_matrix_affine(ref_frame.cam, cur_frame.cam, pxr, pyr, f_ref, 1.0 / fmax(0.000001, d_estimate_inv), T_cur_ref,
                         level, m.A);
  m.reject = false;
  if (is_edgelet(type) && opt.epi_search_edgelet_filtering) {
    double g0 = m.A[0] * gx + m.A[2] * gy, g1 = m.A[1] * gx + m.A[3] * gy;
    normalize2(g0, g1);
    double e0 = m.epi_image[0], e1 = m.epi_image[1];
    normalize2(e0, e1);
    const double cosangle = fabs(g0 * e0 + g1 * e1);
    if (cosangle < opt.epi_search_edgelet_max_angle) {
      m.reject = true;
      return SVOH_MATCH_FAIL_ANGLE;
    }
  }
  m.search_level = get_best_search_level(m.A, ref_frame.n_levels - 1);
  m.epi_length_pyramid = sqrt(m.epi_image[0] * m.epi_image[0] + m.epi_image[1] * m.epi_image[1]) / (1 << m.search_level);
  double ed0 = m.epi_image[0], ed1 = m.epi_image[1];
  normalize2(ed0, ed1);
  ++m.n_warp;
  SVOH_MSTAMP(m, 0);
  bool warp_ok;
  if constexpr (G8 == 1) warp_ok = warp_affine_g8(m.A, ref_frame.lv[level], pxr, pyr, level, m.search_level, m.pwb, m.sub);
  else if constexpr (G8 == 3) warp_ok = warp_affine_w64(m.A, ref_frame.lv[level], pxr, pyr, level, m.search_level, m.pwb);
  else if constexpr (G8 == 2) warp_ok = warp_affine_packed(m.A, ref_frame.lv[level], pxr, pyr, level, m.search_level, m.pwb);
  else warp_ok = warp_affine(m.A, ref_frame.lv[level], pxr, pyr, level, m.search_level, m.pwb);
  SVOH_MSTAMP(m, 1);
  if (!warp_ok) return SVOH_MATCH_FAIL_WARP;

  // The reference refines the match at two places (matcher.cpp:205-215 for an epipolar segment shorter than two
  // pixels, :229-235 after the scan).  Here both kinds of lanes meet at ONE refinement (find_epipolar_match_direct): inlined
  // twice, the alignment loops of the two places run one after the other in every wave that holds both kinds.
  bool refine;
  if (m.epi_length_pyramid < 2.0) {
    m.px_cur[0] = (pAx + pBx) / 2.0;
    m.px_cur[1] = (pAy + pBy) / 2.0;
    refine = true;
  } else {
    // PatchScore constructor (patch_score.h:80-92)
    int sumA = 0, sumAA = 0;
    if constexpr (G8 == 1) patch_sums_g8(m.pwb, m.sub, sumA, sumAA);
    else if constexpr (G8 == 2 || G8 == 3) {
      pack_template(m.pwb, m.tpl);
      unsigned a1 = 0, a2 = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) { a1 = udot4(m.tpl[k], 0x01010101u, a1); a2 = udot4(m.tpl[k], m.tpl[k], a2); }
      sumA = (int)a1; sumAA = (int)a2;
    } else for (int r = 0; r < 64; ++r) { const int n = patch_at(m.pwb, r); sumA += n; sumAA += n * n; }
    const Vec3 C = { Rf.x + T_cur_ref.t.x * d_estimate_inv, Rf.y + T_cur_ref.t.y * d_estimate_inv,
                     Rf.z + T_cur_ref.t.z * d_estimate_inv };
    if constexpr (G8 == 3) {
      if (opt.scan_on_unit_sphere)
        scan_epipolar_unit_sphere_w64(m, opt, cur_frame, A, B, C, m.search_level, sumA, sumAA, m.px_cur[0], m.px_cur[1], zmssd_best);
      else
        scan_epipolar_unit_plane_w64(m, opt, cur_frame, A, B, C, m.search_level, sumA, sumAA, m.px_cur[0], m.px_cur[1], zmssd_best);
    } else if (opt.scan_on_unit_sphere)
      scan_epipolar_unit_sphere<G8>(m, opt, cur_frame, A, B, C, m.search_level, sumA, sumAA, m.px_cur[0], m.px_cur[1], zmssd_best);
    else
      scan_epipolar_unit_plane<G8>(m, opt, cur_frame, A, B, C, m.search_level, sumA, sumAA, m.px_cur[0], m.px_cur[1], zmssd_best);
    if (!(zmssd_best < ZMSSD_THRESHOLD)) {
      SVOH_MSTAMP(m, 2);
      return SVOH_MATCH_FAIL_SCORE;
    }
    refine = opt.subpix_refinement != 0;
  }
  SVOH_MSTAMP(m, 2);
  m.epi_dir[0] = ed0; m.epi_dir[1] = ed1;
  return refine ? kMatchRefinePending : kMatchTriangulatePending;
}

// The rest of Matcher::findEpipolarMatchDirect once the refinement (if any) has run: matcher.cpp:262-289 tail,
// :216-219 / :236-239.  aligned = return value of align1D / align2D, (sx, sy) = its result at the search level.
__device__ int epipolar_match_finish(MatcherState& m, const DevFrameView& cur_frame, const Rigid& T_cur_ref, const Vec3& f_ref,
                                     bool refined, bool aligned, double sx, double sy, double& depth)
{
  if (refined) {
    if (!aligned) return SVOH_MATCH_FAIL_ALIGNMENT;
    m.px_cur[0] = sx * (1 << m.search_level);
    m.px_cur[1] = sy * (1 << m.search_level);
  }
  m.f_cur = back_project3(cur_frame.cam, m.px_cur[0], m.px_cur[1]);
  normalize3(m.f_cur);
  return depth_from_triangulation(T_cur_ref, f_ref, m.f_cur, depth);
}

// Matcher::findEpipolarMatchDirect in one piece.  G8: 0 one lane per unit, 1 eight lanes per unit, 3 one wave per unit
// (the search of geometry 3, the refinement by its eight-lane groups, all of which compute the same)
template <int G8 = 0>
__device__ int find_epipolar_match_direct(MatcherState& m, const svoh_matcher_options& opt, const DevFrameView& ref_frame,
                                          const DevFrameView& cur_frame, const Rigid& T_cur_ref, double pxr, double pyr,
                                          const Vec3& f_ref, double gx, double gy, int level, int type,
                                          double d_estimate_inv, double d_min_inv, double d_max_inv, double& depth)
{
  const int st = epipolar_match_search<G8>(m, opt, ref_frame, cur_frame, T_cur_ref, pxr, pyr, f_ref, gx, gy, level, type,
                                                   d_estimate_inv, d_min_inv, d_max_inv);
  if (st != kMatchRefinePending && st != kMatchTriangulatePending) return st;
  bool aligned = false;
  double sx = 0.0, sy = 0.0;
  if (st == kMatchRefinePending) {
    sx = m.px_cur[0] / (1 << m.search_level); sy = m.px_cur[1] / (1 << m.search_level);
    if (m.align_1d)
      aligned = G8 ? align_1d_g8(cur_frame.lv[m.search_level], m.epi_dir[0], m.epi_dir[1], m.pwb, opt.align_max_iter,
                                 opt.affine_est_offset != 0, opt.affine_est_gain != 0, sx, sy, &m.h_inv, m.n_align_it, m.sub)
                   : align_1d(cur_frame.lv[m.search_level], m.epi_dir[0], m.epi_dir[1], m.pwb, opt.align_max_iter,
                              opt.affine_est_offset != 0, opt.affine_est_gain != 0, sx, sy, &m.h_inv, m.n_align_it);
    else
      aligned = G8 ? align_2d_g8(cur_frame.lv[m.search_level], m.pwb, opt.align_max_iter, opt.affine_est_offset != 0,
                                 opt.affine_est_gain != 0, sx, sy, m.n_align_it, m.sub)
                   : align_2d(cur_frame.lv[m.search_level], m.pwb, opt.align_max_iter, opt.affine_est_offset != 0,
                              opt.affine_est_gain != 0, sx, sy, m.n_align_it);
    SVOH_MSTAMP(m, 3);
  }
  return epipolar_match_finish(m, cur_frame, T_cur_ref, f_ref, st == kMatchRefinePending, aligned, sx, sy, depth);
}

// math_utils.h:186-194
__device__ __forceinline__ double norm_pdf(double x, double mean, double sigma)
{
  double exponent = x - mean;
  exponent *= -exponent;
  exponent /= 2 * sigma * sigma;
  double result = exp(exponent);
  result /= sigma * sqrt(2 * 3.14159265358979323846);
  return result;
}

// depth_filter.cpp:501-552
__device__ bool update_filter_vogiatzis(double z, double tau2, double mu_range, double* st)
{
  double mu = st[0], sigma2 = st[1], a = st[2], b = st[3];
  const double norm_scale = sqrt(sigma2 + tau2);
  if (norm_scale != norm_scale) return false;
  const double oldsigma2 = sigma2;
  const double s2 = 1.0 / (1.0 / sigma2 + 1.0 / tau2);
  const double mm = s2 * (mu / sigma2 + z / tau2);
  const double uniform_x = 1.0 / mu_range;
  double C1 = a / (a + b) * norm_pdf(z, mu, norm_scale);
  double C2 = b / (a + b) * uniform_x;
  const double normalization_constant = C1 + C2;
  C1 /= normalization_constant;
  C2 /= normalization_constant;
  const double f = C1 * (a + 1.0) / (a + b + 1.0) + C2 * a / (a + b + 1.0);
  const double e = C1 * (a + 1.0) * (a + 2.0) / ((a + b + 1.0) * (a + b + 2.0)) +
                   C2 * a * (a + 1.0) / ((a + b + 1.0) * (a + b + 2.0));
  const double mu_new = C1 * mm + C2 * mu;
  sigma2 = C1 * (s2 + mm * mm) + C2 * (sigma2 + mu * mu) - mu_new * mu_new;
  mu = mu_new;
  a = (e - f) / (f - e / f);
  b = a * (1.0 - f) / f;
  bool ok = true;
  if (sigma2 < 0.0) sigma2 = oldsigma2;
  if (mu < 0.0) { mu = 1.0; ok = false; }
  st[0] = mu; st[1] = sigma2; st[2] = a; st[3] = b;
  return ok;
}

// depth_filter.cpp:554-578
__device__ bool update_filter_gaussian(double z, double tau2, double* st)
{
  const double norm_scale = sqrt(st[1] + tau2);
  if (norm_scale != norm_scale) return false;
  const double denom = (st[1] + tau2);
  st[0] = (st[1] * z + tau2 * st[0]) / denom;
  st[1] = st[1] * tau2 / denom;
  return true;
}

// depth_filter.cpp:580-596
__device__ double compute_tau(const Rigid& T_ref_cur, const Vec3& f, double z, double px_error_angle)
{
  const Vec3& t = T_ref_cur.t;
  const Vec3 a = { f.x * z - t.x, f.y * z - t.y, f.z * z - t.z };
  const double t_norm = sqrt((t.x * t.x + t.y * t.y) + t.z * t.z);
  const double a_norm = sqrt((a.x * a.x + a.y * a.y) + a.z * a.z);
  const double alpha = acos(((f.x * t.x + f.y * t.y) + f.z * t.z) / t_norm);
  const double beta = acos(((a.x * -t.x + a.y * -t.y) + a.z * -t.z) / (t_norm * a_norm));
  const double beta_plus = beta + px_error_angle;
  const double gamma_plus = 3.14159265358979323846 - alpha - beta_plus;
  const double z_plus = t_norm * sin(beta_plus) / sin(gamma_plus);
  return (z_plus - z);
}

__device__ __forceinline__ void flush_counters(unsigned int* c, int i, const MatcherState& m, int updated)
{
#ifdef SVOH_SEED_STAMPS
  reinterpret_cast<uint4*>(c)[i] = make_uint4((unsigned)(m.t[0] >> 4), (unsigned)(m.t[1] >> 4), (unsigned)(m.t[2] >> 4), (unsigned)(m.t[3] >> 4));
#else
  reinterpret_cast<uint4*>(c)[i] = make_uint4((unsigned)m.n_warp, (unsigned)m.n_zmssd, (unsigned)m.n_align_it, (unsigned)updated);
#endif
}

// G8: eight lanes per unit (small batches, see g8_sum): unit = threadIdx.x / 8, the lane's patch row = threadIdx.x % 8
template <bool G8>
__device__ __forceinline__ void match_direct_body(const MatcherArgs& a, int block, unsigned char* s_pwb)
{
  const int unit = G8 ? (int)(threadIdx.x >> 3) : (int)threadIdx.x;
  const int i = block * (G8 ? 8 : 64) + unit;
  if (i >= a.n) return;
  if (!feature_indices_ok(a, i)) {
    if (G8 && (threadIdx.x & 7) != 0) return;
    a.result[i] = SVOH_MATCH_NOT_RUN;
    // a unit that is not run reads back as zeros in its optional outputs (the launch does not clear the output block)
    if (a.f_cur) { a.f_cur[3 * i] = 0.0; a.f_cur[3 * i + 1] = 0.0; a.f_cur[3 * i + 2] = 0.0; }
    if (a.search_level) a.search_level[i] = 0;
    if (a.h_inv) a.h_inv[i] = 0.0;
    if (a.A_cur_ref) for (int k = 0; k < 4; ++k) a.A_cur_ref[4 * i + k] = 0.0;
    reinterpret_cast<uint4*>(a.unit_counts)[i] = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  MatcherState m;
  m.pwb = s_pwb + unit * kPwbStride;
  m.sub = G8 ? (int)(threadIdx.x & 7) : 0;
  m.h_inv = 0.0; m.search_level = 0; m.reject = false; m.align_1d = false;
  m.n_warp = 0; m.n_zmssd = 0; m.n_align_it = 0;
  m.A[0] = m.A[1] = m.A[2] = m.A[3] = 0.0;
  m.f_cur = { 0.0, 0.0, 0.0 };
  const DevFrameView& ref = a.ref_frames[a.ref_frame_idx[i]];
  const Vec3 f = { a.f[3 * i], a.f[3 * i + 1], a.f[3 * i + 2] };
  double pcx = a.px_cur[2 * i], pcy = a.px_cur[2 * i + 1];
  const DevFrameView& curf = a.cur_frame[a.cur_frame_idx ? a.cur_frame_idx[i] : 0];
  const int r = find_match_direct<G8>(m, a.mopt, ref, curf, a.px[2 * i], a.px[2 * i + 1], f, a.grad[2 * i],
                                      a.grad[2 * i + 1], a.level[i], a.type[i], a.depth[i], pcx, pcy,
                                      (!G8 && a.landmark_xyz) ? &a.landmark_xyz[3 * i] : nullptr);
  if (G8 && m.sub != 0) return;   // the eight lanes hold the same results: one of them reports
  a.result[i] = r;
  a.px_cur[2 * i] = pcx; a.px_cur[2 * i + 1] = pcy;
  if (a.f_cur) { a.f_cur[3 * i] = m.f_cur.x; a.f_cur[3 * i + 1] = m.f_cur.y; a.f_cur[3 * i + 2] = m.f_cur.z; }
  if (a.search_level) a.search_level[i] = m.search_level;
  if (a.h_inv) a.h_inv[i] = m.h_inv;
  if (a.A_cur_ref) for (int k = 0; k < 4; ++k) a.A_cur_ref[4 * i + k] = m.A[k];
  flush_counters(a.unit_counts, i, m, 0);
}

template <bool G8>
__global__ __launch_bounds__(64) void match_direct_kernel(const MatcherArgs a)
{
  __shared__ __attribute__((aligned(16))) unsigned char s_pwb[64 * kPwbStride];
  match_direct_body<G8>(a, (int)blockIdx.x, s_pwb);
}

// DepthFilter::updateSeeds (depth_filter.cpp:200-233) + depth_filter_utils::updateSeed (:367-499)
// G8: 0 one lane per unit, 1 eight lanes per unit, 3 one wave per unit (scan 64 steps at a time)
template <int G8>
__device__ __forceinline__ void update_seeds_body(const MatcherArgs& a, int block, unsigned char* s_pwb)
{
  const int unit = G8 == 1 ? (int)(threadIdx.x >> 3) : (G8 == 3 ? 0 : (int)threadIdx.x);
  const int i = block * (G8 == 1 ? 8 : (G8 == 3 ? 1 : 64)) + unit;
  if (i >= a.n) return;
  const bool reporter = G8 == 0 || (G8 == 1 && (threadIdx.x & 7) == 0) || (G8 == 3 && (threadIdx.x & 63) == 0);
  if (reporter) {
    a.success[i] = 0;
    if (a.result) a.result[i] = SVOH_MATCH_NOT_RUN;
    // units that are not run: zeros in the optional outputs (the launch does not clear the output block; a unit that
    // does run overwrites them below)
    if (a.px_cur) { a.px_cur[2 * i] = 0.0; a.px_cur[2 * i + 1] = 0.0; }
    if (a.f_cur) { a.f_cur[3 * i] = 0.0; a.f_cur[3 * i + 1] = 0.0; a.f_cur[3 * i + 2] = 0.0; }
    if (a.search_level) a.search_level[i] = 0;
    if (a.A_cur_ref) for (int k = 0; k < 4; ++k) a.A_cur_ref[4 * i + k] = 0.0;
    reinterpret_cast<uint4*>(a.unit_counts)[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (!feature_indices_ok(a, i)) return;
  const int type = a.type[i];
  if (!(type < 6)) return;  // isSeed
  double cur_thresh = a.dopt.seed_convergence_sigma2_thresh;
  if (type == SVOH_FT_MAPPOINT_SEED || type == SVOH_FT_MAPPOINT_SEED_CONVERGED)
    cur_thresh = a.dopt.mappoint_convergence_sigma2_thresh;
  const DevFrameView& ref = a.ref_frames[a.ref_frame_idx[i]];
  const DevFrameView& cur = a.cur_frame[a.cur_frame_idx ? a.cur_frame_idx[i] : 0];
  if (cur.id == ref.id) return;
  if (type == SVOH_FT_OUTLIER) return;
  if ((type == SVOH_FT_CORNER_SEED_CONVERGED || type == SVOH_FT_EDGELET_SEED_CONVERGED ||
       type == SVOH_FT_MAPPOINT_SEED_CONVERGED) && a.dopt.check_convergence)
    return;
  double st[4] = { a.state[4 * i], a.state[4 * i + 1], a.state[4 * i + 2], a.state[4 * i + 3] };
  const Vec3 f = { a.f[3 * i], a.f[3 * i + 1], a.f[3 * i + 2] };
  const Rigid T_cur_ref = T_cur_ref_of(ref, cur);
  if (a.dopt.check_visibility) {
    const double depth = 1.0 / st[0];
    const Vec3 p = { depth * f.x, depth * f.y, depth * f.z };
    double px, py;
    project3(cur.cam, transform(T_cur_ref, p), px, py);
    if (!(px >= 0.0 && py >= 0.0 && px < (double)cur.cam.width && py < (double)cur.cam.height)) return;
    const int pxi0 = (int)px, pxi1 = (int)py;
    const int boundary = 9;
    if (!(pxi0 >= boundary && pxi1 >= boundary && pxi0 < cur.cam.width - boundary && pxi1 < cur.cam.height - boundary)) return;
  }
  MatcherState m;
  m.pwb = s_pwb + unit * kPwbStride;
  m.sub = G8 ? (int)(threadIdx.x & 7) : 0;
  m.h_inv = 0.0; m.search_level = 0; m.reject = false;
  m.n_warp = 0; m.n_zmssd = 0; m.n_align_it = 0;
  m.align_1d = (type == SVOH_FT_EDGELET_SEED || type == SVOH_FT_EDGELET_SEED_CONVERGED);
  m.A[0] = m.A[1] = m.A[2] = m.A[3] = 0.0;
  m.px_cur[0] = m.px_cur[1] = 0.0;
  m.f_cur = { 0.0, 0.0, 0.0 };
#ifdef SVOH_SEED_STAMPS
  m.t[0] = m.t[1] = m.t[2] = m.t[3] = 0; m.tlast = clock64();
#endif
  double depth = 0.0;
  const double inv_min = st[0] + sqrt(st[1]);
  const double inv_max = fmax(st[0] - sqrt(st[1]), 0.00000001);
  const int res = find_epipolar_match_direct<G8>(m, a.mopt, ref, cur, T_cur_ref, a.px[2 * i], a.px[2 * i + 1], f, a.grad[2 * i],
                                                 a.grad[2 * i + 1], a.level[i], type, st[0], inv_min, inv_max, depth);
  if (!reporter) return;   // the lanes of a unit hold the same results: one of them reports and updates the seed
  if (a.result) a.result[i] = res;
  // matcher state as reprojector_utils::matchCandidate reads it after updateSeed (reprojector.cpp:403-413, 473-476)
  if (a.px_cur) { a.px_cur[2 * i] = m.px_cur[0]; a.px_cur[2 * i + 1] = m.px_cur[1]; }
  if (a.f_cur) { a.f_cur[3 * i] = m.f_cur.x; a.f_cur[3 * i + 1] = m.f_cur.y; a.f_cur[3 * i + 2] = m.f_cur.z; }
  if (a.search_level) a.search_level[i] = m.search_level;
  if (a.A_cur_ref) for (int k = 0; k < 4; ++k) a.A_cur_ref[4 * i + k] = m.A[k];
  flush_counters(a.unit_counts, i, m, res == SVOH_MATCH_SUCCESS ? 1 : 0);
  if (res != SVOH_MATCH_SUCCESS) {
    if (!m.reject) a.state[4 * i + 3] = st[3] + 1;  // seed::increaseOutlierProbability
    return;
  }
  const double depth_sigma = compute_tau(inverse(T_cur_ref), f, depth, a.dopt.px_error_angle);
  const double z = 1.0 / depth;
  const double sg = 0.5 * (1.0 / fmax(0.000000000001, depth - depth_sigma) - 1.0 / (depth + depth_sigma));
  const double tau2 = sg * sg;
  bool ok;
  if (a.dopt.use_vogiatzis_update) ok = update_filter_vogiatzis(z, tau2, ref.seed_mu_range, st);
  else ok = update_filter_gaussian(z, tau2, st);
  a.state[4 * i] = st[0]; a.state[4 * i + 1] = st[1]; a.state[4 * i + 2] = st[2]; a.state[4 * i + 3] = st[3];
  if (!ok) { a.type[i] = SVOH_FT_OUTLIER; return; }
  const double thresh = ref.seed_mu_range / cur_thresh;
  if (st[1] < thresh * thresh) {
    if (type == SVOH_FT_CORNER_SEED) a.type[i] = SVOH_FT_CORNER_SEED_CONVERGED;
    else if (type == SVOH_FT_EDGELET_SEED) a.type[i] = SVOH_FT_EDGELET_SEED_CONVERGED;
    else if (type == SVOH_FT_MAPPOINT_SEED) a.type[i] = SVOH_FT_MAPPOINT_SEED_CONVERGED;
  }
  a.success[i] = 1;
#ifdef SVOH_SEED_STAMPS
  SVOH_MSTAMP(m, 0);
  flush_counters(a.unit_counts, i, m, 1);
#endif
}

template <int G8>
__global__ __launch_bounds__(64) void update_seeds_kernel(const MatcherArgs a)
{
  __shared__ __attribute__((aligned(16))) unsigned char s_pwb[64 * kPwbStride];
  update_seeds_body<G8>(a, (int)blockIdx.x, s_pwb);
}

// The direct matches and the seed updates of one reprojection (a deferred section, svoh_matcher_begin_deferred) in ONE
// launch: workgroups [0, n_blocks_direct) run the direct matcher's body, the rest the seed update's.  Both are small
// (a few hundred to a few thousand units on a 256-CU device): one after the other they cost two kernel latencies.
template <bool G8>
__global__ __launch_bounds__(64) void match_mixed_kernel(const MatcherArgs ad, const MatcherArgs as, int n_blocks_direct)
{
  __shared__ __attribute__((aligned(16))) unsigned char s_pwb[64 * kPwbStride];
  if ((int)blockIdx.x < n_blocks_direct) match_direct_body<G8>(ad, (int)blockIdx.x, s_pwb);
  else update_seeds_body<G8 ? 1 : 0>(as, (int)blockIdx.x - n_blocks_direct, s_pwb);
}

// ---- spatial binning of a large seed batch -------------------------------------------------------------------------
// Seeds arrive in the detector's order (by score): neighbours in the batch lie anywhere in the image, so every seed
// pulls its own ~20 cache lines (10-12 rows of the reference patch, as many of the current image) through L2 and
// nothing is shared.  A counting sort by (reference frame, 128x32-pixel tile of the reference pixel) makes the
// seeds of a workgroup neighbours in both images (the frames of an update are close): their rows share lines.
// Results do not depend on the processing order; every seed still writes its own entries.
constexpr size_t kBinMaxKeys = (size_t)1 << 20;

__device__ __forceinline__ unsigned seed_bin_key(const MatcherArgs& a, int i, int tiles_x, int tiles_y)
{
  int ri = a.ref_frame_idx[i];
  if ((unsigned)ri >= (unsigned)a.n_ref_frames) ri = 0;   // such a seed is not run; it only needs some place in the order
  const double x = a.px[2 * i], y = a.px[2 * i + 1];
  int tx = (x >= 0.0 && x < 1e9) ? ((int)x >> kBinShiftX) : 0, ty = (y >= 0.0 && y < 1e9) ? ((int)y >> kBinShiftY) : 0;
  tx = tx < tiles_x ? tx : tiles_x - 1; ty = ty < tiles_y ? ty : tiles_y - 1;
  return ((unsigned)ri * (unsigned)tiles_y + (unsigned)ty) * (unsigned)tiles_x + (unsigned)tx;
}

__global__ __launch_bounds__(256) void seed_bin_count_kernel(const MatcherArgs a, int tiles_x, int tiles_y, unsigned* hist, unsigned* rank)
{
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i < a.n) rank[i] = atomicAdd(&hist[seed_bin_key(a, i, tiles_x, tiles_y)], 1u);
}

// exclusive scan of hist[0 .. n_keys) in place (one workgroup)
__global__ __launch_bounds__(1024) void seed_bin_scan_kernel(unsigned* hist, unsigned n_keys)
{
  __shared__ unsigned s_part[1024];
  const unsigned t = threadIdx.x;
  const unsigned per = (n_keys + 1023u) / 1024u;
  const unsigned lo = t * per, hi = lo + per < n_keys ? lo + per : n_keys;
  unsigned sum = 0;
  for (unsigned k = lo; k < hi; ++k) sum += hist[k];
  s_part[t] = sum;
  __syncthreads();
  for (unsigned o = 1; o < 1024u; o <<= 1) {
    const unsigned v = t >= o ? s_part[t - o] : 0u;
    __syncthreads();
    s_part[t] += v;
    __syncthreads();
  }
  unsigned run = s_part[t] - sum;
  for (unsigned k = lo; k < hi; ++k) { const unsigned c = hist[k]; hist[k] = run; run += c; }
}


// Sorted seed records (binned batches): the binning pass gathers every seed's inputs into ONE 128-byte line at its
// place in the processing order, the packed kernel leaves its results in a 64-byte record next to it, and a last
// pass copies the results back to the caller's arrays in the caller's order (coalesced on that side).  Without it
// every seed of a binned batch would touch ~10 separate lines of the caller's arrays (8 input arrays + outputs).
struct SeedRecIn {    // 128 bytes
  double px[2], f[3], grad[2], state[4];
  int32_t ri, ci, level, type;
  int32_t index;      // the seed's place in the caller's arrays
  int32_t pad[5];
};
struct SeedRecOut {   // 64 bytes
  double state[4];
  unsigned counts[4];
  int32_t result, type, success, pad;
};
static_assert(sizeof(SeedRecIn) == 128 && sizeof(SeedRecOut) == 64, "record sizes");

// SCAN_HERE: `offs` still holds the histogram and every workgroup scans it for itself into LDS (n_keys <= kBinScanHereMaxKeys:
// a few thousand adds per workgroup, all workgroups at once, instead of a one-workgroup kernel in between)
constexpr unsigned kBinScanHereMaxKeys = 8192;
template <bool SCAN_HERE>
__global__ __launch_bounds__(256) void seed_bin_scatter_kernel(const MatcherArgs a, int tiles_x, int tiles_y, const unsigned* offs,
                                                               const unsigned* rank, unsigned* pos_of, SeedRecIn* rec, unsigned n_keys)
{
  __shared__ unsigned s_off[SCAN_HERE ? kBinScanHereMaxKeys : 1];
  __shared__ unsigned s_part[256];
  if constexpr (SCAN_HERE) {
    const unsigned t = threadIdx.x;
    const unsigned per = (n_keys + 255u) / 256u;
    const unsigned lo = t * per, hi = lo + per < n_keys ? lo + per : n_keys;
    unsigned sum = 0;
    for (unsigned k = lo; k < hi; ++k) { const unsigned c = offs[k]; s_off[k] = c; sum += c; }
    s_part[t] = sum;
    __syncthreads();
    for (unsigned o = 1; o < 256u; o <<= 1) {
      const unsigned v = t >= o ? s_part[t - o] : 0u;
      __syncthreads();
      s_part[t] += v;
      __syncthreads();
    }
    unsigned run = s_part[t] - sum;
    for (unsigned k = lo; k < hi; ++k) { const unsigned c = s_off[k]; s_off[k] = run; run += c; }
    __syncthreads();
  }
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= a.n) return;
  const unsigned key = seed_bin_key(a, i, tiles_x, tiles_y);
  const unsigned pos = (SCAN_HERE ? s_off[key] : offs[key]) + rank[i];
  pos_of[i] = pos;
  SeedRecIn r;
  r.px[0] = a.px[2 * i]; r.px[1] = a.px[2 * i + 1];
  r.f[0] = a.f[3 * i]; r.f[1] = a.f[3 * i + 1]; r.f[2] = a.f[3 * i + 2];
  r.grad[0] = a.grad[2 * i]; r.grad[1] = a.grad[2 * i + 1];
  for (int k = 0; k < 4; ++k) r.state[k] = a.state[4 * i + k];
  r.ri = a.ref_frame_idx[i]; r.ci = a.cur_frame_idx ? a.cur_frame_idx[i] : 0; r.level = a.level[i]; r.type = a.type[i];
  r.index = i;
  for (int k = 0; k < 5; ++k) r.pad[k] = 0;
  rec[pos] = r;
}

// results back into the caller's arrays, one thread per seed in the caller's order
// (also leaves the histogram zeroed for the next call: one memset launch less per step)
__global__ __launch_bounds__(256) void seed_unsort_kernel(const MatcherArgs a, const unsigned* pos_of, const SeedRecOut* out,
                                                          unsigned* hist, unsigned n_keys)
{
  for (unsigned k = blockIdx.x * 256 + threadIdx.x; k < n_keys; k += gridDim.x * 256) hist[k] = 0u;
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= a.n) return;
  const SeedRecOut r = out[pos_of[i]];
  for (int k = 0; k < 4; ++k) a.state[4 * i + k] = r.state[k];
  a.type[i] = (uint8_t)r.type;
  a.success[i] = (uint8_t)r.success;
  if (a.result) a.result[i] = r.result;
  reinterpret_cast<uint4*>(a.unit_counts)[i] = make_uint4(r.counts[0], r.counts[1], r.counts[2], r.counts[3]);
}

// DepthFilter::updateSeeds + depth_filter_utils::updateSeed, packed geometry (see "Packed geometry" above)
// Phase E of the packed geometry: the sub-pixel refinements (align1D / align2D) of a 256-unit workgroup as jobs.  Units
// that need one have left their slot in one of two LDS queues (s_q[0]: 2-D, s_q[1]: 1-D) and H^-1 of their patch in
// s_hm (refine_prepare_*, by the unit's own lane in phase A).  A wave is sixteen groups of kPkJobLanes = 4 lanes; a group
// takes a job, runs its iterations (every lane two patch rows, order-dependent sums handed from lane to lane) and takes
// the next one as soon as its own job ends -- no group waits for the slowest job of a round.  A wave works on ONE kind
// at a time and moves to the other queue when its own is empty.
// In: s_u / s_v start position, s_dir direction (1-D), s_stat (cur frame << 8) | search level, the slot's warped patch.
// Out: s_u / s_v result, s_res (iterations << 8) | converged.
constexpr int kPkJobLanes = 4;
__device__ __forceinline__ void packed_refine_phase(const MatcherArgs& a, int tid, const unsigned char* s_pwb, int* s_res, const double* s_dir,
                                                    float* s_u, float* s_v, const int* s_stat, unsigned short (*s_q)[kPkThreads], int* s_qn,
                                                    int* s_qh, const float (*s_hm)[16], bool est_offset, bool est_gain)
{
  constexpr int L = kPkJobLanes;
  const int sub = tid & (L - 1);
  const int max_iter = a.mopt.align_max_iter;
  int kind = (tid >> 6) & 1;          // waves start on different queues
  for (int tried = 0; tried < 2; ++tried, kind ^= 1) {
    const int qn = s_qn[kind];
    int slot = -1, iter = 0, n_it = 0;
    bool dry = false;
    float Hm[16], u = 0.0f, v = 0.0f, mean_diff = 0.0f, alpha = 1.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) Hm[k] = 0.0f;
    double dir0 = 0.0, dir1 = 0.0;
    DevImage img = { nullptr, 0, 0, 0, 0 };
    const unsigned char* pwb = s_pwb;
    // a group runs at most qn jobs of at most max_iter passes (plus one pass per job that takes it): the bound is never
    // reached, it is the exit every wave has whatever the queue holds
    const long long pass_bound = (long long)qn * ((max_iter > 0 ? max_iter : 0) + 1) + 1;
    for (long long pass = 0; pass <= pass_bound; ++pass) {
      if (slot < 0 && !dry) {
        int s2 = -1;
        if (sub == 0) {
          const int j = atomicAdd(&s_qh[kind], 1);
          if (j < qn) s2 = s_q[kind][j];
        }
        s2 = __shfl(s2, 0, L);
        if (s2 >= 0) {
          slot = s2; iter = 0; n_it = 0;
          const int meta = s_stat[slot];
          img = a.cur_frame[meta >> 8].lv[meta & 255];
          pwb = s_pwb + slot * kPwbStride;
#pragma unroll
          for (int k = 0; k < 16; ++k) Hm[k] = s_hm[slot][k];
          u = s_u[slot]; v = s_v[slot];     // align1D / 2D start from the position narrowed to float
          mean_diff = 0.0f; alpha = 1.0f;
          if (kind == 1) { dir0 = s_dir[2 * slot]; dir1 = s_dir[2 * slot + 1]; }
        } else dry = true;
      }
      if (__ballot(slot >= 0) == 0) break;
      if (slot >= 0) {
        int st = ALIGN_STOPPED;
        if (iter < max_iter)
          st = kind == 1 ? align_1d_gl_iter<L>(img, dir0, dir1, pwb, est_offset, est_gain, Hm, u, v, mean_diff, alpha, n_it, sub)
                         : align_2d_gl_iter<L>(img, pwb, est_offset, est_gain, Hm, u, v, mean_diff, alpha, n_it, sub);
        ++iter;
        if (st != ALIGN_CONTINUE || iter >= max_iter) {
          if (sub == 0) {
            if (st != ALIGN_NAN) { s_u[slot] = u; s_v[slot] = v; }   // align1D / 2D hand back px = u, py = v (floats); not on NaN
            s_res[slot] = (n_it << 8) | (st == ALIGN_CONVERGED ? 1 : 0);
          }
          slot = -1;
        }
      }
    }
  }
}

// phase A's hand-over of a unit to phase E: H^-1 (by this lane), start position, direction, queue entry
__device__ __forceinline__ double packed_refine_enqueue(int tid, bool align_1d, double dir0, double dir1, const unsigned char* pwb,
                                                        double sx, double sy, bool est_offset, bool est_gain, float (*s_hm)[16],
                                                        double* s_dir, float* s_u, float* s_v, unsigned short (*s_q)[kPkThreads],
                                                        int* s_qn)
{
  float Hinv[16];
  double h_inv = 0.0;
  if (align_1d) refine_prepare_1d(dir0, dir1, pwb, est_offset, est_gain, Hinv, h_inv);
  else refine_prepare_2d(pwb, est_offset, est_gain, Hinv);
#pragma unroll
  for (int k = 0; k < 16; ++k) s_hm[tid][k] = Hinv[k];
  s_u[tid] = (float)sx;
  s_v[tid] = (float)sy;
  if (align_1d) { s_dir[2 * tid] = dir0; s_dir[2 * tid + 1] = dir1; }
  const int kind = align_1d ? 1 : 0;
  const int j = atomicAdd(&s_qn[kind], 1);
  s_q[kind][j] = (unsigned short)tid;
  return h_inv;
}

// WS: the SVOH_BATCH_WHOLE_SETS form (inputs from the reference frames' tile-ordered resident columns, results in place) as an
// instantiation of its own: as a run-time branch it cost the record form five spilled registers at the three-waves-per-SIMD budget
template <bool WS>
__global__ __launch_bounds__(kPkThreads) __attribute__((amdgpu_waves_per_eu(3))) void update_seeds_packed_kernel(
    const MatcherArgs a, const SeedRecIn* __restrict__ rec_in_, SeedRecOut* __restrict__ rec_out_)
{
  const SeedRecIn* __restrict__ rec_in = WS ? nullptr : rec_in_;
  SeedRecOut* __restrict__ rec_out = WS ? nullptr : rec_out_;
  __shared__ __attribute__((aligned(16))) unsigned char s_pwb[kPkThreads * kPwbStride + 16];
  __shared__ int s_res[kPkThreads];              // out of the refinement: (iterations << 8) | converged
  __shared__ double s_dir[2 * kPkThreads];       // 1-D refinements: the direction
  __shared__ float s_u[kPkThreads], s_v[kPkThreads];   // in: start of the refinement; out: its result
  __shared__ int s_stat[kPkThreads];             // (cur frame << 8) | search level of the slot
  __shared__ unsigned short s_q[2][kPkThreads];  // job queues: slots with a pending 2-D / 1-D refinement
  __shared__ __attribute__((aligned(16))) float s_hm[kPkThreads][16];   // H^-1 of the slot's refinement
  __shared__ int s_qn[2], s_qh[2];
  const int tid = (int)threadIdx.x;
  if (tid < 2) { s_qn[tid] = 0; s_qh[tid] = 0; }
  __syncthreads();
  const int slot_i = (int)blockIdx.x * kPkThreads + tid;
  const bool live = slot_i < a.n;
  const bool est_offset = a.mopt.affine_est_offset != 0, est_gain = a.mopt.affine_est_gain != 0;

#ifdef SVOH_PK_STAMPS
  const long long ts0 = clock64(); long long tsA1 = ts0, tsA2 = ts0, tsE0 = ts0, tsE1 = ts0;
  long long tsg[3] = { 0, 0, 0 };
#endif
  // the seed's inputs: from its sorted record, or straight from the caller's arrays
  int i = slot_i, ri = 0, ci = 0, level = 0, type = 0;
  double pxr = 0.0, pyr = 0.0, gx = 0.0, gy = 0.0;
  int ws_q = 0;                       // whole sets: the slot's place in its reference frame's tile order
  const double* ws_f = nullptr;
  if (WS && live) {
    // slot p of the launch = place q of reference frame ri's tile order = feature perm[q] = unit unit_begin + perm[q] of the caller
    ri = whole_sets_frame_of(a.ref_frames, a.n_ref_frames, slot_i, a.n);
    const DevFrameView& rv = a.ref_frames[ri];
    ws_q = slot_i - rv.unit_begin;
    if (ws_q >= 0 && ws_q < rv.feat_n) {
      i = rv.unit_begin + rv.feat_perm[ws_q];
      pxr = rv.feat_spx[2 * ws_q]; pyr = rv.feat_spx[2 * ws_q + 1]; gx = rv.feat_sgrad[2 * ws_q]; gy = rv.feat_sgrad[2 * ws_q + 1];
      level = rv.feat_slevel[ws_q]; ws_f = rv.feat_sf;
      ci = a.cur_frame_idx ? a.cur_frame_idx[rv.unit_begin] : 0;   // (a set's seeds go into ONE current frame: read at its first unit)
      type = a.type[i];
    } else { ri = -1; }   // (cannot happen: the host checked that the sets' sizes add up to n)
  } else if (!WS && live) {
    if (rec_in) {
      const SeedRecIn& r = rec_in[slot_i];
      pxr = r.px[0]; pyr = r.px[1]; gx = r.grad[0]; gy = r.grad[1];
      ri = r.ri; ci = r.ci; level = r.level; type = r.type; i = r.index;
    } else {
      pxr = a.px[2 * i]; pyr = a.px[2 * i + 1]; gx = a.grad[2 * i]; gy = a.grad[2 * i + 1];
      ri = a.ref_frame_idx[i]; ci = a.cur_frame_idx ? a.cur_frame_idx[i] : 0; level = a.level[i]; type = a.type[i];
    }
  }
  auto load_f = [&]() -> Vec3 {
    if constexpr (WS) return Vec3{ ws_f[3 * ws_q], ws_f[3 * ws_q + 1], ws_f[3 * ws_q + 2] };
    if (rec_in) { const SeedRecIn& r = rec_in[slot_i]; return Vec3{ r.f[0], r.f[1], r.f[2] }; }
    return Vec3{ a.f[3 * i], a.f[3 * i + 1], a.f[3 * i + 2] };
  };
  auto load_state = [&](double (&st)[4]) {
    if (rec_in) { const SeedRecIn& r = rec_in[slot_i]; for (int k = 0; k < 4; ++k) st[k] = r.state[k]; }
    else for (int k = 0; k < 4; ++k) st[k] = a.state[4 * i + k];
  };

  // ---- phase A: one lane per seed, up to the refinement ----
  // code: kMatchRefinePending / kMatchTriangulatePending / a final svoh_match_result / kNotRun
  constexpr int kNotRun = -2000;
  int code = kNotRun;
  int search_level = 0;
  bool reject = false;
  int n_warp = 0, n_zmssd = 0;
  double px_cur0 = 0.0, px_cur1 = 0.0, edir0 = 0.0, edir1 = 0.0;
  bool is_1d = false;
  const bool indices_ok = live && (unsigned)ri < (unsigned)a.n_ref_frames && (unsigned)ci < (unsigned)a.n_cur_frames &&
                          level >= 0 && level < a.ref_frames[(unsigned)ri < (unsigned)a.n_ref_frames ? ri : 0].n_levels;
  if (indices_ok) {
    const DevFrameView& ref = a.ref_frames[ri];
    const DevFrameView& cur = a.cur_frame[ci];
    bool run = type < 6 && cur.id != ref.id && type != SVOH_FT_OUTLIER;
    if ((type == SVOH_FT_CORNER_SEED_CONVERGED || type == SVOH_FT_EDGELET_SEED_CONVERGED ||
         type == SVOH_FT_MAPPOINT_SEED_CONVERGED) && a.dopt.check_convergence)
      run = false;
    if (run) {
      double st[4];
      load_state(st);
      const double st0 = st[0], st1 = st[1];
      const Vec3 f = load_f();
      const Rigid T_cur_ref = T_cur_ref_of(ref, cur);
      if (a.dopt.check_visibility) {
        const double depth = 1.0 / st0;
        const Vec3 p = { depth * f.x, depth * f.y, depth * f.z };
        double px, py;
        project3(cur.cam, transform(T_cur_ref, p), px, py);
        if (!(px >= 0.0 && py >= 0.0 && px < (double)cur.cam.width && py < (double)cur.cam.height)) run = false;
        else {
          const int pxi0 = (int)px, pxi1 = (int)py;
          const int boundary = 9;
          if (!(pxi0 >= boundary && pxi1 >= boundary && pxi0 < cur.cam.width - boundary && pxi1 < cur.cam.height - boundary)) run = false;
        }
      }
      if (run) {
        MatcherState m;
        m.pwb = s_pwb + tid * kPwbStride;
        m.sub = 0;
        m.h_inv = 0.0; m.search_level = 0; m.reject = false;
        m.n_warp = 0; m.n_zmssd = 0; m.n_align_it = 0;
        m.align_1d = (type == SVOH_FT_EDGELET_SEED || type == SVOH_FT_EDGELET_SEED_CONVERGED);
        m.A[0] = m.A[1] = m.A[2] = m.A[3] = 0.0;
        m.px_cur[0] = m.px_cur[1] = 0.0;
        m.f_cur = { 0.0, 0.0, 0.0 };
        m.epi_dir[0] = m.epi_dir[1] = 0.0;
#ifdef SVOH_SEED_STAMPS
        m.t[0] = m.t[1] = m.t[2] = m.t[3] = 0; m.tlast = clock64();
#endif
        const double inv_min = st0 + sqrt(st1);
        const double inv_max = fmax(st0 - sqrt(st1), 0.00000001);
        code = epipolar_match_search<2>(m, a.mopt, ref, cur, T_cur_ref, pxr, pyr, f, gx, gy, level, type, st0, inv_min, inv_max);
        search_level = m.search_level; reject = m.reject;
        n_warp = m.n_warp; n_zmssd = m.n_zmssd;
#if defined(SVOH_PK_STAMPS) && defined(SVOH_SEED_STAMPS)
        tsg[0] = m.t[0]; tsg[1] = m.t[1]; tsg[2] = m.t[2];
#endif
        px_cur0 = m.px_cur[0]; px_cur1 = m.px_cur[1];
        if (a.search_level) a.search_level[i] = m.search_level;
        if (a.A_cur_ref) for (int k = 0; k < 4; ++k) a.A_cur_ref[4 * i + k] = m.A[k];
#ifdef SVOH_PK_STAMPS
        tsA1 = clock64();
#endif
        is_1d = m.align_1d; edir0 = m.epi_dir[0]; edir1 = m.epi_dir[1];
      }
    }
  }
  // every lane leaves (cur frame, search level) of its slot where the lanes of a refinement job find them
  s_stat[tid] = (ci << 8) | search_level;
  // the refinement (feature_alignment.cpp:31-209 / 212-391) is phase E's job
  if (code == kMatchRefinePending)
    packed_refine_enqueue(tid, is_1d, edir0, edir1, s_pwb + tid * kPwbStride, px_cur0 / (1 << search_level),
                          px_cur1 / (1 << search_level), est_offset, est_gain, s_hm, s_dir, s_u, s_v, s_q, s_qn);
#ifdef SVOH_PK_STAMPS
  tsA2 = clock64();
#endif
  __syncthreads();
#ifdef SVOH_PK_STAMPS
  tsE0 = tsE1 = clock64();
#endif

  // ---- phase E: the refinements as jobs of four-lane groups (packed_refine_phase) ----
  packed_refine_phase(a, tid, s_pwb, s_res, s_dir, s_u, s_v, s_stat, s_q, s_qn, s_qh, s_hm, est_offset, est_gain);
#ifdef SVOH_PK_STAMPS
  tsE1 = clock64();
#endif
  __syncthreads();

  // ---- phase F: one lane per seed: triangulation, tau, filter update, outputs ----
  if (!live) return;
  double st[4] = { 0.0, 0.0, 0.0, 0.0 };
  int out_type = type, out_success = 0, out_result = SVOH_MATCH_NOT_RUN;
  unsigned counts[4] = { 0u, 0u, 0u, 0u };
  bool state_changed = false;
  if (rec_out || code != kNotRun) load_state(st);
  if (code != kNotRun) {
    int res = code;
    int n_align_it = 0;
    Vec3 f_cur = { 0.0, 0.0, 0.0 };
    const DevFrameView& ref = a.ref_frames[ri];
    const DevFrameView& cur = a.cur_frame[ci];
    const Vec3 f = load_f();
    const Rigid T_cur_ref = T_cur_ref_of(ref, cur);
    double depth = 0.0;
    if (code == kMatchRefinePending || code == kMatchTriangulatePending) {
      MatcherState m;
      m.search_level = search_level;
      m.px_cur[0] = px_cur0; m.px_cur[1] = px_cur1;
      bool aligned = false;
      double sx = 0.0, sy = 0.0;
      if (code == kMatchRefinePending) {
        const int r = s_res[tid];
        n_align_it = r >> 8;
        aligned = (r & 1) != 0;
        sx = s_u[tid]; sy = s_v[tid];
      }
      res = epipolar_match_finish(m, cur, T_cur_ref, f, code == kMatchRefinePending, aligned, sx, sy, depth);
      px_cur0 = m.px_cur[0]; px_cur1 = m.px_cur[1];
      f_cur = m.f_cur;
    }
    out_result = res;
    if (a.px_cur) { a.px_cur[2 * i] = px_cur0; a.px_cur[2 * i + 1] = px_cur1; }
    if (a.f_cur) { a.f_cur[3 * i] = f_cur.x; a.f_cur[3 * i + 1] = f_cur.y; a.f_cur[3 * i + 2] = f_cur.z; }
    counts[0] = (unsigned)n_warp; counts[1] = (unsigned)n_zmssd; counts[2] = (unsigned)n_align_it;
    counts[3] = res == SVOH_MATCH_SUCCESS ? 1u : 0u;
    if (res != SVOH_MATCH_SUCCESS) {
      if (!reject) { st[3] = st[3] + 1; state_changed = true; }  // seed::increaseOutlierProbability
    } else {
      double cur_thresh = a.dopt.seed_convergence_sigma2_thresh;
      if (type == SVOH_FT_MAPPOINT_SEED || type == SVOH_FT_MAPPOINT_SEED_CONVERGED)
        cur_thresh = a.dopt.mappoint_convergence_sigma2_thresh;
      const double depth_sigma = compute_tau(inverse(T_cur_ref), f, depth, a.dopt.px_error_angle);
      const double z = 1.0 / depth;
      const double sg = 0.5 * (1.0 / fmax(0.000000000001, depth - depth_sigma) - 1.0 / (depth + depth_sigma));
      const double tau2 = sg * sg;
      bool ok;
      if (a.dopt.use_vogiatzis_update) ok = update_filter_vogiatzis(z, tau2, ref.seed_mu_range, st);
      else ok = update_filter_gaussian(z, tau2, st);
      state_changed = true;
      if (!ok) out_type = SVOH_FT_OUTLIER;
      else {
        const double thresh = ref.seed_mu_range / cur_thresh;
        if (st[1] < thresh * thresh) {
          if (type == SVOH_FT_CORNER_SEED) out_type = SVOH_FT_CORNER_SEED_CONVERGED;
          else if (type == SVOH_FT_EDGELET_SEED) out_type = SVOH_FT_EDGELET_SEED_CONVERGED;
          else if (type == SVOH_FT_MAPPOINT_SEED) out_type = SVOH_FT_MAPPOINT_SEED_CONVERGED;
        }
        out_success = 1;
      }
    }
  }
#ifdef SVOH_PK_STAMPS
  {  // diagnostic build: cycles / 16 in search (geometry, warp, scan), refinement set-up, job loop (participating waves), whole kernel
    const long long tsF = clock64();
    counts[0] = (unsigned)((tsA1 - ts0) >> 4); counts[1] = (unsigned)((tsA2 - tsA1) >> 4);
    counts[2] = (unsigned)((tsE1 - tsE0) >> 4); counts[3] = (unsigned)((tsF - ts0) >> 4);
#ifdef SVOH_SEED_STAMPS   // second diagnostic variant: geometry / warp / scan / set-up
    counts[0] = (unsigned)(tsg[0] >> 4); counts[1] = (unsigned)(tsg[1] >> 4); counts[2] = (unsigned)(tsg[2] >> 4);
    counts[3] = (unsigned)((tsA2 - tsA1) >> 4);
#endif
  }
#endif
  if (rec_out) {
    SeedRecOut o;
    for (int k = 0; k < 4; ++k) { o.state[k] = st[k]; o.counts[k] = counts[k]; }
    o.result = out_result; o.type = out_type; o.success = out_success; o.pad = 0;
    rec_out[slot_i] = o;
  } else {
    a.success[i] = (uint8_t)out_success;
    if (a.result) a.result[i] = out_result;
    reinterpret_cast<uint4*>(a.unit_counts)[i] = make_uint4(counts[0], counts[1], counts[2], counts[3]);
    if (state_changed) for (int k = 0; k < 4; ++k) a.state[4 * i + k] = st[k];
    if (out_type != type) a.type[i] = (uint8_t)out_type;
  }
}

// n x Matcher::findEpipolarMatchDirect with an explicit T_cur_ref (matcher.cpp:157-241), align_1d = isEdgelet(type):
// the call StereoTriangulation::compute makes per new feature (stereo_triangulation.cpp:92-104)
template <int G8>
__global__ __launch_bounds__(64) void epipolar_match_kernel(const MatcherArgs a)
{
  __shared__ __attribute__((aligned(16))) unsigned char s_pwb[64 * kPwbStride];
  const int unit = G8 == 1 ? (int)(threadIdx.x >> 3) : (G8 == 3 ? 0 : (int)threadIdx.x);
  const int i = blockIdx.x * (G8 == 1 ? 8 : (G8 == 3 ? 1 : 64)) + unit;
  if (i >= a.n) return;
  const bool reporter = G8 == 0 || (G8 == 1 && (threadIdx.x & 7) == 0) || (G8 == 3 && (threadIdx.x & 63) == 0);
  if (!feature_indices_ok(a, i)) {
    if (reporter) {
      a.result[i] = SVOH_MATCH_NOT_RUN;
      reinterpret_cast<uint4*>(a.unit_counts)[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    return;
  }
  const int ri = a.ref_frame_idx[i], ci = a.cur_frame_idx ? a.cur_frame_idx[i] : 0;
  const DevFrameView& ref = a.ref_frames[ri];
  const DevFrameView& cur = a.cur_frame[ci];
  const Rigid T_cur_ref = a.T_cur_ref ? a.T_cur_ref[(size_t)ri * a.n_cur_frames + ci] : T_cur_ref_of(ref, cur);
  const int type = a.type[i];
  MatcherState m;
  m.pwb = s_pwb + unit * kPwbStride;
  m.sub = G8 ? (int)(threadIdx.x & 7) : 0;
  m.h_inv = 0.0; m.search_level = 0; m.reject = false;
  m.n_warp = 0; m.n_zmssd = 0; m.n_align_it = 0;
  m.align_1d = is_edgelet(type);
  m.A[0] = m.A[1] = m.A[2] = m.A[3] = 0.0;
  m.px_cur[0] = m.px_cur[1] = 0.0;
  m.f_cur = { 0.0, 0.0, 0.0 };
#ifdef SVOH_SEED_STAMPS
  m.t[0] = m.t[1] = m.t[2] = m.t[3] = 0; m.tlast = clock64();
#endif
  const Vec3 f = { a.f[3 * i], a.f[3 * i + 1], a.f[3 * i + 2] };
  const double d_est = a.d_inv ? a.d_inv[3 * i] : a.d_inv_common[0];
  const double d_min = a.d_inv ? a.d_inv[3 * i + 1] : a.d_inv_common[1];
  const double d_max = a.d_inv ? a.d_inv[3 * i + 2] : a.d_inv_common[2];
  double depth = 0.0;
  const int res = find_epipolar_match_direct<G8>(m, a.mopt, ref, cur, T_cur_ref, a.px[2 * i], a.px[2 * i + 1], f, a.grad[2 * i],
                                                 a.grad[2 * i + 1], a.level[i], type, d_est, d_min, d_max, depth);
  if (!reporter) return;
  a.result[i] = res;
  a.depth_out[i] = depth;
  if (a.px_cur) { a.px_cur[2 * i] = m.px_cur[0]; a.px_cur[2 * i + 1] = m.px_cur[1]; }
  if (a.f_cur) { a.f_cur[3 * i] = m.f_cur.x; a.f_cur[3 * i + 1] = m.f_cur.y; a.f_cur[3 * i + 2] = m.f_cur.z; }
  if (a.search_level) a.search_level[i] = m.search_level;
  if (a.h_inv) a.h_inv[i] = m.h_inv;
  if (a.A_cur_ref) for (int k = 0; k < 4; ++k) a.A_cur_ref[4 * i + k] = m.A[k];
  flush_counters(a.unit_counts, i, m, res == SVOH_MATCH_SUCCESS ? 1 : 0);
}


// The packed geometry for the other two matcher entries (large batches of svoh_match_direct_batch /
// svoh_epipolar_match_batch): phase A one lane per unit (geometry, affine warp with the corner test and two-tap loads,
// for the epipolar match the scan with the packed ZMSSD), phase E the refinements as eight-lane jobs
// (packed_refine_phase), phase F one lane per unit (too-far test / triangulation, outputs).  Every unit goes through the
// arithmetic of the one-lane and eight-lane kernels: identical outputs, successes and failures alike.
template <bool DIRECT>
__global__ __launch_bounds__(kPkThreads) __attribute__((amdgpu_waves_per_eu(3))) void match_packed_kernel(const MatcherArgs a)
{
  __shared__ __attribute__((aligned(16))) unsigned char s_pwb[kPkThreads * kPwbStride + 16];
  __shared__ int s_res[kPkThreads];
  __shared__ double s_dir[2 * kPkThreads];
  __shared__ float s_u[kPkThreads], s_v[kPkThreads];
  __shared__ int s_stat[kPkThreads];
  __shared__ unsigned short s_q[2][kPkThreads];
  __shared__ __attribute__((aligned(16))) float s_hm[kPkThreads][16];
  __shared__ int s_qn[2], s_qh[2];
  const int tid = (int)threadIdx.x;
  if (tid < 2) { s_qn[tid] = 0; s_qh[tid] = 0; }
  __syncthreads();
  double h_inv_1d = 0.0, edir0 = 0.0, edir1 = 0.0;
  const int i = (int)blockIdx.x * kPkThreads + tid;
  const bool live = i < a.n;
  const bool est_offset = a.mopt.affine_est_offset != 0, est_gain = a.mopt.affine_est_gain != 0;
  constexpr int kNotRun = -2000;
  int code = kNotRun, search_level = 0, ci = 0, n_warp = 0, n_zmssd = 0;
  double A4[4] = { 0.0, 0.0, 0.0, 0.0 };
  double px0 = 0.0, px1 = 0.0, sx0 = 0.0, sy0 = 0.0;
  bool is_1d = false;
  const bool ok_idx = live && feature_indices_ok(a, i);
  if (ok_idx) {
    const int ri = a.ref_frame_idx[i];
    ci = a.cur_frame_idx ? a.cur_frame_idx[i] : 0;
    const DevFrameView& ref = a.ref_frames[ri];
    const DevFrameView& cur = a.cur_frame[ci];
    const int type = a.type[i], level = a.level[i];
    const Vec3 f = { a.f[3 * i], a.f[3 * i + 1], a.f[3 * i + 2] };
    const double pxr = a.px[2 * i], pyr = a.px[2 * i + 1], gx = a.grad[2 * i], gy = a.grad[2 * i + 1];
    MatcherState m;
    m.pwb = s_pwb + tid * kPwbStride;
    m.sub = 0;
    m.h_inv = 0.0; m.search_level = 0; m.reject = false;
    m.n_warp = 0; m.n_zmssd = 0; m.n_align_it = 0;
    m.align_1d = is_edgelet(type);
    m.A[0] = m.A[1] = m.A[2] = m.A[3] = 0.0;
    m.px_cur[0] = m.px_cur[1] = 0.0;
    m.f_cur = { 0.0, 0.0, 0.0 };
    m.epi_dir[0] = m.epi_dir[1] = 0.0;
#ifdef SVOH_SEED_STAMPS
    m.t[0] = m.t[1] = m.t[2] = m.t[3] = 0; m.tlast = clock64();
#endif
    is_1d = m.align_1d;
    if constexpr (DIRECT) {
      // Matcher::findMatchDirect up to the refinement (matcher.cpp:31-93)
      px0 = a.px_cur[2 * i]; px1 = a.px_cur[2 * i + 1];
      constexpr int kHalfPatchSize = 4;
      const int pxi0 = (int)pxr / (1 << level), pxi1 = (int)pyr / (1 << level);
      const int boundary = kHalfPatchSize + 2;
      if (pxi0 < boundary || pxi1 < boundary || pxi0 >= (int)(ref.cam.width / (1 << level)) - boundary ||
          pxi1 >= (int)(ref.cam.height / (1 << level)) - boundary) {
        code = SVOH_MATCH_FAIL_VISIBILITY;
      } else {
        const Rigid T_cur_ref = T_cur_ref_of(ref, cur);
        get_warp_matrix_affine(ref.cam, cur.cam, pxr, pyr, f, a.depth[i], T_cur_ref, level, m.A);
        m.search_level = get_best_search_level(m.A, ref.n_levels - 1);
        ++m.n_warp;
        if (!warp_affine_packed(m.A, ref.lv[level], pxr, pyr, level, m.search_level, m.pwb)) code = SVOH_MATCH_FAIL_WARP;
        else {
          code = kMatchRefinePending;
          sx0 = px0 / (1 << m.search_level); sy0 = px1 / (1 << m.search_level);
          if (m.align_1d) {
            double d0 = m.A[0] * gx + m.A[2] * gy, d1 = m.A[1] * gx + m.A[3] * gy;
            normalize2(d0, d1);
            m.epi_dir[0] = d0; m.epi_dir[1] = d1;
          }
        }
      }
    } else {
      const Rigid T_cur_ref = a.T_cur_ref ? a.T_cur_ref[(size_t)ri * a.n_cur_frames + ci] : T_cur_ref_of(ref, cur);
      const double d_est = a.d_inv ? a.d_inv[3 * i] : a.d_inv_common[0];
      const double d_min = a.d_inv ? a.d_inv[3 * i + 1] : a.d_inv_common[1];
      const double d_max = a.d_inv ? a.d_inv[3 * i + 2] : a.d_inv_common[2];
      code = epipolar_match_search<2>(m, a.mopt, ref, cur, T_cur_ref, pxr, pyr, f, gx, gy, level, type, d_est, d_min, d_max);
      px0 = m.px_cur[0]; px1 = m.px_cur[1];
      if (code == kMatchRefinePending) { sx0 = m.px_cur[0] / (1 << m.search_level); sy0 = m.px_cur[1] / (1 << m.search_level); }
    }
    search_level = m.search_level;
    n_warp = m.n_warp; n_zmssd = m.n_zmssd;
    for (int k = 0; k < 4; ++k) A4[k] = m.A[k];
    edir0 = m.epi_dir[0]; edir1 = m.epi_dir[1];
  }
  s_stat[tid] = (ci << 8) | search_level;
  if (code == kMatchRefinePending)
    h_inv_1d = packed_refine_enqueue(tid, is_1d, edir0, edir1, s_pwb + tid * kPwbStride, sx0, sy0, est_offset, est_gain, s_hm, s_dir,
                                     s_u, s_v, s_q, s_qn);
  __syncthreads();
  packed_refine_phase(a, tid, s_pwb, s_res, s_dir, s_u, s_v, s_stat, s_q, s_qn, s_qh, s_hm, est_offset, est_gain);
  __syncthreads();
  if (!live) return;
  if (!ok_idx) {
    a.result[i] = SVOH_MATCH_NOT_RUN;
    reinterpret_cast<uint4*>(a.unit_counts)[i] = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  const DevFrameView& cur = a.cur_frame[ci];
  int res = code, n_align_it = 0;
  Vec3 f_cur = { 0.0, 0.0, 0.0 };
  double depth = 0.0;
  bool aligned = false;
  double sx = 0.0, sy = 0.0;
  if (code == kMatchRefinePending) {
    const int r = s_res[tid];
    n_align_it = r >> 8;
    aligned = (r & 1) != 0;
    sx = s_u[tid]; sy = s_v[tid];
  }
  if constexpr (DIRECT) {
    if (code == kMatchRefinePending) {
      // the tail of Matcher::findMatchDirect (matcher.cpp:94-141)
      if (!aligned) res = SVOH_MATCH_FAIL_ALIGNMENT;
      else {
        const double dx = sx - sx0, dy = sy - sy0;   // against the start position before its narrowing to float
        constexpr int kPatchSize = 8;
        if (sqrt(dx * dx + dy * dy) > a.mopt.max_patch_diff_ratio * kPatchSize) res = SVOH_MATCH_FAIL_TOO_FAR;
        else {
          px0 = sx * (1 << search_level); px1 = sy * (1 << search_level);
          f_cur = back_project3(cur.cam, px0, px1);
          normalize3(f_cur);
          res = SVOH_MATCH_SUCCESS;
        }
      }
    }
    a.result[i] = res;
    a.px_cur[2 * i] = px0; a.px_cur[2 * i + 1] = px1;
  } else {
    if (code == kMatchRefinePending || code == kMatchTriangulatePending) {
      const int ri = a.ref_frame_idx[i];
      const DevFrameView& ref = a.ref_frames[ri];
      const Rigid T_cur_ref = a.T_cur_ref ? a.T_cur_ref[(size_t)ri * a.n_cur_frames + ci] : T_cur_ref_of(ref, cur);
      const Vec3 f = { a.f[3 * i], a.f[3 * i + 1], a.f[3 * i + 2] };
      MatcherState m;
      m.search_level = search_level;
      m.px_cur[0] = px0; m.px_cur[1] = px1;
      m.f_cur = { 0.0, 0.0, 0.0 };
      res = epipolar_match_finish(m, cur, T_cur_ref, f, code == kMatchRefinePending, aligned, sx, sy, depth);
      px0 = m.px_cur[0]; px1 = m.px_cur[1];
      f_cur = m.f_cur;
    }
    a.result[i] = res;
    a.depth_out[i] = depth;
    if (a.px_cur) { a.px_cur[2 * i] = px0; a.px_cur[2 * i + 1] = px1; }
  }
  if (a.f_cur) { a.f_cur[3 * i] = f_cur.x; a.f_cur[3 * i + 1] = f_cur.y; a.f_cur[3 * i + 2] = f_cur.z; }
  if (a.search_level) a.search_level[i] = search_level;
  if (a.h_inv) a.h_inv[i] = (code == kMatchRefinePending && is_1d) ? h_inv_1d : 0.0;
  if (a.A_cur_ref) for (int k = 0; k < 4; ++k) a.A_cur_ref[4 * i + k] = A4[k];
  reinterpret_cast<uint4*>(a.unit_counts)[i] = make_uint4((unsigned)n_warp, (unsigned)n_zmssd, (unsigned)n_align_it,
                                                           (!DIRECT && res == SVOH_MATCH_SUCCESS) ? 1u : 0u);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

static int fill_view(svoh_ctx* ctx, const svoh_frame_view& v, DevFrameView* out, const char* what, bool pose_from_results_allowed = false)
{
  if (v.pose_result_index_plus1 != 0 && !pose_from_results_allowed)
    return set_error(ctx, SVOH_ERR_INVALID_ARGUMENT, "%s: pose_result_index_plus1 is for the current frames of a staged seed batch queued from the hook of svoh_optimize_pose_batch_hook", what);
  const Frame* f = find_frame(ctx, v.frame);
  if (!f) return set_error(ctx, SVOH_ERR_BAD_HANDLE, "%s: unknown frame handle %llu", what, (unsigned long long)v.frame);
  if (v.cam.distortion != SVOH_DISTORTION_NONE && v.cam.distortion != SVOH_DISTORTION_RADTAN)
    return set_error(ctx, SVOH_ERR_UNSUPPORTED, "%s: unsupported distortion model", what);
  // the matcher's bounds tests use the camera's size (matcher.cpp:67-70, 340-413): it must be the frame's
  if (v.cam.width != f->lv[0].w || v.cam.height != f->lv[0].h)
    return set_error(ctx, SVOH_ERR_INVALID_ARGUMENT, "%s: camera is %dx%d but the frame's level 0 is %dx%d", what,
                     v.cam.width, v.cam.height, f->lv[0].w, f->lv[0].h);
  for (int l = 0; l < SVOH_MAX_LEVELS; ++l) out->lv[l] = l < f->n_levels ? f->lv[l] : DevImage{ nullptr, 0, 0, 0, 0 };
  out->cam = load_camera(v.cam);
  out->T_f_w = load_rigid(v.T_f_w);
  out->seed_mu_range = v.seed_mu_range;
  out->n_levels = f->n_levels;
  out->id = v.id;
  out->pose_result_index_plus1 = v.pose_result_index_plus1;
  out->feat_n = 0; out->feat_px = out->feat_f = out->feat_grad = nullptr; out->feat_level = nullptr;
  out->feat_spx = out->feat_sf = out->feat_sgrad = nullptr; out->feat_slevel = out->feat_perm = nullptr; out->unit_begin = 0;
  if (v.features) {
    auto it = ctx->feature_sets.find(v.features);
    if (it == ctx->feature_sets.end()) return set_error(ctx, SVOH_ERR_BAD_HANDLE, "%s: unknown feature-set handle %llu", what, (unsigned long long)v.features);
    out->feat_n = it->second.n; out->feat_px = it->second.px; out->feat_f = it->second.f; out->feat_grad = it->second.grad; out->feat_level = it->second.level;
    out->feat_spx = it->second.spx; out->feat_sf = it->second.sf; out->feat_sgrad = it->second.sgrad; out->feat_slevel = it->second.slevel; out->feat_perm = it->second.perm;
  }
  return SVOH_OK;
}

struct Staging {
  std::vector<std::pair<const void*, size_t>> in;   // host source, bytes
  std::vector<size_t> off;
  size_t total = 0;
  size_t add(const void* p, size_t bytes)
  {
    const size_t o = total;
    in.emplace_back(p, bytes);
    off.push_back(o);
    total = (total + bytes + 63) & ~(size_t)63;
    return o;
  }
};

__global__ void count_success_kernel(const uint8_t* success, int n, int* out)
{
  int c = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) c += success[i];
  c = wave_sum_i32_dpp(c);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// The kernels of one batch in geometry g8 (0 one lane per unit, 1 eight lanes, 2 packed, 3 one wave per unit), on the
// context's stream; max_w x max_h = the largest reference frame (bins of the packed seed update).
static int launch_matcher_kernels(svoh_ctx* ctx, bool seeds, int g8, MatcherArgs& a, int n, int n_ref_frames, int max_w, int max_h)
{
  const int units_per_block = g8 == 1 ? 8 : (g8 == 3 ? 1 : 64);
  const dim3 grid((unsigned)((n + units_per_block - 1) / units_per_block)), block(64);
  if (seeds) {
    if (g8 == 2) {
      if (max_w < 1) max_w = 1;
      if (max_h < 1) max_h = 1;
      const int tiles_x = (max_w + (1 << kBinShiftX) - 1) >> kBinShiftX, tiles_y = (max_h + (1 << kBinShiftY) - 1) >> kBinShiftY;
      const size_t n_keys = (size_t)n_ref_frames * tiles_x * tiles_y;
      const SeedRecIn* rec_in = nullptr;
      SeedRecOut* rec_out = nullptr;
      const unsigned* pos_of = nullptr;
      unsigned* hist_ptr = nullptr;
      unsigned hist_keys = 0;
      const dim3 gb((unsigned)((n + 255) / 256));
      if (!a.whole_sets && n_keys <= kBinMaxKeys && SvohKnobs::or_default(ctx->knobs.seed_binning, 1) != 0) {
        // [hist | rank | pos_of | records in (128 B each) | records out (64 B each)]
        const size_t o_rank = (n_keys * sizeof(unsigned) + 255) & ~(size_t)255;
        const size_t o_pos = o_rank + (((size_t)n * sizeof(unsigned) + 255) & ~(size_t)255);
        const size_t o_in = o_pos + (((size_t)n * sizeof(unsigned) + 255) & ~(size_t)255);
        const size_t o_out = o_in + (size_t)n * sizeof(SeedRecIn);
        SVOH_HIP_TRY(ctx, ctx->d_seed_bin.reserve(o_out + (size_t)n * sizeof(SeedRecOut)));
        uint8_t* base = static_cast<uint8_t*>(ctx->d_seed_bin.ptr);
        unsigned* hist = reinterpret_cast<unsigned*>(base);
        unsigned* rank = reinterpret_cast<unsigned*>(base + o_rank);
        unsigned* pos = reinterpret_cast<unsigned*>(base + o_pos);
        // the histogram is left zeroed by the previous call's last pass unless the block moved or the key count grew
        if (ctx->seed_hist_ptr != static_cast<void*>(hist) || n_keys > ctx->seed_hist_clean_keys)
          SVOH_HIP_TRY(ctx, hipMemsetAsync(hist, 0, n_keys * sizeof(unsigned), ctx->stream));
        ctx->seed_hist_ptr = hist;
        ctx->seed_hist_clean_keys = n_keys;
        hist_keys = (unsigned)n_keys;
        hist_ptr = hist;
        hipLaunchKernelGGL(seed_bin_count_kernel, gb, dim3(256), 0, ctx->stream, a, tiles_x, tiles_y, hist, rank);
        if (n_keys <= kBinScanHereMaxKeys) {
          hipLaunchKernelGGL(seed_bin_scatter_kernel<true>, gb, dim3(256), 0, ctx->stream, a, tiles_x, tiles_y, hist, rank, pos,
                             reinterpret_cast<SeedRecIn*>(base + o_in), (unsigned)n_keys);
        } else {
          hipLaunchKernelGGL(seed_bin_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, hist, (unsigned)n_keys);
          hipLaunchKernelGGL(seed_bin_scatter_kernel<false>, gb, dim3(256), 0, ctx->stream, a, tiles_x, tiles_y, hist, rank, pos,
                             reinterpret_cast<SeedRecIn*>(base + o_in), (unsigned)n_keys);
        }
        rec_in = reinterpret_cast<const SeedRecIn*>(base + o_in);
        rec_out = reinterpret_cast<SeedRecOut*>(base + o_out);
        pos_of = pos;
      }
      if (a.whole_sets) hipLaunchKernelGGL(update_seeds_packed_kernel<true>, dim3((unsigned)((n + kPkThreads - 1) / kPkThreads)), dim3(kPkThreads), 0,
                                           ctx->stream, a, rec_in, rec_out);
      else hipLaunchKernelGGL(update_seeds_packed_kernel<false>, dim3((unsigned)((n + kPkThreads - 1) / kPkThreads)), dim3(kPkThreads), 0,
                              ctx->stream, a, rec_in, rec_out);
      if (rec_out) hipLaunchKernelGGL(seed_unsort_kernel, gb, dim3(256), 0, ctx->stream, a, pos_of, static_cast<const SeedRecOut*>(rec_out),
                                      hist_ptr, hist_keys);
    }
    else if (g8 == 3) hipLaunchKernelGGL(update_seeds_kernel<3>, grid, block, 0, ctx->stream, a);
    else if (g8) hipLaunchKernelGGL(update_seeds_kernel<1>, grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL(update_seeds_kernel<0>, grid, block, 0, ctx->stream, a);
  } else {
    if (g8 == 2) hipLaunchKernelGGL(match_packed_kernel<true>, dim3((unsigned)((n + kPkThreads - 1) / kPkThreads)), dim3(kPkThreads), 0, ctx->stream, a);
    else if (g8) hipLaunchKernelGGL(match_direct_kernel<true>, grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL(match_direct_kernel<false>, grid, block, 0, ctx->stream, a);
  }
  SVOH_HIP_TRY(ctx, hipGetLastError());
  return SVOH_OK;
}

static int run_matcher_staged(svoh_ctx* ctx, bool seeds, const svoh_matcher_options* mopt, const svoh_depth_filter_options* dopt,
                              int n_ref_frames, const svoh_frame_view* ref_frames, const svoh_frame_view* cur_frame,
                              const svoh_feature_batch* fb, const double* depth, double* px_cur, int32_t* result, double* f_cur,
                              int32_t* search_level, double* h_inv, double* A_cur_ref, double* state, uint8_t* success, int32_t* n_success);

static int run_matcher(svoh_ctx* ctx, bool seeds, const svoh_matcher_options* mopt, const svoh_depth_filter_options* dopt,
                       int n_ref_frames, const svoh_frame_view* ref_frames, const svoh_frame_view* cur_frame,
                       const svoh_feature_batch* fb, const double* depth, double* px_cur, int32_t* result, double* f_cur,
                       int32_t* search_level, double* h_inv, double* A_cur_ref, double* state, uint8_t* success,
                       int32_t* n_success, const double* landmark_xyz = nullptr)
{
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, mopt && ref_frames && cur_frame && fb && n_ref_frames >= 1, "NULL argument");
  SVOH_REQUIRE(ctx, !landmark_xyz || (!seeds && !ctx->matcher_deferred), "the pixelwise warp: direct matches only, outside a deferred section");
  const int n = fb->n;
  if (n_success) *n_success = 0;
  if (n <= 0) return SVOH_OK;
  if (fb->mem_space == SVOH_MEM_STAGED) {
    SVOH_REQUIRE(ctx, !landmark_xyz, "the pixelwise warp has no staged form");
    return run_matcher_staged(ctx, seeds, mopt, dopt, n_ref_frames, ref_frames, cur_frame, fb, depth, px_cur, result, f_cur, search_level,
                              h_inv, A_cur_ref, state, success, n_success);
  }
  SVOH_REQUIRE(ctx, fb->mem_space == SVOH_MEM_HOST || fb->mem_space == SVOH_MEM_DEVICE, "bad mem_space");
  SVOH_REQUIRE(ctx, !fb->feature_index, "feature_index: staged batches only (svoh_matcher_stage with SVOH_STAGE_RESIDENT_COLUMNS)");
  // SVOH_BATCH_WHOLE_SETS on a DEVICE batch: the seeds' state / type / current-frame index live in the caller's device arrays in the
  // sets' order, the columns are the reference frames' resident ones; always the packed geometry (which walks the tile-ordered copies)
  const bool whole_sets = fb->layout == SVOH_BATCH_WHOLE_SETS;
  SVOH_REQUIRE(ctx, fb->layout == SVOH_BATCH_UNITS || (whole_sets && seeds && fb->mem_space == SVOH_MEM_DEVICE && !ctx->matcher_deferred),
               "svoh_feature_batch::layout: SVOH_BATCH_WHOLE_SETS is for seed batches over resident columns, staged (svoh_matcher_stage) or with device arrays");
  const bool on_device = fb->mem_space == SVOH_MEM_DEVICE;
  SVOH_REQUIRE(ctx, fb->type && (whole_sets || (fb->ref_frame_idx && fb->px && fb->f && fb->grad && fb->level)), "NULL feature array");
  if (seeds) SVOH_REQUIRE(ctx, dopt && state && success, "NULL seed argument");
  else SVOH_REQUIRE(ctx, depth && px_cur && result, "NULL match argument");
  const int n_cur = (fb->cur_frame_idx && fb->n_cur_frames > 0) ? fb->n_cur_frames : 1;
  if (!on_device) {
    for (int i = 0; i < n; ++i) {
      SVOH_REQUIRE(ctx, fb->ref_frame_idx[i] >= 0 && fb->ref_frame_idx[i] < n_ref_frames, "ref_frame_idx out of range");
      SVOH_REQUIRE(ctx, fb->level[i] >= 0 && fb->level[i] < SVOH_MAX_LEVELS, "feature level out of range");
    }
    if (fb->cur_frame_idx)
      for (int i = 0; i < n; ++i)
        SVOH_REQUIRE(ctx, fb->cur_frame_idx[i] >= 0 && fb->cur_frame_idx[i] < n_cur, "cur_frame_idx out of range");
  }
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));

  std::vector<DevFrameView> views((size_t)n_ref_frames + n_cur);
  for (int k = 0; k < n_ref_frames; ++k) {
    int rc = fill_view(ctx, ref_frames[k], &views[k], "reference frame");
    if (rc != SVOH_OK) return rc;
  }
  for (int k = 0; k < n_cur; ++k) {
    int rc = fill_view(ctx, cur_frame[k], &views[n_ref_frames + k], "current frame");
    if (rc != SVOH_OK) return rc;
  }
  if (whole_sets) {
    long long begin = 0;
    for (int k = 0; k < n_ref_frames; ++k) {
      SVOH_REQUIRE(ctx, ref_frames[k].features != 0, "SVOH_BATCH_WHOLE_SETS: a reference frame has no resident columns (svoh_frame_view::features)");
      views[k].unit_begin = (int32_t)begin; begin += views[k].feat_n;
    }
    SVOH_REQUIRE(ctx, begin == n, "SVOH_BATCH_WHOLE_SETS: the reference frames' resident sets do not add up to the batch's n");
  }
  {
    // the search level is chosen up to the reference pyramid's top (patch_warp.cpp:97-110) and then read
    // from the current frame: every current frame needs at least as many levels
    int ref_levels = 0;
    for (int k = 0; k < n_ref_frames; ++k) ref_levels = views[k].n_levels > ref_levels ? views[k].n_levels : ref_levels;
    for (int k = 0; k < n_cur; ++k)
      SVOH_REQUIRE(ctx, views[n_ref_frames + k].n_levels >= ref_levels, "current frame has fewer pyramid levels than a reference frame");
  }
  if (!on_device)
    for (int i = 0; i < n; ++i)
      SVOH_REQUIRE(ctx, fb->level[i] < views[fb->ref_frame_idx[i]].n_levels, "feature level beyond the reference pyramid");

  // Host-resident batches travel through one pinned staging block and one device block:
  // [views | inputs | in/out | outputs]; device-resident batches stage the views only
  // and the kernel reads and writes the caller's arrays in place.
  Staging s;
  const size_t o_views = s.add(views.data(), sizeof(DevFrameView) * views.size());
  size_t o_idx = 0, o_cidx = 0, o_px = 0, o_f = 0, o_grad = 0, o_level = 0, o_type = 0, o_depth = 0, o_pxcur = 0, o_lm = 0,
         o_state = 0, o_result = 0, o_fcur = 0, o_slevel = 0, o_hinv = 0, o_A = 0, o_success = 0;
  if (!on_device) {
    o_idx = s.add(fb->ref_frame_idx, sizeof(int32_t) * n);
    o_cidx = fb->cur_frame_idx ? s.add(fb->cur_frame_idx, sizeof(int32_t) * n) : 0;
    o_px = s.add(fb->px, sizeof(double) * 2 * n);
    o_f = s.add(fb->f, sizeof(double) * 3 * n);
    o_grad = s.add(fb->grad, sizeof(double) * 2 * n);
    o_level = s.add(fb->level, sizeof(int32_t) * n);
    o_type = s.add(fb->type, (size_t)n);
    if (seeds) o_state = s.add(state, sizeof(double) * 4 * n);
    else { o_depth = s.add(depth, sizeof(double) * n); o_pxcur = s.add(px_cur, sizeof(double) * 2 * n); }
    if (landmark_xyz) o_lm = s.add(landmark_xyz, sizeof(double) * 3 * n);
  }
  const size_t in_total = s.total;
  // outputs (device only, appended)
  auto out_add = [&](size_t bytes) { const size_t o = s.total; s.total = (s.total + bytes + 63) & ~(size_t)63; return o; };
  if (!on_device) {
    o_result = out_add(sizeof(int32_t) * n);
    o_fcur = out_add(sizeof(double) * 3 * n);
    o_slevel = out_add(sizeof(int32_t) * n);
    o_hinv = out_add(sizeof(double) * n);
    o_A = out_add(sizeof(double) * 4 * n);
    o_success = out_add((size_t)n);
    if (seeds && px_cur) o_pxcur = out_add(sizeof(double) * 2 * n);  // matcher px_cur_ of the seed update (output only)
  }
  const size_t o_nsucc = out_add(sizeof(int32_t));

  // deferred section (svoh_matcher_begin_deferred): no synchronisation here; each deferred batch stages through
  // buffers of its own kind, which no other call touches (svoh_internal.h)
  const bool defer = ctx->matcher_deferred && !on_device;
  if (defer) {
    SVOH_REQUIRE(ctx, !ctx->matcher_deferred_used[seeds ? 1 : 0], "one batch of each kind per deferred section: collect first");
    ctx->matcher_deferred_used[seeds ? 1 : 0] = true;
  }
  PinnedBuffer& hbuf = defer ? (seeds ? ctx->h_match_seeds : ctx->h_match_direct) : ctx->h_scratch1;
  DevBuffer& dbuf = defer ? (seeds ? ctx->d_match_seeds : ctx->d_match_direct) : ctx->d_scratch1;
  if (defer && seeds) ctx->seed_block.valid = false;   // the seed block is laid out anew (svoh_align_camera::pos_seed_unit)
  SVOH_HIP_TRY(ctx, hbuf.reserve(s.total));
  SVOH_HIP_TRY(ctx, dbuf.reserve(s.total));
  uint8_t* h = static_cast<uint8_t*>(hbuf.ptr);
  uint8_t* d = static_cast<uint8_t*>(dbuf.ptr);
  for (size_t k = 0; k < s.in.size(); ++k) memcpy(h + s.off[k], s.in[k].first, s.in[k].second);
  SVOH_HIP_TRY(ctx, svoh_copy_to_device(ctx, d, h, in_total));
  // geometry: 1 = eight lanes per unit, 0 = one lane per unit, 2 = packed (seed update only; large batches)
  // 3 = one wave per unit (the epipolar scan 64 steps at a time; seed update only, on request: its scans are short)
  int g8 = n <= kG8MaxUnits ? 1 : 2;
  g8 = SvohKnobs::or_default(ctx->knobs.matcher_g8, g8);
  if (g8 < 0 || g8 > 3) g8 = 0;
  if (g8 == 3 && (!seeds || defer)) g8 = 1;   // the direct matcher has no scan
  if (landmark_xyz) g8 = 0;                    // the pixelwise warp exists with one lane per unit only
  if (whole_sets) g8 = 2;                      // (a device batch over whole sets: the packed geometry reads the resident columns itself)
  // Optional outputs of units that return before the matcher runs read back as zeros.  The per-unit kernels write those
  // zeros themselves (match_direct_body, update_seeds_body): a fill of the output block would be one more operation on
  // the stream of every per-frame call (2.6 us each, tools/svoh_call_overhead).  The packed geometry's kernels do not.
  if (!on_device && g8 == 2 && !defer) SVOH_HIP_TRY(ctx, hipMemsetAsync(d + in_total, 0, s.total - in_total, ctx->stream));   // (a deferred batch: at its launch)

  MatcherArgs a;
  memset(&a, 0, sizeof a);
  a.ref_frames = reinterpret_cast<const DevFrameView*>(d + o_views);
  a.cur_frame = a.ref_frames + n_ref_frames;
  a.mopt = *mopt;
  if (dopt) a.dopt = *dopt;
  a.n = n;
  a.n_ref_frames = n_ref_frames;
  a.n_cur_frames = n_cur;
  a.whole_sets = whole_sets ? 1 : 0;
  if (on_device) {
    a.ref_frame_idx = fb->ref_frame_idx; a.cur_frame_idx = fb->cur_frame_idx;
    a.px = fb->px; a.f = fb->f; a.grad = fb->grad; a.level = fb->level; a.type = fb->type;
    a.result = result; a.f_cur = f_cur; a.search_level = search_level; a.h_inv = h_inv; a.A_cur_ref = A_cur_ref;
    a.success = success; a.state = state; a.depth = depth; a.px_cur = px_cur; a.landmark_xyz = landmark_xyz;
  } else {
    a.ref_frame_idx = reinterpret_cast<const int32_t*>(d + o_idx);
    a.cur_frame_idx = fb->cur_frame_idx ? reinterpret_cast<const int32_t*>(d + o_cidx) : nullptr;
    a.px = reinterpret_cast<const double*>(d + o_px);
    a.f = reinterpret_cast<const double*>(d + o_f);
    a.grad = reinterpret_cast<const double*>(d + o_grad);
    a.level = reinterpret_cast<const int32_t*>(d + o_level);
    a.type = d + o_type;
    a.result = reinterpret_cast<int32_t*>(d + o_result);
    a.f_cur = reinterpret_cast<double*>(d + o_fcur);
    a.search_level = reinterpret_cast<int32_t*>(d + o_slevel);
    a.h_inv = reinterpret_cast<double*>(d + o_hinv);
    a.A_cur_ref = reinterpret_cast<double*>(d + o_A);
    a.success = d + o_success;
    if (seeds) {
      a.state = reinterpret_cast<double*>(d + o_state);
      a.px_cur = px_cur ? reinterpret_cast<double*>(d + o_pxcur) : nullptr;
    } else {
      a.depth = reinterpret_cast<const double*>(d + o_depth);
      a.px_cur = reinterpret_cast<double*>(d + o_pxcur);
      a.landmark_xyz = landmark_xyz ? reinterpret_cast<const double*>(d + o_lm) : nullptr;
    }
  }
  // small batches: eight lanes per unit (a launch is as slow as its slowest lane, and eight lanes get a unit done
  // ~3x sooner); large batches: one lane per unit (fewer instructions per unit).  The knob SVOH_MATCHER_G8 (read when the context is made) forces one.
  if (defer) {
    // Deferred section: the launch itself waits for svoh_matcher_flush / _collect, where a direct batch and a seed batch of
    // the same per-unit geometry go out as ONE kernel (match_mixed_kernel).  Remembered: the arguments, the copy of the
    // results back to the pinned block, and the copies from there to the caller's arrays.
    svoh_ctx::DeferredLaunch& dl = ctx->matcher_deferred_launch[seeds ? 1 : 0];
    dl.args.assign(reinterpret_cast<const uint8_t*>(&a), reinterpret_cast<const uint8_t*>(&a) + sizeof a);
    dl.n = n; dl.g8 = g8; dl.valid = true; dl.seed_block_staged = false; dl.d_fidx = nullptr; dl.pose_from_results = false;
    dl.max_w = dl.max_h = 1;
    for (int k = 0; k < n_ref_frames; ++k) { dl.max_w = views[k].lv[0].w > dl.max_w ? views[k].lv[0].w : dl.max_w; dl.max_h = views[k].lv[0].h > dl.max_h ? views[k].lv[0].h : dl.max_h; }
    dl.d_block = d; dl.out_off = in_total; dl.out_bytes = s.total - in_total;
    dl.d2h_dst = h + o_type; dl.d2h_src = d + o_type; dl.d2h_bytes = o_nsucc - o_type;
    dl.views_h = h + o_views; dl.views_d = d + o_views; dl.n_ref = n_ref_frames; dl.n_cur = n_cur;
    dl.cur_frame_handle = cur_frame[0].frame;
    auto later = [&](void* dst, size_t off, size_t bytes) { if (dst && bytes) ctx->matcher_pending.push_back({ dst, h + off, bytes }); };
    if (seeds) {
      later(fb->type, o_type, (size_t)n); later(state, o_state, sizeof(double) * 4 * n); later(success, o_success, (size_t)n);
      later(result, o_result, sizeof(int32_t) * n); later(px_cur, o_pxcur, sizeof(double) * 2 * n);
      later(f_cur, o_fcur, sizeof(double) * 3 * n); later(search_level, o_slevel, sizeof(int32_t) * n);
      later(A_cur_ref, o_A, sizeof(double) * 4 * n);
      if (n_success) ctx->matcher_pending_counts.push_back({ n_success, h + o_success, n });
    } else {
      later(px_cur, o_pxcur, sizeof(double) * 2 * n); later(result, o_result, sizeof(int32_t) * n);
      later(f_cur, o_fcur, sizeof(double) * 3 * n); later(search_level, o_slevel, sizeof(int32_t) * n);
      later(h_inv, o_hinv, sizeof(double) * n); later(A_cur_ref, o_A, sizeof(double) * 4 * n);
    }
    return SVOH_OK;
  }
  {
    unsigned long long* dummy;
    int rc = reset_counters(ctx, &dummy);
    if (rc == SVOH_OK) rc = reserve_unit_counts(ctx, (size_t)n, &a.unit_counts);
    if (rc != SVOH_OK) return rc;
  }
  if (ctx->timing_on()) SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_start, ctx->stream));
  {
    int max_w = 1, max_h = 1;
    for (int k = 0; k < n_ref_frames; ++k) { max_w = views[k].lv[0].w > max_w ? views[k].lv[0].w : max_w; max_h = views[k].lv[0].h > max_h ? views[k].lv[0].h : max_h; }
    const int rc = launch_matcher_kernels(ctx, seeds, g8, a, n, n_ref_frames, max_w, max_h);
    if (rc != SVOH_OK) return rc;
  }
  if (ctx->timing_on()) SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_stop, ctx->stream));
  ctx->misc_timed = ctx->timing_on(); ctx->misc_launched = true;
  {
    int rc = reduce_unit_counts(ctx, (size_t)n);
    if (rc != SVOH_OK) return rc;
  }
  if (on_device) {
    // results stay where the caller put them; only the success count (if asked for) comes back
    if (seeds && n_success) {
      int* d_ns = reinterpret_cast<int*>(d + o_nsucc);
      SVOH_HIP_TRY(ctx, hipMemsetAsync(d_ns, 0, sizeof(int), ctx->stream));
      const int blocks = (n + 255) / 256 > 256 ? 256 : (n + 255) / 256;
      hipLaunchKernelGGL(count_success_kernel, dim3(blocks), dim3(256), 0, ctx->stream, success, n, d_ns);
      SVOH_HIP_TRY(ctx, hipGetLastError());
      SVOH_HIP_TRY(ctx, hipMemcpyAsync(h + o_nsucc, d_ns, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      *n_success = *reinterpret_cast<const int*>(h + o_nsucc);
    }
    return SVOH_OK;
  }
  // everything after the inputs that may have changed comes back in one copy
  const size_t back_from = o_type;
  SVOH_HIP_TRY(ctx, svoh_copy_to_host(ctx, h + back_from, d + back_from, o_nsucc - back_from));
  SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (seeds) {
    memcpy(fb->type, h + o_type, (size_t)n);
    memcpy(state, h + o_state, sizeof(double) * 4 * n);
    memcpy(success, h + o_success, (size_t)n);
    if (result) memcpy(result, h + o_result, sizeof(int32_t) * n);
    if (px_cur) memcpy(px_cur, h + o_pxcur, sizeof(double) * 2 * n);
    if (f_cur) memcpy(f_cur, h + o_fcur, sizeof(double) * 3 * n);
    if (search_level) memcpy(search_level, h + o_slevel, sizeof(int32_t) * n);
    if (A_cur_ref) memcpy(A_cur_ref, h + o_A, sizeof(double) * 4 * n);
    if (n_success) {
      int c = 0;
      for (int i = 0; i < n; ++i) c += success[i];
      *n_success = c;
    }
  } else {
    memcpy(px_cur, h + o_pxcur, sizeof(double) * 2 * n);
    memcpy(result, h + o_result, sizeof(int32_t) * n);
    if (f_cur) memcpy(f_cur, h + o_fcur, sizeof(double) * 3 * n);
    if (search_level) memcpy(search_level, h + o_slevel, sizeof(int32_t) * n);
    if (h_inv) memcpy(h_inv, h + o_hinv, sizeof(double) * n);
    if (A_cur_ref) memcpy(A_cur_ref, h + o_A, sizeof(double) * 4 * n);
  }
  return SVOH_OK;
}

// svoh_epipolar_match_batch: same staging scheme as run_matcher ([views | transforms | inputs | outputs])
static int run_epipolar(svoh_ctx* ctx, const svoh_matcher_options* mopt, int n_ref_frames, const svoh_frame_view* ref_frames,
                        const svoh_frame_view* cur_frame, const svoh_se3* T_cur_ref, const svoh_feature_batch* fb,
                        const double d_inv_common[3], const double* d_inv, const svoh_epipolar_match_outputs* out)
{
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, mopt && ref_frames && cur_frame && fb && out && n_ref_frames >= 1, "NULL argument");
  SVOH_REQUIRE(ctx, d_inv_common || d_inv, "no depth range given");
  const int n = fb->n;
  if (n <= 0) return SVOH_OK;
  SVOH_REQUIRE(ctx, fb->mem_space == SVOH_MEM_HOST || fb->mem_space == SVOH_MEM_DEVICE, "bad mem_space");
  SVOH_REQUIRE(ctx, !fb->feature_index, "feature_index: staged batches only (svoh_matcher_stage with SVOH_STAGE_RESIDENT_COLUMNS)");
  SVOH_REQUIRE(ctx, fb->layout == SVOH_BATCH_UNITS, "svoh_feature_batch::layout: SVOH_BATCH_WHOLE_SETS is for staged seed batches over resident columns");
  const bool on_device = fb->mem_space == SVOH_MEM_DEVICE;
  SVOH_REQUIRE(ctx, fb->ref_frame_idx && fb->px && fb->f && fb->grad && fb->level && fb->type, "NULL feature array");
  SVOH_REQUIRE(ctx, out->result && out->depth, "result and depth outputs are required");
  const int n_cur = (fb->cur_frame_idx && fb->n_cur_frames > 0) ? fb->n_cur_frames : 1;
  if (!on_device) {
    for (int i = 0; i < n; ++i) {
      SVOH_REQUIRE(ctx, fb->ref_frame_idx[i] >= 0 && fb->ref_frame_idx[i] < n_ref_frames, "ref_frame_idx out of range");
      SVOH_REQUIRE(ctx, fb->level[i] >= 0 && fb->level[i] < SVOH_MAX_LEVELS, "feature level out of range");
      if (fb->cur_frame_idx)
        SVOH_REQUIRE(ctx, fb->cur_frame_idx[i] >= 0 && fb->cur_frame_idx[i] < n_cur, "cur_frame_idx out of range");
    }
  }
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::vector<DevFrameView> views((size_t)n_ref_frames + n_cur);
  int ref_levels = 0;
  for (int k = 0; k < n_ref_frames; ++k) {
    const int rc = fill_view(ctx, ref_frames[k], &views[k], "reference frame");
    if (rc != SVOH_OK) return rc;
    ref_levels = views[k].n_levels > ref_levels ? views[k].n_levels : ref_levels;
  }
  for (int k = 0; k < n_cur; ++k) {
    const int rc = fill_view(ctx, cur_frame[k], &views[n_ref_frames + k], "current frame");
    if (rc != SVOH_OK) return rc;
    SVOH_REQUIRE(ctx, views[n_ref_frames + k].n_levels >= ref_levels, "current frame has fewer pyramid levels than a reference frame");
  }
  if (!on_device)
    for (int i = 0; i < n; ++i)
      SVOH_REQUIRE(ctx, fb->level[i] < views[fb->ref_frame_idx[i]].n_levels, "feature level beyond the reference pyramid");
  std::vector<Rigid> T;
  if (T_cur_ref) {
    T.resize((size_t)n_ref_frames * n_cur);
    for (size_t k = 0; k < T.size(); ++k) T[k] = load_rigid(T_cur_ref[k]);
  }

  Staging s;
  const size_t o_views = s.add(views.data(), sizeof(DevFrameView) * views.size());
  const size_t o_T = T_cur_ref ? s.add(T.data(), sizeof(Rigid) * T.size()) : 0;
  size_t o_idx = 0, o_cidx = 0, o_px = 0, o_f = 0, o_grad = 0, o_level = 0, o_type = 0, o_dinv = 0;
  if (!on_device) {
    o_idx = s.add(fb->ref_frame_idx, sizeof(int32_t) * n);
    o_cidx = fb->cur_frame_idx ? s.add(fb->cur_frame_idx, sizeof(int32_t) * n) : 0;
    o_px = s.add(fb->px, sizeof(double) * 2 * n);
    o_f = s.add(fb->f, sizeof(double) * 3 * n);
    o_grad = s.add(fb->grad, sizeof(double) * 2 * n);
    o_level = s.add(fb->level, sizeof(int32_t) * n);
    o_type = s.add(fb->type, (size_t)n);
    o_dinv = d_inv ? s.add(d_inv, sizeof(double) * 3 * n) : 0;
  }
  const size_t in_total = s.total;
  auto out_add = [&](size_t bytes) { const size_t o = s.total; s.total = (s.total + bytes + 63) & ~(size_t)63; return o; };
  size_t o_result = 0, o_depth = 0, o_pxcur = 0, o_fcur = 0, o_slevel = 0, o_hinv = 0, o_A = 0;
  if (!on_device) {
    o_result = out_add(sizeof(int32_t) * n);
    o_depth = out_add(sizeof(double) * n);
    o_pxcur = out_add(sizeof(double) * 2 * n);
    o_fcur = out_add(sizeof(double) * 3 * n);
    o_slevel = out_add(sizeof(int32_t) * n);
    o_hinv = out_add(sizeof(double) * n);
    o_A = out_add(sizeof(double) * 4 * n);
  }
  const size_t o_end = s.total;
  SVOH_HIP_TRY(ctx, ctx->h_scratch1.reserve(s.total));
  SVOH_HIP_TRY(ctx, ctx->d_scratch1.reserve(s.total));
  uint8_t* h = static_cast<uint8_t*>(ctx->h_scratch1.ptr);
  uint8_t* d = static_cast<uint8_t*>(ctx->d_scratch1.ptr);
  for (size_t k = 0; k < s.in.size(); ++k) memcpy(h + s.off[k], s.in[k].first, s.in[k].second);
  SVOH_HIP_TRY(ctx, svoh_copy_to_device(ctx, d, h, in_total));
  if (!on_device && o_end > in_total) SVOH_HIP_TRY(ctx, hipMemsetAsync(d + in_total, 0, o_end - in_total, ctx->stream));

  MatcherArgs a;
  memset(&a, 0, sizeof a);
  a.ref_frames = reinterpret_cast<const DevFrameView*>(d + o_views);
  a.cur_frame = a.ref_frames + n_ref_frames;
  a.T_cur_ref = T_cur_ref ? reinterpret_cast<const Rigid*>(d + o_T) : nullptr;
  a.mopt = *mopt;
  a.n = n;
  a.n_ref_frames = n_ref_frames;
  a.n_cur_frames = n_cur;
  if (d_inv_common) for (int k = 0; k < 3; ++k) a.d_inv_common[k] = d_inv_common[k];
  if (on_device) {
    a.ref_frame_idx = fb->ref_frame_idx; a.cur_frame_idx = fb->cur_frame_idx;
    a.px = fb->px; a.f = fb->f; a.grad = fb->grad; a.level = fb->level; a.type = fb->type;
    a.d_inv = d_inv;
    a.result = out->result; a.depth_out = out->depth; a.px_cur = out->px_cur; a.f_cur = out->f_cur;
    a.search_level = out->search_level; a.h_inv = out->h_inv; a.A_cur_ref = out->A_cur_ref;
  } else {
    a.ref_frame_idx = reinterpret_cast<const int32_t*>(d + o_idx);
    a.cur_frame_idx = fb->cur_frame_idx ? reinterpret_cast<const int32_t*>(d + o_cidx) : nullptr;
    a.px = reinterpret_cast<const double*>(d + o_px);
    a.f = reinterpret_cast<const double*>(d + o_f);
    a.grad = reinterpret_cast<const double*>(d + o_grad);
    a.level = reinterpret_cast<const int32_t*>(d + o_level);
    a.type = d + o_type;
    a.d_inv = d_inv ? reinterpret_cast<const double*>(d + o_dinv) : nullptr;
    a.result = reinterpret_cast<int32_t*>(d + o_result);
    a.depth_out = reinterpret_cast<double*>(d + o_depth);
    a.px_cur = reinterpret_cast<double*>(d + o_pxcur);
    a.f_cur = reinterpret_cast<double*>(d + o_fcur);
    a.search_level = reinterpret_cast<int32_t*>(d + o_slevel);
    a.h_inv = reinterpret_cast<double*>(d + o_hinv);
    a.A_cur_ref = reinterpret_cast<double*>(d + o_A);
  }
  // geometry as in run_matcher: 1 = eight lanes per unit, 0 = one lane per unit, 2 = packed (large batches),
  // 3 = one wave per unit, the scan 64 steps at a time: what a long search (the stereo seam's 500 steps) gets while the
  // batch leaves the device room for a wave per feature (a launch is as slow as its slowest feature: 8 rounds instead of 63)
  constexpr int kW64MaxUnits = 16384;
  int g8 = n <= kG8MaxUnits ? ((mopt->max_epi_search_steps > 128 && n <= kW64MaxUnits) ? 3 : 1) : 2;
  g8 = SvohKnobs::or_default(ctx->knobs.matcher_g8, g8);
  if (g8 < 0 || g8 > 3) g8 = 0;
  const int units_per_block = g8 == 1 ? 8 : (g8 == 3 ? 1 : 64);
  const dim3 grid((unsigned)((n + units_per_block - 1) / units_per_block)), block(64);
  {
    unsigned long long* dummy;
    int rc = reset_counters(ctx, &dummy);
    if (rc == SVOH_OK) rc = reserve_unit_counts(ctx, (size_t)n, &a.unit_counts);
    if (rc != SVOH_OK) return rc;
  }
  if (ctx->timing_on()) SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_start, ctx->stream));
  if (g8 == 2) hipLaunchKernelGGL(match_packed_kernel<false>, dim3((unsigned)((n + kPkThreads - 1) / kPkThreads)), dim3(kPkThreads), 0, ctx->stream, a);
  else if (g8 == 3) hipLaunchKernelGGL(epipolar_match_kernel<3>, grid, block, 0, ctx->stream, a);
  else if (g8) hipLaunchKernelGGL(epipolar_match_kernel<1>, grid, block, 0, ctx->stream, a);
  else hipLaunchKernelGGL(epipolar_match_kernel<0>, grid, block, 0, ctx->stream, a);
  SVOH_HIP_TRY(ctx, hipGetLastError());
  if (ctx->timing_on()) SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_stop, ctx->stream));
  ctx->misc_timed = ctx->timing_on(); ctx->misc_launched = true;
  {
    const int rc = reduce_unit_counts(ctx, (size_t)n);
    if (rc != SVOH_OK) return rc;
  }
  if (on_device) return SVOH_OK;
  SVOH_HIP_TRY(ctx, hipMemcpyAsync(h + o_result, d + o_result, o_end - o_result, hipMemcpyDeviceToHost, ctx->stream));
  SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(out->result, h + o_result, sizeof(int32_t) * n);
  memcpy(out->depth, h + o_depth, sizeof(double) * n);
  if (out->px_cur) memcpy(out->px_cur, h + o_pxcur, sizeof(double) * 2 * n);
  if (out->f_cur) memcpy(out->f_cur, h + o_fcur, sizeof(double) * 3 * n);
  if (out->search_level) memcpy(out->search_level, h + o_slevel, sizeof(int32_t) * n);
  if (out->h_inv) memcpy(out->h_inv, h + o_hinv, sizeof(double) * n);
  if (out->A_cur_ref) memcpy(out->A_cur_ref, h + o_A, sizeof(double) * 4 * n);
  return SVOH_OK;
}


// ---- f-4: candidate projection of Reprojector::reprojectFrames ---------------------------------------------------
// reprojector_utils::getCandidate / projectPointAndCheckVisibility (src/svo/src/reprojector.cpp:489-543) with
// Frame::isVisible (src/svo_common/src/frame.cpp:229-260) for every point of the local map: the position of a
// landmark, or of a seed T_world_kf * (f / mu) (Frame::getSeedPosInFrame), is taken into the current frame, tested
// against the cone of the image's top-left corner and the image box, then against the 8-pixel margin on the
// truncated pixel.  The current frame's pose is given, or composed on the device from the alignment result that
// the launch queued in front of this one has left in device memory (T_f_w = T_cam_imu * T_icur_iref * T_imu_world of
// the reference frame: sparse_img_align.cpp:100-107), so that the projection needs no round trip of its own.
// This file is compiled without FMA contraction and the maths is the host mirror's (svoh_math.h): same bits as
// reprojector_utils::getCandidate of the host layer.
struct CandidateArgs {
  svoh_camera cam;
  svoh_se3 T_a;                       // T_f_w of the current frame, or T_cam_imu of it when `align_result` is set
  svoh_se3 T_b;                       // ... then T_imu_world of the alignment's reference frame
  const svoh_align_result* align_result;
  const svoh_se3* T_world_kf;         // n_kf
  const uint8_t* kind;                // n: 0 world point, 1 seed of keyframe kf[i]
  const int32_t* kf;                  // n
  const double* v;                    // 3 x n: position, or bearing vector of the seed
  const double* mu;                   // n: inverse depth of the seed
  double* px;                         // 2 x n
  uint8_t* visible;                   // n
  int n, n_kf;
};

__global__ __launch_bounds__(256) void project_candidates_kernel(const CandidateArgs a)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  Rigid T_f_w = load_rigid(a.T_a);
  if (a.align_result) T_f_w = mul(mul(T_f_w, load_rigid(a.align_result->T_icur_iref)), load_rigid(a.T_b));
  const CamModel cm = load_camera(a.cam);
  Vec3 xyz = { a.v[3 * i], a.v[3 * i + 1], a.v[3 * i + 2] };
  bool ok = true;
  if (a.kind[i]) {
    const int k = a.kf[i];
    ok = k >= 0 && k < a.n_kf;
    if (ok) {
      const double depth = 1.0 / a.mu[i];                        // seed::getDepth (seed.h:110-113)
      const Vec3 in_f = { xyz.x * depth, xyz.y * depth, xyz.z * depth };
      xyz = transform(load_rigid(a.T_world_kf[k]), in_f);
    }
  }
  double u = 0.0, v = 0.0;
  if (ok) {
    const Vec3 xyz_f = transform(T_f_w, xyz);
    // pinhole: not farther off the optical axis than the image's top-left corner (frame.cpp:233-246)
    const Vec3 f_tl = back_project3(cm, 0.0, 0.0);
    const double min_cos = f_tl.z / sqrt(f_tl.x * f_tl.x + f_tl.y * f_tl.y + f_tl.z * f_tl.z);
    const double cur_cos = xyz_f.z / sqrt(xyz_f.x * xyz_f.x + xyz_f.y * xyz_f.y + xyz_f.z * xyz_f.z);
    ok = !(cur_cos < min_cos);
    if (ok) {
      project3(cm, xyz_f, u, v);
      ok = u >= 0.0 && v >= 0.0 && u < (double)a.cam.width && v < (double)a.cam.height;   // isKeypointVisible
      if (ok) {
        const int pxi0 = (int)u, pxi1 = (int)v;                  // px->cast<int>(), margin kPatchSize = 8 (:526-529)
        ok = pxi0 >= 8 && pxi1 >= 8 && pxi0 < a.cam.width - 8 && pxi1 < a.cam.height - 8;
      }
    }
  }
  a.px[2 * i] = u; a.px[2 * i + 1] = v;
  a.visible[i] = ok ? 1 : 0;
}

static int enqueue_candidates(svoh_ctx* ctx, const svoh_camera* cam, const svoh_se3* T_a, const svoh_se3* T_b, int align_result_index,
                              int n_kf, const svoh_se3* T_world_kf, int n, const uint8_t* kind, const int32_t* kf, const double* v,
                              const double* mu)
{
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, cam && T_a && n >= 1 && n <= (1 << 22) && kind && kf && v && mu, "bad arguments");
  SVOH_REQUIRE(ctx, n_kf >= 0 && n_kf <= 4096 && (n_kf == 0 || T_world_kf), "bad keyframe table");
  SVOH_REQUIRE(ctx, cam->distortion == SVOH_DISTORTION_NONE || cam->distortion == SVOH_DISTORTION_RADTAN, "unsupported distortion model");
  SVOH_REQUIRE(ctx, ctx->cand_pending_n == 0, "a candidate projection is queued already: collect first");
  if (align_result_index >= 0) {
    SVOH_REQUIRE(ctx, T_b != nullptr, "T_post is NULL");
    SVOH_REQUIRE(ctx, (size_t)align_result_index < ctx->align_pending_results && (size_t)align_result_index < ctx->align_result_dev_index.size() && ctx->d_results.ptr,
                 "no queued alignment result with this index");
  }
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t nd = (size_t)n;
  auto al = [](size_t b) { return (b + 63) & ~(size_t)63; };
  const size_t o_kf = 0, o_kind = o_kf + al(sizeof(svoh_se3) * (size_t)(n_kf > 0 ? n_kf : 1)), o_idx = o_kind + al(nd);
  const size_t o_v = o_idx + al(4 * nd), o_mu = o_v + al(24 * nd), in_total = o_mu + al(8 * nd);
  const size_t o_px = in_total, o_vis = o_px + al(16 * nd), total = o_vis + al(nd);
  SVOH_HIP_TRY(ctx, ctx->h_cand.reserve(total));
  SVOH_HIP_TRY(ctx, ctx->d_cand.reserve(total));
  uint8_t* h = static_cast<uint8_t*>(ctx->h_cand.ptr);
  uint8_t* d = static_cast<uint8_t*>(ctx->d_cand.ptr);
  if (n_kf > 0) memcpy(h + o_kf, T_world_kf, sizeof(svoh_se3) * (size_t)n_kf);
  memcpy(h + o_kind, kind, nd); memcpy(h + o_idx, kf, 4 * nd); memcpy(h + o_v, v, 24 * nd); memcpy(h + o_mu, mu, 8 * nd);
  SVOH_HIP_TRY(ctx, svoh_copy_to_device(ctx, d, h, in_total));
  CandidateArgs a;
  memset(&a, 0, sizeof a);
  a.cam = *cam; a.T_a = *T_a;
  if (T_b) a.T_b = *T_b;
  a.align_result = align_result_index >= 0 ? static_cast<const svoh_align_result*>(ctx->d_results.ptr) + ctx->align_result_dev_index[(size_t)align_result_index] : nullptr;
  a.T_world_kf = reinterpret_cast<const svoh_se3*>(d + o_kf);
  a.kind = d + o_kind; a.kf = reinterpret_cast<const int32_t*>(d + o_idx);
  a.v = reinterpret_cast<const double*>(d + o_v); a.mu = reinterpret_cast<const double*>(d + o_mu);
  a.px = reinterpret_cast<double*>(d + o_px); a.visible = d + o_vis;
  a.n = n; a.n_kf = n_kf;
  hipLaunchKernelGGL(project_candidates_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(ctx, SVOH_ERR_HIP, "project_candidates launch failed: %s", hipGetErrorString(e));
  SVOH_HIP_TRY(ctx, svoh_copy_to_host(ctx, h + o_px, d + o_px, total - o_px));
  ctx->cand_pending_n = n;
  ctx->cand_out_off = o_px;
  return SVOH_OK;
}


// ---- a matcher batch staged in place (svoh_matcher_stage, SVOH_MEM_STAGED) -------------------------------------------
// The caller -- the lock-step front end of many camera streams, with one host thread per few streams -- writes a batch's
// inputs straight into the section's page-locked block and reads its outputs from there: no concatenation of per-stream
// arrays, no staging copy inside the call, no copy to caller arrays at collect (at 32 streams those were 3 - 5 MB each way
// per batch, on ONE thread).  Layout: [views (max) | ref idx | cur idx | px | f | grad | level | (direct: depth) |
// type | (direct: px_cur in/out; seeds: state in/out) | result | success | (match outputs) | n_success]; everything from
// `type` on comes back.
static void layout_matcher_stage(bool seeds, int n, int max_views, bool want_outputs, bool resident, svoh_ctx::MatcherStage* st)
{
  size_t total = 0;
  auto add = [&](size_t bytes) { const size_t o = total; total = (total + bytes + 63) & ~(size_t)63; return o; };
  const size_t nn = (size_t)n;
  st->n = n; st->max_views = max_views; st->want_outputs = want_outputs || !seeds; st->resident = resident;
  st->o_views = add(sizeof(DevFrameView) * (size_t)max_views);
  st->o_idx = add(4 * nn); st->o_cidx = add(4 * nn);
  // features named by index: the four columns exist on the device only (behind everything that crosses PCIe, see below)
  st->o_fidx = resident ? add(4 * nn) : 0;
  if (!resident) { st->o_px = add(16 * nn); st->o_f = add(24 * nn); st->o_grad = add(16 * nn); st->o_level = add(4 * nn); }
  st->o_depth = seeds ? 0 : add(8 * nn);
  st->o_type = add(nn);
  st->back_from = st->o_type;
  if (seeds) { st->o_state = add(32 * nn); st->o_pxcur = 0; }
  else { st->o_pxcur = add(16 * nn); st->o_state = 0; }
  st->in_total = total;
  st->o_result = add(4 * nn); st->o_success = add(nn);
  if (st->want_outputs) {
    if (seeds) st->o_pxcur = add(16 * nn);
    st->o_fcur = add(24 * nn); st->o_slevel = add(4 * nn); st->o_hinv = add(8 * nn); st->o_A = add(32 * nn);
  } else {
    st->o_fcur = st->o_slevel = st->o_hinv = st->o_A = 0;
  }
  st->o_nsucc = add(sizeof(int32_t));
  if (resident) { st->o_px = add(16 * nn); st->o_f = add(24 * nn); st->o_grad = add(16 * nn); st->o_level = add(4 * nn); }
  st->total = total;
}

static int run_matcher_staged(svoh_ctx* ctx, bool seeds, const svoh_matcher_options* mopt, const svoh_depth_filter_options* dopt,
                              int n_ref_frames, const svoh_frame_view* ref_frames, const svoh_frame_view* cur_frame,
                              const svoh_feature_batch* fb, const double* depth, double* px_cur, int32_t* result, double* f_cur,
                              int32_t* search_level, double* h_inv, double* A_cur_ref, double* state, uint8_t* success, int32_t* n_success)
{
  const int kind = seeds ? 1 : 0;
  svoh_ctx::MatcherStage& st = ctx->matcher_stage[kind];
  SVOH_REQUIRE(ctx, ctx->matcher_deferred, "a staged batch lives in a deferred section (svoh_matcher_begin_deferred)");
  SVOH_REQUIRE(ctx, st.valid, "no staged block of this kind: svoh_matcher_stage first");
  SVOH_REQUIRE(ctx, !ctx->matcher_deferred_used[kind], "one batch of each kind per deferred section: collect first");
  const int n = fb->n;
  PinnedBuffer& hbuf = seeds ? ctx->h_match_seeds : ctx->h_match_direct;
  DevBuffer& dbuf = seeds ? ctx->d_match_seeds : ctx->d_match_direct;
  uint8_t* h = static_cast<uint8_t*>(hbuf.ptr);
  uint8_t* d = static_cast<uint8_t*>(dbuf.ptr);
  SVOH_REQUIRE(ctx, n == st.n && h && d && hbuf.cap >= st.total && dbuf.cap >= st.total, "not the batch that was staged (n differs)");
  const int n_cur = fb->n_cur_frames > 0 ? fb->n_cur_frames : 1;
  SVOH_REQUIRE(ctx, n_ref_frames + n_cur <= st.max_views, "more frames than the staged block has room for (max_frame_views)");
  // the arrays must be the staged ones: anything else would silently be ignored
  auto at = [&](size_t off) { return static_cast<void*>(h + off); };
  SVOH_REQUIRE(ctx, fb->ref_frame_idx == at(st.o_idx) && fb->cur_frame_idx == at(st.o_cidx) && fb->type == at(st.o_type),
               "a staged batch's feature arrays must be the pointers svoh_matcher_stage handed out");
  const bool whole_sets = fb->layout == SVOH_BATCH_WHOLE_SETS;
  SVOH_REQUIRE(ctx, fb->layout == SVOH_BATCH_UNITS || whole_sets, "svoh_feature_batch::layout: SVOH_BATCH_UNITS or SVOH_BATCH_WHOLE_SETS");
  SVOH_REQUIRE(ctx, !whole_sets || (seeds && st.resident), "SVOH_BATCH_WHOLE_SETS: seed batches staged with SVOH_STAGE_RESIDENT_COLUMNS");
  if (st.resident)
    SVOH_REQUIRE(ctx, fb->feature_index == at(st.o_fidx) && !fb->px && !fb->f && !fb->grad && !fb->level,
                 "a batch staged with SVOH_STAGE_RESIDENT_COLUMNS names its features by feature_index (the staged pointer); px / f / grad / level are NULL");
  else
    SVOH_REQUIRE(ctx, !fb->feature_index && fb->px == at(st.o_px) && fb->f == at(st.o_f) && fb->grad == at(st.o_grad) && fb->level == at(st.o_level),
                 "a staged batch's feature arrays must be the pointers svoh_matcher_stage handed out");
  if (seeds) {
    SVOH_REQUIRE(ctx, dopt && state == at(st.o_state) && success == at(st.o_success), "a staged seed batch's state / success must be the staged pointers");
    SVOH_REQUIRE(ctx, !result || result == at(st.o_result), "result: not the staged pointer");
    SVOH_REQUIRE(ctx, st.want_outputs || (!px_cur && !f_cur && !search_level && !A_cur_ref), "the block was staged without match outputs");
    if (st.want_outputs)
      SVOH_REQUIRE(ctx, (!px_cur || px_cur == at(st.o_pxcur)) && (!f_cur || f_cur == at(st.o_fcur)) && (!search_level || search_level == at(st.o_slevel)) &&
                            (!A_cur_ref || A_cur_ref == at(st.o_A)), "match outputs: not the staged pointers");
  } else {
    SVOH_REQUIRE(ctx, depth == at(st.o_depth) && px_cur == at(st.o_pxcur) && result == at(st.o_result), "a staged direct batch's depth / px_cur / result must be the staged pointers");
    SVOH_REQUIRE(ctx, (!f_cur || f_cur == at(st.o_fcur)) && (!search_level || search_level == at(st.o_slevel)) && (!h_inv || h_inv == at(st.o_hinv)) &&
                          (!A_cur_ref || A_cur_ref == at(st.o_A)), "match outputs: not the staged pointers");
  }
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  DevFrameView* views = reinterpret_cast<DevFrameView*>(h + st.o_views);
  int ref_levels = 0, max_w = 1, max_h = 1;
  bool pose_from_results = false;
  // the same frames as the last staged batch asked for, none released since: that batch's table (a reprojection's direct and
  // seed batch name the same keyframes and current frames; a table of 60 views is 10 us of look-ups on the caller's thread)
  const size_t key_bytes = sizeof(svoh_frame_view) * (size_t)(n_ref_frames + n_cur);
  bool reuse = ctx->staged_views_generation == ctx->handle_generation && ctx->staged_views_key.size() == key_bytes + sizeof(int) &&
               memcmp(ctx->staged_views_key.data(), &n_ref_frames, sizeof(int)) == 0 &&
               memcmp(ctx->staged_views_key.data() + sizeof(int), ref_frames, sizeof(svoh_frame_view) * (size_t)n_ref_frames) == 0 &&
               memcmp(ctx->staged_views_key.data() + sizeof(int) + sizeof(svoh_frame_view) * (size_t)n_ref_frames, cur_frame, sizeof(svoh_frame_view) * (size_t)n_cur) == 0;
  for (int k = 0; k < n_cur && reuse; ++k) reuse = cur_frame[k].pose_result_index_plus1 == 0;   // (those go through the checks below)
  if (reuse) {
    memcpy(views, ctx->staged_views_resolved.data(), sizeof(DevFrameView) * (size_t)(n_ref_frames + n_cur));
    ref_levels = ctx->staged_views_ref_levels; max_w = ctx->staged_views_max_w; max_h = ctx->staged_views_max_h;
    if (st.resident) for (int k = 0; k < n_ref_frames; ++k) SVOH_REQUIRE(ctx, ref_frames[k].features != 0, "a reference frame of a batch with feature_index has no resident columns (svoh_frame_view::features)");
  } else {
    for (int k = 0; k < n_ref_frames; ++k) {
      const int rc = fill_view(ctx, ref_frames[k], &views[k], "reference frame");
      if (rc != SVOH_OK) return rc;
      ref_levels = views[k].n_levels > ref_levels ? views[k].n_levels : ref_levels;
      max_w = views[k].lv[0].w > max_w ? views[k].lv[0].w : max_w; max_h = views[k].lv[0].h > max_h ? views[k].lv[0].h : max_h;
      SVOH_REQUIRE(ctx, !st.resident || ref_frames[k].features != 0, "a reference frame of a batch with feature_index has no resident columns (svoh_frame_view::features)");
    }
    for (int k = 0; k < n_cur; ++k) {
      const int rc = fill_view(ctx, cur_frame[k], &views[n_ref_frames + k], "current frame", seeds && ctx->in_pose_hook);
      if (rc != SVOH_OK) return rc;
      SVOH_REQUIRE(ctx, views[n_ref_frames + k].n_levels >= ref_levels, "current frame has fewer pyramid levels than a reference frame");
      const int pr = views[n_ref_frames + k].pose_result_index_plus1;
      SVOH_REQUIRE(ctx, pr >= 0 && pr <= ctx->n_pose_results, "pose_result_index_plus1: the pose batch in flight has no such result");
      pose_from_results = pose_from_results || pr > 0;
    }
    ctx->staged_views_generation = ~0ull;
    if (!pose_from_results) {
      ctx->staged_views_key.resize(key_bytes + sizeof(int));
      memcpy(ctx->staged_views_key.data(), &n_ref_frames, sizeof(int));
      memcpy(ctx->staged_views_key.data() + sizeof(int), ref_frames, sizeof(svoh_frame_view) * (size_t)n_ref_frames);
      memcpy(ctx->staged_views_key.data() + sizeof(int) + sizeof(svoh_frame_view) * (size_t)n_ref_frames, cur_frame, sizeof(svoh_frame_view) * (size_t)n_cur);
      ctx->staged_views_resolved.assign(reinterpret_cast<const uint8_t*>(views), reinterpret_cast<const uint8_t*>(views) + sizeof(DevFrameView) * (size_t)(n_ref_frames + n_cur));
      ctx->staged_views_ref_levels = ref_levels; ctx->staged_views_max_w = max_w; ctx->staged_views_max_h = max_h;
      ctx->staged_views_generation = ctx->handle_generation;
    }
  }
  if (whole_sets) {   // the units are the reference frames' features, frame after frame: where each frame's units begin
    long long begin = 0;
    for (int k = 0; k < n_ref_frames; ++k) { views[k].unit_begin = (int32_t)begin; begin += views[k].feat_n; }
    SVOH_REQUIRE(ctx, begin == n, "SVOH_BATCH_WHOLE_SETS: the reference frames' resident sets do not add up to the batch's n");
  }
  ctx->matcher_deferred_used[kind] = true;
  st.valid = false;   // consumed: the outputs stay readable, a second batch needs a new svoh_matcher_stage
  SVOH_HIP_TRY(ctx, svoh_copy_to_device(ctx, d, h, st.in_total));
  int g8 = n <= kG8MaxUnits ? 1 : 2;
  g8 = SvohKnobs::or_default(ctx->knobs.matcher_g8, g8);
  if (g8 < 0 || g8 > 3) g8 = 0;
  if (g8 == 3) g8 = 1;
  MatcherArgs a;
  memset(&a, 0, sizeof a);
  a.ref_frames = reinterpret_cast<const DevFrameView*>(d + st.o_views);
  a.cur_frame = a.ref_frames + n_ref_frames;
  a.mopt = *mopt;
  if (dopt) a.dopt = *dopt;
  a.n = n; a.n_ref_frames = n_ref_frames; a.n_cur_frames = n_cur;
  a.whole_sets = whole_sets ? 1 : 0;
  a.ref_frame_idx = reinterpret_cast<const int32_t*>(d + st.o_idx);
  a.cur_frame_idx = reinterpret_cast<const int32_t*>(d + st.o_cidx);
  a.px = reinterpret_cast<const double*>(d + st.o_px); a.f = reinterpret_cast<const double*>(d + st.o_f);
  a.grad = reinterpret_cast<const double*>(d + st.o_grad); a.level = reinterpret_cast<const int32_t*>(d + st.o_level);
  a.type = d + st.o_type;
  a.result = reinterpret_cast<int32_t*>(d + st.o_result);
  a.success = d + st.o_success;
  if (st.want_outputs) {
    a.f_cur = reinterpret_cast<double*>(d + st.o_fcur); a.search_level = reinterpret_cast<int32_t*>(d + st.o_slevel);
    a.h_inv = reinterpret_cast<double*>(d + st.o_hinv); a.A_cur_ref = reinterpret_cast<double*>(d + st.o_A);
  }
  if (seeds) {
    a.state = reinterpret_cast<double*>(d + st.o_state);
    a.px_cur = st.want_outputs ? reinterpret_cast<double*>(d + st.o_pxcur) : nullptr;
  } else {
    a.depth = reinterpret_cast<const double*>(d + st.o_depth);
    a.px_cur = reinterpret_cast<double*>(d + st.o_pxcur);
  }
  svoh_ctx::DeferredLaunch& dl = ctx->matcher_deferred_launch[kind];
  dl.args.assign(reinterpret_cast<const uint8_t*>(&a), reinterpret_cast<const uint8_t*>(&a) + sizeof a);
  dl.n = n; dl.g8 = g8; dl.valid = true; dl.max_w = max_w; dl.max_h = max_h;
  dl.d_block = d; dl.out_off = st.in_total; dl.out_bytes = st.o_nsucc - st.in_total;
  dl.d2h_dst = h + st.back_from; dl.d2h_src = d + st.back_from; dl.d2h_bytes = st.o_nsucc - st.back_from;
  dl.views_h = h + st.o_views; dl.views_d = d + st.o_views; dl.n_ref = n_ref_frames; dl.n_cur = n_cur;
  dl.cur_frame_handle = cur_frame[0].frame;
  dl.pose_from_results = pose_from_results; dl.d_pose_results = pose_from_results ? ctx->d_pose_results : nullptr; dl.n_pose_results = ctx->n_pose_results;
  dl.d_fidx = st.resident ? d + st.o_fidx : nullptr;
  dl.seed_block_staged = seeds;
  if (seeds && n_success) ctx->matcher_pending_counts.push_back({ n_success, h + st.o_success, n });
  return SVOH_OK;
}

// ---- the candidate projections of many current frames in one launch (svoh_project_candidates_stage / ...) ----------
struct MultiCandidateArgs {
  const svoh_candidate_job* jobs;
  const svoh_align_result* align_results;   // d_results
  const uint32_t* result_dev_index;         // per job: where its alignment result lives in align_results (job.align_result_index >= 0)
  const svoh_se3* T_world_kf;
  const int32_t* job;
  const uint8_t* kind;
  const int32_t* kf;
  const double* v;
  const double* mu;
  double* px;
  uint8_t* visible;
  int n, n_jobs;
};

// per point exactly the arithmetic of project_candidates_kernel
__global__ __launch_bounds__(256) void project_candidates_multi_kernel(const MultiCandidateArgs a)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int j = a.job[i];
  if ((unsigned)j >= (unsigned)a.n_jobs) { a.px[2 * i] = 0.0; a.px[2 * i + 1] = 0.0; a.visible[i] = 0; return; }
  const svoh_candidate_job& jb = a.jobs[j];
  Rigid T_f_w = load_rigid(jb.T_f_w_or_T_cam_imu);
  if (jb.align_result_index >= 0) T_f_w = mul(mul(T_f_w, load_rigid(a.align_results[a.result_dev_index[j]].T_icur_iref)), load_rigid(jb.T_imu_world_ref));
  const CamModel cm = load_camera(jb.cam);
  Vec3 xyz = { a.v[3 * i], a.v[3 * i + 1], a.v[3 * i + 2] };
  bool ok = true;
  if (a.kind[i]) {
    const int k = a.kf[i];
    ok = k >= 0 && k < jb.n_kf;
    if (ok) {
      const double depth = 1.0 / a.mu[i];                        // seed::getDepth (seed.h:110-113)
      const Vec3 in_f = { xyz.x * depth, xyz.y * depth, xyz.z * depth };
      xyz = transform(load_rigid(a.T_world_kf[jb.kf_begin + k]), in_f);
    }
  }
  double u = 0.0, v = 0.0;
  if (ok) {
    const Vec3 xyz_f = transform(T_f_w, xyz);
    const Vec3 f_tl = back_project3(cm, 0.0, 0.0);
    const double min_cos = f_tl.z / sqrt(f_tl.x * f_tl.x + f_tl.y * f_tl.y + f_tl.z * f_tl.z);
    const double cur_cos = xyz_f.z / sqrt(xyz_f.x * xyz_f.x + xyz_f.y * xyz_f.y + xyz_f.z * xyz_f.z);
    ok = !(cur_cos < min_cos);
    if (ok) {
      project3(cm, xyz_f, u, v);
      ok = u >= 0.0 && v >= 0.0 && u < (double)jb.cam.width && v < (double)jb.cam.height;
      if (ok) {
        const int pxi0 = (int)u, pxi1 = (int)v;
        ok = pxi0 >= 8 && pxi1 >= 8 && pxi0 < jb.cam.width - 8 && pxi1 < jb.cam.height - 8;
      }
    }
  }
  a.px[2 * i] = u; a.px[2 * i + 1] = v;
  a.visible[i] = ok ? 1 : 0;
}

// ---- svoh_select_matches_batch: the loop of matchCandidates (reprojector.cpp:342-382) over given matches, one workgroup per list --
// Sequentially: candidate k is tried iff its cell is free when the loop gets there; a success takes the cell; the loop ends with the
// success that fills the frame.  In parallel: W(c) = the smallest k among the successes of cell c (cells free at the start only);
// the winners are the k with k == W(cell[k]); the loop ends at K = the m-th winner in order of k, m = max(1, max_n - n_before)
// (or runs through the list); k is tried iff k <= K, its cell was free at the start and k <= W(cell[k]).
struct SelectArgs {
  const int32_t* begin; const int32_t* cell; const uint8_t* success;
  int n_cells; uint8_t* occupancy; const int32_t* max_n; int32_t* num_features;
  uint8_t* visited; int32_t* n_trials; int32_t* n_matches; int32_t* n_consumed;
  int32_t* first_success;   // scratch: n_lists x n_cells
};
__global__ __launch_bounds__(256) void select_matches_kernel(const SelectArgs a)
{
  const int l = blockIdx.x, tid = threadIdx.x;
  const int lo = a.begin[l], n = a.begin[l + 1] - lo;
  const int32_t* cell = a.cell + lo;
  const uint8_t* success = a.success + lo;
  uint8_t* visited = a.visited + lo;
  uint8_t* occ = a.occupancy + (size_t)l * a.n_cells;
  int32_t* W = a.first_success + (size_t)l * a.n_cells;
  __shared__ int s_cnt[256], s_stop, s_trials, s_matches;
  // W is written by atomics (at the L2) and read back by every wave of the workgroup: agent-scope loads, past the CU's vector L1
  auto Wat = [&](int c) { return __hip_atomic_load(&W[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  for (int c = tid; c < a.n_cells; c += 256) __hip_atomic_store(&W[c], 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) { s_stop = n - 1; s_trials = 0; s_matches = 0; }
  __syncthreads();
  // every thread owns a contiguous run of the list, so that counts per thread are counts in visiting order
  const int per = (n + 255) / 256, k0 = tid * per, k1 = k0 + per < n ? k0 + per : n;
  for (int k = k0; k < k1; ++k) {
    const int c = cell[k];
    if (success[k] && c >= 0 && c < a.n_cells && !occ[c]) atomicMin(&W[c], k);
  }
  __threadfence_block();
  __syncthreads();
  int winners = 0;
  for (int k = k0; k < k1; ++k) {
    const int c = cell[k];
    winners += (c >= 0 && c < a.n_cells && Wat(c) == k) ? 1 : 0;
  }
  s_cnt[tid] = winners;
  __syncthreads();
  if (tid == 0) {   // where the m-th winner lies: a scan of 256 counts by one lane (a list has a few thousand candidates)
    const int n_before = a.num_features[l], max_n = a.max_n[l];
    int m = max_n - n_before;
    if (m < 1) m = 1;
    int acc = 0, owner = -1;
    for (int t = 0; t < 256; ++t) { if (acc + s_cnt[t] >= m) { owner = t; break; } acc += s_cnt[t]; }
    if (owner >= 0) {
      int need = m - acc;
      const int b0 = owner * per, b1 = b0 + per < n ? b0 + per : n;
      for (int k = b0; k < b1; ++k) {
        const int c = cell[k];
        if (c >= 0 && c < a.n_cells && Wat(c) == k && --need == 0) { s_stop = k; break; }
      }
    }
  }
  __syncthreads();
  const int K = s_stop;
  int trials = 0, matches = 0;
  for (int k = k0; k < k1; ++k) {
    const int c = cell[k];
    const bool in_grid = c >= 0 && c < a.n_cells;
    const int w = in_grid ? Wat(c) : 0x7fffffff;
    const bool v = k <= K && in_grid && !occ[c] && k <= w;
    visited[k] = v ? 1 : 0;
    trials += v ? 1 : 0;
    matches += (v && w == k) ? 1 : 0;
  }
  if (trials) atomicAdd(&s_trials, trials);
  if (matches) atomicAdd(&s_matches, matches);
  __syncthreads();   // (every read of occ[] above precedes the writes below)
  for (int c = tid; c < a.n_cells; c += 256) if (Wat(c) <= K) occ[c] = 1;
  if (tid == 0) {
    a.n_trials[l] = s_trials; a.n_matches[l] = s_matches;
    a.n_consumed[l] = n > 0 ? K + 1 : 0;
    a.num_features[l] += s_matches;
  }
}

// The ranges form (svoh_project_candidates_stage_ranges): point i is feature i - point_begin of the keyframe whose range holds it;
// a seed's bearing vector comes from the keyframe's resident f column.  Same arithmetic per point.
struct DevCandidateRange { const double* f; int32_t n_feat; int32_t point_begin, n_points; int32_t job; int32_t pad_; };
struct RangeCandidateArgs {
  const svoh_candidate_job* jobs;
  const svoh_align_result* align_results;
  const uint32_t* result_dev_index;
  const svoh_se3* T_world_kf;
  const DevCandidateRange* ranges;
  const uint8_t* kind;
  const double* v;      // landmarks only (NULL when the launch has none)
  const double* mu;
  const int32_t* mu_unit;       // NULL, or per point: >= 0 = the inverse depth of that unit of the seed batch in flight
  const double* unit_state; int n_units;
  double* px;
  uint8_t* visible;
  int n, n_jobs, n_ranges;
};

__global__ __launch_bounds__(256) void project_candidates_ranges_kernel(const RangeCandidateArgs a)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  // the last range that begins at or before i
  int lo = 0, hi = a.n_ranges;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.ranges[mid].point_begin <= i) lo = mid; else hi = mid; }
  const DevCandidateRange rg = a.ranges[lo];
  const int jf = i - rg.point_begin;
  const int j = rg.job;
  const bool in_range = jf >= 0 && jf < rg.n_points && (unsigned)j < (unsigned)a.n_jobs;
  if (!in_range) { a.px[2 * i] = 0.0; a.px[2 * i + 1] = 0.0; a.visible[i] = 0; return; }
  const svoh_candidate_job& jb = a.jobs[j];
  Rigid T_f_w = load_rigid(jb.T_f_w_or_T_cam_imu);
  if (jb.align_result_index >= 0) T_f_w = mul(mul(T_f_w, load_rigid(a.align_results[a.result_dev_index[j]].T_icur_iref)), load_rigid(jb.T_imu_world_ref));
  const CamModel cm = load_camera(jb.cam);
  Vec3 xyz = { 0.0, 0.0, 0.0 };
  bool ok = true;
  if (a.kind[i]) {
    ok = lo >= jb.kf_begin && lo < jb.kf_begin + jb.n_kf && jf < rg.n_feat;
    if (ok) {
      double mu_i = a.mu[i];
      if (a.mu_unit) { const int u = a.mu_unit[i]; if (u >= 0 && u < a.n_units) mu_i = a.unit_state[4 * (size_t)u]; }
      const double depth = 1.0 / mu_i;                           // seed::getDepth (seed.h:110-113)
      const Vec3 in_f = { rg.f[3 * jf] * depth, rg.f[3 * jf + 1] * depth, rg.f[3 * jf + 2] * depth };
      xyz = transform(load_rigid(a.T_world_kf[lo]), in_f);
    }
  } else {
    ok = a.v != nullptr;
    if (ok) { xyz.x = a.v[3 * i]; xyz.y = a.v[3 * i + 1]; xyz.z = a.v[3 * i + 2]; }
  }
  double u = 0.0, v = 0.0;
  if (ok) {
    const Vec3 xyz_f = transform(T_f_w, xyz);
    const Vec3 f_tl = back_project3(cm, 0.0, 0.0);
    const double min_cos = f_tl.z / sqrt(f_tl.x * f_tl.x + f_tl.y * f_tl.y + f_tl.z * f_tl.z);
    const double cur_cos = xyz_f.z / sqrt(xyz_f.x * xyz_f.x + xyz_f.y * xyz_f.y + xyz_f.z * xyz_f.z);
    ok = !(cur_cos < min_cos);
    if (ok) {
      project3(cm, xyz_f, u, v);
      ok = u >= 0.0 && v >= 0.0 && u < (double)jb.cam.width && v < (double)jb.cam.height;
      if (ok) {
        const int pxi0 = (int)u, pxi1 = (int)v;
        ok = pxi0 >= 8 && pxi1 >= 8 && pxi0 < jb.cam.width - 8 && pxi1 < jb.cam.height - 8;
      }
    }
  }
  a.px[2 * i] = u; a.px[2 * i + 1] = v;
  a.visible[i] = ok ? 1 : 0;
}

}  // namespace svoh

using namespace svoh;

extern "C" {

int svoh_select_matches_batch(svoh_ctx* ctx, int n_lists, const int32_t* begin, const int32_t* cell, const uint8_t* success,
                              int n_cells, uint8_t* occupancy, const int32_t* max_n_features, int32_t* num_features,
                              uint8_t* visited, int32_t* n_trials, int32_t* n_matches, int32_t* n_consumed)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, n_lists >= 0 && n_cells >= 1 && n_cells <= (1 << 20), "bad arguments");
  if (n_lists == 0) return SVOH_OK;
  SVOH_REQUIRE(ctx, begin && occupancy && max_n_features && num_features && n_trials && n_matches && n_consumed, "NULL argument");
  SVOH_REQUIRE(ctx, n_lists <= (1 << 16) && begin[0] == 0, "too many lists, or begin[0] != 0");
  for (int l = 0; l < n_lists; ++l) {
    SVOH_REQUIRE(ctx, begin[l + 1] >= begin[l], "begin must not decrease");
    SVOH_REQUIRE(ctx, max_n_features[l] > 0, "max_n_features must be > 0 (with 0 the reference's loop ignores the grid: replay it on the host)");
  }
  const size_t n = (size_t)begin[n_lists];
  SVOH_REQUIRE(ctx, n <= ((size_t)1 << 26) && (n == 0 || (cell && success && visited)), "NULL candidate array, or too many candidates");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
  const size_t nl = (size_t)n_lists, nc = (size_t)n_cells;
  // [begin | max_n | num_features | cell | success | occupancy]  up;  [num_features | occupancy | visited | trials | matches | consumed]  back
  const size_t o_begin = 0, o_max = al(4 * (nl + 1)), o_cell = o_max + al(4 * nl), o_succ = o_cell + al(4 * n), in_only = o_succ + al(n);
  const size_t o_nf = in_only, o_occ = o_nf + al(4 * nl), in_total = o_occ + al(nl * nc);
  const size_t o_vis = in_total, o_tr = o_vis + al(n), o_ma = o_tr + al(4 * nl), o_co = o_ma + al(4 * nl), back_end = o_co + al(4 * nl);
  const size_t o_w = back_end, total = o_w + al(4 * nl * nc);
  SVOH_HIP_TRY(ctx, ctx->h_scratch1.reserve(back_end));
  SVOH_HIP_TRY(ctx, ctx->d_scratch1.reserve(total));
  uint8_t* h = static_cast<uint8_t*>(ctx->h_scratch1.ptr);
  uint8_t* d = static_cast<uint8_t*>(ctx->d_scratch1.ptr);
  memcpy(h + o_begin, begin, 4 * (nl + 1)); memcpy(h + o_max, max_n_features, 4 * nl); memcpy(h + o_nf, num_features, 4 * nl);
  if (n) { memcpy(h + o_cell, cell, 4 * n); memcpy(h + o_succ, success, n); }
  memcpy(h + o_occ, occupancy, nl * nc);
  SVOH_HIP_TRY(ctx, svoh_copy_to_device(ctx, d, h, in_total));
  SelectArgs a;
  a.begin = reinterpret_cast<const int32_t*>(d + o_begin); a.cell = reinterpret_cast<const int32_t*>(d + o_cell); a.success = d + o_succ;
  a.n_cells = n_cells; a.occupancy = d + o_occ; a.max_n = reinterpret_cast<const int32_t*>(d + o_max);
  a.num_features = reinterpret_cast<int32_t*>(d + o_nf);
  a.visited = d + o_vis; a.n_trials = reinterpret_cast<int32_t*>(d + o_tr); a.n_matches = reinterpret_cast<int32_t*>(d + o_ma);
  a.n_consumed = reinterpret_cast<int32_t*>(d + o_co); a.first_success = reinterpret_cast<int32_t*>(d + o_w);
  hipLaunchKernelGGL(select_matches_kernel, dim3((unsigned)n_lists), dim3(256), 0, ctx->stream, a);
  SVOH_HIP_TRY(ctx, hipGetLastError());
  SVOH_HIP_TRY(ctx, svoh_copy_to_host(ctx, h + o_nf, d + o_nf, back_end - o_nf));
  SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(num_features, h + o_nf, 4 * nl); memcpy(occupancy, h + o_occ, nl * nc);
  if (n) memcpy(visited, h + o_vis, n);
  memcpy(n_trials, h + o_tr, 4 * nl); memcpy(n_matches, h + o_ma, 4 * nl); memcpy(n_consumed, h + o_co, 4 * nl);
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_project_candidates_enqueue(svoh_ctx* ctx, const svoh_camera* cam, const svoh_se3* T_f_w_or_T_cam_imu, const svoh_se3* T_imu_world_ref,
                                    int align_result_index, int n_kf, const svoh_se3* T_world_kf, int n, const uint8_t* kind,
                                    const int32_t* kf, const double* v, const double* mu)
try {
  return enqueue_candidates(ctx, cam, T_f_w_or_T_cam_imu, T_imu_world_ref, align_result_index, n_kf, T_world_kf, n, kind, kf, v, mu);
} SVOH_ABI_CATCH(ctx)

int svoh_project_candidates_collect(svoh_ctx* ctx, int n, double* px, uint8_t* visible)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, px && visible && n >= 1 && n == ctx->cand_pending_n, "n is not the number of points of the queued candidate projection");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // returns at once when a fetch of the alignment in front has waited already
  const uint8_t* h = static_cast<const uint8_t*>(ctx->h_cand.ptr) + ctx->cand_out_off;
  memcpy(px, h, sizeof(double) * 2 * (size_t)n);
  memcpy(visible, h + ((sizeof(double) * 2 * (size_t)n + 63) & ~(size_t)63), (size_t)n);
  ctx->cand_pending_n = 0;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_project_candidates(svoh_ctx* ctx, const svoh_camera* cam, const svoh_se3* T_f_w, int n_kf, const svoh_se3* T_world_kf, int n,
                            const uint8_t* kind, const int32_t* kf, const double* v, const double* mu, double* px, uint8_t* visible)
try {
  int rc = enqueue_candidates(ctx, cam, T_f_w, nullptr, -1, n_kf, T_world_kf, n, kind, kf, v, mu);
  if (rc != SVOH_OK) return rc;
  return svoh_project_candidates_collect(ctx, n, px, visible);
} SVOH_ABI_CATCH(ctx)

static int stage_candidates(svoh_ctx* ctx, int n_jobs, int n_kf_total, int n_points_total, svoh_candidate_stage_t* out, bool ranges)
{
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, out && n_jobs >= 1 && n_jobs <= 4096 && n_kf_total >= 0 && n_kf_total <= (1 << 20) && n_points_total >= 1 && n_points_total <= (1 << 24), "bad arguments");
  SVOH_REQUIRE(ctx, ctx->cand_stage.state != 2, "a staged candidate projection is in flight: svoh_project_candidates_wait first");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  svoh_ctx::CandStage& st = ctx->cand_stage;
  size_t total = 0;
  auto add = [&](size_t bytes) { const size_t o = total; total = (total + bytes + 63) & ~(size_t)63; return o; };
  const size_t np = (size_t)n_points_total;
  st.n_jobs = n_jobs; st.n_kf = n_kf_total; st.n_points = n_points_total;
  st.o_jobs = add(sizeof(svoh_candidate_job) * (size_t)n_jobs + sizeof(uint32_t) * (size_t)n_jobs);   // + the jobs' device result indices
  st.o_kf = add(sizeof(svoh_se3) * (size_t)(n_kf_total > 0 ? n_kf_total : 1));
  st.ranges = ranges;
  memset(out, 0, sizeof *out);
  if (ranges) {
    SVOH_REQUIRE(ctx, n_kf_total >= 1, "the ranges form needs a keyframe table");
    // [jobs | T_world_kf | device ranges | kind | mu | v]: v last, uploaded only when a landmark is among the points; the ranges
    // as the caller writes them (handles) stay on the host
    st.o_dev_ranges = add(sizeof(DevCandidateRange) * (size_t)n_kf_total);
    st.o_kind = add(np); st.o_mu = add(8 * np); st.o_mu_unit = add(4 * np);
    st.in_total_without_v = total;
    st.o_v = add(24 * np);
    st.in_total = total;
    st.o_px = add(16 * np); st.o_vis = add(np);
    st.o_ranges = add(sizeof(svoh_candidate_range) * (size_t)n_kf_total);
    st.o_job = st.o_idx = 0;
  } else {
    st.o_job = add(4 * np); st.o_kind = add(np); st.o_idx = add(4 * np); st.o_v = add(24 * np); st.o_mu = add(8 * np);
    st.in_total = total;
    st.o_px = add(16 * np); st.o_vis = add(np);
  }
  st.total = total;
  SVOH_HIP_TRY(ctx, ctx->h_cand_multi.reserve(total));
  SVOH_HIP_TRY(ctx, ctx->d_cand_multi.reserve(total));
  uint8_t* h = static_cast<uint8_t*>(ctx->h_cand_multi.ptr);
  out->jobs = reinterpret_cast<svoh_candidate_job*>(h + st.o_jobs);
  out->T_world_kf = reinterpret_cast<svoh_se3*>(h + st.o_kf);
  out->kind = h + st.o_kind;
  if (ranges) {
    out->ranges = reinterpret_cast<svoh_candidate_range*>(h + st.o_ranges);
    out->mu_unit = reinterpret_cast<int32_t*>(h + st.o_mu_unit);
    memset(out->mu_unit, 0xff, 4 * np);   // -1 everywhere
  }
  else { out->job = reinterpret_cast<int32_t*>(h + st.o_job); out->kf = reinterpret_cast<int32_t*>(h + st.o_idx); }
  out->v = reinterpret_cast<double*>(h + st.o_v); out->mu = reinterpret_cast<double*>(h + st.o_mu);
  out->px = reinterpret_cast<double*>(h + st.o_px); out->visible = h + st.o_vis;
  st.state = 1;
  return SVOH_OK;
}

int svoh_project_candidates_stage(svoh_ctx* ctx, int n_jobs, int n_kf_total, int n_points_total, svoh_candidate_stage_t* out)
try {
  return stage_candidates(ctx, n_jobs, n_kf_total, n_points_total, out, false);
} SVOH_ABI_CATCH(ctx)

int svoh_project_candidates_stage_ranges(svoh_ctx* ctx, int n_jobs, int n_kf_total, int n_points_total, svoh_candidate_stage_t* out)
try {
  return stage_candidates(ctx, n_jobs, n_kf_total, n_points_total, out, true);
} SVOH_ABI_CATCH(ctx)

static int enqueue_staged_candidates(svoh_ctx* ctx, bool with_units)
{
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  svoh_ctx::CandStage& st = ctx->cand_stage;
  SVOH_REQUIRE(ctx, !with_units || (st.ranges && ctx->seed_block.valid),
               "mu_unit: the ranges form only, and a seed batch must have been sent off on this context (its block not laid out anew since)");
  SVOH_REQUIRE(ctx, st.state == 1, "nothing staged (svoh_project_candidates_stage)");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint8_t* h = static_cast<uint8_t*>(ctx->h_cand_multi.ptr);
  uint8_t* d = static_cast<uint8_t*>(ctx->d_cand_multi.ptr);
  svoh_candidate_job* jobs = reinterpret_cast<svoh_candidate_job*>(h + st.o_jobs);
  uint32_t* dev_idx = reinterpret_cast<uint32_t*>(jobs + st.n_jobs);
  bool any_result = false;
  for (int j = 0; j < st.n_jobs; ++j) {
    const svoh_candidate_job& jb = jobs[j];
    SVOH_REQUIRE(ctx, jb.cam.distortion == SVOH_DISTORTION_NONE || jb.cam.distortion == SVOH_DISTORTION_RADTAN, "unsupported distortion model");
    SVOH_REQUIRE(ctx, jb.n_kf >= 0 && jb.kf_begin >= 0 && (int64_t)jb.kf_begin + jb.n_kf <= st.n_kf, "a job's keyframe range leaves the table");
    SVOH_REQUIRE(ctx, jb.n_points >= 0 && jb.point_begin >= 0 && (int64_t)jb.point_begin + jb.n_points <= st.n_points, "a job's point range leaves the arrays");
    dev_idx[j] = 0;
    if (jb.align_result_index >= 0) {
      SVOH_REQUIRE(ctx, (size_t)jb.align_result_index < ctx->align_pending_results && (size_t)jb.align_result_index < ctx->align_result_dev_index.size() && ctx->d_results.ptr,
                   "no queued alignment result with this index");
      dev_idx[j] = ctx->align_result_dev_index[(size_t)jb.align_result_index];
      any_result = true;
    }
  }
  if (st.ranges) {
    const svoh_candidate_range* rg = reinterpret_cast<const svoh_candidate_range*>(h + st.o_ranges);
    DevCandidateRange* dr = reinterpret_cast<DevCandidateRange*>(h + st.o_dev_ranges);
    int64_t at = 0;
    for (int k = 0; k < st.n_kf; ++k) {
      SVOH_REQUIRE(ctx, rg[k].point_begin == at && rg[k].n_points >= 0 && at + rg[k].n_points <= st.n_points, "the ranges must lie back to back inside the point arrays");
      SVOH_REQUIRE(ctx, rg[k].job >= 0 && rg[k].job < st.n_jobs, "a range's job is out of range");
      auto it = ctx->feature_sets.find(rg[k].features);
      if (it == ctx->feature_sets.end()) return set_error(ctx, SVOH_ERR_BAD_HANDLE, "range %d: unknown feature-set handle %llu", k, (unsigned long long)rg[k].features);
      dr[k].f = it->second.f; dr[k].n_feat = it->second.n; dr[k].point_begin = rg[k].point_begin; dr[k].n_points = rg[k].n_points; dr[k].job = rg[k].job; dr[k].pad_ = 0;
      at += rg[k].n_points;
    }
    SVOH_REQUIRE(ctx, at == st.n_points, "the ranges do not cover the staged points");
    const bool any_landmark = memchr(h + st.o_kind, 0, (size_t)st.n_points) != nullptr;
    SVOH_HIP_TRY(ctx, svoh_copy_to_device(ctx, d, h, any_landmark ? st.in_total : st.in_total_without_v));
    RangeCandidateArgs a;
    memset(&a, 0, sizeof a);
    a.jobs = reinterpret_cast<const svoh_candidate_job*>(d + st.o_jobs);
    a.result_dev_index = reinterpret_cast<const uint32_t*>(a.jobs + st.n_jobs);
    a.align_results = any_result ? static_cast<const svoh_align_result*>(ctx->d_results.ptr) : nullptr;
    a.T_world_kf = reinterpret_cast<const svoh_se3*>(d + st.o_kf);
    a.ranges = reinterpret_cast<const DevCandidateRange*>(d + st.o_dev_ranges);
    a.kind = d + st.o_kind; a.mu = reinterpret_cast<const double*>(d + st.o_mu);
    if (with_units) { a.mu_unit = reinterpret_cast<const int32_t*>(d + st.o_mu_unit); a.unit_state = ctx->seed_block.state; a.n_units = ctx->seed_block.n; }
    a.v = any_landmark ? reinterpret_cast<const double*>(d + st.o_v) : nullptr;
    a.px = reinterpret_cast<double*>(d + st.o_px); a.visible = d + st.o_vis;
    a.n = st.n_points; a.n_jobs = st.n_jobs; a.n_ranges = st.n_kf;
    hipLaunchKernelGGL(project_candidates_ranges_kernel, dim3((unsigned)((st.n_points + 255) / 256)), dim3(256), 0, ctx->stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(ctx, SVOH_ERR_HIP, "project_candidates_ranges launch failed: %s", hipGetErrorString(e));
    SVOH_HIP_TRY(ctx, svoh_copy_to_host(ctx, h + st.o_px, d + st.o_px, st.o_ranges - st.o_px));
    st.state = 2;
    return SVOH_OK;
  }
  SVOH_HIP_TRY(ctx, svoh_copy_to_device(ctx, d, h, st.in_total));
  MultiCandidateArgs a;
  memset(&a, 0, sizeof a);
  a.jobs = reinterpret_cast<const svoh_candidate_job*>(d + st.o_jobs);
  a.result_dev_index = reinterpret_cast<const uint32_t*>(a.jobs + st.n_jobs);
  a.align_results = any_result ? static_cast<const svoh_align_result*>(ctx->d_results.ptr) : nullptr;
  a.T_world_kf = reinterpret_cast<const svoh_se3*>(d + st.o_kf);
  a.job = reinterpret_cast<const int32_t*>(d + st.o_job); a.kind = d + st.o_kind; a.kf = reinterpret_cast<const int32_t*>(d + st.o_idx);
  a.v = reinterpret_cast<const double*>(d + st.o_v); a.mu = reinterpret_cast<const double*>(d + st.o_mu);
  a.px = reinterpret_cast<double*>(d + st.o_px); a.visible = d + st.o_vis;
  a.n = st.n_points; a.n_jobs = st.n_jobs;
  hipLaunchKernelGGL(project_candidates_multi_kernel, dim3((unsigned)((st.n_points + 255) / 256)), dim3(256), 0, ctx->stream, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(ctx, SVOH_ERR_HIP, "project_candidates_multi launch failed: %s", hipGetErrorString(e));
  SVOH_HIP_TRY(ctx, svoh_copy_to_host(ctx, h + st.o_px, d + st.o_px, st.total - st.o_px));
  st.state = 2;
  return SVOH_OK;
}

int svoh_project_candidates_enqueue_staged(svoh_ctx* ctx)
try {
  return enqueue_staged_candidates(ctx, false);
} SVOH_ABI_CATCH(ctx)

int svoh_project_candidates_enqueue_staged_units(svoh_ctx* ctx)
try {
  return enqueue_staged_candidates(ctx, true);
} SVOH_ABI_CATCH(ctx)

int svoh_project_candidates_wait(svoh_ctx* ctx)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, ctx->cand_stage.state == 2, "no staged candidate projection in flight");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // returns at once when a fetch of the alignment in front has waited already
  ctx->cand_stage.state = 0;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_matcher_stage(svoh_ctx* ctx, int seeds, int n, int max_frame_views, int flags, svoh_matcher_stage_t* out)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, out && n >= 1 && n <= (1 << 24) && max_frame_views >= 2 && max_frame_views <= (1 << 16), "bad arguments");
  SVOH_REQUIRE(ctx, ctx->matcher_deferred, "svoh_matcher_stage: inside a deferred section only (svoh_matcher_begin_deferred)");
  const int kind = seeds ? 1 : 0;
  SVOH_REQUIRE(ctx, !ctx->matcher_deferred_used[kind], "one batch of each kind per deferred section: collect first");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  svoh_ctx::MatcherStage& st = ctx->matcher_stage[kind];
  SVOH_REQUIRE(ctx, (flags & ~(SVOH_STAGE_MATCH_OUTPUTS | SVOH_STAGE_RESIDENT_COLUMNS)) == 0, "unknown flag");
  if (seeds) ctx->seed_block.valid = false;   // (its uploads are stream-ordered behind whatever still reads the old block)
  layout_matcher_stage(seeds != 0, n, max_frame_views, (flags & SVOH_STAGE_MATCH_OUTPUTS) != 0, (flags & SVOH_STAGE_RESIDENT_COLUMNS) != 0, &st);
  PinnedBuffer& hbuf = seeds ? ctx->h_match_seeds : ctx->h_match_direct;
  DevBuffer& dbuf = seeds ? ctx->d_match_seeds : ctx->d_match_direct;
  SVOH_HIP_TRY(ctx, hbuf.reserve(st.total));
  SVOH_HIP_TRY(ctx, dbuf.reserve(st.total));
  uint8_t* h = static_cast<uint8_t*>(hbuf.ptr);
  memset(out, 0, sizeof *out);
  out->ref_frame_idx = reinterpret_cast<int32_t*>(h + st.o_idx); out->cur_frame_idx = reinterpret_cast<int32_t*>(h + st.o_cidx);
  if (st.resident) out->feature_index = reinterpret_cast<int32_t*>(h + st.o_fidx);
  else {
    out->px = reinterpret_cast<double*>(h + st.o_px); out->f = reinterpret_cast<double*>(h + st.o_f); out->grad = reinterpret_cast<double*>(h + st.o_grad);
    out->level = reinterpret_cast<int32_t*>(h + st.o_level);
  }
  out->type = h + st.o_type;
  out->result = reinterpret_cast<int32_t*>(h + st.o_result); out->success = h + st.o_success;
  if (seeds) out->state = reinterpret_cast<double*>(h + st.o_state);
  else { out->depth = reinterpret_cast<double*>(h + st.o_depth); out->px_cur = reinterpret_cast<double*>(h + st.o_pxcur); }
  if (st.want_outputs) {
    if (seeds) out->px_cur = reinterpret_cast<double*>(h + st.o_pxcur);
    out->f_cur = reinterpret_cast<double*>(h + st.o_fcur); out->search_level = reinterpret_cast<int32_t*>(h + st.o_slevel);
    out->h_inv = reinterpret_cast<double*>(h + st.o_hinv); out->A_cur_ref = reinterpret_cast<double*>(h + st.o_A);
  }
  st.valid = true;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_epipolar_match_batch(svoh_ctx* ctx, const svoh_matcher_options* options, int n_ref_frames,
                              const svoh_frame_view* ref_frames, const svoh_frame_view* cur_frame,
                              const svoh_se3* T_cur_ref, const svoh_feature_batch* features,
                              const double d_inv_common[3], const double* d_inv,
                              const svoh_epipolar_match_outputs* outputs)
try {
  return run_epipolar(ctx, options, n_ref_frames, ref_frames, cur_frame, T_cur_ref, features, d_inv_common, d_inv, outputs);
} SVOH_ABI_CATCH(ctx)

int svoh_matcher_begin_deferred(svoh_ctx* ctx)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, !ctx->matcher_deferred, "a deferred section is already open");
  ctx->matcher_deferred = true;
  ctx->matcher_deferred_used[0] = ctx->matcher_deferred_used[1] = false;
  ctx->matcher_pending.clear();
  ctx->matcher_pending_counts.clear();
  ctx->matcher_deferred_launch[0].valid = ctx->matcher_deferred_launch[1].valid = false;
  ctx->matcher_stage[0].valid = ctx->matcher_stage[1].valid = false;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

// What has to happen on the device between the upload of a queued batch and its kernels, for both kinds in one launch:
//  * svoh_frame_view::pose_result_index_plus1: current view k's T_f_w holds T_cam_imu; its pose is T_cam_imu * T_imu_world of
//    result k-1 of the pose batch that ran just before on this stream (PoseOptimizerHip::finishRun's product, the same inline function);
//  * svoh_feature_batch::feature_index: unit i's px / f / grad / level are those of feature feature_index[i] of its reference
//    frame's resident columns; an index outside the set (or a reference frame outside the table) poisons ref_frame_idx[i], which
//    the kernels' own range check turns into SVOH_MATCH_NOT_RUN.
struct PrologueBatch {
  DevFrameView* views; int n_ref, n_cur;                     // reference views, then the current ones
  int n; int32_t* ref_idx; const int32_t* fidx;              // fidx == NULL: nothing to gather
  int whole_sets;                                            // unit i = feature i - unit_begin of the frame whose units hold it: ref_idx is WRITTEN here
  double* px; double* f; double* grad; int32_t* level;
  const svoh_pose_result* pose_results; int n_pose_results;  // pose_results == NULL: no view takes its pose from the device
};
__global__ void matcher_prologue_kernel(PrologueBatch b0, PrologueBatch b1, int blocks0)
{
  const bool second = (int)blockIdx.x >= blocks0;
  const PrologueBatch& b = second ? b1 : b0;
  const int i = ((int)blockIdx.x - (second ? blocks0 : 0)) * (int)blockDim.x + (int)threadIdx.x;
  if (b.fidx && i < b.n) {
    int r, j;
    if (b.whole_sets) { r = whole_sets_frame_of(b.views, b.n_ref, i, b.n); j = i - b.views[r].unit_begin; b.ref_idx[i] = r; }
    else { r = b.ref_idx[i]; j = b.fidx[i]; }
    bool ok = r >= 0 && r < b.n_ref;
    const DevFrameView* v = ok ? &b.views[r] : nullptr;
    ok = ok && j >= 0 && j < v->feat_n;
    if (ok) {
      b.px[2 * i] = v->feat_px[2 * j]; b.px[2 * i + 1] = v->feat_px[2 * j + 1];
      b.f[3 * i] = v->feat_f[3 * j]; b.f[3 * i + 1] = v->feat_f[3 * j + 1]; b.f[3 * i + 2] = v->feat_f[3 * j + 2];
      b.grad[2 * i] = v->feat_grad[2 * j]; b.grad[2 * i + 1] = v->feat_grad[2 * j + 1];
      b.level[i] = v->feat_level[j];
    } else {
      b.px[2 * i] = b.px[2 * i + 1] = 0.0; b.f[3 * i] = b.f[3 * i + 1] = 0.0; b.f[3 * i + 2] = 1.0; b.grad[2 * i] = 1.0; b.grad[2 * i + 1] = 0.0; b.level[i] = 0;
      b.ref_idx[i] = -1;
    }
  }
  if (b.pose_results && i < b.n_cur) {
    DevFrameView& cv = b.views[b.n_ref + i];
    const int r = cv.pose_result_index_plus1 - 1;
    if (r >= 0 && r < b.n_pose_results) {
      cv.T_f_w = mul(cv.T_f_w, load_rigid(b.pose_results[r].T_imu_world));
      cv.pose_result_index_plus1 = 0;
    }
  }
}

// the kernels of the batches queued in the open deferred section go out now (one kernel when both kinds share a
// geometry), followed by the copies of their results to the pinned blocks; nothing is waited for
static int launch_deferred(svoh_ctx* ctx)
{
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    svoh_ctx::DeferredLaunch& d0 = ctx->matcher_deferred_launch[0];
    svoh_ctx::DeferredLaunch& d1 = ctx->matcher_deferred_launch[1];
    const bool v0 = d0.valid, v1 = d1.valid;
    d0.valid = d1.valid = false;
    if (v0 || v1) {
      MatcherArgs a0, a1;
      memset(&a0, 0, sizeof a0); memset(&a1, 0, sizeof a1);
      if (v0) memcpy(&a0, d0.args.data(), sizeof a0);
      if (v1) memcpy(&a1, d1.args.data(), sizeof a1);
      const size_t n0 = v0 ? (size_t)d0.n : 0, n1 = v1 ? (size_t)d1.n : 0;
      unsigned long long* dummy;
      unsigned int* uc = nullptr;
      int rc = reset_counters(ctx, &dummy);
      if (rc == SVOH_OK) rc = reserve_unit_counts(ctx, n0 + n1, &uc);
      if (rc != SVOH_OK) return rc;
      a0.unit_counts = uc; a1.unit_counts = uc + 4 * n0;
      auto blocks = [](const svoh_ctx::DeferredLaunch& d) { const int u = d.g8 ? 8 : 64; return (unsigned)((d.n + u - 1) / u); };   // per-unit geometries 0 / 1
      // poses composed on the device and features named by index: one small launch ahead of the kernels that read them
      {
        auto prologue_of = [](const svoh_ctx::DeferredLaunch& d, const MatcherArgs& a, bool valid) {
          PrologueBatch b;
          memset(&b, 0, sizeof b);
          const bool gather = d.d_fidx && !(a.whole_sets && d.g8 == 2);   // (whole sets in the packed geometry: the kernel reads the resident columns itself)
          if (!valid || (!gather && !d.pose_from_results)) return b;
          b.views = static_cast<DevFrameView*>(d.views_d); b.n_ref = d.n_ref; b.n_cur = d.n_cur;
          if (gather) {
            b.n = d.n; b.ref_idx = const_cast<int32_t*>(a.ref_frame_idx); b.fidx = static_cast<const int32_t*>(d.d_fidx);
            b.whole_sets = a.whole_sets;
            b.px = const_cast<double*>(a.px); b.f = const_cast<double*>(a.f); b.grad = const_cast<double*>(a.grad); b.level = const_cast<int32_t*>(a.level);
          }
          if (d.pose_from_results) { b.pose_results = static_cast<const svoh_pose_result*>(d.d_pose_results); b.n_pose_results = d.n_pose_results; }
          return b;
        };
        const PrologueBatch b0 = prologue_of(d0, a0, v0), b1 = prologue_of(d1, a1, v1);
        auto blocks_of = [](const PrologueBatch& b) { const int m = std::max(b.fidx ? b.n : 0, b.pose_results ? b.n_cur : 0); return (m + 255) / 256; };
        const int nb0 = blocks_of(b0), nb1 = blocks_of(b1);
        if (nb0 + nb1 > 0) hipLaunchKernelGGL(matcher_prologue_kernel, dim3((unsigned)(nb0 + nb1)), dim3(256), 0, ctx->stream, b0, b1, nb0);
        d0.pose_from_results = d1.pose_from_results = false; d0.d_fidx = d1.d_fidx = nullptr;
      }
      if (ctx->timing_on()) SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_start, ctx->stream));
      if (v0 && v1 && d0.g8 == d1.g8 && d0.g8 != 2) {
        const unsigned b0 = blocks(d0), b1 = blocks(d1);
        if (d0.g8) hipLaunchKernelGGL(match_mixed_kernel<true>, dim3(b0 + b1), dim3(64), 0, ctx->stream, a0, a1, (int)b0);
        else hipLaunchKernelGGL(match_mixed_kernel<false>, dim3(b0 + b1), dim3(64), 0, ctx->stream, a0, a1, (int)b0);
      } else {
        // (the packed geometry's kernels do not write the optional outputs of units that return early: zeroed here)
        if (v0) {
          if (d0.g8 == 2 && d0.out_bytes) SVOH_HIP_TRY(ctx, hipMemsetAsync(static_cast<uint8_t*>(d0.d_block) + d0.out_off, 0, d0.out_bytes, ctx->stream));
          rc = launch_matcher_kernels(ctx, false, d0.g8, a0, d0.n, a0.n_ref_frames, d0.max_w, d0.max_h);
          if (rc != SVOH_OK) return rc;
        }
        if (v1) {
          if (d1.g8 == 2 && d1.out_bytes) SVOH_HIP_TRY(ctx, hipMemsetAsync(static_cast<uint8_t*>(d1.d_block) + d1.out_off, 0, d1.out_bytes, ctx->stream));
          rc = launch_matcher_kernels(ctx, true, d1.g8, a1, d1.n, a1.n_ref_frames, d1.max_w, d1.max_h);
          if (rc != SVOH_OK) return rc;
        }
      }
      SVOH_HIP_TRY(ctx, hipGetLastError());
      if (ctx->timing_on()) SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_stop, ctx->stream));
      ctx->misc_timed = ctx->timing_on(); ctx->misc_launched = true;
      rc = reduce_unit_counts(ctx, n0 + n1);
      if (rc != SVOH_OK) return rc;
      if (v0) SVOH_HIP_TRY(ctx, svoh_copy_to_host(ctx, d0.d2h_dst, d0.d2h_src, d0.d2h_bytes));
      if (v1) SVOH_HIP_TRY(ctx, svoh_copy_to_host(ctx, d1.d2h_dst, d1.d2h_src, d1.d2h_bytes));
      // what collect waits for: these copies, not whatever the caller queues on the stream afterwards
      if (!ctx->ev_matcher_done) SVOH_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_matcher_done, hipEventDisableTiming));
      SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_matcher_done, ctx->stream));
      ctx->matcher_done_recorded = true;
      if (v1 && a1.state) {   // a seed batch of the section (staged or from host arrays): its device block can serve svoh_align_camera::pos_seed_unit until it is laid out anew
        svoh_ctx::SeedBlock& sb = ctx->seed_block;
        sb.valid = true; sb.views = a1.ref_frames; sb.n_ref = a1.n_ref_frames; sb.ref_idx = a1.ref_frame_idx; sb.f = a1.f; sb.state = a1.state; sb.n = d1.n;
      }
    }
  }
  return SVOH_OK;
}

}  // extern "C"

namespace svoh {
// svoh_align_camera::pos_seed_unit: feature i of a job with unit u = unit[i] >= 0 gets the position of seed u of the staged seed
// batch in flight -- T_world_keyframe * (f / mu), Frame::getSeedPosInFrame behind T_world_cam (resolveAlignmentPoints of the host
// mirror, the same inline functions) -- with the state the update has left in the batch's device block
struct PosFromSeedsArgs { const PosFromSeedsJob* jobs; const DevFrameView* views; int n_ref; const int32_t* ref_idx; const double* f; const double* state; int n_units; };
__global__ __launch_bounds__(256) void pos_from_seed_batch_kernel(const PosFromSeedsArgs a)
{
  const PosFromSeedsJob jb = a.jobs[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= jb.n) return;
  const int u = jb.unit[i];
  if (u < 0 || u >= a.n_units) return;
  const int r = a.ref_idx[u];
  if (r < 0 || r >= a.n_ref) return;
  const double depth = 1.0 / a.state[4 * (size_t)u];   // seed::getDepth (seed.h:110-113)
  const Vec3 in_kf = { a.f[3 * (size_t)u] * depth, a.f[3 * (size_t)u + 1] * depth, a.f[3 * (size_t)u + 2] * depth };
  const Vec3 p = transform(inverse(a.views[r].T_f_w), in_kf);
  jb.pos[3 * (size_t)i] = p.x; jb.pos[3 * (size_t)i + 1] = p.y; jb.pos[3 * (size_t)i + 2] = p.z;
}

int svoh_launch_pos_from_seed_batch(svoh_ctx* ctx, int n_jobs, int max_n, const PosFromSeedsJob* jobs_device)
{
  const svoh_ctx::SeedBlock& sb = ctx->seed_block;
  SVOH_REQUIRE(ctx, sb.valid, "pos_seed_unit: no staged seed batch has been sent off on this context (or its block has been staged again)");
  if (n_jobs <= 0 || max_n <= 0) return SVOH_OK;
  PosFromSeedsArgs a;
  a.jobs = jobs_device; a.views = static_cast<const DevFrameView*>(sb.views); a.n_ref = sb.n_ref; a.ref_idx = sb.ref_idx; a.f = sb.f; a.state = sb.state; a.n_units = sb.n;
  hipLaunchKernelGGL(pos_from_seed_batch_kernel, dim3((unsigned)((max_n + 255) / 256), (unsigned)n_jobs), dim3(256), 0, ctx->stream, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(ctx, SVOH_ERR_HIP, "pos_from_seed_batch launch failed: %s", hipGetErrorString(e));
  return SVOH_OK;
}
}  // namespace svoh

extern "C" {

int svoh_matcher_deferred_set_cur_frame(svoh_ctx* ctx, const svoh_frame_view* cur_frame)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, cur_frame != nullptr, "NULL argument");
  SVOH_REQUIRE(ctx, ctx->matcher_deferred, "no deferred section is open");
  svoh_ctx::DeferredLaunch& dl = ctx->matcher_deferred_launch[1];
  SVOH_REQUIRE(ctx, dl.valid && dl.views_h && dl.n_cur == 1, "no queued seed batch with one current frame (or it has been sent off already)");
  SVOH_REQUIRE(ctx, cur_frame->frame == dl.cur_frame_handle, "a different frame: only the view (pose) of the queued batch's current frame can be replaced");
  DevFrameView v;
  const int rc = fill_view(ctx, *cur_frame, &v, "current frame");
  if (rc != SVOH_OK) return rc;
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  // the staged copy is replaced as well: it is the upload's source (pinned: it must keep the bytes until the copy has run)
  DevFrameView* hv = static_cast<DevFrameView*>(dl.views_h) + dl.n_ref;
  *hv = v;
  SVOH_HIP_TRY(ctx, hipMemcpyAsync(static_cast<DevFrameView*>(dl.views_d) + dl.n_ref, hv, sizeof(DevFrameView), hipMemcpyHostToDevice, ctx->stream));
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_matcher_flush(svoh_ctx* ctx)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, ctx->matcher_deferred, "no deferred section is open");
  return launch_deferred(ctx);
} SVOH_ABI_CATCH(ctx)

int svoh_matcher_collect(svoh_ctx* ctx)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, ctx->matcher_deferred, "no deferred section is open");
  ctx->matcher_deferred = false;
  ctx->matcher_deferred_used[0] = ctx->matcher_deferred_used[1] = false;
  const int rc_launch = launch_deferred(ctx);   // whatever svoh_matcher_flush has not sent yet
  if (rc_launch != SVOH_OK) return rc_launch;
  // the section's results: behind their own copies, not behind what was queued on the stream since the flush
  if (ctx->matcher_done_recorded) { SVOH_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_matcher_done)); ctx->matcher_done_recorded = false; }
  else SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (const auto& c : ctx->matcher_pending) memcpy(c.dst, c.src, c.bytes);
  for (const auto& c : ctx->matcher_pending_counts) {
    int k = 0;
    for (int i = 0; i < c.n; ++i) k += c.flags[i];
    *c.dst = k;
  }
  ctx->matcher_pending.clear();
  ctx->matcher_pending_counts.clear();
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_match_direct_batch(svoh_ctx* ctx, const svoh_matcher_options* options, int n_ref_frames,
                            const svoh_frame_view* ref_frames, const svoh_frame_view* cur_frame,
                            const svoh_feature_batch* features, const double* depth, double* px_cur, int32_t* result,
                            double* f_cur, int32_t* search_level, double* h_inv, double* A_cur_ref)
try {
  return run_matcher(ctx, false, options, nullptr, n_ref_frames, ref_frames, cur_frame, features, depth, px_cur, result,
                     f_cur, search_level, h_inv, A_cur_ref, nullptr, nullptr, nullptr);
} SVOH_ABI_CATCH(ctx)

int svoh_match_direct_batch_pixelwise(svoh_ctx* ctx, const svoh_matcher_options* options, int n_ref_frames,
                                      const svoh_frame_view* ref_frames, const svoh_frame_view* cur_frame,
                                      const svoh_feature_batch* features, const double* depth, const double* landmark_xyz,
                                      double* px_cur, int32_t* result, double* f_cur, int32_t* search_level, double* h_inv,
                                      double* A_cur_ref)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, landmark_xyz, "landmark_xyz is NULL (svoh_match_direct_batch is the affine warp)");
  return run_matcher(ctx, false, options, nullptr, n_ref_frames, ref_frames, cur_frame, features, depth, px_cur, result,
                     f_cur, search_level, h_inv, A_cur_ref, nullptr, nullptr, nullptr, landmark_xyz);
} SVOH_ABI_CATCH(ctx)

int svoh_update_seeds_batch(svoh_ctx* ctx, const svoh_matcher_options* matcher_options,
                            const svoh_depth_filter_options* options, int n_ref_frames,
                            const svoh_frame_view* ref_frames, const svoh_frame_view* cur_frame,
                            const svoh_feature_batch* features, double* state, uint8_t* success, int32_t* match_result,
                            int32_t* n_success)
try {
  return run_matcher(ctx, true, matcher_options, options, n_ref_frames, ref_frames, cur_frame, features, nullptr, nullptr,
                     match_result, nullptr, nullptr, nullptr, nullptr, state, success, n_success);
} SVOH_ABI_CATCH(ctx)

int svoh_update_seeds_batch_ex(svoh_ctx* ctx, const svoh_matcher_options* matcher_options,
                               const svoh_depth_filter_options* options, int n_ref_frames,
                               const svoh_frame_view* ref_frames, const svoh_frame_view* cur_frame,
                               const svoh_feature_batch* features, double* state, uint8_t* success,
                               int32_t* match_result, int32_t* n_success, const svoh_seed_match_outputs* outputs)
try {
  const svoh_seed_match_outputs none = { nullptr, nullptr, nullptr, nullptr };
  const svoh_seed_match_outputs& o = outputs ? *outputs : none;
  return run_matcher(ctx, true, matcher_options, options, n_ref_frames, ref_frames, cur_frame, features, nullptr, o.px_cur,
                     match_result, o.f_cur, o.search_level, nullptr, o.A_cur_ref, state, success, n_success);
} SVOH_ABI_CATCH(ctx)

}  // extern "C"
