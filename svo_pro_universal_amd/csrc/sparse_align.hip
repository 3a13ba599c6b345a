// sparse_align.hip -- device-resident sparse image alignment for gfx950.
//
// Replaces, for a batch of independent (reference bundle, current bundle)
// pairs, the reference's
//   SparseImgAlign::run / evaluateError      src/svo_img_align/src/sparse_img_align.cpp:34-156
//   sparse_img_align_utils::*                src/svo_img_align/src/sparse_img_align.cpp:209-541
//   SparseImgAlignBase::update / applyPrior  src/svo_img_align/src/sparse_img_align_base.cpp:64-107
//   MiniLeastSquaresSolver::optimizeGaussNewton / solveDefaultImpl
//        src/vikit/vikit_solver/include/vikit/solver/implementation/mini_least_squares_solver.hpp:42-107,253-262
//   TukeyWeightFunction::weight              src/vikit/vikit_solver/src/robust_cost.cpp:48-60
//
// Design (MI355X-first, not a translation of the reference's loops):
//   * ONE workgroup owns ONE alignment problem for its whole life: all pyramid
//     levels and all Gauss-Newton iterations run inside one launch with no
//     host round trip; the 8x8 solve and the SE3 update happen in LDS.
//   * one thread owns one patch per pass (wave64: 64 patches in flight per
//     wave); per-patch state is two coalesced SoA loads (xyz_ref, uv).
//   * nothing per-pixel is ever written to memory: the reference's
//     jacobian_cache_ / ref_patch_cache_ / residual_cache_ (the 485 B per
//     patch-iteration of SURVEY 8(d)) are recomputed in registers from the
//     u8 pyramid level, which is staged ONCE per level into LDS (both the
//     reference and the current level) when it fits, with 16-byte coalesced
//     loads.  All arithmetic is fp64 like the reference (FloatType = double);
//     MI355X runs vector fp64 at half the fp32 rate, so this is cheap.
//   * the per-thread partial normal equations (upper triangle of J J^T, J r,
//     chi2) are reduced with wave shuffles, then across waves through LDS.
//     The summation order differs from the reference's sequential order; all
//     other arithmetic follows the reference expression by expression.
#include <atomic>
#include <cstdlib>
#include <cstring>

#include <type_traits>

#include "svoh_internal.h"
#include "svoh_device_utils.h"
#include "svoh_math.h"

// every wait for the whole stream in this file: the alignment's staging blocks have been read (svoh_internal.h)
#define SVOH_ALIGN_DRAIN(ctx) do { SVOH_HIP_TRY(ctx, hipStreamSynchronize((ctx)->stream)); (ctx)->align_launches_since_drain = 0; } while (0)

namespace svoh {

struct DevCamDesc {
  DevImage ref[SVOH_MAX_LEVELS];
  DevImage cur[SVOH_MAX_LEVELS];
  svoh_camera cam;
  svoh_se3 ref_T_imu_cam, ref_T_cam_imu, cur_T_cam_imu;
  double ref_pos[3];
  const double* px;
  const double* f;
  const double* pos_world;
  const uint8_t* flags;
  int32_t n_features;
  int32_t feat_off;  // first slot of this camera in the feature workspace
};

struct DevProblemDesc {
  int32_t n_cams, cam_begin;
  svoh_se3 T_init;
  double alpha_init, beta_init;
  svoh_align_prior prior;
};

struct AlignKernelArgs {
  const DevProblemDesc* problems;
  const DevCamDesc* cams;
  svoh_align_result* results;
  // feature workspace (SoA over all features of all problems)
  // per-feature workspace, 3 pairs of doubles per slot, pair-major: wpk[(pair * slots + gi) * 2 + {0,1}]
  //   pair 0 (x, y)  1 (z, u)  2 (v, state)   xyz_ref (a-4), uv in the reference image (level 0);
  //   state: 0 not selected by a-3, 1 selected, 2 selected and visible in the last full pass
  // 16 bytes per lane and pair = one global_load_lds_dwordx4 per pair in the staged patch loop.  The rows of the
  // projection Jacobian (jacobian_proj_cache_, a-4: 96 bytes per feature) are NOT kept: every patch-iteration
  // rebuilds them from xyz_ref (jac_rows below) -- ~50 fp64 operations against two thirds of the kernel's HBM-side
  // traffic in round 2 (3.85 of 5.27 GB per 1024 x 2000 launch were these rows read again every iteration)
  double* wpk;
  int64_t slots;                        // stride of the SoA arrays
  uint8_t* wsel;                        // selected by extractFeaturesSubset (a-3)
  uint8_t* wvis;                        // visibility of the last evaluation
  svoh_align_options opt;
  int32_t lds_img_bytes;                // dynamic LDS available for image staging
  int32_t rig_build;                    // host side only: a small launch with problems of several cameras -> the instantiation that runs a rig's cameras side by side
  int32_t latency_build;                // host side only: fewer problems than compute units -> the one-wave-per-SIMD build of the 256-thread kernel
  int32_t lds_two_per_cu;               // host side only: the launch counts on two workgroups per compute unit (launch_one sizes the image area for it)
  int32_t ws_lds_bytes;                 // > 0: the feature workspace of a (small) problem lives in LDS behind the image area (512-thread geometry)
  int32_t eval_level;                   // <0: full run; >=0: evaluate once at that level
  double* eval_out;                     // [64 H][8 g][chi2][n_meas] for eval mode
  long long* stamps;                    // diagnostic builds only (SVOH_PHASE_STAMPS): 8 per problem
  int32_t n_problems;
  int32_t* queue;                       // work queue head: workgroups pull problem indices from it
  int32_t one_each;                     // the grid has a workgroup per problem: workgroup b solves problem b, nobody asks the queue
  // patch-split mode: evaluate at a caller-owned device state, hand out the undivided sums
  const svoh_align_gn_state* ext_state;
  int32_t raw_sums;
  // cluster mode: ONE problem whose feature shares (one descriptor each, as in the patch-split evaluation) are
  // owned by `cluster` co-resident workgroups that add their normal equations through xchg every iteration
  int32_t cluster;                      // 0 / 1: off
  double* xchg;                         // 2 x cluster x kXchgStride doubles
  unsigned int* bar;                    // arrival counter, zeroed before the launch
#ifdef SVOH_TEST_HOOKS
  int32_t cluster_test_absent;          // libsvo_hip_testhooks.so only (SVOH_ALIGN_CLUSTER_TEST_ABSENT): share 1 never arrives
#endif
};

constexpr int kXchgStride = 64;         // doubles per share and parity (>= 45 + 8 + 1 accumulators + 2)

#ifndef SVOH_ROW_UNROLL_GONLY
#define SVOH_ROW_UNROLL_GONLY 4
#endif
#ifndef SVOH_PIN_MOMENTS
#define SVOH_PIN_MOMENTS 1
#endif
#ifndef SVOH_ALIGN_WAVE_STEP
#define SVOH_ALIGN_WAVE_STEP 1
#endif
#ifndef SVOH_STEP_PRIO
#define SVOH_STEP_PRIO 1
#endif
// rows of the full pass taken per loop trip: measured per configuration (round 3, scripts/ab.sh: 4x4 1.377 -> 1.365 ms with 2,
// 1.44 with 4; 8x8 3.36 -> 3.25 with 4; with the illumination terms' 15 moments live, 1 for 4x4)
#ifndef SVOH_ROW_UNROLL
#define SVOH_ROW_UNROLL (PH == 8 ? 4 : (D == 8 ? 1 : 2))
#endif

#ifdef SVOH_PHASE_STAMPS
// Diagnostic build only: wave 0 / lane 0 accumulates shader-clock cycles per
// phase.  Never enabled in the shipped library (rule: read shares, not length).
#define SVOH_STAMP_DECL long long st_t0 = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const long long st_begin = (long long)__builtin_amdgcn_s_memtime();
#define SVOH_STAMP_START() do { st_t0 = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define SVOH_STAMP_ADD(k) do { long long st_n = (long long)__builtin_amdgcn_s_memtime(); st_acc[k] += st_n - st_t0; st_t0 = st_n; } while (0)
#define SVOH_STAMP_COUNT(k) do { st_acc[k] += 1; } while (0)
#define SVOH_STAMP_FLUSH() do { if (threadIdx.x == 0) { st_acc[7] = (long long)__builtin_amdgcn_s_memtime() - st_begin; for (int k_ = 0; k_ < 8; ++k_) a.stamps[pbi * 20 + k_] = st_acc[k_]; for (int k_ = 0; k_ < 4; ++k_) a.stamps[pbi * 20 + 8 + k_] = g_state.dbg[k_]; for (int k_ = 0; k_ < 8; ++k_) a.stamps[pbi * 20 + 12 + k_] = g_state.dbg2[k_]; } } while (0)
#define SVOH_SERIAL_STAMP(k) do { long long st_n = (long long)__builtin_amdgcn_s_memtime(); s.dbg[k] += st_n - st_s0; st_s0 = st_n; } while (0)
// inside a pass over the patches (thread 0 only; waits for everything in flight first, so the pieces add up)
#define SVOH_PASS_STAMP_BEGIN() long long ps_t0 = 0; if (threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); ps_t0 = (long long)__builtin_amdgcn_s_memtime(); }
#define SVOH_PASS_STAMP(k) do { if (threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); long long ps_n = (long long)__builtin_amdgcn_s_memtime(); g_state.dbg2[k] += ps_n - ps_t0; ps_t0 = ps_n; } } while (0)
#define SVOH_OUTER_STAMP_BEGIN() long long os_t0 = 0; if (threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); os_t0 = (long long)__builtin_amdgcn_s_memtime(); }
#define SVOH_OUTER_STAMP(k) do { if (threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); long long os_n = (long long)__builtin_amdgcn_s_memtime(); g_state.dbg2[k] += os_n - os_t0; os_t0 = os_n; } } while (0)
#else
#define SVOH_PASS_STAMP_BEGIN()
#define SVOH_PASS_STAMP(k) do {} while (0)
#define SVOH_OUTER_STAMP_BEGIN()
#define SVOH_OUTER_STAMP(k) do {} while (0)
#define SVOH_STAMP_DECL
#define SVOH_STAMP_START() do {} while (0)
#define SVOH_STAMP_ADD(k) do {} while (0)
#define SVOH_STAMP_COUNT(k) do {} while (0)
#define SVOH_STAMP_FLUSH() do {} while (0)
#define SVOH_SERIAL_STAMP(k) do {} while (0)
#endif

struct ShState {
  Rigid T, Told;
  double alpha, beta, alpha_old, beta_old;
  Rigid Tcr[SVOH_MAX_CAMS];
  double I_prior[8];
  float alpha_f, beta_f;
  int stop, level_done, nsel, status;
  long long patch_iters;
  // LDL^T factor of the level's (H + prior information), kept while the Hessian is (gn_serial_step)
  double fact[36];
  int fact_tr[8];
  int fact_nonzero;
  // the wave-wide step (gn_wave_step): the factorisation's permutation as an index, the prior's share of the gradient,
  // the solution
  int fact_perm[8];
  double prior_g[8];
  double dx[8];
  // what the one-lane step needs of the problem and camera descriptors, and what it reports per level: in LDS for the
  // problem's life, so that the step makes no round trip to global memory (a load it must wait for, or a store the
  // barrier behind it must wait for) between two passes over the patches
  Rigid cam_cur_T_cam_imu[SVOH_MAX_CAMS], cam_ref_T_imu_cam[SVOH_MAX_CAMS];
  svoh_align_prior prior;
  int lvl_iters[SVOH_MAX_LEVELS], lvl_n_meas[SVOH_MAX_LEVELS];
  double lvl_chi2[SVOH_MAX_LEVELS];
#ifdef SVOH_PHASE_STAMPS
  long long dbg[4];   // diagnostic build: cycles of the one-lane step in set-up / solve / update / camera poses
  long long dbg2[8];  // ... and of thread 0 inside the passes: row arrival / projection / pixels / Jacobian rows + map / loop exit
#endif
};

// The Gauss-Newton state of the workgroup's problem, the summed normal equations and the visible count live at file
// scope: the out-of-line one-lane step (gn_serial_step) then addresses them as LDS (ds_read / ds_write) instead of
// through generic pointers handed to it (flat loads and stores, which wait on both memory counters).
__shared__ ShState g_state;
__shared__ double g_sum[45];   // AccLayout<8>::NACC: upper triangle of H, g, chi2
__shared__ int g_nvis;

// ---- image accessors --------------------------------------------------------
// LDS=true: the level lives in the workgroup's LDS (address space 3 -> ds_read);
// LDS=false: gathers go to global memory (L1/L2).
template <int N>
struct PackedRow { static constexpr int NQ = (N + 7) / 8; unsigned long long v[NQ]; };

template <bool LDS>
struct ImgView;
// row<N>(off, out): N consecutive pixels.  From global memory they come as one unaligned 8- or 16-byte load
// (a row of the (P+3)- or (P+1)-pixel footprint) instead of N byte gathers: the L1 / texture path of a CU
// serves every wave-wide load instruction separately, and 74 byte loads per 4x4 patch were what the levels
// that do not fit in LDS spent their time on.  (Reads up to 15 bytes past the row: kSlabTailPad.)
template <>
struct ImgView<false> {
  const uint8_t* p;
  int pitch;
  __device__ __forceinline__ unsigned at(int off) const { return p[off]; }
  template <int N>
  __device__ __forceinline__ void row(int off, unsigned (&out)[N]) const
  {
    PackedRow<N> r;
    fetch<N>(off, r);
    unpack<N>(r, out);
  }
  // the load and the use of a row, separately: patch_moments requests the next row before it works on this one
  template <int N>
  __device__ __forceinline__ void fetch(int off, PackedRow<N>& r) const { __builtin_memcpy(r.v, p + off, 8 * PackedRow<N>::NQ); }
  template <int N>
  static __device__ __forceinline__ void unpack(const PackedRow<N>& r, unsigned (&out)[N])
  {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = (unsigned)(r.v[i / 8] >> (8 * (i % 8))) & 0xFFu;
  }
};
template <>
struct ImgView<true> {
  const __attribute__((address_space(3))) uint8_t* p;
  int pitch;
  __device__ __forceinline__ unsigned at(int off) const { return p[off]; }
  template <int N>
  __device__ __forceinline__ void row(int off, unsigned (&out)[N]) const
  {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = p[off + i];
  }
};

template <int D>
struct AccLayout {
  static constexpr int NH = D * (D + 1) / 2;
  static constexpr int NACC = NH + D + 1;
  // weighted per-patch moments: Sxx Sxy Syy Sxr Syr Srr [SxI SyI SII SI Sx Sy S1 SIr Sr]
  static constexpr int NMOM = (D == 8) ? 15 : 6;
};

__device__ __forceinline__ float tukey_weight(float e)
{
  const float b = 4.6851f;
  const float b2 = b * b;
  const float x2 = e * e;
  if (x2 <= b2) {
    const float t = 1.0f - x2 / b2;
    return t * t;
  }
  return 0.0f;
}

// Per-camera constants of the Jacobian rows, in LDS for the problem's life (21 doubles per camera):
//   [0..9) R_imu_cam row-major, [9..12) t_imu_cam, [12..21) R_cam_imu row-major
constexpr int kJacConsts = 24;

// Rows of the projection Jacobian of one feature, scaled to the pyramid level:
//   a = jp0 * scale, b = jp1 * scale with (jp0; jp1) = Frame::jacobian_xyz2uv_imu (frame.h:342-357) times the focal
//   length, or Frame::jacobian_xyz2image_imu (frame.cpp:274-290) times -1 (sparse_img_align.cpp:298-309).
// The reference keeps them per feature (jacobian_proj_cache_); here they are rebuilt from xyz_ref in every
// patch-iteration (see AlignKernelArgs::wpk).  p_in_cam of the reference is T_cam_imu * (T_imu_cam * xyz_ref):
// xyz_ref itself up to rounding (the two transformations of a frame are inverses of each other), which is what is
// used; -x/z is taken as -x * (1/z).  Both are last-bit differences, inside the stated tolerance of H and g.
__device__ __forceinline__ void jac_rows(const double* jc, const CamModel& cm, bool use_distortion_jac, const Vec3& X,
                                         double scale, double (&a)[6], double (&b)[6])
{
  // xyz_in_imu = T_imu_cam * xyz_ref
  const Vec3 p = { jc[0] * X.x + jc[1] * X.y + jc[2] * X.z + jc[9], jc[3] * X.x + jc[4] * X.y + jc[5] * X.z + jc[10],
                   jc[6] * X.x + jc[7] * X.y + jc[8] * X.z + jc[11] };
  const double* R = jc + 12;
  double B[6];
  if (!use_distortion_jac) {
    // A = -(1/z) [1 0 -x/z; 0 1 -y/z] * |fx| * scale  (the two zeros are not multiplied out)
    const double rz = 1.0 / X.z;
    const double sm = -rz * (fabs(cm.fx) * scale);
    const double a2 = sm * (-X.x * rz), a5 = sm * (-X.y * rz);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      B[c] = sm * R[c] + a2 * R[6 + c];
      B[3 + c] = sm * R[3 + c] + a5 * R[6 + c];
    }
  } else {
    double A[6];
    project3_jacobian(cm, X, A);
    const double m = -scale;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        B[r * 3 + c] = (A[r * 3 + 0] * R[c] + A[r * 3 + 1] * R[3 + c] + A[r * 3 + 2] * R[6 + c]) * m;
  }
  // G_x = [I, -skew(p)]
  a[0] = B[0]; a[1] = B[1]; a[2] = B[2];
  a[3] = B[1] * (-p.z) + B[2] * p.y;
  a[4] = B[0] * p.z + B[2] * (-p.x);
  a[5] = B[0] * (-p.y) + B[1] * p.x;
  b[0] = B[3]; b[1] = B[4]; b[2] = B[5];
  b[3] = B[4] * (-p.z) + B[5] * p.y;
  b[4] = B[3] * p.z + B[5] * (-p.x);
  b[5] = B[3] * (-p.y) + B[4] * p.x;
}

// One patch, pixel part: interpolated reference patch with border and its
// central differences (a-5), interpolated current patch and residual (a-6),
// robust weight (a-7), reduced to the weighted moments of (dx, dy, I_ref, 1)
// against themselves and against the residual.  Every pixel's Jacobian is
//   J = [ (dx*jp0 + dy*jp1)*scale , -I_ref , -1 ]        (sparse_img_align.cpp:389-398)
// i.e. a fixed linear map of (dx, dy, I_ref, 1), so the patch's contribution to
// H = sum w J J^T and g = -sum w J r is that map applied to these moments.
// This regroups the reference's per-pixel sum algebraically (same real-number
// result; rounding differs at the 1e-16 level, like the reduction order does).
// GONLY: the unweighted Hessian of a level does not change while the set of visible patches does not (inverse
// compositional: the Jacobians are those of the reference patch), so later iterations only need the moments
// against the residual -- Sxr Syr Srr [SIr Sr] -- i.e. the gradient and chi2.
// PH x PW: the rows and columns of the (sub-)patch this call covers.  An 8x8 patch is taken as two 8x4 column
// halves (patch_moments below): the rolling window is (PW+2) wide, and the three interpolated rows, the raw rows and
// the current rows of a 10-wide window are what pushed the 8x8 kernel 200-400 registers over the budget
// (.vgpr_spill_count 209 / 436 in round 1).  The halves add into the same moments; the two window columns they
// share are interpolated twice.
template <int PH, int PW, int D, bool RLDS, bool CLDS, bool GONLY, bool ZERO, bool ROB = false>
__device__ __forceinline__ void patch_moments_part(
    const ImgView<RLDS>& ref, const ImgView<CLDS>& cur, int ru, int rv, double rsu, double rsv, int cu, int cv,
    double csu, double csv, double one_plus_alpha, double beta, bool robust, float weight_scale,
    double (&mom)[AccLayout<D>::NMOM])
{
  constexpr int P = PW;      // width of this call's window
  constexpr int WB = P + 2;
  // bilinear weights (sparse_img_align.cpp:355-360 and :456-461)
  const double rwtl = (1.0 - rsu) * (1.0 - rsv), rwtr = rsu * (1.0 - rsv);
  const double rwbl = (1.0 - rsu) * rsv, rwbr = rsu * rsv;
  // The residual (I_cur * (1 + alpha) + beta) - I_ref (sparse_img_align.cpp:466-472) is taken as ONE chain of four multiply-adds that
  // starts at beta - I_ref, with the gain folded into the four bilinear weights once per patch: five operations per pixel instead of
  // six (mul + 3 fma for the intensity, fma, sub) and no multiply that only heads a chain.  Same real number; the rounding differs
  // from the reference's order in the last place of the residual (inside the stated 1e-10 of H and g; every parity test unchanged).
  // Round 6, measured on one box against the six-operation form: 8x8 launch 2.97 - 3.02 -> 2.93 - 2.95 ms, 4x4 within the noise
  // (profiles/r06_align_diet_ab.txt).
  const double cwtl = ((1.0 - csu) * (1.0 - csv)) * one_plus_alpha, cwtr = (csu * (1.0 - csv)) * one_plus_alpha;
  const double cwbl = ((1.0 - csu) * csv) * one_plus_alpha, cwbr = (csu * csv) * one_plus_alpha;

  const int roff = rv * ref.pitch + ru;
  const int coff = cv * cur.pitch + cu;
  if constexpr (ZERO) {
#pragma unroll
    for (int k = 0; k < AccLayout<D>::NMOM; ++k) mom[k] = 0.0;
  }

  // rolling window: three interpolated reference rows (up / centre / down) and
  // two raw current rows; one new row of each per output row
  unsigned rawA[WB + 1], rawB[WB + 1];
  double it0[WB], it1[WB], it2[WB];
  unsigned curA[P + 1], curB[P + 1];
  // Rows from global memory are requested one row ahead (packed, 2-4 VGPRs each): the round trip to L2 then
  // overlaps the interpolation of the row before instead of preceding every row's arithmetic.
  PackedRow<WB + 1> pref_r;
  PackedRow<P + 1> pref_c;
  if constexpr (!RLDS) {
    PackedRow<WB + 1> r0, r1, r2;
    ref.template fetch<WB + 1>(roff, r0);
    ref.template fetch<WB + 1>(roff + ref.pitch, r1);
    ref.template fetch<WB + 1>(roff + 2 * ref.pitch, r2);
    ref.template fetch<WB + 1>(roff + 3 * ref.pitch, pref_r);
    if constexpr (!CLDS) {
      PackedRow<P + 1> c0;
      cur.template fetch<P + 1>(coff, c0);
      cur.template fetch<P + 1>(coff + cur.pitch, pref_c);
      ImgView<false>::unpack<P + 1>(c0, curB);
    }
    ImgView<false>::unpack<WB + 1>(r0, rawA);
    ImgView<false>::unpack<WB + 1>(r1, rawB);
#pragma unroll
    for (int i = 0; i < WB; ++i)
      it1[i] = rwtl * (double)rawA[i] + rwtr * (double)rawA[i + 1] + rwbl * (double)rawB[i] + rwbr * (double)rawB[i + 1];
    ImgView<false>::unpack<WB + 1>(r2, rawA);
  } else {
    ref.template row<WB + 1>(roff, rawA);
    ref.template row<WB + 1>(roff + ref.pitch, rawB);
#pragma unroll
    for (int i = 0; i < WB; ++i)
      it1[i] = rwtl * (double)rawA[i] + rwtr * (double)rawA[i + 1] + rwbl * (double)rawB[i] + rwbr * (double)rawB[i + 1];
    ref.template row<WB + 1>(roff + 2 * ref.pitch, rawA);
  }
#pragma unroll
  for (int i = 0; i < WB; ++i)
    it2[i] = rwtl * (double)rawB[i] + rwtr * (double)rawB[i + 1] + rwbl * (double)rawA[i] + rwbr * (double)rawA[i + 1];
  if constexpr (RLDS || CLDS) cur.template row<P + 1>(coff, curB);
  // here: rawA = raw row 2, it1 = interp row 0, it2 = interp row 1, curB = cur row 0

  // the gradient-only pass has registers to spare (7 or 9 accumulators instead of 28 or 45): unrolled rows let the
  // next row's LDS / L2 reads overlap this row's arithmetic (1.64 -> 1.59 ms on the headline config)
  // (with Tukey weights a row carries a float division and the weighted forms of all moments: one row per trip -- four
  // unrolled rows of an 8x8 patch were 240-390 spilled registers)
  constexpr int kRowUnroll = GONLY ? SVOH_ROW_UNROLL_GONLY : (ROB ? 1 : SVOH_ROW_UNROLL);
#pragma unroll kRowUnroll
  for (int y = 0; y < PH; ++y) {
    const int rrow = roff + (y + 3) * ref.pitch;
    const int crow = coff + (y + 1) * cur.pitch;
#pragma unroll
    for (int i = 0; i < P + 1; ++i) curA[i] = curB[i];
    if constexpr (!RLDS && !CLDS) {
      ImgView<false>::unpack<WB + 1>(pref_r, rawB);
      ImgView<false>::unpack<P + 1>(pref_c, curB);
      if (y + 1 < PH) {   // request the rows of the next output row now (they are inside the footprint)
        ref.template fetch<WB + 1>(rrow + ref.pitch, pref_r);
        cur.template fetch<P + 1>(crow + cur.pitch, pref_c);
      }
      asm volatile("" ::: "memory");   // keep the requests ahead of this row's arithmetic
    } else {
      ref.template row<WB + 1>(rrow, rawB);
      cur.template row<P + 1>(crow, curB);
    }
#pragma unroll
    for (int i = 0; i < WB; ++i) {
      it0[i] = it1[i];
      it1[i] = it2[i];
      it2[i] = rwtl * (double)rawA[i] + rwtr * (double)rawA[i + 1] + rwbl * (double)rawB[i] +
               rwbr * (double)rawB[i + 1];
    }
#pragma unroll
    for (int i = 0; i < WB + 1; ++i) rawA[i] = rawB[i];
    // output row y: it0 = interp row y (up), it1 = y+1 (centre), it2 = y+2 (down)
#pragma unroll
    for (int x = 0; x < P; ++x) {
      const double ref_val = it1[x + 1];
      // twice the central differences: the factor 1/2 (1/4 for the products of two gradients) is applied to the
      // finished moments in patch_moments -- a power of two commutes with every rounding, so the bits are the same
      const double dx = it1[x + 2] - it1[x];
      const double dy = it2[x + 1] - it0[x + 1];
      const double res = fma(cwtl, (double)curA[x], fma(cwtr, (double)curA[x + 1], fma(cwbl, (double)curB[x], fma(cwbr, (double)curB[x + 1], beta - ref_val))));
      if constexpr (GONLY) {
        mom[3] += dx * res;   // Sxr
        mom[4] += dy * res;   // Syr
        mom[5] += res * res;  // Srr (chi2)
        if constexpr (D == 8) {
          mom[13] += res * ref_val;   // SIr
          mom[14] += res;             // Sr
        }
        continue;
      }
      double wdx = dx, wdy = dy, wr = res;
      if (robust) {
        const double w = (double)tukey_weight((float)(res / (double)weight_scale));
        wdx = dx * w; wdy = dy * w; wr = res * w;
        if constexpr (D == 8) {
          mom[8] += (ref_val * w) * ref_val;  // SII
          mom[9] += ref_val * w;              // SI
          mom[12] += w;                       // S1
        }
      } else if constexpr (D == 8) {
        mom[8] += ref_val * ref_val;
        mom[9] += ref_val;
        mom[12] += 1.0;
      }
      mom[0] += wdx * dx;   // Sxx
      mom[1] += wdx * dy;   // Sxy
      mom[2] += wdy * dy;   // Syy
      mom[3] += wdx * res;  // Sxr
      mom[4] += wdy * res;  // Syr
      mom[5] += wr * res;   // Srr (chi2)
      if constexpr (D == 8) {
        mom[6] += wdx * ref_val;   // SxI
        mom[7] += wdy * ref_val;   // SyI
        mom[10] += wdx;            // Sx
        mom[11] += wdy;            // Sy
        mom[13] += wr * ref_val;   // SIr
        mom[14] += wr;             // Sr
      }
    }
  }
}

template <int P, int D, bool RLDS, bool CLDS, bool GONLY = false, bool ROB = false>
__device__ __forceinline__ void patch_moments(
    const ImgView<RLDS>& ref, const ImgView<CLDS>& cur, int ru, int rv, double rsu, double rsv, int cu, int cv,
    double csu, double csv, double one_plus_alpha, double beta, bool robust, float weight_scale,
    double (&mom)[AccLayout<D>::NMOM])
{
  if constexpr (P == 8) {
    patch_moments_part<8, 4, D, RLDS, CLDS, GONLY, true, ROB>(ref, cur, ru, rv, rsu, rsv, cu, cv, csu, csv, one_plus_alpha, beta,
                                                              robust, weight_scale, mom);
    patch_moments_part<8, 4, D, RLDS, CLDS, GONLY, false, ROB>(ref, cur, ru + 4, rv, rsu, rsv, cu + 4, cv, csu, csv, one_plus_alpha,
                                                               beta, robust, weight_scale, mom);
  } else {
    patch_moments_part<P, P, D, RLDS, CLDS, GONLY, true, ROB>(ref, cur, ru, rv, rsu, rsv, cu, cv, csu, csv, one_plus_alpha, beta,
                                                              robust, weight_scale, mom);
  }
  // dx, dy above are twice the central differences (sparse_img_align.cpp:386-387 has the 0.5)
  mom[3] *= 0.5; mom[4] *= 0.5;
  if constexpr (!GONLY) {
    mom[0] *= 0.25; mom[1] *= 0.25; mom[2] *= 0.25;
    if constexpr (D == 8) { mom[6] *= 0.5; mom[7] *= 0.5; mom[10] *= 0.5; mom[11] *= 0.5; }
  }
}

// ---- rows geometry: LPP lanes per patch, lane r owns the P / LPP output rows from r * P / LPP on ----------------
// north_star's "several lanes per patch".  A problem with few patches (one camera stream: <= 180 at C3) fills a
// fraction of a workgroup's lanes when a lane owns a whole patch, and that lane's pass is one long chain of P rolling
// rows.  Here the LPP consecutive lanes of a group share a patch: each runs the rolling window of patch_moments_part
// over its own P / LPP rows (the window's first two interpolated rows are computed again by every lane: (P / LPP + 2)
// interpolated rows per lane instead of P + 2 per patch).  Same expressions as the lane-per-patch code, so every
// interpolated value and every residual has the same bits; a lane's moments are partial sums of the patch's.  Nothing
// is exchanged between the lanes of a group: the map from moments to normal equations (accumulate_patch) is linear in
// the moments, so every lane applies it to its own rows and the wave reduction adds row groups instead of patches.
// (Exchanging would cost more than it saves: a double moved over DPP is two VALU slots, an interpolated value
// recomputed is four; summing a group's moments first costs 18-36 slots per lane against the 66 of the map.)
template <int P, int LPP, int D, bool RLDS, bool CLDS, bool GONLY = false, bool ROB = false>
__device__ __forceinline__ void patch_rows_moments(
    const ImgView<RLDS>& ref, const ImgView<CLDS>& cur, int ru, int rv, double rsu, double rsv, int cu, int cv,
    double csu, double csv, int r, double one_plus_alpha, double beta, bool robust, float weight_scale,
    double (&mom)[AccLayout<D>::NMOM])
{
  constexpr int PH = P / LPP;
  static_assert(PH * LPP == P && PH >= 1, "lanes per patch must divide the patch height");
  const int rv0 = rv + r * PH, cv0 = cv + r * PH;
  if constexpr (P == 8) {
    patch_moments_part<PH, 4, D, RLDS, CLDS, GONLY, true, ROB>(ref, cur, ru, rv0, rsu, rsv, cu, cv0, csu, csv, one_plus_alpha, beta,
                                                               robust, weight_scale, mom);
    patch_moments_part<PH, 4, D, RLDS, CLDS, GONLY, false, ROB>(ref, cur, ru + 4, rv0, rsu, rsv, cu + 4, cv0, csu, csv, one_plus_alpha,
                                                                beta, robust, weight_scale, mom);
  } else {
    patch_moments_part<PH, P, D, RLDS, CLDS, GONLY, true, ROB>(ref, cur, ru, rv0, rsu, rsv, cu, cv0, csu, csv, one_plus_alpha, beta,
                                                               robust, weight_scale, mom);
  }
  mom[3] *= 0.5; mom[4] *= 0.5;
  if constexpr (!GONLY) {
    mom[0] *= 0.25; mom[1] *= 0.25; mom[2] *= 0.25;
    if constexpr (D == 8) { mom[6] *= 0.5; mom[7] *= 0.5; mom[10] *= 0.5; mom[11] *= 0.5; }
  }
}

// Apply the patch's linear map to its moments: acc += (H upper triangle, g, chi2).
// a = jp0 * scale, b = jp1 * scale (scale is a power of two: (dx*jp0+dy*jp1)*scale == dx*a+dy*b exactly)
template <int D>
__device__ __forceinline__ void accumulate_patch(const double (&mom)[AccLayout<D>::NMOM], const double (&a)[6],
                                                 const double (&b)[6], bool est_alpha, bool est_beta,
                                                 double (&acc)[AccLayout<D>::NACC])
{
  constexpr int NH = AccLayout<D>::NH;
  double u[6], v[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    u[k] = mom[0] * a[k] + mom[1] * b[k];
    v[k] = mom[1] * a[k] + mom[2] * b[k];
  }
  int idx = 0;
#pragma unroll
  for (int i = 0; i < D; ++i) {
#pragma unroll
    for (int j = i; j < D; ++j) {
      if (j < 6) {
        acc[idx] += u[i] * a[j] + v[i] * b[j];
      } else if constexpr (D == 8) {
        if (i < 6) {
          if (j == 6) { if (est_alpha) acc[idx] -= a[i] * mom[6] + b[i] * mom[7]; }
          else { if (est_beta) acc[idx] -= a[i] * mom[10] + b[i] * mom[11]; }
        } else if (i == 6) {
          if (j == 6) { if (est_alpha) acc[idx] += mom[8]; }
          else { if (est_alpha && est_beta) acc[idx] += mom[9]; }
        } else {
          if (est_beta) acc[idx] += mom[12];
        }
      }
      ++idx;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[NH + i] -= a[i] * mom[3] + b[i] * mom[4];
  if constexpr (D == 8) {
    if (est_alpha) acc[NH + 6] += mom[13];  // g6 -= sum w (-I) r
    if (est_beta) acc[NH + 7] += mom[14];   // g7 -= sum w (-1) r
  }
  acc[NH + D] += mom[5];
}

// Gradient-only form of the above for the iterations that reuse the level's Hessian: accg = (g[0..D), chi2).
template <int D>
__device__ __forceinline__ void accumulate_patch_gradient(const double (&mom)[AccLayout<D>::NMOM], const double (&a)[6],
                                                          const double (&b)[6], bool est_alpha, bool est_beta,
                                                          double (&accg)[D + 1])
{
#pragma unroll
  for (int i = 0; i < 6; ++i) accg[i] -= a[i] * mom[3] + b[i] * mom[4];
  if constexpr (D == 8) {
    if (est_alpha) accg[6] += mom[13];
    if (est_beta) accg[7] += mom[14];
  }
  accg[D] += mom[5];
}

template <int NT>
__device__ __forceinline__ void stage_image(unsigned char* dst, const DevImage& im, int tid)
{
  const int total = im.w * im.h;
  if (im.pitch == im.w && (reinterpret_cast<uintptr_t>(im.data) & 15) == 0) {
    // LDS-DMA: every wave copies 1 KB per instruction (16 bytes per lane, lane-contiguous on both sides) without a
    // round trip through VGPRs, and all of a thread's requests are in flight at once.
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    const int n16 = total >> 4;
    const int lane = tid & 63, wave_base = tid - lane;
    for (int i0 = wave_base; i0 < n16; i0 += NT)
      if (i0 + lane < n16)
        __builtin_amdgcn_global_load_lds((gptr)(im.data + (size_t)(i0 + lane) * 16), (lptr)(dst + (size_t)i0 * 16), 16, 0, 0);
    for (int i = (n16 << 4) + tid; i < total; i += NT) dst[i] = im.data[i];
    // no wait here: the requests stay in flight, the caller waits (vmcnt(0)) before the barrier in front of the first read
  } else {
    for (int i = tid; i < total; i += NT) {
      const int y = i / im.w, x = i - y * im.w;
      dst[i] = im.data[(size_t)y * im.pitch + x];
    }
  }
}

// SVOH_ALIGN_STAGE_REGS: a level's two images (reference and current) through registers instead of LDS-DMA -- every thread requests up to six
// 16-byte pieces of each image back to back (all in flight together), waits once and writes them to LDS.  Round 6, by the cycle stamps
// (profiles/r06_align_p4_stamps.txt): the 50 KB of LDS-DMA requests of a problem's start take ~60 K cycles to arrive (a wave's global_load_lds
// instructions return one behind the other), a fifth of the life of a 180-patch problem and 6 % of a 2000-patch one.
#ifndef SVOH_ALIGN_STAGE_REGS
#define SVOH_ALIGN_STAGE_REGS 1
#endif
template <int NT, bool REGS>
__device__ __forceinline__ void stage_image_pair(unsigned char* dst_a, const DevImage& im_a, unsigned char* dst_b, const DevImage& im_b, int tid)
{
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) v4u* lds16;
  const bool fast = im_a.pitch == im_a.w && im_b.pitch == im_b.w && ((reinterpret_cast<uintptr_t>(im_a.data) | reinterpret_cast<uintptr_t>(im_b.data)) & 15) == 0;
  if (!REGS || !fast) { stage_image<NT>(dst_a, im_a, tid); stage_image<NT>(dst_b, im_b, tid); return; }
  const int total_a = im_a.w * im_a.h, total_b = im_b.w * im_b.h;
  const int n16a = total_a >> 4, n16b = total_b >> 4;
  const v4u* src_a = reinterpret_cast<const v4u*>(im_a.data);
  const v4u* src_b = reinterpret_cast<const v4u*>(im_b.data);
  lds16 da = (lds16)dst_a;   // (both LDS offsets are multiples of 16: the levels' sizes are rounded up to that)
  lds16 db = (lds16)dst_b;
  constexpr int K = 6;
  const int n16 = n16a > n16b ? n16a : n16b;
  for (int base = tid; base < n16; base += NT * K) {
    v4u va[K], vb[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = base + k * NT;
      if (i < n16a) va[k] = src_a[i];
      if (i < n16b) vb[k] = src_b[i];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = base + k * NT;
      if (i < n16a) da[i] = va[k];
      if (i < n16b) db[i] = vb[k];
    }
  }
  for (int i = (n16a << 4) + tid; i < total_a; i += NT) dst_a[i] = im_a.data[i];
  for (int i = (n16b << 4) + tid; i < total_b; i += NT) dst_b[i] = im_b.data[i];
}

// ... and ALL the levels that are resident for a problem's life in one go: a table of the images (workgroup-uniform, in LDS), every thread takes up to
// sixteen 16-byte pieces of the whole list, requests them back to back, waits ONCE and writes them: one exposed round trip to HBM per problem instead
// of one per level (the images come cold: 838 MB of pyramids in the benchmark).
struct StageItem { const unsigned char* src; int dst_off, begin16, bytes, pad_; };
constexpr int kMaxStageItems = 16;
template <int NT>
__device__ __forceinline__ void stage_item_list(unsigned char* lds_base, const StageItem* items, int n_items, int total16, int tid)
{
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) v4u* lds16;
  constexpr int K = 16;
  for (int f0 = tid; f0 < total16; f0 += NT * K) {
    v4u v[K];
    int jj[K];
    int j = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int f = f0 + k * NT;
      jj[k] = 0;
      if (f < total16) {
        while (j + 1 < n_items && f >= items[j + 1].begin16) ++j;
        jj[k] = j;
        v[k] = reinterpret_cast<const v4u*>(items[j].src)[f - items[j].begin16];
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int f = f0 + k * NT;
      if (f < total16) ((lds16)(lds_base + items[jj[k]].dst_off))[f - items[jj[k]].begin16] = v[k];
    }
  }
  for (int j = 0; j < n_items; ++j) {   // the images' last bytes (a level of 47 x 30 pixels is 88 pieces and two bytes)
    const int n16 = j + 1 < n_items ? items[j + 1].begin16 - items[j].begin16 : total16 - items[j].begin16;
    for (int i = (n16 << 4) + tid; i < items[j].bytes; i += NT) lds_base[items[j].dst_off + i] = items[j].src[i];
  }
}

// Values read from LDS are wave-uniform here but the compiler cannot know it;
// moving them to SGPRs frees a VGPR pair per double in the patch loop.
__device__ __forceinline__ double uniform_f64(double v)
{
  const unsigned long long b = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ Rigid uniform_rigid(const Rigid& T)
{
  Rigid r;
  r.q.w = uniform_f64(T.q.w); r.q.x = uniform_f64(T.q.x); r.q.y = uniform_f64(T.q.y); r.q.z = uniform_f64(T.q.z);
  r.t.x = uniform_f64(T.t.x); r.t.y = uniform_f64(T.t.y); r.t.z = uniform_f64(T.t.z);
  return r;
}

// ---- what a pass over a camera's patches needs of the camera, per level -------------------------------------------------
// The pass loop used to take it from the camera descriptor in global memory -- the two level images, the camera model,
// the feature range: ~35 scalars through scalar loads whose addresses depend on the level -- in EVERY pass: 2.2 - 2.7 K
// cycles per camera and pass by the cycle stamps (round 4), a fifth of a Gauss-Newton iteration of a 180-patch problem.
// Now the level's start copies them into LDS once (LevelDesc, one thread per camera), and a pass reads them back uniformly
// (ds_read + v_readfirstlane: values in scalar registers, as before).
// MEASURED (round 5, A/B on one box, profiles/r05_align_desc_vote_ab.txt) AND NOT KEPT: a 180-patch problem 5.33 -> 5.68 us per
// iteration with the descriptors in LDS alone, 5.38 - 5.42 with the vote folded as well; the batch of 1024 x 2000 within the
// box's run-to-run spread (1.19 - 1.23 ms all three builds).  The scalar loads hit the scalar cache and overlap the
// wave's other work; what the stamps attributed to them was the stamps' own s_waitcnt(0).  Both stay compiled out
// (SVOH_ALIGN_LDS_DESC=1 / SVOH_ALIGN_FOLD_VOTE=1 build them: same results, all tests green).
// SVOH_ALIGN_CLUSTER_XCD: the shares of a clustered problem placed on workgroups that share an XCD.  Built, measured
// (profiles/r05_align_cluster_xcd_ab.txt), compiled out: up to 8 shares per problem nothing changes (2000 patches 0.154 against
// 0.155 ms), with 16 - 32 shares one XCD is SLOWER than the eight the dispatcher spreads them over (8000 patches, 32 shares:
// 0.334 against 0.266 ms; 20000: 0.352 against 0.279) -- the shares' image reads and workspace rows then go through one L2 and
// one fabric port instead of eight, which costs more than the exchange slots meeting in one L2 saves (74 doubles per share and
// iteration).
#ifndef SVOH_ALIGN_CLUSTER_XCD
#define SVOH_ALIGN_CLUSTER_XCD 0
#endif
// SVOH_ALIGN_ONE_EACH: launches whose grid has a workgroup per problem skip the work queue.  Built, measured
// (profiles/r05_align_one_each_ab.txt: one 180-patch problem 5.35 us per iteration + 27.1 us either way), compiled out: the
// returning atomic on the queue head is not what a lone problem's launch pays for.
#ifndef SVOH_ALIGN_ONE_EACH
#define SVOH_ALIGN_ONE_EACH 0
#endif
#ifndef SVOH_ALIGN_LDS_DESC
#define SVOH_ALIGN_LDS_DESC 0
#endif
// ... and the visibility vote of a gradient-only pass rides the reduction's barrier instead of having one of its own
#ifndef SVOH_ALIGN_FOLD_VOTE
#define SVOH_ALIGN_FOLD_VOTE 0
#endif
struct LevelDesc {
  DevImage ref, cur;
  svoh_camera cam;
  int32_t n_features, feat_off;
};
struct CamPass {   // the uniform copy a pass works with
  svoh_camera cam;
  int32_t n_features, feat_off;
};
__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ DevImage uniform_image(const DevImage& im)
{
  DevImage r;
  const unsigned long long p = reinterpret_cast<unsigned long long>(im.data);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
  r.data = reinterpret_cast<const uint8_t*>(((unsigned long long)hi << 32) | lo);
  r.w = uniform_i32(im.w); r.h = uniform_i32(im.h); r.pitch = uniform_i32(im.pitch); r.pad = 0;
  return r;
}
__device__ __forceinline__ CamPass uniform_cam_pass(const LevelDesc& d)
{
  CamPass r;
  r.cam.fx = uniform_f64(d.cam.fx); r.cam.fy = uniform_f64(d.cam.fy); r.cam.cx = uniform_f64(d.cam.cx); r.cam.cy = uniform_f64(d.cam.cy);
#pragma unroll
  for (int k = 0; k < 4; ++k) r.cam.d[k] = uniform_f64(d.cam.d[k]);
  r.cam.distortion = uniform_i32(d.cam.distortion); r.cam.width = uniform_i32(d.cam.width); r.cam.height = uniform_i32(d.cam.height); r.cam.reserved = 0;
  r.n_features = uniform_i32(d.n_features); r.feat_off = uniform_i32(d.feat_off);
  return r;
}

constexpr int kWsPairs = 3;
__device__ __forceinline__ double* ws_pair(const AlignKernelArgs& a, int pair, int64_t gi) { return a.wpk + ((int64_t)pair * a.slots + gi) * 2; }
// The workspace as the passes of the 512-thread geometry see it: the launch's global arrays, or -- a problem of a few
// hundred features, what one camera stream aligns -- the workgroup's own copy in LDS (a generic pointer into the
// dynamic LDS area: the same code reads either).  A pass then starts with an LDS read instead of a round trip to L2 for
// its 48-byte row (measured on a 180-patch problem: 1.2 K of the 5.7 K cycles of a wave's pass).
// first: the launch-wide slot that is this view's slot 0 (every address formed stays inside the view's memory: a generic
// pointer that leaves the LDS aperture on the way, to come back with the index, is an aperture violation on gfx950)
struct WsView { double* base; int64_t slots; int64_t first; };
__device__ __forceinline__ double* ws_pair(const WsView& w, int pair, int64_t gi) { return w.base + ((int64_t)pair * w.slots + (gi - w.first)) * 2; }

// All patches of one camera at one Gauss-Newton iteration: one thread per patch.
// GONLY (see patch_moments): accumulate the gradient and chi2 only and report in `changed` whether any patch's
// visibility differs from the one recorded by the last full pass (the state value of its workspace row) -- the
// caller then repeats the iteration with a full pass.  A full pass records the visibility.
// jc: the camera's kJacConsts block (LDS).
template <int P, int D, int NT, bool LDS, bool GONLY = false, bool ROB = false>
__device__ __forceinline__ void accumulate_camera(
    const AlignKernelArgs& a, const WsView& ws, const CamPass& cd, const ImgView<LDS>& ref, const ImgView<LDS>& cur, int cw, int ch,
    const Rigid& Tcr, double scale, double one_plus_alpha, double beta_d, bool est_alpha, bool est_beta,
    bool robust, bool dist_jac, float weight_scale, int tid, const double* jc,
    double (&acc)[GONLY ? D + 1 : AccLayout<D>::NACC], int& nvis, int& changed, int stride = NT)
{
  SVOH_PASS_STAMP_BEGIN();
  const CamModel cm = load_camera(cd.cam);
  const double patch_center = (P - 1) / 2.0f;
  const double patch_center_wb = (P + 2 - 1) / 2.0f;
  SVOH_PASS_STAMP(0);

  // tid / stride: the lanes that take this camera and how many they are (the whole workgroup, or the camera's waves
  // when the cameras of a rig run side by side: run_cameras)
  for (int i = tid; i < cd.n_features; i += stride) {
    const int gi = cd.feat_off + i;
    const double2 vs = *reinterpret_cast<const double2*>(ws_pair(ws, 2, gi));
    if (vs.y == 0.0) continue;
    const double2 xy = *reinterpret_cast<const double2*>(ws_pair(ws, 0, gi));
    const double2 zu = *reinterpret_cast<const double2*>(ws_pair(ws, 1, gi));
    SVOH_PASS_STAMP(1);
    const Vec3 X = { xy.x, xy.y, zu.x };
    // ---- a-6 projection into the current level + visibility ----
    const Vec3 Y = transform(Tcr, X);
    bool vis = !(Y.z < 0.0);
    int cu = 0, cv = 0;
    double csu = 0.0, csv = 0.0;
    if (vis) {
      double u, v;
      project3(cm, Y, u, v);
      const double u_tl = u * scale - patch_center;
      const double v_tl = v * scale - patch_center;
      vis = !(u_tl < 0.0 || v_tl < 0.0 || u_tl + P + 2.0 >= cw || v_tl + P + 2.0 >= ch);
      vis = vis && u_tl == u_tl && v_tl == v_tl;  // a NaN passes the reference's test; never index with it
      if (vis) {
        const double fu = floor(u_tl), fv = floor(v_tl);
        cu = (int)fu; cv = (int)fv;
        csu = u_tl - cu; csv = v_tl - cv;
      }
    }
    if constexpr (GONLY) changed |= (int)((vs.y == 2.0) != vis);
    else {
      ws_pair(ws, 2, gi)[1] = vis ? 2.0 : 1.0;
      if (a.eval_level >= 0) a.wvis[gi] = vis ? 1 : 0;   // svoh_sparse_align_evaluate hands the mask out
    }
    if (!vis) continue;
    ++nvis;
    SVOH_PASS_STAMP(2);
    // ---- a-5 reference side (recomputed, never stored) ----
    const double ru_tl = zu.y * scale - patch_center_wb;
    const double rv_tl = vs.x * scale - patch_center_wb;
    const int ru = (int)floor(ru_tl), rv = (int)floor(rv_tl);
    const double rsu = ru_tl - ru, rsv = rv_tl - rv;
    double mom[AccLayout<D>::NMOM];
    patch_moments<P, D, LDS, LDS, GONLY, ROB>(ref, cur, ru, rv, rsu, rsv, cu, cv, csu, csv, one_plus_alpha, beta_d, robust,
                                              weight_scale, mom);
#if SVOH_PIN_MOMENTS
    // see accumulate_camera_rows: the moments are finished here, the pixel arithmetic is not sunk behind what follows.
    // On the LDS-resident levels only: where the rows come from global memory the interleaving hides their latency
    // (A/B on one box, 1024 x 2000 patches: levels 4..2 0.996 -> 0.904-0.954 ms with the pin, level 0 alone 0.787 -> 0.82-0.84)
    if constexpr (LDS) {
#pragma unroll
      for (int k = 0; k < AccLayout<D>::NMOM; ++k)
        if (!GONLY || k == 3 || k == 4 || k == 5 || k == 13 || k == 14) asm volatile("" : "+v"(mom[k]));
    }
#endif
    SVOH_PASS_STAMP(3);
    double ja[6], jb[6];
    jac_rows(jc, cm, dist_jac, X, scale, ja, jb);
    if constexpr (GONLY) accumulate_patch_gradient<D>(mom, ja, jb, est_alpha, est_beta, acc);
    else accumulate_patch<D>(mom, ja, jb, est_alpha, est_beta, acc);
#ifdef SVOH_PHASE_STAMPS
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
#endif
    SVOH_PASS_STAMP(4);
  }
}

// Rows geometry: the LPP lanes of a group (consecutive lanes, so a wave holds 64 / LPP patches) take one patch per pass,
// lane r its rows.  What belongs to the patch as a whole -- the workspace row (the group's lanes read the same 48
// bytes: one request), projection and visibility, the Jacobian rows -- is computed by every lane of the group alike:
// identical values, no exchange, and no divergence inside a group.  The patch is counted and its visibility recorded
// by the group's lane 0.
template <int P, int LPP, int D, int NT, bool LDS, bool GONLY = false, bool ROB = false>
__device__ __forceinline__ void accumulate_camera_rows(
    const AlignKernelArgs& a, const WsView& ws, const CamPass& cd, const ImgView<LDS>& ref, const ImgView<LDS>& cur, int cw, int ch,
    const Rigid& Tcr, double scale, double one_plus_alpha, double beta_d, bool est_alpha, bool est_beta,
    bool robust, bool dist_jac, float weight_scale, int tid, const double* jc,
    double (&acc)[GONLY ? D + 1 : AccLayout<D>::NACC], int& nvis, int& changed, int stride = NT)
{
  static_assert(LPP == 2 || LPP == 4 || LPP == 8, "a group is 2, 4 or 8 consecutive lanes");
  SVOH_PASS_STAMP_BEGIN();
  const CamModel cm = load_camera(cd.cam);
  const double patch_center = (P - 1) / 2.0f;
  const double patch_center_wb = (P + 2 - 1) / 2.0f;
  const int r = tid & (LPP - 1);
  SVOH_PASS_STAMP(0);

  for (int i = tid / LPP; i < cd.n_features; i += stride / LPP) {   // tid / stride: see accumulate_camera
    const int gi = cd.feat_off + i;
    const double2 vs = *reinterpret_cast<const double2*>(ws_pair(ws, 2, gi));
    if (vs.y == 0.0) continue;
    const double2 xy = *reinterpret_cast<const double2*>(ws_pair(ws, 0, gi));
    const double2 zu = *reinterpret_cast<const double2*>(ws_pair(ws, 1, gi));
    SVOH_PASS_STAMP(1);
    const Vec3 X = { xy.x, xy.y, zu.x };
    const Vec3 Y = transform(Tcr, X);
    bool vis = !(Y.z < 0.0);
    int cu = 0, cv = 0;
    double csu = 0.0, csv = 0.0;
    if (vis) {
      double u, v;
      project3(cm, Y, u, v);
      const double u_tl = u * scale - patch_center;
      const double v_tl = v * scale - patch_center;
      vis = !(u_tl < 0.0 || v_tl < 0.0 || u_tl + P + 2.0 >= cw || v_tl + P + 2.0 >= ch);
      vis = vis && u_tl == u_tl && v_tl == v_tl;
      if (vis) {
        const double fu = floor(u_tl), fv = floor(v_tl);
        cu = (int)fu; cv = (int)fv;
        csu = u_tl - cu; csv = v_tl - cv;
      }
    }
    if constexpr (GONLY) changed |= (int)((vs.y == 2.0) != vis);
    else if (r == 0) {
      ws_pair(ws, 2, gi)[1] = vis ? 2.0 : 1.0;
      if (a.eval_level >= 0) a.wvis[gi] = vis ? 1 : 0;
    }
    if (!vis) continue;
    nvis += (r == 0) ? 1 : 0;
    SVOH_PASS_STAMP(2);
    const double ru_tl = zu.y * scale - patch_center_wb;
    const double rv_tl = vs.x * scale - patch_center_wb;
    const int ru = (int)floor(ru_tl), rv = (int)floor(rv_tl);
    const double rsu = ru_tl - ru, rsv = rv_tl - rv;
    double mom[AccLayout<D>::NMOM];
    patch_rows_moments<P, LPP, D, LDS, LDS, GONLY, ROB>(ref, cur, ru, rv, rsu, rsv, cu, cv, csu, csv, r, one_plus_alpha, beta_d, robust,
                                                   weight_scale, mom);
    // the moments are finished HERE: left alone, the compiler sinks the pixel arithmetic (pure) behind the branches of
    // the Jacobian rows, towards its use, while the rows' bytes wait in registers -- spilled ones (measured: the pass of
    // two lanes per patch 20 % slower than the lane-per-patch pass without this line, 25 % faster with it)
#pragma unroll
    for (int k = 0; k < AccLayout<D>::NMOM; ++k)
      if (!GONLY || k == 3 || k == 4 || k == 5 || k == 13 || k == 14) asm volatile("" : "+v"(mom[k]));
    SVOH_PASS_STAMP(3);
    double ja[6], jb[6];
    jac_rows(jc, cm, dist_jac, X, scale, ja, jb);
    if constexpr (GONLY) accumulate_patch_gradient<D>(mom, ja, jb, est_alpha, est_beta, acc);
    else accumulate_patch<D>(mom, ja, jb, est_alpha, est_beta, acc);
#ifdef SVOH_PHASE_STAMPS
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
#endif
    SVOH_PASS_STAMP(4);
  }
}

// The same loop with the workspace row of every patch (48 bytes: xyz_ref, uv, state) brought in by LDS-DMA
// (global_load_lds_dwordx4: 16 bytes per lane straight into LDS, no VGPR and no wait at issue).  Without it the
// row's load sits on the patch's critical path: a round trip to L2 / Infinity Cache before the projection can
// start, with only two waves per SIMD to hide it.  Here the row of the NEXT patch is requested right after this
// patch's has been read and arrives while the pixel loop runs.  Two buffers per wave: xyz_ref is read again
// behind the pixel loop for the Jacobian rows (three doubles that need not stay in registers through it).
// stage: this wave's 2 x 3 x 64 x 16 B staging area.  Control flow is wave-uniform (every lane runs every pass).
constexpr int kStageDoubles = 2 * kWsPairs * 128;
template <int P, int D, int NT, bool LDS, bool GONLY = false, bool ROB = false>
__device__ __forceinline__ void accumulate_camera_staged(
    const AlignKernelArgs& a, const CamPass& cd, const ImgView<LDS>& ref, const ImgView<LDS>& cur, int cw, int ch,
    const Rigid& Tcr, double scale, double one_plus_alpha, double beta_d, bool est_alpha, bool est_beta,
    bool robust, bool dist_jac, float weight_scale, int tid, double* stage, const double* jc,
    double (&acc)[GONLY ? D + 1 : AccLayout<D>::NACC], int& nvis, int& changed, int stride = NT)
{
  typedef const __attribute__((address_space(1))) void* gptr;
  typedef __attribute__((address_space(3))) void* lptr;
  const CamModel cm = load_camera(cd.cam);
  const double patch_center = (P - 1) / 2.0f;
  const double patch_center_wb = (P + 2 - 1) / 2.0f;
  const int lane = tid & 63;   // tid / stride: see accumulate_camera (stride is a multiple of 64: a wave's lanes stay together)
  const int n = cd.n_features;
  const int n_round = (n + stride - 1) / stride * stride;
  auto request = [&](double* buf, int64_t gi) {
#pragma unroll
    for (int pr = 0; pr < kWsPairs; ++pr)
      __builtin_amdgcn_global_load_lds((gptr)ws_pair(a, pr, gi), (lptr)(buf + pr * 128), 16, 0, 0);
  };
  if (tid < n) request(stage, cd.feat_off + tid);
  int which = 0;
  for (int i = tid; i < n_round; i += stride, which ^= 1) {
    const int64_t gi = cd.feat_off + i;
    const bool in_range = i < n;
    double* buf = stage + which * (kWsPairs * 128);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this pass's row has landed in LDS
    const double2 xy = *reinterpret_cast<const double2*>(buf + 0 * 128 + lane * 2);
    const double2 zu = *reinterpret_cast<const double2*>(buf + 1 * 128 + lane * 2);
    const double2 vs = *reinterpret_cast<const double2*>(buf + 2 * 128 + lane * 2);
    // the other buffer: its last readers (the previous pass's Jacobian rows) are done -- lgkmcnt(0) at the loop's end
    if (i + stride < n) request(stage + (which ^ 1) * (kWsPairs * 128), gi + stride);
    const bool sel = in_range && vs.y != 0.0;
    bool vis = false;
    double mom[AccLayout<D>::NMOM];
    if (sel) {
      const Vec3 X = { xy.x, xy.y, zu.x };
      const Vec3 Y = transform(Tcr, X);
      vis = !(Y.z < 0.0);
      int cu = 0, cv = 0;
      double csu = 0.0, csv = 0.0;
      if (vis) {
        double u, v;
        project3(cm, Y, u, v);
        const double u_tl = u * scale - patch_center;
        const double v_tl = v * scale - patch_center;
        vis = !(u_tl < 0.0 || v_tl < 0.0 || u_tl + P + 2.0 >= cw || v_tl + P + 2.0 >= ch);
        vis = vis && u_tl == u_tl && v_tl == v_tl;
        if (vis) {
          const double fu = floor(u_tl), fv = floor(v_tl);
          cu = (int)fu; cv = (int)fv;
          csu = u_tl - cu; csv = v_tl - cv;
        }
      }
      if constexpr (GONLY) changed |= (int)((vs.y == 2.0) != vis);
      else {
        ws_pair(a, 2, gi)[1] = vis ? 2.0 : 1.0;
        if (a.eval_level >= 0) a.wvis[gi] = vis ? 1 : 0;
      }
      if (vis) {
        ++nvis;
        const double ru_tl = zu.y * scale - patch_center_wb;
        const double rv_tl = vs.x * scale - patch_center_wb;
        const int ru = (int)floor(ru_tl), rv = (int)floor(rv_tl);
        const double rsu = ru_tl - ru, rsv = rv_tl - rv;
        patch_moments<P, D, LDS, LDS, GONLY, ROB>(ref, cur, ru, rv, rsu, rsv, cu, cv, csu, csv, one_plus_alpha, beta_d,
                                                  robust, weight_scale, mom);
      }
    }
    if (vis) {
#if SVOH_PIN_MOMENTS
      if constexpr (LDS) {   // see accumulate_camera
#pragma unroll
        for (int k = 0; k < AccLayout<D>::NMOM; ++k)
          if (!GONLY || k == 3 || k == 4 || k == 5 || k == 13 || k == 14) asm volatile("" : "+v"(mom[k]));
      }
#endif

      asm volatile("" ::: "memory");   // xyz_ref comes from LDS again here instead of living through the pixel loop
      const double2 xy2 = *reinterpret_cast<const double2*>(buf + 0 * 128 + lane * 2);
      const double z2 = buf[1 * 128 + lane * 2];
      const Vec3 X = { xy2.x, xy2.y, z2 };
      double ja[6], jb[6];
      jac_rows(jc, cm, dist_jac, X, scale, ja, jb);
      if constexpr (GONLY) accumulate_patch_gradient<D>(mom, ja, jb, est_alpha, est_beta, acc);
      else accumulate_patch<D>(mom, ja, jb, est_alpha, est_beta, acc);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) before the next pass may request this buffer again
  }
}

// Cluster mode: replace vals[0 .. n) (LDS) by their sums over the workgroups of the cluster, added in share order
// so that every workgroup gets the same bits.  Slots alternate between two buffers by epoch: a workgroup can be
// one barrier ahead of the slowest, never two.  The wait is bounded; false = a partner never arrived (the
// caller gives up with status 3 instead of hanging the device).
template <int NT>
__device__ __forceinline__ bool cluster_sum(const AlignKernelArgs& a, int prob, int share, unsigned& epoch, double* vals, int n,
                                            int tid, int* s_flag)
{
  const int G = a.cluster;
  double* buf = a.xchg + ((size_t)prob * 2 + (epoch & 1u)) * G * kXchgStride;   // this problem's slots
  unsigned int* bar = a.bar + prob;
  if (tid < n) buf[(size_t)share * kXchgStride + tid] = vals[tid];
  __syncthreads();
  if (tid == 0) {
    __threadfence();   // this workgroup's slot is visible device-wide before it counts as arrived
    atomicAdd(bar, 1u);
    const unsigned target = (unsigned)G * (epoch + 1u);
    int ok = 0;
    for (long long spin = 0; spin < (1ll << 19); ++spin) {   // ~0.1 s; an exchange normally completes in ~2 us
      if (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    *s_flag = ok;
  }
  __syncthreads();
  const bool ok = *s_flag != 0;
  if (ok && tid < n) {
    __threadfence();   // acquire on the reading lane as well
    double v = 0.0;
    for (int g = 0; g < G; ++g) v += buf[(size_t)g * kXchgStride + tid];
    vals[tid] = v;
  }
  __syncthreads();
  ++epoch;
  return ok;
}

// One Gauss-Newton bookkeeping step, run by a single lane: prior, pivoted LDL^T, SE3
// update, convergence (MiniLeastSquaresSolver::optimizeGaussNewton,
// mini_least_squares_solver.hpp:42-107).  Kept out of line on purpose: inlined into the
// kernel its ~3000 instructions and ~100 live registers share one register allocation
// with the patch loop, and both ends pay for it with spills.
template <int P, int D, bool ILLUM>
__device__ __attribute__((noinline)) void gn_serial_step(const AlignKernelArgs& a, const DevProblemDesc& pb,
                                                         const DevCamDesc* cams, int n_cams, int pbi, int level,
                                                         int iter, bool eval_mode, bool reuse_factor)
{
  constexpr int NH = AccLayout<D>::NH;
  const svoh_align_options& opt = a.opt;
  ShState& s = g_state;
  const double (&s_sum)[45] = g_sum;
  int& s_nvis = g_nvis;

#ifdef SVOH_PHASE_STAMPS
    long long st_s0 = (long long)__builtin_amdgcn_s_memtime();
#endif
    const int n_meas = s_nvis * P * P;
    s.patch_iters += s_nvis;
    s_nvis = 0;
    const double chi2 = s_sum[NH + D] / (double)n_meas;
    double m[36], xg[8];
#pragma unroll
    for (int k = 0; k < 36; ++k) m[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) xg[k] = 0.0;
    {
      int idx = 0;
#pragma unroll
      for (int r = 0; r < D; ++r)
#pragma unroll
        for (int c2 = r; c2 < D; ++c2) SVOH_L(c2, r) = s_sum[idx++];  // H(r,c2) = H(c2,r)
#pragma unroll
      for (int r = 0; r < D; ++r) xg[r] = s_sum[NH + r];
    }
    if (level < SVOH_MAX_LEVELS) {
      s.lvl_iters[level] = iter + 1;
      s.lvl_n_meas[level] = n_meas;
      s.lvl_chi2[level] = chi2;
    }
    if (eval_mode) {
      double* eo = a.eval_out + 74 * pbi;   // one block per problem (the shares of a patch-split evaluation)
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c2 = 0; c2 <= r; ++c2) {
          eo[c2 * 8 + r] = SVOH_L(r, c2);
          eo[r * 8 + c2] = SVOH_L(r, c2);
        }
#pragma unroll
      for (int k = 0; k < 8; ++k) eo[64 + k] = xg[k];
      eo[72] = a.raw_sums ? s_sum[NH + D] : chi2;
      eo[73] = (double)n_meas;
      s.level_done = 1;
    } else {
      if (s.prior.have_prior) {
        // SparseImgAlignBase::applyPrior (sparse_img_align_base.cpp:77-107)
        if (iter == 0) {
          double mt = 0, mr = 0;
#pragma unroll
          for (int j = 0; j < 3; ++j) mt = fmax(mt, fabs(SVOH_L(j, j)));
#pragma unroll
          for (int j = 3; j < 6; ++j) mr = fmax(mr, fabs(SVOH_L(j, j)));
          for (int j = 0; j < 3; ++j) s.I_prior[j] = 1.0 * s.prior.lambda_trans * mt;
          for (int j = 3; j < 6; ++j) s.I_prior[j] = 1.0 * s.prior.lambda_rot * mr;
          s.I_prior[6] = s.prior.lambda_alpha * SVOH_L(6, 6);
          s.I_prior[7] = s.prior.lambda_beta * SVOH_L(7, 7);
        }
        if (!reuse_factor) {
#pragma unroll
          for (int j = 0; j < 8; ++j) SVOH_L(j, j) += s.I_prior[j];
        }
        double lg[6];
        rigid_log(mul(inverse(load_rigid(s.prior.T_prior)), s.T), lg);
#pragma unroll
        for (int j = 0; j < 6; ++j) xg[j] += s.I_prior[j] * lg[j];
        xg[6] += s.I_prior[6] * (s.prior.alpha_prior - s.alpha);
        xg[7] += s.I_prior[7] * (s.prior.beta_prior - s.beta);
      }
      SVOH_SERIAL_STAMP(0);
      // without illumination terms rows/columns 6 and 7 are exactly zero: the
      // pivoted factorisation never selects them before the six pose pivots and
      // they contribute exact zeros, so the 6x6 leading block gives the same bits
      // The factor is computed when the normal matrix is new (a full pass) and kept in s.fact for the iterations
      // that only refresh the gradient: same matrix, same factor, same bits as factorising again.
      if constexpr (ILLUM) {
        int tr[8];
        bool nonzero;
        if (!reuse_factor) {
          nonzero = ldlt_factor_regs<8>(m, tr);
#pragma unroll
          for (int k = 0; k < 36; ++k) s.fact[k] = m[k];
#pragma unroll
          for (int k = 0; k < 8; ++k) s.fact_tr[k] = tr[k];
          s.fact_nonzero = nonzero ? 1 : 0;
        } else {
#pragma unroll
          for (int k = 0; k < 36; ++k) m[k] = s.fact[k];
#pragma unroll
          for (int k = 0; k < 8; ++k) tr[k] = s.fact_tr[k];
          nonzero = s.fact_nonzero != 0;
        }
        if (!ldlt_apply_regs<8>(m, tr, nonzero, xg)) s.stop = 1;
      } else {
        double m6[21], x6[6];
        int tr[6];
        bool nonzero;
        if (!reuse_factor) {
#pragma unroll
          for (int k = 0; k < 21; ++k) m6[k] = m[k];
          nonzero = ldlt_factor_regs<6>(m6, tr);
#pragma unroll
          for (int k = 0; k < 21; ++k) s.fact[k] = m6[k];
#pragma unroll
          for (int k = 0; k < 6; ++k) s.fact_tr[k] = tr[k];
          s.fact_nonzero = nonzero ? 1 : 0;
        } else {
#pragma unroll
          for (int k = 0; k < 21; ++k) m6[k] = s.fact[k];
#pragma unroll
          for (int k = 0; k < 6; ++k) tr[k] = s.fact_tr[k];
          nonzero = s.fact_nonzero != 0;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) x6[k] = xg[k];
        if (!ldlt_apply_regs<6>(m6, tr, nonzero, x6)) s.stop = 1;
#pragma unroll
        for (int k = 0; k < 6; ++k) xg[k] = x6[k];
        xg[6] = 0.0; xg[7] = 0.0;
      }
      SVOH_SERIAL_STAMP(1);
      if (s.stop) {
        // rollback (mini_least_squares_solver.hpp:73-82); stop_ is only cleared by reset()
        s.T = s.Told; s.alpha = s.alpha_old; s.beta = s.beta_old;
        s.status = 2;
        s.level_done = 1;
      } else {
        // SparseImgAlignBase::update (sparse_img_align_base.cpp:64-75)
        double mdx[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) mdx[j] = -xg[j];
        Rigid Tn = mul(s.T, rigid_exp(mdx));
        const double an = (s.alpha - xg[6]) / (1.0 + xg[6]);
        const double bn = (s.beta - xg[7]) / (1.0 + xg[6]);
        Tn.q = normalized(Tn.q);
        s.Told = s.T; s.alpha_old = s.alpha; s.beta_old = s.beta;
        s.T = Tn; s.alpha = an; s.beta = bn;
        double x_norm = -1.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const double v = fabs(xg[j]); if (v > x_norm) x_norm = v; }
        if (x_norm < opt.eps) s.level_done = 1;
      }
      SVOH_SERIAL_STAMP(2);
      for (int c = 0; c < n_cams; ++c)
        s.Tcr[c] = mul(mul(s.cam_cur_T_cam_imu[c], s.T), s.cam_ref_T_imu_cam[c]);
      s.alpha_f = (float)s.alpha; s.beta_f = (float)s.beta;
      SVOH_SERIAL_STAMP(3);
    }
  }

// The same step by the 64 lanes of the workgroup's first wave (round 4).  On one lane the step is ~1 500 instructions
// issued one after the other while every other wave of the workgroup waits at the barrier behind it: 31 % of a single
// 180-patch problem's kernel, 14 % of a workgroup's life in the batch.  What one lane must do stays on lane 0 and runs
// only when it is needed -- the factorisation when the level's Hessian is new, the prior's share of the gradient (a
// logarithm) when there is a prior.  The rest is spread: the substitution by lanes 0 .. N-1 (ldlt_apply_wave: the
// permutation is an index, the N divisions are one, a forward step is one multiply-add for all lanes behind it), the
// four divisions of the quaternion's normalisation and the two of the illumination update in ONE division slot
// (lanes 0 .. 5), the cameras' poses one camera per lane.  Every value is computed by the operations of the one-lane
// step in their order: the states are the same to the last bit or two (multiply-adds the compiler contracts either way).
template <int P, int D, bool ILLUM>
__device__ __attribute__((noinline)) void gn_wave_step(const AlignKernelArgs& a, int n_cams, int level, int iter,
                                                       bool reuse_factor, int lane)
{
  constexpr int NH = AccLayout<D>::NH;
  constexpr int N = ILLUM ? 8 : 6;
  const svoh_align_options& opt = a.opt;
  ShState& s = g_state;
  const int nvis = g_nvis;
  const int n_meas = nvis * P * P;
  const double chi2 = g_sum[NH + D] / (double)n_meas;
  const bool have_prior = s.prior.have_prior != 0;
  const int stop_before = s.stop;
#ifdef SVOH_PHASE_STAMPS
  long long st_s0 = (long long)__builtin_amdgcn_s_memtime();
#define SVOH_WAVE_STAMP(k) do { __builtin_amdgcn_s_waitcnt(0); if (lane == 0) { long long st_n = (long long)__builtin_amdgcn_s_memtime(); s.dbg[k] += st_n - st_s0; st_s0 = st_n; } else { st_s0 = (long long)__builtin_amdgcn_s_memtime(); } } while (0)
#else
#define SVOH_WAVE_STAMP(k) do {} while (0)
#endif
  __builtin_amdgcn_wave_barrier();   // every lane has read what lane 0 is about to overwrite
  if (lane == 0) {
    s.patch_iters += nvis;
    g_nvis = 0;
    if (level < SVOH_MAX_LEVELS) {
      s.lvl_iters[level] = iter + 1;
      s.lvl_n_meas[level] = n_meas;
      s.lvl_chi2[level] = chi2;
    }
    if (!reuse_factor || have_prior) {
      double m[36];
#pragma unroll
      for (int k = 0; k < 36; ++k) m[k] = 0.0;
      if (!reuse_factor || iter == 0) {
        int idx = 0;
#pragma unroll
        for (int r = 0; r < D; ++r)
#pragma unroll
          for (int c2 = r; c2 < D; ++c2) SVOH_L(c2, r) = g_sum[idx++];
      }
      if (have_prior) {
        // SparseImgAlignBase::applyPrior (sparse_img_align_base.cpp:77-107), as in gn_serial_step
        if (iter == 0) {
          double mt = 0, mr = 0;
#pragma unroll
          for (int j = 0; j < 3; ++j) mt = fmax(mt, fabs(SVOH_L(j, j)));
#pragma unroll
          for (int j = 3; j < 6; ++j) mr = fmax(mr, fabs(SVOH_L(j, j)));
          for (int j = 0; j < 3; ++j) s.I_prior[j] = 1.0 * s.prior.lambda_trans * mt;
          for (int j = 3; j < 6; ++j) s.I_prior[j] = 1.0 * s.prior.lambda_rot * mr;
          s.I_prior[6] = s.prior.lambda_alpha * SVOH_L(6, 6);
          s.I_prior[7] = s.prior.lambda_beta * SVOH_L(7, 7);
        }
        if (!reuse_factor) {
#pragma unroll
          for (int j = 0; j < 8; ++j) SVOH_L(j, j) += s.I_prior[j];
        }
        double lg[6];
        rigid_log(mul(inverse(load_rigid(s.prior.T_prior)), s.T), lg);
#pragma unroll
        for (int j = 0; j < 6; ++j) s.prior_g[j] = s.I_prior[j] * lg[j];
        s.prior_g[6] = s.I_prior[6] * (s.prior.alpha_prior - s.alpha);
        s.prior_g[7] = s.I_prior[7] * (s.prior.beta_prior - s.beta);
      }
      if (!reuse_factor) {
        if constexpr (ILLUM) {
          int tr[8];
          const bool nonzero = ldlt_factor_regs<8>(m, tr);
#pragma unroll
          for (int k = 0; k < 36; ++k) s.fact[k] = m[k];
#pragma unroll
          for (int k = 0; k < 8; ++k) s.fact_tr[k] = tr[k];
          ldlt_perm_from_transpositions<8>(tr, s.fact_perm);
          s.fact_nonzero = nonzero ? 1 : 0;
        } else {
          double m6[21];
          int tr[6];
#pragma unroll
          for (int k = 0; k < 21; ++k) m6[k] = m[k];
          const bool nonzero = ldlt_factor_regs<6>(m6, tr);
#pragma unroll
          for (int k = 0; k < 21; ++k) s.fact[k] = m6[k];
#pragma unroll
          for (int k = 0; k < 6; ++k) s.fact_tr[k] = tr[k];
          ldlt_perm_from_transpositions<6>(tr, s.fact_perm);
          s.fact_nonzero = nonzero ? 1 : 0;
        }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  SVOH_WAVE_STAMP(0);
  // ---- the substitution, lane i < N its component ----
  {
    const int pi = s.fact_perm[lane < N ? lane : 0];
    double rhs = g_sum[NH + pi];
    if (have_prior) rhs += s.prior_g[pi];
    const double x = ldlt_apply_wave<N>(s.fact, s.fact_nonzero != 0, rhs, lane);
    if (lane < N) s.dx[pi] = x;
    else if (lane < 8) s.dx[lane] = 0.0;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double xg[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) xg[k] = s.dx[k];
  SVOH_WAVE_STAMP(1);
  const bool stop = stop_before != 0 || xg[0] != xg[0];   // solveDefaultImpl: failure iff dx[0] is NaN; stop_ is sticky
  Rigid T = s.T;
  double alpha = s.alpha, beta = s.beta;
  int level_done = 0;
  if (stop) {
    // rollback (mini_least_squares_solver.hpp:73-82)
    T = s.Told; alpha = s.alpha_old; beta = s.beta_old;
    level_done = 1;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) { s.stop = 1; s.T = T; s.alpha = alpha; s.beta = beta; s.status = 2; s.level_done = 1; }
  } else {
    // SparseImgAlignBase::update (sparse_img_align_base.cpp:64-75)
    double mdx[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) mdx[j] = -xg[j];
    Rigid Tn = mul(T, rigid_exp(mdx));
    const double nrm = sqrt(sqnorm(Tn.q));
    // one division slot: lanes 0..3 the quaternion's components over its norm, lanes 4, 5 the illumination terms
    const double num = lane == 0 ? Tn.q.w : lane == 1 ? Tn.q.x : lane == 2 ? Tn.q.y : lane == 3 ? Tn.q.z
                     : lane == 4 ? (alpha - xg[6]) : (beta - xg[7]);
    const double den = lane < 4 ? nrm : (1.0 + xg[6]);
    const double qd = num / den;
    Tn.q.w = wave_bcast_f64(qd, 0); Tn.q.x = wave_bcast_f64(qd, 1); Tn.q.y = wave_bcast_f64(qd, 2); Tn.q.z = wave_bcast_f64(qd, 3);
    const double an = wave_bcast_f64(qd, 4), bn = wave_bcast_f64(qd, 5);
    double x_norm = -1.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const double v = fabs(xg[j]); if (v > x_norm) x_norm = v; }
    if (x_norm < opt.eps) level_done = 1;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      s.Told = T; s.alpha_old = alpha; s.beta_old = beta;
      s.T = Tn; s.alpha = an; s.beta = bn;
      if (level_done) s.level_done = 1;
    }
    T = Tn; alpha = an; beta = bn;
  }
  SVOH_WAVE_STAMP(2);
  // ---- the cameras' poses, one camera per lane ----
  {
    const int c = lane < n_cams ? lane : 0;
    const Rigid Tcr = mul(mul(s.cam_cur_T_cam_imu[c], T), s.cam_ref_T_imu_cam[c]);
    if (lane < n_cams) s.Tcr[c] = Tcr;
    if (lane == 0) { s.alpha_f = (float)alpha; s.beta_f = (float)beta; }
  }
  SVOH_WAVE_STAMP(3);
}

#ifndef SVOH_ALIGN_STAGED
#define SVOH_ALIGN_STAGED 1
#endif
#ifndef SVOH_ALIGN_REUSE_HESSIAN
#define SVOH_ALIGN_REUSE_HESSIAN 1
#endif
#ifndef SVOH_ALIGN_MIN_WAVES_256
#define SVOH_ALIGN_MIN_WAVES_256 2
#endif
// the lanes per patch a small problem gets by default, as a function of what fits (measured: DESIGN.md 4.2)
#ifndef SVOH_ALIGN_ROWS_DEFAULT
#define SVOH_ALIGN_ROWS_DEFAULT(fit) (fit)
#endif

// CLUSTER: the cluster mode's exchanges are compiled in (256-thread geometry only); the batch instantiation stays
// free of them -- as run-time branches they cost the batch kernel 5 %.
// ROBUST: the Tukey weights (SparseImgAlignOptions::robustification, off in the reference's handlers) as a compile-time
// switch: as a run-time flag the per-pixel branch, its float division and the weighted forms of the moments stayed in
// the full pass of every launch (320 instructions per patch row against 106 in the gradient-only pass); the robust
// instantiation in turn has no gradient-only pass (the weights change every iteration).
// LPP > 1: the rows geometry (accumulate_camera_rows: LPP lanes per patch), in the 512-thread workgroup of the latency
// mode (few problems), at the same 256 registers.
// LAT: the 256-thread kernel built for ONE wave per SIMD (a launch of fewer problems than compute units: nothing shares
// the SIMD anyway).  With the whole register file to itself the kernel spills nothing (274-302 registers instead of
// 256 + 26-56 spilled), and in a launch that small a spilled register is a round trip to memory nobody hides: the
// full passes of a single 180-patch problem took twice the cycles of its gradient-only passes.
// RIG: the instantiation for launches of a few problems of SEVERAL cameras (stereo bundles): cameras side by side on the
// workgroup's waves (run_cameras).  A template parameter, not a run-time branch of every kernel: the bookkeeping costs a
// single-camera problem 1.5 - 4 % of its kernel (measured: register allocation and a scalar-load dependency per pass),
// which the mono chain should not pay.
template <int P, int NT, bool ILLUM, bool CLUSTER = false, bool ROBUST = false, int LPP = 1, bool LAT = false, bool RIG = false>
__global__ __launch_bounds__(NT, (NT == 256 ? (LAT ? 1 : SVOH_ALIGN_MIN_WAVES_256) : 2))
void sparse_align_kernel(const AlignKernelArgs a)
{
  static_assert(!LAT || (NT == 256 && !CLUSTER && LPP == 1), "latency build: the 256-thread lane-per-patch geometry");
  static_assert(!RIG || (!CLUSTER && LPP == 1 && (LAT || NT == 512)), "rig build: the two lane-per-patch geometries of small launches");
  static_assert(LPP == 1 || (!CLUSTER && NT == 512 && LPP <= P), "rows geometry: 512 threads, no cluster mode");
  static_assert(NT == 256 || NT == 512, "workgroups of 256 or 512 threads");
  constexpr bool ROWS = LPP > 1;
  constexpr int D = ILLUM ? 8 : 6;
  constexpr int NACC = AccLayout<D>::NACC;
  constexpr int NW = NT / 64;

  // LDS-DMA staging of the workspace rows (accumulate_camera_staged) in the batch geometry; the wide
  // geometries keep their LDS for finer image levels and read the workspace with ordinary loads
  constexpr bool STAGED = SVOH_ALIGN_STAGED && NT == 256 && !ROWS;
  // the levels' images through registers in the BATCH geometry only (stage_item_list): a workgroup there has a second one beside it on its compute
  // unit to fill the wait; the small-launch geometries keep the LDS-DMA requests, whose flight overlaps the problem's base phase (measured, one
  // 180-patch problem: 27.6 us outside the iterations with LDS-DMA, 31.3 with the blocking copy)
  constexpr bool kStageRegs = SVOH_ALIGN_STAGE_REGS != 0 && NT == 256 && !LAT && !CLUSTER && !ROWS;
  extern __shared__ __align__(16) unsigned char lds_img[];
  __shared__ __align__(16) double s_stage[STAGED ? NW * kStageDoubles : 2];
  __shared__ double s_jc[SVOH_MAX_CAMS][kJacConsts];   // per-camera constants of jac_rows
  __shared__ int s_lvl_off[SVOH_MAX_LEVELS];           // where the level's images start in lds_img; -1: not resident
  __shared__ double s_red[NW][NACC];
  static_assert(NACC <= 45, "g_sum is sized for the 8-parameter case");
  __shared__ double s_x[kXchgStride];   // cluster mode: the block handed to cluster_sum
  __shared__ int s_cluster_ok;
  [[maybe_unused]] __shared__ __align__(16) LevelDesc s_lvd[SVOH_ALIGN_LDS_DESC ? SVOH_MAX_CAMS : 1];   // the cameras at the current level (see LevelDesc)
  [[maybe_unused]] __shared__ int s_changed[SVOH_ALIGN_FOLD_VOTE ? NW : 1];   // a gradient-only pass' visibility vote, wave by wave (SVOH_ALIGN_FOLD_VOTE)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  __shared__ int s_pbi;
  __shared__ StageItem s_stage_items[kMaxStageItems];   // the images of the problem's resident levels (stage_item_list)
  // Persistent workgroups: the grid is sized to what is resident at once and every
  // workgroup pulls the next problem index from a queue head, so problems that need
  // more Gauss-Newton iterations do not leave CUs idle at the end of the launch.
#if SVOH_ALIGN_CLUSTER_XCD
  bool first_pull = true;
#endif
#if SVOH_ALIGN_ONE_EACH
  bool pulled_once = false;
#endif
  for (;;) {
  __syncthreads();  // everyone is done with the previous problem's shared state
#if SVOH_ALIGN_CLUSTER_XCD
  // Cluster mode: a workgroup takes exactly one share, and WHICH one follows from its block index -- the shares of a problem go
  // to blocks b, b + 8, b + 16, ..., which the dispatcher is observed to deal to ONE XCD (round robin over the eight: a label
  // for speed, never for correctness): the shares' exchange slots and the images they all read then meet in one L2.  The
  // launch's grid is 8 x cluster x ceil(problems / 8) blocks; blocks beyond the last problem leave at once.
  if constexpr (CLUSTER) {
    if (tid == 0) {
      int v = a.n_problems;
      if (first_pull) {
        const int x = (int)(blockIdx.x & 7u), srow = (int)(blockIdx.x >> 3);
        const int c = x + 8 * (srow / a.cluster);
        if (c < a.n_problems / a.cluster) v = c * a.cluster + srow % a.cluster;
      }
      s_pbi = v;
    }
    first_pull = false;
  } else
#endif
#if SVOH_ALIGN_ONE_EACH
  // a launch with a workgroup per problem (one problem alone, a camera stream's frame, the problems of a lock-step round): the
  // problem index is the block index -- no returning atomic on the queue head ahead of the first descriptor load
  if (a.one_each) {
    if (tid == 0) s_pbi = pulled_once ? a.n_problems : (int)blockIdx.x;
    pulled_once = true;
  } else
#endif
  if (tid == 0) s_pbi = atomicAdd(a.queue, 1);
  __syncthreads();
  const int pbi = s_pbi;
  if (pbi >= a.n_problems) break;
#ifdef SVOH_TEST_HOOKS
  if (CLUSTER && a.cluster_test_absent && pbi % a.cluster == 1) continue;   // a partner that never arrives
#endif
  const DevProblemDesc& pb = a.problems[pbi];
  const DevCamDesc* cams = a.cams + pb.cam_begin;
  const int n_cams = pb.n_cams;
  const svoh_align_options& opt = a.opt;
  const bool eval_mode = a.eval_level >= 0;
  const int level_hi = eval_mode ? a.eval_level : opt.max_level;
  const int level_lo = eval_mode ? a.eval_level : opt.min_level;
  // LDS bytes of a level's images (all cameras), and their copy into lds_img + off (requests only: no wait)
  auto level_bytes = [&](int l) {
    int need = 0;
    for (int c = 0; c < n_cams; ++c) {
      need += ((cams[c].ref[l].w * cams[c].ref[l].h + 15) & ~15);
      need += ((cams[c].cur[l].w * cams[c].cur[l].h + 15) & ~15);
    }
    return need;
  };
  auto stage_level = [&](int l, int off) {
    for (int c = 0; c < n_cams; ++c) {
      const int off_cur = off + ((cams[c].ref[l].w * cams[c].ref[l].h + 15) & ~15);
      stage_image_pair<NT, kStageRegs>(lds_img + off, cams[c].ref[l], lds_img + off_cur, cams[c].cur[l], tid);
      off = off_cur + ((cams[c].cur[l].w * cams[c].cur[l].h + 15) & ~15);
    }
  };

  SVOH_STAMP_DECL
  SVOH_STAMP_START();
  // The images of the coarse levels are requested NOW, for the problem's whole life, from the coarsest level down
  // as long as they fit side by side (640x480: levels 4, 3, 2 = 50 400 bytes): their way from HBM overlaps the base
  // phase instead of standing in front of every level's first iteration (three exposed copies per problem before:
  // 6 % of a workgroup's life).  A finer level that fits only by itself is staged when its turn comes, over them.
  {
    // (uniform) which levels stay resident, and the table of their images; the list form wants images it can take in 16-byte pieces
    int off = 0, n_items = 0, total16 = 0;
    bool room = true, list_ok = kStageRegs;
    for (int l = level_hi; l >= level_lo; --l) {
      const int need = level_bytes(l);
      room = room && off + need <= a.lds_img_bytes;
      if (!room) continue;
      for (int c = 0; c < n_cams; ++c)
        for (int h = 0; h < 2; ++h) {
          const DevImage& im = h ? cams[c].cur[l] : cams[c].ref[l];
          list_ok = list_ok && n_items < kMaxStageItems && im.pitch == im.w && (reinterpret_cast<uintptr_t>(im.data) & 15) == 0;
          ++n_items;
        }
      off += need;
    }
    off = 0; room = true; n_items = 0;
    for (int l = level_hi; l >= level_lo; --l) {
      const int need = level_bytes(l);
      room = room && off + need <= a.lds_img_bytes;
      if (room) {
        if (!list_ok) stage_level(l, off);
        else if (tid == 0) {
          int o = off;
          for (int c = 0; c < n_cams; ++c)
            for (int h = 0; h < 2; ++h) {
              const DevImage& im = h ? cams[c].cur[l] : cams[c].ref[l];
              const int bytes = im.w * im.h;
              s_stage_items[n_items] = StageItem{ im.data, o, total16, bytes, 0 };
              ++n_items; total16 += bytes >> 4; o += (bytes + 15) & ~15;
            }
        } else {
          for (int c = 0; c < n_cams; ++c)
            for (int h = 0; h < 2; ++h) { const DevImage& im = h ? cams[c].cur[l] : cams[c].ref[l]; ++n_items; total16 += (im.w * im.h) >> 4; }
        }
        if (tid == 0) s_lvl_off[l] = off;
        off += need;
      }
      else if (tid == 0) s_lvl_off[l] = -1;
    }
    if (list_ok && n_items) {
      __syncthreads();   // the table is written
      stage_item_list<NT>(lds_img, s_stage_items, n_items, total16, tid);
    }
  }
#ifdef SVOH_STAGE_WAIT_EXPERIMENT
  // diagnostic build only (with SVOH_PHASE_STAMPS): wait for the levels' LDS-DMA right here and book it as "stage" -- how long the 50 KB take by themselves
  __builtin_amdgcn_s_waitcnt(0x0F70);
  SVOH_STAMP_ADD(1);
#endif
  if (tid == 0) {
    g_state.T = load_rigid(pb.T_init);
    g_state.Told = g_state.T;
    g_state.alpha = pb.alpha_init; g_state.beta = pb.beta_init;
    if (a.ext_state) {
      g_state.T = load_rigid(a.ext_state->T_icur_iref);
      g_state.Told = g_state.T;
      g_state.alpha = a.ext_state->alpha; g_state.beta = a.ext_state->beta;
    }
    g_state.alpha_old = g_state.alpha; g_state.beta_old = g_state.beta;
    g_state.stop = 0; g_state.level_done = 0; g_state.nsel = 0; g_state.status = 0; g_state.patch_iters = 0;
    for (int k = 0; k < 8; ++k) g_state.I_prior[k] = 0.0;
    g_nvis = 0;
    g_state.prior = pb.prior;
    for (int c = 0; c < n_cams; ++c) {
      g_state.cam_cur_T_cam_imu[c] = load_rigid(cams[c].cur_T_cam_imu);
      g_state.cam_ref_T_imu_cam[c] = load_rigid(cams[c].ref_T_imu_cam);
      to_matrix(g_state.cam_ref_T_imu_cam[c].q, &s_jc[c][0]);
      s_jc[c][9] = g_state.cam_ref_T_imu_cam[c].t.x; s_jc[c][10] = g_state.cam_ref_T_imu_cam[c].t.y;
      s_jc[c][11] = g_state.cam_ref_T_imu_cam[c].t.z;
      to_matrix(load_rigid(cams[c].ref_T_cam_imu).q, &s_jc[c][12]);
    }
#ifdef SVOH_PHASE_STAMPS
    for (int k = 0; k < 4; ++k) g_state.dbg[k] = 0;
    for (int k = 0; k < 8; ++k) g_state.dbg2[k] = 0;
#endif
    for (int l = 0; l < SVOH_MAX_LEVELS; ++l) { g_state.lvl_iters[l] = 0; g_state.lvl_n_meas[l] = 0; g_state.lvl_chi2[l] = 0.0; }
  }
  __syncthreads();

  WsView ws = { a.wpk, a.slots, 0 };
  if constexpr (NT == 512) {
    if (a.ws_lds_bytes > 0) {   // this problem's rows (all cameras: consecutive slots from cams[0].feat_off on) in LDS
      int n_local = 0;
      for (int c = 0; c < n_cams; ++c) n_local += cams[c].n_features;
      if (n_local * kWsPairs * 16 <= a.ws_lds_bytes) {   // (the host sized the area for the launch's largest problem)
        ws.base = reinterpret_cast<double*>(lds_img + a.lds_img_bytes);
        ws.slots = n_local;
        ws.first = cams[0].feat_off;
      }
    }
  }
  // ---- a-3 extractFeaturesSubset + a-4 precomputeBaseCaches (depth, xyz_ref) ----
  {
    int my_sel = 0;
    const int patch_size_wb = P + 2;
    const double scale = 1.0f / (1 << opt.max_level);
    const double patch_center_wb = (patch_size_wb - 1) / 2.0f;
    for (int c = 0; c < n_cams; ++c) {
      const DevCamDesc& cd = cams[c];
      const int rows_minus_two = cd.ref[opt.max_level].h - 2;
      const int cols_minus_two = cd.ref[opt.max_level].w - 2;
      // all of a feature's inputs are requested together (they are independent; behind the selection test each
      // would be its own round trip to memory), and one feature ahead
      struct FeatIn { double pu, pv, pwx, pwy, pwz, fx, fy, fz; unsigned flag; };
      auto request = [&](int k, FeatIn& o) {
        o.flag = cd.flags[k];
        o.pu = cd.px[2 * k]; o.pv = cd.px[2 * k + 1];
        o.pwx = cd.pos_world[3 * k + 0]; o.pwy = cd.pos_world[3 * k + 1]; o.pwz = cd.pos_world[3 * k + 2];
        o.fx = cd.f[3 * k + 0]; o.fy = cd.f[3 * k + 1]; o.fz = cd.f[3 * k + 2];
      };
      // kBaseDepth features per thread are requested together and then worked off: their inputs come cold from HBM
      // (65 bytes per feature, read once), and the vector-memory counter is in order -- with one feature requested ahead
      // every trip waited for the PREVIOUS trip's stores as well (round 4 stamps: 99 K cycles per 2000-feature problem,
      // 12 K per trip: 9 % of a workgroup's life in the batch)
#ifndef SVOH_ALIGN_BASE_DEPTH
#define SVOH_ALIGN_BASE_DEPTH 4
#endif
      constexpr int kBaseDepth = SVOH_ALIGN_BASE_DEPTH;
      for (int i0 = tid; i0 < cd.n_features; i0 += kBaseDepth * NT) {
        FeatIn in[kBaseDepth];
#pragma unroll
        for (int k = 0; k < kBaseDepth; ++k) {
          in[k] = FeatIn{ 0, 0, 0, 0, 0, 0, 0, 0, 0u };
          if (i0 + k * NT < cd.n_features) request(i0 + k * NT, in[k]);
        }
#pragma unroll
        for (int k = 0; k < kBaseDepth; ++k) {
          const int i = i0 + k * NT;
          if (i >= cd.n_features) break;
          const int gi = cd.feat_off + i;
          bool sel = in[k].flag != 0;
          const double pu = in[k].pu, pv = in[k].pv;
          const double pwx = in[k].pwx, pwy = in[k].pwy, pwz = in[k].pwz;
          const double fx_ = in[k].fx, fy_ = in[k].fy, fz_ = in[k].fz;
          if (sel) {
            const double u_tl = pu * scale - patch_center_wb;
            const double v_tl = pv * scale - patch_center_wb;
            const int u_tl_i = (int)floor(u_tl);
            const int v_tl_i = (int)floor(v_tl);
            // same test as sparse_img_align.cpp:221-225 with the constant moved to the right-hand side:
            // a wild pixel saturates the int conversion and `+ patch_size_wb` must not wrap around
            sel = !(u_tl_i < 0 || v_tl_i < 0 || u_tl_i >= cols_minus_two - patch_size_wb ||
                    v_tl_i >= rows_minus_two - patch_size_wb);
          }
          a.wsel[gi] = sel ? 1 : 0;
          a.wvis[gi] = 0;
          if (!sel) *reinterpret_cast<double2*>(ws_pair(ws, 2, gi)) = make_double2(0.0, 0.0);
          if (sel) {
            const double dx = pwx - cd.ref_pos[0];
            const double dy = pwy - cd.ref_pos[1];
            const double dz = pwz - cd.ref_pos[2];
            const double depth = sqrt(dx * dx + dy * dy + dz * dz);
            const Vec3 X = { fx_ * depth, fy_ * depth, fz_ * depth };
            *reinterpret_cast<double2*>(ws_pair(ws, 0, gi)) = make_double2(X.x, X.y);
            *reinterpret_cast<double2*>(ws_pair(ws, 1, gi)) = make_double2(X.z, pu);
            *reinterpret_cast<double2*>(ws_pair(ws, 2, gi)) = make_double2(pv, 1.0);
            ++my_sel;
          }
        }
      }
    }
    if (my_sel) atomicAdd(&g_state.nsel, my_sel);
  }
  __syncthreads();
  SVOH_STAMP_ADD(0);
  constexpr bool cluster = CLUSTER;
  // cluster mode: descriptor pbi is share (pbi % G) of problem (pbi / G); share 0 reports the common result in
  // the problem's slot, the other shares write theirs behind the problems' slots
  const int c_prob = cluster ? pbi / a.cluster : pbi;
  const int c_share = cluster ? pbi % a.cluster : 0;
  const int res_idx = cluster ? (c_share == 0 ? c_prob : a.n_problems / a.cluster + pbi) : pbi;
  unsigned cluster_epoch = 0;
  bool cluster_failed = false;
  int n_sel = g_state.nsel;
  if (cluster) {   // the number of selected features of the whole problem, not of this share
    if (tid == 0) s_x[0] = (double)n_sel;
    __syncthreads();
    cluster_failed = !cluster_sum<NT>(a, c_prob, c_share, cluster_epoch, s_x, 1, tid, &s_cluster_ok);
    n_sel = cluster_failed ? 0 : (int)s_x[0];
    __syncthreads();
  }
  if (n_sel == 0) {
    if (tid == 0) {
      svoh_align_result& r = a.results[res_idx];
      r.status = cluster_failed ? 3 : 1; r.n_fts_to_track = 0;
      store_rigid(g_state.T, r.T_icur_iref);
      r.alpha = g_state.alpha; r.beta = g_state.beta;
      r.n_patch_iters = 0;
      for (int l = 0; l < SVOH_MAX_LEVELS; ++l) { r.iters[l] = 0; r.n_meas[l] = 0; r.chi2[l] = 0.0; }
      if (eval_mode) for (int k = 0; k < 74; ++k) a.eval_out[74 * pbi + k] = 0.0;
    }
    SVOH_STAMP_FLUSH();
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): no image request of this problem lands in the next one's
    continue;
  }
  const bool est_alpha = opt.estimate_illumination_gain != 0;
  const bool est_beta = opt.estimate_illumination_offset != 0;
  constexpr bool robust = ROBUST;
  const bool dist_jac = opt.use_distortion_jacobian != 0;
  const float weight_scale = (float)opt.weight_scale;

  // cameras side by side (run_cameras): when the patches of all cameras fit the workgroup together, wave after wave
  constexpr int kLanesPerPatch = ROWS ? LPP : 1;
  bool side_by_side = false;
  if (RIG && n_cams >= 2) {
    int lanes_needed = 0;
    for (int c = 0; c < n_cams; ++c) lanes_needed += (cams[c].n_features * kLanesPerPatch + 63) & ~63;
    side_by_side = lanes_needed <= NT;
  }

  for (int level = level_hi; level >= level_lo; --level) {
    const double scale = 1.0f / (1 << level);
    // ---- the level's images: resident since the problem's start, staged now, or read from global memory ----
    int lvl_off = s_lvl_off[level];
    bool in_lds = lvl_off >= 0;
    if (!in_lds && level_bytes(level) <= a.lds_img_bytes) {
      // not resident (the coarser levels took the room) but it fits by itself: over the levels that are done
      __syncthreads();  // their readers are done with lds_img
      stage_level(level, 0);
      lvl_off = 0;
      in_lds = true;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this thread's image requests (issued here or at the problem's start) have landed
    __syncthreads();
#if SVOH_ALIGN_LDS_DESC
    if (tid >= 64 && tid < 64 + n_cams) {   // (a lane of the second wave per camera: the first wave's lane 0 is busy below)
      const DevCamDesc& cdg = cams[tid - 64];
      LevelDesc& ld = s_lvd[tid - 64];
      ld.ref = cdg.ref[level]; ld.cur = cdg.cur[level]; ld.cam = cdg.cam; ld.n_features = cdg.n_features; ld.feat_off = cdg.feat_off;
    }
#endif
    if (tid == 0) {
      g_state.level_done = 0;
      for (int c = 0; c < n_cams; ++c)
        g_state.Tcr[c] = mul(mul(g_state.cam_cur_T_cam_imu[c], g_state.T), g_state.cam_ref_T_imu_cam[c]);
      g_state.alpha_f = (float)g_state.alpha; g_state.beta_f = (float)g_state.beta;
      g_state.Told = g_state.T; g_state.alpha_old = g_state.alpha; g_state.beta_old = g_state.beta;  // old_state = state (hpp:45)
    }
    __syncthreads();
    SVOH_STAMP_ADD(1);

    for (int iter = 0; iter < opt.max_iter; ++iter) {
      const double one_plus_alpha = uniform_f64(1.0 + (double)g_state.alpha_f);
      const double beta_d = uniform_f64((double)g_state.beta_f);
      // Without robust weights the Hessian of a level only depends on which patches are visible (the Jacobians
      // are the reference patch's: inverse compositional), so after the level's first iteration a pass computes
      // the gradient and chi2 only and the solver reuses the level's factorisation.  A pass that finds a patch
      // whose visibility differs from the last full pass is thrown away and repeated in full: the numbers are
      // those of recomputing the Hessian every iteration, as the reference does.
      bool light = SVOH_ALIGN_REUSE_HESSIAN && !eval_mode && !robust && iter > 0;
      for (;;) {
        int nvis = 0, changed = 0;
        auto run_cameras = [&](auto gonly_tag, auto& acc_ref) {
          constexpr bool G = decltype(gonly_tag)::value;
          int off = lvl_off;
          // A rig of several cameras whose patches fit the workgroup together: the cameras run SIDE BY SIDE, camera c on
          // the waves behind camera c-1's (whole waves: the choice is wave-uniform), instead of one after the other on the
          // same few waves -- a stereo bundle of 2 x 160 patches is one round of 3 + 3 waves (spread over the four SIMDs
          // by the hardware's wave placement), not two rounds of 3.  The normal equations are the sum over the cameras
          // either way (sparse_img_align.cpp:138-154); a lane's partial sum now holds one camera's patches, so sums differ
          // in order only.  Not when the patches do not fit: the cameras then take turns with the whole workgroup.
          // (side_by_side: decided once per problem, above the level loop)
          int cam_base = 0;
          for (int c = 0; c < n_cams; ++c) {
#if SVOH_ALIGN_LDS_DESC
            const CamPass cd = uniform_cam_pass(s_lvd[c]);
            const DevImage rim = uniform_image(s_lvd[c].ref);
            const DevImage cim = uniform_image(s_lvd[c].cur);
#else
            const DevCamDesc& cdg = cams[c];
            CamPass cd;
            cd.cam = cdg.cam; cd.n_features = cdg.n_features; cd.feat_off = cdg.feat_off;
            const DevImage& rim = cdg.ref[level];
            const DevImage& cim = cdg.cur[level];
#endif
            const Rigid Tcr = uniform_rigid(g_state.Tcr[c]);
            unsigned cam_stride = NT, cam_t = (unsigned)tid;
            if (RIG && side_by_side) {   // (uniform)
              const int cam_width = (cd.n_features * kLanesPerPatch + 63) & ~63;
              cam_stride = (unsigned)cam_width;
              cam_t = (unsigned)(tid - cam_base);
              cam_base += cam_width;
            }
            if (in_lds) {
              ImgView<true> ref, cur;
              ref.p = (const __attribute__((address_space(3))) uint8_t*)(lds_img + off); ref.pitch = rim.w;
              off += ((rim.w * rim.h + 15) & ~15);
              cur.p = (const __attribute__((address_space(3))) uint8_t*)(lds_img + off); cur.pitch = cim.w;
              off += ((cim.w * cim.h + 15) & ~15);
              if constexpr (ROWS) {
                if (!RIG || cam_t < cam_stride)
                  accumulate_camera_rows<P, (ROWS ? LPP : 2), D, NT, true, G, ROBUST>(a, ws, cd, ref, cur, cim.w, cim.h, Tcr, scale, one_plus_alpha, beta_d,
                                                                            est_alpha, est_beta, robust, dist_jac, weight_scale, (int)cam_t, s_jc[c],
                                                                            acc_ref, nvis, changed, (int)cam_stride);
              } else if constexpr (STAGED) {
                if (!RIG || cam_t < cam_stride)
                  accumulate_camera_staged<P, D, NT, true, G, ROBUST>(a, cd, ref, cur, cim.w, cim.h, Tcr, scale, one_plus_alpha, beta_d,
                                                            est_alpha, est_beta, robust, dist_jac, weight_scale, (int)cam_t,
                                                            s_stage + wave * kStageDoubles, s_jc[c], acc_ref, nvis, changed, (int)cam_stride);
              }
              else if (!RIG || cam_t < cam_stride)   // (unsigned: the lanes of the cameras in front are "negative")
                accumulate_camera<P, D, NT, true, G, ROBUST>(a, ws, cd, ref, cur, cim.w, cim.h, Tcr, scale, one_plus_alpha, beta_d,
                                                     est_alpha, est_beta, robust, dist_jac, weight_scale, (int)cam_t, s_jc[c], acc_ref,
                                                     nvis, changed, (int)cam_stride);
            } else {
              ImgView<false> ref, cur;
              ref.p = rim.data; ref.pitch = rim.pitch;
              cur.p = cim.data; cur.pitch = cim.pitch;
              if constexpr (ROWS) {
                if (!RIG || cam_t < cam_stride)
                  accumulate_camera_rows<P, (ROWS ? LPP : 2), D, NT, false, G, ROBUST>(a, ws, cd, ref, cur, cim.w, cim.h, Tcr, scale, one_plus_alpha, beta_d,
                                                                             est_alpha, est_beta, robust, dist_jac, weight_scale, (int)cam_t, s_jc[c],
                                                                             acc_ref, nvis, changed, (int)cam_stride);
              } else if constexpr (STAGED) {
                if (!RIG || cam_t < cam_stride)
                  accumulate_camera_staged<P, D, NT, false, G, ROBUST>(a, cd, ref, cur, cim.w, cim.h, Tcr, scale, one_plus_alpha, beta_d,
                                                             est_alpha, est_beta, robust, dist_jac, weight_scale, (int)cam_t,
                                                             s_stage + wave * kStageDoubles, s_jc[c], acc_ref, nvis, changed, (int)cam_stride);
              }
              else if (!RIG || cam_t < cam_stride)   // (unsigned: the lanes of the cameras in front are "negative")
                accumulate_camera<P, D, NT, false, G, ROBUST>(a, ws, cd, ref, cur, cim.w, cim.h, Tcr, scale, one_plus_alpha, beta_d,
                                                      est_alpha, est_beta, robust, dist_jac, weight_scale, (int)cam_t, s_jc[c], acc_ref,
                                                      nvis, changed, (int)cam_stride);
            }
          }
        };
        if (light) {
          double accg[D + 1];
#pragma unroll
          for (int k = 0; k < D + 1; ++k) accg[k] = 0.0;
          SVOH_OUTER_STAMP_BEGIN();
          run_cameras(std::true_type(), accg);
          SVOH_OUTER_STAMP(5);   // includes the pieces 0..4 counted inside
#if SVOH_ALIGN_FOLD_VOTE
          // The vote -- did any patch's visibility differ from the last full pass? -- travels with the reduction: every wave
          // leaves its own verdict in LDS next to its partial sums, ONE barrier, and everybody reads the NW verdicts.  A pass
          // that is thrown away has reduced for nothing (that is rare: none on the benchmark scenes); every other pass has one
          // barrier less (1.1 K cycles of a 180-patch problem's iteration by the stamps).  nvis is only added once the pass counts.
          {
            int ridx;
            bool rvalid;
            wave_reduce_scatter<D + 1>(accg, lane, ridx, rvalid);
            if (rvalid) s_red[wave][ridx] = accg[0];
          }
          {
            const int wave_changed = __builtin_amdgcn_ballot_w64(changed != 0) != 0ull;
            if (lane == 0) s_changed[wave] = wave_changed;
          }
          nvis = wave_sum_i32_dpp(nvis);
          __syncthreads();
          int changed_here = 0;
#pragma unroll
          for (int w = 0; w < NW; ++w) changed_here |= s_changed[w];
          SVOH_OUTER_STAMP(6);
          if (changed_here && !cluster) {   // visibility moved: this iteration in full
            SVOH_STAMP_COUNT(5);
            light = false;
            __syncthreads();   // (everybody has read the verdicts and nobody reads this pass' partial sums: the full pass may write both)
            continue;
          }
          SVOH_STAMP_COUNT(6);
          SVOH_STAMP_ADD(2);
          if (lane == 0) atomicAdd(&g_nvis, nvis);
#else
          const int changed_here = __syncthreads_or(changed);
          SVOH_OUTER_STAMP(6);
          if (changed_here && !cluster) {   // visibility moved: this iteration in full
            SVOH_STAMP_COUNT(5);
            light = false;
            continue;
          }
          SVOH_STAMP_COUNT(6);
          SVOH_STAMP_ADD(2);
          {
            int ridx;
            bool rvalid;
            wave_reduce_scatter<D + 1>(accg, lane, ridx, rvalid);
            if (rvalid) s_red[wave][ridx] = accg[0];
          }
          nvis = wave_sum_i32_dpp(nvis);
          if (lane == 0) atomicAdd(&g_nvis, nvis);
          __syncthreads();
#endif
          if (tid < D + 1) {   // gradient and chi2 only: g_sum[0 .. NH) still holds the level's Hessian
            double v = 0.0;
            for (int w = 0; w < NW; ++w) v += s_red[w][tid];
            g_sum[AccLayout<D>::NH + tid] = v;
          }
          __syncthreads();
          if (cluster) {   // gradient, chi2, visible count and the visibility vote of all shares
            if (tid < D + 1) s_x[tid] = g_sum[AccLayout<D>::NH + tid];
            if (tid == 0) { s_x[D + 1] = (double)g_nvis; s_x[D + 2] = changed_here ? 1.0 : 0.0; }
            __syncthreads();
            if (!cluster_sum<NT>(a, c_prob, c_share, cluster_epoch, s_x, D + 3, tid, &s_cluster_ok)) { cluster_failed = true; break; }
            const bool changed_anywhere = s_x[D + 2] != 0.0;
            __syncthreads();
            if (changed_anywhere) {
              if (tid == 0) g_nvis = 0;
              __syncthreads();
              light = false;
              continue;
            }
            if (tid < D + 1) g_sum[AccLayout<D>::NH + tid] = s_x[tid];
            if (tid == 0) g_nvis = (int)s_x[D + 1];
            __syncthreads();
          }
        } else {
          double acc[NACC];
#pragma unroll
          for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
          run_cameras(std::false_type(), acc);
          SVOH_STAMP_ADD(2);
          // ---- reduce: butterfly reduce-scatter per wave, then across waves through LDS ----
          {
            int ridx;
            bool rvalid;
            wave_reduce_scatter<NACC>(acc, lane, ridx, rvalid);
            if (rvalid) s_red[wave][ridx] = acc[0];
          }
          nvis = wave_sum_i32_dpp(nvis);
          if (lane == 0) atomicAdd(&g_nvis, nvis);
          __syncthreads();
          if (tid < NACC) {
            double v = 0.0;
            for (int w = 0; w < NW; ++w) v += s_red[w][tid];
            g_sum[tid] = v;
          }
          __syncthreads();
          if (cluster) {
            if (tid < NACC) s_x[tid] = g_sum[tid];
            if (tid == 0) s_x[NACC] = (double)g_nvis;
            __syncthreads();
            if (!cluster_sum<NT>(a, c_prob, c_share, cluster_epoch, s_x, NACC + 1, tid, &s_cluster_ok)) { cluster_failed = true; break; }
            if (tid < NACC) g_sum[tid] = s_x[tid];
            if (tid == 0) g_nvis = (int)s_x[NACC];
            __syncthreads();
          }
        }
        break;
      }
      if (cluster_failed) break;

      SVOH_STAMP_ADD(3);
      // ---- serial part: prior, pivoted LDL^T (or the level's factor again), SE3 update, convergence ----
#if SVOH_ALIGN_WAVE_STEP
      if (eval_mode) { if (tid == 0) gn_serial_step<P, D, ILLUM>(a, pb, cams, n_cams, res_idx, level, iter, eval_mode, light); }
      else if (tid < 64) {
        // the step is its workgroup's critical path (three waves wait at the barrier behind it): it goes ahead of the
        // other workgroup's pass waves on its SIMD
#if SVOH_STEP_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        gn_wave_step<P, D, ILLUM>(a, n_cams, level, iter, light, tid);
#if SVOH_STEP_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
      }
#else
      if (tid == 0) gn_serial_step<P, D, ILLUM>(a, pb, cams, n_cams, res_idx, level, iter, eval_mode, light);
#endif
      __syncthreads();
      SVOH_STAMP_ADD(4);
      if (g_state.level_done) break;
    }
    if (cluster_failed) break;
  }

  if (tid == 0) {
    svoh_align_result& r = a.results[res_idx];
    r.status = cluster_failed ? 3 : g_state.status;   // 3: a workgroup of the cluster never arrived (see cluster_sum)
    r.n_fts_to_track = n_sel;
    store_rigid(g_state.T, r.T_icur_iref);
    r.alpha = g_state.alpha; r.beta = g_state.beta;
    r.n_patch_iters = g_state.patch_iters;
    for (int l = 0; l < SVOH_MAX_LEVELS; ++l) { r.iters[l] = g_state.lvl_iters[l]; r.n_meas[l] = g_state.lvl_n_meas[l]; r.chi2[l] = g_state.lvl_chi2[l]; }
  }
  SVOH_STAMP_FLUSH();
  }  // next problem
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

// Shares of one problem evaluated by several workgroups: add their 74-double blocks in share order.
__global__ __launch_bounds__(128)
void sum_shares_kernel(const double* parts, int n_shares, double* out)
{
  const int k = threadIdx.x;
  if (k >= 74) return;
  double v = 0.0;
  for (int s = 0; s < n_shares; ++s) v += parts[74 * s + k];
  out[k] = v;
}

// One Gauss-Newton iteration on normal equations that were summed outside the resident kernel (patch-split
// mode): unpack the 74 doubles into the packed layout, run the very same serial step, write the state back.
template <int P, bool ILLUM>
__global__ __launch_bounds__(64)
void align_gn_update_kernel(const AlignKernelArgs a, const double* sums, svoh_align_gn_state* st, int level, int iter)
{
  constexpr int D = ILLUM ? 8 : 6;
  constexpr int NH = AccLayout<D>::NH;
  ShState& s = g_state;
  double (&s_sum)[45] = g_sum;
  if (threadIdx.x != 0) return;
  const DevProblemDesc& pb = a.problems[0];
  const DevCamDesc* cams = a.cams + pb.cam_begin;
  s.T = load_rigid(st->T_icur_iref);
  s.alpha = st->alpha; s.beta = st->beta;
  if (iter == 0) {   // old_state = state at the start of a level (hpp:45)
    s.Told = s.T; s.alpha_old = s.alpha; s.beta_old = s.beta;
    for (int k = 0; k < 8; ++k) s.I_prior[k] = 0.0;
  } else {
    s.Told = load_rigid(st->T_old); s.alpha_old = st->alpha_old; s.beta_old = st->beta_old;
    for (int k = 0; k < 8; ++k) s.I_prior[k] = st->I_prior[k];
  }
  s.stop = st->stop; s.status = st->status; s.level_done = 0; s.nsel = 0; s.patch_iters = 0;
  {
    int idx = 0;
    for (int r = 0; r < D; ++r)
      for (int c2 = r; c2 < D; ++c2) s_sum[idx++] = sums[c2 * 8 + r];
    for (int r = 0; r < D; ++r) s_sum[NH + r] = sums[64 + r];
    s_sum[NH + D] = sums[72];
  }
  const int n_meas = (int)sums[73];
  g_nvis = n_meas / (P * P);
  s.prior = pb.prior;
  for (int c = 0; c < pb.n_cams; ++c) {
    s.cam_cur_T_cam_imu[c] = load_rigid(cams[c].cur_T_cam_imu);
    s.cam_ref_T_imu_cam[c] = load_rigid(cams[c].ref_T_imu_cam);
  }
  gn_serial_step<P, D, ILLUM>(a, pb, cams, pb.n_cams, 0, level, iter, false, false);
  if (level < SVOH_MAX_LEVELS) {
    svoh_align_result& res = a.results[0];
    res.iters[level] = s.lvl_iters[level]; res.n_meas[level] = s.lvl_n_meas[level]; res.chi2[level] = s.lvl_chi2[level];
  }
  store_rigid(s.T, st->T_icur_iref);
  st->alpha = s.alpha; st->beta = s.beta;
  store_rigid(s.Told, st->T_old);
  st->alpha_old = s.alpha_old; st->beta_old = s.beta_old;
  for (int k = 0; k < 8; ++k) st->I_prior[k] = s.I_prior[k];
  st->chi2 = s_sum[NH + D] / (double)n_meas;
  st->n_meas = n_meas;
  st->stop = s.stop; st->level_done = s.level_done; st->status = s.status;
}

struct LaunchCfg { int nt; size_t lds; };

constexpr size_t kLdsPerCu = 163840;   // gfx950: 160 KB per compute unit
template <int P, int NT, bool ILLUM, bool CLUSTER, bool ROBUST, int LPP = 1, bool LAT = false, bool RIG = false>
static hipError_t launch_one(hipStream_t st, int grid, size_t lds, AlignKernelArgs args)
{
  auto kern = sparse_align_kernel<P, NT, ILLUM, CLUSTER, ROBUST, LPP, LAT, RIG>;
  if (NT == 256 && args.lds_two_per_cu) {
    // Two workgroups per compute unit is what the batch geometry is built on, and the budget is tight (28-29.5 KB of
    // static LDS + 51 KB of images = 79-80.5 of the 80 KB a workgroup may have).  The static part is asked of the code
    // object, not assumed: if it ever grows, the image area shrinks (a level fewer in LDS) instead of the second
    // workgroup silently not fitting -- half the throughput with every test green.
    static std::atomic<size_t> static_lds_cache{ 0 };   // per instantiation; the same for every context and host thread (one code object)
    size_t static_lds = static_lds_cache.load(std::memory_order_relaxed);
    if (static_lds == 0) {
      hipFuncAttributes attr;
      hipError_t ea = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kern));
      if (ea != hipSuccess) return ea;
      static_lds = attr.sharedSizeBytes ? attr.sharedSizeBytes : 1;
      static_lds_cache.store(static_lds, std::memory_order_relaxed);
    }
    const size_t room = kLdsPerCu / 2 > static_lds ? (kLdsPerCu / 2 - static_lds) & ~(size_t)15 : 0;
    if (lds > room) {
      args.lds_img_bytes -= (int32_t)(lds - room);
      if (args.lds_img_bytes < 0) args.lds_img_bytes = 0;
      lds = room;
    }
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, args);
  return hipGetLastError();
}

// lpp: lanes per patch (1: a lane owns a patch; 2, 4, 8 <= P: the rows geometry, 512 threads)
template <int P, bool ILLUM, bool ROBUST>
static hipError_t launch_nt(hipStream_t st, int nt, int lpp, int grid, size_t lds, const AlignKernelArgs& args)
{
  if (args.cluster > 1) return launch_one<P, 256, ILLUM, true, ROBUST>(st, grid, lds, args);
  if (nt == 256) {
    if (args.latency_build) return args.rig_build ? launch_one<P, 256, ILLUM, false, ROBUST, 1, true, true>(st, grid, lds, args)
                                                  : launch_one<P, 256, ILLUM, false, ROBUST, 1, true>(st, grid, lds, args);
    return launch_one<P, 256, ILLUM, false, ROBUST>(st, grid, lds, args);
  }
  if (lpp <= 1 && args.rig_build) return launch_one<P, 512, ILLUM, false, ROBUST, 1, false, true>(st, grid, lds, args);
  switch (lpp) {
    case 2: return launch_one<P, 512, ILLUM, false, ROBUST, 2>(st, grid, lds, args);
    case 4: return launch_one<P, 512, ILLUM, false, ROBUST, 4>(st, grid, lds, args);
    case 8: if constexpr (P == 8) return launch_one<P, 512, ILLUM, false, ROBUST, 8>(st, grid, lds, args);
    default: return launch_one<P, 512, ILLUM, false, ROBUST>(st, grid, lds, args);
  }
}

// cluster mode thresholds (measured): an exchange costs ~2 us, so below ~500 patches the barriers
// cost more than the idle CUs; ~200 patches per workgroup is where the per-iteration time stops falling
constexpr int kWsLdsMaxFeatures = 340;    // features of a problem whose workspace rows are kept in LDS (16 KB)
constexpr int kClusterMinFeatures = 512;
constexpr int kClusterMaxWorkgroups = 32;
constexpr int kClusterMaxProblems = 64;   // arrival counters: one 32-bit word per problem in a 256-byte block

static int validate_options(svoh_ctx* ctx, const svoh_align_options* o)
{
  SVOH_REQUIRE(ctx, o != nullptr, "options is NULL");
  if (o->patch_size != 4 && o->patch_size != 8)
    return set_error(ctx, SVOH_ERR_UNSUPPORTED, "patch_size %d not built (4 and 8 are)", o->patch_size);
  SVOH_REQUIRE(ctx, o->max_level >= o->min_level && o->min_level >= 0 && o->max_level < SVOH_MAX_LEVELS,
               "bad level range");
  SVOH_REQUIRE(ctx, o->max_iter >= 1, "max_iter must be >= 1");
  return SVOH_OK;
}

// Build descriptors, upload host feature arrays if needed, launch.
// patch-split mode: either evaluate at a device-resident state into the caller's buffer, or run one
// Gauss-Newton update on summed normal equations
struct SplitArgs {
  const svoh_align_gn_state* ext_state = nullptr;   // evaluate: state to evaluate at
  double* sums_out = nullptr;                       // evaluate: 74 doubles, device
  const double* sums_in = nullptr;                  // update: summed normal equations, device
  svoh_align_gn_state* state = nullptr;             // update: state to advance, device
  int iter = 0;
  bool update = false;
  int n_shares = 1;                                 // evaluate: workgroups the problem's features are spread over
};

// The launch geometry of the resident kernel: workgroups per problem (cluster mode), threads per workgroup, lanes per
// patch, and which build of the 256 / 512-thread kernel.  The geometries differ in the ORDER of their sums only (poses agree
// to 1e-15), but a caller that wants a problem's result to be the same bits whatever else shares its launch -- the
// lock-step front end of many camera streams, whose streams must reproduce their single-stream runs byte for byte --
// asks for the geometry a launch of the problem alone would get (svoh_sparse_align_geometry_key / _enqueue_keyed).
struct AlignGeometry {
  int cluster_g = 0;       // 0: one workgroup per problem; >= 2: that many co-resident workgroups per problem
  int nt = 256;            // 256 / 512 threads
  int rows = 1;            // lanes per patch (1, 2, 4, 8)
  bool latency = false;    // the one-wave-per-SIMD build of the 256-thread kernel
  bool rig = false;        // the build that runs a rig's cameras side by side
  int32_t key() const { return (1 << 30) | (cluster_g & 0xff) | ((nt == 512 ? 1 : 0) << 8) | ((rows & 0xf) << 9) | ((latency ? 1 : 0) << 13) | ((rig ? 1 : 0) << 14); }
  static bool from_key(int32_t k, AlignGeometry* g)
  {
    if (!(k & (1 << 30)) || (k & ~((1 << 30) | 0x7fff))) return false;
    g->cluster_g = k & 0xff; g->nt = (k >> 8) & 1 ? 512 : 256; g->rows = (k >> 9) & 0xf; g->latency = (k >> 13) & 1; g->rig = (k >> 14) & 1;
    return (g->rows == 1 || g->rows == 2 || g->rows == 4 || g->rows == 8) && g->cluster_g != 1 && g->cluster_g <= kClusterMaxWorkgroups;
  }
};

// cluster mode: a single problem with many features gets several co-resident workgroups of the resident kernel
// (one share each) that add their normal equations through a device-side barrier every iteration, instead of
// one workgroup on one CU.  SVOH_ALIGN_CLUSTER=0 turns it off, =G forces G workgroups.
// A handful of such problems (a stereo pair of streams, a few cameras) are clustered alike, each with its own
// exchange slots, as long as every one of them is large and they all fit on the device at once.
// Returns the workgroups per problem (>= 2) or 0.
static int decide_cluster(const svoh_ctx* ctx, int n_problems, const svoh_align_problem* problems)
{
  if (n_problems > kClusterMaxProblems || ctx->align_no_cluster) return 0;
  int64_t nf_min = INT64_MAX, nf_max = 0;
  for (int p = 0; p < n_problems; ++p) {
    int64_t nf = 0;
    if (problems[p].n_cams >= 1 && problems[p].n_cams <= SVOH_MAX_CAMS)
      for (int c = 0; c < problems[p].n_cams; ++c) nf += problems[p].cams[c].n_features > 0 ? problems[p].cams[c].n_features : 0;
    nf_min = nf < nf_min ? nf : nf_min;
    nf_max = nf > nf_max ? nf : nf_max;
  }
  int g = SvohKnobs::or_default(ctx->knobs.align_cluster, -1);
  // measured (scripts/perf_small_batch.py): up to 16 problems always gain; 32..64 only when each is large
  const bool worth = nf_min >= kClusterMinFeatures && (n_problems <= 16 || nf_min >= 3000);
  // workgroups per problem against its size, measured (profiles/r05_align_cluster_sweep.txt: one problem of 700 ... 20 000 patches,
  // 2 ... 32 workgroups): about 250 patches per workgroup up to eight workgroups, eight up to 5 000 patches, then 12 / 16 / 20 --
  // more workgroups than that pay more per exchange than their shorter passes save (the rule of rounds 2 - 4, one workgroup per 192
  // patches up to 32, was 4 - 18 % slower between 1 000 and 20 000 patches)
  auto workgroups_for = [](int64_t nf) {
    if (nf < 875) return 3;
    if (nf < 1250) return 4;
    if (nf < 1750) return 6;
    if (nf < 5000) return 8;
    if (nf < 7000) return 12;
    if (nf < 16000) return 16;
    return 20;
  };
  if (g < 0) g = worth ? workgroups_for(nf_max) : 0;
  if (g > kClusterMaxWorkgroups) g = kClusterMaxWorkgroups;
  if ((int64_t)g * n_problems > ctx->num_cus) g = ctx->num_cus / n_problems;   // every workgroup on its own CU
  return g >= 2 ? g : 0;
}

// what pass 1 of enqueue_align learns about the problems of a launch, as far as the geometry depends on it
struct LaunchShape { int max_feat_per_problem = 0; bool have_rig = false, rig_wants_512 = false; };
static void add_to_shape(const svoh_align_problem& pb, int S, LaunchShape* z)
{
  int nf = 0;
  for (int c = 0; c < pb.n_cams; ++c) nf += pb.cams[c].n_features > 0 ? pb.cams[c].n_features : 0;
  const int per_share = S > 1 ? nf / S + pb.n_cams : nf;
  if (per_share > z->max_feat_per_problem) z->max_feat_per_problem = per_share;
  if (S == 1 && pb.n_cams >= 2) {
    z->have_rig = true;
    int lanes = 0;
    for (int c = 0; c < pb.n_cams; ++c) lanes += (pb.cams[c].n_features + 63) & ~63;
    z->rig_wants_512 = z->rig_wants_512 || (lanes > 256 && lanes <= 512);
  }
}

// Geometry.  Many problems: 256-thread workgroups, two per CU (256 VGPRs each), so
// that one problem's serial solve overlaps the other's patch work; LDS holds levels
// >= 2 of a 640x480 pyramid.  Few problems (latency mode): 512-thread workgroups.
// measured on MI355X (2000 patches): one problem takes 0.42 ms with 512 threads, 0.50 ms
// with 256 and 0.91 ms with 1024 (128-VGPR budget spills), so 512 is the latency geometry
// (measured, scripts/perf_mid_batch.py: from one problem per CU on, two 256-thread workgroups per CU with the
// LDS-DMA workspace path beat one 512-thread workgroup: 384 problems 1.06 -> 0.82 ms)
static AlignGeometry decide_geometry(const svoh_ctx* ctx, const svoh_align_options* opt, int n_desc, int cluster_g, const LaunchShape& z)
{
  AlignGeometry g;
  const bool cluster = cluster_g >= 2;
  g.cluster_g = cluster ? cluster_g : 0;
  int nt = (n_desc >= ctx->num_cus) ? 256 : 512;
  if (z.max_feat_per_problem <= 256) nt = 256;
  // a few rigs whose cameras fill five to eight waves between them: 512 threads, so that the cameras run side by side with
  // one round each (run_cameras in the kernel) instead of taking turns
  if (!cluster && n_desc < ctx->num_cus && z.rig_wants_512) nt = 512;
  // Rows geometry (LPP lanes per patch, accumulate_camera_rows): a problem with so few patches that they do not give
  // every SIMD of its compute unit a wave gets 2, 4 or 8 lanes per patch, as many as keep it at one wave per SIMD (256
  // lanes) -- a lane's pass is then a chain of P / LPP rolling rows instead of P.  Measured (one problem, levels 4..2,
  // kernel ms, lanes per patch 1 / 2 / 4 / 8): 60 patches of 8x8 0.151 / 0.131 / 0.119 / 0.126; 100 of 4x4 0.092 / 0.087 /
  // 0.092; 180 of 4x4 0.092 / 0.095 / 0.098; 180 of 8x8 0.137 / 0.143 / 0.137 / 0.169; 2000 of 4x4 0.184 / 0.230 / 0.330:
  // beyond one wave per SIMD the compute unit is bound by vector issue, and more lanes per patch are more instructions
  // per patch (every lane repeats the projection, the Jacobian rows and two interpolated rows).
  // SVOH_ALIGN_ROWS: lanes per patch (2, 4 or 8); anything else = a lane per patch.
  int rows = 1;
  if (!cluster && n_desc < ctx->num_cus) {
    while (rows * 2 <= opt->patch_size && rows * 2 <= 8 && (int64_t)z.max_feat_per_problem * rows * 2 <= 256) rows *= 2;
    rows = SVOH_ALIGN_ROWS_DEFAULT(rows);
  }
  rows = SvohKnobs::or_default(ctx->knobs.align_rows, rows);
  if (rows != 2 && rows != 4 && rows != 8) rows = 1;
  if (rows > opt->patch_size || cluster) rows = 1;
  if (rows > 1) nt = 512;
  nt = SvohKnobs::or_default(ctx->knobs.align_threads, nt);
  if (cluster) nt = 256;   // one workgroup per CU at most: all of them are resident together
  if (nt != 256 && nt != 512) nt = 256;
  if (nt == 256) rows = 1;
  g.nt = nt; g.rows = rows;
  // SVOH_ALIGN_LATENCY_BUILD=0 keeps the batch build for small launches too (A/B)
  g.latency = nt == 256 && !cluster && n_desc < ctx->num_cus && SvohKnobs::or_default(ctx->knobs.align_latency_build, 1) != 0;
  g.rig = !cluster && n_desc < ctx->num_cus && z.have_rig;
  return g;
}

// svoh_set_align_geometry_classes(ctx, 1): ONE geometry for every problem below the cluster threshold -- 512 threads, a lane per patch, a rig's
// cameras side by side where they fit -- instead of the five that are each fastest for one problem of their size (rows 2 / 4 below 129 / 65
// patches, the one-wave-per-SIMD 256-thread build up to 256, 512 threads above).  Measured cost for a problem alone (decide_geometry's table:
// 100 patches 0.087 -> 0.092 ms, 180 patches 0.092 -> 0.095): what a lock-step round of streams of different sizes saves is a launch per class.
static bool shared_class_geometry(const svoh_ctx* ctx, const svoh_align_problem& pb, AlignGeometry* g)
{
  if (!ctx->align_shared_classes || ctx->align_no_cluster) return false;
  if (pb.n_cams < 1 || pb.n_cams > SVOH_MAX_CAMS) return false;
  int64_t nf = 0;
  for (int c = 0; c < pb.n_cams; ++c) nf += pb.cams[c].n_features > 0 ? pb.cams[c].n_features : 0;
  if (nf >= kClusterMinFeatures) return false;   // (the large ones keep the rule of their size: decide_cluster)
  *g = AlignGeometry();
  g->cluster_g = 0; g->nt = 512; g->rows = 1; g->latency = false; g->rig = pb.n_cams >= 2;
  return true;
}

// the geometry a launch of this problem ALONE gets
static AlignGeometry geometry_of_single(const svoh_ctx* ctx, const svoh_align_options* opt, const svoh_align_problem& pb)
{
  AlignGeometry shared;
  if (shared_class_geometry(ctx, pb, &shared)) return shared;
  const int g = decide_cluster(ctx, 1, &pb);
  const int S = g >= 2 ? g : 1;
  LaunchShape z;
  add_to_shape(pb, S, &z);
  return decide_geometry(ctx, opt, S, g, z);
}

static int enqueue_align(svoh_ctx* ctx, const svoh_align_options* opt, int n_problems,
                         const svoh_align_problem* problems, int eval_level, const SplitArgs* split = nullptr,
                         const AlignGeometry* forced = nullptr)
{
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  int rc = validate_options(ctx, opt);
  if (rc != SVOH_OK) return rc;
  SVOH_REQUIRE(ctx, n_problems >= 1 && problems, "no problems");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  // a launch of ONE problem under shared classes: the geometry its key names (svoh_set_align_geometry_classes)
  AlignGeometry shared_single;
  if (!forced && !split && eval_level < 0 && n_problems == 1 && shared_class_geometry(ctx, problems[0], &shared_single)) forced = &shared_single;

  // patch-split evaluation: the one problem becomes S descriptors, share s holding features
  // [n*s/S, n*(s+1)/S) of every camera, one workgroup each
  int S = (split && !split->update && split->n_shares > 1) ? split->n_shares : 1;
  bool cluster = false;
  if (!split && eval_level < 0) {
    const int g = forced ? forced->cluster_g : decide_cluster(ctx, n_problems, problems);
    if (g >= 2) { S = g; cluster = true; }
  }
  SVOH_REQUIRE(ctx, !forced || (!split && eval_level < 0), "a forced geometry applies to full runs only");
  SVOH_REQUIRE(ctx, !cluster || (int64_t)S * n_problems <= ctx->num_cus, "cluster mode: more workgroups than compute units");
  SVOH_REQUIRE(ctx, !cluster || n_problems <= kClusterMaxProblems, "cluster mode: more problems in one launch than arrival counters");
  SVOH_REQUIRE(ctx, S == 1 || n_problems == 1 || cluster, "shares apply to a single problem");
  const int n_desc = n_problems * S;

  // pass 1: sizes
  size_t n_cams_total = 0, n_feat_total = 0, host_bytes = 0;
  int n_pos_jobs = 0;   // cameras whose seed positions come from the seed batch in flight (svoh_align_camera::pos_seed_unit)
  LaunchShape shape;   // largest problem (share); some problem has more than one camera; ... whose patches fill five to eight waves, camera by camera
  for (int p = 0; p < n_problems; ++p) {
    const svoh_align_problem& pb = problems[p];
    SVOH_REQUIRE(ctx, pb.n_cams >= 1 && pb.n_cams <= SVOH_MAX_CAMS, "n_cams out of range");
    int nf = 0;
    for (int c = 0; c < pb.n_cams; ++c) {
      const svoh_align_camera& cam = pb.cams[c];
      SVOH_REQUIRE(ctx, cam.n_features >= 0, "negative n_features");
      SVOH_REQUIRE(ctx, cam.n_features == 0 || (cam.px && cam.f && cam.pos_world && cam.flags),
                   "NULL feature array");
      SVOH_REQUIRE(ctx, cam.cam.distortion == SVOH_DISTORTION_NONE || cam.cam.distortion == SVOH_DISTORTION_RADTAN,
                   "unsupported distortion model");
      nf += cam.n_features;
      if (cam.mem_space == SVOH_MEM_HOST && !(split && split->update))
        host_bytes += ((size_t)cam.n_features * (8 * 8 + 1) + 63) & ~(size_t)63;
      if (cam.pos_seed_unit) {
        SVOH_REQUIRE(ctx, cam.mem_space == SVOH_MEM_HOST && !split && eval_level < 0, "pos_seed_unit: host-memory cameras of full runs only");
        host_bytes += (((size_t)cam.n_features * 4 + 63) & ~(size_t)63) + 64;   // the units, and the camera's entry of the job table
        ++n_pos_jobs;
      }
    }
    n_cams_total += (size_t)pb.n_cams * S;
    n_feat_total += nf;
    add_to_shape(pb, S, &shape);
  }
  SVOH_REQUIRE(ctx, n_pos_jobs == 0 || ctx->seed_block.valid,
               "pos_seed_unit: no staged seed batch has been sent off on this context (or its block has been staged again)");
  const int max_feat_per_problem = shape.max_feat_per_problem;
  const size_t feat_slots = n_feat_total ? n_feat_total : 1;

  // descriptors, then (256-byte aligned) the zeroed work-queue head and the cluster arrival counters, then the
  // feature arrays of host-resident cameras: one staging block, one upload
  const size_t desc_only = sizeof(DevProblemDesc) * n_desc + sizeof(DevCamDesc) * n_cams_total;
  const size_t ctl_off = (desc_only + 255) & ~(size_t)255;
  const size_t up_base = ctl_off + 512;
  const size_t desc_bytes = up_base + host_bytes;
  // The pinned staging buffer is reused by every call.  A call queued right behind another (enqueue without fetch,
  // the patch-split entries) must not write a byte of it -- not even the zeroed control block, whose offset moves
  // with the descriptor count and would land inside the earlier call's descriptors -- nor let reserve() replace
  // it, before the earlier call's upload has read it.
  // (two blocks in turn, an event only between launches queued back to back: svoh_internal.h)
  // (the slot is committed only once the upload has been queued: a call that fails in between -- an unknown frame handle,
  // a failed reserve -- must leave the NEXT call on the block this one was about to use, not on the block the last
  // successful launch may still be uploading from)
  const unsigned desc_slot = ctx->align_desc_slot ^ 1u;
  PinnedBuffer& h_desc = desc_slot ? ctx->h_desc_odd : ctx->h_desc;
  if (ctx->align_launches_since_drain >= 2) {
    // this block was last read by the upload of the launch before the last one, which has not been waited for
    if (ctx->align_staged_event_valid) SVOH_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_align_staged));
    else SVOH_ALIGN_DRAIN(ctx);
  }
  SVOH_HIP_TRY(ctx, h_desc.reserve(desc_bytes));
  SVOH_HIP_TRY(ctx, ctx->d_desc.reserve(desc_bytes));
  memset(static_cast<uint8_t*>(h_desc.ptr) + ctl_off, 0, 512);
  // results of launches queued since the last fetch are kept one after the other in pinned host memory -- and on the
  // device as well (a candidate projection queued behind several launches reads the result of any of them):
  // this launch's device block of n_desc + n_problems results starts behind the blocks of the launches before it
  const bool delivers = (S == 1 || cluster) && eval_level < 0;
  const size_t dev_results_off = delivers ? ctx->align_pending_dev : 0;
  {
    const size_t need = sizeof(svoh_align_result) * (dev_results_off + (size_t)n_desc + n_problems);
    if (need > ctx->d_results.cap) {
      if (dev_results_off) {   // earlier launches' results live in the old block: let them finish, take them along
        SVOH_ALIGN_DRAIN(ctx);
        DevBuffer bigger;
        SVOH_HIP_TRY(ctx, bigger.reserve(need * 2));
        SVOH_HIP_TRY(ctx, hipMemcpy(bigger.ptr, ctx->d_results.ptr, sizeof(svoh_align_result) * dev_results_off, hipMemcpyDeviceToDevice));
        std::swap(bigger.ptr, ctx->d_results.ptr);
        std::swap(bigger.cap, ctx->d_results.cap);
      } else {
        SVOH_HIP_TRY(ctx, ctx->d_results.reserve(need));
      }
    }
  }
  if (delivers) {
    SVOH_REQUIRE(ctx, ctx->align_pending_results + (size_t)n_problems <= svoh_ctx::kMaxQueuedResults,
                 "too many alignment results queued without a fetch");
    // room for a queue of launches of this size (svoh_ctx::kAlignEventRing of them), so that a caller that queues
    // steps back to back never waits here for the stream to drain and the block to be replaced
    size_t want_results = ctx->align_pending_results + (size_t)n_problems;
    const size_t queue_of_these = (size_t)svoh_ctx::kAlignEventRing * (size_t)n_problems;
    if (want_results < queue_of_these && queue_of_these <= svoh_ctx::kMaxQueuedResults) want_results = queue_of_these;
    const size_t need = sizeof(svoh_align_result) * want_results;
    if (need > ctx->h_results.cap) {
      if (ctx->align_pending_results) {   // earlier launches still deliver into the old block: let them finish, keep theirs
        SVOH_ALIGN_DRAIN(ctx);
        PinnedBuffer bigger;
        SVOH_HIP_TRY(ctx, bigger.reserve(need * 2));
        memcpy(bigger.ptr, ctx->h_results.ptr, sizeof(svoh_align_result) * ctx->align_pending_results);
        std::swap(bigger.ptr, ctx->h_results.ptr);
        std::swap(bigger.cap, ctx->h_results.cap);
      } else {
        SVOH_HIP_TRY(ctx, ctx->h_results.reserve(need));
      }
    }
  }
  SVOH_HIP_TRY(ctx, ctx->d_feat.reserve(feat_slots * (kWsPairs * 16 + 2) + 256));
  SVOH_HIP_TRY(ctx, ctx->d_eval.reserve(74 * sizeof(double) * S));

  DevProblemDesc* hp = static_cast<DevProblemDesc*>(h_desc.ptr);
  DevCamDesc* hc = reinterpret_cast<DevCamDesc*>(hp + n_desc);
  uint8_t* hup = static_cast<uint8_t*>(h_desc.ptr) + up_base;
  uint8_t* dup = static_cast<uint8_t*>(ctx->d_desc.ptr) + up_base;
  size_t up_off = 0;
  int cam_idx = 0, feat_off = 0;
  const int need_levels = opt->max_level + 1;
  const uint8_t* share_base[SVOH_MAX_CAMS] = {};   // uploaded block of camera c (host arrays, S > 1)
  std::vector<PosFromSeedsJob> pos_jobs;
  int pos_jobs_max_n = 0;
  for (int pd = 0; pd < n_desc; ++pd) {
    const int p = pd / S, sh = pd % S;
    const svoh_align_problem& pb = problems[p];
    DevProblemDesc& d = hp[pd];
    d.n_cams = pb.n_cams;
    d.cam_begin = cam_idx;
    d.T_init = pb.T_icur_iref;
    d.alpha_init = pb.alpha_init; d.beta_init = pb.beta_init;
    d.prior = pb.prior;
    for (int c = 0; c < pb.n_cams; ++c, ++cam_idx) {
      const svoh_align_camera& cam = pb.cams[c];
      DevCamDesc& dc = hc[cam_idx];
      const Frame* fr = find_frame(ctx, cam.ref_frame);
      const Frame* fc = find_frame(ctx, cam.cur_frame);
      if (!fr || !fc) return set_error(ctx, SVOH_ERR_BAD_HANDLE, "problem %d camera %d: unknown frame handle", p, c);
      if (fr->n_levels < need_levels || fc->n_levels < need_levels)
        return set_error(ctx, SVOH_ERR_INVALID_ARGUMENT, "problem %d camera %d: pyramid has fewer than %d levels", p,
                         c, need_levels);
      for (int l = 0; l < SVOH_MAX_LEVELS; ++l) {
        dc.ref[l] = l < fr->n_levels ? fr->lv[l] : DevImage{ nullptr, 0, 0, 0, 0 };
        dc.cur[l] = l < fc->n_levels ? fc->lv[l] : DevImage{ nullptr, 0, 0, 0, 0 };
      }
      dc.cam = cam.cam;
      {
        // The Jacobian rows take p_in_cam = T_cam_imu * (T_imu_cam * xyz_ref) to be xyz_ref itself (jac_rows): the two
        // extrinsic transformations of the reference frame must be inverses of each other, as they are in a
        // reference Frame (frame.h:342-357 builds both from one T_cam_imu).  A pair that is not would silently get
        // a different Jacobian than the reference computes: refused instead.
        const Rigid I = mul(load_rigid(cam.ref_T_cam_imu), load_rigid(cam.ref_T_imu_cam));
        const double dev = fmax(fmax(fabs(fabs(I.q.w) - 1.0), fmax(fabs(I.q.x), fmax(fabs(I.q.y), fabs(I.q.z)))),
                                fmax(fabs(I.t.x), fmax(fabs(I.t.y), fabs(I.t.z))));
        if (!(dev <= 1e-9))
          return set_error(ctx, SVOH_ERR_INVALID_ARGUMENT,
                           "problem %d camera %d: ref_T_cam_imu is not the inverse of ref_T_imu_cam (deviation %.3g)", p, c, dev);
      }
      dc.ref_T_imu_cam = cam.ref_T_imu_cam;
      dc.ref_T_cam_imu = cam.ref_T_cam_imu;
      dc.cur_T_cam_imu = cam.cur_T_cam_imu;
      for (int k = 0; k < 3; ++k) dc.ref_pos[k] = cam.ref_pos[k];
      const size_t lo = (size_t)((int64_t)cam.n_features * sh / S), hi = (size_t)((int64_t)cam.n_features * (sh + 1) / S);
      dc.n_features = (int32_t)(hi - lo);
      dc.feat_off = feat_off;
      feat_off += dc.n_features;
      if (cam.mem_space == SVOH_MEM_DEVICE || cam.n_features == 0 || (split && split->update)) {
        dc.px = cam.px + 2 * lo; dc.f = cam.f + 3 * lo; dc.pos_world = cam.pos_world + 3 * lo; dc.flags = cam.flags + lo;
      } else {
        const size_t n = (size_t)cam.n_features;
        if (sh == 0) {   // the camera's arrays go up once; every share points into the same block
          uint8_t* h = hup + up_off;
          memcpy(h, cam.px, n * 16);
          memcpy(h + n * 16, cam.f, n * 24);
          memcpy(h + n * 40, cam.pos_world, n * 24);
          memcpy(h + n * 64, cam.flags, n);
          share_base[c] = dup + up_off;
          up_off += (n * 65 + 63) & ~(size_t)63;
        }
        if (sh == 0 && cam.pos_seed_unit) {   // its units go up behind the camera's arrays; the job points at the uploaded positions
          memcpy(hup + up_off, cam.pos_seed_unit, n * 4);
          PosFromSeedsJob jb;
          jb.pos = reinterpret_cast<double*>(const_cast<uint8_t*>(share_base[c]) + n * 40);
          jb.unit = reinterpret_cast<const int32_t*>(dup + up_off);
          jb.n = (int32_t)n; jb.pad_ = 0;
          pos_jobs.push_back(jb);
          if ((int)n > pos_jobs_max_n) pos_jobs_max_n = (int)n;
          up_off += (n * 4 + 63) & ~(size_t)63;
        }
        const uint8_t* dv = share_base[c];
        dc.px = reinterpret_cast<const double*>(dv) + 2 * lo;
        dc.f = reinterpret_cast<const double*>(dv + n * 16) + 3 * lo;
        dc.pos_world = reinterpret_cast<const double*>(dv + n * 40) + 3 * lo;
        dc.flags = dv + n * 64 + lo;
      }
    }
  }
  const PosFromSeedsJob* pos_jobs_device = nullptr;
  if (!pos_jobs.empty()) {   // the job table rides the same upload
    memcpy(hup + up_off, pos_jobs.data(), sizeof(PosFromSeedsJob) * pos_jobs.size());
    pos_jobs_device = reinterpret_cast<const PosFromSeedsJob*>(dup + up_off);
    up_off += (sizeof(PosFromSeedsJob) * pos_jobs.size() + 63) & ~(size_t)63;
  }
  SVOH_HIP_TRY(ctx, svoh_copy_to_device(ctx, ctx->d_desc.ptr, h_desc.ptr, up_base + up_off));
  if (pos_jobs_device) {
    const int rcp = svoh_launch_pos_from_seed_batch(ctx, (int)pos_jobs.size(), pos_jobs_max_n, pos_jobs_device);
    if (rcp != SVOH_OK) return rcp;
  }
  ctx->align_desc_slot = desc_slot;
  ctx->align_staged_event_valid = false;
  if (ctx->align_launches_since_drain >= 1) {   // queued behind a launch nobody has waited for: the next one may need this
    SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_align_staged, ctx->stream));
    ctx->align_staged_event_valid = true;
  }
  ++ctx->align_launches_since_drain;

  AlignKernelArgs args;
  args.problems = static_cast<const DevProblemDesc*>(ctx->d_desc.ptr);
  args.cams = reinterpret_cast<const DevCamDesc*>(args.problems + n_desc);
  args.results = static_cast<svoh_align_result*>(ctx->d_results.ptr) + dev_results_off;
  double* w = static_cast<double*>(ctx->d_feat.ptr);
  args.wpk = w;
  args.slots = (int64_t)feat_slots;
  args.wsel = reinterpret_cast<uint8_t*>(w + 2 * kWsPairs * feat_slots);
  args.wvis = args.wsel + feat_slots;
  args.opt = *opt;
  args.eval_level = eval_level;
  args.eval_out = static_cast<double*>(ctx->d_eval.ptr);
  args.stamps = nullptr;
  args.n_problems = n_desc;
  args.ext_state = nullptr;
  args.raw_sums = 0;
  args.cluster = 0;
  args.xchg = nullptr;
  args.bar = nullptr;
#ifdef SVOH_TEST_HOOKS
  args.cluster_test_absent = SvohKnobs::or_default(ctx->knobs.align_cluster_test_absent, 0);
#endif
  if (cluster) {
    const size_t xchg_bytes = 2 * (size_t)S * kXchgStride * sizeof(double) * n_problems;
    SVOH_HIP_TRY(ctx, ctx->d_xchg.reserve(xchg_bytes));
    args.cluster = S;
    args.xchg = static_cast<double*>(ctx->d_xchg.ptr);
    args.bar = reinterpret_cast<unsigned int*>(static_cast<uint8_t*>(ctx->d_desc.ptr) + ctl_off + 256);   // zero, as above
  }
  if (split && !split->update) {
    args.ext_state = split->ext_state;
    if (S == 1) args.eval_out = split->sums_out;
    args.raw_sums = 1;
  }
  if (split && split->update) {
    // the serial step only: one lane, no feature work
    const bool illum_u = opt->estimate_illumination_gain || opt->estimate_illumination_offset;
    if (opt->patch_size == 4) {
      if (illum_u) hipLaunchKernelGGL((align_gn_update_kernel<4, true>), dim3(1), dim3(64), 0, ctx->stream, args, split->sums_in, split->state, eval_level, split->iter);
      else hipLaunchKernelGGL((align_gn_update_kernel<4, false>), dim3(1), dim3(64), 0, ctx->stream, args, split->sums_in, split->state, eval_level, split->iter);
    } else {
      if (illum_u) hipLaunchKernelGGL((align_gn_update_kernel<8, true>), dim3(1), dim3(64), 0, ctx->stream, args, split->sums_in, split->state, eval_level, split->iter);
      else hipLaunchKernelGGL((align_gn_update_kernel<8, false>), dim3(1), dim3(64), 0, ctx->stream, args, split->sums_in, split->state, eval_level, split->iter);
    }
    const hipError_t eu = hipGetLastError();
    if (eu != hipSuccess) return set_error(ctx, SVOH_ERR_HIP, "gn_update launch failed: %s", hipGetErrorString(eu));
    return SVOH_OK;
  }
  args.queue = reinterpret_cast<int32_t*>(static_cast<uint8_t*>(ctx->d_desc.ptr) + ctl_off);   // zero: uploaded with the descriptors
#ifdef SVOH_PHASE_STAMPS
  SVOH_HIP_TRY(ctx, ctx->d_scratch0.reserve(sizeof(long long) * 20 * (size_t)n_problems));
  args.stamps = static_cast<long long*>(ctx->d_scratch0.ptr);
#endif

  // geometry (decide_geometry): chosen from the launch, or the caller's (a keyed launch: the geometry of a problem alone)
  const AlignGeometry geo = forced ? *forced : decide_geometry(ctx, opt, n_desc, cluster ? S : 0, shape);
  const int nt = geo.nt, rows = geo.rows;
  // 256 threads: two workgroups per CU, each with <= 29 KB of static LDS (reduction scratch, the 24 KB LDS-DMA staging
  // area of the workspace rows) -> 51 KB for images: levels 4, 3 and 2 of a 640x480 pyramid side by side (50 400 B);
  // launch_one trims the image area to what the instantiation's static LDS really leaves of half a compute unit
  size_t lds = (nt == 256) ? 52224 : 78 * 1024;
  args.latency_build = geo.latency ? 1 : 0;
  args.rig_build = geo.rig ? 1 : 0;
  args.lds_two_per_cu = (nt == 256 && !cluster && ctx->knobs.align_lds == kKnobUnset && ctx->knobs.align_wg_per_cu == kKnobUnset) ? 1 : 0;
  lds = (size_t)SvohKnobs::or_default(ctx->knobs.align_lds, (int)lds) & ~(size_t)15;   // the workspace behind the image area is read through 16-byte loads
  // 160 KB per workgroup minus the kernel's static LDS
  const size_t lds_cap = (nt == 256 && SVOH_ALIGN_STAGED) ? 163840 - 32768 : 153856;
  if (lds > lds_cap) lds = lds_cap;
  // a small problem's feature workspace in LDS, behind the image area (512-thread geometry: 78 KB hold the 50 KB of the
  // levels 4..2 of a 640x480 pair and 16 KB of rows; the next finer level would not fit either way)
  args.ws_lds_bytes = 0;
  if (nt == 512 && !cluster && max_feat_per_problem <= kWsLdsMaxFeatures && lds >= (size_t)kWsLdsMaxFeatures * kWsPairs * 16 + 4096) {
    args.ws_lds_bytes = ((max_feat_per_problem > 0 ? max_feat_per_problem : 1) * kWsPairs * 16 + 15) & ~15;
  }
  args.lds_img_bytes = (int32_t)(lds - (size_t)args.ws_lds_bytes);   // the launch still asks for all of `lds`

  const bool illum = opt->estimate_illumination_gain || opt->estimate_illumination_offset;
  // resident workgroups per CU: 256-thread groups at 256 VGPRs -> 2; larger groups -> 1
  int grid = ctx->num_cus * SvohKnobs::or_default(ctx->knobs.align_wg_per_cu, nt == 256 ? 2 : 1);
  if (grid > n_desc || grid <= 0) grid = n_desc;
  if (cluster) grid = n_desc;
  args.one_each = (!cluster && grid == n_desc) ? 1 : 0;
#if SVOH_ALIGN_CLUSTER_XCD
  if (cluster) grid = 8 * args.cluster * ((n_desc / args.cluster + 7) / 8);   // (see the kernel's pull)
#endif
  hipError_t e;
  const bool timed = ctx->timing_on();
  const int ev_slot = (int)(ctx->align_timed_launches % svoh_ctx::kAlignEventRing);
  if (timed) SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_align_start[ev_slot], ctx->stream));
  const bool robust = opt->robustification != 0;
#ifdef SVOH_DEV_ONLY_PLAIN   // development builds only (scripts/kernel_resources.sh, quick A/B libraries): the 4x4 / 8x8 kernels without illumination terms and robust weights
  if (robust || illum) return set_error(ctx, SVOH_ERR_UNSUPPORTED, "development build: plain kernels only");
  e = opt->patch_size == 4 ? launch_nt<4, false, false>(ctx->stream, nt, rows, grid, lds, args)
                           : launch_nt<8, false, false>(ctx->stream, nt, rows, grid, lds, args);
#else
  if (opt->patch_size == 4) {
    if (robust) e = illum ? launch_nt<4, true, true>(ctx->stream, nt, rows, grid, lds, args) : launch_nt<4, false, true>(ctx->stream, nt, rows, grid, lds, args);
    else e = illum ? launch_nt<4, true, false>(ctx->stream, nt, rows, grid, lds, args) : launch_nt<4, false, false>(ctx->stream, nt, rows, grid, lds, args);
  } else {
    if (robust) e = illum ? launch_nt<8, true, true>(ctx->stream, nt, rows, grid, lds, args) : launch_nt<8, false, true>(ctx->stream, nt, rows, grid, lds, args);
    else e = illum ? launch_nt<8, true, false>(ctx->stream, nt, rows, grid, lds, args) : launch_nt<8, false, false>(ctx->stream, nt, rows, grid, lds, args);
  }
#endif
  if (e != hipSuccess)
    return set_error(ctx, SVOH_ERR_HIP, "sparse_align launch failed: %s", hipGetErrorString(e));
  if (S > 1 && !cluster) {
    hipLaunchKernelGGL(sum_shares_kernel, dim3(1), dim3(128), 0, ctx->stream, static_cast<const double*>(ctx->d_eval.ptr), S,
                       split->sums_out);
    const hipError_t es = hipGetLastError();
    if (es != hipSuccess) return set_error(ctx, SVOH_ERR_HIP, "sum_shares launch failed: %s", hipGetErrorString(es));
  }
  if (timed) { SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_align_stop[ev_slot], ctx->stream)); ++ctx->align_timed_launches; }
  ctx->align_last_timed = timed;
  ++ctx->align_launches;
  // the results follow the kernel to pinned host memory right away, so that a caller which queues several
  // launches and fetches once still has every launch's output delivered
  if (delivers) {   // cluster: entry 0 is share 0's copy of the common result
    SVOH_HIP_TRY(ctx, svoh_copy_to_host(ctx, static_cast<svoh_align_result*>(ctx->h_results.ptr) + ctx->align_pending_results,
                                        args.results, sizeof(svoh_align_result) * n_problems));
    ctx->align_last_results_off = ctx->align_pending_results;
    // where on the device result #k of the queue lives (svoh_project_candidates_enqueue's align_result_index)
    if (ctx->align_pending_results == 0) ctx->align_result_dev_index.clear();
    for (int p = 0; p < n_problems; ++p) ctx->align_result_dev_index.push_back((uint32_t)(dev_results_off + (size_t)p));
    ctx->align_pending_results += (size_t)n_problems;
    ctx->align_pending_dev = dev_results_off + (size_t)n_desc + (size_t)n_problems;
  }
#ifdef SVOH_PHASE_STAMPS
  {
    std::vector<long long> h((size_t)n_problems * 20);
    SVOH_HIP_TRY(ctx, hipMemcpyAsync(h.data(), args.stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    SVOH_ALIGN_DRAIN(ctx);
    double sum[20] = {0};
    for (int p = 0; p < n_problems; ++p) for (int k = 0; k < 20; ++k) sum[k] += (double)h[(size_t)p * 20 + k];
    fprintf(stderr, "[stamps] one-lane step: set-up %.0f solve %.0f update %.0f camera poses %.0f\n", sum[8] / n_problems, sum[9] / n_problems,
            sum[10] / n_problems, sum[11] / n_problems);
    fprintf(stderr, "[stamps] inside the passes (thread 0): set-up %.0f row arrival %.0f projection %.0f pixels %.0f Jacobian rows + map %.0f; gradient-only passes as a whole %.0f, the vote's barrier behind them %.0f\n", sum[12] / n_problems,
            sum[13] / n_problems, sum[14] / n_problems, sum[15] / n_problems, sum[16] / n_problems, sum[17] / n_problems, sum[18] / n_problems);
    fprintf(stderr, "[stamps] n=%d nt=%d avg cycles/block: base %.0f stage %.0f patch %.0f reduce %.0f serial %.0f total %.0f; "
            "gradient-only passes kept %.2f / discarded %.2f per problem\n",
            n_problems, nt, sum[0] / n_problems, sum[1] / n_problems, sum[2] / n_problems, sum[3] / n_problems,
            sum[4] / n_problems, sum[7] / n_problems, sum[6] / n_problems, sum[5] / n_problems);
  }
#endif
  // what svoh_sparse_align_fetch may hand out: only a launch that delivered into h_results (not an evaluation, not
  // the shares of a patch-split evaluation, whose result slots are not a caller's problems)
  ctx->last_align_n = delivers ? n_problems : 0;
  return SVOH_OK;
}

}  // namespace svoh

using namespace svoh;

extern "C" {

int svoh_sparse_align_enqueue(svoh_ctx* ctx, const svoh_align_options* options, int n_problems,
                              const svoh_align_problem* problems)
try {
  return enqueue_align(ctx, options, n_problems, problems, -1);
} SVOH_ABI_CATCH(ctx)

int svoh_sparse_align_geometry_key(svoh_ctx* ctx, const svoh_align_options* options, const svoh_align_problem* problem, int32_t* key)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  const int rc = validate_options(ctx, options);
  if (rc != SVOH_OK) return rc;
  SVOH_REQUIRE(ctx, problem && key, "NULL argument");
  SVOH_REQUIRE(ctx, problem->n_cams >= 1 && problem->n_cams <= SVOH_MAX_CAMS, "n_cams out of range");
  *key = geometry_of_single(ctx, options, *problem).key();
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_set_align_geometry_classes(svoh_ctx* ctx, int shared)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, shared == 0 || shared == 1, "geometry classes: 0 = the fastest geometry for a problem alone, 1 = shared classes");
  ctx->align_shared_classes = shared != 0;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_sparse_align_enqueue_keyed(svoh_ctx* ctx, const svoh_align_options* options, int n_problems,
                                    const svoh_align_problem* problems, int32_t key)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  AlignGeometry geo;
  SVOH_REQUIRE(ctx, AlignGeometry::from_key(key, &geo), "not a geometry key of svoh_sparse_align_geometry_key");
  SVOH_REQUIRE(ctx, n_problems >= 1 && problems, "no problems");
  // cluster mode wants every workgroup of a launch on a compute unit of its own: more problems than that go out as
  // several launches, one behind the other (their results queue up in problem order)
  // -- and never more than the arrival counters of one launch hold (kClusterMaxProblems words at ctl + 256: the uploaded
  // feature arrays start right behind them)
  int per_launch = n_problems;
  if (geo.cluster_g >= 2) {
    per_launch = ctx->num_cus / geo.cluster_g > 0 ? ctx->num_cus / geo.cluster_g : 1;
    if (per_launch > kClusterMaxProblems) per_launch = kClusterMaxProblems;
  }
  for (int p0 = 0; p0 < n_problems; p0 += per_launch) {
    const int n = n_problems - p0 < per_launch ? n_problems - p0 : per_launch;
    const int rc = enqueue_align(ctx, options, n, problems + p0, -1, nullptr, &geo);
    if (rc != SVOH_OK) return rc;
  }
  // svoh_sparse_align_fetch hands out the LAST launch's results: a keyed call that was split is fetched with _fetch_all
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_sparse_align_fetch(svoh_ctx* ctx, int n_problems, svoh_align_result* results)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, results && n_problems >= 1 && n_problems <= ctx->last_align_n && ctx->h_results.ptr &&
                        ctx->align_pending_results >= (size_t)n_problems, "nothing to fetch");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  SVOH_ALIGN_DRAIN(ctx);   // the copy to h_results was queued behind the kernel
  memcpy(results, static_cast<const svoh_align_result*>(ctx->h_results.ptr) + ctx->align_last_results_off,
         sizeof(svoh_align_result) * n_problems);
  ctx->align_pending_results = 0;
  ctx->align_pending_dev = 0;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_sparse_align_fetch_all(svoh_ctx* ctx, int n_results, svoh_align_result* results)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, results && n_results >= 1 && (size_t)n_results == ctx->align_pending_results,
               "n_results is not the number of results queued since the last fetch");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  SVOH_ALIGN_DRAIN(ctx);
  memcpy(results, ctx->h_results.ptr, sizeof(svoh_align_result) * (size_t)n_results);
  ctx->align_pending_results = 0;
  ctx->align_pending_dev = 0;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_sparse_align_batch(svoh_ctx* ctx, const svoh_align_options* options, int n_problems,
                            const svoh_align_problem* problems, svoh_align_result* results)
try {
  int rc = enqueue_align(ctx, options, n_problems, problems, -1);
  if (rc != SVOH_OK) return rc;
  rc = svoh_sparse_align_fetch(ctx, n_problems, results);
  if (rc != SVOH_OK) return rc;
  // Cluster mode gives up (status 3) when the workgroups of a problem do not all become resident within its
  // bounded wait -- possible when something else holds the device.  The call does not fail for that: it runs
  // the problems again, one workgroup each.
  bool gave_up = false;
  for (int p = 0; p < n_problems; ++p) gave_up = gave_up || results[p].status == 3;
  if (gave_up) {
    ctx->align_no_cluster = true;
    rc = enqueue_align(ctx, options, n_problems, problems, -1);
    ctx->align_no_cluster = false;
    if (rc != SVOH_OK) return rc;
    rc = svoh_sparse_align_fetch(ctx, n_problems, results);
  }
  return rc;
} SVOH_ABI_CATCH(ctx)

int svoh_sparse_align_last_kernel_ms(svoh_ctx* ctx, float* ms)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, ms != nullptr && ctx->align_timed_launches > 0 && ctx->align_last_timed,
               "the last alignment launch was not timed (svoh_set_kernel_timing)");
  const int slot = (int)((ctx->align_timed_launches - 1) % svoh_ctx::kAlignEventRing);
  SVOH_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_align_stop[slot]));
  SVOH_HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev_align_start[slot], ctx->ev_align_stop[slot]));
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_sparse_align_kernel_ms_history(svoh_ctx* ctx, int n, float* ms, int* n_out)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, ms != nullptr && n_out != nullptr && n >= 0, "bad arguments");
  int have = (int)(ctx->align_timed_launches < (unsigned long long)svoh_ctx::kAlignEventRing ? ctx->align_timed_launches
                                                                                             : (unsigned long long)svoh_ctx::kAlignEventRing);
  if (n < have) have = n;
  for (int k = 0; k < have; ++k) {   // oldest of the requested launches first
    const int slot = (int)((ctx->align_timed_launches - (unsigned long long)have + (unsigned long long)k) % svoh_ctx::kAlignEventRing);
    SVOH_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_align_stop[slot]));
    SVOH_HIP_TRY(ctx, hipEventElapsedTime(&ms[k], ctx->ev_align_start[slot], ctx->ev_align_stop[slot]));
  }
  *n_out = have;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_sparse_align_evaluate(svoh_ctx* ctx, const svoh_align_options* options, const svoh_align_problem* problem,
                               int level, double* H64, double* g8, double* chi2, int32_t* n_meas,
                               uint8_t* visibility, int32_t* n_selected)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, problem && H64 && g8, "NULL argument");
  SVOH_REQUIRE(ctx, level >= 0 && options && level <= options->max_level, "level out of range");
  int rc = enqueue_align(ctx, options, 1, problem, level);
  if (rc != SVOH_OK) return rc;
  double out[74];
  SVOH_HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_eval.ptr, sizeof out, hipMemcpyDeviceToHost, ctx->stream));
  int nf = 0;
  for (int c = 0; c < problem->n_cams; ++c) nf += problem->cams[c].n_features;
  std::vector<uint8_t> sel((size_t)nf + 1), vis((size_t)nf + 1);
  // workspace layout: see enqueue_align
  const size_t slots = nf ? (size_t)nf : 1;
  const uint8_t* dsel = reinterpret_cast<const uint8_t*>(static_cast<double*>(ctx->d_feat.ptr) + 2 * kWsPairs * slots);
  if (nf) {
    SVOH_HIP_TRY(ctx, hipMemcpyAsync(sel.data(), dsel, (size_t)nf, hipMemcpyDeviceToHost, ctx->stream));
    SVOH_HIP_TRY(ctx, hipMemcpyAsync(vis.data(), dsel + slots, (size_t)nf, hipMemcpyDeviceToHost, ctx->stream));
  }
  SVOH_ALIGN_DRAIN(ctx);
  memcpy(H64, out, 64 * sizeof(double));
  memcpy(g8, out + 64, 8 * sizeof(double));
  if (chi2) *chi2 = out[72];
  if (n_meas) *n_meas = (int32_t)out[73];
  int k = 0;
  for (int i = 0; i < nf; ++i)
    if (sel[i]) { if (visibility) visibility[k] = vis[i]; ++k; }
  if (n_selected) *n_selected = k;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_sparse_align_split_buffers(svoh_ctx* ctx, svoh_align_gn_state** d_state, double** d_sums)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, d_state && d_sums, "NULL argument");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t state_bytes = (sizeof(svoh_align_gn_state) + 255) & ~(size_t)255;
  SVOH_HIP_TRY(ctx, ctx->d_split.reserve(state_bytes + SVOH_ALIGN_SUMS_DOUBLES * sizeof(double)));
  *d_state = static_cast<svoh_align_gn_state*>(ctx->d_split.ptr);
  *d_sums = reinterpret_cast<double*>(static_cast<uint8_t*>(ctx->d_split.ptr) + state_bytes);
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_sparse_align_split_init(svoh_ctx* ctx, const svoh_align_problem* problem, svoh_align_gn_state* d_state)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, problem && d_state, "NULL argument");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  svoh_align_gn_state h;
  memset(&h, 0, sizeof h);
  h.T_icur_iref = problem->T_icur_iref;
  h.T_old = problem->T_icur_iref;
  h.alpha = h.alpha_old = problem->alpha_init;
  h.beta = h.beta_old = problem->beta_init;
  SVOH_HIP_TRY(ctx, hipMemcpyAsync(d_state, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
  SVOH_ALIGN_DRAIN(ctx);   // h is a stack object
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_sparse_align_partial_sums(svoh_ctx* ctx, const svoh_align_options* options, const svoh_align_problem* problem,
                                   int level, int n_workgroups, const svoh_align_gn_state* d_state, double* d_sums)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, problem && d_state && d_sums, "NULL argument");
  SVOH_REQUIRE(ctx, options && level >= 0 && level <= options->max_level, "level out of range");
  SVOH_REQUIRE(ctx, n_workgroups >= 0 && n_workgroups <= 4096, "n_workgroups out of range");
  SVOH_REQUIRE(ctx, problem->n_cams >= 1 && problem->n_cams <= SVOH_MAX_CAMS, "n_cams out of range");
  if (n_workgroups == 0) {
    // about 256 patches per workgroup (one per lane of the 256-thread geometry), at most two workgroups per CU
    int64_t nf = 0;
    for (int c = 0; c < problem->n_cams; ++c) nf += problem->cams[c].n_features > 0 ? problem->cams[c].n_features : 0;
    int64_t w = (nf + 255) / 256;
    if (w > 2 * ctx->num_cus) w = 2 * ctx->num_cus;
    n_workgroups = w < 1 ? 1 : (int)w;
  }
  SplitArgs sp;
  sp.ext_state = d_state;
  sp.sums_out = d_sums;
  sp.n_shares = n_workgroups;
  return enqueue_align(ctx, options, 1, problem, level, &sp);
} SVOH_ABI_CATCH(ctx)

int svoh_sparse_align_gn_update(svoh_ctx* ctx, const svoh_align_options* options, const svoh_align_problem* problem,
                                int level, int iter, const double* d_sums, svoh_align_gn_state* d_state,
                                svoh_align_gn_state* h_state)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, problem && d_state && d_sums, "NULL argument");
  SVOH_REQUIRE(ctx, options && level >= 0 && level <= options->max_level && iter >= 0, "level / iteration out of range");
  SplitArgs sp;
  sp.sums_in = d_sums;
  sp.state = d_state;
  sp.iter = iter;
  sp.update = true;
  int rc = enqueue_align(ctx, options, 1, problem, level, &sp);
  if (rc != SVOH_OK) return rc;
  if (h_state)
    SVOH_HIP_TRY(ctx, hipMemcpyAsync(h_state, d_state, sizeof *h_state, hipMemcpyDeviceToHost, ctx->stream));
  SVOH_ALIGN_DRAIN(ctx);
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

}  // extern "C"
