// pose.hip -- batched pose optimiser for gfx950.
//
// Replaces PoseOptimizer::run (src/svo/src/pose_optimizer.cpp:39-113): start errors -> MAD scale
// (robust_cost.cpp:21-27), Gauss-Newton on T_imu_world (mini_least_squares_solver.hpp:42-107) over the
// six residual kinds of pose_optimizer_utils (:338-627) with Tukey weights, optional rotation prior
// (:320-334), update T <- exp(dx) T (:309-318), removeOutliers (:198-307).
//
// Same skeleton as the alignment kernel: one workgroup owns one frame bundle for the whole
// optimisation, one thread per feature, fp64 like the reference, the 27 normal-equation accumulators go
// through the wave reduce-scatter, one lane runs the 6x6 pivoted LDL^T and the update.  The problem is
// tiny (<= a few hundred 2-D residuals): the point of running it here is batching many bundles per
// launch (multi-stream) and keeping the per-frame chain on the device; one bundle alone is latency-bound.
// Medians (MAD scale, statistics) are rank selections in LDS -- order-independent, so identical to
// nth_element; sums differ from the sequential reference in rounding only.
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <thread>
#include <type_traits>
#include <vector>

#include "svoh_internal.h"
#include "svoh_math.h"
#include "svoh_device_utils.h"

namespace svoh {

constexpr int kPoseThreads = 256;
constexpr int kPoseMaxMeas = 4096;   // measurements per bundle (LDS-resident error list)
constexpr size_t kPoseParallelStagingFeatures = 65536;   // host staging of a batch this large is split over threads

struct DevPoseCam {
  svoh_camera cam;
  svoh_se3 T_cam_imu;
  int n_features;
  int feat_off;                // offset of this camera's features in the problem's concatenated arrays
};

struct DevPoseProblem {
  int n_cams, n_total;
  int cam_begin;               // into cams[]
  long long arr_off;           // offset (in features) of the problem's arrays
  svoh_se3 T_imu_world;
};

struct PoseArgs {
  svoh_pose_options opt;
  const DevPoseProblem* problems;
  const DevPoseCam* cams;
  // concatenated over all problems and cameras
  const double* px; const double* f; const double* grad; const int32_t* level; const uint8_t* type;
  const double* xyz; const uint8_t* usable;
  uint8_t* outlier; double* final_error;
  svoh_pose_result* results;
  int n_problems;
  int max_meas;   // capacity of the LDS error list (largest bundle of the batch, rounded up to 64)
};

__device__ __forceinline__ float tukey_weight_f(float e)
{
  const float b2 = 4.6851f * 4.6851f;
  const float x2 = e * e;
  if (x2 <= b2) { const float t = 1.0f - x2 / b2; return t * t; }
  return 0.0f;
}

__device__ __forceinline__ bool is_edgelet_type(int t) { return t == SVOH_FT_EDGELET || t == SVOH_FT_EDGELET_SEED || t == SVOH_FT_EDGELET_SEED_CONVERGED; }

// J (rows x 6) = A (rows x 3) * R_cam_imu * [I | -skew(p_in_imu)]   (frame.h:342-397)
template <int ROWS>
__device__ __forceinline__ void chain_G(const double* A, const double* R, const Vec3& p, double* J)
{
  const double S[9] = { 0, p.z, -p.y, -p.z, 0, p.x, p.y, -p.x, 0 };
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    double AR[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) AR[c] = A[r * 3 + 0] * R[0 + c] + A[r * 3 + 1] * R[3 + c] + A[r * 3 + 2] * R[6 + c];
#pragma unroll
    for (int c = 0; c < 3; ++c) J[r * 6 + c] = AR[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) J[r * 6 + 3 + c] = AR[0] * S[0 + c] + AR[1] * S[3 + c] + AR[2] * S[6 + c];
  }
}

// one measurement (pose_optimizer.cpp:338-627); acc = upper triangle of H (21) + g (6), row-major packing
// ET (the error type) and WANT_J are compile-time: each kernel instance carries one error model only, which is what
// keeps it within 256 registers (two waves per SIMD)
template <int ET, bool WANT_J>
__device__ __forceinline__ void pose_residual(const CamModel& cm, const Rigid& T_cam_imu, const double* Rci,
                              const Rigid& T_imu_world, const double* f, const double* px, const double* grad,
                              const Vec3& xyz_world, bool edgelet, double measurement_sigma, double& unwhitened_error,
                              double* acc /* may be NULL */)
{
  const Vec3 p_imu = transform(T_imu_world, xyz_world);
  const Vec3 p_cam = transform(T_cam_imu, p_imu);
  const double Rw = 1.0 / measurement_sigma;
  double e[3] = { 0, 0, 0 };
  double J[18];
  int rows;
  constexpr bool want_J = WANT_J;
  if (ET == SVOH_POSE_ERR_UNIT_PLANE) {
    const double d0 = f[0] / f[2] - p_cam.x / p_cam.z, d1 = f[1] / f[2] - p_cam.y / p_cam.z;
    double Juv[12];
    if (want_J) {
      const double s = -1.0 / p_cam.z;
      const double A[6] = { s * 1.0, 0.0, s * (-p_cam.x / p_cam.z), 0.0, s * 1.0, s * (-p_cam.y / p_cam.z) };
      chain_G<2>(A, Rci, p_imu, Juv);
    }
    if (!edgelet) {
      rows = 2; e[0] = d0; e[1] = d1;
      if (want_J) for (int c = 0; c < 12; ++c) J[c] = Juv[c];
    } else {
      rows = 1; e[0] = grad[0] * d0 + grad[1] * d1;
      if (want_J) for (int c = 0; c < 6; ++c) J[c] = grad[0] * Juv[c] + grad[1] * Juv[6 + c];
    }
  } else if (ET == SVOH_POSE_ERR_IMAGE_PLANE) {
    double u, v, Jc[6], Jimg[12];
    project3(cm, p_cam, u, v);
    project3_jacobian(cm, p_cam, Jc);
    const double d0 = px[0] - u, d1 = px[1] - v;
    if (want_J) { chain_G<2>(Jc, Rci, p_imu, Jimg); for (int c = 0; c < 12; ++c) Jimg[c] = (-1.0) * Jimg[c]; }
    if (!edgelet) {
      rows = 2; e[0] = d0; e[1] = d1;
      if (want_J) for (int c = 0; c < 12; ++c) J[c] = Jimg[c];
    } else {
      rows = 1; e[0] = grad[0] * d0 + grad[1] * d1;
      if (want_J) for (int c = 0; c < 6; ++c) J[c] = grad[0] * Jimg[c] + grad[1] * Jimg[6 + c];
    }
  } else {
    const double x2 = p_cam.x * p_cam.x, y2 = p_cam.y * p_cam.y, z2 = p_cam.z * p_cam.z;
    const double nrm = sqrt(x2 + y2 + z2);
    const double fd[3] = { f[0] - p_cam.x / nrm, f[1] - p_cam.y / nrm, f[2] - p_cam.z / nrm };
    double Jb[18];
    if (want_J) {
      const double xy = p_cam.x * p_cam.y, yz = p_cam.y * p_cam.z, zx = p_cam.z * p_cam.x;
      const double k = 1 / pow(x2 + y2 + z2, 1.5);
      const double A[9] = { k * (y2 + z2), k * -xy, k * -zx, k * -xy, k * (x2 + z2), k * -yz, k * -zx, k * -yz, k * (x2 + y2) };
      chain_G<3>(A, Rci, p_imu, Jb);
    }
    if (!edgelet) {
      rows = 3; e[0] = fd[0]; e[1] = fd[1]; e[2] = fd[2];
      if (want_J) for (int c = 0; c < 18; ++c) J[c] = (-1.0) * Jb[c];
    } else {
      double u, v, Jc[6];
      project3(cm, p_cam, u, v);
      project3_jacobian(cm, p_cam, Jc);
      const double pd[2] = { px[0] - u, px[1] - v };
      const double pd2 = pd[0] * pd[0] + pd[1] * pd[1];
      const double fd2 = fd[0] * fd[0] + fd[1] * fd[1] + fd[2] * fd[2];
      const double e_img = grad[0] * pd[0] + grad[1] * pd[1];
      const double scale_ratio = sqrt(fd2) / sqrt(pd2);
      rows = 1; e[0] = e_img * scale_ratio;
      if (want_J) {
        double Jp[12];
        chain_G<2>(Jc, Rci, p_imu, Jp);
        for (int c = 0; c < 6; ++c) {
          const double J_img = (grad[0] * (-1.0)) * Jp[c] + (grad[1] * (-1.0)) * Jp[6 + c];
          const double J_ftf = (2 * fd[0] * (-1.0)) * Jb[c] + (2 * fd[1] * (-1.0)) * Jb[6 + c] + (2 * fd[2] * (-1.0)) * Jb[12 + c];
          const double J_ptp = (2 * pd[0] * (-1.0)) * Jp[c] + (2 * pd[1] * (-1.0)) * Jp[6 + c];
          const double J_ratio = (0.5) * (1.0 / (scale_ratio)) * (1 / (pd2 * pd2)) * (J_ftf * pd2 - J_ptp * fd2);
          J[c] = e_img * J_ratio + scale_ratio * J_img;
        }
      }
    }
  }
  // rows is 1, 2 or 3: the loops below are unrolled to three with a row test, so that J and e stay in registers
  double en2 = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r) if (r < rows) en2 += e[r] * e[r];
  unwhitened_error = rows == 1 ? fabs(e[0]) : sqrt(en2);
  if (!want_J) return;
#pragma unroll
  for (int r = 0; r < 3; ++r) if (r < rows) e[r] *= Rw;
  const double en = rows == 1 ? e[0] : sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
  const double weight = (double)tukey_weight_f((float)en);
#pragma unroll
  for (int r = 0; r < 3; ++r)
    if (r < rows) {
#pragma unroll
      for (int c = 0; c < 6; ++c) J[r * 6 + c] *= Rw;
    }
  int idx = 0;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
#pragma unroll
    for (int b = a; b < 6; ++b) {
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r) if (r < rows) s += J[r * 6 + a] * J[r * 6 + b];
      acc[idx++] += s * weight;
    }
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) if (r < rows) s += J[r * 6 + a] * e[r];
    acc[21 + a] -= s * weight;
  }
}

// value of rank k (0-based, ties by index) among vals[0..n): what nth_element leaves at position k
template <int NT, typename T>
__device__ T rank_select(const T* vals, int n, int k, int tid, T* s_out)
{
  for (int i = tid; i < n; i += NT) {
    const T v = vals[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const T u = vals[j];
      rank += (u < v) || (u == v && j < i);
    }
    if (rank == k) *s_out = v;
  }
  __syncthreads();
  return *s_out;
}

// NT = 512: a single bundle of more than 256 features (a stereo rig's 2 x 160): one feature per lane instead of two
// rounds; two waves per SIMD, so the 256-register error types only (the bearing-vector difference needs 334).
// NT = 256: one bundle spread over four waves (lowest latency of a single bundle).  NT = 64: one wave per bundle,
// about three features per lane; four bundles share a compute unit (the kernel needs >256 registers, so a SIMD
// holds one wave), and the one-lane solve of one bundle overlaps the residuals of the other three: the geometry for
// batches that outnumber the compute units.
// pyramid level of a feature as the shift count of its scale: host arrays are checked to be 0..29 before the launch,
// device-resident ones are clamped here (a level outside that range is a caller's error; it must not become a
// negative scale or an undefined shift)
__device__ __forceinline__ int pose_level(int l) { return l < 0 ? 0 : (l > 29 ? 29 : l); }

template <int NT, int ET>
__global__ __launch_bounds__(NT) void pose_optimize_kernel(const PoseArgs a)
{
  constexpr int NACC = 27, NW = NT / 64;
  // start errors as float values, later final errors (double); sized by the launch to the largest bundle of the
  // batch (a.max_meas), so that bundles of a few hundred features leave room for eight workgroups per compute unit
  extern __shared__ double s_err[];
  __shared__ double s_red[NW][NACC];
  __shared__ double s_sum[NACC];
  __shared__ Rigid s_T, s_Told;
  __shared__ double s_sigma, s_Iprior, s_median;
  __shared__ float s_medf;
  __shared__ int s_n, s_done, s_stop, s_del_e, s_del_c;
  // the wave-wide solve (ldlt_apply_wave, svoh_device_utils.h): factor, permutation, right-hand side, solution
  __shared__ double s_fact[21], s_rhs[6], s_dx[6];
  __shared__ int s_perm[6], s_nonzero;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pbi = blockIdx.x;
  if (pbi >= a.n_problems) return;
  const DevPoseProblem& pb = a.problems[pbi];
  const DevPoseCam* cams = a.cams + pb.cam_begin;
  const svoh_pose_options& opt = a.opt;
  float* s_errf = reinterpret_cast<float*>(s_err);
  if (tid == 0) {
    s_T = load_rigid(pb.T_imu_world); s_Told = s_T;
    s_n = 0; s_done = 0; s_stop = 0; s_del_e = 0; s_del_c = 0; s_Iprior = 0.0;
  }
  __syncthreads();

  // helper: iterate this thread's features.  The cameras' features form ONE index range (camera after camera) that the
  // workgroup strides through: a rig of two cameras with 160 features each is 320 lanes' worth of work -- one round of
  // the 512-thread geometry, not two rounds per camera.  A lane's camera only ever advances; its model is reloaded then.
  int n_total = 0;
  for (int c = 0; c < pb.n_cams; ++c) n_total += cams[c].n_features;
  auto for_each_feature = [&](auto&& fn) {
    int c = -1, lo = 0, hi = 0;   // camera of the current feature and its index range [lo, hi)
    CamModel cm = {};
    Rigid T_cam_imu = {};
    double Rci[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int j = tid; j < n_total; j += NT) {
      if (j >= hi) {
        do { ++c; lo = hi; hi += cams[c].n_features; } while (j >= hi);
        cm = load_camera(cams[c].cam);
        T_cam_imu = load_rigid(cams[c].T_cam_imu);
        to_matrix(T_cam_imu.q, Rci);
      }
      const long long gi = pb.arr_off + cams[c].feat_off + (j - lo);
      if (!a.usable[gi]) continue;
      fn(cm, T_cam_imu, Rci, gi);
    }
  };

  // ---- start errors and the MAD scale (pose_optimizer.cpp:48-52) ----
  {
    const Rigid T = s_T;
    for_each_feature([&](const CamModel& cm, const Rigid& T_cam_imu, const double* Rci, long long gi) {
      const int scale = 1 << pose_level(a.level[gi]);
      double ue;
      const Vec3 X = { a.xyz[3 * gi], a.xyz[3 * gi + 1], a.xyz[3 * gi + 2] };
      pose_residual<ET, false>(cm, T_cam_imu, Rci, T, &a.f[3 * gi], &a.px[2 * gi], &a.grad[2 * gi], X,
                               is_edgelet_type(a.type[gi]), 1.0, ue, nullptr);
      const int slot = atomicAdd(&s_n, 1);
      if (slot < a.max_meas) s_errf[slot] = (float)(ue / scale);
    });
  }
  __syncthreads();
  const int n_meas = s_n < a.max_meas ? s_n : a.max_meas;
  svoh_pose_result& res = a.results[pbi];
  if (n_meas == 0) {
    if (tid == 0) {
      memset(&res, 0, sizeof res);
      store_rigid(s_T, res.T_imu_world);
      res.status = 1;
    }
    for (int c = 0; c < pb.n_cams; ++c)
      for (int i = tid; i < cams[c].n_features; i += NT) {
        a.outlier[pb.arr_off + cams[c].feat_off + i] = 0;
        a.final_error[pb.arr_off + cams[c].feat_off + i] = 0.0;
      }
    return;
  }
  const float med = rank_select<NT, float>(s_errf, n_meas, n_meas / 2, tid, &s_medf);
  if (tid == 0) s_sigma = (double)(1.48f * med);
  __syncthreads();
  const double measurement_sigma = s_sigma;

  // ---- optimizeGaussNewton ----
  int iters = 0;
  for (int iter = 0; iter < opt.max_iter; ++iter) {
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
    const Rigid T = s_T;
    for_each_feature([&](const CamModel& cm, const Rigid& T_cam_imu, const double* Rci, long long gi) {
      const int scale = 1 << pose_level(a.level[gi]);
      const bool edgelet = is_edgelet_type(a.type[gi]);
      double sigma = measurement_sigma * scale;
      if (edgelet) sigma *= 2.0;   // kEdgeletSigmaExtraFactor
      double ue;
      const Vec3 X = { a.xyz[3 * gi], a.xyz[3 * gi + 1], a.xyz[3 * gi + 2] };
      pose_residual<ET, true>(cm, T_cam_imu, Rci, T, &a.f[3 * gi], &a.px[2 * gi], &a.grad[2 * gi], X, edgelet, sigma, ue, acc);
    });
    {
      int ridx;
      bool rvalid;
      wave_reduce_scatter<NACC>(acc, lane, ridx, rvalid);
      if (rvalid) s_red[wave][ridx] = acc[0];
    }
    __syncthreads();
    if (tid < NACC) {
      double v = 0.0;
      for (int w = 0; w < NW; ++w) v += s_red[w][tid];
      s_sum[tid] = v;
    }
    __syncthreads();
    ++iters;
    // The step between two passes, by the first wave (round 4; as the alignment kernel's gn_wave_step): lane 0 unpacks,
    // adds the prior and factorises (the Hessian is new in every iteration here); the substitution is taken by lanes
    // 0..5 (permutation as an index, six divisions in one, a forward step one multiply-add for all lanes behind it) and
    // the four divisions of the quaternion's normalisation by lanes 0..3 -- the operations of the one-lane step in
    // their order, so the same bits.  On one lane the step was three quarters of a single bundle's kernel.
    if (tid < 64) {
      if (tid == 0) {
        double m[21], xg[6];
        {
          int idx = 0;
          for (int r = 0; r < 6; ++r)
            for (int c = r; c < 6; ++c) SVOH_L(c, r) = s_sum[idx++];
          for (int r = 0; r < 6; ++r) xg[r] = s_sum[21 + r];
        }
        if (opt.have_rotation_prior) {   // applyPrior (pose_optimizer.cpp:320-334)
          if (iter == 0) {
            double hmax = 0;
            for (int j = 3; j < 6; ++j) hmax = fmax(hmax, fabs(SVOH_L(j, j)));
            s_Iprior = hmax * opt.prior_lambda;
          }
          for (int j = 3; j < 6; ++j) SVOH_L(j, j) += s_Iprior;
          Rigid prior;
          prior.q = { opt.R_prior[0], opt.R_prior[1], opt.R_prior[2], opt.R_prior[3] };
          prior.t = { 0.0, 0.0, 0.0 };
          double lg[6];
          rigid_log(mul(s_T, inverse(prior)), lg);
          for (int j = 3; j < 6; ++j) xg[j] -= s_Iprior * lg[j];
        }
        int tr[6];
        const bool nonzero = ldlt_factor_regs<6>(m, tr);
#pragma unroll
        for (int k = 0; k < 21; ++k) s_fact[k] = m[k];
        ldlt_perm_from_transpositions<6>(tr, s_perm);
        s_nonzero = nonzero ? 1 : 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) s_rhs[k] = xg[k];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      {
        const int pi = s_perm[tid < 6 ? tid : 0];
        const double x = ldlt_apply_wave<6>(s_fact, s_nonzero != 0, s_rhs[pi], tid);
        if (tid < 6) s_dx[pi] = x;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      double xg[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) xg[k] = s_dx[k];
      const bool stop = s_stop != 0 || xg[0] != xg[0];
      const Rigid T_old = s_T, T_older = s_Told;
      __builtin_amdgcn_wave_barrier();   // every lane has read the state lane 0 is about to replace
      if (stop) {
        if (tid == 0) { s_stop = 1; s_T = T_older; s_done = 1; }
      } else {
        Rigid Tn = mul(rigid_exp(xg), T_old);   // T_new = exp(dx) * T_old (pose_optimizer.cpp:309-318)
        const double nrm = sqrt(sqnorm(Tn.q));
        const double num = tid == 0 ? Tn.q.w : tid == 1 ? Tn.q.x : tid == 2 ? Tn.q.y : Tn.q.z;
        const double qd = num / nrm;             // normalized(Tn.q), one component per lane
        Tn.q.w = wave_bcast_f64(qd, 0); Tn.q.x = wave_bcast_f64(qd, 1); Tn.q.y = wave_bcast_f64(qd, 2); Tn.q.z = wave_bcast_f64(qd, 3);
        double x_norm = -1.0;
        for (int j = 0; j < 6; ++j) { const double v = fabs(xg[j]); if (v > x_norm) x_norm = v; }
        if (tid == 0) {
          s_Told = T_old;
          s_T = Tn;
          if (x_norm < opt.eps) s_done = 1;
        }
      }
    }
    __syncthreads();
    if (s_done) break;
  }

  // ---- removeOutliers (pose_optimizer.cpp:198-307) + statistics ----
  if (tid == 0) s_n = 0;
  __syncthreads();
  {
    const Rigid T = s_T;
    for (int c = 0; c < pb.n_cams; ++c)
      for (int i = tid; i < cams[c].n_features; i += NT) {
        const long long gi = pb.arr_off + cams[c].feat_off + i;
        if (!a.usable[gi]) { a.outlier[gi] = 0; a.final_error[gi] = 0.0; }
      }
    for_each_feature([&](const CamModel& cm, const Rigid& T_cam_imu, const double* Rci, long long gi) {
      const bool edgelet = is_edgelet_type(a.type[gi]);
      double ue;
      const Vec3 X = { a.xyz[3 * gi], a.xyz[3 * gi + 1], a.xyz[3 * gi + 2] };
      pose_residual<ET, false>(cm, T_cam_imu, Rci, T, &a.f[3 * gi], &a.px[2 * gi], &a.grad[2 * gi], X, edgelet, 1.0, ue, nullptr);
      ue *= 1.0 / (1 << pose_level(a.level[gi]));
      a.final_error[gi] = ue;
      const bool out = fabs(ue) > opt.outlier_threshold;
      a.outlier[gi] = out ? 1 : 0;
      if (out) atomicAdd(edgelet ? &s_del_e : &s_del_c, 1);
      const int slot = atomicAdd(&s_n, 1);
      if (slot < a.max_meas) s_err[slot] = ue;
    });
  }
  __syncthreads();
  const int n_final = s_n < a.max_meas ? s_n : a.max_meas;
  const double med_after = rank_select<NT, double>(s_err, n_final, n_final / 2, tid, &s_median);
  if (tid == 0) {
    store_rigid(s_T, res.T_imu_world);
    res.measurement_sigma = measurement_sigma;
    res.reproj_error_before = (double)med;
    res.reproj_error_after = med_after;
    res.n_meas = n_meas;
    res.n_deleted_edges = s_del_e;
    res.n_deleted_corners = s_del_c;
    res.iters = iters;
    res.status = s_stop ? 2 : 0;
    res.reserved = 0;
  }
}


// ---- Point::optimize (second half) ---------------------------------------------------------
// src/svo_common/src/point.cpp:216-325: 3-DoF Gauss-Newton on one landmark over its observations (unit plane, or
// unit sphere for omnidirectional cameras), pivoted LDL^T of the 3x3 system, stop when the error grows (the step
// is rolled back), on NaN, or when max|dp| <= 1e-10.  One thread per landmark: a landmark has a handful of
// observations and the batch is a keyframe's few hundred landmarks (FrameHandlerBase::optimizeStructure,
// frame_handler_base.cpp:779-825), so there is nothing to reduce across lanes.
struct PointArgs {
  const svoh_se3* T_f_w;      // per view
  const int32_t* obs_begin;   // n_points + 1
  const int32_t* obs_view;    // per observation
  const double* obs_f;        // 3 per observation
  double* pos;                // 3 per point, in/out
  int32_t* iters;             // per point
  int n_points, n_iter, on_sphere;
};

__global__ __launch_bounds__(256)
void point_optimize_kernel(const PointArgs a)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_points) return;
  const int o0 = a.obs_begin[i], o1 = a.obs_begin[i + 1];
  Vec3 pos = { a.pos[3 * i], a.pos[3 * i + 1], a.pos[3 * i + 2] };
  Vec3 old_point = pos;
  double chi2 = 0.0;
  int it = 0;
  if (o1 - o0 >= 2) {   // "optimizing point with less than two observations": left alone (:255-259)
    for (it = 0; it < a.n_iter; ++it) {
      double A[6] = { 0, 0, 0, 0, 0, 0 };   // packed lower triangle, (r,c) at r(r+1)/2 + c
      double b[3] = { 0, 0, 0 };
      double new_chi2 = 0.0;
      for (int o = o0; o < o1; ++o) {
        const Rigid T = load_rigid(a.T_f_w[a.obs_view[o]]);
        double R[9];
        to_matrix(T.q, R);
        const Vec3 p = transform(T, pos);
        const double fx = a.obs_f[3 * o], fy = a.obs_f[3 * o + 1], fz = a.obs_f[3 * o + 2];
        if (a.on_sphere) {
          // updateHessianGradientUnitSphere (:232-246), jacobian_xyz2f (point.h:187-204)
          const double x2 = p.x * p.x, y2 = p.y * p.y, z2 = p.z * p.z;
          const double xy = p.x * p.y, yz = p.y * p.z, zx = p.z * p.x;
          double Jn[9] = { y2 + z2, -xy, -zx, -xy, x2 + z2, -yz, -zx, -yz, x2 + y2 };
          const double s = 1.0 / pow(x2 + y2 + z2, 1.5);
          double J[9];
#pragma unroll
          for (int k = 0; k < 9; ++k) Jn[k] *= s;
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
              J[r * 3 + c] = (-1.0 * Jn[r * 3 + 0]) * R[0 + c] + (-1.0 * Jn[r * 3 + 1]) * R[3 + c] + (-1.0 * Jn[r * 3 + 2]) * R[6 + c];
          const double nrm = sqrt(x2 + y2 + z2);
          const double e[3] = { fx - p.x / nrm, fy - p.y / nrm, fz - p.z / nrm };
#pragma unroll
          for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c <= r; ++c) A[r * (r + 1) / 2 + c] += J[0 + r] * J[0 + c] + J[3 + r] * J[3 + c] + J[6 + r] * J[6 + c];
            b[r] -= J[0 + r] * e[0] + J[3 + r] * e[1] + J[6 + r] * e[2];
          }
          new_chi2 += e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
        } else {
          // updateHessianGradientUnitPlane (:216-230), jacobian_xyz2uv (point.h:170-184)
          const double z_inv = 1.0 / p.z;
          const double z_inv_sq = z_inv * z_inv;
          const double Jp[6] = { z_inv, 0.0, -p.x * z_inv_sq, 0.0, z_inv, -p.y * z_inv_sq };
          double J[6];
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
              J[r * 3 + c] = (-Jp[r * 3 + 0]) * R[0 + c] + (-Jp[r * 3 + 1]) * R[3 + c] + (-Jp[r * 3 + 2]) * R[6 + c];
          const double e[2] = { fx / fz - p.x / p.z, fy / fz - p.y / p.z };   // vk::project2
#pragma unroll
          for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c <= r; ++c) A[r * (r + 1) / 2 + c] += J[0 + r] * J[0 + c] + J[3 + r] * J[3 + c];
            b[r] -= J[0 + r] * e[0] + J[3 + r] * e[1];
          }
          new_chi2 += e[0] * e[0] + e[1] * e[1];
        }
      }
      (void)ldlt_solve_regs<3>(A, b);   // b <- dp
      if ((it > 0 && new_chi2 > chi2) || b[0] != b[0]) {
        pos = old_point;   // roll-back
        ++it;
        break;
      }
      old_point = pos;
      pos.x += b[0]; pos.y += b[1]; pos.z += b[2];
      chi2 = new_chi2;
      const double nm = fmax(fabs(b[0]), fmax(fabs(b[1]), fabs(b[2])));
      if (nm <= 0.0000000001) { ++it; break; }
    }
  }
  a.pos[3 * i] = pos.x; a.pos[3 * i + 1] = pos.y; a.pos[3 * i + 2] = pos.z;
  if (a.iters) a.iters[i] = it;
}

}  // namespace svoh

using namespace svoh;

// packed != NULL: the per-feature arrays are the caller's DEVICE arrays (concatenated in problem, then camera order),
// used in place; only the descriptors travel.
static int run_pose_batch(svoh_ctx* ctx, const svoh_pose_options* options, int n_problems, const svoh_pose_problem* problems,
                          const svoh_pose_packed_arrays* packed, svoh_pose_result* results,
                          void (*after_launch)(void*) = nullptr, void* user = nullptr)
{
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, options && n_problems >= 0, "bad arguments");
  if (n_problems == 0) return SVOH_OK;
  SVOH_REQUIRE(ctx, problems && results, "NULL argument");
  SVOH_REQUIRE(ctx, options->max_iter >= 1 && options->error_type >= 0 && options->error_type <= 2, "bad max_iter / error_type");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));

  size_t n_cams_total = 0, n_feat_total = 0, max_meas = 0;
  for (int p = 0; p < n_problems; ++p) {
    const svoh_pose_problem& pb = problems[p];
    SVOH_REQUIRE(ctx, pb.n_cams >= 1 && pb.n_cams <= SVOH_MAX_CAMS, "n_cams out of range");
    size_t n = 0;
    for (int c = 0; c < pb.n_cams; ++c) {
      const svoh_pose_camera& cam = pb.cams[c];
      SVOH_REQUIRE(ctx, cam.n_features >= 0, "negative n_features");
      SVOH_REQUIRE(ctx, packed || cam.n_features == 0 ||
                            (cam.px && cam.f && cam.grad && cam.level && cam.type && cam.xyz_world && cam.usable),
                   "NULL feature array");
      SVOH_REQUIRE(ctx, cam.cam.distortion == SVOH_DISTORTION_NONE || cam.cam.distortion == SVOH_DISTORTION_RADTAN,
                   "unsupported distortion model");
      if (!packed)   // device-resident levels are clamped to 0..29 by the kernel instead (pose_level)
        for (int i = 0; i < cam.n_features; ++i) SVOH_REQUIRE(ctx, cam.level[i] >= 0 && cam.level[i] < 30, "feature level out of range");
      n += (size_t)cam.n_features;
    }
    SVOH_REQUIRE(ctx, n <= (size_t)kPoseMaxMeas, "more than 4096 features in one bundle");
    n_cams_total += (size_t)pb.n_cams;
    n_feat_total += n;
    max_meas = std::max(max_meas, n);
  }
  if (packed) {
    SVOH_REQUIRE(ctx, (size_t)packed->n_features_total == n_feat_total, "n_features_total does not match the problems");
    SVOH_REQUIRE(ctx, n_feat_total == 0 || (packed->px && packed->f && packed->grad && packed->level && packed->type &&
                                            packed->xyz_world && packed->usable && packed->outlier && packed->final_error),
                 "NULL packed array");
  }
  const size_t nf = packed ? 1 : (n_feat_total ? n_feat_total : 1);
  // one staging block: [problems | cams | px | f | grad | xyz | level | type | usable]  -> device; outputs appended
  auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
  const size_t o_pb = 0, o_cam = al(sizeof(DevPoseProblem) * (size_t)n_problems), o_px = o_cam + al(sizeof(DevPoseCam) * n_cams_total);
  const size_t o_f = o_px + al(16 * nf), o_grad = o_f + al(24 * nf), o_xyz = o_grad + al(16 * nf), o_level = o_xyz + al(24 * nf);
  const size_t o_type = o_level + al(4 * nf), o_usable = o_type + al(nf), in_total = o_usable + al(nf);
  const size_t o_outlier = in_total, o_ferr = o_outlier + al(nf), o_res = o_ferr + al(8 * nf);
  const size_t total = o_res + al(sizeof(svoh_pose_result) * (size_t)n_problems);
  SVOH_HIP_TRY(ctx, ctx->h_scratch0.reserve(total));
  SVOH_HIP_TRY(ctx, ctx->d_scratch0.reserve(total));
  uint8_t* h = static_cast<uint8_t*>(ctx->h_scratch0.ptr);
  uint8_t* d = static_cast<uint8_t*>(ctx->d_scratch0.ptr);
  DevPoseProblem* hp = reinterpret_cast<DevPoseProblem*>(h + o_pb);
  DevPoseCam* hc = reinterpret_cast<DevPoseCam*>(h + o_cam);
  size_t cam_i = 0, off = 0;
  for (int p = 0; p < n_problems; ++p) {   // descriptors and offsets (serial: each depends on the one before)
    const svoh_pose_problem& pb = problems[p];
    hp[p].n_cams = pb.n_cams; hp[p].cam_begin = (int)cam_i; hp[p].arr_off = (long long)off; hp[p].T_imu_world = pb.T_imu_world;
    int local = 0;
    for (int c = 0; c < pb.n_cams; ++c) {
      const svoh_pose_camera& cam = pb.cams[c];
      hc[cam_i].cam = cam.cam; hc[cam_i].T_cam_imu = cam.T_cam_imu; hc[cam_i].n_features = cam.n_features; hc[cam_i].feat_off = local;
      local += cam.n_features;
      ++cam_i;
    }
    hp[p].n_total = local;
    off += (size_t)local;
  }
  // the feature arrays (86 bytes per feature) into the pinned block; a large batch is copied by a few threads
  auto stage_range = [&](int p0, int p1) {
    for (int p = p0; p < p1; ++p) {
      const svoh_pose_problem& pb = problems[p];
      for (int c = 0; c < pb.n_cams; ++c) {
        const svoh_pose_camera& cam = pb.cams[c];
        const size_t n = (size_t)cam.n_features, g = (size_t)hp[p].arr_off + (size_t)hc[hp[p].cam_begin + c].feat_off;
        if (!n) continue;
        memcpy(h + o_px + 16 * g, cam.px, 16 * n); memcpy(h + o_f + 24 * g, cam.f, 24 * n);
        memcpy(h + o_grad + 16 * g, cam.grad, 16 * n); memcpy(h + o_xyz + 24 * g, cam.xyz_world, 24 * n);
        memcpy(h + o_level + 4 * g, cam.level, 4 * n); memcpy(h + o_type + g, cam.type, n); memcpy(h + o_usable + g, cam.usable, n);
      }
    }
  };
  const int n_stage_threads = packed ? 0 : (n_feat_total >= kPoseParallelStagingFeatures
      ? (int)std::min<size_t>({ (size_t)8, (size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)n_problems }) : 1);
  if (n_stage_threads == 0) {
    // nothing to stage
  } else if (n_stage_threads <= 1) {
    stage_range(0, n_problems);
  } else {
    // a thread that cannot be started (pid limit of a container) must not take the call down: its range, and every
    // later one, is copied by this thread, and whatever was started is joined before anything can be thrown past it
    std::vector<std::thread> workers;
    workers.reserve((size_t)n_stage_threads);
    int first_serial = n_stage_threads;
    for (int t = 1; t < n_stage_threads; ++t) {
      try {
        workers.emplace_back(stage_range, (int)((long long)n_problems * t / n_stage_threads), (int)((long long)n_problems * (t + 1) / n_stage_threads));
      } catch (...) { first_serial = t; break; }
    }
    stage_range(0, n_problems / n_stage_threads);
    for (int t = first_serial; t < n_stage_threads; ++t)
      stage_range((int)((long long)n_problems * t / n_stage_threads), (int)((long long)n_problems * (t + 1) / n_stage_threads));
    for (auto& w : workers) w.join();
  }
  SVOH_HIP_TRY(ctx, svoh_copy_to_device(ctx, d, h, in_total));
  PoseArgs a;
  a.opt = *options;
  a.problems = reinterpret_cast<const DevPoseProblem*>(d + o_pb);
  a.cams = reinterpret_cast<const DevPoseCam*>(d + o_cam);
  a.px = reinterpret_cast<const double*>(d + o_px); a.f = reinterpret_cast<const double*>(d + o_f);
  a.grad = reinterpret_cast<const double*>(d + o_grad); a.xyz = reinterpret_cast<const double*>(d + o_xyz);
  a.level = reinterpret_cast<const int32_t*>(d + o_level); a.type = d + o_type; a.usable = d + o_usable;
  a.outlier = d + o_outlier; a.final_error = reinterpret_cast<double*>(d + o_ferr);
  if (packed) {
    a.px = packed->px; a.f = packed->f; a.grad = packed->grad; a.xyz = packed->xyz_world; a.level = packed->level;
    a.type = packed->type; a.usable = packed->usable; a.outlier = packed->outlier; a.final_error = packed->final_error;
  }
  a.results = reinterpret_cast<svoh_pose_result*>(d + o_res);
  a.n_problems = n_problems;
  a.max_meas = (int)((max_meas + 63) & ~(size_t)63);
  if (a.max_meas == 0) a.max_meas = 64;
  const size_t err_bytes = sizeof(double) * (size_t)a.max_meas;
  if (ctx->timing_on()) SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_start, ctx->stream));
  // geometry: see pose_optimize_kernel.  SVOH_POSE_THREADS=64/256 forces one (tests run both).
  int nt = n_problems > ctx->num_cus ? 64 : kPoseThreads;
  // a few large bundles: one feature per lane (measured, scripts/perf_pose_scaling.py: 2 x 160 features 0.064 ms with 256 threads)
  if (n_problems * 2 <= ctx->num_cus && (int)max_meas > kPoseThreads && options->error_type != SVOH_POSE_ERR_BEARING_DIFF) nt = 512;
  { const int v = ctx->knobs.pose_threads; if (v == 64 || v == kPoseThreads || (v == 512 && options->error_type != SVOH_POSE_ERR_BEARING_DIFF)) nt = v; }
  auto launch = [&](auto et) {
    constexpr int ET = decltype(et)::value;
    if (nt == 64) hipLaunchKernelGGL((pose_optimize_kernel<64, ET>), dim3((unsigned)n_problems), dim3(64), err_bytes, ctx->stream, a);
    else if (nt == 512) {
      if constexpr (ET != SVOH_POSE_ERR_BEARING_DIFF) hipLaunchKernelGGL((pose_optimize_kernel<512, ET>), dim3((unsigned)n_problems), dim3(512), err_bytes, ctx->stream, a);
    }
    else hipLaunchKernelGGL((pose_optimize_kernel<kPoseThreads, ET>), dim3((unsigned)n_problems), dim3(kPoseThreads), err_bytes, ctx->stream, a);
  };
  if (options->error_type == SVOH_POSE_ERR_UNIT_PLANE) launch(std::integral_constant<int, SVOH_POSE_ERR_UNIT_PLANE>{});
  else if (options->error_type == SVOH_POSE_ERR_IMAGE_PLANE) launch(std::integral_constant<int, SVOH_POSE_ERR_IMAGE_PLANE>{});
  else launch(std::integral_constant<int, SVOH_POSE_ERR_BEARING_DIFF>{});
  SVOH_HIP_TRY(ctx, hipGetLastError());
  if (ctx->timing_on()) SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_stop, ctx->stream));
  ctx->misc_timed = ctx->timing_on(); ctx->misc_launched = true;
  if (packed) {   // per-feature outputs stay on the device; the per-bundle results come back
    SVOH_HIP_TRY(ctx, svoh_copy_to_host(ctx, h + o_res, d + o_res, total - o_res));
    SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(results, h + o_res, sizeof(svoh_pose_result) * (size_t)n_problems);
    return SVOH_OK;
  }
  SVOH_HIP_TRY(ctx, svoh_copy_to_host(ctx, h + o_outlier, d + o_outlier, total - o_outlier));
  if (after_launch) {
    // the hook's work goes behind an event and is not waited for: this call's staging (h, d) is not touched by it --
    // the hook must not enter a blocking call of the context, and the depth filter's staging has blocks of its own
    if (!ctx->ev_pose_done) SVOH_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_pose_done, hipEventDisableTiming));
    SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_pose_done, ctx->stream));
    // (what the hook queues may read this batch's results in place: svoh_frame_view::pose_result_index_plus1)
    ctx->in_pose_hook = true; ctx->d_pose_results = d + o_res; ctx->n_pose_results = n_problems;
    after_launch(user);
    ctx->in_pose_hook = false; ctx->d_pose_results = nullptr; ctx->n_pose_results = 0;
    SVOH_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_pose_done));
  } else {
    SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  memcpy(results, h + o_res, sizeof(svoh_pose_result) * (size_t)n_problems);
  off = 0;
  for (int p = 0; p < n_problems; ++p)
    for (int c = 0; c < problems[p].n_cams; ++c) {
      const svoh_pose_camera& cam = problems[p].cams[c];
      const size_t n = (size_t)cam.n_features;
      if (cam.outlier && n) memcpy(cam.outlier, h + o_outlier + off, n);
      if (cam.final_error && n) memcpy(cam.final_error, h + o_ferr + 8 * off, 8 * n);
      off += n;
    }
  return SVOH_OK;
}

extern "C" int svoh_optimize_pose_batch(svoh_ctx* ctx, const svoh_pose_options* options, int n_problems,
                                        const svoh_pose_problem* problems, svoh_pose_result* results)
try {
  return run_pose_batch(ctx, options, n_problems, problems, nullptr, results);
} SVOH_ABI_CATCH(ctx)

extern "C" int svoh_optimize_pose_batch_hook(svoh_ctx* ctx, const svoh_pose_options* options, int n_problems,
                                             const svoh_pose_problem* problems, svoh_pose_result* results,
                                             void (*after_launch)(void* user), void* user)
try {
  return run_pose_batch(ctx, options, n_problems, problems, nullptr, results, after_launch, user);
} SVOH_ABI_CATCH(ctx)

extern "C" int svoh_optimize_pose_batch_packed(svoh_ctx* ctx, const svoh_pose_options* options, int n_problems,
                                               const svoh_pose_problem* problems, const svoh_pose_packed_arrays* arrays,
                                               svoh_pose_result* results)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, arrays != nullptr, "NULL argument");
  return run_pose_batch(ctx, options, n_problems, problems, arrays, results);
} SVOH_ABI_CATCH(ctx)

static int run_points_batch(svoh_ctx* ctx, bool queued, int n_iter, int using_bearing_vector, int n_views,
                            const svoh_se3* T_f_w, int n_points, const int32_t* obs_begin,
                            const int32_t* obs_view, const double* obs_f, double* pos, int32_t* iters)
{
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, n_points >= 0 && n_views >= 0 && n_iter >= 0, "negative count");
  if (n_points == 0) return SVOH_OK;
  SVOH_REQUIRE(ctx, obs_begin && pos, "NULL argument");
  SVOH_REQUIRE(ctx, obs_begin[0] == 0, "obs_begin must start at 0");
  for (int i = 0; i < n_points; ++i) SVOH_REQUIRE(ctx, obs_begin[i + 1] >= obs_begin[i], "obs_begin must not decrease");
  const size_t n_obs = (size_t)obs_begin[n_points];
  SVOH_REQUIRE(ctx, n_obs == 0 || (obs_view && obs_f && T_f_w), "NULL observation array");
  for (size_t o = 0; o < n_obs; ++o) SVOH_REQUIRE(ctx, obs_view[o] >= 0 && obs_view[o] < n_views, "observation refers to an unknown view");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
  const size_t o_T = 0, o_begin = al(sizeof(svoh_se3) * (size_t)(n_views ? n_views : 1));
  const size_t o_view = o_begin + al(4 * ((size_t)n_points + 1)), o_f = o_view + al(4 * (n_obs ? n_obs : 1));
  const size_t o_pos = o_f + al(24 * (n_obs ? n_obs : 1)), o_it = o_pos + al(24 * (size_t)n_points);
  const size_t total = o_it + al(4 * (size_t)n_points);
  // queued: buffers of its own (the blocking calls' scratch blocks stay free for whoever calls in between) and an event behind the copy back
  hipStream_t stream = ctx->stream;
  SVOH_REQUIRE(ctx, !queued || ctx->points_pending == 0, "a points batch is queued: svoh_optimize_points_batch_collect first");
  PinnedBuffer& hb = queued ? ctx->h_points_q : ctx->h_scratch0;
  DevBuffer& db = queued ? ctx->d_points_q : ctx->d_scratch0;
  SVOH_HIP_TRY(ctx, hb.reserve(total));
  SVOH_HIP_TRY(ctx, db.reserve(total));
  uint8_t* h = static_cast<uint8_t*>(hb.ptr);
  uint8_t* d = static_cast<uint8_t*>(db.ptr);
  if (n_views) memcpy(h + o_T, T_f_w, sizeof(svoh_se3) * (size_t)n_views);
  memcpy(h + o_begin, obs_begin, 4 * ((size_t)n_points + 1));
  if (n_obs) { memcpy(h + o_view, obs_view, 4 * n_obs); memcpy(h + o_f, obs_f, 24 * n_obs); }
  memcpy(h + o_pos, pos, 24 * (size_t)n_points);
  SVOH_HIP_TRY(ctx, svoh_copy_to_device(ctx, d, h, o_it));
  PointArgs a;
  a.T_f_w = reinterpret_cast<const svoh_se3*>(d + o_T);
  a.obs_begin = reinterpret_cast<const int32_t*>(d + o_begin);
  a.obs_view = reinterpret_cast<const int32_t*>(d + o_view);
  a.obs_f = reinterpret_cast<const double*>(d + o_f);
  a.pos = reinterpret_cast<double*>(d + o_pos);
  a.iters = reinterpret_cast<int32_t*>(d + o_it);
  a.n_points = n_points; a.n_iter = n_iter; a.on_sphere = using_bearing_vector != 0;
  const bool timed = !queued && ctx->timing_on();
  if (timed) SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_start, stream));
  hipLaunchKernelGGL(point_optimize_kernel, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, stream, a);
  SVOH_HIP_TRY(ctx, hipGetLastError());
  if (timed) SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_misc_stop, stream));
  if (!queued) { ctx->misc_timed = timed; ctx->misc_launched = true; }
  SVOH_HIP_TRY(ctx, svoh_copy_to_host(ctx, h + o_pos, d + o_pos, total - o_pos));
  if (queued) {
    if (!ctx->ev_points) SVOH_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_points, hipEventDisableTiming));
    SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_points, stream));
    ctx->points_pending = n_points; ctx->points_o_pos = o_pos; ctx->points_o_it = o_it;
    return SVOH_OK;
  }
  SVOH_HIP_TRY(ctx, hipStreamSynchronize(stream));
  memcpy(pos, h + o_pos, 24 * (size_t)n_points);
  if (iters) memcpy(iters, h + o_it, 4 * (size_t)n_points);
  return SVOH_OK;
}

extern "C" int svoh_optimize_points_batch(svoh_ctx* ctx, int n_iter, int using_bearing_vector, int n_views,
                                          const svoh_se3* T_f_w, int n_points, const int32_t* obs_begin,
                                          const int32_t* obs_view, const double* obs_f, double* pos, int32_t* iters)
try {
  return run_points_batch(ctx, false, n_iter, using_bearing_vector, n_views, T_f_w, n_points, obs_begin, obs_view, obs_f, pos, iters);
} SVOH_ABI_CATCH(ctx)

extern "C" int svoh_optimize_points_batch_enqueue(svoh_ctx* ctx, int n_iter, int using_bearing_vector, int n_views,
                                                  const svoh_se3* T_f_w, int n_points, const int32_t* obs_begin,
                                                  const int32_t* obs_view, const double* obs_f, const double* pos)
try {
  return run_points_batch(ctx, true, n_iter, using_bearing_vector, n_views, T_f_w, n_points, obs_begin, obs_view, obs_f, const_cast<double*>(pos), nullptr);
} SVOH_ABI_CATCH(ctx)

extern "C" int svoh_optimize_points_batch_collect(svoh_ctx* ctx, int n_points, double* pos, int32_t* iters)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, n_points >= 0 && n_points == ctx->points_pending && (n_points == 0 || pos), "n_points is not the size of the queued batch");
  if (n_points == 0) return SVOH_OK;
  ctx->points_pending = 0;
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  SVOH_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_points));   // THAT batch: not what the caller has queued on the context since
  const uint8_t* h = static_cast<const uint8_t*>(ctx->h_points_q.ptr);
  memcpy(pos, h + ctx->points_o_pos, 24 * (size_t)n_points);
  if (iters) memcpy(iters, h + ctx->points_o_it, 4 * (size_t)n_points);
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)
