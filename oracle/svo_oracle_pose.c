/* svo_oracle_pose.c -- CPU restatement of PoseOptimizer::run.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 * PARITY UNPINNED, except for one property the reference itself asserts: the Jacobians agree with finite
 * differences (src/svo/test/test_frame.cpp:131-157, tol 1e-6 / 1e-5), repeated in tests/test_oracle_pose_cpu.py.
 *
 * Follows
 *   PoseOptimizer::run / evaluateErrorImpl / removeOutliers / update / applyPrior
 *                                      src/svo/src/pose_optimizer.cpp:39-334
 *   pose_optimizer_utils::calculate{Feature,Edgelet}Residual{UnitPlane,ImagePlane,BearingVectorDiff}   :338-627
 *   Frame::jacobian_xyz2uv_imu / xyz2img_imu / xyz2f_imu      src/svo_common/include/svo/common/frame.h:342-397
 *   MiniLeastSquaresSolver::optimizeGaussNewton               vikit/solver/implementation/mini_least_squares_solver.hpp:42-107
 *   MADScaleEstimator::compute, TukeyWeightFunction::weight   src/vikit/vikit_solver/src/robust_cost.cpp:21-60
 * The state is T_imu_world; update is T_new = exp(dx) * T_old with minkindr's decoupled exp, then the
 * quaternion is normalised (:309-318).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "svo_oracle.h"

static float tukey_weight_f(float error)
{
  const float b_square = 4.6851f * 4.6851f;
  const float x_square = error * error;
  if (x_square <= b_square) { const float tmp = 1.0f - x_square / b_square; return tmp * tmp; }
  return 0.0f;
}

static void mat3_of_se3(const svoh_se3* T, double R[9])
{
  const double w = T->q[0], x = T->q[1], y = T->q[2], z = T->q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}

/* J (rows x 6) = A (rows x 3) * R_cam_imu * [I | -skew(p_in_imu)] */
static void chain_G(const double* A, int rows, const double R[9], const double p[3], double* J)
{
  double AR[9];
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < 3; ++c) AR[r * 3 + c] = A[r * 3 + 0] * R[0 + c] + A[r * 3 + 1] * R[3 + c] + A[r * 3 + 2] * R[6 + c];
  /* -skew(p) = [[0, pz, -py], [-pz, 0, px], [py, -px, 0]] */
  const double S[9] = { 0, p[2], -p[1], -p[2], 0, p[0], p[1], -p[0], 0 };
  for (int r = 0; r < rows; ++r) {
    for (int c = 0; c < 3; ++c) J[r * 6 + c] = AR[r * 3 + c];
    for (int c = 0; c < 3; ++c) J[r * 6 + 3 + c] = AR[r * 3 + 0] * S[0 + c] + AR[r * 3 + 1] * S[3 + c] + AR[r * 3 + 2] * S[6 + c];
  }
}

void orc_jacobian_xyz2uv_imu(const svoh_se3* T_cam_imu, const double p_in_imu[3], double J[12])
{
  double R[9], pc[3];
  mat3_of_se3(T_cam_imu, R);
  orc_se3_transform(T_cam_imu, p_in_imu, pc);
  const double s = -1.0 / pc[2];
  const double A[6] = { s * 1.0, 0.0, s * (-pc[0] / pc[2]), 0.0, s * 1.0, s * (-pc[1] / pc[2]) };
  chain_G(A, 2, R, p_in_imu, J);
}

void orc_jacobian_xyz2img_imu(const svoh_se3* T_cam_imu, const double p_in_imu[3], const double J_cam[6], double J[12])
{
  double R[9];
  mat3_of_se3(T_cam_imu, R);
  chain_G(J_cam, 2, R, p_in_imu, J);
}

void orc_jacobian_xyz2f_imu(const svoh_se3* T_cam_imu, const double p_in_imu[3], double J[18])
{
  double R[9], pc[3];
  mat3_of_se3(T_cam_imu, R);
  orc_se3_transform(T_cam_imu, p_in_imu, pc);
  const double x2 = pc[0] * pc[0], y2 = pc[1] * pc[1], z2 = pc[2] * pc[2];
  const double xy = pc[0] * pc[1], yz = pc[1] * pc[2], zx = pc[2] * pc[0];
  const double k = 1 / pow(x2 + y2 + z2, 1.5);
  const double A[9] = { k * (y2 + z2), k * -xy, k * -zx, k * -xy, k * (x2 + z2), k * -yz, k * -zx, k * -yz, k * (x2 + y2) };
  chain_G(A, 3, R, p_in_imu, J);
}

/* one measurement: unwhitened error, chi2, and (if H) H += J^T J w, g -= J^T e w; returns 0 */
static void residual(const svoh_pose_options* opt, const svoh_pose_camera* cam, int i, const svoh_se3* T_imu_world,
                     double measurement_sigma, double* unwhitened_error, double* chi2_error, double* H, double* g)
{
  const int edgelet = cam->type[i] == SVOH_FT_EDGELET || cam->type[i] == SVOH_FT_EDGELET_SEED ||
                      cam->type[i] == SVOH_FT_EDGELET_SEED_CONVERGED;
  double p_imu[3], p_cam[3];
  orc_se3_transform(T_imu_world, &cam->xyz_world[3 * i], p_imu);
  orc_se3_transform(&cam->T_cam_imu, p_imu, p_cam);
  const double* f = &cam->f[3 * i];
  const double* px = &cam->px[2 * i];
  const double* grad = &cam->grad[2 * i];
  const double R = 1.0 / measurement_sigma;
  double e[3] = { 0, 0, 0 };
  int rows = 0;
  double J[18];
  const int want_J = H != NULL;
  if (opt->error_type == SVOH_POSE_ERR_UNIT_PLANE) {
    const double d0 = f[0] / f[2] - p_cam[0] / p_cam[2], d1 = f[1] / f[2] - p_cam[1] / p_cam[2];
    double Juv[12];
    if (want_J) orc_jacobian_xyz2uv_imu(&cam->T_cam_imu, p_imu, Juv);
    if (!edgelet) {
      rows = 2; e[0] = d0; e[1] = d1;
      if (want_J) memcpy(J, Juv, sizeof Juv);
    } else {
      rows = 1; e[0] = grad[0] * d0 + grad[1] * d1;
      if (want_J) for (int c = 0; c < 6; ++c) J[c] = grad[0] * Juv[c] + grad[1] * Juv[6 + c];
    }
  } else if (opt->error_type == SVOH_POSE_ERR_IMAGE_PLANE) {
    double px_est[2], J_cam[6], Jimg[12];
    orc_project3(&cam->cam, p_cam, px_est, J_cam);
    const double d0 = px[0] - px_est[0], d1 = px[1] - px_est[1];
    if (want_J) { orc_jacobian_xyz2img_imu(&cam->T_cam_imu, p_imu, J_cam, Jimg); for (int c = 0; c < 12; ++c) Jimg[c] = (-1.0) * Jimg[c]; }
    if (!edgelet) {
      rows = 2; e[0] = d0; e[1] = d1;
      if (want_J) memcpy(J, Jimg, sizeof Jimg);
    } else {
      rows = 1; e[0] = grad[0] * d0 + grad[1] * d1;
      if (want_J) for (int c = 0; c < 6; ++c) J[c] = grad[0] * Jimg[c] + grad[1] * Jimg[6 + c];
    }
  } else {
    const double nrm = sqrt(p_cam[0] * p_cam[0] + p_cam[1] * p_cam[1] + p_cam[2] * p_cam[2]);
    const double fd[3] = { f[0] - p_cam[0] / nrm, f[1] - p_cam[1] / nrm, f[2] - p_cam[2] / nrm };
    double Jb[18];
    if (want_J) orc_jacobian_xyz2f_imu(&cam->T_cam_imu, p_imu, Jb);
    if (!edgelet) {
      rows = 3; e[0] = fd[0]; e[1] = fd[1]; e[2] = fd[2];
      if (want_J) for (int c = 0; c < 18; ++c) J[c] = (-1.0) * Jb[c];
    } else {
      /* the image-plane edgelet residual scaled to the unit sphere (:565-627) */
      double px_est[2], J_cam[6];
      orc_project3(&cam->cam, p_cam, px_est, J_cam);
      const double pd[2] = { px[0] - px_est[0], px[1] - px_est[1] };
      const double pd2 = pd[0] * pd[0] + pd[1] * pd[1];
      const double fd2 = fd[0] * fd[0] + fd[1] * fd[1] + fd[2] * fd[2];
      const double e_img = grad[0] * pd[0] + grad[1] * pd[1];
      const double scale_ratio = sqrt(fd2) / sqrt(pd2);
      rows = 1; e[0] = e_img * scale_ratio;
      if (want_J) {
        double Jp[12];
        orc_jacobian_xyz2img_imu(&cam->T_cam_imu, p_imu, J_cam, Jp);
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
  double en2 = 0.0;
  for (int r = 0; r < rows; ++r) en2 += e[r] * e[r];
  *unwhitened_error = rows == 1 ? fabs(e[0]) : sqrt(en2);
  for (int r = 0; r < rows; ++r) e[r] *= R;
  /* robust_weight.weight(const float&): the double argument narrows to float */
  const double en = rows == 1 ? e[0] : sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
  const double weight = (double)tukey_weight_f((float)en);
  double es2 = 0.0;
  for (int r = 0; r < rows; ++r) es2 += e[r] * e[r];
  *chi2_error = 0.5 * es2 * weight;
  if (want_J) {
    for (int c = 0; c < rows * 6; ++c) J[c] *= R;
    for (int a = 0; a < 6; ++a) {
      for (int b = 0; b < 6; ++b) {
        double s = 0.0;
        for (int r = 0; r < rows; ++r) s += J[r * 6 + a] * J[r * 6 + b];
        H[a * 6 + b] += s * weight;
      }
      double s = 0.0;
      for (int r = 0; r < rows; ++r) s += J[r * 6 + a] * e[r];
      g[a] -= s * weight;
    }
  }
}

static int cmp_float(const void* a, const void* b)
{
  const float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}
static int cmp_double(const void* a, const void* b)
{
  const double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}

/* evaluateErrorImpl (:115-196): chi2 sum; errors (may be NULL) receives unwhitened_error / 2^level per measurement */
static double evaluate(const svoh_pose_options* opt, const svoh_pose_problem* pb, const svoh_se3* T_imu_world,
                       double measurement_sigma_base, double* H, double* g, float* errors, int* n_meas)
{
  double chi2_sum = 0.0;
  int n = 0;
  for (int c = 0; c < pb->n_cams; ++c) {
    const svoh_pose_camera* cam = &pb->cams[c];
    for (int i = 0; i < cam->n_features; ++i) {
      if (!cam->usable[i]) continue;
      const int scale = 1 << cam->level[i];
      double sigma = measurement_sigma_base * scale;
      const int edgelet = cam->type[i] == SVOH_FT_EDGELET || cam->type[i] == SVOH_FT_EDGELET_SEED ||
                          cam->type[i] == SVOH_FT_EDGELET_SEED_CONVERGED;
      if (edgelet) sigma *= 2.0;  /* kEdgeletSigmaExtraFactor */
      double ue, chi2;
      residual(opt, cam, i, T_imu_world, sigma, &ue, &chi2, H, g);
      if (errors) errors[n] = (float)(ue / scale);
      chi2_sum += chi2;
      ++n;
    }
  }
  *n_meas = n;
  return chi2_sum;
}

/* PoseOptimizer::run (:39-113).  outlier (per camera, n_features bytes) and final_error (may be NULL) are outputs. */
void orc_optimize_pose(const svoh_pose_options* opt, const svoh_pose_problem* pb, svoh_pose_result* res)
{
  int n_total = 0;
  for (int c = 0; c < pb->n_cams; ++c) n_total += pb->cams[c].n_features;
  float* start_errors = (float*)malloc(sizeof(float) * (size_t)(n_total > 0 ? n_total : 1));
  svoh_se3 T = pb->T_imu_world;
  memset(res, 0, sizeof *res);
  int n_meas = 0;
  evaluate(opt, pb, &T, 1.0, NULL, NULL, start_errors, &n_meas);   /* measurement_sigma_ is 1.0 here (fresh optimizer) */
  res->n_meas = n_meas;
  if (n_meas == 0) { res->T_imu_world = T; res->status = 1; free(start_errors); return; }
  float* tmp = (float*)malloc(sizeof(float) * (size_t)n_meas);
  memcpy(tmp, start_errors, sizeof(float) * (size_t)n_meas);
  qsort(tmp, (size_t)n_meas, sizeof(float), cmp_float);
  const float med = tmp[n_meas / 2];                   /* nth_element at floor(n/2) */
  const double measurement_sigma = (double)(1.48f * med);
  res->measurement_sigma = measurement_sigma;
  res->reproj_error_before = (double)med;
  free(tmp);

  /* optimizeGaussNewton */
  svoh_se3 old_T = T;
  double I_prior = 0.0;
  int stop = 0;
  for (int iter = 0; iter < opt->max_iter; ++iter) {
    double H[36], g[6], dx[8];
    memset(H, 0, sizeof H); memset(g, 0, sizeof g);
    evaluate(opt, pb, &T, measurement_sigma, H, g, NULL, &n_meas);
    res->iters = iter + 1;
    if (opt->have_rotation_prior) {   /* applyPrior (:320-334) */
      if (iter == 0) {
        double hmax = 0;
        for (int j = 3; j < 6; ++j) hmax = fmax(hmax, fabs(H[j * 6 + j]));
        I_prior = hmax * opt->prior_lambda;
      }
      for (int j = 3; j < 6; ++j) H[j * 6 + j] += I_prior;
      svoh_se3 prior, prior_inv, d;
      prior.q[0] = opt->R_prior[0]; prior.q[1] = opt->R_prior[1]; prior.q[2] = opt->R_prior[2]; prior.q[3] = opt->R_prior[3];
      prior.t[0] = prior.t[1] = prior.t[2] = 0.0;
      orc_se3_inverse(&prior, &prior_inv);
      orc_se3_mul(&T, &prior_inv, &d);
      double lg[6];
      orc_se3_log(&d, lg);
      for (int j = 3; j < 6; ++j) g[j] -= I_prior * lg[j];
    }
    if (!orc_ldlt_solve(6, H, g, dx)) stop = 1;   /* H symmetric: row-major == col-major */
    if (stop) { T = old_T; res->status = 2; break; }
    svoh_se3 ex, Tn;
    orc_se3_exp(dx, &ex);
    orc_se3_mul(&ex, &T, &Tn);                  /* T_new = exp(dx) * T_old */
    {
      const double nn = sqrt(Tn.q[0] * Tn.q[0] + Tn.q[1] * Tn.q[1] + Tn.q[2] * Tn.q[2] + Tn.q[3] * Tn.q[3]);
      for (int k = 0; k < 4; ++k) Tn.q[k] /= nn;
    }
    old_T = T;
    T = Tn;
    double x_norm = -1.0;
    for (int j = 0; j < 6; ++j) if (fabs(dx[j]) > x_norm) x_norm = fabs(dx[j]);
    if (x_norm < opt->eps) break;
  }
  res->T_imu_world = T;
  res->n_meas = n_meas;

  /* removeOutliers (:198-307) on every frame with measurement_sigma 0 -> only the unwhitened error is used */
  double* final_errors = (double*)malloc(sizeof(double) * (size_t)(n_total > 0 ? n_total : 1));
  int nf = 0;
  for (int c = 0; c < pb->n_cams; ++c) {
    const svoh_pose_camera* cam = &pb->cams[c];
    for (int i = 0; i < cam->n_features; ++i) {
      if (cam->outlier) cam->outlier[i] = 0;
      if (cam->final_error) cam->final_error[i] = 0.0;
      if (!cam->usable[i]) continue;
      double ue, chi2;
      residual(opt, cam, i, &T, 0.0, &ue, &chi2, NULL, NULL);
      ue *= 1.0 / (1 << cam->level[i]);
      final_errors[nf++] = ue;
      if (cam->final_error) cam->final_error[i] = ue;
      if (fabs(ue) > opt->outlier_threshold) {
        const int edgelet = cam->type[i] == SVOH_FT_EDGELET || cam->type[i] == SVOH_FT_EDGELET_SEED ||
                            cam->type[i] == SVOH_FT_EDGELET_SEED_CONVERGED;
        if (edgelet) ++res->n_deleted_edges; else ++res->n_deleted_corners;
        if (cam->outlier) cam->outlier[i] = 1;
      }
    }
  }
  if (nf) { qsort(final_errors, (size_t)nf, sizeof(double), cmp_double); res->reproj_error_after = final_errors[nf / 2]; }
  free(final_errors); free(start_errors);
}

/* ======================================================================== */
/* f-3 (second half)  Point::optimize                                        */
/* ======================================================================== */
/* src/svo_common/src/point.cpp:216-325 with Point::jacobian_xyz2uv / xyz2f
 * (src/svo_common/include/svo/common/point.h:170-204).  3-DoF Gauss-Newton on one landmark over its
 * observations; A.ldlt().solve(b) is the same Eigen LDLT as everywhere else (orc_ldlt_solve, n = 3);
 * stops when the error grows (rolling the last step back), on NaN, or when max|dp| <= 1e-10.
 * Returns the number of iterations started; a point with fewer than two observations is left alone (:255-259). */
int orc_optimize_point(int n_iter, int using_bearing_vector, int n_obs, const svoh_se3* const* T_f_w,
                       const double* f /* 3 per observation */, double pos[3])
{
  double old_point[3] = { pos[0], pos[1], pos[2] };
  double chi2 = 0.0;
  const double eps = 0.0000000001;
  if (n_obs < 2) return 0;
  int i;
  for (i = 0; i < n_iter; ++i) {
    double A[9] = { 0 }, b[3] = { 0 };
    double new_chi2 = 0.0;
    for (int o = 0; o < n_obs; ++o) {
      double R[9], p[3];
      mat3_of_se3(T_f_w[o], R);
      orc_se3_transform(T_f_w[o], pos, p);
      const double* fo = f + 3 * o;
      if (using_bearing_vector) {
        /* updateHessianGradientUnitSphere, point.cpp:232-246 */
        const double x2 = p[0] * p[0], y2 = p[1] * p[1], z2 = p[2] * p[2];
        const double xy = p[0] * p[1], yz = p[1] * p[2], zx = p[2] * p[0];
        double Jn[9] = { y2 + z2, -xy, -zx, -xy, x2 + z2, -yz, -zx, -yz, x2 + y2 };
        const double s = 1.0 / pow(x2 + y2 + z2, 1.5);
        for (int k = 0; k < 9; ++k) Jn[k] *= s;
        double J[9];
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c)
            J[r * 3 + c] = (-1.0 * Jn[r * 3 + 0]) * R[0 + c] + (-1.0 * Jn[r * 3 + 1]) * R[3 + c] + (-1.0 * Jn[r * 3 + 2]) * R[6 + c];
        const double nrm = sqrt(x2 + y2 + z2);
        const double e[3] = { fo[0] - p[0] / nrm, fo[1] - p[1] / nrm, fo[2] - p[2] / nrm };
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < 3; ++c) A[c * 3 + r] += J[0 + r] * J[0 + c] + J[3 + r] * J[3 + c] + J[6 + r] * J[6 + c];
          b[r] -= J[0 + r] * e[0] + J[3 + r] * e[1] + J[6 + r] * e[2];
        }
        new_chi2 += e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
      } else {
        /* updateHessianGradientUnitPlane, point.cpp:216-230 */
        const double z_inv = 1.0 / p[2];
        const double z_inv_sq = z_inv * z_inv;
        const double Jp[6] = { z_inv, 0.0, -p[0] * z_inv_sq, 0.0, z_inv, -p[1] * z_inv_sq };
        double J[6];
        for (int r = 0; r < 2; ++r)
          for (int c = 0; c < 3; ++c)
            J[r * 3 + c] = (-Jp[r * 3 + 0]) * R[0 + c] + (-Jp[r * 3 + 1]) * R[3 + c] + (-Jp[r * 3 + 2]) * R[6 + c];
        const double e[2] = { fo[0] / fo[2] - p[0] / p[2], fo[1] / fo[2] - p[1] / p[2] };   /* vk::project2 */
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < 3; ++c) A[c * 3 + r] += J[0 + r] * J[0 + c] + J[3 + r] * J[3 + c];
          b[r] -= J[0 + r] * e[0] + J[3 + r] * e[1];
        }
        new_chi2 += e[0] * e[0] + e[1] * e[1];
      }
    }
    double dp[3];
    orc_ldlt_solve(3, A, b, dp);
    if ((i > 0 && new_chi2 > chi2) || dp[0] != dp[0]) {
      pos[0] = old_point[0]; pos[1] = old_point[1]; pos[2] = old_point[2];   /* roll-back */
      return i + 1;
    }
    for (int k = 0; k < 3; ++k) { old_point[k] = pos[k]; pos[k] = pos[k] + dp[k]; }
    chi2 = new_chi2;
    double nm = -1.0;
    for (int k = 0; k < 3; ++k) { const double v = fabs(dp[k]); if (v > nm) nm = v; }
    if (nm <= eps) return i + 1;
  }
  return i;
}
