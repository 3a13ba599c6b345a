// svoh_math.h -- small fixed-size double-precision maths shared by host and
// device code of libsvo_hip: unit quaternions / rigid transforms with the
// semantics of the reference's minkindr types, the pinhole (+radtan) camera,
// and an 8x8 pivoted LDL^T solve with Eigen-3.4 semantics.
//
// Reference behaviour restated here (paths relative to the reference tree):
//   3rd/minkindr/include/kindr/minimal/implementation/rotation-quaternion-inl.h
//       :435-442 operator*, :580-589 normalizationHelper, :478-540 log/exp
//   3rd/minkindr/include/kindr/minimal/implementation/quat-transformation-inl.h
//       :79-84 (ctor from 6-vector), :155-168, :212-238
//   src/vikit/vikit_cameras/include/vikit/cameras/implementation/pinhole_projection.hpp:30-64
//   src/vikit/vikit_cameras/include/vikit/cameras/radial_tangential_distortion.h:46-106
//   src/vikit/vikit_solver/include/vikit/solver/implementation/mini_least_squares_solver.hpp:253-262
#pragma once

#include <math.h>
#include <float.h>
#include "../../include/svo_hip.h"

#if defined(__HIPCC__)
#define SVOH_HD __host__ __device__ __forceinline__
#else
#define SVOH_HD inline
#endif

namespace svoh {

struct Quat { double w, x, y, z; };
struct Vec3 { double x, y, z; };
struct Rigid { Quat q; Vec3 t; };  // p' = q * p + t

SVOH_HD Rigid load_rigid(const svoh_se3& s)
{
  Rigid r;
  r.q.w = s.q[0]; r.q.x = s.q[1]; r.q.y = s.q[2]; r.q.z = s.q[3];
  r.t.x = s.t[0]; r.t.y = s.t[1]; r.t.z = s.t[2];
  return r;
}

SVOH_HD void store_rigid(const Rigid& r, svoh_se3& s)
{
  s.q[0] = r.q.w; s.q[1] = r.q.x; s.q[2] = r.q.y; s.q[3] = r.q.z;
  s.t[0] = r.t.x; s.t[1] = r.t.y; s.t[2] = r.t.z;
}

SVOH_HD double sqnorm(const Quat& q) { return q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; }

SVOH_HD Quat normalized(const Quat& q)
{
  const double n = sqrt(sqnorm(q));
  Quat r = { q.w / n, q.x / n, q.y / n, q.z / n };
  return r;
}

// Hamilton product followed by minkindr's "renormalise only when |n^2-1| > 1e-4"
SVOH_HD Quat mul(const Quat& a, const Quat& b)
{
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  if (fabs(sqnorm(r) - 1.0) > 1.0e-4) r = normalized(r);
  return r;
}

// v + w*(2 qv x v) + qv x (2 qv x v): the two-cross-product form Eigen uses
SVOH_HD Vec3 rotate(const Quat& q, const Vec3& v)
{
  double ux = q.y * v.z - q.z * v.y;
  double uy = q.z * v.x - q.x * v.z;
  double uz = q.x * v.y - q.y * v.x;
  ux += ux; uy += uy; uz += uz;
  Vec3 r;
  r.x = v.x + q.w * ux + (q.y * uz - q.z * uy);
  r.y = v.y + q.w * uy + (q.z * ux - q.x * uz);
  r.z = v.z + q.w * uz + (q.x * uy - q.y * ux);
  return r;
}

// rotation by conj(q)/|q|^2 (Eigen's Quaternion::inverse())
SVOH_HD Vec3 inverse_rotate(const Quat& q, const Vec3& v)
{
  const double n2 = sqnorm(q);
  Quat qi = { 0, 0, 0, 0 };
  if (n2 > 0.0) { qi.w = q.w / n2; qi.x = -q.x / n2; qi.y = -q.y / n2; qi.z = -q.z / n2; }
  return rotate(qi, v);
}

SVOH_HD Vec3 transform(const Rigid& T, const Vec3& p)
{
  Vec3 r = rotate(T.q, p);
  r.x += T.t.x; r.y += T.t.y; r.z += T.t.z;
  return r;
}

SVOH_HD Rigid mul(const Rigid& a, const Rigid& b)
{
  Rigid r;
  r.q = mul(a.q, b.q);
  Vec3 rt = rotate(a.q, b.t);
  r.t.x = a.t.x + rt.x; r.t.y = a.t.y + rt.y; r.t.z = a.t.z + rt.z;
  return r;
}

SVOH_HD Rigid inverse(const Rigid& a)
{
  Rigid r;
  r.q.w = a.q.w; r.q.x = -a.q.x; r.q.y = -a.q.y; r.q.z = -a.q.z;
  Vec3 it = inverse_rotate(a.q, a.t);
  r.t.x = -it.x; r.t.y = -it.y; r.t.z = -it.z;
  return r;
}

// 4th root of DBL_EPSILON = 2^-13
#define SVOH_EPS_4TH_ROOT 1.220703125e-4

SVOH_HD Quat quat_exp(const Vec3& w)
{
  const double theta = sqrt(w.x * w.x + w.y * w.y + w.z * w.z);
  double na, sh, ch;
  sincos(theta * 0.5, &sh, &ch);   // one argument reduction for both (the one-lane Gauss-Newton step pays for every instruction)
  if (theta < SVOH_EPS_4TH_ROOT) na = 0.5 + (theta * theta) * (1.0 / 48.0);
  else na = sh / theta;
  Quat q = { ch, w.x * na, w.y * na, w.z * na };
  return q;
}

SVOH_HD Vec3 quat_log(const Quat& q)
{
  const double na = sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  const double eta = q.w;
  double scale;
  if (fabs(eta) < na) {
    scale = (eta >= 0) ? acos(eta) / na : -acos(-eta) / na;
  } else {
    const double a = (fabs(na) < SVOH_EPS_4TH_ROOT) ? 1.0 + na * na * (1.0 / 6.0) : asin(na) / na;
    scale = (eta > 0) ? a : -a;
  }
  Vec3 r = { q.x * (2.0 * scale), q.y * (2.0 * scale), q.z * (2.0 * scale) };
  return r;
}

// minkindr Transformation::exp(v) = { t = v[0:3], q = Exp(v[3:6]) } (decoupled)
SVOH_HD Rigid rigid_exp(const double v[6])
{
  Rigid r;
  Vec3 w = { v[3], v[4], v[5] };
  r.q = quat_exp(w);
  r.t.x = v[0]; r.t.y = v[1]; r.t.z = v[2];
  return r;
}

SVOH_HD void rigid_log(const Rigid& T, double v[6])
{
  v[0] = T.t.x; v[1] = T.t.y; v[2] = T.t.z;
  Vec3 w = quat_log(T.q);
  v[3] = w.x; v[4] = w.y; v[5] = w.z;
}

// row-major 3x3
SVOH_HD void to_matrix(const Quat& q, double R[9])
{
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// ---- camera ---------------------------------------------------------------

struct CamModel {
  double fx, fy, cx, cy;
  double k1, k2, p1, p2;
  int distortion;
  int width, height;
};

SVOH_HD CamModel load_camera(const svoh_camera& c)
{
  CamModel m;
  m.fx = c.fx; m.fy = c.fy; m.cx = c.cx; m.cy = c.cy;
  m.k1 = c.d[0]; m.k2 = c.d[1]; m.p1 = c.d[2]; m.p2 = c.d[3];
  m.distortion = c.distortion; m.width = c.width; m.height = c.height;
  return m;
}

SVOH_HD void radtan_distort(const CamModel& c, double& x, double& y)
{
  const double xx = x * x, yy = y * y, xy = x * y;
  const double xy2 = 2.0 * xy;
  const double r2 = xx + yy;
  const double cdist = (c.k1 + c.k2 * r2) * r2;
  const double nx = x + x * cdist + c.p1 * xy2 + c.p2 * (r2 + 2.0 * xx);
  const double ny = y + y * cdist + c.p2 * xy2 + c.p1 * (r2 + 2.0 * yy);
  x = nx; y = ny;
}

SVOH_HD void project3(const CamModel& c, const Vec3& p, double& u, double& v)
{
  const double z_inv = 1 / p.z;
  double x = p.x * z_inv, y = p.y * z_inv;
  if (c.distortion == SVOH_DISTORTION_RADTAN) radtan_distort(c, x, y);
  u = c.fx * x + c.cx;
  v = c.fy * y + c.cy;
}

// projection + 2x3 Jacobian (row-major) as PinholeProjection::project3 computes
// it: diag(fx,fy) * distortion.jacobian(uv_undistorted) * d(uv)/d(xyz)
SVOH_HD void project3_jacobian(const CamModel& c, const Vec3& p, double J[6])
{
  const double z_inv = 1 / p.z;
  const double x = p.x * z_inv, y = p.y * z_inv;
  const double d[6] = { z_inv, 0.0, -p.x * z_inv * z_inv, 0.0, z_inv, -p.y * z_inv * z_inv };
  double J00 = 1.0, J01 = 0.0, J10 = 0.0, J11 = 1.0;
  if (c.distortion == SVOH_DISTORTION_RADTAN) {
    const double xx = x * x, yy = y * y, xy = x * y;
    const double r2 = xx + yy;
    const double cdist = (c.k1 + c.k2 * r2) * r2;
    const double k2_r2_x4 = c.k2 * r2 * 4.0;
    const double cdist_p1 = cdist + 1.0;
    J00 = cdist_p1 + c.k1 * 2.0 * xx + k2_r2_x4 * xx + 2.0 * c.p1 * y + 6.0 * c.p2 * x;
    J11 = cdist_p1 + c.k1 * 2.0 * yy + k2_r2_x4 * yy + 2.0 * c.p2 * x + 6.0 * c.p1 * y;
    J10 = 2.0 * c.k1 * xy + k2_r2_x4 * xy + 2.0 * c.p1 * x + 2.0 * c.p2 * y;
    J01 = J10;
  }
  for (int k = 0; k < 3; ++k) {
    J[k] = c.fx * (J00 * d[k] + J01 * d[3 + k]);
    J[3 + k] = c.fy * (J10 * d[k] + J11 * d[3 + k]);
  }
}

SVOH_HD Vec3 back_project3(const CamModel& c, double u, double v)
{
  double x = (u - c.cx) * (1.0 / c.fx);
  double y = (v - c.cy) * (1.0 / c.fy);
  if (c.distortion == SVOH_DISTORTION_RADTAN) {
    const double x0 = x, y0 = y;
    for (int i = 0; i < 5; ++i) {
      const double xx = x * x, yy = y * y, xy = x * y;
      const double xy2 = 2 * xy;
      const double r2 = xx + yy;
      const double icdist = 1.0 / (1.0 + (c.k1 + c.k2 * r2) * r2);
      const double dx = c.p1 * xy2 + c.p2 * (r2 + 2.0 * xx);
      const double dy = c.p2 * xy2 + c.p1 * (r2 + 2.0 * yy);
      x = (x0 - dx) * icdist;
      y = (y0 - dy) * icdist;
    }
  }
  Vec3 r = { x, y, 1.0 };
  return r;
}

// ---- 8x8 LDL^T with diagonal pivoting (Eigen 3.4 LDLT<Lower> semantics) ----
// H col-major (symmetric, only the lower triangle is read), solves H dx = g.
// Zero pivots give zero solution components (pseudo-inverse of D), which is
// what makes the illumination-off case (rows 6,7 of H all zero) work.
// Returns false iff dx[0] is NaN (MiniLeastSquaresSolver::solveDefaultImpl).
// m: N*N workspace holding H on entry (destroyed); x: g on entry, dx on exit;
// tr/tmp: N-element scratch.  The caller supplies the storage so that device
// code can keep it in LDS (the pivoting needs run-time indexing).
template <int N>
SVOH_HD bool ldlt_solve_inplace(double* m, double* x, int* tr, double* tmp)
{
#define SVOH_M(r, c) m[(c) * N + (r)]
  for (int k = 0; k < N; ++k) {
    int big = k;
    double bigv = fabs(SVOH_M(k, k));
    for (int i = k + 1; i < N; ++i) {
      const double v = fabs(SVOH_M(i, i));
      if (v > bigv) { bigv = v; big = i; }
    }
    tr[k] = big;
    if (k != big) {
      for (int c = 0; c < k; ++c) { const double t = SVOH_M(k, c); SVOH_M(k, c) = SVOH_M(big, c); SVOH_M(big, c) = t; }
      for (int r = big + 1; r < N; ++r) { const double t = SVOH_M(r, k); SVOH_M(r, k) = SVOH_M(r, big); SVOH_M(r, big) = t; }
      { const double t = SVOH_M(k, k); SVOH_M(k, k) = SVOH_M(big, big); SVOH_M(big, big) = t; }
      for (int i = k + 1; i < big; ++i) { const double t = SVOH_M(i, k); SVOH_M(i, k) = SVOH_M(big, i); SVOH_M(big, i) = t; }
    }
    if (k > 0) {
      for (int c = 0; c < k; ++c) tmp[c] = SVOH_M(c, c) * SVOH_M(k, c);
      double acc = 0.0;
      for (int c = 0; c < k; ++c) acc += SVOH_M(k, c) * tmp[c];
      SVOH_M(k, k) -= acc;
      for (int r = k + 1; r < N; ++r) {
        double a = 0.0;
        for (int c = 0; c < k; ++c) a += SVOH_M(r, c) * tmp[c];
        SVOH_M(r, k) -= a;
      }
    }
    const double akk = SVOH_M(k, k);
    const bool pivot_ok = fabs(akk) > 0.0;
    if (k == 0 && !pivot_ok) {
      for (int j = 0; j < N; ++j) tr[j] = j;
      break;
    }
    if (pivot_ok)
      for (int r = k + 1; r < N; ++r) SVOH_M(r, k) /= akk;
  }
  for (int k = 0; k < N; ++k)
    if (tr[k] != k) { const double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
  for (int i = 0; i < N; ++i) {
    double a = x[i];
    for (int c = 0; c < i; ++c) a -= SVOH_M(i, c) * x[c];
    x[i] = a;
  }
  for (int i = 0; i < N; ++i) {
    const double d = SVOH_M(i, i);
    x[i] = (fabs(d) > DBL_MIN) ? x[i] / d : 0.0;
  }
  for (int i = N - 1; i >= 0; --i) {
    double a = x[i];
    for (int c = i + 1; c < N; ++c) a -= SVOH_M(c, i) * x[c];
    x[i] = a;
  }
  for (int k = N - 1; k >= 0; --k)
    if (tr[k] != k) { const double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
#undef SVOH_M
  return !(x[0] != x[0]);
}

}  // namespace svoh
