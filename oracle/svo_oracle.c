/*
 * svo_oracle.c -- CPU restatement of the reference's direct front end, part 1:
 * image pyramid, SE3/camera maths, the mini least-squares solver and
 * SparseImgAlign.   TEST INFRASTRUCTURE ONLY -- see svo_oracle.h.
 *
 * PARITY UNPINNED (no golden vectors in the reference, reference unbuildable
 * here): see the header of svo_oracle.h and DESIGN.md.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference tree).  Arithmetic is IEEE double, evaluated in the order the
 * reference writes it; build with -ffp-contract=off and without -ffast-math so
 * fixtures are reproducible on any x86-64 host.
 */
#include "svo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================== */
/* a-0  image pyramid                                                       */
/* ======================================================================== */

/* vision.cpp:19-44 halfSampleSSE2: _mm_avg_epu8 of the two rows (round half up),
 * then _mm_avg_epu16 of the even/odd columns (round half up again). */
static inline uint8_t half_sse2(uint8_t a, uint8_t b, uint8_t c, uint8_t d)
{
  /* a b = top row (x, x+1), c d = bottom row */
  unsigned v0 = ((unsigned)a + c + 1u) >> 1; /* avg_epu8(here,next) column x   */
  unsigned v1 = ((unsigned)b + d + 1u) >> 1; /* column x+1 */
  return (uint8_t)((v0 + v1 + 1u) >> 1);     /* avg_epu16 */
}

/* vision.cpp:108 scalar path: (a+b+c+d)/4 truncating */
static inline uint8_t half_scalar(uint8_t a, uint8_t b, uint8_t c, uint8_t d)
{
  return (uint8_t)(((unsigned)a + b + c + d) / 4u);
}

void orc_half_sample(const uint8_t* in, int in_w, int in_h, int in_pitch,
                     uint8_t* out, int out_pitch, int rounding)
{
  const int out_w = in_w / 2, out_h = in_h / 2;
  int use_sse = 0;
  if (rounding == SVOH_HALFSAMPLE_SSE2) use_sse = 1;
  else if (rounding == SVOH_HALFSAMPLE_REFERENCE)
    /* vision.cpp:80-87: aligned data, cols%16==0, continuous rows */
    use_sse = ((in_w % 16) == 0) && (in_pitch == in_w);
  if (use_sse) {
    /* vision.cpp:24-43: sw = w>>4 blocks of 16 columns, sh = h>>1 rows */
    const int sw = in_w >> 4, sh = in_h >> 1;
    for (int y = 0; y < sh; ++y) {
      const uint8_t* top = in + (size_t)(2 * y) * in_pitch;
      const uint8_t* bot = top + in_pitch;
      uint8_t* p = out + (size_t)y * out_pitch;
      for (int x = 0; x < sw * 8; ++x)
        p[x] = half_sse2(top[2 * x], top[2 * x + 1], bot[2 * x], bot[2 * x + 1]);
    }
    return;
  }
  /* vision.cpp:97-110 */
  for (int y = 0; y < out_h && (2 * y + 1) < in_h; ++y) {
    const uint8_t* top = in + (size_t)(2 * y) * in_pitch;
    const uint8_t* bot = top + in_pitch;
    uint8_t* p = out + (size_t)y * out_pitch;
    for (int x = 0; x < out_w; ++x)
      p[x] = half_scalar(top[2 * x], top[2 * x + 1], bot[2 * x], bot[2 * x + 1]);
  }
}

/* frame.cpp:372-386 */
void orc_create_img_pyramid(const uint8_t* img0, int w, int h, int pitch,
                            int n_levels, int rounding, uint8_t* const* out_levels)
{
  const uint8_t* prev = img0;
  int pw = w, ph = h, ppitch = pitch;
  if (out_levels[0] && out_levels[0] != img0) {
    for (int y = 0; y < h; ++y) memcpy(out_levels[0] + (size_t)y * w, img0 + (size_t)y * pitch, (size_t)w);
    prev = out_levels[0];
    ppitch = w;
  }
  for (int i = 1; i < n_levels; ++i) {
    const int nw = pw / 2, nh = ph / 2;
    orc_half_sample(prev, pw, ph, ppitch, out_levels[i], nw, rounding);
    prev = out_levels[i];
    pw = nw; ph = nh; ppitch = nw;
  }
}

/* ======================================================================== */
/* SE3 / quaternion: minkindr on top of Eigen::Quaternion                   */
/* ======================================================================== */

/* Eigen/src/Geometry/Quaternion.h quat_product (generic, non-vectorised) */
static void quat_mul_raw(const double a[4], const double b[4], double r[4])
{
  const double aw = a[0], ax = a[1], ay = a[2], az = a[3];
  const double bw = b[0], bx = b[1], by = b[2], bz = b[3];
  r[0] = aw * bw - ax * bx - ay * by - az * bz;
  r[1] = aw * bx + ax * bw + ay * bz - az * by;
  r[2] = aw * by + ay * bw + az * bx - ax * bz;
  r[3] = aw * bz + az * bw + ax * by - ay * bx;
}

static double quat_sqnorm(const double q[4])
{
  return q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
}

static void quat_normalize(double q[4])
{
  const double n = sqrt(quat_sqnorm(q));
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/* rotation-quaternion-inl.h:435-442 operator* + :580-589 normalizationHelper */
void orc_quat_mul(const double a[4], const double b[4], double out[4])
{
  double r[4];
  quat_mul_raw(a, b, r);
  if (fabs(quat_sqnorm(r) - 1.0) > 1.0e-4) quat_normalize(r);
  memcpy(out, r, sizeof r);
}

/* Eigen QuaternionBase::_transformVector: uv = q.vec x v; uv += uv;
 * v + w*uv + q.vec x uv   (rotation-quaternion-inl.h:321-326 rotate) */
void orc_quat_rotate(const double q[4], const double v[3], double out[3])
{
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  double uvx = y * v[2] - z * v[1];
  double uvy = z * v[0] - x * v[2];
  double uvz = x * v[1] - y * v[0];
  uvx += uvx; uvy += uvy; uvz += uvz;
  const double cx = y * uvz - z * uvy;
  const double cy = z * uvx - x * uvz;
  const double cz = x * uvy - y * uvx;
  out[0] = v[0] + w * uvx + cx;
  out[1] = v[1] + w * uvy + cy;
  out[2] = v[2] + w * uvz + cz;
}

/* rotation-quaternion-inl.h:353-357 inverseRotate: q_A_B_.inverse()*v with
 * Eigen's Quaternion::inverse() = conjugate / squaredNorm */
static void quat_inverse_rotate(const double q[4], const double v[3], double out[3])
{
  const double n2 = quat_sqnorm(q);
  double qi[4];
  if (n2 > 0.0) { qi[0] = q[0] / n2; qi[1] = -q[1] / n2; qi[2] = -q[2] / n2; qi[3] = -q[3] / n2; }
  else { qi[0] = qi[1] = qi[2] = qi[3] = 0.0; }
  orc_quat_rotate(qi, v, out);
}

/* rotation-quaternion-inl.h:92-104 */
static int less_than_eps_4th_root(double x)
{
  const double eps4 = pow(DBL_EPSILON, 1.0 / 4.0);
  return x < eps4;
}
static double arc_sin_x_over_x(double x)
{
  if (less_than_eps_4th_root(fabs(x))) return 1.0 + x * x * (1.0 / 6.0);
  return asin(x) / x;
}

/* rotation-quaternion-inl.h:522-540 */
void orc_quat_exp(const double dx[3], double q[4])
{
  const double theta = sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]);
  double na;
  if (less_than_eps_4th_root(theta)) {
    const double one_over_48 = 1.0 / 48.0;
    na = 0.5 + (theta * theta) * one_over_48;
  } else {
    na = sin(theta * 0.5) / theta;
  }
  const double ct = cos(theta * 0.5);
  q[0] = ct; q[1] = dx[0] * na; q[2] = dx[1] * na; q[3] = dx[2] * na;
}

/* rotation-quaternion-inl.h:478-519 */
void orc_quat_log(const double q[4], double out[3])
{
  const double ax = q[1], ay = q[2], az = q[3];
  const double na = sqrt(ax * ax + ay * ay + az * az);
  const double eta = q[0];
  double scale;
  if (fabs(eta) < na) {
    if (eta >= 0) scale = acos(eta) / na;
    else scale = -acos(-eta) / na;
  } else {
    if (eta > 0) scale = arc_sin_x_over_x(na);
    else scale = -arc_sin_x_over_x(na);
  }
  out[0] = ax * (2.0 * scale);
  out[1] = ay * (2.0 * scale);
  out[2] = az * (2.0 * scale);
}

/* Eigen QuaternionBase::toRotationMatrix; R row-major */
void orc_quat_to_matrix(const double q[4], double R[9])
{
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

/* quat-transformation-inl.h:155-161 operator* */
void orc_se3_mul(const svoh_se3* a, const svoh_se3* b, svoh_se3* out)
{
  svoh_se3 r;
  double rt[3];
  orc_quat_mul(a->q, b->q, r.q);
  orc_quat_rotate(a->q, b->t, rt);
  r.t[0] = a->t[0] + rt[0]; r.t[1] = a->t[1] + rt[1]; r.t[2] = a->t[2] + rt[2];
  *out = r;
}

/* quat-transformation-inl.h:212-216 inverse(): (q.inverse(), -q.inverseRotate(t));
 * RotationQuaternion::inverse() = conjugated() (rotation-quaternion-inl.h:296-316) */
void orc_se3_inverse(const svoh_se3* a, svoh_se3* out)
{
  svoh_se3 r;
  double it[3];
  r.q[0] = a->q[0]; r.q[1] = -a->q[1]; r.q[2] = -a->q[2]; r.q[3] = -a->q[3];
  quat_inverse_rotate(a->q, a->t, it);
  r.t[0] = -it[0]; r.t[1] = -it[1]; r.t[2] = -it[2];
  *out = r;
}

/* quat-transformation-inl.h:163-168 transform */
void orc_se3_transform(const svoh_se3* T, const double p[3], double out[3])
{
  double r[3];
  orc_quat_rotate(T->q, p, r);
  out[0] = r[0] + T->t[0]; out[1] = r[1] + T->t[1]; out[2] = r[2] + T->t[2];
}

/* quat-transformation-inl.h:79-84 ctor from Vector6 (t = head3, q = Exp(tail3)),
 * :229-231 exp(vec) */
void orc_se3_exp(const double v[6], svoh_se3* out)
{
  orc_quat_exp(v + 3, out->q);
  out->t[0] = v[0]; out->t[1] = v[1]; out->t[2] = v[2];
}

/* quat-transformation-inl.h:233-238 */
void orc_se3_log(const svoh_se3* T, double v[6])
{
  v[0] = T->t[0]; v[1] = T->t[1]; v[2] = T->t[2];
  orc_quat_log(T->q, v + 3);
}

/* ======================================================================== */
/* a-15  camera                                                             */
/* ======================================================================== */

/* radial_tangential_distortion.h:57-67 distort(Vector2d) */
static void radtan_distort(const double k[4], const double in[2], double out[2])
{
  const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3];
  const double xx = in[0] * in[0];
  const double yy = in[1] * in[1];
  const double xy = in[0] * in[1];
  const double xy2 = 2.0 * xy;
  const double r2 = xx + yy;
  const double cdist = (k1 + k2 * r2) * r2;
  out[0] = in[0] + in[0] * cdist + p1 * xy2 + p2 * (r2 + 2.0 * xx);
  out[1] = in[1] + in[1] * cdist + p2 * xy2 + p1 * (r2 + 2.0 * yy);
}

/* radial_tangential_distortion.h:69-88 jacobian (row-major 2x2) */
static void radtan_jacobian(const double k[4], const double px[2], double J[4])
{
  const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3];
  const double xx = px[0] * px[0];
  const double yy = px[1] * px[1];
  const double xy = px[0] * px[1];
  const double r2 = xx + yy;
  const double cdist = (k1 + k2 * r2) * r2;
  const double k2_r2_x4 = k2 * r2 * 4.0;
  const double cdist_p1 = cdist + 1.0;
  J[0] = cdist_p1 + k1 * 2.0 * xx + k2_r2_x4 * xx + 2.0 * p1 * px[1] + 6.0 * p2 * px[0];
  J[3] = cdist_p1 + k1 * 2.0 * yy + k2_r2_x4 * yy + 2.0 * p2 * px[0] + 6.0 * p1 * px[1];
  J[2] = 2.0 * k1 * xy + k2_r2_x4 * xy + 2.0 * p1 * px[0] + 2.0 * p2 * px[1];
  J[1] = J[2];
}

/* radial_tangential_distortion.h:90-106 undistort: 5 fixed-point iterations */
static void radtan_undistort(const double k[4], double* x, double* y)
{
  const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3];
  const double x0 = *x, y0 = *y;
  for (int i = 0; i < 5; ++i) {
    const double xx = (*x) * (*x);
    const double yy = (*y) * (*y);
    const double xy = (*x) * (*y);
    const double xy2 = 2 * xy;
    const double r2 = xx + yy;
    const double icdist = 1.0 / (1.0 + (k1 + k2 * r2) * r2);
    const double dx = p1 * xy2 + p2 * (r2 + 2.0 * xx);
    const double dy = p2 * xy2 + p1 * (r2 + 2.0 * yy);
    *x = (x0 - dx) * icdist;
    *y = (y0 - dy) * icdist;
  }
}

/* pinhole_projection.hpp:44-64 */
void orc_project3(const svoh_camera* cam, const double p[3], double uv_out[2], double* J23)
{
  const double z_inv = 1 / p[2];
  const double uv[2] = { p[0] * z_inv, p[1] * z_inv };
  double ud[2];
  if (cam->distortion == SVOH_DISTORTION_RADTAN) radtan_distort(cam->d, uv, ud);
  else { ud[0] = uv[0]; ud[1] = uv[1]; }
  uv_out[0] = cam->fx * ud[0] + cam->cx;
  uv_out[1] = cam->fy * ud[1] + cam->cy;
  if (J23) {
    /* duv_dxy = [I*z_inv, -p.head2*z_inv*z_inv] */
    const double d[6] = { z_inv, 0.0, -p[0] * z_inv * z_inv,
                          0.0, z_inv, -p[1] * z_inv * z_inv };
    double Jd[4] = { 1.0, 0.0, 0.0, 1.0 };
    if (cam->distortion == SVOH_DISTORTION_RADTAN) radtan_jacobian(cam->d, uv, Jd);
    /* focal_matrix * Jd * duv_dxy */
    for (int c = 0; c < 3; ++c) {
      const double a0 = Jd[0] * d[c] + Jd[1] * d[3 + c];
      const double a1 = Jd[2] * d[c] + Jd[3] * d[3 + c];
      J23[c] = cam->fx * a0;
      J23[3 + c] = cam->fy * a1;
    }
  }
}

/* pinhole_projection.hpp:30-42 */
void orc_back_project3(const svoh_camera* cam, const double kp[2], double f[3])
{
  const double fx_inv = 1.0 / cam->fx, fy_inv = 1.0 / cam->fy;
  double x = (kp[0] - cam->cx) * fx_inv;
  double y = (kp[1] - cam->cy) * fy_inv;
  if (cam->distortion == SVOH_DISTORTION_RADTAN) radtan_undistort(cam->d, &x, &y);
  f[0] = x; f[1] = y; f[2] = 1.0;
}

/* ======================================================================== */
/* a-2  Eigen 3.4 LDLT (Lower, unblocked, diagonal pivoting) + solve        */
/* ======================================================================== */

extern int g_orc_ldlt_mode;   /* svo_oracle_matcher.c: orc_set_third_party_modes */

int orc_ldlt_solve(int n, const double* H, const double* g, double* dx)
{
  double m[64];
  int tr[8];
  double temp[8];
  memcpy(m, H, sizeof(double) * (size_t)n * n);
#define M(r, c) m[(c) * n + (r)]
  int all_zero_diag = 0;
  /* Eigen/src/Cholesky/LDLT.h ldlt_inplace<Lower>::unblocked */
  for (int k = 0; k < n; ++k) {
    int big = k;
    double bigv = fabs(M(k, k));
    for (int i = k + 1; i < n; ++i) {
      const double v = fabs(M(i, i));
      if (v > bigv) { bigv = v; big = i; }
    }
    tr[k] = big;
    if (k != big) {
      const int s = n - big - 1;
      for (int c = 0; c < k; ++c) { double t = M(k, c); M(k, c) = M(big, c); M(big, c) = t; }
      for (int r = 0; r < s; ++r) {
        double t = M(big + 1 + r, k); M(big + 1 + r, k) = M(big + 1 + r, big); M(big + 1 + r, big) = t;
      }
      { double t = M(k, k); M(k, k) = M(big, big); M(big, big) = t; }
      for (int i = k + 1; i < big; ++i) {
        double t = M(i, k); M(i, k) = M(big, i); M(big, i) = t;
      }
    }
    const int rs = n - k - 1;
    if (k > 0) {
      for (int c = 0; c < k; ++c) temp[c] = M(c, c) * M(k, c);
      if (g_orc_ldlt_mode == 1) {
        /* sensitivity mode: the same sums taken as a two-lane packet reduction (even + odd partial sums, then the
         * tail), the order a vectorised dot product would use -- same real numbers, other rounding */
        double pe = 0.0, po = 0.0;
        int c = 0;
        for (; c + 1 < k; c += 2) { pe += M(k, c) * temp[c]; po += M(k, c + 1) * temp[c + 1]; }
        double acc = pe + po;
        if (c < k) acc += M(k, c) * temp[c];
        M(k, k) -= acc;
        for (int r = 0; r < rs; ++r) {
          double qe = 0.0, qo = 0.0;
          int cc = 0;
          for (; cc + 1 < k; cc += 2) { qe += M(k + 1 + r, cc) * temp[cc]; qo += M(k + 1 + r, cc + 1) * temp[cc + 1]; }
          double a = qe + qo;
          if (cc < k) a += M(k + 1 + r, cc) * temp[cc];
          M(k + 1 + r, k) -= a;
        }
      } else {
      double acc = 0.0;
      for (int c = 0; c < k; ++c) acc += M(k, c) * temp[c];
      M(k, k) -= acc;
      for (int r = 0; r < rs; ++r) {
        double a = 0.0;
        for (int c = 0; c < k; ++c) a += M(k + 1 + r, c) * temp[c];
        M(k + 1 + r, k) -= a;
      }
      }
    }
    const double akk = M(k, k);
    const int pivot_is_valid = fabs(akk) > 0.0;
    if (k == 0 && !pivot_is_valid) {
      for (int j = 0; j < n; ++j) tr[j] = j;
      all_zero_diag = 1;
      break;
    }
    if (rs > 0 && pivot_is_valid)
      for (int r = 0; r < rs; ++r) M(k + 1 + r, k) /= akk;
  }
  (void)all_zero_diag;
  /* LDLT::_solve_impl_transposed */
  double x[8];
  for (int i = 0; i < n; ++i) x[i] = g[i];
  for (int k = 0; k < n; ++k)
    if (tr[k] != k) { double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
  for (int i = 0; i < n; ++i) {          /* unit-lower forward substitution */
    double a = x[i];
    for (int c = 0; c < i; ++c) a -= M(i, c) * x[c];
    x[i] = a;
  }
  for (int i = 0; i < n; ++i) {          /* pseudo-inverse of D, tolerance = DBL_MIN */
    const double d = M(i, i);
    if (fabs(d) > DBL_MIN) x[i] /= d; else x[i] = 0.0;
  }
  for (int i = n - 1; i >= 0; --i) {     /* unit-upper (L^T) back substitution */
    double a = x[i];
    for (int c = i + 1; c < n; ++c) a -= M(c, i) * x[c];
    x[i] = a;
  }
  for (int k = n - 1; k >= 0; --k)
    if (tr[k] != k) { double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
#undef M
  for (int i = 0; i < n; ++i) dx[i] = x[i];
  return isnan(dx[0]) ? 0 : 1;
}

/* ======================================================================== */
/* a-1 .. a-8  SparseImgAlign                                               */
/* ======================================================================== */

typedef struct align_state {
  svoh_se3 T;      /* T_icur_iref */
  double alpha, beta;
} align_state;

typedef struct align_caches {
  int n;                 /* n_fts_to_track */
  int patch_size, patch_area;
  double* uv;            /* 2 x n */
  double* xyz_ref;       /* 3 x n */
  double* jac_proj;      /* 6 x 2n (col-major: column 2i = row 0 of frame_jac) */
  double* jac;           /* 8 x (n*area) */
  double* residual;      /* area x n */
  uint8_t* visible;      /* n */
  double* ref_patch;     /* area x n */
  int n_per_cam[SVOH_MAX_CAMS];
} align_caches;

static void caches_free(align_caches* c)
{
  free(c->uv); free(c->xyz_ref); free(c->jac_proj); free(c->jac);
  free(c->residual); free(c->visible); free(c->ref_patch);
  memset(c, 0, sizeof *c);
}

/* sparse_img_align.cpp:209-260 */
int orc_extract_features_subset(const orc_align_camera* cam, int max_level,
                                int patch_size_wb, int32_t* out_idx)
{
  const double scale = 1.0f / (1 << max_level);
  const orc_image* ref_img = &cam->ref_pyr.level[max_level];
  const int rows_minus_two = ref_img->height - 2;
  const int cols_minus_two = ref_img->width - 2;
  const double patch_center_wb = (patch_size_wb - 1) / 2.0f;
  int n = 0;
  for (int i = 0; i < cam->n_features; ++i) {
    if (!cam->flags[i]) continue; /* :239-245 folded into flags by the caller */
    const double u_tl = cam->px[2 * i + 0] * scale - patch_center_wb;
    const double v_tl = cam->px[2 * i + 1] * scale - patch_center_wb;
    const int u_tl_i = (int)floor(u_tl);
    const int v_tl_i = (int)floor(v_tl);
    if (!(u_tl_i < 0 || v_tl_i < 0
          || u_tl_i + patch_size_wb >= cols_minus_two
          || v_tl_i + patch_size_wb >= rows_minus_two))
      out_idx[n++] = i;
  }
  return n;
}

/* frame.h:342-357 Frame::jacobian_xyz2uv_imu; J row-major 2x6 */
static void jacobian_xyz2uv_imu(const svoh_se3* T_cam_imu, const double p_in_imu[3], double J[12])
{
  double Gx[18]; /* row-major 3x6: [I, -skew(p)] */
  const double px = p_in_imu[0], py = p_in_imu[1], pz = p_in_imu[2];
  /* vk::skew(v) = [0 -z y; z 0 -x; -y x 0]  (vikit/math_utils.h:85-92) */
  const double G[18] = { 1, 0, 0, -0.0, pz, -py,
                         0, 1, 0, -pz, -0.0, px,
                         0, 0, 1, py, -px, -0.0 };
  memcpy(Gx, G, sizeof G);
  double p_in_cam[3];
  orc_se3_transform(T_cam_imu, p_in_imu, p_in_cam);
  const double Jp[6] = { 1, 0, -p_in_cam[0] / p_in_cam[2],
                         0, 1, -p_in_cam[1] / p_in_cam[2] };
  double R[9];
  orc_quat_to_matrix(T_cam_imu->q, R);
  const double s = -1.0 / p_in_cam[2];
  double A[6], B[6];
  for (int i = 0; i < 6; ++i) A[i] = s * Jp[i];
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 3; ++c)
      B[r * 3 + c] = A[r * 3 + 0] * R[0 * 3 + c] + A[r * 3 + 1] * R[1 * 3 + c] + A[r * 3 + 2] * R[2 * 3 + c];
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 6; ++c)
      J[r * 6 + c] = B[r * 3 + 0] * Gx[0 * 6 + c] + B[r * 3 + 1] * Gx[1 * 6 + c] + B[r * 3 + 2] * Gx[2 * 6 + c];
}

/* frame.cpp:274-290 Frame::jacobian_xyz2image_imu; J row-major 2x6 */
static void jacobian_xyz2image_imu(const svoh_camera* cam, const svoh_se3* T_cam_imu,
                                   const double p_in_imu[3], double J[12])
{
  const double px = p_in_imu[0], py = p_in_imu[1], pz = p_in_imu[2];
  const double Gx[18] = { 1, 0, 0, -0.0, pz, -py,
                          0, 1, 0, -pz, -0.0, px,
                          0, 0, 1, py, -px, -0.0 };
  double p_in_cam[3], uv[2], Jp[6], R[9], B[6];
  orc_se3_transform(T_cam_imu, p_in_imu, p_in_cam);
  orc_project3(cam, p_in_cam, uv, Jp);
  orc_quat_to_matrix(T_cam_imu->q, R);
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 3; ++c)
      B[r * 3 + c] = Jp[r * 3 + 0] * R[0 * 3 + c] + Jp[r * 3 + 1] * R[1 * 3 + c] + Jp[r * 3 + 2] * R[2 * 3 + c];
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 6; ++c)
      J[r * 6 + c] = B[r * 3 + 0] * Gx[0 * 6 + c] + B[r * 3 + 1] * Gx[1 * 6 + c] + B[r * 3 + 2] * Gx[2 * 6 + c];
}

/* sparse_img_align.cpp:262-317 */
static void precompute_base_caches(const orc_align_camera* cam, const int32_t* fts, int n_fts,
                                   int use_distortion_jac, int* feature_counter,
                                   align_caches* c)
{
  const double focal_length = fabs(cam->cam.fx); /* pinhole_projection.hpp:66-70 errorMultiplier */
  for (int k = 0; k < n_fts; ++k) {
    const int i = fts[k];
    const int fc = *feature_counter;
    c->uv[2 * fc + 0] = cam->px[2 * i + 0];
    c->uv[2 * fc + 1] = cam->px[2 * i + 1];
    const double dx = cam->pos_world[3 * i + 0] - cam->ref_pos[0];
    const double dy = cam->pos_world[3 * i + 1] - cam->ref_pos[1];
    const double dz = cam->pos_world[3 * i + 2] - cam->ref_pos[2];
    const double depth = sqrt(dx * dx + dy * dy + dz * dz);
    const double xyz_ref[3] = { cam->f[3 * i + 0] * depth, cam->f[3 * i + 1] * depth, cam->f[3 * i + 2] * depth };
    c->xyz_ref[3 * fc + 0] = xyz_ref[0];
    c->xyz_ref[3 * fc + 1] = xyz_ref[1];
    c->xyz_ref[3 * fc + 2] = xyz_ref[2];
    double xyz_in_imu[3];
    orc_se3_transform(&cam->ref_T_imu_cam, xyz_ref, xyz_in_imu);
    double J[12];
    if (!use_distortion_jac) { /* camera type is always pinhole here */
      jacobian_xyz2uv_imu(&cam->ref_T_cam_imu, xyz_in_imu, J);
      for (int j = 0; j < 12; ++j) J[j] *= focal_length;
    } else {
      jacobian_xyz2image_imu(&cam->cam, &cam->ref_T_cam_imu, xyz_in_imu, J);
      for (int j = 0; j < 12; ++j) J[j] *= (-1.0);
    }
    /* jacobian_proj_cache.col(2*fc) = row 0, col(2*fc+1) = row 1 */
    for (int j = 0; j < 6; ++j) {
      c->jac_proj[(size_t)(2 * fc) * 6 + j] = J[j];
      c->jac_proj[(size_t)(2 * fc + 1) * 6 + j] = J[6 + j];
    }
    ++(*feature_counter);
  }
}

/* sparse_img_align.cpp:319-403 */
static void precompute_jacobians_and_ref_patches(const orc_image* ref_img, int level, int patch_size,
                                                 int nr_features, int estimate_alpha, int estimate_beta,
                                                 int* feature_counter, align_caches* c)
{
  const int stride = ref_img->pitch;
  const double scale = 1.0f / (1 << level);
  const int patch_area = patch_size * patch_size;
  const int border = 1;
  const int patch_size_wb = patch_size + 2 * border;
  const int patch_area_wb = patch_size_wb * patch_size_wb;
  const double patch_center_wb = (patch_size_wb - 1) / 2.0f;
  double interp_patch_array[(8 + 2) * (8 + 2)];
  (void)patch_area_wb;

  for (int i = 0; i < nr_features; ++i, ++(*feature_counter)) {
    const int fc = *feature_counter;
    const double u_tl = c->uv[2 * fc + 0] * scale - patch_center_wb;
    const double v_tl = c->uv[2 * fc + 1] * scale - patch_center_wb;
    const int u_tl_i = (int)floor(u_tl);
    const int v_tl_i = (int)floor(v_tl);
    const double subpix_u_tl = u_tl - u_tl_i;
    const double subpix_v_tl = v_tl - v_tl_i;
    const double wtl = (1.0 - subpix_u_tl) * (1.0 - subpix_v_tl);
    const double wtr = subpix_u_tl * (1.0 - subpix_v_tl);
    const double wbl = (1.0 - subpix_u_tl) * subpix_v_tl;
    const double wbr = subpix_u_tl * subpix_v_tl;
    const int jacobian_proj_col = 2 * fc;

    int pixel_counter = 0;
    for (int y = 0; y < patch_size_wb; ++y) {
      const uint8_t* r = ref_img->data + (ptrdiff_t)(v_tl_i + y) * stride + u_tl_i;
      for (int x = 0; x < patch_size_wb; ++x, ++r, ++pixel_counter)
        interp_patch_array[pixel_counter] = wtl * r[0] + wtr * r[1] + wbl * r[stride] + wbr * r[stride + 1];
    }

    pixel_counter = 0;
    for (int y = 0; y < patch_size; ++y) {
      for (int x = 0; x < patch_size; ++x, ++pixel_counter) {
        const int offset_center = (x + border) + patch_size_wb * (y + border);
        c->ref_patch[(size_t)fc * patch_area + pixel_counter] = interp_patch_array[offset_center];
        const double dx = 0.5f * (interp_patch_array[offset_center + 1] - interp_patch_array[offset_center - 1]);
        const double dy = 0.5f * (interp_patch_array[offset_center + patch_size_wb]
                                  - interp_patch_array[offset_center - patch_size_wb]);
        const size_t jacobian_col = (size_t)fc * patch_area + pixel_counter;
        const double* jp0 = &c->jac_proj[(size_t)jacobian_proj_col * 6];
        const double* jp1 = &c->jac_proj[(size_t)(jacobian_proj_col + 1) * 6];
        double* Jc = &c->jac[jacobian_col * 8];
        for (int j = 0; j < 6; ++j) Jc[j] = (dx * jp0[j] + dy * jp1[j]) * scale;
        Jc[6] = estimate_alpha ? -(interp_patch_array[offset_center]) : 0.0;
        Jc[7] = estimate_beta ? -1.0 : 0.0;
      }
    }
  }
}

/* sparse_img_align.cpp:405-498 */
static void compute_residuals_of_frame(const orc_image* cur_img, const svoh_camera* cam, int level,
                                       int patch_size, int nr_features, const svoh_se3* T_cur_ref,
                                       float alpha, float beta, int* feature_counter, align_caches* c)
{
  const int stride = cur_img->pitch;
  const double scale = 1.0f / (1 << level);
  const int patch_area = patch_size * patch_size;
  const double patch_center = (patch_size - 1) / 2.0f;

  for (int i = 0; i < nr_features; ++i, ++(*feature_counter)) {
    const int fc = *feature_counter;
    double xyz_cur[3];
    orc_se3_transform(T_cur_ref, &c->xyz_ref[3 * fc], xyz_cur);
    if (xyz_cur[2] < 0.0) { c->visible[fc] = 0; continue; }
    double uv_cur[2];
    orc_project3(cam, xyz_cur, uv_cur, NULL);
    const double uv_cur_pyr[2] = { uv_cur[0] * scale, uv_cur[1] * scale };
    const double u_tl = uv_cur_pyr[0] - patch_center;
    const double v_tl = uv_cur_pyr[1] - patch_center;
    if (u_tl < 0.0 || v_tl < 0.0
        || u_tl + patch_size + 2.0 >= cur_img->width
        || v_tl + patch_size + 2.0 >= cur_img->height) {
      c->visible[fc] = 0;
      continue;
    }
    c->visible[fc] = 1;
    const int u_tl_i = (int)floor(u_tl);
    const int v_tl_i = (int)floor(v_tl);
    const double subpix_u_tl = u_tl - u_tl_i;
    const double subpix_v_tl = v_tl - v_tl_i;
    const double wtl = (1.0 - subpix_u_tl) * (1.0 - subpix_v_tl);
    const double wtr = subpix_u_tl * (1.0 - subpix_v_tl);
    const double wbl = (1.0 - subpix_u_tl) * subpix_v_tl;
    const double wbr = subpix_u_tl * subpix_v_tl;
    int pixel_counter = 0;
    for (int y = 0; y < patch_size; ++y) {
      const uint8_t* p = cur_img->data + (ptrdiff_t)(v_tl_i + y) * stride + u_tl_i;
      for (int x = 0; x < patch_size; ++x, ++pixel_counter, ++p) {
        const double intensity_cur = wtl * p[0] + wtr * p[1] + wbl * p[stride] + wbr * p[stride + 1];
        const double res = (double)(intensity_cur * (1.0 + alpha) + beta)
                           - c->ref_patch[(size_t)fc * patch_area + pixel_counter];
        c->residual[(size_t)fc * patch_area + pixel_counter] = res;
      }
    }
  }
}

/* robust_cost.cpp:44-60 TukeyWeightFunction, b = 4.6851f (robust_cost.h:70) */
static float tukey_weight(float error)
{
  const float b = 4.6851f;
  const float b_square = b * b;
  const float x_square = error * error;
  if (x_square <= b_square) {
    const float tmp = 1.0f - x_square / b_square;
    return tmp * tmp;
  }
  return 0.0f;
}

/* sparse_img_align.cpp:500-541 */
static double compute_hessian_and_gradient(const align_caches* c, float weight_scale, int robust,
                                           double* H, double* g, int* n_meas_out)
{
  float chi2 = 0.0f;
  size_t n_meas = 0;
  const int patch_area = c->patch_area;
  for (int i = 0; i < c->n; ++i) {
    if (c->visible[i]) {
      const size_t patch_offset = (size_t)i * patch_area;
      for (int j = 0; j < patch_area; ++j) {
        const double res = c->residual[patch_offset + j];
        float weight = 1.0f;
        if (robust) weight = tukey_weight((float)(res / weight_scale));
        chi2 += res * res * weight;
        ++n_meas;
        const double* J = &c->jac[(patch_offset + j) * 8];
        for (int cc = 0; cc < 8; ++cc)
          for (int rr = 0; rr < 8; ++rr)
            H[cc * 8 + rr] += J[rr] * J[cc] * weight;
        for (int rr = 0; rr < 8; ++rr)
          g[rr] -= J[rr] * res * weight;
      }
    }
  }
  if (n_meas_out) *n_meas_out = (int)n_meas;
  return (double)(chi2 / n_meas);
}

typedef struct align_ctx {
  const svoh_align_options* opt;
  const orc_align_problem* pb;
  align_caches c;
  int level;
  int have_cache;
  double I_prior[8]; /* diagonal */
  int iter;
} align_ctx;

/* sparse_img_align.cpp:115-156 */
static double evaluate_error(align_ctx* a, const align_state* state, double* H, double* g, int* n_meas)
{
  const svoh_align_options* opt = a->opt;
  const orc_align_problem* pb = a->pb;
  if (!a->have_cache) {
    int fc = 0;
    for (int i = 0; i < pb->n_cams; ++i)
      precompute_jacobians_and_ref_patches(&pb->cams[i].ref_pyr.level[a->level], a->level, opt->patch_size,
                                           a->c.n_per_cam[i], opt->estimate_illumination_gain,
                                           opt->estimate_illumination_offset, &fc, &a->c);
    a->have_cache = 1;
  }
  int fc = 0;
  for (int i = 0; i < pb->n_cams; ++i) {
    svoh_se3 tmp, T_cur_ref;
    orc_se3_mul(&pb->cams[i].cur_T_cam_imu, &state->T, &tmp);
    orc_se3_mul(&tmp, &pb->cams[i].ref_T_imu_cam, &T_cur_ref);
    compute_residuals_of_frame(&pb->cams[i].cur_pyr.level[a->level], &pb->cams[i].cam, a->level,
                               opt->patch_size, a->c.n_per_cam[i], &T_cur_ref,
                               (float)state->alpha, (float)state->beta, &fc, &a->c);
  }
  return compute_hessian_and_gradient(&a->c, (float)opt->weight_scale, opt->robustification, H, g, n_meas);
}

/* sparse_img_align_base.cpp:77-107 */
static void apply_prior(align_ctx* a, const align_state* state, double* H, double* g)
{
  const svoh_align_prior* pr = &a->pb->prior;
  if (a->iter == 0) {
    double H_max_diag_trans = 0;
    for (int j = 0; j < 3; ++j) H_max_diag_trans = fmax(H_max_diag_trans, fabs(H[j * 8 + j]));
    double H_max_diag_rot = 0;
    for (int j = 3; j < 6; ++j) H_max_diag_rot = fmax(H_max_diag_rot, fabs(H[j * 8 + j]));
    for (int j = 0; j < 3; ++j) a->I_prior[j] = 1.0 * pr->lambda_trans * H_max_diag_trans;
    for (int j = 3; j < 6; ++j) a->I_prior[j] = 1.0 * pr->lambda_rot * H_max_diag_rot;
    a->I_prior[6] = pr->lambda_alpha * H[6 * 8 + 6];
    a->I_prior[7] = pr->lambda_beta * H[7 * 8 + 7];
  }
  for (int j = 0; j < 8; ++j) H[j * 8 + j] += a->I_prior[j];
  svoh_se3 Tinv, Td;
  double lg[6];
  orc_se3_inverse(&pr->T_prior, &Tinv);
  orc_se3_mul(&Tinv, &state->T, &Td);
  orc_se3_log(&Td, lg);
  for (int j = 0; j < 6; ++j) g[j] += a->I_prior[j] * lg[j];
  g[6] += a->I_prior[6] * (pr->alpha_prior - state->alpha);
  g[7] += a->I_prior[7] * (pr->beta_prior - state->beta);
}

/* sparse_img_align_base.cpp:64-75 */
static void update_state(const align_state* old_s, const double dx[8], align_state* new_s)
{
  double mdx[6];
  svoh_se3 E;
  for (int j = 0; j < 6; ++j) mdx[j] = -dx[j];
  orc_se3_exp(mdx, &E);
  orc_se3_mul(&old_s->T, &E, &new_s->T);
  new_s->alpha = (old_s->alpha - dx[6]) / (1.0 + dx[6]);
  new_s->beta = (old_s->beta - dx[7]) / (1.0 + dx[6]);
  quat_normalize(new_s->T.q);
}

static int build_selection(const svoh_align_options* opt, const orc_align_problem* pb,
                           align_caches* c, int32_t** fts_out)
{
  int total = 0;
  for (int i = 0; i < pb->n_cams; ++i) {
    fts_out[i] = (int32_t*)malloc(sizeof(int32_t) * (size_t)(pb->cams[i].n_features + 1));
    c->n_per_cam[i] = orc_extract_features_subset(&pb->cams[i], opt->max_level, opt->patch_size + 2, fts_out[i]);
    total += c->n_per_cam[i];
  }
  return total;
}

static void alloc_caches(align_caches* c, int n, int patch_size)
{
  const int area = patch_size * patch_size;
  c->n = n; c->patch_size = patch_size; c->patch_area = area;
  const size_t nn = (size_t)(n > 0 ? n : 1);
  c->uv = (double*)calloc(nn * 2, sizeof(double));
  c->xyz_ref = (double*)calloc(nn * 3, sizeof(double));
  c->jac_proj = (double*)calloc(nn * 12, sizeof(double));
  c->jac = (double*)calloc(nn * area * 8, sizeof(double));
  c->residual = (double*)calloc(nn * area, sizeof(double));
  c->visible = (uint8_t*)calloc(nn, 1);
  c->ref_patch = (double*)calloc(nn * area, sizeof(double));
}

/* sparse_img_align.cpp:34-113 with mini_least_squares_solver.hpp:42-107 inlined */
int orc_sparse_align_run(const svoh_align_options* opt, const orc_align_problem* pb,
                         svoh_align_result* res, orc_align_trace* trace)
{
  align_ctx a;
  memset(&a, 0, sizeof a);
  memset(res, 0, sizeof *res);
  a.opt = opt; a.pb = pb;
  int32_t* fts[SVOH_MAX_CAMS] = { 0 };
  const int n = build_selection(opt, pb, &a.c, fts);
  res->T_icur_iref = pb->T_icur_iref;
  res->alpha = pb->alpha_init; res->beta = pb->beta_init;
  res->n_fts_to_track = n;
  if (trace) trace->count = 0;
  if (n == 0) {
    for (int i = 0; i < pb->n_cams; ++i) free(fts[i]);
    res->status = 1;
    return 0;
  }
  {
    int keep[SVOH_MAX_CAMS];
    memcpy(keep, a.c.n_per_cam, sizeof keep);
    alloc_caches(&a.c, n, opt->patch_size);
    memcpy(a.c.n_per_cam, keep, sizeof keep);
  }
  align_state state;
  state.T = pb->T_icur_iref;
  state.alpha = pb->alpha_init;
  state.beta = pb->beta_init;

  int fc = 0;
  for (int i = 0; i < pb->n_cams; ++i)
    precompute_base_caches(&pb->cams[i], fts[i], a.c.n_per_cam[i], opt->use_distortion_jacobian, &fc, &a.c);

  int stop = 0; /* MiniLeastSquaresSolver::stop_, cleared by reset() only */
  for (a.level = opt->max_level; a.level >= opt->min_level; --a.level) {
    a.have_cache = 0;
    /* optimizeGaussNewton, mini_least_squares_solver.hpp:42-107 */
    align_state old_state = state;
    int n_eval = 0;
    for (a.iter = 0; a.iter < opt->max_iter; ++a.iter) {
      double H[64], g[8], dx[8];
      memset(H, 0, sizeof H); memset(g, 0, sizeof g);
      int n_meas = 0;
      const double new_chi2 = evaluate_error(&a, &state, H, g, &n_meas);
      ++n_eval;
      res->n_patch_iters += n_meas / (opt->patch_size * opt->patch_size);
      if (a.level < SVOH_MAX_LEVELS) {
        res->iters[a.level] = n_eval;
        res->n_meas[a.level] = n_meas;
        res->chi2[a.level] = new_chi2;
      }
      if (trace && trace->count < trace->capacity) {
        const int k = trace->count++;
        trace->level[k] = a.level;
        memcpy(&trace->H[(size_t)k * 64], H, sizeof H);
        memcpy(&trace->g[(size_t)k * 8], g, sizeof g);
        trace->chi2[k] = new_chi2;
        trace->n_meas[k] = n_meas;
        memcpy(&trace->state[(size_t)k * 9], state.T.q, 4 * sizeof(double));
        memcpy(&trace->state[(size_t)k * 9 + 4], state.T.t, 3 * sizeof(double));
        trace->state[(size_t)k * 9 + 7] = state.alpha;
        trace->state[(size_t)k * 9 + 8] = state.beta;
      }
      if (pb->prior.have_prior) apply_prior(&a, &state, H, g);
      if (!orc_ldlt_solve(8, H, g, dx)) stop = 1;
      if (stop) { /* stop_when_error_increases is false (mini_least_squares_solver.h:40) */
        state = old_state;
        res->status = 2;
        break;
      }
      align_state new_state;
      update_state(&state, dx, &new_state);
      old_state = state;
      state = new_state;
      double x_norm = -1; /* utils::norm_max, mini_least_squares_solver.hpp:10-21 */
      for (int j = 0; j < 8; ++j) { const double v = fabs(dx[j]); if (v > x_norm) x_norm = v; }
      if (x_norm < opt->eps) break;
    }
  }
  res->T_icur_iref = state.T;
  res->alpha = state.alpha;
  res->beta = state.beta;
  for (int i = 0; i < pb->n_cams; ++i) free(fts[i]);
  caches_free(&a.c);
  return n;
}

int orc_sparse_align_evaluate(const svoh_align_options* opt, const orc_align_problem* pb,
                              int level, double* H64, double* g8, double* chi2,
                              int32_t* n_meas, uint8_t* visibility, int32_t* n_selected)
{
  align_ctx a;
  memset(&a, 0, sizeof a);
  a.opt = opt; a.pb = pb;
  int32_t* fts[SVOH_MAX_CAMS] = { 0 };
  const int n = build_selection(opt, pb, &a.c, fts);
  if (n_selected) *n_selected = n;
  memset(H64, 0, 64 * sizeof(double));
  memset(g8, 0, 8 * sizeof(double));
  if (n == 0) { for (int i = 0; i < pb->n_cams; ++i) free(fts[i]); return 0; }
  {
    int keep[SVOH_MAX_CAMS];
    memcpy(keep, a.c.n_per_cam, sizeof keep);
    alloc_caches(&a.c, n, opt->patch_size);
    memcpy(a.c.n_per_cam, keep, sizeof keep);
  }
  int fc = 0;
  for (int i = 0; i < pb->n_cams; ++i)
    precompute_base_caches(&pb->cams[i], fts[i], a.c.n_per_cam[i], opt->use_distortion_jacobian, &fc, &a.c);
  align_state state;
  state.T = pb->T_icur_iref; state.alpha = pb->alpha_init; state.beta = pb->beta_init;
  a.level = level; a.have_cache = 0;
  int nm = 0;
  const double c2 = evaluate_error(&a, &state, H64, g8, &nm);
  if (chi2) *chi2 = c2;
  if (n_meas) *n_meas = nm;
  if (visibility) memcpy(visibility, a.c.visible, (size_t)n);
  for (int i = 0; i < pb->n_cams; ++i) free(fts[i]);
  caches_free(&a.c);
  return n;
}
