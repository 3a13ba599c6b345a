/*
 * svo_oracle_matcher.c -- CPU restatement, part 3: affine patch warp, ZMSSD,
 * 1-D/2-D sub-pixel alignment, Matcher (direct + epipolar) and the depth filter
 * seed update (a-10 ... a-14).  TEST INFRASTRUCTURE ONLY, PARITY UNPINNED --
 * see svo_oracle.h.
 *
 * Follows (paths relative to the reference tree)
 *   src/svo_direct/src/patch_warp.cpp:20-60, 97-156
 *   src/svo_direct/include/svo/direct/patch_utils.h:18-30
 *   src/svo_direct/include/svo/direct/patch_score.h:44-285 (scalar branch)
 *   src/svo_direct/src/feature_alignment.cpp:31-209 (align1D), 212-391 (align2D)
 *   src/svo_direct/src/matcher.cpp:31-505
 *   src/svo_direct/src/depth_filter.cpp:200-233, 367-596
 *   src/svo_common/include/svo/common/seed.h:106-169
 *   src/vikit/vikit_common/include/vikit/math_utils.h:186-194 (normPdf)
 *   src/vikit/vikit_cameras/.../camera_geometry_base.hpp:6-29, camera_geometry.hpp:28-39
 * Third-party arithmetic restated from the published algorithm (Eigen 3.4, generic
 * non-vectorised paths): Matrix2d/2f::inverse, Matrix3f::inverse (cofactors of
 * column 0, determinant, adjugate * invdet), Matrix4f::inverse (cofactor_4x4,
 * division by col(0).row(0)), AngleAxis::toRotationMatrix, normalize().
 * The reference's own x86 build uses Eigen's SSE 4x4 float inverse, whose
 * operation order differs: float results agree to rounding only.
 */
#define _USE_MATH_DEFINES
#define _DEFAULT_SOURCE
#include <float.h>
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <stdlib.h>
#include <string.h>

#include "svo_oracle.h"

/* ---- small helpers --------------------------------------------------------- */
static void mat2d_inverse(const double m[4] /*col-major*/, double r[4])
{
  const double det = m[0] * m[3] - m[1] * m[2];
  const double invdet = 1.0 / det;
  r[0] = m[3] * invdet;   /* (0,0) */
  r[1] = -m[1] * invdet;  /* (1,0) */
  r[2] = -m[2] * invdet;  /* (0,1) */
  r[3] = m[0] * invdet;   /* (1,1) */
}

static void normalize2(double v[2])
{
  const double z = v[0] * v[0] + v[1] * v[1];
  if (z > 0.0) { const double n = sqrt(z); v[0] /= n; v[1] /= n; }
}

static void normalize3(double v[3])
{
  const double z = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  if (z > 0.0) { const double n = sqrt(z); v[0] /= n; v[1] /= n; v[2] /= n; }
}

/* ---- a-10 warp ------------------------------------------------------------- */
/* patch_warp.cpp:20-60 (pinhole branch: xyz_du_ref *= xyz_ref[2]); A col-major 2x2 */
void orc_get_warp_matrix_affine(const svoh_camera* cam_ref, const svoh_camera* cam_cur, const double px_ref[2],
                                const double f_ref[3], double depth_ref, const svoh_se3* T_cur_ref, int level_ref,
                                double A_cur_ref[4])
{
  const int kHalfPatchSize = 5;
  const double xyz_ref[3] = { f_ref[0] * depth_ref, f_ref[1] * depth_ref, f_ref[2] * depth_ref };
  double xyz_du_ref[3], xyz_dv_ref[3];
  const double pdu[2] = { px_ref[0] + (double)kHalfPatchSize * (1 << level_ref), px_ref[1] + 0.0 * (1 << level_ref) };
  const double pdv[2] = { px_ref[0] + 0.0 * (1 << level_ref), px_ref[1] + (double)kHalfPatchSize * (1 << level_ref) };
  orc_back_project3(cam_ref, pdu, xyz_du_ref);
  orc_back_project3(cam_ref, pdv, xyz_dv_ref);
  for (int k = 0; k < 3; ++k) { xyz_du_ref[k] *= xyz_ref[2]; xyz_dv_ref[k] *= xyz_ref[2]; }
  double p[3], px_cur[2], px_du_cur[2], px_dv_cur[2];
  orc_se3_transform(T_cur_ref, xyz_ref, p);     orc_project3(cam_cur, p, px_cur, NULL);
  orc_se3_transform(T_cur_ref, xyz_du_ref, p);  orc_project3(cam_cur, p, px_du_cur, NULL);
  orc_se3_transform(T_cur_ref, xyz_dv_ref, p);  orc_project3(cam_cur, p, px_dv_cur, NULL);
  A_cur_ref[0] = (px_du_cur[0] - px_cur[0]) / kHalfPatchSize;
  A_cur_ref[1] = (px_du_cur[1] - px_cur[1]) / kHalfPatchSize;
  A_cur_ref[2] = (px_dv_cur[0] - px_cur[0]) / kHalfPatchSize;
  A_cur_ref[3] = (px_dv_cur[1] - px_cur[1]) / kHalfPatchSize;
}

/* patch_warp.cpp:97-110 */
int orc_get_best_search_level(const double A[4], int max_level)
{
  int search_level = 0;
  double D = A[0] * A[3] - A[1] * A[2];
  while (D > 3.0 && search_level < max_level) {
    search_level += 1;
    D *= 0.25;
  }
  return search_level;
}

/* patch_warp.cpp:112-156 */
int orc_warp_affine(const double A_cur_ref[4], const orc_image* img_ref, const double px_ref[2], int level_ref,
                    int search_level, int halfpatch_size, uint8_t* patch)
{
  double Ai[4];
  mat2d_inverse(A_cur_ref, Ai);
  const float s = (float)(1 << search_level);
  const float a00 = (float)Ai[0] * s, a10 = (float)Ai[1] * s, a01 = (float)Ai[2] * s, a11 = (float)Ai[3] * s;
  if (isnan(a00)) return 0;
  uint8_t* patch_ptr = patch;
  const float prx = (float)px_ref[0] / (float)(1 << level_ref);
  const float pry = (float)px_ref[1] / (float)(1 << level_ref);
  const int stride = img_ref->pitch;
  for (int y = -halfpatch_size; y < halfpatch_size; ++y) {
    for (int x = -halfpatch_size; x < halfpatch_size; ++x, ++patch_ptr) {
      const float fx = (float)x, fy = (float)y;
      const float pxx = (a00 * fx + a01 * fy) + prx;
      const float pxy = (a10 * fx + a11 * fy) + pry;
      const int xi = (int)floorf(pxx);
      const int yi = (int)floorf(pxy);
      if (xi < 0 || yi < 0 || xi + 1 >= img_ref->width || yi + 1 >= img_ref->height) return 0;
      const float subpix_x = pxx - xi;
      const float subpix_y = pxy - yi;
      const float w00 = (1.0f - subpix_x) * (1.0f - subpix_y);
      const float w01 = (1.0f - subpix_x) * subpix_y;
      const float w10 = subpix_x * (1.0f - subpix_y);
      const float w11 = 1.0f - w00 - w01 - w10;
      const uint8_t* ptr = img_ref->data + (ptrdiff_t)yi * stride + xi;
      *patch_ptr = (uint8_t)(w00 * ptr[0] + w01 * ptr[stride] + w10 * ptr[1] + w11 * ptr[stride + 1]);
    }
  }
  return 1;
}

/* patch_warp.cpp:158-230 warpPixelwise: every pixel of the patch (at the search level of the current frame) is
 * back-projected at the landmark's distance from the current camera, carried into the reference frame and sampled
 * there.  landmark_xyz = ref_ftr.landmark->pos() (world). */
int orc_warp_pixelwise(const orc_frame_view* cur_frame, const orc_frame_view* ref_frame, const double px_ref[2],
                       const double landmark_xyz[3], int level_ref, int level_cur, int halfpatch_size, uint8_t* patch)
{
  svoh_se3 T_w_ref, T_w_cur, T_cur_ref, T_ref_cur;
  orc_se3_inverse(&ref_frame->T_f_w, &T_w_ref);   /* Frame::pos() = T_world_cam().getPosition() (frame.h:261,306) */
  orc_se3_inverse(&cur_frame->T_f_w, &T_w_cur);
  const double dr[3] = { T_w_ref.t[0] - landmark_xyz[0], T_w_ref.t[1] - landmark_xyz[1], T_w_ref.t[2] - landmark_xyz[2] };
  const double dc[3] = { T_w_cur.t[0] - landmark_xyz[0], T_w_cur.t[1] - landmark_xyz[1], T_w_cur.t[2] - landmark_xyz[2] };
  const double depth_ref = sqrt(dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2]);
  const double depth_cur = sqrt(dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2]);
  double xyz_ref[3];
  orc_back_project3(&ref_frame->cam, px_ref, xyz_ref);
  normalize3(xyz_ref);
  xyz_ref[0] *= depth_ref; xyz_ref[1] *= depth_ref; xyz_ref[2] *= depth_ref;
  orc_se3_mul(&cur_frame->T_f_w, &T_w_ref, &T_cur_ref);
  double xyz_cur[3], px_cur[2];
  orc_se3_transform(&T_cur_ref, xyz_ref, xyz_cur);
  orc_project3(&cur_frame->cam, xyz_cur, px_cur, NULL);
  const double pcs[2] = { px_cur[0] / (1 << level_cur), px_cur[1] / (1 << level_cur) };
  orc_se3_mul(&ref_frame->T_f_w, &T_w_cur, &T_ref_cur);
  const orc_image* img_ref = &ref_frame->pyr.level[level_ref];
  const int stride = img_ref->pitch;
  uint8_t* patch_ptr = patch;
  for (int y = -halfpatch_size; y < halfpatch_size; ++y) {
    for (int x = -halfpatch_size; x < halfpatch_size; ++x, ++patch_ptr) {
      const double ele_search[2] = { (double)x + pcs[0], (double)y + pcs[1] };
      const double ele_px[2] = { ele_search[0] * (1 << level_cur), ele_search[1] * (1 << level_cur) };
      double e_cur[3], e_ref[3], ele_ref[2];
      orc_back_project3(&cur_frame->cam, ele_px, e_cur);
      normalize3(e_cur);
      e_cur[0] *= depth_cur; e_cur[1] *= depth_cur; e_cur[2] *= depth_cur;
      orc_se3_transform(&T_ref_cur, e_cur, e_ref);
      orc_project3(&ref_frame->cam, e_ref, ele_ref, NULL);
      ele_ref[0] = ele_ref[0] / (1 << level_ref); ele_ref[1] = ele_ref[1] / (1 << level_ref);
      const int xi = (int)floor(ele_ref[0]);
      const int yi = (int)floor(ele_ref[1]);
      if (xi < 0 || yi < 0 || xi + 1 >= img_ref->width || yi + 1 >= img_ref->height) return 0;
      const float subpix_x = (float)(ele_ref[0] - xi);
      const float subpix_y = (float)(ele_ref[1] - yi);
      const float w00 = (1.0f - subpix_x) * (1.0f - subpix_y);
      const float w01 = (1.0f - subpix_x) * subpix_y;
      const float w10 = subpix_x * (1.0f - subpix_y);
      const float w11 = 1.0f - w00 - w01 - w10;
      const uint8_t* ptr = img_ref->data + (ptrdiff_t)yi * stride + xi;
      *patch_ptr = (uint8_t)(w00 * ptr[0] + w01 * ptr[stride] + w10 * ptr[1] + w11 * ptr[stride + 1]);
    }
  }
  return 1;
}

/* patch_utils.h:18-30 */
static void create_patch_from_patch_with_border(const uint8_t* pwb, int patch_size, uint8_t* patch)
{
  for (int y = 1; y < patch_size + 1; ++y)
    for (int x = 0; x < patch_size; ++x)
      patch[(y - 1) * patch_size + x] = pwb[y * (patch_size + 2) + 1 + x];
}

/* ---- a-11 ZMSSD<4> ---------------------------------------------------------- */
typedef struct zmssd8 { const uint8_t* ref_patch; int sumA, sumAA; } zmssd8;

static void zmssd_init(zmssd8* z, const uint8_t* ref_patch)
{
  uint32_t sumA = 0, sumAA = 0;
  for (int r = 0; r < 64; ++r) { const uint8_t n = ref_patch[r]; sumA += n; sumAA += (uint32_t)n * n; }
  z->ref_patch = ref_patch; z->sumA = (int)sumA; z->sumAA = (int)sumAA;
}

int orc_zmssd_score(const uint8_t* ref_patch, const uint8_t* cur_patch, int stride)
{
  zmssd8 z;
  zmssd_init(&z, ref_patch);
  uint32_t sumB_u = 0, sumBB_u = 0, sumAB_u = 0;
  for (int y = 0, r = 0; y < 8; ++y) {
    const uint8_t* p = cur_patch + (ptrdiff_t)y * stride;
    for (int x = 0; x < 8; ++x, ++r) {
      const uint8_t c = p[x];
      sumB_u += c; sumBB_u += (uint32_t)c * c; sumAB_u += (uint32_t)c * z.ref_patch[r];
    }
  }
  const int sumB = (int)sumB_u, sumBB = (int)sumBB_u, sumAB = (int)sumAB_u;
  return z.sumAA - 2 * sumAB + sumBB - (z.sumA * z.sumA - 2 * z.sumA * sumB + sumB * sumB) / 64;
}

#define ZMSSD_THRESHOLD (2000 * 64)

/* ---- a-12 align1D / align2D -------------------------------------------------- */
static void mat3f_inverse(const float m[9] /*row-major*/, float r[9])
{
#define M3(i, j) m[(i) * 3 + (j)]
#define COF3(i, j) (M3(((i) + 1) % 3, ((j) + 1) % 3) * M3(((i) + 2) % 3, ((j) + 2) % 3) - M3(((i) + 1) % 3, ((j) + 2) % 3) * M3(((i) + 2) % 3, ((j) + 1) % 3))
  const float c00 = COF3(0, 0), c10 = COF3(1, 0), c20 = COF3(2, 0);
  const float det = (c00 * M3(0, 0) + c10 * M3(1, 0)) + c20 * M3(2, 0);
  const float invdet = 1.0f / det;
  r[0 * 3 + 0] = c00 * invdet; r[0 * 3 + 1] = c10 * invdet; r[0 * 3 + 2] = c20 * invdet;
  r[1 * 3 + 0] = COF3(0, 1) * invdet; r[1 * 3 + 1] = COF3(1, 1) * invdet; r[1 * 3 + 2] = COF3(2, 1) * invdet;
  r[2 * 3 + 0] = COF3(0, 2) * invdet; r[2 * 3 + 1] = COF3(1, 2) * invdet; r[2 * 3 + 2] = COF3(2, 2) * invdet;
#undef COF3
#undef M3
}

static float det3_helper(const float* m, int i1, int i2, int i3, int j1, int j2, int j3)
{
#define M4(i, j) m[(i) * 4 + (j)]
  return M4(i1, j1) * (M4(i2, j2) * M4(i3, j3) - M4(i2, j3) * M4(i3, j2));
}
static float cofactor4(const float* m, int i, int j)
{
  const int i1 = (i + 1) % 4, i2 = (i + 2) % 4, i3 = (i + 3) % 4;
  const int j1 = (j + 1) % 4, j2 = (j + 2) % 4, j3 = (j + 3) % 4;
  return det3_helper(m, i1, i2, i3, j1, j2, j3) + det3_helper(m, i2, i3, i1, j1, j2, j3) + det3_helper(m, i3, i1, i2, j1, j2, j3);
}
/* ---- sensitivity mode: Eigen 3.4's VECTORISED 4x4 float inverse -----------------------------------------------
 * The reference's x86 build takes Eigen/src/LU/arch/InverseSize4.h (compute_inverse_size4<Architecture::Target,
 * float, ...>: the 2x2 block formula of Intel's AP-928 note in Eigen's generic packet operations) instead of the
 * scalar cofactor path restated above.  Restated here from the published Eigen 3.4.0 source, packets as float[4]:
 *   input = [[A, B], [C, D]] (2x2 blocks), AB = A# B, DC = D# C (# = adjugate), det = |A||D| + |B||C| - tr(AB DC),
 *   iA = A|D| - B DC, iD = D|A| - C AB, iB = C|B| - D AB#, iC = B|C| - A DC#, all times (+,-,-,+)/det.
 * Same real-number result, different rounding: orc_set_third_party_modes(1, x) switches to it so that the number
 * of result codes / iteration counts that depend on the choice can be counted (tests/golden/eigen_sensitivity.json). */
static int g_inv4_mode = 0;   /* 0 = generic cofactor path, 1 = vectorised block formula */
int g_orc_ldlt_mode = 0;      /* used by svo_oracle.c: 0 = sequential sums, 1 = packet-of-two partial sums */
void orc_set_third_party_modes(int inverse4_mode, int ldlt_mode) { g_inv4_mode = inverse4_mode; g_orc_ldlt_mode = ldlt_mode; }

typedef struct { float v[4]; } pk4;
static pk4 pk_swz(pk4 a, pk4 b, int p, int q, int r, int s) { pk4 o = { { a.v[p], a.v[q], b.v[r], b.v[s] } }; return o; }  /* vec4f_swizzle2 */
static pk4 pk_movelh(pk4 a, pk4 b) { pk4 o = { { a.v[0], a.v[1], b.v[0], b.v[1] } }; return o; }
static pk4 pk_movehl(pk4 a, pk4 b) { pk4 o = { { b.v[2], b.v[3], a.v[2], a.v[3] } }; return o; }
static pk4 pk_dup(pk4 a, int p) { pk4 o = { { a.v[p], a.v[p], a.v[p], a.v[p] } }; return o; }
static pk4 pk_mul(pk4 a, pk4 b) { pk4 o; for (int k = 0; k < 4; ++k) o.v[k] = a.v[k] * b.v[k]; return o; }
static pk4 pk_add(pk4 a, pk4 b) { pk4 o; for (int k = 0; k < 4; ++k) o.v[k] = a.v[k] + b.v[k]; return o; }
static pk4 pk_sub(pk4 a, pk4 b) { pk4 o; for (int k = 0; k < 4; ++k) o.v[k] = a.v[k] - b.v[k]; return o; }

static void mat4f_inverse_vectorised(const float m[16] /*row-major*/, float r[16])
{
  /* Eigen's Matrix4f is column-major: packet L_k = column k; source and result orders match (movelh / movehl branch) */
  pk4 L1, L2, L3, L4;
  for (int k = 0; k < 4; ++k) { L1.v[k] = m[k * 4 + 0]; L2.v[k] = m[k * 4 + 1]; L3.v[k] = m[k * 4 + 2]; L4.v[k] = m[k * 4 + 3]; }
  const pk4 A = pk_movelh(L1, L2), B = pk_movehl(L2, L1), C = pk_movelh(L3, L4), D = pk_movehl(L4, L3);
  pk4 AB = pk_mul(pk_swz(A, A, 3, 3, 0, 0), B);
  AB = pk_sub(AB, pk_mul(pk_swz(A, A, 1, 1, 2, 2), pk_swz(B, B, 2, 3, 0, 1)));
  pk4 DC = pk_mul(pk_swz(D, D, 3, 3, 0, 0), C);
  DC = pk_sub(DC, pk_mul(pk_swz(D, D, 1, 1, 2, 2), pk_swz(C, C, 2, 3, 0, 1)));
  pk4 dA = pk_mul(pk_swz(A, A, 3, 3, 1, 1), A); dA = pk_sub(dA, pk_movehl(dA, dA));
  pk4 dB = pk_mul(pk_swz(B, B, 3, 3, 1, 1), B); dB = pk_sub(dB, pk_movehl(dB, dB));
  pk4 dC = pk_mul(pk_swz(C, C, 3, 3, 1, 1), C); dC = pk_sub(dC, pk_movehl(dC, dC));
  pk4 dD = pk_mul(pk_swz(D, D, 3, 3, 1, 1), D); dD = pk_sub(dD, pk_movehl(dD, dD));
  pk4 d = pk_mul(pk_swz(DC, DC, 0, 2, 1, 3), AB);
  d = pk_add(d, pk_movehl(d, d));
  d = pk_add(d, pk_swz(d, d, 1, 0, 0, 0));
  const pk4 d1 = pk_mul(dA, dD), d2 = pk_mul(dB, dC);
  const pk4 det = pk_dup(pk_sub(pk_add(d1, d2), d), 0);
  pk4 rd;
  for (int k = 0; k < 4; ++k) rd.v[k] = 1.0f / det.v[k];
  pk4 iD = pk_mul(pk_swz(C, C, 0, 0, 2, 2), pk_movelh(AB, AB));
  iD = pk_add(iD, pk_mul(pk_swz(C, C, 1, 1, 3, 3), pk_movehl(AB, AB)));
  iD = pk_sub(pk_mul(D, pk_dup(dA, 0)), iD);
  pk4 iA = pk_mul(pk_swz(B, B, 0, 0, 2, 2), pk_movelh(DC, DC));
  iA = pk_add(iA, pk_mul(pk_swz(B, B, 1, 1, 3, 3), pk_movehl(DC, DC)));
  iA = pk_sub(pk_mul(A, pk_dup(dD, 0)), iA);
  pk4 iB = pk_mul(D, pk_swz(AB, AB, 3, 0, 3, 0));
  iB = pk_sub(iB, pk_mul(pk_swz(D, D, 1, 0, 3, 2), pk_swz(AB, AB, 2, 1, 2, 1)));
  iB = pk_sub(pk_mul(C, pk_dup(dB, 0)), iB);
  pk4 iC = pk_mul(A, pk_swz(DC, DC, 3, 0, 3, 0));
  iC = pk_sub(iC, pk_mul(pk_swz(A, A, 1, 0, 3, 2), pk_swz(DC, DC, 2, 1, 2, 1)));
  iC = pk_sub(pk_mul(B, pk_dup(dC, 0)), iC);
  rd.v[1] = -rd.v[1]; rd.v[2] = -rd.v[2];    /* sign mask (+, -, -, +) */
  iA = pk_mul(iA, rd); iB = pk_mul(iB, rd); iC = pk_mul(iC, rd); iD = pk_mul(iD, rd);
  const pk4 c0 = pk_swz(iA, iB, 3, 1, 3, 1), c1 = pk_swz(iA, iB, 2, 0, 2, 0), c2 = pk_swz(iC, iD, 3, 1, 3, 1), c3 = pk_swz(iC, iD, 2, 0, 2, 0);
  for (int k = 0; k < 4; ++k) { r[k * 4 + 0] = c0.v[k]; r[k * 4 + 1] = c1.v[k]; r[k * 4 + 2] = c2.v[k]; r[k * 4 + 3] = c3.v[k]; }   /* column j of the result */
}

static void mat4f_inverse(const float m[16] /*row-major*/, float r[16])
{
  if (g_inv4_mode == 1) { mat4f_inverse_vectorised(m, r); return; }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      const float c = cofactor4(m, i, j);
      r[j * 4 + i] = ((i + j) & 1) ? -c : c;  /* result(j,i) = (-1)^(i+j) cofactor<i,j> */
    }
  /* result /= (matrix.col(0) . result.row(0)) */
  const float d = ((M4(0, 0) * r[0] + M4(1, 0) * r[1]) + M4(2, 0) * r[2]) + M4(3, 0) * r[3];
  for (int k = 0; k < 16; ++k) r[k] /= d;
#undef M4
}

/* feature_alignment.cpp:31-209; returns converged; px in/out (double), h_inv out (may be NULL) */
int orc_align_1d(const orc_image* cur_img, const double dir[2], const uint8_t* ref_patch_with_border,
                 const uint8_t* ref_patch, int n_iter, int affine_est_offset, int affine_est_gain,
                 double cur_px_estimate[2], double* h_inv)
{
  enum { kHalfPatchSize = 4, kPatchSize = 8, kPatchArea = 64, ref_step = 10 };
  int converged = 0;
  float ref_patch_dv[kPatchArea];
  float H[9] = { 0 };
  float* it_dv = ref_patch_dv;
  for (int y = 0; y < kPatchSize; ++y) {
    const uint8_t* it = ref_patch_with_border + (y + 1) * ref_step + 1;
    for (int x = 0; x < kPatchSize; ++x, ++it, ++it_dv) {
      float J[3];
      const float dx = (float)it[1] - (float)it[-1];
      const float dy = (float)it[ref_step] - (float)it[-ref_step];
      J[0] = (float)(0.5f * (dir[0] * dx + dir[1] * dy));
      J[1] = affine_est_offset ? 1.0f : 0.0f;
      J[2] = affine_est_gain ? -1.0f * it[0] : 0.0f;
      *it_dv = J[0];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) H[r * 3 + c] += J[r] * J[c];
    }
  }
  if (!affine_est_offset) H[1 * 3 + 1] = 1.0f;
  if (!affine_est_gain) H[2 * 3 + 2] = 1.0f;
  if (h_inv) *h_inv = 1.0 / H[0] * kPatchSize * kPatchSize;
  float Hinv[9];
  mat3f_inverse(H, Hinv);
  float mean_diff = 0;
  float alpha = 1.0;
  float u = (float)cur_px_estimate[0];
  float v = (float)cur_px_estimate[1];
  const float min_update_squared = (float)(0.03 * 0.03);
  const int cur_step = cur_img->pitch;
  for (int iter = 0; iter < n_iter; ++iter) {
    const int u_r = (int)floorf(u);
    const int v_r = (int)floorf(v);
    if (u_r < kHalfPatchSize || v_r < kHalfPatchSize || u_r >= cur_img->width - kHalfPatchSize ||
        v_r >= cur_img->height - kHalfPatchSize)
      break;
    if (isnan(u) || isnan(v)) return 0;
    const float subpix_x = u - u_r;
    const float subpix_y = v - v_r;
    const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
    const float wTR = (float)(subpix_x * (1.0 - subpix_y));
    const float wBL = (float)((1.0 - subpix_x) * subpix_y);
    const float wBR = subpix_x * subpix_y;
    const uint8_t* it_ref = ref_patch;
    const float* it_ref_dv = ref_patch_dv;
    float Jres[3] = { 0, 0, 0 };
    for (int y = 0; y < kPatchSize; ++y) {
      const uint8_t* it = cur_img->data + (ptrdiff_t)(v_r + y - kHalfPatchSize) * cur_step + u_r - kHalfPatchSize;
      for (int x = 0; x < kPatchSize; ++x, ++it, ++it_ref, ++it_ref_dv) {
        const float cur_intensity = wTL * it[0] + wTR * it[1] + wBL * it[cur_step] + wBR * it[cur_step + 1];
        const float res = cur_intensity - alpha * (*it_ref) + mean_diff;
        Jres[0] -= res * (*it_ref_dv);
        if (affine_est_offset) Jres[1] -= res;
        if (affine_est_gain) Jres[2] -= (-1) * res * (*it_ref);
      }
    }
    if (!affine_est_offset) Jres[1] = 0.0f;
    if (!affine_est_gain) Jres[2] = 0.0f;
    float update[3];
    for (int r = 0; r < 3; ++r) update[r] = (Hinv[r * 3 + 0] * Jres[0] + Hinv[r * 3 + 1] * Jres[1]) + Hinv[r * 3 + 2] * Jres[2];
    u = (float)(u + update[0] * dir[0]);
    v = (float)(v + update[0] * dir[1]);
    mean_diff += update[1];
    alpha += update[2];
    if (update[0] * update[0] < min_update_squared) { converged = 1; break; }
  }
  cur_px_estimate[0] = u;
  cur_px_estimate[1] = v;
  return converged;
}

/* feature_alignment.cpp:212-391 */
int orc_align_2d(const orc_image* cur_img, const uint8_t* ref_patch_with_border, const uint8_t* ref_patch, int n_iter,
                 int affine_est_offset, int affine_est_gain, double cur_px_estimate[2])
{
  enum { halfpatch_size_ = 4, patch_size_ = 8, patch_area_ = 64, ref_step = 10 };
  int converged = 0;
  float ref_patch_dx[patch_area_], ref_patch_dy[patch_area_];
  float H[16] = { 0 };
  float* it_dx = ref_patch_dx;
  float* it_dy = ref_patch_dy;
  for (int y = 0; y < patch_size_; ++y) {
    const uint8_t* it = ref_patch_with_border + (y + 1) * ref_step + 1;
    for (int x = 0; x < patch_size_; ++x, ++it, ++it_dx, ++it_dy) {
      float J[4];
      J[0] = (float)(0.5 * (it[1] - it[-1]));
      J[1] = (float)(0.5 * (it[ref_step] - it[-ref_step]));
      J[2] = affine_est_offset ? 1.0f : 0.0f;
      J[3] = affine_est_gain ? (float)(-1.0 * it[0]) : 0.0f;
      *it_dx = J[0];
      *it_dy = J[1];
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) H[r * 4 + c] += J[r] * J[c];
    }
  }
  if (!affine_est_offset) H[2 * 4 + 2] = 1.0f;
  if (!affine_est_gain) H[3 * 4 + 3] = 1.0f;
  float Hinv[16];
  mat4f_inverse(H, Hinv);
  float mean_diff = 0;
  float alpha = 1.0;
  float u = (float)cur_px_estimate[0];
  float v = (float)cur_px_estimate[1];
  const float min_update_squared = (float)(0.03 * 0.03);
  const int cur_step = cur_img->pitch;
  float update[4] = { 0, 0, 0, 0 };
  for (int iter = 0; iter < n_iter; ++iter) {
    const int u_r = (int)floorf(u);
    const int v_r = (int)floorf(v);
    if (u_r < halfpatch_size_ || v_r < halfpatch_size_ || u_r >= cur_img->width - halfpatch_size_ ||
        v_r >= cur_img->height - halfpatch_size_)
      break;
    if (isnan(u) || isnan(v)) return 0;
    const float subpix_x = u - u_r;
    const float subpix_y = v - v_r;
    const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
    const float wTR = (float)(subpix_x * (1.0 - subpix_y));
    const float wBL = (float)((1.0 - subpix_x) * subpix_y);
    const float wBR = subpix_x * subpix_y;
    const uint8_t* it_ref = ref_patch;
    const float* it_ref_dx = ref_patch_dx;
    const float* it_ref_dy = ref_patch_dy;
    float Jres[4] = { 0, 0, 0, 0 };
    for (int y = 0; y < patch_size_; ++y) {
      const uint8_t* it = cur_img->data + (ptrdiff_t)(v_r + y - halfpatch_size_) * cur_step + u_r - halfpatch_size_;
      for (int x = 0; x < patch_size_; ++x, ++it, ++it_ref, ++it_ref_dx, ++it_ref_dy) {
        const float search_pixel = wTL * it[0] + wTR * it[1] + wBL * it[cur_step] + wBR * it[cur_step + 1];
        const float res = search_pixel - alpha * (*it_ref) + mean_diff;
        Jres[0] -= res * (*it_ref_dx);
        Jres[1] -= res * (*it_ref_dy);
        if (affine_est_offset) Jres[2] -= res;
        if (affine_est_gain) Jres[3] -= (-1) * res * (*it_ref);
      }
    }
    if (!affine_est_offset) Jres[2] = 0.0f;
    if (!affine_est_gain) Jres[3] = 0.0f;
    for (int r = 0; r < 4; ++r)
      update[r] = ((Hinv[r * 4 + 0] * Jres[0] + Hinv[r * 4 + 1] * Jres[1]) + Hinv[r * 4 + 2] * Jres[2]) + Hinv[r * 4 + 3] * Jres[3];
    u += update[0];
    v += update[1];
    mean_diff += update[2];
    alpha += update[3];
    if (update[0] * update[0] + update[1] * update[1] < min_update_squared) { converged = 1; break; }
  }
  cur_px_estimate[0] = u;
  cur_px_estimate[1] = v;
  return converged;
}

/* ---- a-13 Matcher ------------------------------------------------------------ */
typedef struct orc_matcher {
  svoh_matcher_options opt;
  int align_1d;
  uint8_t patch[64];
  uint8_t patch_with_border[100];
  double A_cur_ref[4];
  double epi_image[2];
  double epi_length_pyramid;
  double h_inv;
  int search_level;
  int reject;
  double px_cur[2];
  double f_cur[3];
} orc_matcher;

static int is_edgelet(int t) { return t == SVOH_FT_EDGELET || t == SVOH_FT_EDGELET_SEED || t == SVOH_FT_EDGELET_SEED_CONVERGED; }
static int is_seed(int t) { return t < 6; }

static void T_cur_ref_from_frames(const orc_frame_view* ref, const orc_frame_view* cur, svoh_se3* T)
{
  svoh_se3 inv;
  orc_se3_inverse(&ref->T_f_w, &inv);
  orc_se3_mul(&cur->T_f_w, &inv, T);
}

/* matcher.cpp:31-141 */
int orc_find_match_direct_lm(orc_matcher* m, const orc_frame_view* ref_frame, const orc_frame_view* cur_frame,
                             const double px_ref[2], const double f_ref[3], const double grad_ref[2], int level,
                             int type, double ref_depth, const double* landmark_xyz, double px_cur[2]);
int orc_find_match_direct(orc_matcher* m, const orc_frame_view* ref_frame, const orc_frame_view* cur_frame,
                          const double px_ref[2], const double f_ref[3], const double grad_ref[2], int level,
                          int type, double ref_depth, double px_cur[2])
{
  return orc_find_match_direct_lm(m, ref_frame, cur_frame, px_ref, f_ref, grad_ref, level, type, ref_depth, NULL, px_cur);
}

/* landmark_xyz != NULL: Matcher::Options::use_affine_warp_ == false (matcher.cpp:67-81), the patch by warpPixelwise */
int orc_find_match_direct_lm(orc_matcher* m, const orc_frame_view* ref_frame, const orc_frame_view* cur_frame,
                             const double px_ref[2], const double f_ref[3], const double grad_ref[2], int level,
                             int type, double ref_depth, const double* landmark_xyz, double px_cur[2])
{
  enum { kHalfPatchSize = 4, kPatchSize = 8 };
  const int pxi0 = (int)px_ref[0] / (1 << level), pxi1 = (int)px_ref[1] / (1 << level);
  const int boundary = kHalfPatchSize + 2;
  if (pxi0 < boundary || pxi1 < boundary || pxi0 >= (int)(ref_frame->cam.width / (1 << level)) - boundary ||
      pxi1 >= (int)(ref_frame->cam.height / (1 << level)) - boundary)
    return SVOH_MATCH_FAIL_VISIBILITY;
  svoh_se3 T_cur_ref;
  T_cur_ref_from_frames(ref_frame, cur_frame, &T_cur_ref);
  orc_get_warp_matrix_affine(&ref_frame->cam, &cur_frame->cam, px_ref, f_ref, ref_depth, &T_cur_ref, level, m->A_cur_ref);
  m->search_level = orc_get_best_search_level(m->A_cur_ref, ref_frame->pyr.n_levels - 1);
  if (landmark_xyz) {
    if (!orc_warp_pixelwise(cur_frame, ref_frame, px_ref, landmark_xyz, level, m->search_level, kHalfPatchSize + 1,
                            m->patch_with_border))
      return SVOH_MATCH_FAIL_WARP;
  } else if (!orc_warp_affine(m->A_cur_ref, &ref_frame->pyr.level[level], px_ref, level, m->search_level, kHalfPatchSize + 1,
                              m->patch_with_border))
    return SVOH_MATCH_FAIL_WARP;
  create_patch_from_patch_with_border(m->patch_with_border, kPatchSize, m->patch);
  double px_scaled[2] = { px_cur[0] / (1 << m->search_level), px_cur[1] / (1 << m->search_level) };
  const double px_scaled_start[2] = { px_scaled[0], px_scaled[1] };
  int ok;
  if (is_edgelet(type)) {
    double dir_cur[2] = { m->A_cur_ref[0] * grad_ref[0] + m->A_cur_ref[2] * grad_ref[1],
                          m->A_cur_ref[1] * grad_ref[0] + m->A_cur_ref[3] * grad_ref[1] };
    normalize2(dir_cur);
    ok = orc_align_1d(&cur_frame->pyr.level[m->search_level], dir_cur, m->patch_with_border, m->patch,
                      m->opt.align_max_iter, m->opt.affine_est_offset, m->opt.affine_est_gain, px_scaled, &m->h_inv);
  } else {
    ok = orc_align_2d(&cur_frame->pyr.level[m->search_level], m->patch_with_border, m->patch, m->opt.align_max_iter,
                      m->opt.affine_est_offset, m->opt.affine_est_gain, px_scaled);
  }
  if (!ok) return SVOH_MATCH_FAIL_ALIGNMENT;
  const double dx = px_scaled[0] - px_scaled_start[0], dy = px_scaled[1] - px_scaled_start[1];
  if (sqrt(dx * dx + dy * dy) > m->opt.max_patch_diff_ratio * kPatchSize) return SVOH_MATCH_FAIL_TOO_FAR;
  px_cur[0] = px_scaled[0] * (1 << m->search_level);
  px_cur[1] = px_scaled[1] * (1 << m->search_level);
  m->px_cur[0] = px_cur[0]; m->px_cur[1] = px_cur[1];
  orc_back_project3(&cur_frame->cam, m->px_cur, m->f_cur);
  normalize3(m->f_cur);
  return SVOH_MATCH_SUCCESS;
}

/* matcher.cpp:262-289 */
static int find_local_match(orc_matcher* m, const orc_frame_view* frame, const double direction[2], int patch_level,
                            double px_cur[2])
{
  double px_scaled[2] = { px_cur[0] / (1 << patch_level), px_cur[1] / (1 << patch_level) };
  int res;
  if (m->align_1d)
    res = orc_align_1d(&frame->pyr.level[patch_level], direction, m->patch_with_border, m->patch, m->opt.align_max_iter,
                       m->opt.affine_est_offset, m->opt.affine_est_gain, px_scaled, &m->h_inv);
  else
    res = orc_align_2d(&frame->pyr.level[patch_level], m->patch_with_border, m->patch, m->opt.align_max_iter,
                       m->opt.affine_est_offset, m->opt.affine_est_gain, px_scaled);
  if (!res) return SVOH_MATCH_FAIL_ALIGNMENT;
  px_cur[0] = px_scaled[0] * (1 << patch_level);
  px_cur[1] = px_scaled[1] * (1 << patch_level);
  return SVOH_MATCH_SUCCESS;
}

/* matcher.cpp:292-322 */
static int update_zmssd(const orc_frame_view* frame, const int pxi[2], int patch_level, const uint8_t* ref_patch,
                        int* zmssd_best)
{
  const orc_image* im = &frame->pyr.level[patch_level];
  const uint8_t* cur_patch_ptr = im->data + (ptrdiff_t)(pxi[1] - 4) * im->pitch + (pxi[0] - 4);
  const int zmssd = orc_zmssd_score(ref_patch, cur_patch_ptr, im->pitch);
  if (zmssd < *zmssd_best) { *zmssd_best = zmssd; return 1; }
  return 0;
}

static int is_patch_within_image(const orc_frame_view* frame, const int pxi[2], int patch_level)
{
  enum { kPatchSize = 8 };
  return !(pxi[0] < kPatchSize || pxi[1] < kPatchSize ||
           pxi[0] >= ((int)(frame->cam.width / (1 << patch_level)) - kPatchSize) ||
           pxi[1] >= ((int)(frame->cam.height / (1 << patch_level)) - kPatchSize));
}

/* matcher.cpp:340-413 */
static void scan_epipolar_unit_plane(orc_matcher* m, const orc_frame_view* frame, const double A[3], const double B[3],
                                     const double C[3], int patch_level, double image_best[2], int* zmssd_best)
{
  size_t n_steps = (size_t)(m->epi_length_pyramid / 0.7);
  const double pa[2] = { A[0] / A[2], A[1] / A[2] }, pb[2] = { B[0] / B[2], B[1] / B[2] };
  double step[2] = { (pa[0] - pb[0]) / n_steps, (pa[1] - pb[1]) / n_steps };
  if (n_steps > (size_t)m->opt.max_epi_search_steps) n_steps = (size_t)m->opt.max_epi_search_steps;
  const double uv_C[2] = { C[0] / C[2], C[1] / C[2] };
  double uv[2] = { uv_C[0], uv_C[1] };
  double uv_best[2] = { uv[0], uv[1] };
  int forward = 1;
  int last_checked_pxi[2] = { 0, 0 };
  for (size_t i = 0; i < n_steps; ++i, uv[0] += step[0], uv[1] += step[1]) {
    double px[2];
    const double p3[3] = { uv[0], uv[1], 1.0 };
    orc_project3(&frame->cam, p3, px, NULL);
    const int pxi[2] = { (int)(px[0] / (1 << patch_level) + 0.5), (int)(px[1] / (1 << patch_level) + 0.5) };
    if (pxi[0] == last_checked_pxi[0] && pxi[1] == last_checked_pxi[1]) continue;
    last_checked_pxi[0] = pxi[0]; last_checked_pxi[1] = pxi[1];
    if (!is_patch_within_image(frame, pxi, patch_level)) {
      if (forward) {
        i = (size_t)(n_steps * 0.5);
        step[0] = -step[0]; step[1] = -step[1];
        uv[0] = uv_C[0]; uv[1] = uv_C[1];
        forward = 0;
        continue;
      } else
        break;
    }
    if (update_zmssd(frame, pxi, patch_level, m->patch, zmssd_best)) { uv_best[0] = uv[0]; uv_best[1] = uv[1]; }
    if (forward && i > n_steps * 0.5) {
      step[0] = -step[0]; step[1] = -step[1];
      uv[0] = uv_C[0]; uv[1] = uv_C[1];
      forward = 0;
    }
  }
  const double p3[3] = { uv_best[0], uv_best[1], 1.0 };
  orc_project3(&frame->cam, p3, image_best, NULL);
}

/* Eigen AngleAxis::toRotationMatrix() * v */
static void angle_axis_rotate(const double axis[3], double angle, const double v[3], double out[3])
{
  const double s = sin(angle), c = cos(angle);
  const double sin_axis[3] = { s * axis[0], s * axis[1], s * axis[2] };
  const double cos1_axis[3] = { (1.0 - c) * axis[0], (1.0 - c) * axis[1], (1.0 - c) * axis[2] };
  double R[9];
  double tmp;
  tmp = cos1_axis[0] * axis[1]; R[0 * 3 + 1] = tmp - sin_axis[2]; R[1 * 3 + 0] = tmp + sin_axis[2];
  tmp = cos1_axis[0] * axis[2]; R[0 * 3 + 2] = tmp + sin_axis[1]; R[2 * 3 + 0] = tmp - sin_axis[1];
  tmp = cos1_axis[1] * axis[2]; R[1 * 3 + 2] = tmp - sin_axis[0]; R[2 * 3 + 1] = tmp + sin_axis[0];
  R[0] = cos1_axis[0] * axis[0] + c; R[4] = cos1_axis[1] * axis[1] + c; R[8] = cos1_axis[2] * axis[2] + c;
  for (int r = 0; r < 3; ++r) out[r] = (R[r * 3 + 0] * v[0] + R[r * 3 + 1] * v[1]) + R[r * 3 + 2] * v[2];
}

/* matcher.cpp:415-488 */
static void scan_epipolar_unit_sphere(orc_matcher* m, const orc_frame_view* frame, const double A[3], const double B[3],
                                      const double C[3], int patch_level, double image_best[2], int* zmssd_best)
{
  size_t n_steps = (size_t)(m->epi_length_pyramid / 0.7);
  n_steps = n_steps > (size_t)m->opt.max_epi_search_steps ? (size_t)m->opt.max_epi_search_steps : n_steps;
  const size_t half_steps = n_steps / 2;
  double f_A[3] = { A[0], A[1], A[2] }, f_B[3] = { B[0], B[1], B[2] }, f_C[3] = { C[0], C[1], C[2] };
  normalize3(f_A); normalize3(f_B); normalize3(f_C);
  const double step = acos((f_A[0] * f_B[0] + f_A[1] * f_B[1]) + f_A[2] * f_B[2]) / n_steps;
  double axis[3] = { f_B[1] * f_A[2] - f_B[2] * f_A[1], f_B[2] * f_A[0] - f_B[0] * f_A[2], f_B[0] * f_A[1] - f_B[1] * f_A[0] };
  normalize3(axis);
  double f[3] = { f_C[0], f_C[1], f_C[2] };
  double f_best[3] = { f_C[0], f_C[1], f_C[2] };
  int last_checked_pxi[2] = { 0, 0 };
  for (size_t i = 0; i < n_steps; i++) {
    double angle;
    if (i < half_steps) angle = i * step;
    else angle = (i - half_steps) * (-step);
    angle_axis_rotate(axis, angle, f_C, f);
    double px[2];
    orc_project3(&frame->cam, f, px, NULL);
    const int pxi[2] = { (int)(px[0] / (1 << patch_level) + 0.5), (int)(px[1] / (1 << patch_level) + 0.5) };
    if (pxi[0] == last_checked_pxi[0] && pxi[1] == last_checked_pxi[1]) continue;
    last_checked_pxi[0] = pxi[0]; last_checked_pxi[1] = pxi[1];
    if (!is_patch_within_image(frame, pxi, patch_level)) {
      if (i < half_steps) { i = half_steps; continue; }
      else break;
    }
    if (update_zmssd(frame, pxi, patch_level, m->patch, zmssd_best)) { f_best[0] = f[0]; f_best[1] = f[1]; f_best[2] = f[2]; }
  }
  orc_project3(&frame->cam, f_best, image_best, NULL);
}

/* matcher.cpp:492-505 */
static int depth_from_triangulation(const svoh_se3* T_search_ref, const double f_ref[3], const double f_cur[3], double* depth)
{
  double a[3];
  orc_quat_rotate(T_search_ref->q, f_ref, a);
  const double* b = f_cur;
  const double AtA[4] = { (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2], (b[0] * a[0] + b[1] * a[1]) + b[2] * a[2],
                          (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2], (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2] }; /* col-major */
  if (AtA[0] * AtA[3] - AtA[1] * AtA[2] < 0.000001) return SVOH_MATCH_FAIL_TRIANGULATION;
  double inv[4];
  mat2d_inverse(AtA, inv);
  /* (-inv) * A^T (2x3), then * t */
  double Mrow0[3], Mrow1[3];
  for (int k = 0; k < 3; ++k) {
    Mrow0[k] = (-inv[0]) * a[k] + (-inv[2]) * b[k];
    Mrow1[k] = (-inv[1]) * a[k] + (-inv[3]) * b[k];
  }
  const double* t = T_search_ref->t;
  const double d0 = (Mrow0[0] * t[0] + Mrow0[1] * t[1]) + Mrow0[2] * t[2];
  (void)Mrow1;
  *depth = fabs(d0);
  return SVOH_MATCH_SUCCESS;
}

/* matcher.cpp:157-241 */
int orc_find_epipolar_match_direct(orc_matcher* m, const orc_frame_view* ref_frame, const orc_frame_view* cur_frame,
                                   const svoh_se3* T_cur_ref, const double px_ref[2], const double f_ref[3],
                                   const double grad_ref[2], int level, int type, double d_estimate_inv,
                                   double d_min_inv, double d_max_inv, double* depth)
{
  enum { kHalfPatchSize = 4, kPatchSize = 8 };
  int zmssd_best = ZMSSD_THRESHOLD;
  double Rf[3];
  orc_quat_rotate(T_cur_ref->q, f_ref, Rf);
  const double A[3] = { Rf[0] + T_cur_ref->t[0] * d_min_inv, Rf[1] + T_cur_ref->t[1] * d_min_inv, Rf[2] + T_cur_ref->t[2] * d_min_inv };
  const double B[3] = { Rf[0] + T_cur_ref->t[0] * d_max_inv, Rf[1] + T_cur_ref->t[1] * d_max_inv, Rf[2] + T_cur_ref->t[2] * d_max_inv };
  double px_A[2], px_B[2];
  orc_project3(&cur_frame->cam, A, px_A, NULL);
  orc_project3(&cur_frame->cam, B, px_B, NULL);
  m->epi_image[0] = px_A[0] - px_B[0];
  m->epi_image[1] = px_A[1] - px_B[1];
  orc_get_warp_matrix_affine(&ref_frame->cam, &cur_frame->cam, px_ref, f_ref, 1.0 / fmax(0.000001, d_estimate_inv),
                             T_cur_ref, level, m->A_cur_ref);
  m->reject = 0;
  if (is_edgelet(type) && m->opt.epi_search_edgelet_filtering) {
    double grad_cur[2] = { m->A_cur_ref[0] * grad_ref[0] + m->A_cur_ref[2] * grad_ref[1],
                           m->A_cur_ref[1] * grad_ref[0] + m->A_cur_ref[3] * grad_ref[1] };
    normalize2(grad_cur);
    double en[2] = { m->epi_image[0], m->epi_image[1] };
    normalize2(en);
    const double cosangle = fabs(grad_cur[0] * en[0] + grad_cur[1] * en[1]);
    if (cosangle < m->opt.epi_search_edgelet_max_angle) {
      m->reject = 1;
      return SVOH_MATCH_FAIL_ANGLE;
    }
  }
  m->search_level = orc_get_best_search_level(m->A_cur_ref, ref_frame->pyr.n_levels - 1);
  m->epi_length_pyramid = sqrt(m->epi_image[0] * m->epi_image[0] + m->epi_image[1] * m->epi_image[1]) / (1 << m->search_level);
  double epi_dir_image[2] = { m->epi_image[0], m->epi_image[1] };
  normalize2(epi_dir_image);
  if (!orc_warp_affine(m->A_cur_ref, &ref_frame->pyr.level[level], px_ref, level, m->search_level, kHalfPatchSize + 1,
                       m->patch_with_border))
    return SVOH_MATCH_FAIL_WARP;
  create_patch_from_patch_with_border(m->patch_with_border, kPatchSize, m->patch);

  if (m->epi_length_pyramid < 2.0) {
    m->px_cur[0] = (px_A[0] + px_B[0]) / 2.0;
    m->px_cur[1] = (px_A[1] + px_B[1]) / 2.0;
    const int res = find_local_match(m, cur_frame, epi_dir_image, m->search_level, m->px_cur);
    if (res != SVOH_MATCH_SUCCESS) return res;
    orc_back_project3(&cur_frame->cam, m->px_cur, m->f_cur);
    normalize3(m->f_cur);
    return depth_from_triangulation(T_cur_ref, f_ref, m->f_cur, depth);
  }

  const double C[3] = { Rf[0] + T_cur_ref->t[0] * d_estimate_inv, Rf[1] + T_cur_ref->t[1] * d_estimate_inv,
                        Rf[2] + T_cur_ref->t[2] * d_estimate_inv };
  if (m->opt.scan_on_unit_sphere)
    scan_epipolar_unit_sphere(m, cur_frame, A, B, C, m->search_level, m->px_cur, &zmssd_best);
  else
    scan_epipolar_unit_plane(m, cur_frame, A, B, C, m->search_level, m->px_cur, &zmssd_best);

  if (zmssd_best < ZMSSD_THRESHOLD) {
    if (m->opt.subpix_refinement) {
      const int res = find_local_match(m, cur_frame, epi_dir_image, m->search_level, m->px_cur);
      if (res != SVOH_MATCH_SUCCESS) return res;
    }
    orc_back_project3(&cur_frame->cam, m->px_cur, m->f_cur);
    normalize3(m->f_cur);
    return depth_from_triangulation(T_cur_ref, f_ref, m->f_cur, depth);
  }
  return SVOH_MATCH_FAIL_SCORE;
}

/* ---- a-14 depth filter -------------------------------------------------------- */
/* math_utils.h:186-194 */
static double norm_pdf(double x, double mean, double sigma)
{
  double exponent = x - mean;
  exponent *= -exponent;
  exponent /= 2 * sigma * sigma;
  double result = exp(exponent);
  result /= sigma * sqrt(2 * M_PI);
  return result;
}

/* depth_filter.cpp:501-552 */
static int update_filter_vogiatzis(double z, double tau2, double mu_range, double st[4])
{
  double* mu = &st[0]; double* sigma2 = &st[1]; double* a = &st[2]; double* b = &st[3];
  const double norm_scale = sqrt(*sigma2 + tau2);
  if (isnan(norm_scale)) return 0;
  const double oldsigma2 = *sigma2;
  const double s2 = 1.0 / (1.0 / *sigma2 + 1.0 / tau2);
  const double m = s2 * (*mu / *sigma2 + z / tau2);
  const double uniform_x = 1.0 / mu_range;
  double C1 = *a / (*a + *b) * norm_pdf(z, *mu, norm_scale);
  double C2 = *b / (*a + *b) * uniform_x;
  const double normalization_constant = C1 + C2;
  C1 /= normalization_constant;
  C2 /= normalization_constant;
  const double f = C1 * (*a + 1.0) / (*a + *b + 1.0) + C2 * *a / (*a + *b + 1.0);
  const double e = C1 * (*a + 1.0) * (*a + 2.0) / ((*a + *b + 1.0) * (*a + *b + 2.0)) +
                   C2 * *a * (*a + 1.0) / ((*a + *b + 1.0) * (*a + *b + 2.0));
  const double mu_new = C1 * m + C2 * *mu;
  *sigma2 = C1 * (s2 + m * m) + C2 * (*sigma2 + *mu * *mu) - mu_new * mu_new;
  *mu = mu_new;
  *a = (e - f) / (f - e / f);
  *b = *a * (1.0 - f) / f;
  if (*sigma2 < 0.0) *sigma2 = oldsigma2;
  if (*mu < 0.0) { *mu = 1.0; return 0; }
  return 1;
}

/* depth_filter.cpp:554-578 */
static int update_filter_gaussian(double z, double tau2, double st[4])
{
  const double norm_scale = sqrt(st[1] + tau2);
  if (isnan(norm_scale)) return 0;
  const double denom = (st[1] + tau2);
  st[0] = (st[1] * z + tau2 * st[0]) / denom;
  st[1] = st[1] * tau2 / denom;
  return 1;
}

/* depth_filter.cpp:580-596 */
static double compute_tau(const svoh_se3* T_ref_cur, const double f[3], double z, double px_error_angle)
{
  const double* t = T_ref_cur->t;
  const double a[3] = { f[0] * z - t[0], f[1] * z - t[1], f[2] * z - t[2] };
  const double t_norm = sqrt((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
  const double a_norm = sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
  const double alpha = acos(((f[0] * t[0] + f[1] * t[1]) + f[2] * t[2]) / t_norm);
  const double beta = acos(((a[0] * -t[0] + a[1] * -t[1]) + a[2] * -t[2]) / (t_norm * a_norm));
  const double beta_plus = beta + px_error_angle;
  const double gamma_plus = M_PI - alpha - beta_plus;
  const double z_plus = t_norm * sin(beta_plus) / sin(gamma_plus);
  return (z_plus - z);
}

/* depth_filter.cpp:367-499; st = [mu, sigma2, a, b]; type in/out; match_result out */
int orc_update_seed(orc_matcher* m, const svoh_depth_filter_options* opt, const orc_frame_view* cur_frame,
                    const orc_frame_view* ref_frame, const double px_ref[2], const double f_ref[3],
                    const double grad_ref[2], int level, uint8_t* type_io, double st[4],
                    double sigma2_convergence_threshold, int* match_result)
{
  *match_result = SVOH_MATCH_NOT_RUN;
  if (cur_frame->id == ref_frame->id) return 0;
  const int type = *type_io;
  if (type == SVOH_FT_OUTLIER) return 0;
  if ((type == SVOH_FT_CORNER_SEED_CONVERGED || type == SVOH_FT_EDGELET_SEED_CONVERGED ||
       type == SVOH_FT_MAPPOINT_SEED_CONVERGED) && opt->check_convergence)
    return 0;
  svoh_se3 T_cur_ref;
  T_cur_ref_from_frames(ref_frame, cur_frame, &T_cur_ref);
  if (opt->check_visibility) {
    const double depth = 1.0 / st[0];
    const double p[3] = { depth * f_ref[0], depth * f_ref[1], depth * f_ref[2] };
    double xyz_f[3], px[2];
    orc_se3_transform(&T_cur_ref, p, xyz_f);
    orc_project3(&cur_frame->cam, xyz_f, px, NULL);
    if (!(px[0] >= 0.0 && px[1] >= 0.0 && px[0] < (double)cur_frame->cam.width && px[1] < (double)cur_frame->cam.height))
      return 0;
    const int pxi0 = (int)px[0], pxi1 = (int)px[1];
    const int boundary = 9;
    if (!(pxi0 >= boundary && pxi1 >= boundary && pxi0 < cur_frame->cam.width - boundary &&
          pxi1 < cur_frame->cam.height - boundary))
      return 0;
  }
  m->align_1d = (type == SVOH_FT_EDGELET_SEED || type == SVOH_FT_EDGELET_SEED_CONVERGED);
  double depth;
  const double inv_min = st[0] + sqrt(st[1]);                 /* seed.h:120-123 getInvMinDepth */
  const double inv_max = fmax(st[0] - sqrt(st[1]), 0.00000001); /* seed.h:125-128 getInvMaxDepth */
  const int res = orc_find_epipolar_match_direct(m, ref_frame, cur_frame, &T_cur_ref, px_ref, f_ref, grad_ref, level, type,
                                                 st[0], inv_min, inv_max, &depth);
  *match_result = res;
  if (res != SVOH_MATCH_SUCCESS) {
    if (!m->reject) st[3] += 1;  /* increaseOutlierProbability */
    return 0;
  }
  svoh_se3 T_ref_cur;
  orc_se3_inverse(&T_cur_ref, &T_ref_cur);
  const double depth_sigma = compute_tau(&T_ref_cur, f_ref, depth, opt->px_error_angle);
  const double z = 1.0 / depth;  /* getMeanFromDepth */
  const double sg = 0.5 * (1.0 / fmax(0.000000000001, depth - depth_sigma) - 1.0 / (depth + depth_sigma));
  const double tau2 = sg * sg;   /* getSigma2FromDepthSigma */
  if (opt->use_vogiatzis_update) {
    if (!update_filter_vogiatzis(z, tau2, ref_frame->seed_mu_range, st)) { *type_io = SVOH_FT_OUTLIER; return 0; }
  } else {
    if (!update_filter_gaussian(z, tau2, st)) { *type_io = SVOH_FT_OUTLIER; return 0; }
  }
  const double thresh = ref_frame->seed_mu_range / sigma2_convergence_threshold;  /* seed.h:143-151 */
  if (st[1] < thresh * thresh) {
    if (type == SVOH_FT_CORNER_SEED) *type_io = SVOH_FT_CORNER_SEED_CONVERGED;
    else if (type == SVOH_FT_EDGELET_SEED) *type_io = SVOH_FT_EDGELET_SEED_CONVERGED;
    else if (type == SVOH_FT_MAPPOINT_SEED) *type_io = SVOH_FT_MAPPOINT_SEED_CONVERGED;
  }
  return 1;
}

/* ---- batch entry points with the product ABI's semantics ---------------------- */
void orc_match_direct_batch(const svoh_matcher_options* options, int n_ref_frames, const orc_frame_view* ref_frames,
                            const orc_frame_view* cur_frame, const orc_feature_batch* fb, const double* depth,
                            double* px_cur, int32_t* result, double* f_cur, int32_t* search_level, double* h_inv,
                            double* A_cur_ref)
{
  orc_match_direct_batch_ex(options, n_ref_frames, ref_frames, cur_frame, fb, depth, NULL, px_cur, result, f_cur, search_level,
                            h_inv, A_cur_ref);
}

/* landmark_xyz (3 x n, world) != NULL: use_affine_warp_ == false */
void orc_match_direct_batch_ex(const svoh_matcher_options* options, int n_ref_frames, const orc_frame_view* ref_frames,
                               const orc_frame_view* cur_frame, const orc_feature_batch* fb, const double* depth,
                               const double* landmark_xyz, double* px_cur, int32_t* result, double* f_cur,
                               int32_t* search_level, double* h_inv, double* A_cur_ref)
{
  (void)n_ref_frames;
  for (int i = 0; i < fb->n; ++i) {
    orc_matcher m;
    memset(&m, 0, sizeof m);
    m.opt = *options;
    double p[2] = { px_cur[2 * i], px_cur[2 * i + 1] };
    const orc_frame_view* cf = fb->cur_frame_idx ? &cur_frame[fb->cur_frame_idx[i]] : cur_frame;
    const int r = orc_find_match_direct_lm(&m, &ref_frames[fb->ref_frame_idx[i]], cf, &fb->px[2 * i], &fb->f[3 * i],
                                           &fb->grad[2 * i], fb->level[i], fb->type[i], depth[i],
                                           landmark_xyz ? &landmark_xyz[3 * i] : NULL, p);
    result[i] = r;
    px_cur[2 * i] = p[0]; px_cur[2 * i + 1] = p[1];
    if (f_cur) { f_cur[3 * i] = m.f_cur[0]; f_cur[3 * i + 1] = m.f_cur[1]; f_cur[3 * i + 2] = m.f_cur[2]; }
    if (search_level) search_level[i] = m.search_level;
    if (h_inv) h_inv[i] = m.h_inv;
    if (A_cur_ref) for (int k = 0; k < 4; ++k) A_cur_ref[4 * i + k] = m.A_cur_ref[k];
  }
}

/* DepthFilter::updateSeeds, synchronous branch (depth_filter.cpp:200-233) */
int orc_update_seeds_batch(const svoh_matcher_options* mopt, const svoh_depth_filter_options* opt, int n_ref_frames,
                           const orc_frame_view* ref_frames, const orc_frame_view* cur_frame,
                           const orc_feature_batch* fb, double* state, uint8_t* success, int32_t* match_result)
{
  (void)n_ref_frames;
  int n_success = 0;
  for (int i = 0; i < fb->n; ++i) {
    success[i] = 0;
    if (match_result) match_result[i] = SVOH_MATCH_NOT_RUN;
    const int type = fb->type[i];
    if (!is_seed(type)) continue;
    double cur_thresh = opt->seed_convergence_sigma2_thresh;
    if (type == SVOH_FT_MAPPOINT_SEED || type == SVOH_FT_MAPPOINT_SEED_CONVERGED)
      cur_thresh = opt->mappoint_convergence_sigma2_thresh;
    orc_matcher m;
    memset(&m, 0, sizeof m);
    m.opt = *mopt;
    int mr;
    const orc_frame_view* cf = fb->cur_frame_idx ? &cur_frame[fb->cur_frame_idx[i]] : cur_frame;
    const int ok = orc_update_seed(&m, opt, cf, &ref_frames[fb->ref_frame_idx[i]], &fb->px[2 * i], &fb->f[3 * i],
                                   &fb->grad[2 * i], fb->level[i], &fb->type[i], &state[4 * i], cur_thresh, &mr);
    if (match_result) match_result[i] = mr;
    success[i] = (uint8_t)ok;
    n_success += ok;
  }
  return n_success;
}

/* ---- reprojector_utils::matchCandidates (reprojector.cpp:342-382) + matchCandidate (:384-486) ---- */
int orc_match_candidates(const svoh_matcher_options* mopt, const svoh_depth_filter_options* dopt, int n_ref_frames,
                         const orc_frame_view* ref_frames, const orc_frame_view* cur_frame, int n_candidates,
                         orc_candidate* cands, int max_n_features_per_frame, int* num_features_io, int cell_size,
                         int n_cols, int n_rows, uint8_t* occupancy, uint8_t* visited, int32_t* result,
                         orc_new_feature* out, int* n_out, int* n_trials, int* n_matches,
                         int* n_failed_reproj, int* n_succeeded_reproj)
{
  (void)n_ref_frames; (void)n_rows;
  int i = 0;
  *n_out = 0;
  for (int k = 0; k < n_candidates; ++k) { visited[k] = 0; result[k] = -1; }
  for (int k = 0; k < n_candidates; ++k) {
    orc_candidate* c = &cands[k];
    ++i;
    /* grid.getCellIndex(candidate.cur_px.x(), candidate.cur_px.y(), 1): the doubles narrow to the int
     * parameters (occupancy_grid_2d.h:91-94), then floor(y / cell_size) * n_cols + floor(x / cell_size) */
    const int xi = (int)c->cur_px[0], yi = (int)c->cur_px[1];
    const size_t grid_index = (size_t)(floor((double)yi / cell_size) * n_cols + floor((double)xi / cell_size));
    if (max_n_features_per_frame > 0 && occupancy[grid_index]) continue;
    ++*n_trials;
    visited[k] = 1;
    orc_matcher m;
    memset(&m, 0, sizeof m);
    m.opt = *mopt;
    const orc_frame_view* rf = &ref_frames[c->ref_frame_idx];
    int ok = 0;
    if (c->kind == 0 || c->kind == 2) {
      const int r = orc_find_match_direct(&m, rf, cur_frame, c->px, c->f, c->grad, c->level, c->ref_type, c->depth, c->cur_px);
      result[k] = r;
      ok = r == SVOH_MATCH_SUCCESS;
      if (c->kind == 2) { if (ok) ++*n_succeeded_reproj; else ++*n_failed_reproj; }
    } else if (c->kind == 1) {
      int mr;
      /* updateSeed(*frame, *c.ref_frame, c.ref_index, matcher, seed_sigma2_thresh, false, false) */
      ok = orc_update_seed(&m, dopt, cur_frame, rf, c->px, c->f, c->grad, c->level, &c->ref_type, c->state,
                           dopt->seed_convergence_sigma2_thresh, &mr);
      result[k] = mr;
    } else {
      result[k] = 1000;  /* getCloseViewObs returned false */
    }
    if (!ok) continue;
    orc_new_feature* o = &out[(*n_out)++];
    memset(o, 0, sizeof *o);
    o->candidate = k;
    if (is_edgelet(c->type)) {  /* (matcher.A_cur_ref_ * grad_ref).normalized() */
      double g0 = m.A_cur_ref[0] * c->grad[0] + m.A_cur_ref[2] * c->grad[1];
      double g1 = m.A_cur_ref[1] * c->grad[0] + m.A_cur_ref[3] * c->grad[1];
      const double z = g0 * g0 + g1 * g1;
      if (z > 0.0) { const double nn = sqrt(z); g0 /= nn; g1 /= nn; }
      o->grad[0] = g0; o->grad[1] = g1;
    }
    o->type = c->type;
    o->px[0] = m.px_cur[0]; o->px[1] = m.px_cur[1];
    for (int j = 0; j < 3; ++j) o->f[j] = m.f_cur[j];
    o->level = m.search_level;
    o->score = c->score;
    for (int j = 0; j < 4; ++j) o->state[j] = c->state[j];
    ++*n_matches;
    ++*num_features_io;
    occupancy[grid_index] = 1;
    if (max_n_features_per_frame > 0 && *num_features_io >= max_n_features_per_frame) break;
  }
  return i;
}

/* ---- stereo seam: n x Matcher::findEpipolarMatchDirect with an explicit T_cur_ref, as
 * StereoTriangulation::compute calls it (src/svo/src/stereo_triangulation.cpp:92-104: a fresh Matcher,
 * max_epi_search_steps = 500, subpix_refinement on, align_1d = isEdgelet(type)); the 7-argument overload
 * (matcher.cpp:143-155) when T_cur_ref is NULL.  d_inv = [estimate, min, max] inverse depths, common to all
 * features (d_inv_common) or 3 per feature (d_inv, may be NULL). ---- */
void orc_epipolar_match_batch(const svoh_matcher_options* options, int n_ref_frames, const orc_frame_view* ref_frames,
                              const orc_frame_view* cur_frame, const svoh_se3* T_cur_ref, const orc_feature_batch* fb,
                              const double d_inv_common[3], const double* d_inv, int32_t* result, double* depth,
                              double* px_cur, double* f_cur, int32_t* search_level, double* h_inv, double* A_cur_ref)
{
  (void)n_ref_frames;
  const int n_cur = (fb->cur_frame_idx && fb->n_cur_frames > 0) ? fb->n_cur_frames : 1;
  for (int i = 0; i < fb->n; ++i) {
    orc_matcher m;
    memset(&m, 0, sizeof m);
    m.opt = *options;
    m.align_1d = is_edgelet(fb->type[i]);
    const int ci = fb->cur_frame_idx ? fb->cur_frame_idx[i] : 0;
    const orc_frame_view* rf = &ref_frames[fb->ref_frame_idx[i]];
    const orc_frame_view* cf = &cur_frame[ci];
    svoh_se3 T;
    if (T_cur_ref) T = T_cur_ref[(size_t)fb->ref_frame_idx[i] * n_cur + ci];
    else T_cur_ref_from_frames(rf, cf, &T);
    const double* d = d_inv ? &d_inv[3 * i] : d_inv_common;
    double z = 0.0;
    result[i] = orc_find_epipolar_match_direct(&m, rf, cf, &T, &fb->px[2 * i], &fb->f[3 * i], &fb->grad[2 * i], fb->level[i],
                                               fb->type[i], d[0], d[1], d[2], &z);
    depth[i] = z;
    if (px_cur) { px_cur[2 * i] = m.px_cur[0]; px_cur[2 * i + 1] = m.px_cur[1]; }
    if (f_cur) { f_cur[3 * i] = m.f_cur[0]; f_cur[3 * i + 1] = m.f_cur[1]; f_cur[3 * i + 2] = m.f_cur[2]; }
    if (search_level) search_level[i] = m.search_level;
    if (h_inv) h_inv[i] = m.h_inv;
    if (A_cur_ref) for (int k = 0; k < 4; ++k) A_cur_ref[4 * i + k] = m.A_cur_ref[k];
  }
}

/* StereoTriangulation::compute, the loop over the shuffled new features (stereo_triangulation.cpp:86-139):
 * indices[n_indices] = features of frame0 in visiting order; stops once n_desired have been triangulated.
 * out[k] for the k-th success: i_ref, xyz in frame0's camera frame (f * depth; the caller applies T_world_cam),
 * and what frame1 receives (px, f, normalised A_cur_ref * grad).  Returns the number of successes. */
int orc_stereo_triangulate(const orc_frame_view* frame0, const orc_frame_view* frame1, const svoh_se3* T_f1f0,
                           const orc_feature_batch* fb, int n_indices, const int32_t* indices, int n_desired,
                           const double d_inv[3], orc_stereo_match* out, int32_t* result, int* n_failed)
{
  int n_succeeded = 0;
  *n_failed = 0;
  for (int i = 0; i < fb->n; ++i) result[i] = -1;   /* not visited */
  for (int k = 0; k < n_indices; ++k) {
    const int i = indices[k];
    orc_matcher m;
    memset(&m, 0, sizeof m);
    m.opt.align_max_iter = 10; m.opt.max_epi_search_steps = 500; m.opt.subpix_refinement = 1;   /* matcher.h:39-54 + :93-94 */
    m.opt.epi_search_edgelet_filtering = 1; m.opt.scan_on_unit_sphere = 1; m.opt.affine_est_offset = 1;
    m.opt.affine_est_gain = 0; m.opt.epi_search_edgelet_max_angle = 0.7; m.opt.max_patch_diff_ratio = 2.0;
    m.align_1d = is_edgelet(fb->type[i]);
    double depth = 0.0;
    const int res = orc_find_epipolar_match_direct(&m, frame0, frame1, T_f1f0, &fb->px[2 * i], &fb->f[3 * i], &fb->grad[2 * i],
                                                   fb->level[i], fb->type[i], d_inv[0], d_inv[1], d_inv[2], &depth);
    result[i] = res;
    if (res == SVOH_MATCH_SUCCESS) {
      orc_stereo_match* o = &out[n_succeeded];
      memset(o, 0, sizeof *o);
      o->i_ref = i;
      for (int j = 0; j < 3; ++j) o->xyz_cam0[j] = fb->f[3 * i + j] * depth;
      o->px[0] = m.px_cur[0]; o->px[1] = m.px_cur[1];
      for (int j = 0; j < 3; ++j) o->f[j] = m.f_cur[j];
      double g[2] = { m.A_cur_ref[0] * fb->grad[2 * i] + m.A_cur_ref[2] * fb->grad[2 * i + 1],
                      m.A_cur_ref[1] * fb->grad[2 * i] + m.A_cur_ref[3] * fb->grad[2 * i + 1] };
      normalize2(g);
      o->grad[0] = g[0]; o->grad[1] = g[1];
      o->depth = depth;
      ++n_succeeded;
    } else {
      ++*n_failed;
    }
    if (n_succeeded >= n_desired) break;
  }
  return n_succeeded;
}

/* test hook: Matrix4f::inverse() in the given mode (row-major in and out) */
void orc_mat4f_inverse(const float m[16], float r[16], int mode)
{
  const int old = g_inv4_mode;
  g_inv4_mode = mode;
  mat4f_inverse(m, r);
  g_inv4_mode = old;
}


/* reprojector.cpp:342-382, the matches given (see svo_oracle.h) */
int orc_select_matches(int n, const int32_t* cell, const uint8_t* success, int n_cells, uint8_t* occupancy, int max_n_features_per_frame,
                       int* num_features_io, uint8_t* visited, int* n_trials, int* n_matches)
{
  int i = 0;
  *n_trials = 0; *n_matches = 0;
  for (int k = 0; k < n; ++k) visited[k] = 0;
  for (int k = 0; k < n; ++k) {
    ++i;                                                                  /* :358 */
    const int c = cell[k];
    if (c < 0 || c >= n_cells) continue;                                  /* (never for a visible candidate) */
    if (max_n_features_per_frame > 0 && occupancy[c]) continue;           /* :361-364 */
    ++*n_trials;                                                          /* :366 */
    visited[k] = 1;
    if (success[k]) {                                                     /* :368-369 */
      ++*n_matches;                                                       /* :371 */
      ++*num_features_io;                                                 /* :372 */
      occupancy[c] = 1;                                                   /* :373 */
      if (max_n_features_per_frame > 0 && *num_features_io >= max_n_features_per_frame) break;   /* :374-378 */
    }
  }
  return i;                                                               /* :381 */
}
