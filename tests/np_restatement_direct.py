"""Independent NumPy restatement of the reference's matcher / depth-filter path (SURVEY.md 8 rows a-10 ... a-14).

A SECOND reading of the reference, used only by tests: written from the reference's source files
(`file:line` below, paths relative to the reference tree) WITHOUT consulting oracle/*.c or csrc/*.hip, with
different machinery (whole-patch NumPy arrays instead of per-pixel loops, float32 running sums via cumsum), so that
a line misread by the author of the C oracle and of the kernels does not pass unnoticed because both sides share
it.  tests/test_np_second_opinion_cpu.py compares the C oracle with this file on thousands of units.

Restated:
  warp::getWarpMatrixAffine / getBestSearchLevel / warpAffine / warpPixelwise   src/svo_direct/src/patch_warp.cpp:20-60, 97-230
  patch_utils::createPatchFromPatchWithBorder                      src/svo_direct/include/svo/direct/patch_utils.h:18-30
  patch_score::ZMSSD<4>                                            src/svo_direct/include/svo/direct/patch_score.h:44-285
  feature_alignment::align1D / align2D                             src/svo_direct/src/feature_alignment.cpp:31-391
  Matcher::findMatchDirect / findEpipolarMatchDirect / findLocalMatch / scanEpipolarUnitPlane / UnitSphere,
  matcher_utils::depthFromTriangulation                            src/svo_direct/src/matcher.cpp:31-505
  depth_filter_utils::updateSeed / updateFilterVogiatzis / updateFilterGaussian / computeTau
                                                                   src/svo_direct/src/depth_filter.cpp:367-596
  seed::*                                                          src/svo_common/include/svo/common/seed.h:110-169
  PinholeProjection / RadialTangentialDistortion                   src/vikit/vikit_cameras/include/vikit/cameras/
                                                                   implementation/pinhole_projection.hpp:30-64,
                                                                   radial_tangential_distortion.h:34-95
  minkindr quaternion / transformation algebra                     3rd/minkindr/include/kindr/minimal/implementation/

Third-party arithmetic (Eigen 3.4, not in the reference tree) is written from the published algorithms: 2x2 / 3x3 / 4x4
inverses by cofactors (Eigen/src/LU/InverseImpl.h), fixed-size sums as Eigen's unrolled halving reduction
(Eigen/src/Core/Redux.h: a0 + (a1 + a2); (a0 + a1) + (a2 + a3)), AngleAxis::toRotationMatrix.

float32 arithmetic is done in np.float32 operation by operation (NumPy does not contract a*b+c), order-dependent
sums as np.cumsum(..., dtype=float32), which adds strictly left to right.
"""
import math

import numpy as np

f32 = np.float32

# svo::FeatureType (src/svo_common/include/svo/common/types.h:60-73)
EDGELET_SEED, CORNER_SEED, MAPPOINT_SEED = 0, 1, 2
EDGELET_SEED_CONV, CORNER_SEED_CONV, MAPPOINT_SEED_CONV = 3, 4, 5
EDGELET, CORNER, MAPPOINT, FIXED_LANDMARK, OUTLIER = 6, 7, 8, 9, 10
# Matcher::MatchResult (src/svo_direct/include/svo/direct/matcher.h:56-68)
SUCCESS, FAIL_SCORE, FAIL_TRIANGULATION, FAIL_VISIBILITY, FAIL_WARP, FAIL_ALIGNMENT = 0, 1, 2, 3, 4, 5
FAIL_RANGE, FAIL_ANGLE, FAIL_CLOSE_VIEW, FAIL_LOCK, FAIL_TOO_FAR = 6, 7, 8, 9, 10
NOT_RUN = 100   # updateSeed returned before the matcher ran (the C ABI's extra code)

K_HALF_PATCH = 4
K_PATCH = 8


def is_edgelet(t):   # types.h:116-121
    return t in (EDGELET, EDGELET_SEED, EDGELET_SEED_CONV)


def is_seed(t):      # types.h:78-81
    return t < 6


# ---------------------------------------------------------------------------------------------------------------
# minkindr / Eigen algebra (double)
# ---------------------------------------------------------------------------------------------------------------

def q_rot(q, v):
    """Eigen QuaternionBase::_transformVector (what RotationQuaternion::rotate calls, rotation-quaternion-inl.h:323-326):
    uv = 2 (q.vec x v); v + w uv + q.vec x uv."""
    qv = np.array([q[1], q[2], q[3]])
    uv = np.cross(qv, v)
    uv = uv + uv
    return v + q[0] * uv + np.cross(qv, uv)


def q_mul(a, b):
    """Eigen quaternion product, then minkindr's normalizationHelper (rotation-quaternion-inl.h:437-442, 580-589)."""
    r = np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                  a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                  a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
                  a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])
    if abs(float(r @ r) - 1.0) > 1e-4:
        r = r / math.sqrt(float(r @ r))
    return r


class Tf(object):
    """kindr::minimal::QuatTransformation: q (w, x, y, z), t."""

    def __init__(self, q, t):
        self.q, self.t = np.asarray(q, np.float64), np.asarray(t, np.float64)

    @staticmethod
    def from7(v):
        return Tf(v[:4], v[4:7])

    def __mul__(self, o):     # quat-transformation-inl.h:151-156
        return Tf(q_mul(self.q, o.q), self.t + q_rot(self.q, o.t))

    def inverse(self):        # :212-215: (q^-1, -(q^-1 * t)); Eigen's inverse() of a unit quaternion = conjugate / |q|^2
        n2 = float(self.q @ self.q)
        qi = np.array([self.q[0], -self.q[1], -self.q[2], -self.q[3]]) / n2
        return Tf(np.array([self.q[0], -self.q[1], -self.q[2], -self.q[3]]), -q_rot(qi, self.t))

    def apply(self, p):       # :158-163
        return q_rot(self.q, p) + self.t


def angle_axis_matrix(angle, axis):
    """Eigen::AngleAxis::toRotationMatrix (kindr AngleAxis::rotate = C_A_B_ * v, angle-axis-inl.h:192-195)."""
    s, c = math.sin(angle), math.cos(angle)
    sin_axis = s * axis
    cos1_axis = (1.0 - c) * axis
    R = np.zeros((3, 3))
    tmp = cos1_axis[0] * axis[1]
    R[0, 1] = tmp - sin_axis[2]; R[1, 0] = tmp + sin_axis[2]
    tmp = cos1_axis[0] * axis[2]
    R[0, 2] = tmp + sin_axis[1]; R[2, 0] = tmp - sin_axis[1]
    tmp = cos1_axis[1] * axis[2]
    R[1, 2] = tmp - sin_axis[0]; R[2, 1] = tmp + sin_axis[0]
    R[0, 0] = cos1_axis[0] * axis[0] + c
    R[1, 1] = cos1_axis[1] * axis[1] + c
    R[2, 2] = cos1_axis[2] * axis[2] + c
    return R


def normalized(v):
    n = math.sqrt(float(v @ v))
    return v / n if n > 0 else v


# ---------------------------------------------------------------------------------------------------------------
# camera (pinhole, optional radial-tangential distortion)
# ---------------------------------------------------------------------------------------------------------------

class Cam(object):
    def __init__(self, width, height, fx, fy, cx, cy, dist=None):
        self.width, self.height = int(width), int(height)
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.dist = None if dist is None else [float(d) for d in dist]

    @staticmethod
    def of(c):
        return Cam(c.width, c.height, c.fx, c.fy, c.cx, c.cy, c.dist)

    def distort(self, x, y):              # radial_tangential_distortion.h:46-56
        if self.dist is None:
            return x, y
        k1, k2, p1, p2 = self.dist
        xx, yy, xy = x * x, y * y, x * y
        xy2 = 2.0 * xy
        r2 = xx + yy
        cdist = (k1 + k2 * r2) * r2
        return (x + x * cdist + p1 * xy2 + p2 * (r2 + 2.0 * xx),
                y + y * cdist + p2 * xy2 + p1 * (r2 + 2.0 * yy))

    def distort_jacobian(self, x, y):     # :58-77
        if self.dist is None:
            return np.eye(2)
        k1, k2, p1, p2 = self.dist
        xx, yy, xy = x * x, y * y, x * y
        r2 = xx + yy
        cdist = (k1 + k2 * r2) * r2
        k2_r2_x4 = k2 * r2 * 4.0
        cdist_p1 = cdist + 1.0
        J = np.zeros((2, 2))
        J[0, 0] = cdist_p1 + k1 * 2.0 * xx + k2_r2_x4 * xx + 2.0 * p1 * y + 6.0 * p2 * x
        J[1, 1] = cdist_p1 + k1 * 2.0 * yy + k2_r2_x4 * yy + 2.0 * p2 * x + 6.0 * p1 * y
        J[1, 0] = 2.0 * k1 * xy + k2_r2_x4 * xy + 2.0 * p1 * x + 2.0 * p2 * y
        J[0, 1] = J[1, 0]
        return J

    def undistort(self, x, y):            # :79-95: five fixed-point iterations
        if self.dist is None:
            return x, y
        k1, k2, p1, p2 = self.dist
        x0, y0 = x, y
        for _ in range(5):
            xx, yy, xy = x * x, y * y, x * y
            xy2 = 2 * xy
            r2 = xx + yy
            icdist = 1.0 / (1.0 + (k1 + k2 * r2) * r2)
            dx = p1 * xy2 + p2 * (r2 + 2.0 * xx)
            dy = p2 * xy2 + p1 * (r2 + 2.0 * yy)
            x = (x0 - dx) * icdist
            y = (y0 - dy) * icdist
        return x, y

    def project3(self, p):                # pinhole_projection.hpp:44-64
        z_inv = 1.0 / p[2]
        x, y = self.distort(p[0] * z_inv, p[1] * z_inv)
        return np.array([self.fx * x + self.cx, self.fy * y + self.cy])

    def project3_jacobian(self, p):       # :55-63: diag(fx, fy) * J_dist(uv) * d(uv)/d(xyz)
        z_inv = 1.0 / p[2]
        uv = (p[0] * z_inv, p[1] * z_inv)
        duv = np.array([[z_inv, 0.0, -p[0] * z_inv * z_inv], [0.0, z_inv, -p[1] * z_inv * z_inv]])
        return np.diag([self.fx, self.fy]) @ self.distort_jacobian(uv[0], uv[1]) @ duv

    def back_project3(self, px):          # :30-42
        x = (px[0] - self.cx) * (1.0 / self.fx)
        y = (px[1] - self.cy) * (1.0 / self.fy)
        x, y = self.undistort(x, y)
        return np.array([x, y, 1.0])

    def is_keypoint_visible(self, px):    # camera_geometry_base.hpp:6-16
        return px[0] >= 0.0 and px[1] >= 0.0 and px[0] < float(self.width) and px[1] < float(self.height)

    def is_keypoint_visible_with_margin(self, pxi, margin):   # :18-29 (integer keypoint)
        return (pxi[0] >= margin and pxi[1] >= margin and pxi[0] < self.width - margin and pxi[1] < self.height - margin)

    def angle_error(self, img_err):       # pinhole_projection.hpp:73-76
        return math.atan(img_err / (2.0 * self.fx)) + math.atan(img_err / (2.0 * self.fy))


class FrameView(object):
    """What the matcher reads of a Frame: the pyramid (list of u8 arrays), the camera, T_f_w, id, seed_mu_range."""

    def __init__(self, levels, cam, T_f_w, frame_id=0, seed_mu_range=0.0):
        self.levels = [np.ascontiguousarray(l, np.uint8) for l in levels]
        self.cam, self.T_f_w, self.id, self.seed_mu_range = cam, T_f_w, frame_id, seed_mu_range


# ---------------------------------------------------------------------------------------------------------------
# a-10 warp
# ---------------------------------------------------------------------------------------------------------------

def get_warp_matrix_affine(cam_ref, cam_cur, px_ref, f_ref, depth_ref, T_cur_ref, level_ref):
    """patch_warp.cpp:20-60 (pinhole branch: the back-projected rays are scaled by the z of xyz_ref)."""
    half = 5
    xyz_ref = f_ref * depth_ref
    xyz_du = cam_ref.back_project3(px_ref + np.array([half, 0.0]) * (1 << level_ref)) * xyz_ref[2]
    xyz_dv = cam_ref.back_project3(px_ref + np.array([0.0, half]) * (1 << level_ref)) * xyz_ref[2]
    px_cur = cam_cur.project3(T_cur_ref.apply(xyz_ref))
    px_du = cam_cur.project3(T_cur_ref.apply(xyz_du))
    px_dv = cam_cur.project3(T_cur_ref.apply(xyz_dv))
    A = np.zeros((2, 2))
    A[:, 0] = (px_du - px_cur) / half
    A[:, 1] = (px_dv - px_cur) / half
    return A


def get_best_search_level(A_cur_ref, max_level):
    """patch_warp.cpp:97-110."""
    level = 0
    D = A_cur_ref[0, 0] * A_cur_ref[1, 1] - A_cur_ref[1, 0] * A_cur_ref[0, 1]
    while D > 3.0 and level < max_level:
        level += 1
        D *= 0.25
    return level


def inverse2(A):
    """Eigen's 2x2 inverse: adjugate times 1/det."""
    invdet = 1.0 / (A[0, 0] * A[1, 1] - A[1, 0] * A[0, 1])
    return np.array([[A[1, 1] * invdet, -A[0, 1] * invdet], [-A[1, 0] * invdet, A[0, 0] * invdet]])


def warp_affine(A_cur_ref, img_ref, px_ref, level_ref, search_level, halfpatch):
    """patch_warp.cpp:112-156: (2 halfpatch)^2 u8 patch (row-major) or None.  float32 sampling coordinates and
    weights, truncating float -> u8."""
    A_ref_cur = inverse2(A_cur_ref).astype(f32) * f32(1 << search_level)
    if np.isnan(A_ref_cur[0, 0]):
        return None
    px_ref_pyr = np.asarray(px_ref, np.float64).astype(f32) / f32(1 << level_ref)
    r = np.arange(-halfpatch, halfpatch, dtype=f32)
    X, Y = np.meshgrid(r, r)                       # X[y, x] = x, Y[y, x] = y
    pxx = (A_ref_cur[0, 0] * X + A_ref_cur[0, 1] * Y) + px_ref_pyr[0]
    pxy = (A_ref_cur[1, 0] * X + A_ref_cur[1, 1] * Y) + px_ref_pyr[1]
    if not (np.isfinite(pxx).all() and np.isfinite(pxy).all()):
        return None
    xi = np.floor(pxx).astype(np.int64)
    yi = np.floor(pxy).astype(np.int64)
    rows, cols = img_ref.shape
    if (xi < 0).any() or (yi < 0).any() or (xi + 1 >= cols).any() or (yi + 1 >= rows).any():
        return None
    sx = pxx - xi.astype(f32)
    sy = pxy - yi.astype(f32)
    one = f32(1.0)
    w00 = (one - sx) * (one - sy)
    w01 = (one - sx) * sy
    w10 = sx * (one - sy)
    w11 = ((one - w00) - w01) - w10
    p00 = img_ref[yi, xi].astype(f32)
    p01 = img_ref[yi + 1, xi].astype(f32)          # ptr[stride]
    p10 = img_ref[yi, xi + 1].astype(f32)          # ptr[1]
    p11 = img_ref[yi + 1, xi + 1].astype(f32)
    val = ((w00 * p00 + w01 * p01) + w10 * p10) + w11 * p11
    return val.astype(np.uint8)                    # C's float -> uint8_t: truncation (values are in [0, 255])


def warp_pixelwise(cur, ref, px_ref, landmark_pos, level_ref, level_cur, halfpatch):
    """patch_warp.cpp:158-230: (2 halfpatch)^2 u8 patch or None.  The patch's pixels (search level of the current frame)
    are back-projected to the landmark's distance from the current camera and sampled in the reference level; double
    coordinates, float32 weights.  Frame::pos() = T_world_cam().getPosition() (frame.h:261, 306)."""
    lm = np.asarray(landmark_pos, np.float64)
    T_w_ref, T_w_cur = ref.T_f_w.inverse(), cur.T_f_w.inverse()
    depth_ref = math.sqrt(float(np.sum((T_w_ref.t - lm) ** 2)))
    depth_cur = math.sqrt(float(np.sum((T_w_cur.t - lm) ** 2)))
    xyz_ref = normalized(ref.cam.back_project3(px_ref)) * depth_ref
    xyz_cur = (cur.T_f_w * T_w_ref).apply(xyz_ref)
    px_cur_search = cur.cam.project3(xyz_cur) / (1 << level_cur)
    T_ref_cur = ref.T_f_w * T_w_cur
    img_ref = ref.levels[level_ref]
    rows, cols = img_ref.shape
    out = np.zeros((2 * halfpatch, 2 * halfpatch), np.uint8)
    one = f32(1.0)
    for iy, y in enumerate(range(-halfpatch, halfpatch)):
        for ix, x in enumerate(range(-halfpatch, halfpatch)):
            ele_search = np.array([float(x), float(y)]) + px_cur_search
            e_cur = normalized(cur.cam.back_project3(ele_search * (1 << level_cur))) * depth_cur
            ele_ref = ref.cam.project3(T_ref_cur.apply(e_cur)) / (1 << level_ref)
            xi, yi = int(math.floor(ele_ref[0])), int(math.floor(ele_ref[1]))
            if xi < 0 or yi < 0 or xi + 1 >= cols or yi + 1 >= rows:
                return None
            sx, sy = f32(ele_ref[0] - xi), f32(ele_ref[1] - yi)
            w00 = (one - sx) * (one - sy)
            w01 = (one - sx) * sy
            w10 = sx * (one - sy)
            w11 = ((one - w00) - w01) - w10
            val = ((w00 * f32(img_ref[yi, xi]) + w01 * f32(img_ref[yi + 1, xi])) + w10 * f32(img_ref[yi, xi + 1])) + w11 * f32(img_ref[yi + 1, xi + 1])
            out[iy, ix] = np.uint8(int(val))          # truncation
    return out


def patch_from_patch_with_border(pwb, patch_size=K_PATCH):
    """patch_utils.h:18-30."""
    return np.ascontiguousarray(pwb[1:patch_size + 1, 1:patch_size + 1])


# ---------------------------------------------------------------------------------------------------------------
# a-11 ZMSSD
# ---------------------------------------------------------------------------------------------------------------

ZMSSD_THRESHOLD = 2000 * 64   # patch_score.h:49


class ZMSSD(object):
    def __init__(self, ref_patch):      # patch_score.h:53-89
        a = ref_patch.astype(np.int64).ravel()
        self.a, self.sumA, self.sumAA = a, int(a.sum()), int((a * a).sum())

    def score(self, img, x0, y0):       # :196-283: the 8x8 block whose top-left pixel is (x0, y0)
        b = img[y0:y0 + 8, x0:x0 + 8].astype(np.int64).ravel()
        sumB, sumBB, sumAB = int(b.sum()), int((b * b).sum()), int((b * self.a).sum())
        mean_term = (self.sumA * self.sumA - 2 * self.sumA * sumB + sumB * sumB)
        return self.sumAA - 2 * sumAB + sumBB - int(mean_term // 64)   # numerator = (sumA - sumB)^2 >= 0: C's '/' = floor


# ---------------------------------------------------------------------------------------------------------------
# a-12 align1D / align2D (float32)
# ---------------------------------------------------------------------------------------------------------------

def _seq_sum(x):
    """float32 sum strictly left to right."""
    return np.cumsum(x.astype(f32), dtype=f32)[-1]


def _inverse3_f32(m):
    """Eigen 3x3 inverse (InverseImpl.h: cofactors of column 0, det = their dot with column 0, transposed cofactors
    times 1/det), float32."""
    def cof(i, j):
        i1, i2, j1, j2 = (i + 1) % 3, (i + 2) % 3, (j + 1) % 3, (j + 2) % 3
        return f32(f32(m[i1, j1] * m[i2, j2]) - f32(m[i1, j2] * m[i2, j1]))
    c0 = [cof(0, 0), cof(1, 0), cof(2, 0)]
    prod = [f32(c0[k] * m[k, 0]) for k in range(3)]
    det = f32(prod[0] + f32(prod[1] + prod[2]))
    invdet = f32(f32(1.0) / det)
    r = np.zeros((3, 3), f32)
    for i in range(3):
        for j in range(3):
            r[i, j] = f32(cof(j, i) * invdet)
    return r


def _inverse4_f32(m):
    """Eigen 4x4 inverse, generic (non-vectorised) path: signed cofactors, divided by the expansion along column 0."""
    def det3(i1, i2, i3, j1, j2, j3):
        return f32(m[i1, j1] * f32(f32(m[i2, j2] * m[i3, j3]) - f32(m[i2, j3] * m[i3, j2])))

    def cof(i, j):
        i1, i2, i3 = (i + 1) % 4, (i + 2) % 4, (i + 3) % 4
        j1, j2, j3 = (j + 1) % 4, (j + 2) % 4, (j + 3) % 4
        return f32(f32(det3(i1, i2, i3, j1, j2, j3) + det3(i2, i3, i1, j1, j2, j3)) + det3(i3, i1, i2, j1, j2, j3))
    r = np.zeros((4, 4), f32)
    for i in range(4):
        for j in range(4):
            c = cof(i, j)
            r[j, i] = c if (i + j) % 2 == 0 else f32(-c)
    p = [f32(m[k, 0] * r[0, k]) for k in range(4)]
    s = f32(f32(p[0] + p[1]) + f32(p[2] + p[3]))
    return (r / s).astype(f32)


def _bilinear_weights(u, v, u_r, v_r):
    """feature_alignment.cpp:115-121 / 304-309: the subpixel offsets are float, the products are formed in double
    (1.0 is a double literal) and stored as float."""
    sx = f32(u - f32(u_r))
    sy = f32(v - f32(v_r))
    sxd, syd = float(sx), float(sy)
    return f32((1.0 - sxd) * (1.0 - syd)), f32(sxd * (1.0 - syd)), f32((1.0 - sxd) * syd), f32(sxd * syd)


def _interp_patch(img, u_r, v_r, w):
    """The 8x8 bilinear samples around (u_r, v_r): wTL it[0] + wTR it[1] + wBL it[step] + wBR it[step+1], float32,
    left to right."""
    x0, y0 = u_r - K_HALF_PATCH, v_r - K_HALF_PATCH
    tl = img[y0:y0 + 8, x0:x0 + 8].astype(f32)
    tr = img[y0:y0 + 8, x0 + 1:x0 + 9].astype(f32)
    bl = img[y0 + 1:y0 + 9, x0:x0 + 8].astype(f32)
    br = img[y0 + 1:y0 + 9, x0 + 1:x0 + 9].astype(f32)
    return (((w[0] * tl + w[1] * tr) + w[2] * bl) + w[3] * br).ravel()


def align1d(cur_img, direction, pwb, patch, n_iter, affine_est_offset, affine_est_gain, px):
    """feature_alignment.cpp:31-209.  px: (u, v) double in; returns (converged, px_out (2,) double, h_inv)."""
    rows, cols = cur_img.shape
    b = pwb.astype(np.int64)
    # Jacobian and Hessian (:52-86): dx, dy float differences of the bordered patch; J0 formed in double, stored float
    dx = (b[1:9, 2:10] - b[1:9, 0:8]).astype(f32).astype(np.float64).ravel()
    dy = (b[2:10, 1:9] - b[0:8, 1:9]).astype(f32).astype(np.float64).ravel()
    J0 = (0.5 * (direction[0] * dx + direction[1] * dy)).astype(f32)
    J1 = np.full(64, 1.0 if affine_est_offset else 0.0, f32)
    J2 = (-b[1:9, 1:9].astype(f32).ravel()) if affine_est_gain else np.zeros(64, f32)
    J = [J0, J1, J2]
    H = np.zeros((3, 3), f32)
    for i in range(3):
        for j in range(3):
            H[i, j] = _seq_sum(J[i] * J[j])
    if not affine_est_offset:
        H[1, 1] = f32(1.0)
    if not affine_est_gain:
        H[2, 2] = f32(1.0)
    h_inv = 1.0 / float(H[0, 0]) * K_PATCH * K_PATCH
    Hinv = _inverse3_f32(H)
    mean_diff, alpha = f32(0.0), f32(1.0)
    u, v = f32(px[0]), f32(px[1])
    min_update_squared = f32(0.03 * 0.03)
    ref = patch.astype(f32).ravel()
    converged = False
    for _ in range(n_iter):
        if not (math.isfinite(float(u)) and math.isfinite(float(v))):
            # floor() of a NaN is undefined as an int; the reference's bounds test or its isnan test ends the loop
            if math.isnan(float(u)) or math.isnan(float(v)):
                return False, np.array([float(u), float(v)]), h_inv
            break
        u_r, v_r = int(math.floor(float(u))), int(math.floor(float(v)))
        if u_r < K_HALF_PATCH or v_r < K_HALF_PATCH or u_r >= cols - K_HALF_PATCH or v_r >= rows - K_HALF_PATCH:
            break
        w = _bilinear_weights(u, v, u_r, v_r)
        cur = _interp_patch(cur_img, u_r, v_r, w)
        res = (cur - alpha * ref) + mean_diff
        Jres = np.zeros(3, f32)
        Jres[0] = -_seq_sum(res * J0)                    # Jres[0] -= res * dv, from 0
        if affine_est_offset:
            Jres[1] = -_seq_sum(res)
        if affine_est_gain:
            Jres[2] = -_seq_sum((f32(-1.0) * res) * ref)
        # Matrix3f * Vector3f: per row a0 b0 + (a1 b1 + a2 b2) (Eigen's unrolled reduction of three terms)
        upd = np.zeros(3, f32)
        for i in range(3):
            t = [f32(Hinv[i, k] * Jres[k]) for k in range(3)]
            upd[i] = f32(t[0] + f32(t[1] + t[2]))
        u = f32(float(u) + float(upd[0]) * direction[0])
        v = f32(float(v) + float(upd[0]) * direction[1])
        mean_diff = f32(mean_diff + upd[1])
        alpha = f32(alpha + upd[2])
        if f32(upd[0] * upd[0]) < min_update_squared:
            converged = True
            break
    return converged, np.array([float(u), float(v)]), h_inv


def align2d(cur_img, pwb, patch, n_iter, affine_est_offset, affine_est_gain, px):
    """feature_alignment.cpp:212-391.  Returns (converged, px_out)."""
    rows, cols = cur_img.shape
    b = pwb.astype(np.int64)
    J0 = (0.5 * (b[1:9, 2:10] - b[1:9, 0:8])).astype(f32).ravel()
    J1 = (0.5 * (b[2:10, 1:9] - b[0:8, 1:9])).astype(f32).ravel()
    J2 = np.full(64, 1.0 if affine_est_offset else 0.0, f32)
    J3 = (-1.0 * b[1:9, 1:9]).astype(f32).ravel() if affine_est_gain else np.zeros(64, f32)
    J = [J0, J1, J2, J3]
    H = np.zeros((4, 4), f32)
    for i in range(4):
        for j in range(4):
            H[i, j] = _seq_sum(J[i] * J[j])
    if not affine_est_offset:
        H[2, 2] = f32(1.0)
    if not affine_est_gain:
        H[3, 3] = f32(1.0)
    Hinv = _inverse4_f32(H)
    mean_diff, alpha = f32(0.0), f32(1.0)
    u, v = f32(px[0]), f32(px[1])
    min_update_squared = f32(0.03 * 0.03)
    ref = patch.astype(f32).ravel()
    converged = False
    for _ in range(n_iter):
        if math.isnan(float(u)) or math.isnan(float(v)):
            return False, np.array([float(u), float(v)])
        if not (math.isfinite(float(u)) and math.isfinite(float(v))):
            break
        u_r, v_r = int(math.floor(float(u))), int(math.floor(float(v)))
        if u_r < K_HALF_PATCH or v_r < K_HALF_PATCH or u_r >= cols - K_HALF_PATCH or v_r >= rows - K_HALF_PATCH:
            break
        w = _bilinear_weights(u, v, u_r, v_r)
        cur = _interp_patch(cur_img, u_r, v_r, w)
        res = (cur - alpha * ref) + mean_diff
        Jres = np.zeros(4, f32)
        Jres[0] = -_seq_sum(res * J0)
        Jres[1] = -_seq_sum(res * J1)
        if affine_est_offset:
            Jres[2] = -_seq_sum(res)
        if affine_est_gain:
            Jres[3] = -_seq_sum((f32(-1.0) * res) * ref)
        # Matrix4f * Vector4f as the reference's SSE build evaluates it: column by column, ((c0 b0 + c1 b1) + c2 b2) + c3 b3
        upd = np.zeros(4, f32)
        for i in range(4):
            acc = f32(Hinv[i, 0] * Jres[0])
            for k in range(1, 4):
                acc = f32(acc + f32(Hinv[i, k] * Jres[k]))
            upd[i] = acc
        u = f32(u + upd[0]); v = f32(v + upd[1])
        mean_diff = f32(mean_diff + upd[2]); alpha = f32(alpha + upd[3])
        if f32(f32(upd[0] * upd[0]) + f32(upd[1] * upd[1])) < min_update_squared:
            converged = True
            break
    return converged, np.array([float(u), float(v)])


# ---------------------------------------------------------------------------------------------------------------
# a-13 Matcher
# ---------------------------------------------------------------------------------------------------------------

class MatcherOptions(object):
    """Matcher::Options (src/svo_direct/include/svo/direct/matcher.h:39-54)."""

    def __init__(self, **kw):
        self.align_1d = False
        self.align_max_iter = 10
        self.max_epi_search_steps = 100
        self.subpix_refinement = True
        self.epi_search_edgelet_filtering = True
        self.scan_on_unit_sphere = True
        self.epi_search_edgelet_max_angle = 0.7
        self.affine_est_offset = True
        self.affine_est_gain = False
        self.max_patch_diff_ratio = 2.0
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)


class Matcher(object):
    def __init__(self, options=None):
        self.options = options or MatcherOptions()
        self.A_cur_ref = np.zeros((2, 2)); self.search_level = 0; self.reject = False
        self.px_cur = np.zeros(2); self.f_cur = np.zeros(3); self.h_inv = 0.0
        self.pwb = None; self.patch = None; self.epi_image = np.zeros(2); self.epi_length_pyramid = 0.0

    # -- matcher.cpp:31-141
    def find_match_direct(self, ref, cur, px, f, grad, level, ftype, ref_depth, px_cur, landmark_pos=None):
        """landmark_pos given = options_.use_affine_warp_ false (matcher.cpp:67-81): the patch by warpPixelwise, with
        the search level of the affine warp."""
        o = self.options
        pxi = (int(px[0]) // (1 << level), int(px[1]) // (1 << level))   # cast<int>() truncates; pixels are >= 0 here
        if px[0] < 0 or px[1] < 0:
            pxi = (int(int(px[0]) / (1 << level)), int(int(px[1]) / (1 << level)))   # C division truncates toward zero
        boundary = K_HALF_PATCH + 2
        if (pxi[0] < boundary or pxi[1] < boundary or pxi[0] >= int(ref.cam.width // (1 << level)) - boundary
                or pxi[1] >= int(ref.cam.height // (1 << level)) - boundary):
            return FAIL_VISIBILITY, px_cur
        T_cur_ref = cur.T_f_w * ref.T_f_w.inverse()
        self.A_cur_ref = get_warp_matrix_affine(ref.cam, cur.cam, px, f, ref_depth, T_cur_ref, level)
        self.search_level = get_best_search_level(self.A_cur_ref, len(ref.levels) - 1)
        if landmark_pos is None:
            self.pwb = warp_affine(self.A_cur_ref, ref.levels[level], px, level, self.search_level, K_HALF_PATCH + 1)
        else:
            self.pwb = warp_pixelwise(cur, ref, px, landmark_pos, level, self.search_level, K_HALF_PATCH + 1)
        if self.pwb is None:
            return FAIL_WARP, px_cur
        self.patch = patch_from_patch_with_border(self.pwb)
        px_scaled = np.asarray(px_cur, np.float64) / (1 << self.search_level)
        px_start = px_scaled.copy()
        if is_edgelet(ftype):
            dir_cur = normalized(self.A_cur_ref @ grad)
            ok, px_scaled, self.h_inv = align1d(cur.levels[self.search_level], dir_cur, self.pwb, self.patch, o.align_max_iter,
                                                o.affine_est_offset, o.affine_est_gain, px_scaled)
        else:
            ok, px_scaled = align2d(cur.levels[self.search_level], self.pwb, self.patch, o.align_max_iter,
                                    o.affine_est_offset, o.affine_est_gain, px_scaled)
        if ok:
            d = px_scaled - px_start
            if math.sqrt(float(d @ d)) > o.max_patch_diff_ratio * K_PATCH:
                return FAIL_TOO_FAR, px_cur
            px_out = px_scaled * (1 << self.search_level)
            self.px_cur = px_out
            self.f_cur = normalized(cur.cam.back_project3(px_out))
            return SUCCESS, px_out
        return FAIL_ALIGNMENT, px_cur

    # -- matcher.cpp:264-292
    def find_local_match(self, frame, direction, patch_level, px_cur):
        o = self.options
        px_scaled = px_cur / (1 << patch_level)
        if o.align_1d:
            ok, px_scaled, self.h_inv = align1d(frame.levels[patch_level], direction, self.pwb, self.patch, o.align_max_iter,
                                                o.affine_est_offset, o.affine_est_gain, px_scaled)
        else:
            ok, px_scaled = align2d(frame.levels[patch_level], self.pwb, self.patch, o.align_max_iter,
                                    o.affine_est_offset, o.affine_est_gain, px_scaled)
        if not ok:
            return FAIL_ALIGNMENT, px_cur
        return SUCCESS, px_scaled * (1 << patch_level)

    # -- matcher.cpp:294-322
    @staticmethod
    def is_patch_within_image(frame, pxi, patch_level):
        return not (pxi[0] < K_PATCH or pxi[1] < K_PATCH
                    or pxi[0] >= int(frame.cam.width // (1 << patch_level)) - K_PATCH
                    or pxi[1] >= int(frame.cam.height // (1 << patch_level)) - K_PATCH)

    @staticmethod
    def _round_px(px, patch_level):
        """Eigen::Vector2i(px/2^L + 0.5): double -> int conversion truncates toward zero."""
        return (int(px[0] / (1 << patch_level) + 0.5), int(px[1] / (1 << patch_level) + 0.5))

    # -- matcher.cpp:340-413
    def scan_unit_plane(self, frame, A, B, C, score, patch_level, zmssd_best):
        o = self.options
        n_steps = int(self.epi_length_pyramid / 0.7)
        step = (A[:2] / A[2] - B[:2] / B[2]) / n_steps
        if n_steps > o.max_epi_search_steps:
            n_steps = o.max_epi_search_steps
        uv_C = C[:2] / C[2]
        uv = uv_C.copy()
        uv_best = uv.copy()
        forward = True
        last = (0, 0)
        img = frame.levels[patch_level]
        i = 0
        while i < n_steps:
            px = frame.cam.project3(np.array([uv[0], uv[1], 1.0]))
            pxi = self._round_px(px, patch_level)
            if pxi != last:
                last = pxi
                if not self.is_patch_within_image(frame, pxi, patch_level):
                    if forward:
                        i = int(n_steps * 0.5)
                        step = -step
                        uv = uv_C.copy()
                        forward = False
                    else:
                        break
                else:
                    z = score.score(img, pxi[0] - K_HALF_PATCH, pxi[1] - K_HALF_PATCH)
                    if z < zmssd_best:
                        zmssd_best = z
                        uv_best = uv.copy()
                    if forward and i > n_steps * 0.5:
                        step = -step
                        uv = uv_C.copy()
                        forward = False
            i += 1
            uv = uv + step
        return frame.cam.project3(np.array([uv_best[0], uv_best[1], 1.0])), zmssd_best

    # -- matcher.cpp:415-488
    def scan_unit_sphere(self, frame, A, B, C, score, patch_level, zmssd_best):
        o = self.options
        n_steps = int(self.epi_length_pyramid / 0.7)
        n_steps = o.max_epi_search_steps if n_steps > o.max_epi_search_steps else n_steps
        half_steps = n_steps // 2
        f_A, f_B = normalized(A), normalized(B)
        step = math.acos(float(f_A @ f_B)) / n_steps
        axis = normalized(np.cross(f_B, f_A))
        f_C = normalized(C)
        f_best = f_C.copy()
        last = (0, 0)
        img = frame.levels[patch_level]
        i = 0
        while i < n_steps:
            angle = i * step if i < half_steps else (i - half_steps) * (-step)
            f = angle_axis_matrix(angle, axis) @ f_C
            px = frame.cam.project3(f)
            pxi = self._round_px(px, patch_level)
            if pxi != last:
                last = pxi
                if not self.is_patch_within_image(frame, pxi, patch_level):
                    if i < half_steps:
                        i = half_steps
                    else:
                        break
                else:
                    z = score.score(img, pxi[0] - K_HALF_PATCH, pxi[1] - K_HALF_PATCH)
                    if z < zmssd_best:
                        zmssd_best = z
                        f_best = f.copy()
            i += 1
        return frame.cam.project3(f_best), zmssd_best

    # -- matcher.cpp:157-241
    def find_epipolar_match_direct(self, ref, cur, T_cur_ref, px, f, grad, level, ftype, d_estimate_inv, d_min_inv, d_max_inv):
        """Returns (result, depth)."""
        o = self.options
        zmssd_best = ZMSSD_THRESHOLD
        rf = q_rot(T_cur_ref.q, f)
        A = rf + T_cur_ref.t * d_min_inv
        B = rf + T_cur_ref.t * d_max_inv
        px_A, px_B = cur.cam.project3(A), cur.cam.project3(B)
        self.epi_image = px_A - px_B
        self.A_cur_ref = get_warp_matrix_affine(ref.cam, cur.cam, px, f, 1.0 / max(0.000001, d_estimate_inv), T_cur_ref, level)
        self.reject = False
        if is_edgelet(ftype) and o.epi_search_edgelet_filtering:
            grad_cur = normalized(self.A_cur_ref @ grad)
            cosangle = abs(float(grad_cur @ normalized(self.epi_image)))
            if cosangle < o.epi_search_edgelet_max_angle:
                self.reject = True
                return FAIL_ANGLE, 0.0
        self.search_level = get_best_search_level(self.A_cur_ref, len(ref.levels) - 1)
        self.epi_length_pyramid = math.sqrt(float(self.epi_image @ self.epi_image)) / (1 << self.search_level)
        epi_dir_image = normalized(self.epi_image)
        self.pwb = warp_affine(self.A_cur_ref, ref.levels[level], px, level, self.search_level, K_HALF_PATCH + 1)
        if self.pwb is None:
            return FAIL_WARP, 0.0
        self.patch = patch_from_patch_with_border(self.pwb)
        if self.epi_length_pyramid < 2.0:
            self.px_cur = (px_A + px_B) / 2.0
            res, self.px_cur = self.find_local_match(cur, epi_dir_image, self.search_level, self.px_cur)
            if res != SUCCESS:
                return res, 0.0
            self.f_cur = normalized(cur.cam.back_project3(self.px_cur))
            return depth_from_triangulation(T_cur_ref, f, self.f_cur)
        score = ZMSSD(self.patch)
        C = rf + T_cur_ref.t * d_estimate_inv
        if o.scan_on_unit_sphere:
            self.px_cur, zmssd_best = self.scan_unit_sphere(cur, A, B, C, score, self.search_level, zmssd_best)
        else:
            self.px_cur, zmssd_best = self.scan_unit_plane(cur, A, B, C, score, self.search_level, zmssd_best)
        if zmssd_best < ZMSSD_THRESHOLD:
            if o.subpix_refinement:
                res, self.px_cur = self.find_local_match(cur, epi_dir_image, self.search_level, self.px_cur)
                if res != SUCCESS:
                    return res, 0.0
            self.f_cur = normalized(cur.cam.back_project3(self.px_cur))
            return depth_from_triangulation(T_cur_ref, f, self.f_cur)
        return FAIL_SCORE, 0.0


def depth_from_triangulation(T_search_ref, f_ref, f_cur):
    """matcher.cpp:492-505."""
    A = np.stack([q_rot(T_search_ref.q, f_ref), f_cur], axis=1)      # 3 x 2
    AtA = A.T @ A
    if AtA[0, 0] * AtA[1, 1] - AtA[1, 0] * AtA[0, 1] < 0.000001:
        return FAIL_TRIANGULATION, 0.0
    depth2 = (-inverse2(AtA)) @ A.T @ T_search_ref.t
    return SUCCESS, abs(float(depth2[0]))


# ---------------------------------------------------------------------------------------------------------------
# a-14 depth filter
# ---------------------------------------------------------------------------------------------------------------

def norm_pdf(x, mean, sigma):          # vikit/math_utils.h:186-194
    e = x - mean
    e *= -e
    e /= 2 * sigma * sigma
    r = math.exp(e)
    r /= sigma * math.sqrt(2 * math.pi)
    return r


def update_filter_vogiatzis(z, tau2, mu_range, st):
    """depth_filter.cpp:501-552; st = [mu, sigma2, a, b], updated in place; returns bool."""
    mu, sigma2, a, b = st
    s = sigma2 + tau2
    norm_scale = math.sqrt(s) if s >= 0 else float("nan")
    if math.isnan(norm_scale):
        return False
    oldsigma2 = sigma2
    s2 = 1.0 / (1.0 / sigma2 + 1.0 / tau2)
    m = s2 * (mu / sigma2 + z / tau2)
    uniform_x = 1.0 / mu_range
    C1 = a / (a + b) * norm_pdf(z, mu, norm_scale)
    C2 = b / (a + b) * uniform_x
    nc = C1 + C2
    C1 /= nc
    C2 /= nc
    f = C1 * (a + 1.0) / (a + b + 1.0) + C2 * a / (a + b + 1.0)
    e = (C1 * (a + 1.0) * (a + 2.0) / ((a + b + 1.0) * (a + b + 2.0))
         + C2 * a * (a + 1.0) / ((a + b + 1.0) * (a + b + 2.0)))
    mu_new = C1 * m + C2 * mu
    sigma2 = C1 * (s2 + m * m) + C2 * (sigma2 + mu * mu) - mu_new * mu_new
    mu = mu_new
    a = (e - f) / (f - e / f)
    b = a * (1.0 - f) / f
    if sigma2 < 0.0:
        sigma2 = oldsigma2
    ok = True
    if mu < 0.0:
        mu = 1.0
        ok = False
    st[:] = [mu, sigma2, a, b]
    return ok


def update_filter_gaussian(z, tau2, st):
    """depth_filter.cpp:554-578."""
    mu, sigma2 = st[0], st[1]
    s = sigma2 + tau2
    if s < 0 or math.isnan(s):
        return False
    st[0] = (sigma2 * z + tau2 * mu) / s
    st[1] = sigma2 * tau2 / s
    return True


def compute_tau(T_ref_cur, f, z, px_error_angle):
    """depth_filter.cpp:580-596."""
    t = T_ref_cur.t
    a = f * z - t
    t_norm = math.sqrt(float(t @ t))
    a_norm = math.sqrt(float(a @ a))
    alpha = math.acos(float(f @ t) / t_norm)
    beta = math.acos(float(a @ (-t)) / (t_norm * a_norm))
    beta_plus = beta + px_error_angle
    gamma_plus = math.pi - alpha - beta_plus
    z_plus = t_norm * math.sin(beta_plus) / math.sin(gamma_plus)
    return z_plus - z


def update_seed(cur, ref, px, f, grad, level, ftype, st, matcher, sigma2_convergence_threshold, px_error_angle,
                check_visibility=True, check_convergence=False, use_vogiatzis_update=True):
    """depth_filter_utils::updateSeed (depth_filter.cpp:367-499).  st: 4 doubles, updated in place.
    Returns (success, new_type, match_result)."""
    if cur.id == ref.id:
        return False, ftype, NOT_RUN
    if ftype == OUTLIER:
        return False, ftype, NOT_RUN
    if ftype in (CORNER_SEED_CONV, EDGELET_SEED_CONV, MAPPOINT_SEED_CONV) and check_convergence:
        return False, ftype, NOT_RUN
    T_cur_ref = cur.T_f_w * ref.T_f_w.inverse()
    if check_visibility:
        xyz_f = T_cur_ref.apply((1.0 / st[0]) * f)                   # seed::getDepth = 1 / mu (seed.h:110-113)
        pxp = cur.cam.project3(xyz_f)
        if not (pxp[0] == pxp[0] and pxp[1] == pxp[1]) or not cur.cam.is_keypoint_visible(pxp):
            return False, ftype, NOT_RUN
        pxi = (int(pxp[0]), int(pxp[1]))
        if not cur.cam.is_keypoint_visible_with_margin(pxi, 9):
            return False, ftype, NOT_RUN
    matcher.options.align_1d = ftype in (EDGELET_SEED, EDGELET_SEED_CONV)
    # seed.h:115-128: inverse depth, mu + sigma, max(mu - sigma, 1e-8)
    sig = math.sqrt(st[1]) if st[1] >= 0 else float("nan")
    res, depth = matcher.find_epipolar_match_direct(ref, cur, T_cur_ref, px, f, grad, level, ftype, st[0], st[0] + sig,
                                                    max(st[0] - sig, 0.00000001))
    if res != SUCCESS:
        if not matcher.reject:
            st[3] += 1                                                # seed::increaseOutlierProbability
        return False, ftype, res
    depth_sigma = compute_tau(T_cur_ref.inverse(), f, depth, px_error_angle)
    z = 1.0 / depth                                                   # seed::getMeanFromDepth
    sg = 0.5 * (1.0 / max(0.000000000001, depth - depth_sigma) - 1.0 / (depth + depth_sigma))   # getSigma2FromDepthSigma
    tau2 = sg * sg
    ok = update_filter_vogiatzis(z, tau2, ref.seed_mu_range, st) if use_vogiatzis_update else update_filter_gaussian(z, tau2, st)
    if not ok:
        return False, OUTLIER, res
    thresh = ref.seed_mu_range / sigma2_convergence_threshold          # seed::isConverged (seed.h:143-151)
    new_type = ftype
    if st[1] < thresh * thresh:
        new_type = {CORNER_SEED: CORNER_SEED_CONV, EDGELET_SEED: EDGELET_SEED_CONV, MAPPOINT_SEED: MAPPOINT_SEED_CONV}.get(ftype, ftype)
    return True, new_type, res


def update_seeds(cur, refs, ref_idx, px, f, grad, level, ftype, state, mopt, seed_thresh=200.0, mappoint_thresh=500.0,
                 px_error_angle=None, check_visibility=True, check_convergence=False, use_vogiatzis_update=True):
    """The synchronous branch of DepthFilter::updateSeeds (depth_filter.cpp:200-233) over a flat feature list:
    arrays as in the C ABI (px 2n, f 3n, grad 2n, state 4n).  Returns dict(state, type, success, match_result, px_cur,
    search_level)."""
    n = len(level)
    st = np.asarray(state, np.float64).reshape(n, 4).copy()
    types = np.asarray(ftype, np.uint8).copy()
    success = np.zeros(n, np.uint8)
    mres = np.full(n, NOT_RUN, np.int32)
    pxc = np.zeros((n, 2)); slv = np.zeros(n, np.int32)
    m = Matcher(mopt)
    if px_error_angle is None:
        px_error_angle = cur.cam.angle_error(1.0)
    for i in range(n):
        t = int(types[i])
        if not is_seed(t):
            continue
        thr = mappoint_thresh if t in (MAPPOINT_SEED, MAPPOINT_SEED_CONV) else seed_thresh
        s = [float(x) for x in st[i]]
        ok, nt, r = update_seed(cur, refs[ref_idx[i]], np.asarray(px[2 * i:2 * i + 2], np.float64), np.asarray(f[3 * i:3 * i + 3], np.float64),
                                np.asarray(grad[2 * i:2 * i + 2], np.float64), int(level[i]), t, s, m, thr, px_error_angle,
                                check_visibility, check_convergence, use_vogiatzis_update)
        st[i] = s; types[i] = nt; success[i] = 1 if ok else 0; mres[i] = r
        if r != NOT_RUN:
            pxc[i] = m.px_cur; slv[i] = m.search_level
    return dict(state=st.ravel(), type=types, success=success, match_result=mres, px_cur=pxc.ravel(), search_level=slv)


def match_direct_batch(cur, refs, ref_idx, px, f, grad, level, ftype, depth, px_cur, mopt, landmark_xyz=None):
    """n x Matcher::findMatchDirect (landmark_xyz, n x 3: with the pixelwise warp)."""
    n = len(level)
    out = dict(result=np.zeros(n, np.int32), px_cur=np.asarray(px_cur, np.float64).copy(), search_level=np.zeros(n, np.int32),
               f_cur=np.zeros(3 * n), h_inv=np.zeros(n), A=np.zeros(4 * n))
    m = Matcher(mopt)
    for i in range(n):
        r, pc = m.find_match_direct(refs[ref_idx[i]], cur, np.asarray(px[2 * i:2 * i + 2], np.float64), np.asarray(f[3 * i:3 * i + 3], np.float64),
                                    np.asarray(grad[2 * i:2 * i + 2], np.float64), int(level[i]), int(ftype[i]), float(depth[i]),
                                    out["px_cur"][2 * i:2 * i + 2].copy(),
                                    None if landmark_xyz is None else np.asarray(landmark_xyz, np.float64).reshape(-1, 3)[i])
        out["result"][i] = r
        out["px_cur"][2 * i:2 * i + 2] = pc
        if r not in (FAIL_VISIBILITY,):
            out["search_level"][i] = m.search_level
            out["A"][4 * i:4 * i + 4] = m.A_cur_ref.T.ravel()      # col-major
        if r == SUCCESS:
            out["f_cur"][3 * i:3 * i + 3] = m.f_cur
            if is_edgelet(int(ftype[i])):
                out["h_inv"][i] = m.h_inv
    return out


# ---------------------------------------------------------------------------------------------------------------
# StereoTriangulation::compute, the loop over the new features (src/svo/src/stereo_triangulation.cpp:88-137)
# ---------------------------------------------------------------------------------------------------------------
def stereo_triangulate(frame0, frame1, T_f1f0, px, f, grad, level, ftype, indices, n_desired, mean_depth_inv,
                       min_depth_inv, max_depth_inv):
    """frame0 / frame1: FrameView of the left / right image; the new features of frame0 (arrays per feature) are visited
    in the order `indices` (the reference shuffles corners and the rest separately with std::random_shuffle: the order
    is an input here).  ONE Matcher for the whole loop, max_epi_search_steps 500, subpix_refinement on, align_1d set per
    feature to isEdgelet(type) (:92-98).  A success makes a landmark at f * depth (frame0's camera frame) and a feature
    in frame1: the matched pixel, its bearing vector, the reference feature's level / type, the gradient A_cur_ref * grad
    normalised (:106-124).  The loop ends when n_desired features have succeeded (:131-132).
    Returns (matches, result per visited index, n_failed)."""
    matcher = Matcher(MatcherOptions(max_epi_search_steps=500, subpix_refinement=True))
    matches, results = [], []
    n_succeeded = n_failed = 0
    for i_ref in indices:
        i_ref = int(i_ref)
        matcher.options.align_1d = is_edgelet(int(ftype[i_ref]))
        res, depth = matcher.find_epipolar_match_direct(frame0, frame1, T_f1f0, px[i_ref], f[i_ref], grad[i_ref], int(level[i_ref]),
                                                        int(ftype[i_ref]), mean_depth_inv, min_depth_inv, max_depth_inv)
        results.append(res)
        if res == SUCCESS:
            g = matcher.A_cur_ref @ grad[i_ref]
            matches.append(dict(i_ref=i_ref, xyz_cam0=f[i_ref] * depth, px=np.array(matcher.px_cur, np.float64),
                                f=np.array(matcher.f_cur, np.float64), grad=normalized(g), depth=depth))
            n_succeeded += 1
        else:
            n_failed += 1
        if n_succeeded >= n_desired:
            break
    return matches, np.array(results, np.int32), n_failed


# ---------------------------------------------------------------------------------------------------------------
# reprojector_utils::getCandidate / projectPointAndCheckVisibility (src/svo/src/reprojector.cpp:489-543) with
# Frame::isVisible (src/svo_common/src/frame.cpp:229-260)
# ---------------------------------------------------------------------------------------------------------------
K_REPROJ_PATCH = 8     # kPatchSize of projectPointAndCheckVisibility (reprojector.cpp:538)


def frame_is_visible(cam, T_f_w, xyz_w):
    """frame.cpp:229-260 for a pinhole-type camera: the point must lie inside the cone through the image's top-left
    corner (cos of the angle to the optical axis against that of backProject3((0, 0))), then project inside the image.
    Returns (visible, px)."""
    xyz_f = T_f_w.apply(xyz_w)
    f_top_left = normalized(cam.back_project3(np.zeros(2)))
    min_cos = f_top_left[2]
    cur_cos = normalized(xyz_f)[2]
    if cur_cos < min_cos:
        return False, np.zeros(2)
    px = cam.project3(xyz_f)
    return bool(cam.is_keypoint_visible(px)), px


def project_point_and_check_visibility(cam, T_f_w, xyz_w):
    """reprojector.cpp:525-543: visible in the frame, and the truncated pixel at least kPatchSize from every border."""
    ok, px = frame_is_visible(cam, T_f_w, xyz_w)
    if not ok:
        return False, px
    pxi = (int(px[0]), int(px[1]))                      # px->cast<int>()
    if not cam.is_keypoint_visible_with_margin(pxi, K_REPROJ_PATCH):
        return False, px
    return True, px


def get_candidate(cam_cur, T_f_w_cur, T_f_w_ref, landmark_pos, f_ref, inv_depth):
    """reprojector.cpp:489-523: the landmark's position if the feature has one (landmark_pos not None), else the seed's
    position T_world_cam * (f * depth) with depth = 1 / mu (seed.h:110-113, the inverse-depth parametrisation the
    reference is built with); then projectPointAndCheckVisibility.  Returns (is a candidate, pixel in the current frame)."""
    if landmark_pos is not None:
        xyz_world = np.asarray(landmark_pos, np.float64)
    else:
        xyz_world = T_f_w_ref.inverse().apply(np.asarray(f_ref, np.float64) * (1.0 / inv_depth))
    return project_point_and_check_visibility(cam_cur, T_f_w_cur, xyz_world)
