"""Independent NumPy restatement ("second opinion") of the sparse-image-alignment
maths, written from the reference's formulas with different machinery than the
C oracle (rotation matrices instead of quaternion cross products, whole-array
operations instead of per-feature loops).  Used only to cross-check the oracle;
agreement is expected to ~1e-10 relative, not bit for bit.

Follows: src/svo_img_align/src/sparse_img_align.cpp:209-541,
src/svo_common/include/svo/common/frame.h:342-357,
src/vikit/vikit_cameras/.../pinhole_projection.hpp:44-54,
radial_tangential_distortion.h:46-56.
"""
import numpy as np


def quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def se3_apply(q, t, P):
    return quat_to_R(q) @ P + np.asarray(t)[:, None]


def se3_compose(qa, ta, qb, tb):
    """returns (R, t) of A*B"""
    Ra, Rb = quat_to_R(qa), quat_to_R(qb)
    return Ra @ Rb, Ra @ np.asarray(tb) + np.asarray(ta)


def project(cam, P):
    x = P[0] / P[2]
    y = P[1] / P[2]
    if cam.dist is not None:
        k1, k2, p1, p2 = cam.dist
        r2 = x * x + y * y
        cd = (k1 + k2 * r2) * r2
        x, y = (x + x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x),
                y + y * cd + 2 * p2 * x * y + p1 * (r2 + 2 * y * y))
    return cam.fx * x + cam.cx, cam.fy * y + cam.cy


def bilinear(img, u, v):
    """img HxW uint8; u, v float arrays (same shape); floor-based bilinear"""
    ui = np.floor(u).astype(np.int64)
    vi = np.floor(v).astype(np.int64)
    su, sv = u - ui, v - vi
    I = img.astype(np.float64)
    return ((1 - su) * (1 - sv) * I[vi, ui] + su * (1 - sv) * I[vi, ui + 1]
            + (1 - su) * sv * I[vi + 1, ui] + su * sv * I[vi + 1, ui + 1])


def select_features(px, flags, ref_top_level, max_level, P):
    """extractFeaturesSubset"""
    scale = 1.0 / (1 << max_level)
    wb = P + 2
    c = (wb - 1) / 2.0
    u = np.floor(px[0] * scale - c).astype(np.int64)
    v = np.floor(px[1] * scale - c).astype(np.int64)
    h, w = ref_top_level.shape
    ok = (flags != 0) & ~((u < 0) | (v < 0) | (u + wb >= w - 2) | (v + wb >= h - 2))
    return np.nonzero(ok)[0]


def evaluate(scene, ref_levels, cur_levels, level, P, T_icur_iref_q, T_icur_iref_t, max_level=4,
             alpha=0.0, beta=0.0, est_alpha=False, est_beta=False, robust=False, weight_scale=10.0):
    """H (8x8), g (8), chi2, n_meas, visibility for one camera."""
    cam = scene.cam
    px = scene.px.reshape(-1, 2).T
    f = scene.f.reshape(-1, 3).T
    pw = scene.pos_world.reshape(-1, 3).T
    idx = select_features(px, scene.flags, ref_levels[max_level], max_level, P)
    px, f, pw = px[:, idx], f[:, idx], pw[:, idx]
    n = idx.size
    depth = np.linalg.norm(pw - scene.ref_pos[:, None], axis=0)
    xyz_ref = f * depth
    # projection Jacobian wrt the IMU pose (frame.h:342-357), times focal length
    R_ci = quat_to_R(scene.T_cam_imu.q)
    p_imu = se3_apply(scene.T_imu_cam.q, scene.T_imu_cam.t, xyz_ref)
    p_cam = R_ci @ p_imu + scene.T_cam_imu.t[:, None]
    Jfull = np.zeros((n, 2, 6))
    for i in range(n):
        x, y, z = p_cam[:, i]
        Jp = np.array([[1, 0, -x / z], [0, 1, -y / z]])
        px_, py_, pz_ = p_imu[:, i]
        skew = np.array([[0, -pz_, py_], [pz_, 0, -px_], [-py_, px_, 0]])
        G = np.hstack([np.eye(3), -skew])
        Jfull[i] = (-1.0 / z) * Jp @ R_ci @ G * abs(cam.fx)
    scale = 1.0 / (1 << level)
    ref_img, cur_img = ref_levels[level], cur_levels[level]
    # reference patch with border, gradients
    c_wb = (P + 2 - 1) / 2.0
    u0 = px[0] * scale - c_wb
    v0 = px[1] * scale - c_wb
    gx, gy = np.meshgrid(np.arange(P + 2), np.arange(P + 2))
    U = u0[:, None, None] + gx[None]
    V = v0[:, None, None] + gy[None]
    patch = bilinear(ref_img, U, V)  # n x (P+2) x (P+2)
    ref_val = patch[:, 1:-1, 1:-1]
    dx = 0.5 * (patch[:, 1:-1, 2:] - patch[:, 1:-1, :-2])
    dy = 0.5 * (patch[:, 2:, 1:-1] - patch[:, :-2, 1:-1])
    J = np.zeros((n, P, P, 8))
    J[..., :6] = (dx[..., None] * Jfull[:, None, None, 0, :] + dy[..., None] * Jfull[:, None, None, 1, :]) * scale
    if est_alpha:
        J[..., 6] = -ref_val
    if est_beta:
        J[..., 7] = -1.0
    # residuals
    Rcr, tcr = se3_compose(scene.T_cam_imu.q, scene.T_cam_imu.t, T_icur_iref_q, T_icur_iref_t)
    Rt, tt = quat_to_R(scene.T_imu_cam.q), scene.T_imu_cam.t
    Rcr, tcr = Rcr @ Rt, Rcr @ tt + tcr
    xyz_cur = Rcr @ xyz_ref + tcr[:, None]
    uc, vc = project(cam, xyz_cur)
    c = (P - 1) / 2.0
    utl = uc * scale - c
    vtl = vc * scale - c
    h, w = cur_img.shape
    vis = ~(xyz_cur[2] < 0) & ~((utl < 0) | (vtl < 0) | (utl + P + 2.0 >= w) | (vtl + P + 2.0 >= h))
    gx, gy = np.meshgrid(np.arange(P), np.arange(P))
    utl_s = np.where(vis, utl, 0.0)
    vtl_s = np.where(vis, vtl, 0.0)
    Icur = bilinear(cur_img, utl_s[:, None, None] + gx[None], vtl_s[:, None, None] + gy[None])
    a32, b32 = np.float32(alpha), np.float32(beta)
    res = Icur * (1.0 + float(a32)) + float(b32) - ref_val
    wgt = np.ones_like(res)
    if robust:
        e = (res / np.float32(weight_scale)).astype(np.float32)
        b2 = np.float32(4.6851) * np.float32(4.6851)
        x2 = e * e
        t = np.float32(1.0) - x2 / b2
        wgt = np.where(x2 <= b2, t * t, np.float32(0)).astype(np.float64)
    m = vis[:, None, None] * np.ones_like(res, dtype=bool)
    Jm = J[m]
    rm = res[m]
    wm = wgt[m]
    H = (Jm * wm[:, None]).T @ Jm
    g = -(Jm * (rm * wm)[:, None]).sum(0)
    chi2 = float((rm * rm * wm).sum() / max(1, rm.size))
    return H, g, chi2, int(rm.size), vis.astype(np.uint8)


def align_pyr_2d(pyr_ref, pyr_cur, max_level, min_level, patch_sizes, n_iter, min_update_squared, px_ref_level_0,
                 px_cur_level_0):
    """feature_alignment::alignPyr2D (src/svo_direct/src/feature_alignment.cpp:761-973, scalar branch), written
    independently of oracle/svo_oracle_klt.c with numpy float32 scalars and whole-patch integer arithmetic.
    Every sum in it is a sum of integers below 2^24, so the order of the additions does not matter.
    Returns (converged, (x, y))."""
    f32 = np.float32
    px_cur = [float(px_cur_level_0[0]), float(px_cur_level_0[1])]
    converged = False
    for level in range(max_level, min_level - 1, -1):
        P = int(patch_sizes[level])
        half = P // 2
        scale = 1 << level
        ref, cur = pyr_ref[level].astype(np.int64), pyr_cur[level].astype(np.int64)
        height, width = ref.shape
        px_ref_flt = [f32(f32(px_ref_level_0[k]) / f32(scale)) - f32(half) for k in range(2)]
        px_ref = [int(px_ref_flt[k]) for k in range(2)]                       # cast<int>: truncation
        off = [f32(px_ref_flt[k] - f32(px_ref[k])) for k in range(2)]
        if px_ref[0] < 1 or px_ref[1] < 1 or px_ref[0] >= width - P - 1 or px_ref[1] >= height - P - 1:
            continue
        x0, y0 = px_ref
        tmpl = ref[y0:y0 + P, x0:x0 + P]
        dx = ref[y0:y0 + P, x0 + 1:x0 + P + 1] - ref[y0:y0 + P, x0 - 1:x0 + P - 1]
        dy = ref[y0 + 1:y0 + P + 1, x0:x0 + P] - ref[y0 - 1:y0 + P - 1, x0:x0 + P]
        H00, H01, H11 = f32((dx * dx).sum()), f32((dx * dy).sum()), f32((dy * dy).sum())
        assert (dx * dx).sum() < 2 ** 24 and (dy * dy).sum() < 2 ** 24
        det = f32(f32(H00 * H11) - f32(H01 * H01))
        with np.errstate(divide="ignore", invalid="ignore"):
            invdet = f32(f32(1.0) / det)
            Hi00, Hi01, Hi10, Hi11 = f32(H11 * invdet), f32(f32(-H01) * invdet), f32(f32(-H01) * invdet), f32(H00 * invdet)
        u = f32(px_cur[0] / scale - half - float(off[0]))
        v = f32(px_cur[1] / scale - half - float(off[1]))
        go_to_next_level = False
        converged = False
        for _ in range(n_iter):
            if np.isnan(u) or np.isnan(v):
                return False, tuple(px_cur)
            go_to_next_level = False
            u_r, v_r = int(np.floor(u)), int(np.floor(v))
            if u_r < 0 or v_r < 0 or u_r >= width - P or v_r >= height - P:
                go_to_next_level = True
                break
            sx, sy = f32(u - f32(u_r)), f32(v - f32(v_r))
            one = f32(1.0)
            wTL = int(f32(f32(f32(one - sx) * f32(one - sy)) * f32(128)))
            wTR = int(f32(f32(sx * f32(one - sy)) * f32(128)))
            wBL = int(f32(f32(f32(one - sx) * sy) * f32(128)))
            wBR = 128 - wTL - wTR - wBL
            a = cur[v_r:v_r + P, u_r:u_r + P]
            b = cur[v_r:v_r + P, u_r + 1:u_r + P + 1]
            c = cur[v_r + 1:v_r + P + 1, u_r:u_r + P]
            d = cur[v_r + 1:v_r + P + 1, u_r + 1:u_r + P + 1]
            interp = (wTL * a + wTR * b + wBL * c + wBR * d + 64) >> 7
            res = interp - tmpl
            Jres0, Jres1 = f32(-(res * dx).sum()), f32(-(res * dy).sum())
            assert abs((res * dx).sum()) < 2 ** 24 and abs((res * dy).sum()) < 2 ** 24
            up0 = f32(f32(f32(Hi00 * Jres0) + f32(Hi01 * Jres1)) * f32(2.0))
            up1 = f32(f32(f32(Hi10 * Jres0) + f32(Hi11 * Jres1)) * f32(2.0))
            u, v = f32(u + up0), f32(v + up1)
            if f32(f32(up0 * up0) + f32(up1 * up1)) < f32(min_update_squared):
                converged = True
                break
        px_cur = [float(f32(f32(f32(u + f32(half)) + off[0]) * f32(scale))), float(f32(f32(f32(v + f32(half)) + off[1]) * f32(scale)))]
        if not converged and not go_to_next_level:
            return False, tuple(px_cur)
    return converged, tuple(px_cur)


def half_sample(img):
    """vk::halfSample (src/vikit/vikit_common/src/vision.cpp:70-112) on an x86 build: the SSE2 routine (:19-44: _mm_avg_epu8
    of the two rows, then _mm_avg_epu16 of the even and odd columns -- two roundings, each half up) when the width is
    a multiple of 16 (aligned, continuous rows), else the scalar sum of four divided by 4 (truncating).  Whole-array
    NumPy; a second reading beside oracle/svo_oracle.c."""
    h, w = img.shape
    oh, ow = h // 2, w // 2
    a = img.astype(np.uint16)
    if w % 16 == 0:
        v = (a[0:2 * oh:2] + a[1:2 * oh:2] + 1) >> 1
        return ((v[:, 0::2] + v[:, 1::2] + 1) >> 1).astype(np.uint8)
    top, bot = a[0:2 * oh:2], a[1:2 * oh:2]
    return ((top[:, 0:2 * ow:2] + top[:, 1:2 * ow:2] + bot[:, 0:2 * ow:2] + bot[:, 1:2 * ow:2]) // 4).astype(np.uint8)


def create_img_pyramid(img, n_levels):
    """frame_utils::createImgPyramid (src/svo_common/src/frame.cpp:372-386)."""
    pyr = [np.ascontiguousarray(img, np.uint8)]
    for _ in range(1, n_levels):
        pyr.append(half_sample(pyr[-1]))
    return pyr
