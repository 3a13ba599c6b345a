"""Independent NumPy restatement of PoseOptimizer::run, the second reading beside the C oracle
(see tests/np_restatement_direct.py for the rules: written from the reference files, not from oracle/*.c).

  PoseOptimizer::run / evaluateErrorImpl / removeOutliers / update / applyPrior / setRotationPrior
                                                        src/svo/src/pose_optimizer.cpp:30-336
  pose_optimizer_utils::calculate{Feature,Edgelet}Residual{UnitPlane,ImagePlane,BearingVectorDiff}
                                                        src/svo/src/pose_optimizer.cpp:338-627
  Frame::jacobian_xyz2uv_imu / _xyz2img_imu / _xyz2f_imu src/svo_common/include/svo/common/frame.h:342-397
  MiniLeastSquaresSolver::optimizeGaussNewton            src/vikit/vikit_solver/include/vikit/solver/implementation/
                                                        mini_least_squares_solver.hpp:42-107
  MADScaleEstimator / TukeyWeightFunction                src/vikit/vikit_solver/src/robust_cost.cpp:19-60
  minkindr exp / log                                     3rd/minkindr/.../rotation-quaternion-inl.h:476-540

The linear solve is numpy's (LU), not Eigen's LDLT: the same real-number solution, rounding differs at 1e-16 * cond.
"""
import math

import numpy as np

from np_restatement_direct import Tf, q_rot, q_mul, is_edgelet, EDGELET, f32

UNIT_PLANE, BEARING_DIFF, IMAGE_PLANE = 0, 1, 2   # the C ABI's numbering of PoseOptimizer::ErrorType


def tukey_weight(e):
    """robust_cost.cpp:44-60, float arithmetic, b = 4.6851."""
    e = f32(e)
    b2 = f32(f32(4.6851) * f32(4.6851))
    x2 = f32(e * e)
    if x2 <= b2:
        t = f32(f32(1.0) - f32(x2 / b2))
        return float(f32(t * t))
    return 0.0


def quat_exp(w):
    theta = math.sqrt(float(w @ w))
    if theta < 1.220703125e-4:          # 4th root of the double epsilon
        na = 0.5 + (theta * theta) * (1.0 / 48.0)
    else:
        na = math.sin(theta * 0.5) / theta
    return np.array([math.cos(theta * 0.5), w[0] * na, w[1] * na, w[2] * na])


def quat_log(q):
    a = q[1:]
    na = math.sqrt(float(a @ a))
    eta = q[0]
    if abs(eta) < na:
        scale = math.acos(eta) / na if eta >= 0 else -math.acos(-eta) / na
    else:
        u = 1.0 + na * na / 6.0 if abs(na) < 1.220703125e-4 else math.asin(na) / na    # detail::arcSinXOverX
        scale = u if eta > 0 else -u
    return a * (2.0 * scale)


def tf_exp(v):      # minkindr Transformation::exp: t = v[0:3], q = Exp(v[3:6]) (decoupled)
    return Tf(quat_exp(np.asarray(v[3:6])), np.asarray(v[0:3]))


def tf_log(T):
    return np.concatenate([T.t, quat_log(T.q)])


def skew(p):
    return np.array([[0.0, -p[2], p[1]], [p[2], 0.0, -p[0]], [-p[1], p[0], 0.0]])


def rot_matrix(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _G(p_in_imu):
    return np.hstack([np.eye(3), -skew(p_in_imu)])


def jacobian_xyz2uv_imu(T_cam_imu, p_in_imu):          # frame.h:342-357
    pc = T_cam_imu.apply(p_in_imu)
    Jp = np.array([[1.0, 0.0, -pc[0] / pc[2]], [0.0, 1.0, -pc[1] / pc[2]]])
    return -1.0 / pc[2] * Jp @ rot_matrix(T_cam_imu.q) @ _G(p_in_imu)


def jacobian_xyz2img_imu(T_cam_imu, p_in_imu, J_cam):   # :359-371
    return J_cam @ rot_matrix(T_cam_imu.q) @ _G(p_in_imu)


def jacobian_xyz2f_imu(T_cam_imu, p_in_imu):            # :373-397
    pc = T_cam_imu.apply(p_in_imu)
    x2, y2, z2 = pc[0] * pc[0], pc[1] * pc[1], pc[2] * pc[2]
    xy, yz, zx = pc[0] * pc[1], pc[1] * pc[2], pc[2] * pc[0]
    Jn = np.array([[y2 + z2, -xy, -zx], [-xy, x2 + z2, -yz], [-zx, -yz, x2 + y2]]) * (1 / (x2 + y2 + z2) ** 1.5)
    return Jn @ rot_matrix(T_cam_imu.q) @ _G(p_in_imu)


def residual(err_type, cam, T_imu_world, T_cam_imu, px, f, grad, xyz_world, edgelet, sigma, want_jac):
    """One of the six residual functions (pose_optimizer.cpp:338-627).  Returns (unwhitened_error, chi2, H 6x6, g 6);
    sigma == 0 (removeOutliers' calls) gives R = inf like the reference: only the unwhitened error is meaningful then."""
    p_imu = T_imu_world.apply(xyz_world)
    p_cam = T_cam_imu.apply(p_imu)
    R = 1.0 / sigma if sigma != 0.0 else float("inf")
    H = g = None
    if err_type == UNIT_PLANE:
        d = f[:2] / f[2] - p_cam[:2] / p_cam[2]
        if not edgelet:
            e = d.copy()
            ue = math.sqrt(float(e @ e))
            e = e * R
            w = tukey_weight(math.sqrt(float(e @ e))) if sigma != 0.0 else 0.0
            chi2 = 0.5 * float(e @ e) * w
            if want_jac:
                J = jacobian_xyz2uv_imu(T_cam_imu, p_imu) * R
                H, g = J.T @ J * w, -(J.T @ e) * w
        else:
            e = float(grad @ d)
            ue = abs(e)
            e *= R
            w = tukey_weight(e) if sigma != 0.0 else 0.0
            chi2 = 0.5 * e * e * w
            if want_jac:
                J = (grad @ jacobian_xyz2uv_imu(T_cam_imu, p_imu)) * R
                H, g = np.outer(J, J) * w, -(J * e) * w
        return ue, chi2, H, g
    if err_type == IMAGE_PLANE:
        px_est = cam.project3(p_cam)
        J_cam = cam.project3_jacobian(p_cam)
        d = px - px_est
        if not edgelet:
            ue = math.sqrt(float(d @ d))
            e = d * R
            w = tukey_weight(math.sqrt(float(e @ e))) if sigma != 0.0 else 0.0
            chi2 = 0.5 * float(e @ e) * w
            if want_jac:
                J = (-1.0) * jacobian_xyz2img_imu(T_cam_imu, p_imu, J_cam) * R
                H, g = J.T @ J * w, -(J.T @ e) * w
        else:
            e = float(grad @ d)
            ue = abs(e)
            e *= R
            w = tukey_weight(e) if sigma != 0.0 else 0.0
            chi2 = 0.5 * e * e * w
            if want_jac:
                J = (grad @ ((-1.0) * jacobian_xyz2img_imu(T_cam_imu, p_imu, J_cam))) * R
                H, g = np.outer(J, J) * w, -(J * e) * w
        return ue, chi2, H, g
    # bearing-vector difference
    f_est = p_cam / math.sqrt(float(p_cam @ p_cam))
    if not edgelet:
        e = f - f_est
        ue = math.sqrt(float(e @ e))
        e = e * R
        w = tukey_weight(math.sqrt(float(e @ e))) if sigma != 0.0 else 0.0
        chi2 = 0.5 * float(e @ e) * w
        if want_jac:
            J = (-1.0) * jacobian_xyz2f_imu(T_cam_imu, p_imu) * R
            H, g = J.T @ J * w, -(J.T @ e) * w
        return ue, chi2, H, g
    px_est = cam.project3(p_cam)
    J_cam = cam.project3_jacobian(p_cam)
    px_diff = px - px_est
    pn2 = float(px_diff @ px_diff)
    f_diff = f - f_est
    fn2 = float(f_diff @ f_diff)
    e_img = float(grad @ px_diff)
    ratio = math.sqrt(fn2) / math.sqrt(pn2)
    e = e_img * ratio
    ue = abs(e)
    e *= R
    w = tukey_weight(e) if sigma != 0.0 else 0.0
    chi2 = 0.5 * e * e * w
    if want_jac:
        J_proj = jacobian_xyz2img_imu(T_cam_imu, p_imu, J_cam)
        J_bear = jacobian_xyz2f_imu(T_cam_imu, p_imu)
        J_img = grad @ ((-1.0) * J_proj)
        J_ftf = 2 * f_diff @ ((-1.0) * J_bear)
        J_ptp = 2 * px_diff @ ((-1.0) * J_proj)
        J_ratio = 0.5 * (1.0 / ratio) * (1 / (pn2 * pn2)) * (J_ftf * pn2 - J_ptp * fn2)
        J = (e_img * J_ratio + ratio * J_img) * R
        H, g = np.outer(J, J) * w, -(J * e) * w
    return ue, chi2, H, g


def optimize_pose(err_type, cams, T_imu_world, outlier_threshold, max_iter=10, eps=1e-6, R_prior=None, prior_lambda=0.0):
    """PoseOptimizer::run.  cams: list of dict(cam (np_restatement_direct.Cam), T_cam_imu (Tf), px 2n, f 3n, grad 2n, level n,
    type n, xyz_world 3n, usable n).  Returns dict(T, sigma, iters, n_meas, outlier (list of arrays), n_deleted_edges,
    n_deleted_corners, err_before, err_after, status)."""
    def feats():
        for ci, c in enumerate(cams):
            for i in range(len(c["level"])):
                if c["usable"][i]:
                    yield ci, c, i

    def eval_all(T, want_jac, sigma, start_errors=None):
        H, g = np.zeros((6, 6)), np.zeros(6)
        n = 0
        for ci, c, i in feats():
            scale = 1 << int(c["level"][i])
            edge = is_edgelet(int(c["type"][i]))
            ms = sigma * scale * (2.0 if edge else 1.0)
            ue, chi2, Hi, gi = residual(err_type, c["cam"], T, c["T_cam_imu"], c["px"][2 * i:2 * i + 2], c["f"][3 * i:3 * i + 3],
                                        c["grad"][2 * i:2 * i + 2], c["xyz_world"][3 * i:3 * i + 3], edge, ms, want_jac)
            if start_errors is not None:
                start_errors.append(f32(ue / scale))
            if want_jac:
                H += Hi; g += gi
            n += 1
        return H, g, n

    # start errors -> MAD scale (the residual functions divide by sigma: any non-zero value serves for the first call)
    start = []
    _, _, n_meas = eval_all(T_imu_world, False, 1.0, start)
    if n_meas == 0:
        return dict(status=1, n_meas=0)
    srt = np.sort(np.array(start, f32))
    sigma = float(f32(f32(1.48) * srt[len(srt) // 2]))            # nth_element at floor(n/2), 1.48f * median (float)
    T = T_imu_world
    T_old = T
    iters, status = 0, 0
    I_prior = np.zeros((6, 6))
    have_prior = R_prior is not None
    prior = Tf(np.asarray(R_prior, np.float64), np.zeros(3)) if have_prior else None
    for it in range(max_iter):
        iters = it + 1
        H, g, _ = eval_all(T, True, sigma)
        if have_prior:
            if it == 0:
                I_prior = np.zeros((6, 6))
                I_prior[3:, 3:] = np.eye(3) * (max(abs(H[j, j]) for j in range(3, 6)) * prior_lambda)
            H = H + I_prior
            g = g - I_prior @ tf_log(T * prior.inverse())
        try:
            dx = np.linalg.solve(H, g)
        except np.linalg.LinAlgError:
            dx = np.full(6, np.nan)
        if np.isnan(dx[0]):
            T = T_old
            status = 2
            break
        Tn = tf_exp(dx) * T
        Tn = Tf(Tn.q / math.sqrt(float(Tn.q @ Tn.q)), Tn.t)
        T_old, T = T, Tn
        if np.max(np.abs(dx)) < eps:
            break
    # removeOutliers with the optimised pose
    outlier, final = [], []
    n_edges = n_corners = 0
    for c in cams:
        o = np.zeros(len(c["level"]), np.uint8)
        for i in range(len(c["level"])):
            if not c["usable"][i]:
                continue
            edge = is_edgelet(int(c["type"][i]))
            ue, _, _, _ = residual(err_type, c["cam"], T, c["T_cam_imu"], c["px"][2 * i:2 * i + 2], c["f"][3 * i:3 * i + 3],
                                   c["grad"][2 * i:2 * i + 2], c["xyz_world"][3 * i:3 * i + 3], edge, 0.0, False)
            ue *= 1.0 / (1 << int(c["level"][i]))
            final.append(ue)
            if abs(ue) > outlier_threshold:
                o[i] = 1
                if edge:
                    n_edges += 1
                else:
                    n_corners += 1
        outlier.append(o)
    fs = np.sort(np.array(final))
    return dict(status=status, T=T, sigma=sigma, iters=iters, n_meas=n_meas, outlier=outlier, n_deleted_edges=n_edges,
                n_deleted_corners=n_corners, err_before=float(srt[len(srt) // 2]), err_after=float(fs[len(fs) // 2]))


# ---------------------------------------------------------------------------------------------------------------
# Point::optimize  (src/svo_common/src/point.cpp:248-325; Jacobians src/svo_common/include/svo/common/point.h:170-204)
# Written from those lines: a 3-DoF Gauss-Newton on the landmark's world position over its observations, residual on
# the unit plane (project2(f) - project2(p_in_f)) or between unit bearing vectors (f - p_in_f / |p_in_f|); the step is
# rolled back when the error grew (from the second iteration on) or the solve gave a NaN; stop when max |dp| <= 1e-10.
# Machinery: numpy matrices and numpy.linalg.solve for A.ldlt().solve(b) (3x3, positive definite for >= 2 views).
# ---------------------------------------------------------------------------------------------------------------
def point_jacobian_xyz2uv(p_in_f, R_f_w):             # point.h:170-184
    z_inv = 1.0 / p_in_f[2]
    J = np.array([[z_inv, 0.0, -p_in_f[0] * z_inv * z_inv],
                  [0.0, z_inv, -p_in_f[1] * z_inv * z_inv]])
    return -J @ R_f_w


def point_jacobian_xyz2f(p_in_f, R_f_w):              # point.h:187-204
    x, y, z = p_in_f
    Jn = np.array([[y * y + z * z, -x * y, -z * x],
                   [-x * y, x * x + z * z, -y * z],
                   [-z * x, -y * z, x * x + y * y]]) * (1.0 / (x * x + y * y + z * z) ** 1.5)
    return -Jn @ R_f_w


def point_optimize(obs, pos, n_iter, using_bearing_vector):
    """obs: list of (T_f_w as Tf, bearing vector f) -- the observations whose frame is still alive; pos: start position.
    Returns (position, iterations executed: the number of times the normal equations were built)."""
    pos = np.array(pos, np.float64)
    if len(obs) < 2:                                  # point.cpp:255-259: nothing is done
        return pos, 0
    old_point = pos.copy()
    chi2 = 0.0
    eps = 0.0000000001
    iters = 0
    for i in range(n_iter):
        A, b, new_chi2 = np.zeros((3, 3)), np.zeros(3), 0.0
        for T_f_w, f in obs:
            p_in_f = T_f_w.apply(pos)
            R_f_w = rot_matrix(T_f_w.q)
            if using_bearing_vector:                  # updateHessianGradientUnitSphere, :232-246
                J = point_jacobian_xyz2f(p_in_f, R_f_w)
                e = f - p_in_f / math.sqrt(float(p_in_f @ p_in_f))
            else:                                     # updateHessianGradientUnitPlane, :216-230; vk::project2 = (x/z, y/z)
                J = point_jacobian_xyz2uv(p_in_f, R_f_w)
                e = np.array([f[0] / f[2], f[1] / f[2]]) - np.array([p_in_f[0] / p_in_f[2], p_in_f[1] / p_in_f[2]])
            A += J.T @ J
            b -= J.T @ e
            new_chi2 += float(e @ e)
        iters += 1
        with np.errstate(all="ignore"):
            try:
                dp = np.linalg.solve(A, b)
            except np.linalg.LinAlgError:             # exactly singular: Eigen's LDLT returns something finite or NaN; treat as NaN
                dp = np.full(3, np.nan)
        if (i > 0 and new_chi2 > chi2) or np.isnan(dp[0]):
            pos = old_point                           # roll-back
            break
        new_point = pos + dp
        old_point = pos
        pos = new_point
        chi2 = new_chi2
        if np.max(np.abs(dp)) <= eps:
            break
    return pos, iters
