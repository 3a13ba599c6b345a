"""CPU checks of the pose-optimiser restatement (oracle/svo_oracle_pose.c)."""
import numpy as np
import pytest

from svo_pro_universal_amd import _capi as capi, frontend as fe, synth

import pose_helpers as ph


def test_jacobians_against_finite_differences(oracle_lib):
    """The one asserted numeric test of the reference on this path (src/svo/test/test_frame.cpp:131-157):
    Frame::jacobian_xyz2uv_imu & co. agree with finite differences of the perturbed pose, tol 1e-6 / 1e-5."""
    rng = np.random.RandomState(3)
    cam = synth.Camera.euroc_like(752, 480)
    for trial in range(20):
        T_cam_imu = synth.SE3(synth.quat_from_axis_angle(rng.normal(size=3), rng.uniform(0, 1.0)), rng.uniform(-0.3, 0.3, 3))
        p_imu = T_cam_imu.inverse().transform(np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(1.5, 6.0)]))
        Juv, Jf, Jimg = oracle_lib.pose_jacobians(T_cam_imu, p_imu, cam)

        def funcs(dx):
            # the point moves as exp(dx) applied in the imu frame: p' = p + dx[:3] + dx[3:] x p (first order)
            T = synth.SE3(synth.quat_from_axis_angle(dx[3:] if np.linalg.norm(dx[3:]) > 0 else [1, 0, 0], np.linalg.norm(dx[3:])), dx[:3])
            pc = T_cam_imu.transform(T.transform(p_imu))
            return pc[:2] / pc[2], pc / np.linalg.norm(pc), cam.project(pc.reshape(3, 1))[:, 0]
        eps = 1e-7
        n_uv, n_f, n_img = np.zeros((2, 6)), np.zeros((3, 6)), np.zeros((2, 6))
        for k in range(6):
            d = np.zeros(6); d[k] = eps
            a, b = funcs(d), funcs(-d)
            n_uv[:, k] = (a[0] - b[0]) / (2 * eps); n_f[:, k] = (a[1] - b[1]) / (2 * eps); n_img[:, k] = (a[2] - b[2]) / (2 * eps)
        # jacobian_xyz2uv_imu is the Jacobian of the ERROR (measurement - projection): minus the projection's
        assert np.abs(Juv + n_uv).max() < 1e-6
        assert np.abs(Jf - n_f).max() < 1e-6
        assert np.abs(Jimg - n_img).max() < 1e-5 * max(1.0, np.abs(n_img).max())


@pytest.mark.parametrize("error_type", [capi.POSE_ERR_UNIT_PLANE, capi.POSE_ERR_BEARING_DIFF, capi.POSE_ERR_IMAGE_PLANE])
@pytest.mark.parametrize("n_cams", [1, 2])
def test_pose_recovery_and_outlier_rejection(oracle_lib, error_type, n_cams):
    sc = ph.make_pose_scene(11 + n_cams, n_cams=n_cams)
    opt = capi.default_pose_options(sc["cam"], error_type=error_type)
    pb, keep = fe.make_pose_problem(sc["cams"], sc["T_imu_world_init"])
    r = oracle_lib.optimize_pose(opt, pb)
    assert r.status == 0 and 2 <= r.iters <= 10
    e0 = synth.se3_error(sc["T_imu_world_init"], sc["T_imu_world_gt"])
    e1 = synth.se3_error(synth.SE3.from7(fe.se3_to_numpy(r.T_imu_world)), sc["T_imu_world_gt"])
    assert e1[0] < 0.1 * e0[0] and e1[1] < 0.1 * e0[1], (e0, e1)
    n_usable = sum(int(c["usable"].sum()) for c in sc["cams"])
    assert r.n_meas == n_usable
    flagged = np.concatenate([k["outlier"][:len(c["level"])] for k, c in zip(keep, sc["cams"])]).astype(bool)
    inlier = np.concatenate(sc["inlier"]); usable = np.concatenate([c["usable"] for c in sc["cams"]]).astype(bool)
    gross = usable & ~inlier
    assert flagged[gross].mean() > 0.75                  # the gross outliers are removed (an edgelet only sees the offset across its edge) ...
    assert flagged[usable & inlier].mean() < 0.1         # ... and the inliers kept
    assert r.n_deleted_edges + r.n_deleted_corners == int(flagged.sum()) and not flagged[~usable].any()
    assert r.reproj_error_after < r.reproj_error_before
    assert r.measurement_sigma == pytest.approx(1.48 * r.reproj_error_before, rel=1e-6)


def test_rotation_prior_and_degenerate_inputs(oracle_lib):
    sc = ph.make_pose_scene(21, n=60, outlier_fraction=0.0)
    opt0 = capi.default_pose_options(sc["cam"])
    pb, keep = fe.make_pose_problem(sc["cams"], sc["T_imu_world_init"])
    r0 = oracle_lib.optimize_pose(opt0, pb)
    # a strong prior on the (wrong) initial rotation keeps the rotation near it
    q_init = sc["T_imu_world_init"].as7()[:4]
    opt1 = capi.default_pose_options(sc["cam"], have_rotation_prior=1, prior_lambda=1e4, R_prior=q_init)
    r1 = oracle_lib.optimize_pose(opt1, pb)
    rot = lambda r, ref: synth.se3_error(synth.SE3.from7(fe.se3_to_numpy(r.T_imu_world)), ref)[0]
    assert rot(r1, sc["T_imu_world_init"]) < 0.1 * rot(r0, sc["T_imu_world_init"])
    # nothing usable -> status 1, pose untouched
    for c in sc["cams"]:
        c["usable"][:] = 0
    pbe, keepe = fe.make_pose_problem(sc["cams"], sc["T_imu_world_init"])
    re = oracle_lib.optimize_pose(opt0, pbe)
    assert re.status == 1 and re.n_meas == 0 and np.array_equal(fe.se3_to_numpy(re.T_imu_world), sc["T_imu_world_init"].as7())
