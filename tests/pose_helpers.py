"""Synthetic pose-optimisation problems shared by the CPU and GPU tests."""
import numpy as np

from svo_pro_universal_amd import _capi as capi, synth


def make_pose_scene(seed, n=180, cam=None, n_cams=1, noise_px=0.3, outlier_fraction=0.1, edgelet_fraction=0.3,
                    pose_err=(0.01, 0.03)):
    """Random 3-D points in front of the rig, observed with pixel noise (+ gross outliers), an initial pose off
    by pose_err (rad, m).  Returns dict(cams=[...], T_imu_world_gt, T_imu_world_init, inlier=[...])."""
    rng = np.random.RandomState(seed)
    cam = cam or synth.Camera.euroc_like(752, 480)
    T_imu_world_gt = synth.SE3(synth.quat_from_axis_angle(rng.normal(size=3), rng.uniform(0, 0.8)), rng.uniform(-1, 1, 3))
    cams, inliers = [], []
    for c in range(n_cams):
        T_cam_imu = synth.SE3(synth.quat_from_axis_angle(rng.normal(size=3), 0.1), rng.uniform(-0.1, 0.1, 3) + [0.1 * c, 0, 0])
        T_cam_world = T_cam_imu * T_imu_world_gt
        px_true = np.stack([rng.uniform(20, cam.width - 20, n), rng.uniform(20, cam.height - 20, n)])
        x, y = cam.undistorted_xy(px_true[0], px_true[1])
        depth = rng.uniform(1.0, 8.0, n)
        p_cam = np.stack([x, y, np.ones(n)]) * depth
        xyz_world = T_cam_world.inverse().transform(p_cam)
        level = rng.choice([0, 1, 2], n).astype(np.int32)
        px = px_true + rng.normal(0, noise_px, px_true.shape) * (1 << level)
        inl = np.ones(n, bool)
        n_out = int(outlier_fraction * n)
        if n_out:
            k = rng.choice(n, n_out, replace=False)
            px[:, k] += rng.uniform(15, 40, (2, n_out)) * rng.choice([-1, 1], (2, n_out))
            inl[k] = False
        xo, yo = cam.undistorted_xy(px[0], px[1])
        f = np.stack([xo, yo, np.ones(n)])
        f /= np.linalg.norm(f, axis=0)
        typ = np.where(rng.uniform(size=n) < edgelet_fraction, capi.FT_EDGELET, capi.FT_CORNER).astype(np.uint8)
        typ[::11] = capi.FT_CORNER_SEED_CONVERGED
        ang = rng.uniform(0, 2 * np.pi, n)
        grad = np.stack([np.cos(ang), np.sin(ang)])
        usable = np.ones(n, np.uint8)
        usable[::13] = 0
        inl &= usable.astype(bool)
        cams.append(dict(cam=cam, T_cam_imu=T_cam_imu, px=np.ascontiguousarray(px.T).ravel(), f=np.ascontiguousarray(f.T).ravel(),
                         grad=np.ascontiguousarray(grad.T).ravel(), level=level, type=typ,
                         xyz_world=np.ascontiguousarray(xyz_world.T).ravel(), usable=usable))
        inliers.append(inl)
    d = synth.SE3(synth.quat_from_axis_angle(rng.normal(size=3), pose_err[0]), rng.normal(size=3) / np.sqrt(3) * pose_err[1])
    return dict(cams=cams, T_imu_world_gt=T_imu_world_gt, T_imu_world_init=d * T_imu_world_gt, inlier=inliers, cam=cam)
