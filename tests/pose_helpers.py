"""Synthetic pose-optimisation problems shared by the CPU and GPU tests."""
import numpy as np

from svo_pro_universal_amd import _capi as capi, synth


def make_pose_scene(seed, n=180, cam=None, n_cams=1, noise_px=0.3, outlier_fraction=0.1, edgelet_fraction=0.3,
                    pose_err=(0.01, 0.03), T_cam0_world=None):
    """Random 3-D points in front of the rig, observed with pixel noise (+ gross outliers), an initial pose off
    by pose_err (rad, m).  T_cam0_world: camera 0's true pose (the rig's is derived from it).  Returns dict(cams=[...], T_imu_world_gt, T_imu_world_init, inlier=[...])."""
    rng = np.random.RandomState(seed)
    cam = cam or synth.Camera.euroc_like(752, 480)
    T_imu_world_gt = synth.SE3(synth.quat_from_axis_angle(rng.normal(size=3), rng.uniform(0, 0.8)), rng.uniform(-1, 1, 3))
    cams, inliers = [], []
    for c in range(n_cams):
        T_cam_imu = synth.SE3(synth.quat_from_axis_angle(rng.normal(size=3), 0.1), rng.uniform(-0.1, 0.1, 3) + [0.1 * c, 0, 0])
        if c == 0 and T_cam0_world is not None:   # the rig's true pose is the one that puts camera 0 at the given pose
            T_imu_world_gt = T_cam_imu.inverse() * T_cam0_world
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


def make_structure_scene(seed, n_points=300, n_views=5, noise=2e-3, start_err=0.08, degenerate=True):
    """Landmarks seen from a few keyframes (FrameHandlerBase::optimizeStructure's input): random views looking
    at a cloud of points, each point observed in 2..n_views of them with noisy bearing vectors, a starting
    position off by start_err (relative to the depth).  With `degenerate`: some points with 0/1 observations
    (left alone), one with two identical observations (singular system), one behind a camera.
    Returns dict(views=[7-vectors T_f_w], obs_begin, obs_view, obs_f, pos0, pos_gt)."""
    rng = np.random.RandomState(seed)
    views = []
    for v in range(n_views):
        T_w_f = synth.SE3(synth.quat_from_axis_angle(rng.normal(size=3), rng.uniform(0, 0.25)), rng.uniform(-0.6, 0.6, 3))
        views.append(T_w_f.inverse())
    pos_gt = np.stack([rng.uniform(-2, 2, n_points), rng.uniform(-1.5, 1.5, n_points), rng.uniform(3, 9, n_points)], 1)
    obs_begin, obs_view, obs_f = [0], [], []
    for i in range(n_points):
        k = rng.randint(2, n_views + 1)
        if degenerate and i % 37 == 5:
            k = i % 2          # 0 or 1 observation
        sel = rng.choice(n_views, k, replace=False)
        if degenerate and i == 11:
            sel = np.array([sel[0], sel[0]])     # the same view twice: rank-deficient normal equations
        for v in sel:
            p = views[v].transform(pos_gt[i])
            f = p / np.linalg.norm(p) + rng.normal(0, noise, 3)
            obs_view.append(int(v))
            obs_f.append(f / np.linalg.norm(f))
        obs_begin.append(len(obs_view))
    pos0 = pos_gt * (1.0 + rng.normal(0, start_err, (n_points, 1))) + rng.normal(0, 0.02, (n_points, 3))
    if degenerate and n_points > 23:
        pos0[23] = views[obs_view[obs_begin[23]]].inverse().transform(np.array([0.1, 0.1, -2.0]))   # behind its first view
    return dict(views=[v.as7() for v in views], obs_begin=np.array(obs_begin, np.int32), obs_view=np.array(obs_view, np.int32),
                obs_f=np.array(obs_f).reshape(-1, 3), pos0=pos0, pos_gt=pos_gt)
