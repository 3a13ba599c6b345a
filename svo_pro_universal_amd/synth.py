"""Deterministic synthetic worlds for the SVO direct front end.

A scene is a textured plane (analytic texture = sum of sinusoids, so both
frames are rendered exactly, without resampling an image) seen by a pinhole
camera (optionally with radial-tangential distortion) from a reference pose
and a current pose.  Features are placed on a jittered grid in the reference
image with ground-truth depth.

The array code is written against a tiny "xp" subset shared by numpy and
torch so that bench.py can render thousands of frames on the GPU while the
tests use numpy on the CPU.  Nothing here calls the oracle.
"""
import math

import numpy as np

# ----------------------------------------------------------------------------
# small SE3 helpers (numpy, host side; q = (w, x, y, z)) -- test/bench plumbing
# ----------------------------------------------------------------------------


def quat_from_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    n = np.linalg.norm(axis)
    if n == 0.0 or angle == 0.0:
        return np.array([1.0, 0.0, 0.0, 0.0])
    axis = axis / n
    s = math.sin(angle * 0.5)
    return np.array([math.cos(angle * 0.5), axis[0] * s, axis[1] * s, axis[2] * s])


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx])


def quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


class SE3(object):
    """q (w,x,y,z) unit quaternion + t; p_out = R(q) p + t."""

    def __init__(self, q=(1.0, 0.0, 0.0, 0.0), t=(0.0, 0.0, 0.0)):
        self.q = np.asarray(q, dtype=np.float64).copy()
        self.t = np.asarray(t, dtype=np.float64).copy()

    def R(self):
        return quat_to_R(self.q)

    def __mul__(self, o):
        return SE3(quat_mul(self.q, o.q), self.t + self.R() @ o.t)

    def inverse(self):
        qi = np.array([self.q[0], -self.q[1], -self.q[2], -self.q[3]])
        return SE3(qi, -(quat_to_R(qi) @ self.t))

    def transform(self, p):
        p = np.asarray(p, dtype=np.float64)
        return (self.R() @ p.reshape(3, -1)).reshape(p.shape) + (self.t if p.ndim == 1 else self.t[:, None])

    def as7(self):
        return np.concatenate([self.q, self.t])

    @staticmethod
    def from7(v):
        return SE3(v[:4], v[4:7])


def se3_error(A, B):
    """(rotation angle [rad], translation distance [m]) between two SE3."""
    D = A.inverse() * B
    w = min(1.0, abs(float(D.q[0])))
    return 2.0 * math.acos(w), float(np.linalg.norm(D.t))


# ----------------------------------------------------------------------------
# camera
# ----------------------------------------------------------------------------


class Camera(object):
    def __init__(self, width=640, height=480, fx=320.0, fy=320.0, cx=320.0, cy=240.0, dist=None):
        self.width, self.height = int(width), int(height)
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.dist = None if dist is None else [float(x) for x in dist]  # k1 k2 p1 p2

    @staticmethod
    def test_camera():
        """PinholeGeometry::createTestCamera(): 640x480, f = 320 (camera_geometry.hpp:76-82)."""
        return Camera()

    @staticmethod
    def euroc_like(width=640, height=480):
        """EuRoC cam0 radtan coefficients (examples/param/calib/euroc_mono.yaml) on a 640x480 sensor."""
        return Camera(width, height, 458.654 * width / 752.0, 457.296, 367.215 * width / 752.0, 248.375,
                      dist=[-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05])

    # normalised, undistorted image-plane coordinates of pixel grids (xp arrays)
    def undistorted_xy(self, u, v):
        x = (u - self.cx) / self.fx
        y = (v - self.cy) / self.fy
        if self.dist is None:
            return x, y
        k1, k2, p1, p2 = self.dist
        x0, y0 = x, y
        for _ in range(12):  # more iterations than the reference's 5: rendering wants the true inverse
            xx, yy, xy = x * x, y * y, x * y
            r2 = xx + yy
            ic = 1.0 / (1.0 + (k1 + k2 * r2) * r2)
            dx = p1 * 2 * xy + p2 * (r2 + 2.0 * xx)
            dy = p2 * 2 * xy + p1 * (r2 + 2.0 * yy)
            x = (x0 - dx) * ic
            y = (y0 - dy) * ic
        return x, y

    def project(self, p):
        """numpy, p: 3xN -> 2xN (with distortion)."""
        x = p[0] / p[2]
        y = p[1] / p[2]
        if self.dist is not None:
            k1, k2, p1, p2 = self.dist
            xx, yy, xy = x * x, y * y, x * y
            r2 = xx + yy
            cd = (k1 + k2 * r2) * r2
            x, y = (x + x * cd + p1 * 2 * xy + p2 * (r2 + 2 * xx),
                    y + y * cd + p2 * 2 * xy + p1 * (r2 + 2 * yy))
        return np.stack([self.fx * x + self.cx, self.fy * y + self.cy])


# ----------------------------------------------------------------------------
# scene
# ----------------------------------------------------------------------------


class Texture(object):
    """I(s,t) = 128 + sum_k A_k sin(2 pi (fs_k s + ft_k t) + phi_k)."""

    def __init__(self, rng, n_waves=28, lam_min=0.03, lam_max=2.5, contrast=46.0):
        lam = np.exp(rng.uniform(math.log(lam_min), math.log(lam_max), n_waves))
        ang = rng.uniform(0.0, 2 * math.pi, n_waves)
        self.fs = np.cos(ang) / lam
        self.ft = np.sin(ang) / lam
        self.phi = rng.uniform(0.0, 2 * math.pi, n_waves)
        amp = lam ** 0.6
        self.amp = amp * (contrast / math.sqrt(0.5 * np.sum(amp ** 2)))

    def eval(self, s, t, xp=np):
        acc = None
        for k in range(len(self.amp)):
            term = float(self.amp[k]) * xp.sin((2 * math.pi * float(self.fs[k])) * s
                                                + (2 * math.pi * float(self.ft[k])) * t + float(self.phi[k]))
            acc = term if acc is None else acc + term
        return acc + 128.0


class Plane(object):
    """n . X = h in world coordinates, with an in-plane orthonormal basis."""

    def __init__(self, n, h):
        n = np.asarray(n, dtype=np.float64)
        self.n = n / np.linalg.norm(n)
        self.h = float(h)
        a = np.array([1.0, 0.0, 0.0]) if abs(self.n[0]) < 0.9 else np.array([0.0, 1.0, 0.0])
        e1 = a - self.n * (a @ self.n)
        self.e1 = e1 / np.linalg.norm(e1)
        self.e2 = np.cross(self.n, self.e1)


def render(cam, T_w_c, plane, tex, xp=np, device=None, dtype=None, gain=1.0, offset=0.0):
    """Render the u8 image seen by `cam` at pose T_w_c (camera -> world)."""
    if xp is np:
        u = np.arange(cam.width, dtype=np.float64)[None, :].repeat(cam.height, 0)
        v = np.arange(cam.height, dtype=np.float64)[:, None].repeat(cam.width, 1)
    else:
        dtype = dtype or xp.float64
        u = xp.arange(cam.width, dtype=dtype, device=device)[None, :].expand(cam.height, cam.width)
        v = xp.arange(cam.height, dtype=dtype, device=device)[:, None].expand(cam.height, cam.width)
    x, y = cam.undistorted_xy(u, v)
    R = T_w_c.R()
    o = T_w_c.t
    dx = float(R[0, 0]) * x + float(R[0, 1]) * y + float(R[0, 2])
    dy = float(R[1, 0]) * x + float(R[1, 1]) * y + float(R[1, 2])
    dz = float(R[2, 0]) * x + float(R[2, 1]) * y + float(R[2, 2])
    n = plane.n
    denom = float(n[0]) * dx + float(n[1]) * dy + float(n[2]) * dz
    lam = (plane.h - float(n @ o)) / denom
    X = float(o[0]) + lam * dx
    Y = float(o[1]) + lam * dy
    Z = float(o[2]) + lam * dz
    e1, e2 = plane.e1, plane.e2
    s = float(e1[0]) * X + float(e1[1]) * Y + float(e1[2]) * Z
    t = float(e2[0]) * X + float(e2[1]) * Y + float(e2[2]) * Z
    img = tex.eval(s, t, xp) * gain + offset
    if xp is np:
        return np.clip(np.floor(img + 0.5), 0, 255).astype(np.uint8)
    return xp.clip(xp.floor(img + 0.5), 0, 255).to(xp.uint8)


def render_batch_torch(cam, poses, planes, texs, device, chunk=32, gains=None, offsets=None):
    """torch only: render len(poses) u8 images (N x H x W) in chunks, vectorised
    over the scene parameters (same arithmetic as render(), fp64).  gains / offsets: optional per-image
    illumination change I -> gain * I + offset before quantisation."""
    import torch
    n = len(poses)
    out = torch.empty((n, cam.height, cam.width), dtype=torch.uint8, device=device)
    u = torch.arange(cam.width, dtype=torch.float64, device=device)[None, :].expand(cam.height, cam.width)
    v = torch.arange(cam.height, dtype=torch.float64, device=device)[:, None].expand(cam.height, cam.width)
    x, y = cam.undistorted_xy(u, v)
    x, y = x[None], y[None]

    def col(vals):
        return torch.tensor(np.asarray(vals, dtype=np.float64), device=device)

    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        R = col([poses[i].R() for i in range(c0, c1)])            # b x 3 x 3
        o = col([poses[i].t for i in range(c0, c1)])              # b x 3
        nn = col([planes[i].n for i in range(c0, c1)])
        hh = col([planes[i].h for i in range(c0, c1)])
        e1 = col([planes[i].e1 for i in range(c0, c1)])
        e2 = col([planes[i].e2 for i in range(c0, c1)])
        B3 = lambda t, k: t[:, k][:, None, None]
        dx = R[:, 0, 0][:, None, None] * x + R[:, 0, 1][:, None, None] * y + R[:, 0, 2][:, None, None]
        dy = R[:, 1, 0][:, None, None] * x + R[:, 1, 1][:, None, None] * y + R[:, 1, 2][:, None, None]
        dz = R[:, 2, 0][:, None, None] * x + R[:, 2, 1][:, None, None] * y + R[:, 2, 2][:, None, None]
        denom = B3(nn, 0) * dx + B3(nn, 1) * dy + B3(nn, 2) * dz
        lam = ((hh - (nn * o).sum(1))[:, None, None]) / denom
        X = B3(o, 0) + lam * dx
        Y = B3(o, 1) + lam * dy
        Z = B3(o, 2) + lam * dz
        s = B3(e1, 0) * X + B3(e1, 1) * Y + B3(e1, 2) * Z
        t = B3(e2, 0) * X + B3(e2, 1) * Y + B3(e2, 2) * Z
        amp = col([texs[i].amp for i in range(c0, c1)])
        fs = col([texs[i].fs for i in range(c0, c1)])
        ft = col([texs[i].ft for i in range(c0, c1)])
        phi = col([texs[i].phi for i in range(c0, c1)])
        acc = torch.full_like(s, 128.0)
        for k in range(amp.shape[1]):
            acc += B3(amp, k) * torch.sin((2 * math.pi) * (B3(fs, k) * s + B3(ft, k) * t) + B3(phi, k))
        if gains is not None:
            acc = acc * col(gains[c0:c1])[:, None, None]
        if offsets is not None:
            acc = acc + col(offsets[c0:c1])[:, None, None]
        out[c0:c1] = torch.clip(torch.floor(acc + 0.5), 0, 255).to(torch.uint8)
    return out


class AlignScene(object):
    """One (reference frame, current frame) pair with features and ground truth."""
    pass


def make_align_scene(seed, n_features=2000, patch_size=4, cam=None, max_level=4,
                     rot_deg=(0.0, 2.0), trans_m=(0.0, 0.05), with_extrinsics=True,
                     border_features=0, invalid_fraction=0.0, gain=1.0, offset=0.0,
                     xp=np, device=None, render_images=True):
    """SURVEY.md 8(d) config C2 (P=4) / C2' (P=8).

    Returns an AlignScene with
      cam, img_ref, img_cur (HxW u8, level 0), T_ref_f_w, T_cur_f_w_gt (cam<-world),
      T_cam_imu, px (2xN), f (3xN), pos_world (3xN), flags (N),
      T_icur_iref_gt (SE3), and the initial guess T_icur_iref_init = identity.
    """
    rng = np.random.RandomState(seed)
    cam = cam or Camera.test_camera()
    sc = AlignScene()
    sc.seed, sc.cam, sc.patch_size = seed, cam, patch_size

    # plane in front of the reference camera, tilted up to ~20 degrees
    depth0 = rng.uniform(1.5, 6.0)
    tilt = np.deg2rad(rng.uniform(0.0, 20.0))
    tdir = rng.uniform(0, 2 * math.pi)
    n_c = np.array([math.sin(tilt) * math.cos(tdir), math.sin(tilt) * math.sin(tdir), math.cos(tilt)])

    # reference pose in the world: arbitrary, so world != camera frame
    q_wr = quat_from_axis_angle(rng.normal(size=3), rng.uniform(0.0, 0.6))
    T_w_ref = SE3(q_wr, rng.uniform(-2.0, 2.0, 3))
    n_w = T_w_ref.R() @ n_c
    p_on_plane = T_w_ref.transform(np.array([0.0, 0.0, depth0]))
    plane = Plane(n_w, float(n_w @ p_on_plane))
    tex = Texture(rng)

    # relative motion (current camera expressed in the reference camera)
    ang = np.deg2rad(rng.uniform(*rot_deg))
    q_rc = quat_from_axis_angle(rng.normal(size=3), ang)
    tdirv = rng.normal(size=3)
    tdirv /= np.linalg.norm(tdirv)
    T_ref_cur = SE3(q_rc, tdirv * rng.uniform(*trans_m))
    T_w_cur = T_w_ref * T_ref_cur

    if with_extrinsics:
        T_cam_imu = SE3(quat_from_axis_angle(rng.normal(size=3), 0.05), rng.uniform(-0.05, 0.05, 3))
    else:
        T_cam_imu = SE3()
    sc.T_cam_imu = T_cam_imu
    sc.T_imu_cam = T_cam_imu.inverse()
    sc.T_ref_f_w = T_w_ref.inverse()
    sc.T_cur_f_w_gt = T_w_cur.inverse()
    sc.plane, sc.tex = plane, tex
    sc.T_w_ref, sc.T_w_cur = T_w_ref, T_w_cur

    if render_images:
        sc.img_ref = render(cam, T_w_ref, plane, tex, xp=xp, device=device)
        sc.img_cur = render(cam, T_w_cur, plane, tex, xp=xp, device=device, gain=gain, offset=offset)

    # jittered grid of features, far enough from the border for every level
    margin = (1 << max_level) * (patch_size + 3)
    w_in, h_in = cam.width - 2 * margin, cam.height - 2 * margin
    nx = max(1, int(math.ceil(math.sqrt(n_features * w_in / float(h_in)))))
    ny = max(1, int(math.ceil(n_features / float(nx))))
    gx, gy = np.meshgrid(np.arange(nx), np.arange(ny))
    gx, gy = gx.ravel()[:n_features], gy.ravel()[:n_features]
    px = np.stack([margin + (gx + rng.uniform(0.05, 0.95, gx.size)) * (w_in / float(nx)),
                   margin + (gy + rng.uniform(0.05, 0.95, gy.size)) * (h_in / float(ny))])
    if border_features > 0:  # features that a-3 must reject / that leave the image
        bx = rng.uniform(0, cam.width - 1, border_features)
        by = rng.uniform(0, cam.height - 1, border_features)
        side = rng.randint(0, 4, border_features)
        bx = np.where(side == 0, rng.uniform(0, margin, border_features), bx)
        bx = np.where(side == 1, cam.width - 1 - rng.uniform(0, margin, border_features), bx)
        by = np.where(side == 2, rng.uniform(0, margin, border_features), by)
        by = np.where(side == 3, cam.height - 1 - rng.uniform(0, margin, border_features), by)
        bpx = np.stack([bx, by])
        k = rng.permutation(px.shape[1] + border_features)
        px = np.concatenate([px, bpx], axis=1)[:, k]
    n = px.shape[1]
    x, y = cam.undistorted_xy(px[0], px[1])
    ray = np.stack([x, y, np.ones_like(x)])
    f = ray / np.linalg.norm(ray, axis=0, keepdims=True)
    # ground-truth depth along the bearing vector
    n_cam = T_w_ref.R().T @ plane.n
    h_cam = plane.h - float(plane.n @ T_w_ref.t)
    dist = h_cam / (n_cam @ f)
    pos_cam = f * dist
    sc.px = np.ascontiguousarray(px.T).ravel().copy()            # col-major 2xN
    sc.f = np.ascontiguousarray(f.T).ravel().copy()
    sc.pos_world = np.ascontiguousarray(T_w_ref.transform(pos_cam).T).ravel().copy()
    flags = np.ones(n, dtype=np.uint8)
    if invalid_fraction > 0:
        flags[rng.uniform(size=n) < invalid_fraction] = 0
    sc.flags = flags
    sc.n_features = n
    sc.depth = dist

    # ground truth for the optimised state: T_icur_iref = cur.T_imu_world * ref.T_imu_world^-1
    T_imu_world_ref = sc.T_imu_cam * sc.T_ref_f_w
    T_imu_world_cur = sc.T_imu_cam * sc.T_cur_f_w_gt
    sc.T_icur_iref_gt = T_imu_world_cur * T_imu_world_ref.inverse()
    sc.T_icur_iref_init = SE3()
    sc.ref_pos = T_w_ref.t.copy()
    return sc


# ----------------------------------------------------------------------------
# KLT tracks and depth-filter seeds on top of an AlignScene
# ----------------------------------------------------------------------------


def make_track_set(sc, n_tracks=400, seed=0, margin=40):
    """KLT-synth: integer reference pixels on a jittered grid, the
    tracker's start value (= last position = reference pixel) and the true
    position in the current frame."""
    rng = np.random.RandomState(seed + 7919)
    cam = sc.cam
    nx = int(math.ceil(math.sqrt(n_tracks * cam.width / float(cam.height))))
    ny = int(math.ceil(n_tracks / float(nx)))
    gx, gy = np.meshgrid(np.arange(nx), np.arange(ny))
    gx, gy = gx.ravel()[:n_tracks], gy.ravel()[:n_tracks]
    w_in, h_in = cam.width - 2 * margin, cam.height - 2 * margin
    px = np.stack([margin + (gx + rng.uniform(0.1, 0.9, gx.size)) * (w_in / float(nx)),
                   margin + (gy + rng.uniform(0.1, 0.9, gy.size)) * (h_in / float(ny))])
    px_ref = np.floor(px).astype(np.int32)                      # detector returns integer positions
    x, y = cam.undistorted_xy(px_ref[0].astype(np.float64), px_ref[1].astype(np.float64))
    ray = np.stack([x, y, np.ones_like(x)])
    n_cam = sc.T_w_ref.R().T @ sc.plane.n
    h_cam = sc.plane.h - float(sc.plane.n @ sc.T_w_ref.t)
    lam = h_cam / (n_cam @ ray)
    Xw = sc.T_w_ref.transform(ray * lam)
    Xc = sc.T_w_cur.inverse().transform(Xw)
    px_true = cam.project(Xc)
    return dict(px_ref=np.ascontiguousarray(px_ref.T).ravel().copy(),
                px_cur_init=np.ascontiguousarray(px_ref.T.astype(np.float64)).ravel().copy(),
                px_true=np.ascontiguousarray(px_true.T).ravel().copy())


def make_seed_set(sc, n_seeds=3000, seed=0, margin=30, edgelet_fraction=0.3, depth_noise=(0.7, 1.3),
                  levels=(0, 1, 2)):
    """C4-synth: seeds of a reference keyframe with inverse-depth state
    [mu, sigma2, a, b] initialised like depth_filter_utils::initializeSeeds
    (depth_filter.cpp:349-361): mu = 1/d_guess, sigma2 = mu_range^2/36, a = b = 10."""
    rng = np.random.RandomState(seed + 104729)
    cam = sc.cam
    px = np.stack([rng.uniform(margin, cam.width - margin, n_seeds), rng.uniform(margin, cam.height - margin, n_seeds)])
    px = np.floor(px)  # detector positions
    x, y = cam.undistorted_xy(px[0], px[1])
    ray = np.stack([x, y, np.ones_like(x)])
    f = ray / np.linalg.norm(ray, axis=0, keepdims=True)
    n_cam = sc.T_w_ref.R().T @ sc.plane.n
    h_cam = sc.plane.h - float(sc.plane.n @ sc.T_w_ref.t)
    dist = h_cam / (n_cam @ f)                                   # true depth along the bearing vector
    d_min = float(dist.min()) * 0.5
    mu_range = 1.0 / d_min                                       # seed::getMeanRangeFromDepthMinMax
    mu0 = 1.0 / (dist * rng.uniform(depth_noise[0], depth_noise[1], n_seeds))
    state = np.stack([mu0, np.full(n_seeds, mu_range * mu_range / 36.0), np.full(n_seeds, 10.0), np.full(n_seeds, 10.0)])
    is_edge = rng.uniform(size=n_seeds) < edgelet_fraction
    ftype = np.where(is_edge, 0, 1).astype(np.uint8)             # kEdgeletSeed / kCornerSeed
    ang = rng.uniform(0, 2 * math.pi, n_seeds)
    grad = np.stack([np.cos(ang), np.sin(ang)])
    level = rng.choice(np.asarray(levels, dtype=np.int32), n_seeds).astype(np.int32)
    return dict(px=np.ascontiguousarray(px.T).ravel().copy(), f=np.ascontiguousarray(f.T).ravel().copy(),
                grad=np.ascontiguousarray(grad.T).ravel().copy(), level=level, type=ftype,
                state=np.ascontiguousarray(state.T).ravel().copy(), mu_range=mu_range, true_depth=dist,
                ref_frame_idx=np.zeros(n_seeds, np.int32))
