"""Thin Python plumbing over the C ABI (include/svo_hip.h) for tests and bench.

This is not the product's host layer (that is C++, svo_pro_universal_amd/host/,
mirroring the reference's SparseImgAlignBase etc.); it only marshals numpy /
device pointers into the POD structs and raises on non-zero status codes.
"""
import ctypes as C

import numpy as np

from . import _capi as capi


class SvohError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "svoh error %d: %s" % (code, msg))
        self.code = code


def _se3(T):
    s = capi.svoh_se3()
    v = T.as7() if hasattr(T, "as7") else np.asarray(T, dtype=np.float64)
    for i in range(4):
        s.q[i] = float(v[i])
    for i in range(3):
        s.t[i] = float(v[4 + i])
    return s


def _camera(cam):
    c = capi.svoh_camera()
    c.fx, c.fy, c.cx, c.cy = cam.fx, cam.fy, cam.cx, cam.cy
    c.width, c.height = cam.width, cam.height
    if cam.dist is None:
        c.distortion = capi.SVOH_DISTORTION_NONE
    else:
        c.distortion = capi.SVOH_DISTORTION_RADTAN
        for i in range(4):
            c.d[i] = cam.dist[i]
    return c


def se3_to_numpy(s):
    return np.array([s.q[0], s.q[1], s.q[2], s.q[3], s.t[0], s.t[1], s.t[2]])


class Context(object):
    """One svoh_ctx (one HIP stream).  Not thread-safe, like the ABI."""

    def __init__(self, device=0, kernel_timing=True, lib=None):
        """kernel_timing: bracket every launch with an event pair so that the *_kernel_ms calls work (what the
        benchmark and the tests want; the library's own default is off: svoh_set_kernel_timing).
        lib: another build of libsvo_hip loaded with capi.load(path) (the test-hook build)."""
        self.lib = lib or capi.load()
        h = C.c_void_p()
        rc = self.lib.svoh_create(int(device), C.byref(h))
        if rc != 0:
            raise SvohError(rc, self.lib.svoh_last_error_string(None).decode())
        self.h = h
        self._keep = []
        self.set_kernel_timing(kernel_timing)

    def set_kernel_timing(self, on):
        self._check(self.lib.svoh_set_kernel_timing(self.h, 1 if on else 0))

    def set_align_geometry_classes(self, shared):
        """svoh_set_align_geometry_classes: True = one launch geometry for every alignment problem below 512 patches (a lock-step round of
        streams of different sizes is then one or two launches instead of up to four)."""
        self._check(self.lib.svoh_set_align_geometry_classes(self.h, 1 if shared else 0))

    def close(self):
        if self.h:
            self.lib.svoh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise SvohError(rc, self.lib.svoh_last_error_string(self.h).decode())

    def synchronize(self):
        self._check(self.lib.svoh_synchronize(self.h))

    def stream(self):
        return self.lib.svoh_stream(self.h)

    def reload_knobs(self):
        """The SVOH_* tuning knobs are read from the environment when the context is made; read them again."""
        self._check(self.lib.svoh_reload_knobs(self.h))

    def camera_maths(self, cam, xyz):
        """svoh_camera_maths: (px 2n, J 6n, f_back 3n) of the n points xyz (3n) as the DEVICE evaluates svoh_math.h."""
        xyz = np.ascontiguousarray(xyz, np.float64).ravel()
        n = xyz.size // 3
        px, J, fb = np.zeros(2 * n), np.zeros(6 * n), np.zeros(3 * n)
        c = _camera(cam)
        self._check(self.lib.svoh_camera_maths(self.h, C.byref(c), n, xyz.ctypes.data, px.ctypes.data, J.ctypes.data,
                                               fb.ctypes.data))
        return px, J, fb

    # ---- frames -----------------------------------------------------------
    def upload_pyramid(self, levels):
        n = len(levels)
        levels = [np.ascontiguousarray(lv, dtype=np.uint8) for lv in levels]
        ptrs = (C.c_void_p * n)(*[lv.ctypes.data for lv in levels])
        w = (C.c_int * n)(*[lv.shape[1] for lv in levels])
        h = (C.c_int * n)(*[lv.shape[0] for lv in levels])
        p = (C.c_int * n)(*[lv.strides[0] for lv in levels])
        out = capi.svoh_frame_t()
        self._check(self.lib.svoh_upload_pyramid(self.h, n, ptrs, w, h, p, C.byref(out)))
        return out.value

    def build_pyramid(self, img, n_levels, rounding=capi.SVOH_HALFSAMPLE_REFERENCE, return_levels=False):
        """img: HxW uint8 numpy array (host).  Returns handle [, list of level arrays]."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        out = capi.svoh_frame_t()
        host = None
        if return_levels:
            lv = [np.zeros((h >> i, w >> i), dtype=np.uint8) for i in range(n_levels)]
            host = (C.c_void_p * n_levels)(*[a.ctypes.data for a in lv])
        self._check(self.lib.svoh_build_pyramid(self.h, img.ctypes.data, w, h, img.strides[0], capi.SVOH_MEM_HOST,
                                                n_levels, rounding, host, C.byref(out)))
        return (out.value, lv) if return_levels else out.value

    def build_pyramid_batch_device(self, dev_ptr, image_stride, n_images, width, height, pitch, n_levels,
                                   rounding=capi.SVOH_HALFSAMPLE_REFERENCE):
        out = (capi.svoh_frame_t * n_images)()
        self._check(self.lib.svoh_build_pyramid_batch(self.h, C.c_void_p(dev_ptr), image_stride, n_images, width,
                                                      height, pitch, capi.SVOH_MEM_DEVICE, n_levels, rounding, out))
        return [int(x) for x in out]

    def build_pyramid_batch_host(self, imgs, n_levels, rounding=capi.SVOH_HALFSAMPLE_REFERENCE):
        """imgs: NxHxW uint8"""
        imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
        n, h, w = imgs.shape
        out = (capi.svoh_frame_t * n)()
        self._check(self.lib.svoh_build_pyramid_batch(self.h, imgs.ctypes.data, h * w, n, w, h, w,
                                                      capi.SVOH_MEM_HOST, n_levels, rounding, out))
        return [int(x) for x in out]

    def download_level(self, frame, level):
        w, h = C.c_int(), C.c_int()
        self._check(self.lib.svoh_download_level(self.h, frame, level, None, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), dtype=np.uint8)
        self._check(self.lib.svoh_download_level(self.h, frame, level, out.ctypes.data, C.byref(w), C.byref(h)))
        return out

    def release_frame(self, frame):
        self._check(self.lib.svoh_release_frame(self.h, frame))

    # ---- sparse image alignment --------------------------------------------
    def sparse_align(self, opt, problems):
        """problems: ctypes array of svoh_align_problem (see make_align_problems)."""
        n = len(problems)
        res = (capi.svoh_align_result * n)()
        self._check(self.lib.svoh_sparse_align_batch(self.h, C.byref(opt), n, problems, res))
        return res

    def sparse_align_enqueue(self, opt, problems):
        self._check(self.lib.svoh_sparse_align_enqueue(self.h, C.byref(opt), len(problems), problems))

    def sparse_align_fetch(self, n):
        res = (capi.svoh_align_result * n)()
        self._check(self.lib.svoh_sparse_align_fetch(self.h, n, res))
        return res

    def sparse_align_fetch_all(self, n_total):
        """Results of every launch queued since the last fetch, in launch order."""
        res = (capi.svoh_align_result * n_total)()
        self._check(self.lib.svoh_sparse_align_fetch_all(self.h, n_total, res))
        return res

    def sparse_align_evaluate(self, opt, problem, level):
        H = np.zeros(64)
        g = np.zeros(8)
        chi2, nm, nsel = C.c_double(), C.c_int32(), C.c_int32()
        ntot = sum(problem.cams[i].n_features for i in range(problem.n_cams))
        vis = np.zeros(max(1, ntot), np.uint8)
        self._check(self.lib.svoh_sparse_align_evaluate(self.h, C.byref(opt), C.byref(problem), level,
                                                        H.ctypes.data, g.ctypes.data, C.byref(chi2), C.byref(nm),
                                                        vis.ctypes.data, C.byref(nsel)))
        return H.reshape(8, 8).T.copy(), g, chi2.value, nm.value, vis[:nsel.value].copy()

    # ---- patch-split Gauss-Newton; the loop itself is split_align.gauss_newton_split ----
    def split_init(self, problem, d_state):
        self._check(self.lib.svoh_sparse_align_split_init(self.h, C.byref(problem), C.c_void_p(d_state)))

    def partial_sums(self, opt, problem, level, d_state, d_sums, n_workgroups=0):
        self._check(self.lib.svoh_sparse_align_partial_sums(self.h, C.byref(opt), C.byref(problem), level,
                                                            n_workgroups, C.c_void_p(d_state), C.c_void_p(d_sums)))

    def gn_update(self, opt, problem, level, it, d_sums, d_state):
        st = capi.svoh_align_gn_state()
        self._check(self.lib.svoh_sparse_align_gn_update(self.h, C.byref(opt), C.byref(problem), level, it,
                                                         C.c_void_p(d_sums), C.c_void_p(d_state), C.byref(st)))
        return st


def fill_align_camera(cam_struct, scene, ref_frame, cur_frame, keep, device_ptrs=None):
    """Fill one svoh_align_camera from a synth.AlignScene.  device_ptrs: optional
    dict(px=, f=, pos_world=, flags=) of device addresses (SVOH_MEM_DEVICE)."""
    cam_struct.ref_frame = ref_frame
    cam_struct.cur_frame = cur_frame
    cam_struct.cam = _camera(scene.cam)
    cam_struct.ref_T_imu_cam = _se3(scene.T_imu_cam)
    cam_struct.ref_T_cam_imu = _se3(scene.T_cam_imu)
    cam_struct.cur_T_cam_imu = _se3(scene.T_cam_imu)
    for k in range(3):
        cam_struct.ref_pos[k] = float(scene.ref_pos[k])
    cam_struct.n_features = int(scene.n_features)
    if device_ptrs is not None:
        cam_struct.mem_space = capi.SVOH_MEM_DEVICE
        cam_struct.px, cam_struct.f = device_ptrs["px"], device_ptrs["f"]
        cam_struct.pos_world, cam_struct.flags = device_ptrs["pos_world"], device_ptrs["flags"]
    else:
        arrs = [np.ascontiguousarray(scene.px, dtype=np.float64), np.ascontiguousarray(scene.f, dtype=np.float64),
                np.ascontiguousarray(scene.pos_world, dtype=np.float64),
                np.ascontiguousarray(scene.flags, dtype=np.uint8)]
        keep.extend(arrs)
        cam_struct.mem_space = capi.SVOH_MEM_HOST
        cam_struct.px, cam_struct.f, cam_struct.pos_world, cam_struct.flags = [a.ctypes.data for a in arrs]


def make_align_problems(items, T_init=None, prior=None, alpha_init=0.0, beta_init=0.0):
    """items: list of problems; each problem is a list (one per camera) of
    (scene, ref_frame_handle, cur_frame_handle[, device_ptrs]).
    Returns (ctypes array, keepalive list)."""
    n = len(items)
    arr = (capi.svoh_align_problem * n)()
    keep = []
    for i, cams in enumerate(items):
        pb = arr[i]
        pb.n_cams = len(cams)
        for c, it in enumerate(cams):
            sc, rf, cf = it[0], it[1], it[2]
            dp = it[3] if len(it) > 3 else None
            fill_align_camera(pb.cams[c], sc, rf, cf, keep, dp)
        sc0 = cams[0][0]
        Ti = T_init[i] if isinstance(T_init, (list, tuple)) else T_init
        pb.T_icur_iref = _se3(Ti if Ti is not None else sc0.T_icur_iref_init)
        pb.alpha_init, pb.beta_init = alpha_init, beta_init
        if prior is not None:
            pb.prior = prior[i] if isinstance(prior, (list, tuple)) else prior
    return arr, keep


# ---- KLT / matcher / depth filter ------------------------------------------------

def make_frame_view(frame_handle, cam, T_f_w, seed_mu_range=0.0, frame_id=0):
    v = capi.svoh_frame_view()
    v.frame = frame_handle
    v.cam = _camera(cam)
    v.T_f_w = _se3(T_f_w)
    v.seed_mu_range = float(seed_mu_range)
    v.id = int(frame_id)
    return v


def make_feature_batch(ref_frame_idx, px, f, grad, level, ftype):
    arrs = dict(ref_frame_idx=np.ascontiguousarray(ref_frame_idx, np.int32), px=np.ascontiguousarray(px, np.float64),
                f=np.ascontiguousarray(f, np.float64), grad=np.ascontiguousarray(grad, np.float64),
                level=np.ascontiguousarray(level, np.int32), type=np.ascontiguousarray(ftype, np.uint8).copy())
    fb = capi.svoh_feature_batch()
    fb.n = int(arrs["level"].size)
    for k, a in arrs.items():
        setattr(fb, k, a.ctypes.data)
    return fb, arrs


def _klt_track_batch(self, opt, ref_frames, cur_frame, px_ref, px_cur):
    n = len(px_ref) // 2
    if isinstance(ref_frames, int):
        ref_frames = [ref_frames] * n
    rf = (capi.svoh_frame_t * max(n, 1))(*ref_frames)
    px_ref = np.ascontiguousarray(px_ref, np.int32)
    out = np.ascontiguousarray(px_cur, np.float64).copy()
    status = np.zeros(max(n, 1), np.uint8)
    self._check(self.lib.svoh_klt_track_batch(self.h, C.byref(opt), n, rf, cur_frame, px_ref.ctypes.data,
                                              out.ctypes.data, status.ctypes.data))
    return out, status[:n]


def _match_direct_batch(self, mopt, ref_views, cur_view, fb, depth, px_cur, landmark_xyz=None):
    """landmark_xyz (n x 3, world): the pixelwise warp (Matcher::Options::use_affine_warp_ == false)."""
    n = fb.n
    rv = (capi.svoh_frame_view * len(ref_views))(*ref_views)
    depth = np.ascontiguousarray(depth, np.float64)
    out = dict(px_cur=np.ascontiguousarray(px_cur, np.float64).copy(), result=np.zeros(n, np.int32),
               f_cur=np.zeros(3 * n), search_level=np.zeros(n, np.int32), h_inv=np.zeros(n), A=np.zeros(4 * n))
    if landmark_xyz is not None:
        lm = np.ascontiguousarray(landmark_xyz, np.float64)
        assert lm.size == 3 * n
        self._check(self.lib.svoh_match_direct_batch_pixelwise(self.h, C.byref(mopt), len(ref_views), rv, C.byref(cur_view),
                                                               C.byref(fb), depth.ctypes.data, lm.ctypes.data,
                                                               out["px_cur"].ctypes.data, out["result"].ctypes.data,
                                                               out["f_cur"].ctypes.data, out["search_level"].ctypes.data,
                                                               out["h_inv"].ctypes.data, out["A"].ctypes.data))
        return out
    self._check(self.lib.svoh_match_direct_batch(self.h, C.byref(mopt), len(ref_views), rv, C.byref(cur_view),
                                                 C.byref(fb), depth.ctypes.data, out["px_cur"].ctypes.data,
                                                 out["result"].ctypes.data, out["f_cur"].ctypes.data,
                                                 out["search_level"].ctypes.data, out["h_inv"].ctypes.data,
                                                 out["A"].ctypes.data))
    return out


def _update_seeds_batch(self, mopt, dopt, ref_views, cur_view, fb, state):
    n = fb.n
    rv = (capi.svoh_frame_view * len(ref_views))(*ref_views)
    st = np.ascontiguousarray(state, np.float64).copy()
    success = np.zeros(max(n, 1), np.uint8)
    mr = np.zeros(max(n, 1), np.int32)
    ns = C.c_int32()
    cv, _ = _views(cur_view)          # one current frame, or a list (fb.cur_frame_idx picks per feature)
    self._check(self.lib.svoh_update_seeds_batch(self.h, C.byref(mopt), C.byref(dopt), len(ref_views), rv,
                                                 cv, C.byref(fb), st.ctypes.data, success.ctypes.data,
                                                 mr.ctypes.data, C.byref(ns)))
    return ns.value, st, success[:n], mr[:n]


def make_feature_batch_device(n, ref_frame_idx, px, f, grad, level, ftype, cur_frame_idx=0, n_cur_frames=0):
    """Feature batch whose arrays are DEVICE pointers (ints, e.g. torch.Tensor.data_ptr()); the caller keeps
    the owning tensors alive until the stream has run the call."""
    fb = capi.svoh_feature_batch()
    fb.n = int(n)
    fb.ref_frame_idx, fb.px, fb.f, fb.grad, fb.level, fb.type = ref_frame_idx, px, f, grad, level, ftype
    fb.cur_frame_idx = cur_frame_idx or None
    fb.n_cur_frames = int(n_cur_frames)
    fb.mem_space = capi.SVOH_MEM_DEVICE
    return fb


def _views(views):
    if isinstance(views, (list, tuple)):
        return (capi.svoh_frame_view * len(views))(*views), len(views)
    return C.byref(views), 1


def _klt_track_indexed(self, opt, frames, n_tracks, ref_idx, cur_idx, px_ref, px_cur, status,
                       mem_space=capi.SVOH_MEM_DEVICE):
    """svoh_klt_track_indexed with raw pointers (ints); device-resident by default, stream-ordered."""
    tab = (capi.svoh_frame_t * len(frames))(*frames)
    self._check(self.lib.svoh_klt_track_indexed(self.h, C.byref(opt), len(frames), tab, int(n_tracks), ref_idx, cur_idx,
                                                px_ref, px_cur, status, int(mem_space)))


def _update_seeds_device(self, mopt, dopt, ref_views, cur_views, fb, state, success, match_result=None,
                         want_count=False):
    """Device-resident svoh_update_seeds_batch: state / success / match_result are device pointers (ints)
    updated in place; returns the success count only if want_count (that synchronises)."""
    rv, n_ref = _views(ref_views)
    cv, _ = _views(cur_views)
    ns = C.c_int32()
    self._check(self.lib.svoh_update_seeds_batch(self.h, C.byref(mopt), C.byref(dopt), n_ref, rv, cv, C.byref(fb),
                                                 state, success, match_result, C.byref(ns) if want_count else None))
    return ns.value if want_count else None


def _match_direct_device(self, mopt, ref_views, cur_views, fb, depth, px_cur, result, f_cur=None, search_level=None,
                         h_inv=None, A_cur_ref=None, landmark_xyz=None):
    """landmark_xyz (device pointer, 3 x n, world): the pixelwise warp (svoh_match_direct_batch_pixelwise)."""
    rv, n_ref = _views(ref_views)
    cv, _ = _views(cur_views)
    if landmark_xyz is not None:
        self._check(self.lib.svoh_match_direct_batch_pixelwise(self.h, C.byref(mopt), n_ref, rv, cv, C.byref(fb), depth, landmark_xyz,
                                                               px_cur, result, f_cur, search_level, h_inv, A_cur_ref))
        return
    self._check(self.lib.svoh_match_direct_batch(self.h, C.byref(mopt), n_ref, rv, cv, C.byref(fb), depth, px_cur,
                                                 result, f_cur, search_level, h_inv, A_cur_ref))


def _epipolar_match_batch(self, mopt, ref_views, cur_views, fb, d_inv_common=None, d_inv=None, T_cur_ref=None):
    """svoh_epipolar_match_batch on host arrays: n x Matcher::findEpipolarMatchDirect (the stereo triangulation's call).
    T_cur_ref: list of 7-vectors (n_ref x n_cur, row-major) or None.  Returns a dict of per-feature arrays."""
    rv, n_ref = _views(ref_views)
    cv, _ = _views(cur_views)
    n = fb.n
    out = dict(result=np.zeros(max(n, 1), np.int32), depth=np.zeros(max(n, 1)), px_cur=np.zeros(2 * max(n, 1)),
               f_cur=np.zeros(3 * max(n, 1)), search_level=np.zeros(max(n, 1), np.int32), h_inv=np.zeros(max(n, 1)),
               A=np.zeros(4 * max(n, 1)))
    o = capi.svoh_epipolar_match_outputs()
    o.result, o.depth, o.px_cur, o.f_cur = (out[k].ctypes.data for k in ("result", "depth", "px_cur", "f_cur"))
    o.search_level, o.h_inv, o.A_cur_ref = (out[k].ctypes.data for k in ("search_level", "h_inv", "A"))
    dc = (C.c_double * 3)(*d_inv_common) if d_inv_common is not None else None
    di = None if d_inv is None else np.ascontiguousarray(d_inv, np.float64)
    T = None if T_cur_ref is None else (capi.svoh_se3 * len(T_cur_ref))(*[_se3(t) for t in T_cur_ref])
    self._check(self.lib.svoh_epipolar_match_batch(self.h, C.byref(mopt), n_ref, rv, cv, T, C.byref(fb), dc,
                                                   None if di is None else di.ctypes.data, C.byref(o)))
    return {k: v[:n * (v.size // max(n, 1))] for k, v in out.items()}


def make_pose_problem(cams, T_imu_world):
    """cams: list of dict(cam=synth.Camera, T_cam_imu=SE3, px, f, grad, level, type, xyz_world, usable); returns
    (svoh_pose_problem, keepalive) with per-feature outputs `outlier` / `final_error` in keepalive[i]."""
    pb = capi.svoh_pose_problem()
    pb.n_cams = len(cams)
    pb.T_imu_world = _se3(T_imu_world)
    keep = []
    for c, d in enumerate(cams):
        n = int(np.asarray(d["level"]).size)
        a = dict(px=np.ascontiguousarray(d["px"], np.float64), f=np.ascontiguousarray(d["f"], np.float64),
                 grad=np.ascontiguousarray(d["grad"], np.float64), level=np.ascontiguousarray(d["level"], np.int32),
                 type=np.ascontiguousarray(d["type"], np.uint8), xyz_world=np.ascontiguousarray(d["xyz_world"], np.float64),
                 usable=np.ascontiguousarray(d["usable"], np.uint8), outlier=np.zeros(max(n, 1), np.uint8),
                 final_error=np.zeros(max(n, 1), np.float64))
        pc = pb.cams[c]
        pc.cam = _camera(d["cam"])
        pc.T_cam_imu = _se3(d["T_cam_imu"])
        pc.n_features = n
        for k, v in a.items():
            setattr(pc, k, v.ctypes.data)
        keep.append(a)
    return pb, keep


def _optimize_pose(self, opt, problems):
    """svoh_optimize_pose_batch over a list of svoh_pose_problem; returns the list of svoh_pose_result."""
    n = len(problems)
    arr = (capi.svoh_pose_problem * n)(*problems)
    res = (capi.svoh_pose_result * n)()
    self._check(self.lib.svoh_optimize_pose_batch(self.h, C.byref(opt), n, arr, res))
    return list(res)


def _optimize_points(self, views, obs_begin, obs_view, obs_f, pos, n_iter=5, using_bearing_vector=False, side=False):
    """svoh_optimize_points_batch (Point::optimize for a batch of landmarks).  views: list of 7-vectors
    (q wxyz, t) T_f_w; obs_begin [n+1], obs_view [n_obs], obs_f [n_obs,3], pos [n,3].  Returns (pos, iters).
    side: through svoh_optimize_points_batch_enqueue + _collect (the queued form)."""
    T = (capi.svoh_se3 * max(1, len(views)))(*[_se3(v) for v in views])
    obs_begin = np.ascontiguousarray(obs_begin, dtype=np.int32)
    obs_view = np.ascontiguousarray(obs_view, dtype=np.int32)
    obs_f = np.ascontiguousarray(obs_f, dtype=np.float64)
    out = np.array(pos, dtype=np.float64, order="C", copy=True)
    n = out.shape[0]
    iters = np.zeros(max(1, n), np.int32)
    if side:
        self._check(self.lib.svoh_optimize_points_batch_enqueue(self.h, int(n_iter), int(bool(using_bearing_vector)), len(views), T, n, obs_begin.ctypes.data,
                                                                obs_view.ctypes.data, obs_f.ctypes.data, out.ctypes.data))
        self._check(self.lib.svoh_optimize_points_batch_collect(self.h, n, out.ctypes.data, iters.ctypes.data))
    else:
        self._check(self.lib.svoh_optimize_points_batch(self.h, int(n_iter), int(bool(using_bearing_vector)), len(views), T, n, obs_begin.ctypes.data,
                                                        obs_view.ctypes.data, obs_f.ctypes.data, out.ctypes.data, iters.ctypes.data))
    return out, iters[:n]


def _detect_features(self, opt, frame, width, height, occupancy=None, mask=None, max_n_features=None):
    """svoh_detect_features: dict(px [n,2], score, level, grad [n,2], type) like FastGradDetector::detect."""
    n_cells = int(np.ceil(width / opt.cell_size)) * int(np.ceil(height / opt.cell_size))
    if max_n_features is None:
        max_n_features = n_cells
    px = np.zeros(2 * n_cells); score = np.zeros(n_cells); level = np.zeros(n_cells, np.int32)
    grad = np.zeros(2 * n_cells); typ = np.zeros(n_cells, np.uint8)
    occ = None if occupancy is None else np.ascontiguousarray(occupancy, np.uint8)
    msk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    n = C.c_int32()
    self._check(self.lib.svoh_detect_features(self.h, frame, C.byref(opt), None if occ is None else occ.ctypes.data,
                                              None if msk is None else msk.ctypes.data,
                                              0 if msk is None else msk.strides[0], int(max_n_features), px.ctypes.data,
                                              score.ctypes.data, level.ctypes.data, grad.ctypes.data, typ.ctypes.data,
                                              C.byref(n)))
    n = n.value
    return dict(px=px[:2 * n].reshape(-1, 2).copy(), score=score[:n].copy(), level=level[:n].copy(),
                grad=grad[:2 * n].reshape(-1, 2).copy(), type=typ[:n].copy())


Context.epipolar_match_batch = _epipolar_match_batch
Context.detect_features = _detect_features
Context.optimize_pose = _optimize_pose
Context.optimize_points = _optimize_points
Context.klt_track_batch = _klt_track_batch
Context.klt_track_indexed = _klt_track_indexed
Context.update_seeds_device = _update_seeds_device
Context.match_direct_device = _match_direct_device
Context.match_direct_batch = _match_direct_batch
Context.update_seeds_batch = _update_seeds_batch
