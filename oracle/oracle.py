"""ctypes wrapper of liboracle.so -- TEST INFRASTRUCTURE ONLY (see svo_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  PARITY UNPINNED: see svo_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from svo_pro_universal_amd import _capi as capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


class orc_image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("pitch", C.c_int32), ("reserved", C.c_int32)]


class orc_pyramid(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("reserved", C.c_int32),
                ("level", orc_image * capi.SVOH_MAX_LEVELS)]


class orc_align_camera(C.Structure):
    _fields_ = [("ref_pyr", orc_pyramid), ("cur_pyr", orc_pyramid), ("cam", capi.svoh_camera),
                ("ref_T_imu_cam", capi.svoh_se3), ("ref_T_cam_imu", capi.svoh_se3),
                ("cur_T_cam_imu", capi.svoh_se3), ("ref_pos", C.c_double * 3),
                ("n_features", C.c_int32), ("reserved", C.c_int32),
                ("px", C.c_void_p), ("f", C.c_void_p), ("pos_world", C.c_void_p), ("flags", C.c_void_p)]


class orc_align_problem(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("reserved", C.c_int32),
                ("cams", orc_align_camera * capi.SVOH_MAX_CAMS), ("T_icur_iref", capi.svoh_se3),
                ("alpha_init", C.c_double), ("beta_init", C.c_double), ("prior", capi.svoh_align_prior)]


class orc_align_trace(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("count", C.c_int32), ("level", C.c_void_p),
                ("H", C.c_void_p), ("g", C.c_void_p), ("chi2", C.c_void_p),
                ("n_meas", C.c_void_p), ("state", C.c_void_p)]


def build(fast=False):
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle_fast.so" if fast else "liboracle.so"])


def load(fast=False):
    key = "fast" if fast else "strict"
    if key in _LIBS:
        return _LIBS[key]
    path = os.path.join(_HERE, "liboracle_fast.so" if fast else "liboracle.so")
    # SVO_ORACLE_LIB: another build of the same sources (tests/san/liboracle_asan.so: the oracle under AddressSanitizer,
    # with the sanitizer runtime preloaded into the interpreter by tests/test_sanitizers_cpu.py)
    if not fast and os.environ.get("SVO_ORACLE_LIB"):
        path = os.environ["SVO_ORACLE_LIB"]
    if not os.path.exists(path):
        build(fast)
    lib = C.CDLL(path)
    P = C.POINTER
    lib.orc_half_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    lib.orc_half_sample.restype = None
    lib.orc_create_img_pyramid.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           P(C.c_void_p)]
    lib.orc_create_img_pyramid.restype = None
    lib.orc_sparse_align_run.argtypes = [P(capi.svoh_align_options), P(orc_align_problem),
                                         P(capi.svoh_align_result), P(orc_align_trace)]
    lib.orc_sparse_align_evaluate.argtypes = [P(capi.svoh_align_options), P(orc_align_problem), C.c_int,
                                              C.c_void_p, C.c_void_p, P(C.c_double), P(C.c_int32),
                                              C.c_void_p, P(C.c_int32)]
    lib.orc_extract_features_subset.argtypes = [P(orc_align_camera), C.c_int, C.c_int, C.c_void_p]
    lib.orc_ldlt_solve.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    for nm in ("orc_quat_mul", "orc_quat_rotate", "orc_quat_exp", "orc_quat_log", "orc_quat_to_matrix"):
        getattr(lib, nm).restype = None
    lib.orc_se3_mul.argtypes = [P(capi.svoh_se3)] * 3
    lib.orc_se3_mul.restype = None
    lib.orc_se3_inverse.argtypes = [P(capi.svoh_se3)] * 2
    lib.orc_se3_inverse.restype = None
    lib.orc_se3_exp.argtypes = [C.c_void_p, P(capi.svoh_se3)]
    lib.orc_se3_exp.restype = None
    lib.orc_se3_log.argtypes = [P(capi.svoh_se3), C.c_void_p]
    lib.orc_se3_log.restype = None
    lib.orc_project3.argtypes = [P(capi.svoh_camera), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_project3.restype = None
    lib.orc_back_project3.argtypes = [P(capi.svoh_camera), C.c_void_p, C.c_void_p]
    lib.orc_back_project3.restype = None
    _LIBS[key] = lib
    return lib


# ---------------------------------------------------------------------------
# conversions
# ---------------------------------------------------------------------------

def to_se3(T):
    """synth.SE3 (or 7-vector q,t) -> svoh_se3"""
    s = capi.svoh_se3()
    v = T.as7() if hasattr(T, "as7") else np.asarray(T, dtype=np.float64)
    for i in range(4):
        s.q[i] = float(v[i])
    for i in range(3):
        s.t[i] = float(v[4 + i])
    return s


def from_se3(s):
    from svo_pro_universal_amd.synth import SE3
    return SE3([s.q[i] for i in range(4)], [s.t[i] for i in range(3)])


def to_camera(cam):
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


def create_img_pyramid(img0, n_levels, rounding=capi.SVOH_HALFSAMPLE_REFERENCE, fast=False):
    """frame_utils::createImgPyramid on the CPU; returns list of HxW u8 arrays."""
    lib = load(fast)
    img0 = np.ascontiguousarray(img0, dtype=np.uint8)
    h, w = img0.shape
    levels = [img0]
    for i in range(1, n_levels):
        levels.append(np.zeros((levels[-1].shape[0] // 2, levels[-1].shape[1] // 2), dtype=np.uint8))
    ptrs = (C.c_void_p * n_levels)(*[lv.ctypes.data for lv in levels])
    lib.orc_create_img_pyramid(img0.ctypes.data, w, h, w, n_levels, rounding, ptrs)
    return levels


def make_pyramid_struct(levels):
    p = orc_pyramid()
    p.n_levels = len(levels)
    for i, lv in enumerate(levels):
        assert lv.dtype == np.uint8 and lv.flags["C_CONTIGUOUS"]
        p.level[i].data = lv.ctypes.data
        p.level[i].width = lv.shape[1]
        p.level[i].height = lv.shape[0]
        p.level[i].pitch = lv.strides[0]
    return p


class AlignProblem(object):
    """Keeps numpy buffers alive next to the ctypes struct."""

    def __init__(self):
        self.c = orc_align_problem()
        self._keep = []


def problem_from_scenes(scenes_pyr, T_init=None, prior=None, alpha_init=0.0, beta_init=0.0):
    """scenes_pyr: list (one per camera) of (AlignScene, ref_levels, cur_levels)."""
    pb = AlignProblem()
    pb.c.n_cams = len(scenes_pyr)
    for i, (sc, ref_lv, cur_lv) in enumerate(scenes_pyr):
        cam = pb.c.cams[i]
        cam.ref_pyr = make_pyramid_struct(ref_lv)
        cam.cur_pyr = make_pyramid_struct(cur_lv)
        cam.cam = to_camera(sc.cam)
        cam.ref_T_imu_cam = to_se3(sc.T_imu_cam)
        cam.ref_T_cam_imu = to_se3(sc.T_cam_imu)
        cam.cur_T_cam_imu = to_se3(sc.T_cam_imu)
        for k in range(3):
            cam.ref_pos[k] = float(sc.ref_pos[k])
        cam.n_features = sc.n_features
        arrs = [np.ascontiguousarray(sc.px, dtype=np.float64), np.ascontiguousarray(sc.f, dtype=np.float64),
                np.ascontiguousarray(sc.pos_world, dtype=np.float64),
                np.ascontiguousarray(sc.flags, dtype=np.uint8)]
        cam.px, cam.f, cam.pos_world, cam.flags = [a.ctypes.data for a in arrs]
        pb._keep += arrs + list(ref_lv) + list(cur_lv)
    sc0 = scenes_pyr[0][0]
    pb.c.T_icur_iref = to_se3(T_init if T_init is not None else sc0.T_icur_iref_init)
    pb.c.alpha_init, pb.c.beta_init = alpha_init, beta_init
    if prior is not None:
        pb.c.prior = prior
    return pb


def sparse_align_run(opt, pb, trace_capacity=0, fast=False):
    lib = load(fast)
    res = capi.svoh_align_result()
    tr = None
    trp = None
    if trace_capacity > 0:
        tr = {
            "level": np.zeros(trace_capacity, np.int32), "H": np.zeros((trace_capacity, 64)),
            "g": np.zeros((trace_capacity, 8)), "chi2": np.zeros(trace_capacity),
            "n_meas": np.zeros(trace_capacity, np.int32), "state": np.zeros((trace_capacity, 9)),
        }
        t = orc_align_trace()
        t.capacity = trace_capacity
        for k in ("level", "H", "g", "chi2", "n_meas", "state"):
            setattr(t, k, tr[k].ctypes.data)
        trp = C.byref(t)
    n = lib.orc_sparse_align_run(C.byref(opt), C.byref(pb.c), C.byref(res), trp)
    if tr is not None:
        cnt = t.count
        tr = {k: v[:cnt] for k, v in tr.items()}
    return n, res, tr


def sparse_align_evaluate(opt, pb, level, fast=False):
    lib = load(fast)
    H = np.zeros(64)
    g = np.zeros(8)
    chi2 = C.c_double()
    nm = C.c_int32()
    nsel = C.c_int32()
    ntot = sum(pb.c.cams[i].n_features for i in range(pb.c.n_cams))
    vis = np.zeros(max(ntot, 1), np.uint8)
    lib.orc_sparse_align_evaluate(C.byref(opt), C.byref(pb.c), level, H.ctypes.data, g.ctypes.data,
                                  C.byref(chi2), C.byref(nm), vis.ctypes.data, C.byref(nsel))
    return H.reshape(8, 8).T.copy(), g, chi2.value, nm.value, vis[:nsel.value].copy()


# ---------------------------------------------------------------------------
# KLT / matcher / depth filter (svo_oracle_klt.c, svo_oracle_matcher.c)
# ---------------------------------------------------------------------------

class orc_frame_view(C.Structure):
    _fields_ = [("pyr", orc_pyramid), ("cam", capi.svoh_camera), ("T_f_w", capi.svoh_se3),
                ("seed_mu_range", C.c_double), ("id", C.c_int32), ("reserved", C.c_int32)]


def _bind_part2(lib):
    if getattr(lib, "_part2", False):
        return
    P = C.POINTER
    lib.orc_klt_track_batch.argtypes = [P(P(orc_pyramid)), P(orc_pyramid), C.c_int, C.c_int, C.c_void_p, C.c_int,
                                        C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_klt_track_batch.restype = None
    lib.orc_match_direct_batch.argtypes = [P(capi.svoh_matcher_options), C.c_int, P(orc_frame_view), P(orc_frame_view),
                                           P(capi.svoh_feature_batch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_match_direct_batch.restype = None
    lib.orc_match_direct_batch_ex.argtypes = [P(capi.svoh_matcher_options), C.c_int, P(orc_frame_view), P(orc_frame_view),
                                              P(capi.svoh_feature_batch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_match_direct_batch_ex.restype = None
    lib.orc_update_seeds_batch.argtypes = [P(capi.svoh_matcher_options), P(capi.svoh_depth_filter_options), C.c_int,
                                           P(orc_frame_view), P(orc_frame_view), P(capi.svoh_feature_batch),
                                           C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_zmssd_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_warp_affine.argtypes = [C.c_void_p, P(orc_image), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.orc_get_warp_matrix_affine.argtypes = [P(capi.svoh_camera), P(capi.svoh_camera), C.c_void_p, C.c_void_p,
                                               C.c_double, P(capi.svoh_se3), C.c_int, C.c_void_p]
    lib.orc_get_warp_matrix_affine.restype = None
    lib.orc_get_best_search_level.argtypes = [C.c_void_p, C.c_int]
    lib.orc_align_2d.argtypes = [P(orc_image), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.orc_align_1d.argtypes = [P(orc_image), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p]
    lib.orc_epipolar_match_batch.argtypes = [P(capi.svoh_matcher_options), C.c_int, P(orc_frame_view), P(orc_frame_view),
                                             P(capi.svoh_se3), P(capi.svoh_feature_batch), P(C.c_double), C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p]
    lib.orc_epipolar_match_batch.restype = None
    lib._part2 = True


def klt_track_batch(opt, ref_levels_list, cur_levels, px_ref, px_cur, fast=False):
    """ref_levels_list: one pyramid (list of arrays) per track, or a single pyramid for all."""
    lib = load(fast)
    _bind_part2(lib)
    n = len(px_ref) // 2
    if not isinstance(ref_levels_list[0], (list, tuple)):
        ref_levels_list = [ref_levels_list] * n
    uniq = {}
    for lv in ref_levels_list:
        uniq.setdefault(id(lv), make_pyramid_struct(lv))
    ptrs = (C.POINTER(orc_pyramid) * n)(*[C.pointer(uniq[id(lv)]) for lv in ref_levels_list])
    cur = make_pyramid_struct(cur_levels)
    px_ref = np.ascontiguousarray(px_ref, np.int32)
    out = np.ascontiguousarray(px_cur, np.float64).copy()
    status = np.zeros(n, np.uint8)
    ps = (C.c_int32 * capi.SVOH_MAX_LEVELS)(*list(opt.patch_sizes))
    lib.orc_klt_track_batch(ptrs, C.byref(cur), opt.max_level, opt.min_level, ps, opt.max_iter,
                            opt.min_update_squared, n, px_ref.ctypes.data, out.ctypes.data, status.ctypes.data)
    return out, status


def make_frame_view(levels, cam, T_f_w, seed_mu_range=0.0, frame_id=0):
    v = orc_frame_view()
    v.pyr = make_pyramid_struct(levels)
    v.cam = to_camera(cam)
    v.T_f_w = to_se3(T_f_w)
    v.seed_mu_range = seed_mu_range
    v.id = frame_id
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


def match_direct_batch(mopt, ref_views, cur_view, fb, depth, px_cur, fast=False, landmark_xyz=None):
    """landmark_xyz (n x 3, world): Matcher::Options::use_affine_warp_ == false (warpPixelwise)."""
    lib = load(fast)
    _bind_part2(lib)
    n = fb.n
    rv = (orc_frame_view * len(ref_views))(*ref_views)
    depth = np.ascontiguousarray(depth, np.float64)
    lm = None if landmark_xyz is None else np.ascontiguousarray(landmark_xyz, np.float64)
    out = dict(px_cur=np.ascontiguousarray(px_cur, np.float64).copy(), result=np.zeros(n, np.int32),
               f_cur=np.zeros(3 * n), search_level=np.zeros(n, np.int32), h_inv=np.zeros(n), A=np.zeros(4 * n))
    lib.orc_match_direct_batch_ex(C.byref(mopt), len(ref_views), rv, C.byref(cur_view), C.byref(fb), depth.ctypes.data,
                                  lm.ctypes.data if lm is not None else None,
                                  out["px_cur"].ctypes.data, out["result"].ctypes.data, out["f_cur"].ctypes.data,
                                  out["search_level"].ctypes.data, out["h_inv"].ctypes.data, out["A"].ctypes.data)
    return out


def warp_pixelwise(cur_view, ref_view, px_ref, landmark_xyz, level_ref, level_cur, halfpatch=5, fast=False):
    """warp::warpPixelwise (patch_warp.cpp:158-230): the (2 halfpatch)^2 patch, or None where the reference returns false."""
    lib = load(fast)
    lib.orc_warp_pixelwise.argtypes = [C.POINTER(orc_frame_view), C.POINTER(orc_frame_view), C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p]
    lib.orc_warp_pixelwise.restype = C.c_int
    px = np.ascontiguousarray(px_ref, np.float64); lm = np.ascontiguousarray(landmark_xyz, np.float64)
    out = np.zeros((2 * halfpatch, 2 * halfpatch), np.uint8)
    ok = lib.orc_warp_pixelwise(C.byref(cur_view), C.byref(ref_view), px.ctypes.data, lm.ctypes.data, level_ref, level_cur,
                                halfpatch, out.ctypes.data)
    return out if ok else None


def epipolar_match_batch(mopt, ref_views, cur_views, fb, d_inv_common=None, d_inv=None, T_cur_ref=None, fast=False):
    """n x Matcher::findEpipolarMatchDirect with align_1d = isEdgelet(type) (stereo_triangulation.cpp:92-104)."""
    lib = load(fast)
    _bind_part2(lib)
    n = fb.n
    rv = (orc_frame_view * len(ref_views))(*ref_views)
    if not isinstance(cur_views, (list, tuple)):
        cur_views = [cur_views]
    cv = (orc_frame_view * len(cur_views))(*cur_views)
    out = dict(result=np.zeros(n, np.int32), depth=np.zeros(n), px_cur=np.zeros(2 * n), f_cur=np.zeros(3 * n),
               search_level=np.zeros(n, np.int32), h_inv=np.zeros(n), A=np.zeros(4 * n))
    dc = (C.c_double * 3)(*d_inv_common) if d_inv_common is not None else None
    di = None if d_inv is None else np.ascontiguousarray(d_inv, np.float64)
    T = None
    if T_cur_ref is not None:
        T = (capi.svoh_se3 * len(T_cur_ref))()
        for k, t in enumerate(T_cur_ref):
            T[k] = to_se3(t)
    lib.orc_epipolar_match_batch(C.byref(mopt), len(ref_views), rv, cv, T, C.byref(fb), dc,
                                 None if di is None else di.ctypes.data, out["result"].ctypes.data, out["depth"].ctypes.data,
                                 out["px_cur"].ctypes.data, out["f_cur"].ctypes.data, out["search_level"].ctypes.data,
                                 out["h_inv"].ctypes.data, out["A"].ctypes.data)
    return out



class orc_stereo_match(C.Structure):
    _fields_ = [("i_ref", C.c_int32), ("pad", C.c_int32), ("xyz_cam0", C.c_double * 3), ("px", C.c_double * 2),
                ("f", C.c_double * 3), ("grad", C.c_double * 2), ("depth", C.c_double)]


def stereo_triangulate(frame0, frame1, T_f1f0, fb, indices, n_desired, d_inv, fast=False):
    """The loop of StereoTriangulation::compute (orc_stereo_triangulate, stereo_triangulation.cpp:92-137) over the new
    features in the given (already shuffled) order.  Returns (matches: list of dicts, result per feature of the batch with -1 for the ones never visited, n_failed)."""
    lib = load(fast)
    _bind_part2(lib)
    P = C.POINTER
    lib.orc_stereo_triangulate.argtypes = [P(orc_frame_view), P(orc_frame_view), P(capi.svoh_se3), P(capi.svoh_feature_batch),
                                           C.c_int, C.c_void_p, C.c_int, P(C.c_double), P(orc_stereo_match), C.c_void_p,
                                           P(C.c_int)]
    lib.orc_stereo_triangulate.restype = C.c_int
    idx = np.ascontiguousarray(indices, np.int32)
    out = (orc_stereo_match * max(1, int(n_desired)))()
    result = np.full(int(fb.n), -1, np.int32)   # per feature of the batch; -1: not visited
    n_failed = C.c_int(0)
    T = to_se3(T_f1f0)
    di = (C.c_double * 3)(*d_inv)
    n = lib.orc_stereo_triangulate(C.byref(frame0), C.byref(frame1), C.byref(T), C.byref(fb), int(idx.size), idx.ctypes.data,
                                   int(n_desired), di, out, result.ctypes.data, C.byref(n_failed))
    matches = [dict(i_ref=int(m.i_ref), xyz_cam0=np.array(m.xyz_cam0[:]), px=np.array(m.px[:]), f=np.array(m.f[:]),
                    grad=np.array(m.grad[:]), depth=float(m.depth)) for m in out[:n]]
    return matches, result, int(n_failed.value)


def update_seeds_batch(mopt, dopt, ref_views, cur_view, fb, state, fast=False):
    lib = load(fast)
    _bind_part2(lib)
    n = fb.n
    rv = (orc_frame_view * len(ref_views))(*ref_views)
    st = np.ascontiguousarray(state, np.float64).copy()
    success = np.zeros(n, np.uint8)
    mr = np.zeros(n, np.int32)
    ns = lib.orc_update_seeds_batch(C.byref(mopt), C.byref(dopt), len(ref_views), rv, C.byref(cur_view), C.byref(fb),
                                    st.ctypes.data, success.ctypes.data, mr.ctypes.data)
    return ns, st, success, mr


# ---- keyframe feature detector ----------------------------------

def _bind_detector(lib):
    if getattr(lib, "_det", False):
        return
    P = C.POINTER
    lib.orc_fast_corner_detect_10.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    lib.orc_fast_corner_score_10.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.orc_fast_corner_score_10.restype = None
    lib.orc_fast_nonmax_3x3.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.orc_gaussian_blur_3x3.argtypes = [P(orc_image), C.c_void_p]
    lib.orc_gaussian_blur_3x3.restype = None
    lib.orc_scharr_16s.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.orc_scharr_16s.restype = None
    lib.orc_angle_at_pixel_using_histogram.argtypes = [P(orc_image), C.c_int, C.c_int, C.c_int]
    lib.orc_angle_at_pixel_using_histogram.restype = C.c_double
    lib.orc_detect_features.argtypes = [P(orc_pyramid), P(capi.svoh_detector_options), C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib._det = True


def _image_struct(img):
    assert img.dtype == np.uint8 and img.flags["C_CONTIGUOUS"]
    s = orc_image()
    s.data, s.width, s.height, s.pitch = img.ctypes.data, img.shape[1], img.shape[0], img.strides[0]
    return s


def fast_corners(img, barrier, fast=False):
    """(xy [n,2] in raster order, score [n], indices surviving the 3x3 non-maximum suppression)."""
    lib = load(fast); _bind_detector(lib)
    cap = img.size
    xy = np.zeros(2 * cap, np.int32)
    n = lib.orc_fast_corner_detect_10(img.ctypes.data, img.shape[1], img.shape[0], img.strides[0], int(barrier), xy.ctypes.data, cap)
    xy = xy[:2 * n].copy()
    sc = np.zeros(max(n, 1), np.int32)
    lib.orc_fast_corner_score_10(img.ctypes.data, img.strides[0], xy.ctypes.data, n, int(barrier), sc.ctypes.data)
    nm = np.zeros(max(n, 1), np.int32)
    k = lib.orc_fast_nonmax_3x3(xy.ctypes.data, sc.ctypes.data, n, nm.ctypes.data)
    return xy.reshape(-1, 2), sc[:n], nm[:k]


# ---- oracle/_ref: the reference's OWN FAST code, compiled from /root/reference (oracle/ref_fast/Makefile) --------

REF_FAST_LIB = os.path.join(_HERE, "_ref", "libfast_ref.so")


def ref_fast_available(build_if_possible=True):
    """True when oracle/_ref/libfast_ref.so exists (it is built where /root/reference exists and travels prebuilt)."""
    if not os.path.exists(REF_FAST_LIB) and build_if_possible and os.path.isdir("/root/reference/src/fast_neon"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "ref_fast")])
    return os.path.exists(REF_FAST_LIB)


def _padded(img):
    # the SSE2 detector loads 16 bytes at p + 2 + 2 * stride of the last row it visits: keep the image inside a
    # larger allocation (16-byte aligned start, like a cv::Mat), same width / height / pitch for the callee; a view with
    # a pitch larger than its width (a cv::Mat ROI) keeps that pitch
    assert img.dtype == np.uint8 and img.ndim == 2 and (img.shape[1] <= 1 or img.strides[1] == 1)
    h, w = img.shape
    pitch = max(img.strides[0], w) if h > 1 else w
    buf = np.zeros(h * pitch + 128, np.uint8)
    off = (-buf.ctypes.data) % 16
    view = np.lib.stride_tricks.as_strided(buf[off:], shape=(h, w), strides=(pitch, 1))
    view[...] = img
    return view, buf


def ref_fast_corners(img, barrier):
    """fast_corner_detect_10_sse2 + fast_corner_score_10 + fast_nonmax_3x3 of the REFERENCE (same tuple as fast_corners)."""
    if "ref_fast" not in _LIBS:
        assert ref_fast_available(), "oracle/_ref/libfast_ref.so is missing and /root/reference is not here to build it"
        lib = C.CDLL(REF_FAST_LIB)
        lib.fast_ref_detect_score_nonmax.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_int, C.c_void_p]
        lib.fast_ref_detect_plain.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        _LIBS["ref_fast"] = lib
    lib = _LIBS["ref_fast"]
    view, keep = _padded(img)
    cap = max(view.size, 1)
    xy = np.zeros(2 * cap, np.int32); sc = np.zeros(cap, np.int32); nm = np.zeros(cap, np.int32)
    k = C.c_int32(0)
    n = lib.fast_ref_detect_score_nonmax(view.ctypes.data, view.shape[1], view.shape[0], view.strides[0], int(barrier),
                                         xy.ctypes.data, sc.ctypes.data, nm.ctypes.data, cap, C.byref(k))
    assert n <= cap
    del keep
    return xy[:2 * n].reshape(-1, 2).copy(), sc[:n].copy(), nm[:k.value].copy()


def ref_fast_corners_plain(img, barrier):
    """fast_corner_detect_10 (the plain decision tree) of the REFERENCE: xy [n,2]."""
    ref_fast_corners(np.zeros((8, 8), np.uint8), 10)
    lib = _LIBS["ref_fast"]
    view, keep = _padded(img)
    cap = max(view.size, 1)
    xy = np.zeros(2 * cap, np.int32)
    n = lib.fast_ref_detect_plain(view.ctypes.data, view.shape[1], view.shape[0], view.strides[0], int(barrier), xy.ctypes.data, cap)
    del keep
    return xy[:2 * n].reshape(-1, 2).copy()


def gaussian_blur_3x3(img, fast=False):
    lib = load(fast); _bind_detector(lib)
    out = np.zeros(img.shape, np.uint8)
    s = _image_struct(img)
    lib.orc_gaussian_blur_3x3(C.byref(s), out.ctypes.data)
    return out


def scharr_16s(img, x_derivative, fast=False):
    lib = load(fast); _bind_detector(lib)
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros(img.shape, np.int16)
    lib.orc_scharr_16s(img.ctypes.data, img.shape[1], img.shape[0], int(bool(x_derivative)), out.ctypes.data)
    return out


def angle_at_pixel(img, x, y, halfpatch=4, fast=False):
    lib = load(fast); _bind_detector(lib)
    s = _image_struct(img)
    return lib.orc_angle_at_pixel_using_histogram(C.byref(s), int(x), int(y), int(halfpatch))


def detect_features(opt, levels, occupancy=None, mask=None, max_n_features=None, fast=False):
    """FastDetector / FastGradDetector::detect restatement: dict(px [n,2], score, level, grad [n,2], type)."""
    lib = load(fast); _bind_detector(lib)
    pyr = make_pyramid_struct(levels)
    h, w = levels[0].shape
    n_cells = int(np.ceil(w / opt.cell_size)) * int(np.ceil(h / opt.cell_size))
    if max_n_features is None:
        max_n_features = n_cells
    px = np.zeros(2 * n_cells); score = np.zeros(n_cells); level = np.zeros(n_cells, np.int32)
    grad = np.zeros(2 * n_cells); typ = np.zeros(n_cells, np.uint8)
    occ = None if occupancy is None else np.ascontiguousarray(occupancy, np.uint8)
    msk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    n = lib.orc_detect_features(C.byref(pyr), C.byref(opt), None if occ is None else occ.ctypes.data,
                                None if msk is None else msk.ctypes.data, 0 if msk is None else msk.strides[0],
                                int(max_n_features), px.ctypes.data, score.ctypes.data, level.ctypes.data,
                                grad.ctypes.data, typ.ctypes.data)
    return dict(px=px[:2 * n].reshape(-1, 2).copy(), score=score[:n].copy(), level=level[:n].copy(),
                grad=grad[:2 * n].reshape(-1, 2).copy(), type=typ[:n].copy())


# ---- pose optimiser ----------------------------------------------

def _bind_pose(lib):
    if getattr(lib, "_pose", False):
        return
    P = C.POINTER
    lib.orc_jacobian_xyz2uv_imu.argtypes = [P(capi.svoh_se3), C.c_void_p, C.c_void_p]
    lib.orc_jacobian_xyz2uv_imu.restype = None
    lib.orc_jacobian_xyz2img_imu.argtypes = [P(capi.svoh_se3), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_jacobian_xyz2img_imu.restype = None
    lib.orc_jacobian_xyz2f_imu.argtypes = [P(capi.svoh_se3), C.c_void_p, C.c_void_p]
    lib.orc_jacobian_xyz2f_imu.restype = None
    lib.orc_optimize_pose.argtypes = [P(capi.svoh_pose_options), P(capi.svoh_pose_problem), P(capi.svoh_pose_result)]
    lib.orc_optimize_pose.restype = None
    lib.orc_project3.argtypes = [P(capi.svoh_camera), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_project3.restype = None
    lib._pose = True


def pose_jacobians(T_cam_imu, p_in_imu, cam=None, fast=False):
    """(J_uv 2x6, J_f 3x6, J_img 2x6 or None) of frame.h:342-397 at p_in_imu."""
    lib = load(fast); _bind_pose(lib)
    T = to_se3(T_cam_imu)
    p = np.ascontiguousarray(p_in_imu, np.float64)
    Juv, Jf = np.zeros(12), np.zeros(18)
    lib.orc_jacobian_xyz2uv_imu(C.byref(T), p.ctypes.data, Juv.ctypes.data)
    lib.orc_jacobian_xyz2f_imu(C.byref(T), p.ctypes.data, Jf.ctypes.data)
    Jimg = None
    if cam is not None:
        pc = np.ascontiguousarray(T_cam_imu.transform(p), np.float64)
        uv, Jc, Jimg = np.zeros(2), np.zeros(6), np.zeros(12)
        c = to_camera(cam)
        lib.orc_project3(C.byref(c), pc.ctypes.data, uv.ctypes.data, Jc.ctypes.data)
        lib.orc_jacobian_xyz2img_imu(C.byref(T), p.ctypes.data, Jc.ctypes.data, Jimg.ctypes.data)
        Jimg = Jimg.reshape(2, 6)
    return Juv.reshape(2, 6), Jf.reshape(3, 6), Jimg


def optimize_pose(opt, problem, fast=False):
    lib = load(fast); _bind_pose(lib)
    res = capi.svoh_pose_result()
    lib.orc_optimize_pose(C.byref(opt), C.byref(problem), C.byref(res))
    return res


def optimize_points(views, obs_begin, obs_view, obs_f, pos, n_iter=5, using_bearing_vector=False, fast=False):
    """Point::optimize (orc_optimize_point) for every landmark of a batch; same arguments as
    frontend.Context.optimize_points.  Returns (pos, iters)."""
    lib = load(fast)
    P = C.POINTER
    lib.orc_optimize_point.argtypes = [C.c_int, C.c_int, C.c_int, P(P(capi.svoh_se3)), C.c_void_p, C.c_void_p]
    lib.orc_optimize_point.restype = C.c_int
    T = [to_se3(v) for v in views]
    obs_f = np.ascontiguousarray(obs_f, dtype=np.float64).reshape(-1, 3)
    out = np.array(pos, dtype=np.float64, order="C", copy=True)
    iters = np.zeros(out.shape[0], np.int32)
    for i in range(out.shape[0]):
        o0, o1 = int(obs_begin[i]), int(obs_begin[i + 1])
        ptrs = (P(capi.svoh_se3) * max(1, o1 - o0))(*[C.pointer(T[int(obs_view[o])]) for o in range(o0, o1)])
        f = np.ascontiguousarray(obs_f[o0:o1])
        p = out[i].copy()
        iters[i] = lib.orc_optimize_point(int(n_iter), int(bool(using_bearing_vector)), o1 - o0, ptrs, f.ctypes.data,
                                          p.ctypes.data)
        out[i] = p
    return out, iters
