"""ctypes mirror of include/svo_hip.h and loader of libsvo_hip.so.

The library is the product; there is no CPU fallback.  Importing this module
never touches the GPU; `load()` raises if the HIP extension has not been built
(run `python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

SVOH_ABI_VERSION = 2   # include/svo_hip.h
SVOH_BATCH_UNITS, SVOH_BATCH_WHOLE_SETS = 0, 1   # svoh_feature_batch.layout
SVOH_MAX_LEVELS = 8
SVOH_MAX_CAMS = 4

SVOH_OK = 0
SVOH_MEM_HOST = 0
SVOH_MEM_DEVICE = 1
SVOH_MEM_STAGED = 2
SVOH_STAGE_MATCH_OUTPUTS = 1
SVOH_STAGE_RESIDENT_COLUMNS = 2
SVOH_MEM_HOST_PINNED = 3

SVOH_DISTORTION_NONE = 0
SVOH_DISTORTION_RADTAN = 1

SVOH_HALFSAMPLE_REFERENCE = 0
SVOH_HALFSAMPLE_SCALAR = 1
SVOH_HALFSAMPLE_SSE2 = 2

svoh_frame_t = C.c_uint64


class svoh_se3(C.Structure):
    _fields_ = [("q", C.c_double * 4), ("t", C.c_double * 3)]


class svoh_camera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("d", C.c_double * 4), ("distortion", C.c_int32), ("width", C.c_int32),
                ("height", C.c_int32), ("reserved", C.c_int32)]


class svoh_align_options(C.Structure):
    _fields_ = [("max_level", C.c_int32), ("min_level", C.c_int32), ("patch_size", C.c_int32),
                ("max_iter", C.c_int32), ("eps", C.c_double),
                ("estimate_illumination_gain", C.c_int32), ("estimate_illumination_offset", C.c_int32),
                ("use_distortion_jacobian", C.c_int32), ("robustification", C.c_int32),
                ("weight_scale", C.c_double)]


class svoh_align_prior(C.Structure):
    _fields_ = [("have_prior", C.c_int32), ("reserved", C.c_int32), ("T_prior", svoh_se3),
                ("alpha_prior", C.c_double), ("beta_prior", C.c_double),
                ("lambda_rot", C.c_double), ("lambda_trans", C.c_double),
                ("lambda_alpha", C.c_double), ("lambda_beta", C.c_double)]


class svoh_align_camera(C.Structure):
    _fields_ = [("ref_frame", svoh_frame_t), ("cur_frame", svoh_frame_t), ("cam", svoh_camera),
                ("ref_T_imu_cam", svoh_se3), ("ref_T_cam_imu", svoh_se3), ("cur_T_cam_imu", svoh_se3),
                ("ref_pos", C.c_double * 3), ("n_features", C.c_int32), ("mem_space", C.c_int32),
                ("px", C.c_void_p), ("f", C.c_void_p), ("pos_world", C.c_void_p), ("flags", C.c_void_p), ("pos_seed_unit", C.c_void_p)]


class svoh_align_problem(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("reserved", C.c_int32),
                ("cams", svoh_align_camera * SVOH_MAX_CAMS), ("T_icur_iref", svoh_se3),
                ("alpha_init", C.c_double), ("beta_init", C.c_double), ("prior", svoh_align_prior)]


class svoh_align_result(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_fts_to_track", C.c_int32), ("T_icur_iref", svoh_se3),
                ("alpha", C.c_double), ("beta", C.c_double),
                ("iters", C.c_int32 * SVOH_MAX_LEVELS), ("n_meas", C.c_int32 * SVOH_MAX_LEVELS),
                ("chi2", C.c_double * SVOH_MAX_LEVELS), ("n_patch_iters", C.c_int64)]


class svoh_align_gn_state(C.Structure):
    _fields_ = [("T_icur_iref", svoh_se3), ("alpha", C.c_double), ("beta", C.c_double),
                ("T_old", svoh_se3), ("alpha_old", C.c_double), ("beta_old", C.c_double),
                ("I_prior", C.c_double * 8), ("chi2", C.c_double), ("n_meas", C.c_int32), ("stop", C.c_int32),
                ("level_done", C.c_int32), ("status", C.c_int32)]


SVOH_ALIGN_SUMS_DOUBLES = 74


class svoh_klt_options(C.Structure):
    _fields_ = [("max_level", C.c_int32), ("min_level", C.c_int32), ("patch_sizes", C.c_int32 * SVOH_MAX_LEVELS),
                ("max_iter", C.c_int32), ("min_update_squared", C.c_float), ("reserved", C.c_int32)]


class svoh_matcher_options(C.Structure):
    _fields_ = [("align_max_iter", C.c_int32), ("max_epi_search_steps", C.c_int32), ("subpix_refinement", C.c_int32),
                ("epi_search_edgelet_filtering", C.c_int32), ("scan_on_unit_sphere", C.c_int32),
                ("affine_est_offset", C.c_int32), ("affine_est_gain", C.c_int32), ("reserved", C.c_int32),
                ("epi_search_edgelet_max_angle", C.c_double), ("max_patch_diff_ratio", C.c_double)]


class svoh_frame_view(C.Structure):
    _fields_ = [("frame", svoh_frame_t), ("cam", svoh_camera), ("T_f_w", svoh_se3), ("seed_mu_range", C.c_double),
                ("id", C.c_int32), ("pose_result_index_plus1", C.c_int32), ("features", C.c_uint64)]


class svoh_feature_batch(C.Structure):
    _fields_ = [("n", C.c_int32), ("layout", C.c_int32), ("ref_frame_idx", C.c_void_p), ("px", C.c_void_p),
                ("f", C.c_void_p), ("grad", C.c_void_p), ("level", C.c_void_p), ("type", C.c_void_p),
                ("cur_frame_idx", C.c_void_p), ("n_cur_frames", C.c_int32), ("mem_space", C.c_int32), ("feature_index", C.c_void_p)]


class svoh_matcher_stage_t(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("ref_frame_idx", "cur_frame_idx", "px", "f", "grad", "level", "type", "depth", "px_cur", "state",
                                            "result", "success", "f_cur", "search_level", "h_inv", "A_cur_ref", "feature_index")]


class svoh_candidate_job(C.Structure):
    _fields_ = [("cam", svoh_camera), ("T_f_w_or_T_cam_imu", svoh_se3), ("T_imu_world_ref", svoh_se3), ("align_result_index", C.c_int32),
                ("kf_begin", C.c_int32), ("n_kf", C.c_int32), ("point_begin", C.c_int32), ("n_points", C.c_int32), ("reserved", C.c_int32)]


class svoh_candidate_stage_t(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("jobs", "T_world_kf", "job", "kind", "kf", "v", "mu", "px", "visible", "ranges", "mu_unit")]


class svoh_candidate_range(C.Structure):
    _fields_ = [("features", C.c_uint64), ("point_begin", C.c_int32), ("n_points", C.c_int32), ("job", C.c_int32), ("reserved", C.c_int32)]


class svoh_depth_filter_options(C.Structure):
    _fields_ = [("seed_convergence_sigma2_thresh", C.c_double), ("mappoint_convergence_sigma2_thresh", C.c_double),
                ("px_error_angle", C.c_double), ("check_visibility", C.c_int32), ("check_convergence", C.c_int32),
                ("use_vogiatzis_update", C.c_int32), ("reserved", C.c_int32)]


# svo::FeatureType (types.h:60-73)
class svoh_epipolar_match_outputs(C.Structure):
    _fields_ = [("result", C.c_void_p), ("depth", C.c_void_p), ("px_cur", C.c_void_p), ("f_cur", C.c_void_p),
                ("search_level", C.c_void_p), ("h_inv", C.c_void_p), ("A_cur_ref", C.c_void_p)]


class svoh_seed_match_outputs(C.Structure):
    _fields_ = [("px_cur", C.c_void_p), ("f_cur", C.c_void_p), ("search_level", C.c_void_p), ("A_cur_ref", C.c_void_p)]


POSE_ERR_UNIT_PLANE, POSE_ERR_BEARING_DIFF, POSE_ERR_IMAGE_PLANE = 0, 1, 2


class svoh_pose_options(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("error_type", C.c_int32), ("eps", C.c_double),
                ("outlier_threshold", C.c_double), ("have_rotation_prior", C.c_int32), ("reserved", C.c_int32),
                ("prior_lambda", C.c_double), ("R_prior", C.c_double * 4)]


class svoh_pose_camera(C.Structure):
    _fields_ = [("cam", svoh_camera), ("T_cam_imu", svoh_se3), ("n_features", C.c_int32), ("reserved", C.c_int32),
                ("px", C.c_void_p), ("f", C.c_void_p), ("grad", C.c_void_p), ("level", C.c_void_p), ("type", C.c_void_p),
                ("xyz_world", C.c_void_p), ("usable", C.c_void_p), ("outlier", C.c_void_p), ("final_error", C.c_void_p)]


class svoh_pose_problem(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("reserved", C.c_int32), ("cams", svoh_pose_camera * SVOH_MAX_CAMS),
                ("T_imu_world", svoh_se3)]


class svoh_pose_packed_arrays(C.Structure):
    _fields_ = [("n_features_total", C.c_int64), ("px", C.c_void_p), ("f", C.c_void_p), ("grad", C.c_void_p),
                ("level", C.c_void_p), ("type", C.c_void_p), ("xyz_world", C.c_void_p), ("usable", C.c_void_p),
                ("outlier", C.c_void_p), ("final_error", C.c_void_p)]


class svoh_pose_result(C.Structure):
    _fields_ = [("T_imu_world", svoh_se3), ("measurement_sigma", C.c_double), ("reproj_error_before", C.c_double),
                ("reproj_error_after", C.c_double), ("n_meas", C.c_int32), ("n_deleted_edges", C.c_int32),
                ("n_deleted_corners", C.c_int32), ("iters", C.c_int32), ("status", C.c_int32), ("reserved", C.c_int32)]


def default_pose_options(cam=None, reproj_thresh_px=2.0, **kw):
    """PoseOptimizer::getDefaultSolverOptions + the outlier threshold of removeOutliers for `cam`
    (pose_optimizer.cpp:22-29, 211-218; poseoptim_thresh = 2.0, svo_factory.cpp:147)."""
    import math
    o = svoh_pose_options(max_iter=10, error_type=POSE_ERR_UNIT_PLANE, eps=1e-6, outlier_threshold=0.0,
                          have_rotation_prior=0, prior_lambda=0.0)
    o.R_prior[0] = 1.0
    for k, v in kw.items():
        if k == "R_prior":
            for i in range(4):
                o.R_prior[i] = float(v[i])
        elif not hasattr(o, k):
            raise AttributeError(k)
        else:
            setattr(o, k, v)
    if cam is not None and "outlier_threshold" not in kw:
        if o.error_type == POSE_ERR_UNIT_PLANE:
            o.outlier_threshold = reproj_thresh_px / abs(cam.fx)
        elif o.error_type == POSE_ERR_BEARING_DIFF:
            ang = math.atan(reproj_thresh_px / (2.0 * cam.fx)) + math.atan(reproj_thresh_px / (2.0 * cam.fy))
            o.outlier_threshold = abs(2 * math.sin(0.5 * ang))
        else:
            o.outlier_threshold = reproj_thresh_px
    return o


class svoh_detector_options(C.Structure):
    _fields_ = [("cell_size", C.c_int32), ("max_level", C.c_int32), ("min_level", C.c_int32), ("border", C.c_int32),
                ("detect_edgelets", C.c_int32), ("reserved", C.c_int32),
                ("threshold_primary", C.c_double), ("threshold_secondary", C.c_double)]


def default_detector_options(**kw):
    """DetectorOptions defaults (feature_detection_types.h:49-84) with the FAST_GRAD detector of pinhole.yaml."""
    o = svoh_detector_options(cell_size=30, max_level=2, min_level=0, border=8, detect_edgelets=1,
                              threshold_primary=10.0, threshold_secondary=100.0)
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


FT_EDGELET_SEED, FT_CORNER_SEED, FT_MAPPOINT_SEED = 0, 1, 2
FT_EDGELET_SEED_CONVERGED, FT_CORNER_SEED_CONVERGED, FT_MAPPOINT_SEED_CONVERGED = 3, 4, 5
FT_EDGELET, FT_CORNER, FT_MAPPOINT, FT_FIXED_LANDMARK, FT_OUTLIER = 6, 7, 8, 9, 10
MATCH_SUCCESS = 0
MATCH_NOT_RUN = 100


def default_klt_options(**kw):
    """FeatureTrackerOptions klt_* defaults (feature_tracking_types.h:15-29)."""
    o = svoh_klt_options(max_level=4, min_level=0, max_iter=30, min_update_squared=0.001)
    for i, v in enumerate([16, 16, 16, 8, 8, 8, 8, 8]):
        o.patch_sizes[i] = v
    for k, v in kw.items():
        if k == "patch_sizes":
            for i, x in enumerate(v):
                o.patch_sizes[i] = x
        else:
            if not hasattr(o, k):
                raise AttributeError(k)
            setattr(o, k, v)
    return o


def default_matcher_options(**kw):
    """Matcher::Options defaults (matcher.h:39-54) with the depth filter's unit-plane scan
    (svo_factory.cpp:261)."""
    o = svoh_matcher_options(align_max_iter=10, max_epi_search_steps=100, subpix_refinement=1,
                             epi_search_edgelet_filtering=1, scan_on_unit_sphere=0, affine_est_offset=1,
                             affine_est_gain=0, epi_search_edgelet_max_angle=0.7, max_patch_diff_ratio=2.0)
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def default_depth_filter_options(cam=None, **kw):
    import math
    o = svoh_depth_filter_options(seed_convergence_sigma2_thresh=200.0, mappoint_convergence_sigma2_thresh=500.0,
                                  px_error_angle=0.0, check_visibility=1, check_convergence=0, use_vogiatzis_update=1)
    if cam is not None:  # PinholeProjection::getAngleError(1.0) (pinhole_projection.hpp:72-76)
        o.px_error_angle = math.atan(1.0 / (2.0 * cam.fx)) + math.atan(1.0 / (2.0 * cam.fy))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def default_align_options(**kw):
    """SparseImgAlignOptions defaults (sparse_img_align_base.h:37-46) + solver
    defaults (sparse_img_align_base.cpp:35-42)."""
    o = svoh_align_options(max_level=4, min_level=1, patch_size=4, max_iter=10, eps=0.0005,
                           estimate_illumination_gain=0, estimate_illumination_offset=0,
                           use_distortion_jacobian=0, robustification=0, weight_scale=10.0)
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


_LIB = None
# SVOH_LIB: diagnostic builds only (e.g. the phase-stamp build used while profiling)
LIB_PATH = os.environ.get("SVOH_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc",
                                                     "libsvo_hip.so")

# every symbol include/svo_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "svoh_abi_version", "svoh_create", "svoh_destroy", "svoh_last_error_string",
    "svoh_synchronize", "svoh_stream",
    "svoh_upload_pyramid", "svoh_build_pyramid", "svoh_build_pyramid_batch",
    "svoh_download_level", "svoh_frame_info", "svoh_release_frame", "svoh_camera_maths", "svoh_context_stats", "svoh_reload_knobs", "svoh_set_kernel_timing", "svoh_set_copy_policy", "svoh_set_align_geometry_classes",
    "svoh_sparse_align_batch", "svoh_sparse_align_enqueue", "svoh_sparse_align_fetch", "svoh_sparse_align_fetch_all",
    "svoh_sparse_align_evaluate", "svoh_sparse_align_last_kernel_ms", "svoh_sparse_align_kernel_ms_history",
    "svoh_sparse_align_split_buffers", "svoh_sparse_align_split_init", "svoh_sparse_align_partial_sums", "svoh_sparse_align_gn_update",
    "svoh_klt_track_batch", "svoh_klt_track_multi", "svoh_klt_track_indexed", "svoh_last_kernel_ms", "svoh_last_kernel_counters",
    "svoh_match_direct_batch", "svoh_match_direct_batch_pixelwise", "svoh_matcher_begin_deferred", "svoh_matcher_collect", "svoh_matcher_flush", "svoh_matcher_deferred_set_cur_frame", "svoh_optimize_pose_batch_hook",
    "svoh_update_seeds_batch", "svoh_update_seeds_batch_ex", "svoh_epipolar_match_batch",
    "svoh_project_candidates_enqueue", "svoh_project_candidates_collect", "svoh_project_candidates",
    "svoh_detect_features", "svoh_optimize_pose_batch", "svoh_optimize_pose_batch_packed", "svoh_optimize_points_batch", "svoh_optimize_points_batch_enqueue", "svoh_optimize_points_batch_collect",
    # round 5: what the lock-step front end of many camera streams stages in place and launches once per stage
    "svoh_host_alloc", "svoh_host_free", "svoh_build_pyramid_multi", "svoh_build_pyramid_multi_prefetch", "svoh_prefetch_fence",
    "svoh_sparse_align_geometry_key", "svoh_sparse_align_enqueue_keyed",
    "svoh_project_candidates_stage", "svoh_project_candidates_stage_ranges", "svoh_project_candidates_enqueue_staged", "svoh_project_candidates_enqueue_staged_units", "svoh_project_candidates_wait",
    "svoh_matcher_stage", "svoh_detect_cells_batch", "svoh_detect_cells_batch_enqueue", "svoh_detect_cells_batch_collect", "svoh_detect_fill_features", "svoh_histogram_angle_bins", "svoh_features_upload", "svoh_features_release", "svoh_select_matches_batch",
]


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).
    Two HIP runtimes in one process cannot both own the GPU, so if torch is
    installed (it is only plumbing here: device buffers and torch.distributed in
    bench.py/tests) make libsvo_hip bind to torch's copy, whichever is imported
    first.  Set SVOH_SYSTEM_HIP=1 to bind to /opt/rocm's runtime instead."""
    import importlib.util
    import sys
    if os.environ.get("SVOH_SYSTEM_HIP") == "1" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


# the same library with the test hooks compiled in (-DSVOH_TEST_HOOKS: make testhooks); loaded by the tests that need a
# hook, never by the product
TESTHOOKS_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libsvo_hip_testhooks.so")


def load(path=None):
    """dlopen libsvo_hip.so (in-tree).  Raises RuntimeError if it is missing:
    the product has no CPU path.  path: another build of the same library (tests: TESTHOOKS_LIB_PATH)."""
    global _LIB
    if path is None and _LIB is not None:
        return _LIB
    lib_path = path or LIB_PATH
    if not os.path.exists(lib_path):
        raise RuntimeError(
            "libsvo_hip.so not built (%s). Build it with __graft_entry__.build(); "
            "there is no CPU fallback." % lib_path)
    _share_hip_runtime_with_torch()
    lib = C.CDLL(lib_path)
    P = C.POINTER
    lib.svoh_abi_version.restype = C.c_int
    if lib.svoh_abi_version() != SVOH_ABI_VERSION:   # the ctypes structs below mirror ONE layout of include/svo_hip.h
        raise RuntimeError("%s has ABI version %d, these bindings mirror version %d: rebuild it (__graft_entry__.build())"
                           % (lib_path, lib.svoh_abi_version(), SVOH_ABI_VERSION))
    lib.svoh_create.argtypes = [C.c_int, P(C.c_void_p)]
    lib.svoh_destroy.argtypes = [C.c_void_p]
    lib.svoh_last_error_string.argtypes = [C.c_void_p]
    lib.svoh_last_error_string.restype = C.c_char_p
    lib.svoh_synchronize.argtypes = [C.c_void_p]
    lib.svoh_stream.argtypes = [C.c_void_p]
    lib.svoh_stream.restype = C.c_void_p
    lib.svoh_upload_pyramid.argtypes = [C.c_void_p, C.c_int, P(C.c_void_p), P(C.c_int), P(C.c_int),
                                        P(C.c_int), P(svoh_frame_t)]
    lib.svoh_build_pyramid.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, P(C.c_void_p), P(svoh_frame_t)]
    lib.svoh_build_pyramid_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             P(svoh_frame_t)]
    lib.svoh_download_level.argtypes = [C.c_void_p, svoh_frame_t, C.c_int, C.c_void_p, P(C.c_int), P(C.c_int)]
    lib.svoh_frame_info.argtypes = [C.c_void_p, svoh_frame_t, P(C.c_int), P(C.c_int), P(C.c_int)]
    lib.svoh_release_frame.argtypes = [C.c_void_p, svoh_frame_t]
    lib.svoh_project_candidates_enqueue.argtypes = [C.c_void_p, P(svoh_camera), P(svoh_se3), P(svoh_se3), C.c_int, C.c_int, C.c_void_p,
                                                    C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.svoh_project_candidates_collect.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.svoh_project_candidates.argtypes = [C.c_void_p, P(svoh_camera), P(svoh_se3), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.svoh_camera_maths.argtypes = [C.c_void_p, P(svoh_camera), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.svoh_context_stats.argtypes = [C.c_void_p, C.c_void_p]
    lib.svoh_reload_knobs.argtypes = [C.c_void_p]
    lib.svoh_set_kernel_timing.argtypes = [C.c_void_p, C.c_int]
    lib.svoh_set_align_geometry_classes.argtypes = [C.c_void_p, C.c_int]
    lib.svoh_matcher_begin_deferred.argtypes = [C.c_void_p]
    lib.svoh_matcher_collect.argtypes = [C.c_void_p]
    lib.svoh_matcher_flush.argtypes = [C.c_void_p]
    lib.svoh_matcher_deferred_set_cur_frame.argtypes = [C.c_void_p, P(svoh_frame_view)]
    lib.svoh_sparse_align_batch.argtypes = [C.c_void_p, P(svoh_align_options), C.c_int,
                                            P(svoh_align_problem), P(svoh_align_result)]
    lib.svoh_sparse_align_enqueue.argtypes = [C.c_void_p, P(svoh_align_options), C.c_int,
                                              P(svoh_align_problem)]
    lib.svoh_sparse_align_fetch.argtypes = [C.c_void_p, C.c_int, P(svoh_align_result)]
    lib.svoh_sparse_align_fetch_all.argtypes = [C.c_void_p, C.c_int, P(svoh_align_result)]
    lib.svoh_sparse_align_evaluate.argtypes = [C.c_void_p, P(svoh_align_options), P(svoh_align_problem),
                                               C.c_int, C.c_void_p, C.c_void_p, P(C.c_double),
                                               P(C.c_int32), C.c_void_p, P(C.c_int32)]
    lib.svoh_sparse_align_last_kernel_ms.argtypes = [C.c_void_p, P(C.c_float)]
    lib.svoh_sparse_align_kernel_ms_history.argtypes = [C.c_void_p, C.c_int, P(C.c_float), P(C.c_int)]
    lib.svoh_sparse_align_split_buffers.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_void_p)]
    lib.svoh_sparse_align_split_init.argtypes = [C.c_void_p, P(svoh_align_problem), C.c_void_p]
    lib.svoh_sparse_align_partial_sums.argtypes = [C.c_void_p, P(svoh_align_options), P(svoh_align_problem), C.c_int,
                                                   C.c_int, C.c_void_p, C.c_void_p]
    lib.svoh_sparse_align_gn_update.argtypes = [C.c_void_p, P(svoh_align_options), P(svoh_align_problem), C.c_int,
                                                C.c_int, C.c_void_p, C.c_void_p, P(svoh_align_gn_state)]
    lib.svoh_klt_track_batch.argtypes = [C.c_void_p, P(svoh_klt_options), C.c_int, C.c_void_p, svoh_frame_t,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
    lib.svoh_klt_track_multi.argtypes = [C.c_void_p, P(svoh_klt_options), C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
    lib.svoh_klt_track_indexed.argtypes = [C.c_void_p, P(svoh_klt_options), C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.svoh_last_kernel_ms.argtypes = [C.c_void_p, P(C.c_float)]
    lib.svoh_last_kernel_counters.argtypes = [C.c_void_p, C.c_void_p]
    lib.svoh_match_direct_batch.argtypes = [C.c_void_p, P(svoh_matcher_options), C.c_int, P(svoh_frame_view),
                                            P(svoh_frame_view), P(svoh_feature_batch), C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.svoh_match_direct_batch_pixelwise.argtypes = [C.c_void_p, P(svoh_matcher_options), C.c_int, P(svoh_frame_view),
                                                      P(svoh_frame_view), P(svoh_feature_batch), C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.svoh_update_seeds_batch.argtypes = [C.c_void_p, P(svoh_matcher_options), P(svoh_depth_filter_options),
                                            C.c_int, P(svoh_frame_view), P(svoh_frame_view), P(svoh_feature_batch),
                                            C.c_void_p, C.c_void_p, C.c_void_p, P(C.c_int32)]
    lib.svoh_detect_features.argtypes = [C.c_void_p, svoh_frame_t, P(svoh_detector_options), C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         P(C.c_int32)]
    lib.svoh_optimize_points_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, P(svoh_se3), C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.svoh_optimize_points_batch_enqueue.argtypes = lib.svoh_optimize_points_batch.argtypes[:-1]
    lib.svoh_optimize_points_batch_collect.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.svoh_optimize_pose_batch.argtypes = [C.c_void_p, P(svoh_pose_options), C.c_int, P(svoh_pose_problem),
                                             P(svoh_pose_result)]
    lib.svoh_optimize_pose_batch_packed.argtypes = [C.c_void_p, P(svoh_pose_options), C.c_int, P(svoh_pose_problem),
                                                    P(svoh_pose_packed_arrays), P(svoh_pose_result)]
    lib.svoh_update_seeds_batch_ex.argtypes = lib.svoh_update_seeds_batch.argtypes + [P(svoh_seed_match_outputs)]
    lib.svoh_epipolar_match_batch.argtypes = [C.c_void_p, P(svoh_matcher_options), C.c_int, P(svoh_frame_view),
                                              P(svoh_frame_view), P(svoh_se3), P(svoh_feature_batch), P(C.c_double),
                                              C.c_void_p, P(svoh_epipolar_match_outputs)]
    lib.svoh_host_alloc.argtypes = [C.c_void_p, C.c_size_t, P(C.c_void_p)]
    lib.svoh_host_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.svoh_build_pyramid_multi.argtypes = [C.c_void_p, P(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P(svoh_frame_t)]
    lib.svoh_sparse_align_geometry_key.argtypes = [C.c_void_p, P(svoh_align_options), P(svoh_align_problem), P(C.c_int32)]
    lib.svoh_sparse_align_enqueue_keyed.argtypes = [C.c_void_p, P(svoh_align_options), C.c_int, P(svoh_align_problem), C.c_int32]
    lib.svoh_detect_cells_batch.argtypes = [C.c_void_p, C.c_int, P(svoh_frame_t), P(svoh_detector_options), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.svoh_detect_cells_batch_enqueue.argtypes = [C.c_void_p, C.c_int, P(svoh_frame_t), P(svoh_detector_options), C.c_void_p]
    lib.svoh_detect_cells_batch_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.svoh_histogram_angle_bins.argtypes = [C.c_void_p, C.c_int, P(svoh_frame_t), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.svoh_detect_fill_features.argtypes = [P(svoh_detector_options), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, P(C.c_int32)]
    lib.svoh_build_pyramid_multi_prefetch.argtypes = lib.svoh_build_pyramid_multi.argtypes
    lib.svoh_prefetch_fence.argtypes = [C.c_void_p]
    lib.svoh_matcher_stage.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, P(svoh_matcher_stage_t)]
    lib.svoh_features_upload.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, P(C.c_uint64)]
    lib.svoh_features_release.argtypes = [C.c_void_p, C.c_uint64]
    lib.svoh_select_matches_batch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 7
    lib.svoh_project_candidates_stage.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, P(svoh_candidate_stage_t)]
    lib.svoh_project_candidates_stage_ranges.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, P(svoh_candidate_stage_t)]
    lib.svoh_project_candidates_enqueue_staged.argtypes = [C.c_void_p]
    lib.svoh_project_candidates_wait.argtypes = [C.c_void_p]
    lib.svoh_matcher_begin_deferred.argtypes = [C.c_void_p]
    lib.svoh_matcher_collect.argtypes = [C.c_void_p]
    lib.svoh_matcher_flush.argtypes = [C.c_void_p]
    if path is None:
        _LIB = lib
    return lib
