"""The asserted numeric cases the reference's own test-suite holds near the path (a-15), restated against the
C oracle and the independent NumPy restatement (the device maths: tests/test_reference_held_cases_gpu.py):

  src/vikit/vikit_cameras/test/test_cameras.cpp:83-121   RadialTangentialDistortion: distort -> undistort of (0.5, 0.8) with
      k = (-0.3, 0.1, 9.52e-05, -0.00057) returns to it within 1e-2 (five fixed-point iterations); the distortion's
      analytic Jacobian against finite differences, step 1e-4, tolerance 1e-4
  src/vikit/vikit_cameras/test/test_cameras.cpp:162-173  CameraProjection: project3 -> backProject3 of (0.1, 0.2, 2.0) through the
      camera of test/data/calib_cam.yaml (pinhole, no distortion) returns xyz / z within 1e-8
  src/vikit/vikit_common/test/test_math_utils.cpp:8-20   project2((2, 2, 2)) == (1, 1) exactly

The numbers below (distortion coefficients, intrinsics, points, tolerances) are those files' test data."""
import ctypes as C

import numpy as np

from svo_pro_universal_amd import synth
import np_restatement_direct as nd

RADTAN_K = (-0.3, 0.1, 9.52e-05, -0.00057)                      # test_cameras.cpp:86-87
CALIB_CAM = dict(width=1241, height=376, fx=7.188560000000e+02, fy=7.188560000000e+02, cx=6.071928000000e+02,
                 cy=1.852157000000e+02)                          # test/data/calib_cam.yaml
K_STEP, K_EPS_JACOBIAN = 1e-4, 1e-4                              # test_cameras.cpp:117-118


def unit_radtan_camera():
    """fx = fy = 1, cx = cy = 0: project3 of (x, y, 1) IS distort(x, y), backProject3 IS undistort."""
    return synth.Camera(width=4, height=4, fx=1.0, fy=1.0, cx=0.0, cy=0.0, dist=RADTAN_K)


def oracle_project(orc, cam, p):
    lib = orc.load(); orc._bind_pose(lib)
    c = orc.to_camera(cam)
    p = np.ascontiguousarray(p, np.float64); uv = np.zeros(2); J = np.zeros(6)
    lib.orc_project3(C.byref(c), p.ctypes.data, uv.ctypes.data, J.ctypes.data)
    return uv, J.reshape(2, 3)


def oracle_back_project(orc, cam, uv):
    lib = orc.load()
    lib.orc_back_project3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_back_project3.restype = None
    c = orc.to_camera(cam)
    uv = np.ascontiguousarray(uv, np.float64); f = np.zeros(3)
    lib.orc_back_project3(C.byref(c), uv.ctypes.data, f.ctypes.data)
    return f


def check_radtan_case(project, back_project):
    """project(p3) -> (uv, J 2x3); back_project(uv) -> f (3).  The reference's RadialTangentialDistortion test."""
    uv, _ = project(np.array([0.5, 0.8, 1.0]))                    # distort (x, y)
    f = back_project(uv)                                          # undistort
    assert abs(f[0] - 0.5) < 1e-2 and abs(f[1] - 0.8) < 1e-2     # EXPECT_NEAR(..., 1e-2)
    assert f[2] == 1.0
    # TEST_JACOBIAN_FINITE_DIFFERENCE at the point the test has reached: (x, y) after distort -> undistort
    x, y = f[0], f[1]
    _, J = project(np.array([x, y, 1.0]))
    Jd = J[:, :2]                                                 # z = 1, f = 1: d(uv)/d(x, y) = the distortion's Jacobian
    num = np.zeros((2, 2))
    for k in range(2):
        d = np.zeros(3); d[k] = K_STEP
        a, _ = project(np.array([x, y, 1.0]) + d)
        b, _ = project(np.array([x, y, 1.0]) - d)
        num[:, k] = (a - b) / (2 * K_STEP)
    assert np.abs(Jd - num).max() < K_EPS_JACOBIAN
    return uv, f


def check_projection_case(project, back_project):
    """The reference's CameraProjection test."""
    xyz = np.array([0.1, 0.2, 2.0])
    px, _ = project(xyz)
    res = back_project(px)
    assert np.linalg.norm(xyz / xyz[2] - res) < 0.00000001
    return px, res


def test_radtan_distort_undistort_and_jacobian_oracle_and_numpy(oracle_lib):
    cam = unit_radtan_camera()
    uv_o, f_o = check_radtan_case(lambda p: oracle_project(oracle_lib, cam, p), lambda uv: oracle_back_project(oracle_lib, cam, uv))
    ncam = nd.Cam.of(cam)
    uv_n, f_n = check_radtan_case(lambda p: (ncam.project3(p), ncam.project3_jacobian(p)), ncam.back_project3)
    # the two readings agree to rounding
    assert np.abs(uv_o - uv_n).max() < 1e-15 and np.abs(f_o - f_n).max() < 1e-15
    # five iterations do NOT invert the distortion to machine precision at this point (the reference's own NOTE, :91-92)
    assert 1e-7 < max(abs(f_o[0] - 0.5), abs(f_o[1] - 0.8)) < 1e-2


def test_project_back_project_round_trip_oracle_and_numpy(oracle_lib):
    cam = synth.Camera(**CALIB_CAM)
    px_o, r_o = check_projection_case(lambda p: oracle_project(oracle_lib, cam, p), lambda uv: oracle_back_project(oracle_lib, cam, uv))
    ncam = nd.Cam.of(cam)
    px_n, r_n = check_projection_case(lambda p: (ncam.project3(p), None), ncam.back_project3)
    assert np.array_equal(px_o, px_n) and np.abs(r_o - r_n).max() < 1e-15
    # known answer: fx * 0.05 + cx, fy * 0.1 + cy
    assert np.allclose(px_o, [718.856 * 0.05 + 607.1928, 718.856 * 0.1 + 185.2157], rtol=0, atol=1e-12)


def test_project2_exact():
    """test_math_utils.cpp:8-20: vk::project2 = head<2>() / v(2) (math_utils.h:143-148), as the unit-plane scan uses it."""
    v = np.array([2.0, 2.0, 2.0])
    x2 = v[:2] / v[2]
    assert x2[0] == 1.0 and x2[1] == 1.0
    v32 = np.array([2.0, 2.0, 2.0], np.float32)
    x2f = v32[:2] / v32[2]
    assert x2f.dtype == np.float32 and x2f[0] == 1.0 and x2f[1] == 1.0
