"""The reference-held camera cases (tests/test_reference_held_cases_cpu.py) against the DEVICE compilation of the camera
maths every kernel uses (csrc/svoh_math.h through the parity entry svoh_camera_maths), and device = oracle on a cloud
of points."""
import numpy as np
import pytest

from svo_pro_universal_amd import synth
import test_reference_held_cases_cpu as cases

pytestmark = pytest.mark.gpu


def test_radtan_case_on_the_device(gpu_ctx, oracle_lib):
    cam = cases.unit_radtan_camera()
    px, J, fb = gpu_ctx.camera_maths(cam, [0.5, 0.8, 1.0])            # distort, Jacobian, undistort of the distorted point
    assert abs(fb[0] - 0.5) < 1e-2 and abs(fb[1] - 0.8) < 1e-2 and fb[2] == 1.0
    x, y = fb[0], fb[1]
    pts = np.array([[x, y, 1.0], [x + cases.K_STEP, y, 1.0], [x - cases.K_STEP, y, 1.0], [x, y + cases.K_STEP, 1.0], [x, y - cases.K_STEP, 1.0]])
    p2, J2, _ = gpu_ctx.camera_maths(cam, pts.ravel())
    p2 = p2.reshape(-1, 2)
    num = np.stack([(p2[1] - p2[2]) / (2 * cases.K_STEP), (p2[3] - p2[4]) / (2 * cases.K_STEP)], axis=1)
    assert np.abs(J2[:6].reshape(2, 3)[:, :2] - num).max() < cases.K_EPS_JACOBIAN
    # the device agrees with the oracle to rounding (FMA contraction on the device)
    uv_o, _ = cases.oracle_project(oracle_lib, cam, np.array([0.5, 0.8, 1.0]))
    f_o = cases.oracle_back_project(oracle_lib, cam, uv_o)
    assert np.abs(px - uv_o).max() < 1e-15 and np.abs(fb - f_o).max() < 1e-14


def test_projection_round_trip_on_the_device(gpu_ctx, oracle_lib):
    cam = synth.Camera(**cases.CALIB_CAM)
    xyz = np.array([0.1, 0.2, 2.0])
    px, J, fb = gpu_ctx.camera_maths(cam, xyz)
    assert np.linalg.norm(xyz / xyz[2] - fb) < 0.00000001
    px_o, _ = cases.oracle_project(oracle_lib, cam, xyz)
    assert np.abs(px - px_o).max() < 1e-12


def test_device_camera_maths_equals_the_oracle_on_a_cloud(gpu_ctx, oracle_lib):
    rng = np.random.RandomState(5)
    for cam in (synth.Camera.test_camera(), synth.Camera.euroc_like(752, 480)):
        pts = np.stack([rng.uniform(-2, 2, 500), rng.uniform(-1.5, 1.5, 500), rng.uniform(1.0, 8.0, 500)], axis=1)
        px, J, fb = gpu_ctx.camera_maths(cam, pts.ravel())
        for i in range(0, 500, 7):
            uv_o, J_o = cases.oracle_project(oracle_lib, cam, pts[i])
            f_o = cases.oracle_back_project(oracle_lib, cam, uv_o)
            assert np.abs(px[2 * i:2 * i + 2] - uv_o).max() < 1e-10
            assert np.allclose(J[6 * i:6 * i + 6].reshape(2, 3), J_o, rtol=1e-12, atol=1e-12)
            assert np.abs(fb[3 * i:3 * i + 3] - f_o).max() < 1e-12
