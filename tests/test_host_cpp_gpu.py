"""Runs the C++ host-layer test (tests/cpp/test_host_align.cpp): SparseImgAlignHip,
the mirror of the reference's SparseImgAlign interface, against the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from svo_pro_universal_amd import synth

import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def dump_scene(path, sc, use_prior):
    cam = sc.cam
    with open(path, "wb") as f:
        f.write(struct.pack("4i", cam.width, cam.height, sc.n_features, int(use_prior)))
        dist = cam.dist or [0, 0, 0, 0]
        np.array([cam.fx, cam.fy, cam.cx, cam.cy] + list(dist) + [0.0 if cam.dist is None else 1.0]).tofile(f)
        sc.T_cam_imu.as7().tofile(f)
        sc.T_ref_f_w.as7().tofile(f)
        sc.T_ref_f_w.as7().tofile(f)  # the new frame starts at the last frame's pose
        np.ascontiguousarray(sc.px, np.float64).tofile(f)
        np.ascontiguousarray(sc.f, np.float64).tofile(f)
        np.ascontiguousarray(sc.pos_world, np.float64).tofile(f)
        np.ascontiguousarray(sc.flags, np.uint8).tofile(f)
        sc.img_ref.tofile(f)
        sc.img_cur.tofile(f)


@pytest.mark.parametrize("cam_kind,use_prior", [("pinhole", 0), ("radtan", 1)])
def test_cpp_host_layer_matches_oracle(tmp_path, oracle_lib, cam_kind, use_prior):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "svo_pro_universal_amd", "host")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    cam = synth.Camera.test_camera() if cam_kind == "pinhole" else synth.Camera.euroc_like()
    sc = helpers.small_scene(77, n=400, cam=cam, border_features=40, invalid_fraction=0.05)
    path = str(tmp_path / "scene.bin")
    dump_scene(path, sc, use_prior)
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_host_align"), path], capture_output=True, text=True)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "PASS" in out.stdout


def test_cpp_depth_filter_and_klt_mirrors_match_oracle(tmp_path, oracle_lib):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "svo_pro_universal_amd", "host")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    cam = synth.Camera.euroc_like()
    sc = synth.make_align_scene(78, n_features=10, cam=cam, rot_deg=(0.4, 1.0), trans_m=(0.06, 0.12))
    sd = synth.make_seed_set(sc, 800, margin=12)
    sd["type"][::13] = 2  # map point seeds
    tr = synth.make_track_set(sc, 120, margin=12)
    path = str(tmp_path / "seeds.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("4i", cam.width, cam.height, 800, 120))
        np.array([cam.fx, cam.fy, cam.cx, cam.cy] + list(cam.dist) + [1.0]).tofile(f)
        sc.T_ref_f_w.as7().tofile(f); sc.T_cur_f_w_gt.as7().tofile(f)
        np.array([sd["mu_range"]]).tofile(f)
        for k in ("px", "f", "grad", "state"):
            np.ascontiguousarray(sd[k], np.float64).tofile(f)
        np.ascontiguousarray(sd["level"], np.int32).tofile(f)
        np.ascontiguousarray(sd["type"], np.uint8).tofile(f)
        np.ascontiguousarray(tr["px_ref"], np.int32).tofile(f)
        np.ascontiguousarray(tr["px_cur_init"], np.float64).tofile(f)
        sc.img_ref.tofile(f); sc.img_cur.tofile(f)
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_host_seeds_klt"), path], capture_output=True, text=True)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "PASS" in out.stdout


@pytest.mark.parametrize("max_n,device_select", [(60, 0), (0, 0), (60, 1)])
def test_cpp_reprojector_match_candidates_mirror_matches_oracle(tmp_path, oracle_lib, max_n, device_select):
    """reprojector_utils::matchCandidates (reprojector.cpp:342-486): speculative GPU batches + ordered host
    replay vs the oracle's sequential loop -- converged seeds, seed updates, landmarks with and without a close
    view, pre-occupied grid cells, the early break at max_n_features_per_frame."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "svo_pro_universal_amd", "host")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    cam = synth.Camera.euroc_like()
    sc = synth.make_align_scene(91, n_features=10, cam=cam, rot_deg=(0.4, 1.0), trans_m=(0.06, 0.12))
    n = 400
    sd = synth.make_seed_set(sc, n, margin=14, levels=(0, 1, 2))
    rng = np.random.RandomState(4)
    typ = sd["type"].copy()                       # 0 edgelet seed / 1 corner seed
    state = sd["state"].copy().reshape(-1, 4)
    kind = rng.randint(0, 10, n)
    conv = kind < 3                               # converged seeds: matched directly at the seed depth
    state[conv, 0] = 1.0 / sd["true_depth"][conv]
    typ[conv] = typ[conv] + 3
    mp = kind == 3                                # map-point seeds (unconverged)
    typ[mp] = 2
    lm = (kind == 4) | (kind == 5)                # landmarks: types kEdgelet / kCorner
    typ[lm] = np.where(sd["type"][lm] == 0, 6, 7)
    lm_far = kind == 6                            # landmarks observed only from far away
    typ[lm_far] = 7
    lm_kind = np.zeros(n, np.uint8); lm_kind[lm] = 1; lm_kind[lm_far] = 2
    x_ref = sd["f"].reshape(-1, 3).T * sd["true_depth"]
    x_w = sc.T_w_ref.transform(x_ref)
    px_true = sc.cam.project(sc.T_w_cur.inverse().transform(x_w))
    cur_px = np.ascontiguousarray((px_true + rng.uniform(-1.5, 1.5, px_true.shape)).T).ravel()
    order = rng.permutation(n).astype(np.int32)
    # a frame on the far side of the scene plane: its observation is never a close view (cos < 0.4)
    T_far_w = (sc.T_w_ref * synth.SE3(synth.quat_from_axis_angle([0, 1, 0], 2.5), [0.0, 0.0, 25.0])).inverse()
    path = str(tmp_path / "reproj.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("4i", cam.width, cam.height, n, max_n))
        np.array([cam.fx, cam.fy, cam.cx, cam.cy] + list(cam.dist) + [1.0]).tofile(f)
        sc.T_ref_f_w.as7().tofile(f); sc.T_cur_f_w_gt.as7().tofile(f); T_far_w.as7().tofile(f)
        np.array([sd["mu_range"]]).tofile(f)
        np.ascontiguousarray(sd["px"], np.float64).tofile(f)
        np.ascontiguousarray(sd["f"], np.float64).tofile(f)
        np.ascontiguousarray(sd["grad"], np.float64).tofile(f)
        np.ascontiguousarray(state.ravel(), np.float64).tofile(f)
        np.ascontiguousarray(x_w.T.ravel(), np.float64).tofile(f)
        np.ascontiguousarray(sd["level"], np.int32).tofile(f)
        np.ascontiguousarray(typ, np.uint8).tofile(f)
        lm_kind.tofile(f)
        order.tofile(f)
        cur_px.tofile(f)
        rng.uniform(10, 100, n).tofile(f)
        sc.img_ref.tofile(f); sc.img_cur.tofile(f)
    # device_select: which candidates a pass tries and where it ends comes from svoh_select_matches_batch instead of the host's walk
    # over the grid (SVOH_REPROJ_DEVICE_SELECT=1) -- same features, counters, grid, trash list
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_host_reprojector"), path], capture_output=True, text=True,
                         env=dict(os.environ, SVOH_REPROJ_DEVICE_SELECT=str(device_select)))
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "PASS" in out.stdout


@pytest.mark.parametrize("first_obs", [1, 0])
def test_cpp_feature_tracker_mirror_matches_oracle(tmp_path, oracle_lib, first_obs):
    """FeatureTracker::trackFrameBundle (feature_tracker.cpp:52-122) over a 2-camera bundle and three time
    steps: template from the first / the last observation, start from the last position, truncated reference
    pixel, terminated tracks, new frame's px / track ids / bearing vectors."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "svo_pro_universal_amd", "host")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    cam = synth.Camera.euroc_like()
    n = 150
    imgs, px0 = [[], [], []], []
    for c, seed in enumerate((95, 96)):
        sc = synth.make_align_scene(seed, n_features=10, cam=cam, rot_deg=(0.4, 1.0), trans_m=(0.03, 0.08))
        T_ref_cur = sc.T_w_ref.inverse() * sc.T_w_cur
        T_w_2 = sc.T_w_cur * T_ref_cur                       # the same motion once more
        imgs[0].append(sc.img_ref); imgs[1].append(sc.img_cur)
        imgs[2].append(synth.render(cam, T_w_2, sc.plane, sc.tex))
        tr = synth.make_track_set(sc, n, seed=c, margin=10)   # some start close to the border and get lost
        px0.append(np.asarray(tr["px_ref"], np.int32))
    path = str(tmp_path / "tracker.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("4i", cam.width, cam.height, n, first_obs))
        np.array([cam.fx, cam.fy, cam.cx, cam.cy] + list(cam.dist) + [1.0]).tofile(f)
        np.concatenate(px0).astype(np.int32).tofile(f)
        for t in range(3):
            for c in range(2):
                np.ascontiguousarray(imgs[t][c], np.uint8).tofile(f)
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_host_tracker"), path], capture_output=True, text=True)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "PASS" in out.stdout


@pytest.mark.parametrize("error_type", [0, 1, 2])
def test_cpp_pose_optimizer_mirror_matches_oracle(tmp_path, oracle_lib, error_type):
    """PoseOptimizer::run as FrameHandlerBase::optimizePose calls it: points from landmarks and seed references,
    pose written back into the frame, outliers marked."""
    import pose_helpers as ph
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "svo_pro_universal_amd", "host"), "libsvo_hip_host.so"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    sc = ph.make_pose_scene(70 + error_type, n=220)
    c = sc["cams"][0]
    cam = sc["cam"]
    T_kf_w = synth.SE3(synth.quat_from_axis_angle([0.3, -0.2, 0.9], 0.4), [0.5, -0.3, 0.2])
    path = str(tmp_path / "pose.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("4i", cam.width, cam.height, len(c["level"]), error_type))
        np.array([cam.fx, cam.fy, cam.cx, cam.cy] + list(cam.dist) + [1.0]).tofile(f)
        c["T_cam_imu"].as7().tofile(f); sc["T_imu_world_init"].as7().tofile(f); T_kf_w.as7().tofile(f)
        for k in ("px", "f", "grad", "xyz_world"):
            np.ascontiguousarray(c[k], np.float64).tofile(f)
        np.ascontiguousarray(c["level"], np.int32).tofile(f)
        np.ascontiguousarray(c["type"], np.uint8).tofile(f)
        np.ascontiguousarray(c["usable"], np.uint8).tofile(f)
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_host_pose"), path], capture_output=True, text=True)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "PASS" in out.stdout


def test_cpp_structure_optimisation_and_map_mirrors():
    """optimizeStructure (FrameHandlerBase::optimizeStructure -> Point::optimize on the device) against the oracle
    per landmark, and the Map / key-point mirrors against brute force (tests/cpp/test_host_map_structure.cpp)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "svo_pro_universal_amd", "host"), "libsvo_hip_host.so"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_host_map_structure")], capture_output=True, text=True)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "PASS" in out.stdout


@pytest.mark.parametrize("n_want,n_have", [(60, 5), (4000, 0)])
def test_cpp_stereo_triangulation_mirror_matches_oracle(tmp_path, oracle_lib, n_want, n_have):
    """StereoTriangulationHip::compute (stereo_triangulation.cpp:23-140) vs the oracle's sequential loop: the early
    stop at n_desired (60 wanted, 5 landmarks already there) and the run through every new feature (4000 wanted)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "svo_pro_universal_amd", "host")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    cam = synth.Camera.euroc_like()
    # left = the scene's reference view, right = its current view: a 9-12 cm baseline with a slight vergence
    sc = synth.make_align_scene(81, n_features=10, cam=cam, rot_deg=(0.2, 0.6), trans_m=(0.09, 0.12))
    T_c0_b = sc.T_cam_imu
    T_b_w = T_c0_b.inverse() * sc.T_ref_f_w
    T_c1_b = sc.T_cur_f_w_gt * T_b_w.inverse()
    path = str(tmp_path / "stereo.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("4i", cam.width, cam.height, n_want, n_have))
        np.array([cam.fx, cam.fy, cam.cx, cam.cy] + list(cam.dist) + [1.0]).tofile(f)
        T_c0_b.as7().tofile(f); T_c1_b.as7().tofile(f); T_b_w.as7().tofile(f)
        np.array([1.0 / 3.0, 1.0 / 1.0, 1.0 / 50.0]).tofile(f)      # StereoTriangulationOptions defaults
        sc.img_ref.tofile(f); sc.img_cur.tofile(f)
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_host_stereo"), path], capture_output=True, text=True)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "PASS" in out.stdout


def test_cpp_500_frames_keep_device_memory_bounded(oracle_lib):
    """500 frames through SparseImgAlignHip / DepthFilterHip with the adapter's per-Frame handle cache
    (DeviceFrameCache): live device frames and device bytes stay bounded, everything is released at the end."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "svo_pro_universal_amd", "host")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_host_leak"), "500"], capture_output=True, text=True, timeout=280)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "PASS" in out.stdout
