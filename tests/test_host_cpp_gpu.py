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
