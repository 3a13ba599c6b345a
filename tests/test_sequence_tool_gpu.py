"""the sequence runner (tools/svoh_track_sequence.cpp) on a synthetic EuRoC-layout dataset:
PNG + data.csv + calibration YAML in, GPU pyramid + FeatureTracker::trackAndDetect per frame, tracks out.
Checked against the scene's ground truth: a track's pixel in frame k is where the 3-D point seen at its first
observation projects in frame k."""
import os
import subprocess

import numpy as np
import pytest

from svo_pro_universal_amd import synth
from test_io_cpu import write_png

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_track_sequence_tool(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "svo_pro_universal_amd", "host")])
    cam = synth.Camera.euroc_like(752, 480)
    sc = synth.make_align_scene(150, n_features=8, cam=cam, rot_deg=(0.2, 0.4), trans_m=(0.01, 0.02))
    step = sc.T_w_ref.inverse() * sc.T_w_cur
    n_frames = 10
    poses = [sc.T_w_ref]
    for k in range(1, n_frames):
        poses.append(poses[-1] * step)
    data = tmp_path / "ds" / "mav0" / "cam0" / "data"
    data.mkdir(parents=True)
    stamps = [1403636579763555584 + 50000000 * k for k in range(n_frames)]
    for k, T in enumerate(poses):
        img = synth.render(cam, T, sc.plane, sc.tex)
        write_png(str(data / ("%d.png" % stamps[k])), img if k % 2 else np.stack([img] * 3, -1), chunk=65536)
    (tmp_path / "ds" / "mav0" / "cam0" / "data.csv").write_text("#timestamp [ns],filename\n" + "".join("%d,%d.png\n" % (t, t) for t in stamps))
    (tmp_path / "calib.yaml").write_text("""label: synthetic
cameras:
- camera:
    label: cam0
    image_height: %d
    image_width: %d
    type: pinhole
    intrinsics:
      data: [%.17g, %.17g, %.17g, %.17g]
    distortion:
      type: radial-tangential
      parameters:
        data: [%.17g, %.17g, %.17g, %.17g]
  T_B_C:
    data: [1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0]
""" % ((cam.height, cam.width, cam.fx, cam.fy, cam.cx, cam.cy) + tuple(cam.dist)))
    (tmp_path / "params.yaml").write_text("grid_size: 30\nn_pyr_levels: 3\ndetector_threshold_secondary: 100\nklt_max_level: 4\n")
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    tool = os.path.join(ROOT, "svo_pro_universal_amd", "host", "svoh_track_sequence")
    r = subprocess.run([tool, str(tmp_path / "ds"), str(tmp_path / "calib.yaml"), str(tmp_path / "params.yaml"), str(out_dir)],
                       capture_output=True, text=True)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    tr = np.loadtxt(str(out_dir / "tracks.csv"), delimiter=",", skiprows=1)
    tm = np.loadtxt(str(out_dir / "timing.csv"), delimiter=",", skiprows=1)
    assert len(tm) == n_frames and tm[0, 2] > 200            # the first frame detects a full grid of features
    frames, ids, xy = tr[:, 0].astype(int), tr[:, 2].astype(int), tr[:, 3:5]
    first = {}
    errs = []
    n_cam = np.asarray(sc.plane.n, float)
    for f, i, p in zip(frames, ids, xy):
        if i not in first:
            first[i] = (f, p)
            continue
        f0, p0 = first[i]
        # back-project the first observation onto the scene plane, project into frame f
        x, y = cam.undistorted_xy(np.array([p0[0]]), np.array([p0[1]]))
        ray_w = poses[f0].R() @ np.array([x[0], y[0], 1.0])
        o = poses[f0].t
        lam = (sc.plane.h - n_cam @ o) / (n_cam @ ray_w)
        X = o + lam * ray_w
        q = cam.project(poses[f].inverse().transform(X.reshape(3, 1)))[:, 0]
        errs.append(np.linalg.norm(q - p))
    errs = np.asarray(errs)
    long_tracks = sum(1 for i in first if (ids == i).sum() == n_frames)
    assert len(errs) > 1000 and np.median(errs) < 0.15 and np.percentile(errs, 95) < 0.6, (np.median(errs), np.percentile(errs, 95))
    assert long_tracks > 150                                 # most tracks survive the whole sequence
    # a missing image is an error message and a non-zero exit, not a crash
    os.remove(str(data / ("%d.png" % stamps[3])))
    r = subprocess.run([tool, str(tmp_path / "ds"), str(tmp_path / "calib.yaml"), "-", str(out_dir)], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr
