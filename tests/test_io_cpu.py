"""dataset / configuration plumbing of the host layer (svo_hip::io) and scripts/ate.py.
Inputs are written here in the formats of the reference's own files (same keys and structure as
examples/param/pinhole.yaml and examples/param/calib/euroc_mono.yaml; values are this test's)."""
import os
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
BIN = os.path.join(ROOT, "tests", "cpp", "test_io")


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(os.path.join(ROOT, "svo_pro_universal_amd", "csrc", "libsvo_hip.so")):
        pytest.skip("libsvo_hip.so not built (run __graft_entry__.build())")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "svo_pro_universal_amd", "host"), "libsvo_hip_host.so"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "test_io"])


def run(*args):
    out = subprocess.run([BIN] + list(args), capture_output=True, text=True)
    return out.returncode, dict(l.split(" ", 1) for l in out.stdout.strip().splitlines() if " " in l), out.stdout


def write_png(path, arr, filters=(0, 1, 2, 3, 4), chunk=777):
    """Minimal PNG encoder: arr HxW (grey) or HxWxC, all five scanline filters in rotation, IDAT split in chunks."""
    a = np.asarray(arr, np.uint8)
    h, w = a.shape[:2]
    ch = 1 if a.ndim == 2 else a.shape[2]
    ct = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    a = a.reshape(h, w * ch).astype(np.int32)
    raw = bytearray()
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        cur = a[y]
        left = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        ul = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        ft = filters[y % len(filters)]
        if ft == 0: line = cur
        elif ft == 1: line = cur - left
        elif ft == 2: line = cur - prev
        elif ft == 3: line = cur - ((left + prev) >> 1)
        else:
            p = left + prev - ul
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            line = cur - pred
        raw.append(ft); raw += bytes((line & 255).astype(np.uint8))
        prev = cur
    comp = zlib.compress(bytes(raw), 6)

    def chunk_bytes(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk_bytes(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ct, 0, 0, 0)))
        f.write(chunk_bytes(b"tEXt", b"Comment\x00written by tests/test_io_cpu.py"))
        for i in range(0, len(comp), chunk):
            f.write(chunk_bytes(b"IDAT", comp[i:i + chunk]))
        f.write(chunk_bytes(b"IEND", b""))


def test_yaml_subset(tmp_path):
    p = tmp_path / "t.yaml"
    p.write_text("""# comment
a:
  b:
    c: 42   # trailing comment
list: [1, 2.5,
       3.75, 4]
items:
- name: first
  v: 1
- name: "second # not a comment"
  v: 2
text: 'quoted: value'
T_world_imuinit/qw: 1
""")
    rc, d, out = run("yaml", str(p))
    assert rc == 0, out
    assert d["a.b.c"] == "42" and d["list"] == "4 3.75" and d["seq"] == "2 first second # not a comment"
    assert d["str"] == "quoted: value" and d["keyslash"] == "1" and d["missing"] == "-7"


def test_camera_rig_and_params(tmp_path):
    R = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])   # 90 degrees about z
    calib = tmp_path / "calib.yaml"
    calib.write_text("""label: "rig"
id: 412eab8e4058621f7036b5e765dfe812
cameras:
- camera:
    label: cam0
    id: 54812562fa109c40fe90b29a59dd7798
    line-delay-nanoseconds: 0
    image_height: 480
    image_width: 752
    type: pinhole
    intrinsics:
      cols: 1
      rows: 4
      data: [458.654, 457.296, 367.215, 248.375]
    distortion:
      type: radial-tangential
      parameters:
        cols: 1
        rows: 4
        data: [-0.28340811, 0.07395907, 0.00019359,
    1.76187114e-05]
  T_B_C:
    cols: 4
    rows: 4
    data: [%s, 0.1,
           %s, -0.2,
           %s, 0.3,
           0.0, 0.0, 0.0, 1.0]

imu_params:
  delay_imu_cam: 0.0
""" % tuple(", ".join("%.17g" % v for v in R[i]) for i in range(3)))
    rc, d, out = run("rig", str(calib))
    assert rc == 0, out
    assert d["label"] == "cam0" and d["size"] == "752 480"
    assert [float(x) for x in d["intrinsics"].split()] == [458.654, 457.296, 367.215, 248.375]
    dist = d["distortion"].split()
    assert dist[0] == "1" and float(dist[4]) == 1.76187114e-05
    T = [float(x) for x in d["T_B_C"].split()]
    assert np.allclose(T[:4], [np.sqrt(0.5), 0, 0, np.sqrt(0.5)]) and T[4:] == [0.1, -0.2, 0.3]
    # defaults of svo_factory.cpp when the file is empty, and overrides
    rc, d0, out = run("params", "-")
    assert d0["img_align"] == "4 2 0 0 0 0" and d0["reprojector"].startswith("160 35 200 1 0")
    assert d0["depth_filter"] == "1 200 500 0 1 0 360" and d0["detector"] == "35 2 10 200 1" and d0["tracker"] == "4 0 pyr 5"
    params = tmp_path / "p.yaml"
    params.write_text("""pipeline_is_stereo: False
max_fts: 180
grid_size: 30
n_pyr_levels: 3
detector_threshold_primary: 10
detector_threshold_secondary: 100
use_edgelets: False
img_align_max_level: 4
img_align_min_level: 1
img_align_est_illumination_gain: True
img_align_prior_lambda_rot: 0.5
use_threaded_depthfilter: False
scan_epi_unit_sphere: True
max_seeds_ratio: 2.0
klt_min_level: 1
""")
    rc, d1, out = run("params", str(params))
    assert d1["img_align"] == "4 1 0 0 1 0" and d1["prior"] == "0.5 0" and d1["reprojector"].startswith("180 30 ")
    assert d1["depth_filter"] == "0 200 500 1 1 0 360" and d1["detector"] == "30 2 10 100 0" and d1["tracker"] == "4 1 pyr 5"


@pytest.mark.parametrize("channels", [1, 2, 3, 4])
def test_png_reader(tmp_path, channels):
    rng = np.random.RandomState(channels)
    h, w = 37, 53
    base = (np.add.outer(np.arange(h) * 3, np.arange(w) * 2) % 256).astype(np.uint8)
    if channels == 1:
        arr, want = base, base
    elif channels == 2:
        arr = np.stack([base, rng.randint(0, 256, (h, w)).astype(np.uint8)], -1); want = base
    else:
        rgb = np.stack([base, np.roll(base, 5, 1), rng.randint(0, 256, (h, w)).astype(np.uint8)], -1)
        arr = rgb if channels == 3 else np.concatenate([rgb, np.full((h, w, 1), 255, np.uint8)], -1)
        r, g, b = [rgb[..., i].astype(np.int64) for i in range(3)]
        want = ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)
    p = str(tmp_path / "img.png")
    write_png(p, arr)
    rc, d, out = run("png", p, str(tmp_path / "out.raw"))
    assert rc == 0 and d["size"] == "%d %d" % (w, h), out
    got = np.fromfile(str(tmp_path / "out.raw"), np.uint8).reshape(h, w)
    assert np.array_equal(got, want)
    # grey stored as RGB comes back unchanged (EuRoC images read with cv::imread's default + BGR2GRAY)
    if channels == 3:
        write_png(p, np.stack([base] * 3, -1))
        run("png", p, str(tmp_path / "out.raw"))
        assert np.array_equal(np.fromfile(str(tmp_path / "out.raw"), np.uint8).reshape(h, w), base)
    # corruption is an error, not garbage
    blob = bytearray(open(p, "rb").read()); blob[60] ^= 0xFF
    open(p, "wb").write(bytes(blob))
    rc, d, out = run("png", p)
    assert rc == 1 and "error" in out


def test_euroc_folder_and_trajectory(tmp_path):
    cam0 = tmp_path / "mav0" / "cam0"
    (cam0 / "data").mkdir(parents=True)
    stamps = [1403636579763555584 + 50000000 * k for k in range(4)]
    (cam0 / "data.csv").write_text("#timestamp [ns],filename\n" + "".join("%d,%d.png\r\n" % (t, t) for t in stamps) + "\n")
    rc, d, out = run("euroc", str(tmp_path))
    assert rc == 0 and d["n"] == "4"
    lines = [l for l in out.splitlines() if l.startswith("frame")]
    assert lines[3].split()[1] == str(stamps[3]) and lines[3].endswith("/mav0/cam0/data/%d.png" % stamps[3])
    traj = str(tmp_path / "traj.txt")
    assert run("traj", traj)[0] == 0
    row = [l for l in open(traj) if l[0] != "#"][0].split()
    assert row[0] == "1403636579.763555584" and [float(x) for x in row[1:]] == [1.0, -2.0, 3.25, 0.5, -0.5, 0.5, 0.5]


def test_ate_script(tmp_path):
    import ate
    rng = np.random.RandomState(0)
    n = 200
    t = np.arange(n) * 0.05
    gt = np.column_stack([t, np.cumsum(rng.normal(0, 0.05, (n, 3)), 0), np.tile([0, 0, 0, 1.0], (n, 1))])
    ang = 0.7
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    est = gt.copy()
    est[:, 1:4] = (R @ (gt[:, 1:4] / 2.5).T).T + [3.0, -1.0, 0.5]          # similarity-transformed copy
    est[:, 0] += 0.001
    r = ate.ate(est, gt, with_scale=True)
    assert r["n"] == n and r["rmse"] < 1e-9 and abs(r["scale"] - 2.5) < 1e-9
    assert ate.ate(est, gt, with_scale=False)["rmse"] > 0.05               # a rigid fit cannot absorb the scale
    noisy = est.copy(); noisy[:, 1:4] += rng.normal(0, 0.01, (n, 3))
    assert 0.03 < ate.ate(noisy, gt, with_scale=True)["rmse"] < 0.06      # 0.01 * 2.5 * sqrt(3)
    e, g = str(tmp_path / "e.txt"), str(tmp_path / "g.txt")
    np.savetxt(e, est, fmt="%.9f"); np.savetxt(g, gt, fmt="%.9f")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ate.py"), e, g, "--scale"], capture_output=True, text=True)
    assert out.returncode == 0 and "ATE rmse 0.0000" in out.stdout


def test_map_and_key_point_mirrors_host_logic():
    """Frame::setKeyPoints and Map::getClosestNKeyframesWithOverlap & co. are host logic: the C++ test's map part
    runs without a GPU (the structure-optimisation part of the same binary is a GPU test)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "svo_pro_universal_amd", "host"), "libsvo_hip_host.so"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "test_host_map_structure"])
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_host_map_structure"), "--map-only"], capture_output=True, text=True)
    assert out.returncode == 0 and "PASS" in out.stdout, out.stdout + out.stderr


REF_PARAM = "/root/reference/examples/param"


def _dump(what, path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "svo_pro_universal_amd", "host")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "test_io"])
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_io"), what, path], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    return dict(l.split(" ", 1) for l in out.stdout.strip().splitlines())


@pytest.mark.skipif(not os.path.isdir(REF_PARAM), reason="the reference tree exists in the build container only")
def test_reference_configuration_files_parse_to_survey_appendix_a():
    """The loader on the reference's OWN files (read in place, never copied): examples/param/pinhole.yaml,
    calib/euroc_mono.yaml, calib/euroc_stereo.yaml, frontend_imu/euroc_stereo_imu.yaml -> the values of SURVEY.md
    Appendix A (the loader had only ever been fed YAML the tests wrote)."""
    # --- calib/euroc_mono.yaml: 752x480 pinhole + radtan (examples/param/calib/euroc_mono.yaml:8-21)
    rig = _dump("rig", os.path.join(REF_PARAM, "calib", "euroc_mono.yaml"))
    assert rig["label"] == "cam0" and rig["size"] == "752 480"
    fx, fy, cx, cy = map(float, rig["intrinsics"].split())
    assert (fx, fy, cx, cy) == (458.6548807207614, 457.2966964634893, 367.2158039615726, 248.37534060980727)
    d = rig["distortion"].split()
    assert int(d[0]) == 1                                             # SVOH_DISTORTION_RADTAN
    assert [float(x) for x in d[1:]] == [-0.28340811217029355, 0.07395907389290132, 0.00019359502856909603, 1.7618711454538528e-05]
    q = np.array(list(map(float, rig["T_B_C"].split())))
    R = np.array([[0.0148655429818, -0.999880929698, 0.00414029679422], [0.999557249008, 0.0149672133247, 0.025715529948],
                  [-0.0257744366974, 0.00375618835797, 0.999660727178]])
    w, x, y, z = q[:4]
    Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    assert np.abs(Rq - R).max() < 1e-6 and np.allclose(q[4:], [-0.0216401454975, -0.064676986768, 0.00981073058949])
    # --- pinhole.yaml (the README's mono example): max_fts 180, grid 30, levels 4..2, FAST 10 / edgelet 200
    p = _dump("params", os.path.join(REF_PARAM, "pinhole.yaml"))
    assert p["img_align"] == "4 2 0 0 0 0"                            # max / min level, no robust weight, no illumination terms
    assert p["prior"].split() == ["0", "0"]
    mf, grid, sigma2, ro, rg = p["reprojector"].split()
    assert (int(mf), int(grid), float(sigma2), int(ro), int(rg)) == (180, 30, 200.0, 1, 0)
    thr, s2, mp2, sphere, ao, ag, nseeds = p["depth_filter"].split()
    assert int(thr) == 1                                              # use_threaded_depthfilter: the file leaves the default (true)
    assert (float(s2), int(sphere), int(ao), int(ag)) == (200.0, 0, 1, 0) and int(nseeds) == 180 * 3
    cell, maxlvl, t1, t2, grad = p["detector"].split()
    assert (int(cell), int(maxlvl), float(t1), float(t2)) == (30, 2, 10.0, 200.0)
    assert p["tracker"].split()[-1] == "5"                            # img_align_max_level + 1 pyramid levels
    # --- frontend_imu/euroc_stereo_imu.yaml (C4): illumination gain + offset on, depth filter thread OFF, grid 35
    s = _dump("params", os.path.join(REF_PARAM, "frontend_imu", "euroc_stereo_imu.yaml"))
    assert s["img_align"].split()[4:] == ["1", "1"]
    assert s["depth_filter"].split()[0] == "0" and s["depth_filter"].split()[4:6] == ["1", "1"]
    mf, grid, _, ro, rg = s["reprojector"].split()
    assert (int(mf), int(grid), int(ro), int(rg)) == (160, 35, 1, 1)   # max_fts absent -> the factory's default 160
    assert int(s["depth_filter"].split()[-1]) == 120 * 3               # depth-filter default max_fts 120 x max_seeds_ratio 3
    # --- calib/euroc_stereo.yaml: two cameras
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_io"), "rig", os.path.join(REF_PARAM, "calib", "euroc_stereo.yaml")],
                         capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.count("label ") == 2 and out.stdout.count("size 752 480") == 2


def test_gyroscope_rotation_prior_of_an_euroc_folder(tmp_path):
    """io::readEurocImu + io::relativeRotationPrior (ImuHandler::getRelativeRotationPrior, imu_handler.cpp:157-233,
    270-297) against an independent integration in NumPy: which measurements take part (newest at or before t_old with
    its time moved to t_old, up to the newest before t_new, which counts up to t_new), the order of the product, and
    the cases where the reference gives up."""
    d = tmp_path / "mav0" / "imu0"
    d.mkdir(parents=True)
    rng = np.random.RandomState(3)
    ts = 1403636579_000000000 + (np.arange(400) * 5_000_000).astype(np.int64)      # 200 Hz
    w = rng.normal(scale=0.4, size=(400, 3)); acc = rng.normal(size=(400, 3))
    with open(d / "data.csv", "w") as f:
        f.write("#timestamp [ns],w_RS_S_x [rad s^-1],w_RS_S_y [rad s^-1],w_RS_S_z [rad s^-1],a_RS_S_x [m s^-2],a_RS_S_y [m s^-2],a_RS_S_z [m s^-2]\n")
        for t, wi, ai in zip(ts, w, acc):
            f.write("%d,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g\n" % (t, wi[0], wi[1], wi[2], ai[0], ai[1], ai[2]))
    tsec = ts * 1e-9

    def qmul(a, b):
        return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                         a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])

    def qexp(v):
        th = np.linalg.norm(v)
        return np.array([1.0, 0, 0, 0]) if th < 1e-12 else np.concatenate([[np.cos(th / 2)], np.sin(th / 2) * v / th])

    def expected(t_old, t_new, bias, max_dt):
        older = np.nonzero(tsec <= t_old)[0]; before = np.nonzero(tsec < t_new)[0]
        if len(older) == 0 or len(before) == 0 or older[-1] == before[-1] or t_new - tsec[before[-1]] > max_dt:
            return None
        i1, i2 = older[-1], before[-1]
        q = np.array([1.0, 0, 0, 0])
        for j in range(i1, i2 + 1):
            tj = t_old if j == i1 else tsec[j]
            dt = (t_new - tsec[j]) if j == i2 else (tsec[j + 1] - tj)
            q = qmul(q, qexp((w[j] - bias) * dt))
        return q

    cases = [(tsec[10] + 0.0012, tsec[14] + 0.0031, (0, 0, 0), 0.01),          # between measurements, 5 take part
             (tsec[20], tsec[30], (0.01, -0.02, 0.005), 0.01),                   # on measurement times: [t_old] included, [t_new] not
             (tsec[50] + 0.001, tsec[50] + 0.004, (0, 0, 0), 0.01),              # one measurement only -> false (it1 == it2)
             (tsec[0] - 0.01, tsec[5], (0, 0, 0), 0.01),                         # nothing at or before t_old -> false
             (tsec[390], tsec[399] + 0.05, (0, 0, 0), 0.01),                     # newest measurement too old -> false
             (tsec[100] + 0.002, tsec[110] + 0.002, (0, 0, 0), 0.01)]            # a 50 ms frame interval
    for t_old, t_new, bias, max_dt in cases:
        rc, out, raw = run("imu", str(tmp_path), "%.17g" % t_old, "%.17g" % t_new, "%.17g" % max_dt, *["%.17g" % b for b in bias])
        assert rc == 0, raw
        assert int(out["n"]) == 400
        want = expected(t_old, t_new, np.array(bias), max_dt)
        q = np.array([float(x) for x in out["q"].split()])
        if want is None:
            assert out["ok"] == "0" and np.array_equal(q, [1, 0, 0, 0])
        else:
            assert out["ok"] == "1" and np.abs(q - want).max() < 1e-14 and abs(np.linalg.norm(q) - 1) < 1e-14
    # no imu folder: empty, and no prior
    rc, out, raw = run("imu", str(tmp_path / "nowhere"), "0", "1", "0.01")
    assert rc == 0 and out["n"] == "0" and out["ok"] == "0"
