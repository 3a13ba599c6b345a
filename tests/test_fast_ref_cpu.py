"""FAST stage of SURVEY.md 8(f-2) PINNED against the reference's own code: tests/golden/fast_ref.npz holds the outputs
of fast::fast_corner_detect_10_sse2 / fast_corner_score_10 / fast_nonmax_3x3 compiled from /root/reference
(oracle/ref_fast/Makefile -> oracle/_ref/libfast_ref.so; generator tests/golden/make_golden_fast_ref.py).  Here: the
oracle's definitional restatement (oracle/svo_oracle_detector.c:36-116) and the NumPy second reading against that
fixture, bit for bit; and, where the reference tree exists (the build container), the fixture regenerated from a fresh
build of the reference equals the committed one.  The GPU side is tests/test_fast_ref_gpu.py."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers

FIX = os.path.join(os.path.dirname(helpers.GOLDEN), "fast_ref.npz")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, np.int32).tobytes()).hexdigest()


def fixture_cases(z):
    for name in z["names"]:
        name = str(name)
        img = z["img_" + name]
        if "roi_" + name in z.files:
            x0, y0, w, h = z["roi_" + name]
            img = img[y0:y0 + h, x0:x0 + w]
        for thr in z["thresholds"]:
            yield name, img, int(thr), "%s_t%d" % (name, int(thr))


def check_against_fixture(z, key, xy, sc, nm):
    n = z["n_" + key]
    assert (len(sc), len(nm)) == (int(n[0]), int(n[1])), (key, len(sc), len(nm), n)
    if "xy_" + key in z.files:   # small case: the lists themselves, for a readable failure
        assert np.array_equal(xy, z["xy_" + key]) and np.array_equal(sc, z["sc_" + key]) and np.array_equal(nm, z["nm_" + key]), key
    assert [sha(xy), sha(sc), sha(nm)] == [str(s) for s in z["sha_" + key]], key
    sv = np.concatenate([xy[nm], sc[nm, None]], axis=1).reshape(-1, 3)
    assert np.array_equal(sv, z["sv_" + key]), key


def test_oracle_fast_equals_the_reference_fixture(oracle_lib):
    z = np.load(FIX)
    n_cases = n_corners = 0
    for name, img, thr, key in fixture_cases(z):
        xy, sc, nm = oracle_lib.fast_corners(img, thr)
        check_against_fixture(z, key, xy, sc, nm)
        n_cases += 1
        n_corners += len(sc)
    assert n_cases == 80 and n_corners > 300000


def test_numpy_second_reading_equals_the_reference_fixture():
    """the dense NumPy reading (tests/np_restatement_detector.py) on the fixture's smaller images: corner set, scores"""
    import np_restatement_detector as npd
    z = np.load(FIX)
    done = 0
    for name, img, thr, key in fixture_cases(z):
        if img.size > 131 * 97 or img.shape[0] < 7 or img.shape[1] < 7:
            continue
        S = npd.fast_score_image(np.ascontiguousarray(img))        # largest barrier at which a pixel is a corner
        ys, xs = np.nonzero(S >= thr)
        xy = np.stack([xs, ys], 1).astype(np.int32)
        assert len(xy) == int(z["n_" + key][0]), key
        if "xy_" + key in z.files:
            assert np.array_equal(xy, z["xy_" + key]) and np.array_equal(S[ys, xs], z["sc_" + key]), key
        sv_np = np.array(npd.fast_corners(np.ascontiguousarray(img), thr), np.int16).reshape(-1, 3)
        assert np.array_equal(sv_np, z["sv_" + key]), key
        done += 1
    assert done >= 20


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/fast_neon"), reason="the reference tree exists in the build container only")
def test_fixture_is_what_a_fresh_reference_build_gives(tmp_path):
    """rebuild oracle/_ref from the reference's sources and regenerate: equal to the committed fixture, array by array"""
    subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle", "ref_fast")])
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden_fast_ref.py"), "--check"], text=True)
    assert "reproduced" in out


def test_prebuilt_reference_library_against_fixture_when_present(oracle_lib):
    """oracle/_ref/libfast_ref.so travels prebuilt (it is not in .gpurunignore): where it is present WITHOUT the
    reference tree (the GPU box), it must still give the fixture -- it is the checker of the GPU test beside this one."""
    if not oracle_lib.ref_fast_available(build_if_possible=True):
        pytest.skip("no oracle/_ref/libfast_ref.so and no reference tree to build it from")
    z = np.load(FIX)
    for name, img, thr, key in fixture_cases(z):
        if name.startswith("s640_l0") or name.startswith("refjpg_l0"):
            xy, sc, nm = oracle_lib.ref_fast_corners(img, thr)
            check_against_fixture(z, key, xy, sc, nm)
            if img.shape[1] >= 22:   # the plain decision tree of the same file set finds the same corners as the SSE2 path
                assert np.array_equal(oracle_lib.ref_fast_corners_plain(img, thr), xy)


def test_oracle_detector_from_reference_survivors(oracle_lib):
    """the oracle's whole FastDetector::detect (orc_detect_features) against the REFERENCE's survivors: (a) one cell per
    pixel = the survivor list itself, (b) the reference's settings = fd_utils::fastDetector's grid step
    (feature_detection_utils.cpp:168-192, restated in the test) on the survivors of levels 0..2.  The GPU twin of this
    test is tests/test_fast_ref_gpu.py."""
    from svo_pro_universal_amd import _capi as capi
    z = np.load(FIX)
    for name, img, thr, key in fixture_cases(z):
        if min(img.shape) < 7 or img.size > 200 * 200:
            continue
        opt = capi.default_detector_options(cell_size=1, min_level=0, max_level=0, border=3, detect_edgelets=0, threshold_primary=float(thr))
        d = oracle_lib.detect_features(opt, [np.ascontiguousarray(img)])
        got = np.concatenate([d["px"].astype(np.int64), d["score"].astype(np.int64)[:, None]], axis=1).reshape(-1, 3)
        got = got[np.lexsort((got[:, 0], got[:, 1]))]
        sv = z["sv_" + key].astype(np.int64)
        want = sv[sv[:, 2] > thr]
        assert np.array_equal(got, want[np.lexsort((want[:, 0], want[:, 1]))]), key
    for stem, (w, h) in (("s640", (640, 480)), ("refjpg", (752, 480))):
        for thr in (10, 20):
            cell, border = 30, 8
            n_cols = int(np.ceil(w / cell))
            best = {}
            for level in range(3):
                scale = 1 << level
                lw, lh = z["img_%s_l%d" % (stem, level)].shape[::-1]
                for x, y, s in z["sv_%s_l%d_t%d" % (stem, level, thr)].astype(np.int64):
                    if x < border or y < border or x >= lw - border or y >= lh - border:
                        continue
                    k = int(np.floor(y * scale / cell) * n_cols + np.floor(x * scale / cell))
                    if s > best.get(k, (0, 0, thr, 0))[2]:
                        best[k] = (x * scale, y * scale, s, level)
            want = np.array(sorted(best.values(), key=lambda c: (c[1], c[0])), np.int64).reshape(-1, 4)
            levels = [z["img_%s_l%d" % (stem, lv)] for lv in range(3)]
            d = oracle_lib.detect_features(capi.default_detector_options(cell_size=cell, detect_edgelets=0, threshold_primary=float(thr)), levels)
            got = np.concatenate([d["px"].astype(np.int64), d["score"].astype(np.int64)[:, None], d["level"].astype(np.int64)[:, None]], axis=1)
            assert len(want) > 100 and np.array_equal(got[np.lexsort((got[:, 0], got[:, 1]))], want), (stem, thr)
