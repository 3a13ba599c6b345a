"""FAST stage of SURVEY.md 8(f-2) on the DEVICE against the reference's own code: csrc/detector.hip through
svoh_detect_features vs tests/golden/fast_ref.npz (outputs of fast_corner_detect_10_sse2 / fast_corner_score_10 /
fast_nonmax_3x3 compiled from the reference; see tests/test_fast_ref_cpu.py).  No oracle call in this file.

How the dense device detector is asked for the reference's survivor list: with cell_size = 1 every pixel is its own grid
cell, so fd_utils::fastDetector's per-cell best (feature_detection_utils.cpp:178-192) keeps every 3x3 survivor whose
score is > the threshold (the grid's corners start at score = threshold, feature_detection.cpp:121-123) and that lies
inside the border (3 = the FAST ring: no survivor is nearer).  Bar: the set of (x, y, score) equal, exactly."""
import os

import numpy as np
import pytest

import helpers
from svo_pro_universal_amd import _capi as capi
from test_fast_ref_cpu import FIX, fixture_cases

pytestmark = pytest.mark.gpu


def as_set(px, score):
    a = np.concatenate([np.asarray(px, np.int64).reshape(-1, 2), np.asarray(score, np.int64).reshape(-1, 1)], axis=1)
    return a[np.lexsort((a[:, 0], a[:, 1]))]


def test_device_survivors_equal_the_reference_fixture(gpu_ctx):
    z = np.load(FIX)
    n_cases = n_feat = 0
    for name, img, thr, key in fixture_cases(z):
        h, w = img.shape
        if h < 7 or w < 7:
            continue
        sv = z["sv_" + key].astype(np.int64)
        want = sv[sv[:, 2] > thr]
        fr = gpu_ctx.build_pyramid(np.ascontiguousarray(img), 1)
        opt = capi.default_detector_options(cell_size=1, min_level=0, max_level=0, border=3, detect_edgelets=0,
                                            threshold_primary=float(thr))
        d = gpu_ctx.detect_features(opt, fr, w, h)
        gpu_ctx.release_frame(fr)
        got = as_set(d["px"], d["score"])
        want = want[np.lexsort((want[:, 0], want[:, 1]))]
        assert np.array_equal(got, want), (key, len(got), len(want))
        assert (d["level"] == 0).all() and (d["type"] == capi.FT_CORNER).all()
        n_cases += 1
        n_feat += len(want)
    assert n_cases >= 70 and n_feat > 50000


@pytest.mark.parametrize("stem, shape", [("s640", (640, 480)), ("refjpg", (752, 480))])
@pytest.mark.parametrize("thr", [10, 20])
def test_device_fast_detector_at_its_defaults_from_reference_survivors(gpu_ctx, stem, shape, thr):
    """FastDetector::detect at the reference's settings (30-pixel cells, levels 0..2, border 8): the device's features
    against fd_utils::fastDetector's grid step (feature_detection_utils.cpp:168-192, restated here in ten lines) applied
    to the REFERENCE's survivors of the three levels.  The device builds levels 1, 2 itself (a-0, bit-exact rows)."""
    z = np.load(FIX)
    w, h = shape
    cell, border = 30, 8
    n_cols, n_rows = int(np.ceil(w / cell)), int(np.ceil(h / cell))
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
    fr = gpu_ctx.build_pyramid(z["img_%s_l0" % stem], 3)
    opt = capi.default_detector_options(cell_size=cell, detect_edgelets=0, threshold_primary=float(thr))
    d = gpu_ctx.detect_features(opt, fr, w, h)
    gpu_ctx.release_frame(fr)
    got = np.concatenate([d["px"].astype(np.int64), d["score"].astype(np.int64)[:, None], d["level"].astype(np.int64)[:, None]], axis=1)
    got = got[np.lexsort((got[:, 0], got[:, 1]))]
    assert len(want) > 100 and np.array_equal(got, want), (len(got), len(want))
    # fillFeatures: sorted by score, best first
    assert (np.diff(d["score"]) <= 0).all()
