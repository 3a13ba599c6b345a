"""SURVEY.md Appendix B, gotcha 1: the bounds tests of the path differ from stage to stage.  Exact boundary values
(one step inside / one step outside) for the two integer-exact ones, on the oracle; tests/test_boundaries_gpu.py
repeats them through the C ABI."""
import ctypes as C

import numpy as np

from svo_pro_universal_amd import _capi as capi

import helpers

# 640x480, level 4 is 40x30.  extractFeaturesSubset (sparse_img_align.cpp:213-229): u_tl = px/16 - 2.5 floored;
# kept iff 0 <= u_tl_i and u_tl_i + 6 < 40 - 2, same for v with 30 rows
A3_PX = np.array([[39.99, 200.0], [40.0, 200.0], [551.99, 200.0], [552.0, 200.0],
                  [300.0, 39.99], [300.0, 40.0], [300.0, 391.99], [300.0, 392.0]])
A3_KEPT = [False, True, True, False, False, True, True, False]

# alignPyr2D at level 0 with a 16x16 patch (feature_alignment.cpp:789-797, 862-868): template corner px-8 must be in
# [1, width-17); the current corner floor(u) must be in [0, width-16)
KLT_REF_X = [8, 9, 640 - 10, 640 - 9]
KLT_REF_OK = [False, True, True, False]


def boundary_scene(n=A3_PX.shape[0]):
    sc = helpers.small_scene(61, n=n)
    sc.px = A3_PX.ravel().copy()
    return sc


def test_a3_selection_boundaries(oracle_lib):
    orc = oracle_lib
    sc = boundary_scene()
    ref, cur = helpers.scene_pyramids(orc, sc)
    pb = orc.problem_from_scenes([(sc, ref, cur)])
    idx = np.zeros(sc.n_features, np.int32)
    n = orc.load().orc_extract_features_subset(C.byref(pb.c.cams[0]), 4, 6, idx.ctypes.data)
    assert sorted(idx[:n]) == [i for i, k in enumerate(A3_KEPT) if k]


def klt_boundary_tracks():
    px_ref = np.array([[x, 240] for x in KLT_REF_X] + [[240, y] for y in (8, 9, 480 - 10, 480 - 9)], np.int32)
    px_cur = px_ref.astype(np.float64)
    # current position: corner at exactly width-16 (outside) and a hair inside
    extra_ref = np.array([[320, 240], [320, 240]], np.int32)
    extra_cur = np.array([[640 - 16 + 8.0, 240.0], [640 - 16 + 8.0 - 0.01, 240.0]])
    return np.concatenate([px_ref, extra_ref]).ravel(), np.concatenate([px_cur, extra_cur]).ravel()


def test_klt_template_and_current_boundaries(oracle_lib):
    orc = oracle_lib
    sc = helpers.small_scene(62, n=10)
    ref = orc.create_img_pyramid(sc.img_ref, 5)
    opt = capi.default_klt_options(max_level=0, min_level=0)
    px_ref, px_cur = klt_boundary_tracks()
    n = px_ref.size // 2
    # same image on both sides: a track that is allowed to run converges where it starts
    po, so = orc.klt_track_batch(opt, [ref] * n, ref, px_ref, px_cur)
    assert list(so[:4]) == [int(k) for k in KLT_REF_OK]
    assert list(so[4:8]) == [int(k) for k in KLT_REF_OK]
    assert so[8] == 0            # the current patch would touch column width: left alone, reported lost
    assert np.array_equal(po.reshape(-1, 2)[[1, 2, 5, 6]], px_cur.reshape(-1, 2)[[1, 2, 5, 6]])
