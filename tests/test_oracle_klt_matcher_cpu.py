"""CPU tests of the oracle's KLT / warp / ZMSSD / align / matcher / depth-filter
restatement: hand-computed known answers for the integer rules, NumPy cross-checks,
and ground-truth properties on rendered scenes.  PARITY UNPINNED (no reference
vectors exist for these functions)."""
import ctypes as C

import numpy as np
import pytest

from svo_pro_universal_amd import _capi as capi, synth


def _img(orc, arr):
    arr = np.ascontiguousarray(arr, np.uint8)
    im = orc.orc_image()
    im.data, im.width, im.height, im.pitch = arr.ctypes.data, arr.shape[1], arr.shape[0], arr.strides[0]
    return im, arr


def test_zmssd_known_answers(oracle_lib):
    orc = oracle_lib
    lib = orc.load(); orc._bind_part2(lib)
    rng = np.random.RandomState(0)
    a = rng.randint(0, 256, 64).astype(np.uint8)
    img = rng.randint(0, 256, (20, 24)).astype(np.uint8)
    b = img[3:11, 5:13].astype(np.int64).ravel()
    A = a.astype(np.int64)
    # patch_score.h:283: sumAA - 2 sumAB + sumBB - (sumA^2 - 2 sumA sumB + sumB^2)/64, integer division
    exp = int((A * A).sum() - 2 * (A * b).sum() + (b * b).sum() - ((A.sum() - b.sum()) ** 2) // 64)
    got = lib.orc_zmssd_score(a.ctypes.data, img.ctypes.data + 3 * 24 + 5, 24)
    assert got == exp
    # identical patch -> 0; constant offset -> 0 (zero mean); threshold 2000*64
    assert lib.orc_zmssd_score(a.ctypes.data, a.ctypes.data, 8) == 0
    c = np.clip(a.astype(int) // 2 + 7, 0, 255).astype(np.uint8); d = (c + 20).astype(np.uint8)
    assert lib.orc_zmssd_score(c.ctypes.data, d.ctypes.data, 8) == 0


def test_warp_affine_identity_and_truncation(oracle_lib):
    orc = oracle_lib
    lib = orc.load(); orc._bind_part2(lib)
    rng = np.random.RandomState(1)
    im, arr = _img(orc, rng.randint(0, 256, (60, 80)))
    A = np.array([1.0, 0.0, 0.0, 1.0])  # col-major identity
    px = np.array([40.0, 30.0])
    patch = np.zeros(100, np.uint8)
    assert lib.orc_warp_affine(A.ctypes.data, C.byref(im), px.ctypes.data, 0, 0, 5, patch.ctypes.data) == 1
    assert np.array_equal(patch.reshape(10, 10), arr[25:35, 35:45])
    # half-pixel shift: truncating cast of the float bilinear value (patch_warp.cpp:151, SURVEY gotcha 6)
    px = np.array([40.5, 30.0])
    lib.orc_warp_affine(A.ctypes.data, C.byref(im), px.ctypes.data, 0, 0, 5, patch.ctypes.data)
    a32 = arr.astype(np.float32)
    exp = (np.float32(0.5) * a32[25:35, 35:45] + np.float32(0.5) * a32[25:35, 36:46]).astype(np.uint8)
    assert np.array_equal(patch.reshape(10, 10), exp)
    # out of image -> fail
    px = np.array([2.0, 2.0])
    assert lib.orc_warp_affine(A.ctypes.data, C.byref(im), px.ctypes.data, 0, 0, 5, patch.ctypes.data) == 0
    # search level 1 doubles the sampling step in the reference image
    px = np.array([40.0, 30.0])
    lib.orc_warp_affine(A.ctypes.data, C.byref(im), px.ctypes.data, 0, 1, 5, patch.ctypes.data)
    assert np.array_equal(patch.reshape(10, 10), arr[20:40:2, 30:50:2])
    # best search level: det 1 -> 0, det 16 -> 2 (16 > 3 -> 4 > 3 -> 1), capped by max_level
    assert lib.orc_get_best_search_level(np.array([1.0, 0, 0, 1.0]).ctypes.data, 4) == 0
    assert lib.orc_get_best_search_level(np.array([4.0, 0, 0, 4.0]).ctypes.data, 4) == 2
    assert lib.orc_get_best_search_level(np.array([40.0, 0, 0, 40.0]).ctypes.data, 2) == 2


def test_warp_matrix_matches_finite_differences(oracle_lib):
    orc = oracle_lib
    lib = orc.load(); orc._bind_part2(lib)
    sc = synth.make_align_scene(5, n_features=5, render_images=False, rot_deg=(1, 2), trans_m=(0.1, 0.2))
    cam = orc.to_camera(sc.cam)
    T = orc.to_se3(sc.T_cur_f_w_gt * sc.T_ref_f_w.inverse())
    px = sc.px[:2].copy(); f = sc.f[:3].copy()
    A = np.zeros(4)
    lib.orc_get_warp_matrix_affine(C.byref(cam), C.byref(cam), px.ctypes.data, f.ctypes.data, float(sc.depth[0]), C.byref(T), 1, A.ctypes.data)
    # numpy: project the reference pixel and pixels 5*2^level away at the plane depth z
    Tc = sc.T_cur_f_w_gt * sc.T_ref_f_w.inverse()
    def proj(p2):
        x, y = sc.cam.undistorted_xy(p2[0], p2[1])
        X = np.array([x, y, 1.0]) * (f[2] * sc.depth[0])
        return sc.cam.project(Tc.transform(X)[:, None])[:, 0]
    c = sc.cam.project(Tc.transform(f * sc.depth[0])[:, None])[:, 0]
    exp = np.concatenate([(proj(px + [10, 0]) - c) / 5, (proj(px + [0, 10]) - c) / 5])
    assert np.abs(A - exp).max() < 1e-9


def test_align2d_recovers_translation(oracle_lib):
    orc = oracle_lib
    lib = orc.load(); orc._bind_part2(lib)
    sc = synth.make_align_scene(6, n_features=5)
    img = sc.img_ref
    im, arr = _img(orc, img)
    for (cx, cy, dx, dy) in ((200, 150, 0.8, -0.6), (400, 300, -1.3, 0.4)):
        pwb = np.ascontiguousarray(img[cy - 5:cy + 5, cx - 5:cx + 5]).ravel().copy()
        patch = np.ascontiguousarray(img[cy - 4:cy + 4, cx - 4:cx + 4]).ravel().copy()
        p = np.array([cx + dx, cy + dy])
        assert lib.orc_align_2d(C.byref(im), pwb.ctypes.data, patch.ctypes.data, 10, 1, 0, p.ctypes.data) == 1
        assert np.abs(p - [cx, cy]).max() < 0.05
        # 1-D alignment along the displacement direction
        d = np.array([dx, dy]) / np.hypot(dx, dy)
        p = np.array([cx + dx, cy + dy]); hinv = C.c_double()
        assert lib.orc_align_1d(C.byref(im), d.ctypes.data, pwb.ctypes.data, patch.ctypes.data, 10, 1, 0, p.ctypes.data, C.byref(hinv)) == 1
        assert np.abs(p - [cx, cy]).max() < 0.08 and hinv.value > 0


def test_klt_tracks_to_ground_truth_and_level_skipping(oracle_lib):
    orc = oracle_lib
    sc = synth.make_align_scene(7, n_features=10, rot_deg=(0.5, 1.5), trans_m=(0.05, 0.15))
    ref = orc.create_img_pyramid(sc.img_ref, 5); cur = orc.create_img_pyramid(sc.img_cur, 5)
    tr = synth.make_track_set(sc, 300)
    out, status = orc.klt_track_batch(capi.default_klt_options(), ref, cur, tr["px_ref"], tr["px_cur_init"])
    ok = status == 1
    err = np.linalg.norm((out - tr["px_true"]).reshape(-1, 2), axis=1)
    assert ok.mean() > 0.95 and np.median(err[ok]) < 0.15 and err[ok].max() < 1.0  # affine patch change limits a translation-only KLT
    # a reference pixel 20 px from the border: levels 4 and 3 are skipped (ref patch within 1 px of the
    # border, feature_alignment.cpp:802-809), tracking still succeeds on the finer levels
    pr = np.array([20, 20], np.int32)
    p, s = orc.klt_track_batch(capi.default_klt_options(), ref, cur, pr, np.array([20.0, 20.0]))
    p2, s2 = orc.klt_track_batch(capi.default_klt_options(max_level=2), ref, cur, pr, np.array([20.0, 20.0]))
    assert np.array_equal(p, p2) and s[0] == s2[0]


def test_depth_filter_converges_to_true_depth(oracle_lib):
    orc = oracle_lib
    sc = synth.make_align_scene(8, n_features=10, rot_deg=(0.2, 0.6), trans_m=(0.3, 0.4))  # wide baseline: informative
    ref = orc.create_img_pyramid(sc.img_ref, 5); cur = orc.create_img_pyramid(sc.img_cur, 5)
    sd = synth.make_seed_set(sc, 1500, edgelet_fraction=0.0, margin=60)
    mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(sc.cam)
    rv = orc.make_frame_view(ref, sc.cam, sc.T_ref_f_w, sd["mu_range"], 1)
    cv = orc.make_frame_view(cur, sc.cam, sc.T_cur_f_w_gt, 0.0, 2)
    fb, keep = orc.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
    state = sd["state"]
    sig0 = state.reshape(-1, 4)[:, 1].copy()
    for _ in range(8):
        ns, state, succ, mr = orc.update_seeds_batch(mopt, dopt, [rv], cv, fb, state)
    st = state.reshape(-1, 4)
    rel = np.abs(1 / st[:, 0] - sd["true_depth"]) / sd["true_depth"]
    good = succ == 1
    rel0 = np.abs(1 / sd["state"].reshape(-1, 4)[:, 0] - sd["true_depth"]) / sd["true_depth"]
    # accuracy is bounded by the one-pixel disparity error (tau): require a clear improvement, not a number
    assert good.mean() > 0.8 and np.median(rel[good]) < 0.5 * np.median(rel0[good])
    assert np.median(st[good, 1] / sig0[good]) < 0.2           # variance shrinks
    z = np.abs(st[good, 0] - 1 / sd["true_depth"][good]) / np.sqrt(st[good, 1])
    assert np.median(z) < 3.0                                   # the filter's own sigma is honest
    # a seed that cannot be matched only gets b += 1 (depth_filter.cpp:445-456)
    bad = np.nonzero((mr != 0) & (mr != 7) & (mr != 100))[0]
    if bad.size:
        assert np.all(st[bad, 3] > 10.0)


def test_vogiatzis_update_formula_against_numpy(oracle_lib):
    """One Vogiatzis update reproduced with plain numpy from the paper's moment matching."""
    orc = oracle_lib
    sc = synth.make_align_scene(9, n_features=10, rot_deg=(0.5, 1.0), trans_m=(0.1, 0.15))
    ref = orc.create_img_pyramid(sc.img_ref, 5); cur = orc.create_img_pyramid(sc.img_cur, 5)
    sd = synth.make_seed_set(sc, 50, edgelet_fraction=0.0)
    mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(sc.cam)
    rv = orc.make_frame_view(ref, sc.cam, sc.T_ref_f_w, sd["mu_range"], 1)
    cv = orc.make_frame_view(cur, sc.cam, sc.T_cur_f_w_gt, 0.0, 2)
    fb, keep = orc.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
    ns, st1, succ, mr = orc.update_seeds_batch(mopt, dopt, [rv], cv, fb, sd["state"])
    st0 = sd["state"].reshape(-1, 4); st1 = st1.reshape(-1, 4)
    for i in np.nonzero(succ)[0][:10]:
        mu, s2, a, b = st0[i]
        mu1, s21, a1, b1 = st1[i]
        # invert the update for the measurement: f and e determine (a', b'); check the Beta moments
        f = a1 / (a1 + b1)
        e = a1 * (a1 + 1) / ((a1 + b1) * (a1 + b1 + 1))
        assert 0 < f < 1 and 0 < e < f
        # the posterior mean lies between prior mean and the measurement-fused mean; variance decreased
        assert s21 <= s2 * (1 + 1e-9)


def test_klt_oracle_against_an_independent_numpy_restatement(oracle_lib):
    """Two restatements of feature_alignment.cpp:761-973 written separately (C in oracle/, numpy float32 in
    tests/np_restatement.py) must agree bit for bit: tracks inside the image, at the border, starting far off."""
    import np_restatement as npr
    orc = oracle_lib
    sc = synth.make_align_scene(77, n_features=10, rot_deg=(0.5, 1.5), trans_m=(0.05, 0.15))
    ref = orc.create_img_pyramid(sc.img_ref, 5)
    cur = orc.create_img_pyramid(sc.img_cur, 5)
    tr = synth.make_track_set(sc, 40, seed=5)
    px_ref = tr["px_ref"].copy().reshape(-1, 2)
    px0 = tr["px_cur_init"].copy().reshape(-1, 2)
    px_ref[0] = [9, 200]; px_ref[1] = [640 - 10, 240]; px_ref[2] = [300, 9]      # template bounds of the coarse levels
    px0[3] += 60.0; px0[4] = [2.0, 3.0]; px0[5] = [636.0, 476.0]                  # far off / at the borders
    opt = capi.default_klt_options()
    po, so = orc.klt_track_batch(opt, [ref] * 40, cur, px_ref.ravel(), px0.ravel())
    sizes = [opt.patch_sizes[k] for k in range(5)]
    n_conv = 0
    for i in range(40):
        ok, p = npr.align_pyr_2d(ref, cur, opt.max_level, opt.min_level, sizes, opt.max_iter, opt.min_update_squared,
                                 [int(px_ref[i, 0]), int(px_ref[i, 1])], px0[i])
        assert int(ok) == int(so[i]), i
        assert p[0] == po[2 * i] and p[1] == po[2 * i + 1], (i, p, po[2 * i:2 * i + 2])
        n_conv += int(ok)
    assert 25 <= n_conv < 40
