"""The C oracle against a SECOND, independently written reading of the reference (tests/np_restatement_direct.py, NumPy,
written from the reference files without consulting oracle/*.c): matcher, warp, ZMSSD, align1D/2D, both epipolar scans,
updateSeed + Vogiatzis + computeTau on thousands of units.  A line of matcher.cpp / depth_filter.cpp misread by the
author of the oracle AND of the kernels would be in both of those; it is unlikely to be in this file as well.
Parity with the reference binary stays unpinned: neither reading can be run against it."""
import numpy as np
import pytest

from svo_pro_universal_amd import _capi as capi, synth
import np_restatement_direct as nd
import np_restatement_detector as ndet


def _views(orc, sc, sd, cam_kind):
    ref = orc.create_img_pyramid(sc.img_ref, 5); cur = orc.create_img_pyramid(sc.img_cur, 5)
    rv = orc.make_frame_view(ref, sc.cam, sc.T_ref_f_w, sd["mu_range"], 1)
    cv = orc.make_frame_view(cur, sc.cam, sc.T_cur_f_w_gt, 0.0, 2)
    ncam = nd.Cam.of(sc.cam)
    nrv = nd.FrameView(ref, ncam, nd.Tf.from7(sc.T_ref_f_w.as7()), 1, sd["mu_range"])
    ncv = nd.FrameView(cur, ncam, nd.Tf.from7(sc.T_cur_f_w_gt.as7()), 2, 0.0)
    return rv, cv, nrv, ncv


def _nd_options(mopt):
    return nd.MatcherOptions(align_max_iter=mopt.align_max_iter, max_epi_search_steps=mopt.max_epi_search_steps,
                             subpix_refinement=bool(mopt.subpix_refinement),
                             epi_search_edgelet_filtering=bool(mopt.epi_search_edgelet_filtering),
                             scan_on_unit_sphere=bool(mopt.scan_on_unit_sphere),
                             epi_search_edgelet_max_angle=mopt.epi_search_edgelet_max_angle,
                             affine_est_offset=bool(mopt.affine_est_offset), affine_est_gain=bool(mopt.affine_est_gain),
                             max_patch_diff_ratio=mopt.max_patch_diff_ratio)


@pytest.mark.parametrize("cam_kind,mkw,n", [
    ("pinhole", dict(scan_on_unit_sphere=0), 2000),                       # depth filter default: unit plane
    ("radtan", dict(scan_on_unit_sphere=1), 700),                         # Matcher default: unit sphere
    ("pinhole", dict(scan_on_unit_sphere=0, affine_est_gain=1), 500),     # illumination gain estimated too
])
def test_update_seeds_oracle_vs_numpy_second_opinion(oracle_lib, cam_kind, mkw, n):
    orc = oracle_lib
    cam = synth.Camera.test_camera() if cam_kind == "pinhole" else synth.Camera.euroc_like()
    sc = synth.make_align_scene(63, n_features=10, cam=cam, rot_deg=(0.5, 1.5), trans_m=(0.08, 0.2))
    sd = synth.make_seed_set(sc, n, margin=3, levels=(0, 1, 2, 3))
    rv, cv, nrv, ncv = _views(orc, sc, sd, cam_kind)
    mopt, dopt = capi.default_matcher_options(**mkw), capi.default_depth_filter_options(sc.cam)
    state = sd["state"].copy(); types = sd["type"].copy()
    for rnd in range(2):          # second round: updated states and types fed back
        fb, keep = orc.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], types)
        ns, st_o, succ_o, mr_o = orc.update_seeds_batch(mopt, dopt, [rv], cv, fb, state)
        got = nd.update_seeds(ncv, [nrv], sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], types, state,
                              _nd_options(mopt), dopt.seed_convergence_sigma2_thresh, dopt.mappoint_convergence_sigma2_thresh,
                              dopt.px_error_angle, bool(dopt.check_visibility), bool(dopt.check_convergence),
                              bool(dopt.use_vogiatzis_update))
        # integer outputs: match result codes, success flags, feature types -- exact
        assert np.array_equal(mr_o, got["match_result"]), np.nonzero(mr_o != got["match_result"])[0][:10]
        assert np.array_equal(succ_o, got["success"])
        assert np.array_equal(keep["type"], got["type"])
        assert ns == int(got["success"].sum())
        # the seed state: doubles computed from a float32 sub-pixel position.  Observed: ~95 % of the seeds agree to 1e-12
        # (the float32 alignment paths of the two readings are bit-identical), all to 2e-10 -- except with the
        # illumination gain estimated too, where the 4x4 system is ill-conditioned in alpha and a last-bit difference of
        # the restated Eigen inverse / product order moves 3 of 500 seeds by up to 1.4e-6 (Eigen sensitivity)
        a, b = st_o.reshape(-1, 4), got["state"].reshape(-1, 4)
        rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-300)
        tol = 1e-5 if mkw.get("affine_est_gain") else 1e-8
        assert rel.max() <= tol, (rel.max(), np.unravel_index(rel.argmax(), rel.shape))
        assert np.mean(rel.max(axis=1) <= 1e-12) > 0.9
        assert len(set(mr_o.tolist())) >= 4 and succ_o.sum() > 0.5 * n        # successes and several failure kinds
        state, types = st_o, keep["type"].copy()


@pytest.mark.parametrize("cam_kind", ["pinhole", "radtan"])
def test_match_direct_oracle_vs_numpy_second_opinion(oracle_lib, cam_kind):
    orc = oracle_lib
    cam = synth.Camera.test_camera() if cam_kind == "pinhole" else synth.Camera.euroc_like()
    sc = synth.make_align_scene(64, n_features=10, cam=cam, rot_deg=(0.5, 1.5), trans_m=(0.05, 0.15))
    n = 2000 if cam_kind == "pinhole" else 600
    sd = synth.make_seed_set(sc, n, margin=3, levels=(0, 1, 2, 3))
    rv, cv, nrv, ncv = _views(orc, sc, sd, cam_kind)
    x = sd["f"].reshape(-1, 3).T * sd["true_depth"]
    px_true = sc.cam.project(sc.T_w_cur.inverse().transform(sc.T_w_ref.transform(x)))
    px_init = np.ascontiguousarray((px_true + np.random.RandomState(1).uniform(-2.0, 2.0, px_true.shape)).T).ravel()
    px_init[:20] += 40.0
    ftype = np.where(sd["type"] == 0, capi.FT_EDGELET, capi.FT_CORNER)
    for mkw in (dict(), dict(affine_est_gain=1)):
        mopt = capi.default_matcher_options(**mkw)
        fb, keep = orc.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], ftype)
        oo = orc.match_direct_batch(mopt, [rv], cv, fb, sd["true_depth"], px_init)
        gg = nd.match_direct_batch(ncv, [nrv], sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], ftype,
                                   sd["true_depth"], px_init, _nd_options(mopt))
        assert np.array_equal(oo["result"], gg["result"]), np.nonzero(oo["result"] != gg["result"])[0][:10]
        ran = oo["result"] != nd.FAIL_VISIBILITY
        assert np.array_equal(oo["search_level"][ran], gg["search_level"][ran])
        ok = oo["result"] == 0
        assert ok.sum() > 0.5 * n and len(set(oo["result"].tolist())) >= 3
        # float32 sub-pixel positions: a few ulp of a float at 640 px (6e-5)
        assert np.abs(oo["px_cur"] - gg["px_cur"]).max() <= 1e-4
        assert np.allclose(oo["A"][np.repeat(ran, 4)], gg["A"][np.repeat(ran, 4)], rtol=1e-11, atol=1e-13)
        assert np.abs(oo["f_cur"] - gg["f_cur"])[np.repeat(ok, 3)].max() < 1e-6
        edge_ok = ok & (ftype == capi.FT_EDGELET)
        assert np.allclose(oo["h_inv"][edge_ok], gg["h_inv"][edge_ok], rtol=1e-5)


@pytest.mark.parametrize("cam_kind", ["pinhole", "radtan"])
def test_pixelwise_warp_oracle_vs_numpy_second_opinion(oracle_lib, cam_kind):
    """Matcher::Options::use_affine_warp_ == false (matcher.cpp:67-81, patch_warp.cpp:158-230): nothing in the reference
    clears the flag, the branch is restated all the same.  Patches byte for byte, then the matches built on them."""
    orc = oracle_lib
    cam = synth.Camera.test_camera() if cam_kind == "pinhole" else synth.Camera.euroc_like()
    sc = synth.make_align_scene(66, n_features=10, cam=cam, rot_deg=(0.5, 1.5), trans_m=(0.05, 0.15))
    n = 300
    sd = synth.make_seed_set(sc, n, margin=3, levels=(0, 1, 2, 3))
    rv, cv, nrv, ncv = _views(orc, sc, sd, cam_kind)
    x = sd["f"].reshape(-1, 3).T * sd["true_depth"]
    lm = np.ascontiguousarray(sc.T_w_ref.transform(x).T)                 # landmark positions (world), n x 3
    px_true = sc.cam.project(sc.T_w_cur.inverse().transform(sc.T_w_ref.transform(x)))
    px_init = np.ascontiguousarray((px_true + np.random.RandomState(2).uniform(-2.0, 2.0, px_true.shape)).T).ravel()
    ftype = np.where(sd["type"] == 0, capi.FT_EDGELET, capi.FT_CORNER)
    n_patch = n_none = n_px_diff = 0
    for i in range(0, n, 3):
        lvl = int(sd["level"][i])
        for search_level in (lvl, min(lvl + 1, 4)):
            po = orc.warp_pixelwise(cv, rv, sd["px"][2 * i:2 * i + 2], lm[i], lvl, search_level)
            pn = nd.warp_pixelwise(ncv, nrv, sd["px"][2 * i:2 * i + 2], lm[i], lvl, search_level, 5)
            assert (po is None) == (pn is None)
            if po is None:
                n_none += 1
                continue
            n_patch += 1
            d = np.abs(po.astype(int) - pn.astype(int))
            assert d.max() <= 1          # a weight on a truncation edge may differ in the last bit of a norm
            n_px_diff += int((d != 0).sum())
    assert n_patch > 100 and n_px_diff <= n_patch      # at most one pixel in a hundred patches ...  (observed: none)
    mopt = capi.default_matcher_options()
    fb, keep = orc.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], ftype)
    oo = orc.match_direct_batch(mopt, [rv], cv, fb, sd["true_depth"], px_init, landmark_xyz=lm)
    gg = nd.match_direct_batch(ncv, [nrv], sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], ftype,
                               sd["true_depth"], px_init, _nd_options(mopt), landmark_xyz=lm)
    aff = orc.match_direct_batch(mopt, [rv], cv, fb, sd["true_depth"], px_init)
    assert np.array_equal(oo["result"], gg["result"])
    ok = oo["result"] == 0
    assert ok.sum() > 0.5 * n
    assert np.abs(oo["px_cur"] - gg["px_cur"]).max() <= 1e-4
    # the two warps agree on what matches where (same plane-induced motion), not bit for bit
    both = ok & (aff["result"] == 0)
    assert both.sum() > 0.45 * n
    dpx = np.abs(oo["px_cur"] - aff["px_cur"]).reshape(-1, 2)[both]
    assert np.median(dpx) < 0.1 and (dpx > 0).any()


@pytest.mark.parametrize("error_type", [capi.POSE_ERR_UNIT_PLANE, capi.POSE_ERR_BEARING_DIFF, capi.POSE_ERR_IMAGE_PLANE])
@pytest.mark.parametrize("n_cams,prior", [(1, False), (2, True)])
def test_pose_optimizer_oracle_vs_numpy_second_opinion(oracle_lib, error_type, n_cams, prior):
    """PoseOptimizer::run (f-3): MAD sigma, iteration count, outlier flags and counters exact, pose to 1e-9 between the C
    oracle and tests/np_restatement_pose.py (all three error types, corner and edgelet residuals, rotation prior)."""
    from svo_pro_universal_amd import frontend as fe
    import pose_helpers as ph
    import np_restatement_pose as npp
    for seed in (11, 12, 13):
        sc = ph.make_pose_scene(seed + n_cams, n=180, n_cams=n_cams)
        kw = dict(error_type=error_type)
        if prior:
            kw.update(have_rotation_prior=1, prior_lambda=0.3, R_prior=sc["T_imu_world_gt"].as7()[:4])
        opt = capi.default_pose_options(sc["cam"], **kw)
        pb, keep = fe.make_pose_problem(sc["cams"], sc["T_imu_world_init"])
        r = oracle_lib.optimize_pose(opt, pb)
        ncams = [dict(cam=nd.Cam.of(c["cam"]), T_cam_imu=nd.Tf.from7(c["T_cam_imu"].as7()), px=c["px"], f=c["f"], grad=c["grad"],
                      level=c["level"], type=c["type"], xyz_world=c["xyz_world"], usable=c["usable"]) for c in sc["cams"]]
        g = npp.optimize_pose(error_type, ncams, nd.Tf.from7(sc["T_imu_world_init"].as7()), opt.outlier_threshold, opt.max_iter, opt.eps,
                              R_prior=[opt.R_prior[k] for k in range(4)] if prior else None, prior_lambda=opt.prior_lambda)
        assert r.status == g["status"] == 0
        assert r.measurement_sigma == g["sigma"]                       # float arithmetic on the median: exact
        assert r.iters == g["iters"] and r.n_meas == g["n_meas"]
        assert (r.n_deleted_edges, r.n_deleted_corners) == (g["n_deleted_edges"], g["n_deleted_corners"])
        for k, o in zip(keep, g["outlier"]):
            assert np.array_equal(k["outlier"][:len(o)], o)
        T = fe.se3_to_numpy(r.T_imu_world)
        Tg = np.concatenate([g["T"].q, g["T"].t])
        assert np.abs(T - Tg).max() <= 1e-9, np.abs(T - Tg).max()
        assert r.reproj_error_before == pytest.approx(g["err_before"], rel=1e-7)
        assert r.reproj_error_after == pytest.approx(g["err_after"], rel=1e-7)


@pytest.mark.parametrize("shape", [(640, 480), (327, 243)])
def test_detector_oracle_vs_numpy_second_opinion(oracle_lib, shape):
    """f-2: FAST-10 + score + 3x3 non-maximum suppression + grid, Scharr edgelets + angle histogram, fillFeatures -- the
    oracle against the dense-array reading of tests/np_restatement_detector.py: positions, levels, types, scores exact,
    directions exact (same libm on both sides here)."""
    orc = oracle_lib
    w, h = shape
    cam = synth.Camera.euroc_like(w, h)
    sc = synth.make_align_scene(150 + w, n_features=8, cam=cam)
    levels = orc.create_img_pyramid(sc.img_ref, 5)
    n_cells = int(np.ceil(w / 30)) * int(np.ceil(h / 30))
    rng = np.random.RandomState(w)
    n_total = n_edgelets = 0
    for kw, occ, mask, max_n in (
            (dict(), None, None, None),
            (dict(threshold_secondary=25.0, threshold_primary=30.0), None, None, None),
            (dict(detect_edgelets=0), (rng.uniform(size=n_cells) < 0.3).astype(np.uint8), None, None),
            (dict(threshold_primary=20.0, threshold_secondary=60.0, border=5, max_level=3, min_level=1), None, None, None),
            (dict(cell_size=17), None, (rng.uniform(size=(h, w)) < 0.7).astype(np.uint8) * 255, None)):
        opt = capi.default_detector_options(**kw)
        do = orc.detect_features(opt, levels, occ, mask, max_n)
        dn = ndet.detect(levels, cell_size=opt.cell_size, max_level=opt.max_level, min_level=opt.min_level, border=opt.border,
                         detect_edgelets=bool(opt.detect_edgelets), threshold_primary=opt.threshold_primary,
                         threshold_secondary=opt.threshold_secondary, occupancy=occ, mask=mask, max_n_features=max_n)
        assert len(do["score"]) > 10
        n_edgelets += int((do["type"] == capi.FT_EDGELET).sum())
        assert np.array_equal(do["type"], dn["type"])
        assert np.array_equal(do["px"], dn["px"]) and np.array_equal(do["level"], dn["level"])
        assert np.array_equal(do["score"], dn["score"])
        assert np.abs(do["grad"] - dn["grad"]).max() < 1e-6
        n_total += len(do["score"])
    assert n_total > 300 and n_edgelets > 50
    # the FAST pieces on their own: the score image against the oracle's per-corner calls, on a level with saturated pixels
    img = levels[1].copy()
    img[::7, ::5] = 255; img[3::11, 2::9] = 0
    img = np.ascontiguousarray(img)
    s = ndet.fast_score_image(img)
    ys, xs = np.nonzero(s >= 10)
    assert len(xs) > 100
    xy_o, sc_o, nm_o = orc.fast_corners(img, 10)
    assert np.array_equal(xy_o, np.stack([xs, ys], axis=1)) and np.array_equal(sc_o, s[ys, xs])      # same set, same scores
    kept = ndet.fast_corners(img, 10)
    assert [(int(x), int(y)) for x, y in xy_o[nm_o]] == [(x, y) for x, y, _ in kept]                  # list walk == dense 3x3
    # blur / Scharr / histogram angle piece by piece
    assert np.array_equal(orc.gaussian_blur_3x3(img), ndet.gaussian_blur_3x3(img))
    for xd in (True, False):
        assert np.array_equal(orc.scharr_16s(img, xd), ndet.scharr(img, xd))
    for (x, y) in ((20, 20), (1, 1), (img.shape[1] - 2, img.shape[0] - 3), (57, 33)):
        assert orc.angle_at_pixel(img, x, y) == ndet.angle_at_pixel_using_histogram(img, x, y, 4)


@pytest.mark.parametrize("shape", [(640, 480), (752, 480), (327, 243), (48, 33), (32, 2)])
def test_pyramid_oracle_vs_numpy_second_opinion(oracle_lib, shape):
    """a-0: createImgPyramid / halfSample, the SSE2 rule where the level's width is a multiple of 16 and the scalar rule
    elsewhere (752 -> 376 -> 188 switches after the first level), odd sizes included."""
    import np_restatement as n0
    w, h = shape
    rng = np.random.RandomState(w * 7 + h)
    img = rng.randint(0, 256, (h, w)).astype(np.uint8)
    img[: h // 3] = np.where(rng.uniform(size=(h // 3, w)) < 0.5, 255, 254)     # sums on the rounding edges
    n_levels = 5 if min(w, h) >= 32 else 2
    po = oracle_lib.create_img_pyramid(img, n_levels)
    pn = n0.create_img_pyramid(img, n_levels)
    for a, b in zip(po, pn):
        assert a.shape == b.shape and np.array_equal(a, b)


# ---------------------------------------------------------------------------------------------------------------
# SparseImgAlign's Gauss-Newton LOOP (a-1, a-2, a-8): update, applyPrior with I_prior_ rebuilt at iteration 0 of every
# level, the stop / roll-back rules -- tests/np_restatement_gn.py on top of np_restatement.evaluate
# ---------------------------------------------------------------------------------------------------------------
def _gn_compare(orc, cams, opt, T_init=None, alpha_init=0.0, beta_init=0.0, prior_c=None, prior_np=None, tol_pose=1e-9):
    import helpers
    import np_restatement_gn as ngn
    pb = orc.problem_from_scenes(cams, T_init=T_init, prior=prior_c, alpha_init=alpha_init, beta_init=beta_init)
    n, ro, _ = orc.sparse_align_run(opt, pb)
    sc0 = cams[0][0]
    Ti = T_init if T_init is not None else sc0.T_icur_iref_init
    rn = ngn.run(cams, opt.max_level, opt.min_level, opt.patch_size, Ti, alpha_init, beta_init, max_iter=opt.max_iter,
                 eps=opt.eps, est_alpha=bool(opt.estimate_illumination_gain), est_beta=bool(opt.estimate_illumination_offset),
                 robust=bool(opt.robustification), weight_scale=opt.weight_scale, prior=prior_np)
    for level in range(opt.max_level, opt.min_level - 1, -1):
        assert ro.iters[level] == rn["iters"][level], (level, list(ro.iters), rn["iters"])
        assert ro.n_meas[level] == rn["n_meas"][level]
    assert ro.status == rn["status"]
    qo = np.array([ro.T_icur_iref.q[i] for i in range(4)]); to = np.array([ro.T_icur_iref.t[i] for i in range(3)])
    dq = min(np.abs(qo - rn["T"].q).max(), np.abs(qo + rn["T"].q).max())
    assert max(dq, np.abs(to - rn["T"].t).max()) < tol_pose
    assert abs(ro.alpha - rn["alpha"]) < 1e-9 and abs(ro.beta - rn["beta"]) < 1e-7
    return ro, rn


@pytest.mark.parametrize("cam_kind", ["pinhole", "radtan"])
def test_gauss_newton_loop_oracle_vs_numpy_second_opinion(oracle_lib, cam_kind):
    import helpers
    orc = oracle_lib
    cam = synth.Camera.test_camera() if cam_kind == "pinhole" else synth.Camera.euroc_like()
    sc = helpers.small_scene(71, n=220, cam=cam, border_features=30, invalid_fraction=0.05, gain=1.03, offset=2.0)
    ref, cur = helpers.scene_pyramids(orc, sc)
    cams = [(sc, ref, cur)]
    total_iters = 0
    # the handler's levels, all levels, an iteration cap that cuts levels short, an eps nothing reaches
    for kw in (dict(min_level=2), dict(min_level=0), dict(min_level=1, max_iter=3), dict(min_level=3, eps=1e-12, max_iter=6)):
        ro, rn = _gn_compare(orc, cams, capi.default_align_options(**kw))
        total_iters += sum(rn["iters"].values())
    # illumination gain and offset estimated, with and without robust weights, from non-zero starting values
    for robust in (0, 1):
        opt = capi.default_align_options(min_level=1, estimate_illumination_gain=1, estimate_illumination_offset=1,
                                         robustification=robust)
        _gn_compare(orc, cams, opt, alpha_init=0.01, beta_init=-0.5)
    # only one of the two illumination terms: the other's row and column stay exactly zero
    _gn_compare(orc, cams, capi.default_align_options(min_level=2, estimate_illumination_gain=1))
    _gn_compare(orc, cams, capi.default_align_options(min_level=2, estimate_illumination_offset=1))
    assert total_iters > 20


def test_gauss_newton_prior_oracle_vs_numpy_second_opinion(oracle_lib):
    """applyPrior over three levels (I_prior_ rebuilt from each level's first Hessian), prior x illumination."""
    import helpers
    import np_restatement_direct as nd2
    orc = oracle_lib
    sc = helpers.small_scene(72, n=260, gain=1.02, offset=1.0)
    ref, cur = helpers.scene_pyramids(orc, sc)
    cams = [(sc, ref, cur)]
    Tp = synth.SE3(synth.quat_from_axis_angle([0.3, -1, 0.2], 0.004), [0.003, -0.002, 0.001])
    Tp_np = nd2.Tf.from7(Tp.as7())
    for lam_r, lam_t, la, lb in ((0.5, 0.0, 0.0, 0.0), (2.0, 3.0, 0.0, 0.0), (0.1, 0.1, 0.5, 0.5), (0.0, 0.7, 0.0, 0.3)):
        illum = int(la > 0 or lb > 0)
        prior_c = helpers.make_prior(Tp, lam_r, lam_t, alpha=0.01, beta=-0.5, lambda_alpha=la, lambda_beta=lb)
        prior_np = dict(T=Tp_np, alpha=0.01, beta=-0.5, lambda_rot=lam_r, lambda_trans=lam_t, lambda_alpha=la, lambda_beta=lb)
        opt = capi.default_align_options(min_level=2, estimate_illumination_gain=illum, estimate_illumination_offset=illum)
        ro, rn = _gn_compare(orc, cams, opt, prior_c=prior_c, prior_np=prior_np)
        assert sum(rn["iters"].values()) >= 3
        # the prior really acts: the unconstrained run ends somewhere else
        r_free = orc.sparse_align_run(opt, orc.problem_from_scenes(cams))[1]
        if lam_r >= 0.5:
            assert helpers.se3_max_abs_diff(r_free.T_icur_iref, ro.T_icur_iref) > 1e-6


def test_gauss_newton_stereo_and_failed_solve_oracle_vs_numpy_second_opinion(oracle_lib):
    import helpers
    orc = oracle_lib
    a = helpers.small_scene(73, n=200, border_features=20)
    b = synth.make_align_scene(73, n_features=180, cam=synth.Camera.euroc_like(), border_features=10)
    ra, ca = helpers.scene_pyramids(orc, a)
    rb, cb = helpers.scene_pyramids(orc, b)
    # two cameras with different extrinsics and intrinsics: one H, one g (sparse_img_align.cpp:138-154)
    _gn_compare(orc, [(a, ra, ca), (b, rb, cb)], capi.default_align_options(min_level=1))
    # a solve that fails -- every patch invisible: H = 0, g = 0, Eigen's LDLT returns zeros, NOT a NaN: no failure, the
    # step is zero, the level ends at once (max |dx| = 0 < eps)
    far = synth.SE3(synth.quat_from_axis_angle([0, 1, 0], 3.0), [0.0, 0.0, 0.0])   # everything behind the camera
    ro, rn = _gn_compare(orc, [(a, ra, ca)], capi.default_align_options(min_level=2), T_init=far)
    assert all(rn["iters"][l] == 1 for l in (4, 3, 2)) and all(rn["n_meas"][l] == 0 for l in (4, 3, 2))


# ---------------------------------------------------------------------------------------------------------------
# Point::optimize (f-3, second half): point.cpp:248-325 restated in tests/np_restatement_pose.py
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sphere", [False, True])
@pytest.mark.parametrize("seed,n_points,n_views,n_iter", [(6, 300, 5, 5), (7, 400, 8, 10), (8, 5, 2, 3)])
def test_point_optimize_oracle_vs_numpy_second_opinion(oracle_lib, sphere, seed, n_points, n_views, n_iter):
    import np_restatement_pose as npp
    import pose_helpers as ph
    orc = oracle_lib
    sc = ph.make_structure_scene(seed, n_points=n_points, n_views=n_views)
    po, io = orc.optimize_points(sc["views"], sc["obs_begin"], sc["obs_view"], sc["obs_f"], sc["pos0"], n_iter=n_iter,
                                 using_bearing_vector=sphere)
    views = [nd.Tf.from7(np.asarray(v, np.float64)) for v in sc["views"]]
    pn, it_n = np.zeros_like(po), np.zeros(n_points, np.int32)
    for i in range(n_points):
        o0, o1 = int(sc["obs_begin"][i]), int(sc["obs_begin"][i + 1])
        obs = [(views[int(sc["obs_view"][o])], sc["obs_f"][o]) for o in range(o0, o1)]
        pn[i], it_n[i] = npp.point_optimize(obs, sc["pos0"][i], n_iter, sphere)
    # constructed degenerate landmarks (pose_helpers): 11 sees one view twice (singular along the ray: the solution
    # there is rounding noise over a tiny pivot), 23 starts behind a camera and runs away -- compared loosely below
    wild = np.zeros(n_points, bool)
    if n_points > 23:
        wild[[11, 23]] = True
    lone = np.diff(sc["obs_begin"]) < 2
    assert np.array_equal(pn[lone], sc["pos0"][lone]) and np.array_equal(po[lone], sc["pos0"][lone])
    assert (it_n[lone] == 0).all() and (io[lone] == 0).all()
    # iteration counts: in lockstep until convergence; once converged chi2 only moves in its last bits, so the "error
    # grew" stop can fire one iteration apart between the two readings (different solvers)
    ok = ~wild & ~lone
    assert np.abs(it_n - io)[ok].max() <= 1 and np.mean((it_n != io)[ok]) < 0.05
    if n_iter <= 5:
        assert np.array_equal(it_n[ok], io[ok])
    d = np.abs(pn - po).max(1)
    fin = np.isfinite(po).all(1)
    assert np.array_equal(fin, np.isfinite(pn).all(1))
    assert d[ok & fin].max() < 1e-6 and np.median(d[ok & fin]) < 1e-12
    # (landmark 11's singular 3x3 system has no defined solution: what comes out is the solver's -- Eigen's LDLT in the
    # oracle and the kernel, LAPACK here -- so it is not compared; the GPU test compares it with the oracle, same solver)


# ---------------------------------------------------------------------------------------------------------------
# The loop of StereoTriangulation::compute (row *J): one Matcher, align_1d per feature, landmark + frame1 feature per
# success, stop at n_desired -- stereo_triangulation.cpp:88-137 restated in tests/np_restatement_direct.py
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cam_kind,n_desired", [("pinhole", 60), ("radtan", 10_000)])
def test_stereo_triangulation_loop_oracle_vs_numpy_second_opinion(oracle_lib, cam_kind, n_desired):
    orc = oracle_lib
    cam = synth.Camera.test_camera() if cam_kind == "pinhole" else synth.Camera.euroc_like()
    sc = synth.make_align_scene(66, n_features=10, cam=cam, rot_deg=(0.5, 1.5), trans_m=(0.05, 0.15))   # "left" = ref, "right" = cur
    n = 240
    sd = synth.make_seed_set(sc, n, margin=6, levels=(0, 1, 2))
    ftype = np.where(sd["type"] == 0, capi.FT_EDGELET, capi.FT_CORNER).astype(np.uint8)   # detector output, not seeds
    rv, cv, nrv, ncv = _views(orc, sc, dict(mu_range=0.0), cam_kind)
    T = sc.T_cur_f_w_gt * sc.T_ref_f_w.inverse()
    d_mean = float(np.median(sd["true_depth"]))
    d_inv = [1.0 / d_mean, 1.0 / (0.3 * d_mean), 1.0 / (15.0 * d_mean)]
    # the reference's order: corners shuffled, then the rest shuffled (:76-84) -- any order is an input to both readings
    rng = np.random.RandomState(5)
    corners = np.nonzero(ftype == capi.FT_CORNER)[0]; rest = np.nonzero(ftype != capi.FT_CORNER)[0]
    order = np.concatenate([rng.permutation(corners), rng.permutation(rest)]).astype(np.int32)
    fb, keep = orc.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], ftype)
    mo, ro, fo = orc.stereo_triangulate(rv, cv, T, fb, order, n_desired, d_inv)
    mn, rn, fn = nd.stereo_triangulate(nrv, ncv, nd.Tf.from7(T.as7()), sd["px"].reshape(-1, 2), sd["f"].reshape(-1, 3),
                                       sd["grad"].reshape(-1, 2), sd["level"], ftype, order, n_desired, *d_inv)
    visited = len(rn)
    # the same features visited, the same codes (the oracle reports per feature index, -1 = never visited)
    assert np.array_equal(ro[order[:visited]], rn) and (ro[order[visited:]] == -1).all()
    assert fo == fn and len(mo) == len(mn) and len(mn) == min(n_desired, int((rn == nd.SUCCESS).sum()))
    if n_desired < 100:
        assert len(mn) == n_desired and visited < n      # the early stop (:131-132)
    else:
        assert visited == n and fn > 0 and len(set(rn.tolist())) >= 3
    for a, b in zip(mo, mn):
        assert a["i_ref"] == b["i_ref"]
        assert abs(a["depth"] - b["depth"]) <= 1e-9 * abs(b["depth"])
        assert np.abs(a["xyz_cam0"] - b["xyz_cam0"]).max() <= 1e-9 * abs(b["depth"])
        assert np.abs(a["px"] - b["px"]).max() <= 1e-4 and np.abs(a["f"] - b["f"]).max() < 1e-6
        assert np.abs(a["grad"] - b["grad"]).max() < 1e-9


# ---------------------------------------------------------------------------------------------------------------
# reprojector_utils::getCandidate (f-4): reprojector.cpp:489-543 + Frame::isVisible restated in
# tests/np_restatement_direct.py, against the HOST MIRROR's getCandidate (svo_hip_host.cpp) run on the CPU through
# tests/cpp/host_candidates_cpu (the device kernel is compared with the same restatement in test_sparse_align_gpu.py)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cam_kind", ["pinhole", "radtan"])
def test_get_candidate_host_mirror_vs_numpy_second_opinion(tmp_path, cam_kind):
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "svo_pro_universal_amd", "host"), "libsvo_hip_host.so"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "tests", "cpp"), "host_candidates_cpu"])
    cam = synth.Camera.test_camera() if cam_kind == "pinhole" else synth.Camera.euroc_like()
    rng = np.random.RandomState(17)
    T_w_cur = synth.SE3(synth.quat_from_axis_angle([0.2, 1, -0.1], 0.15), (0.4, -0.1, 0.2))
    T_w_ref = T_w_cur * synth.SE3(synth.quat_from_axis_angle([0, 1, 0.3], 0.25), (0.3, 0.05, 0.1))
    T_cur, T_ref = T_w_cur.inverse(), T_w_ref.inverse()
    n = 4000
    kind = (rng.uniform(size=n) < 0.5).astype(int)
    v, mu = np.zeros((n, 3)), np.ones(n)
    for i in range(n):
        if kind[i]:    # a seed of the reference keyframe: bearing vector and inverse depth
            f = np.array([rng.uniform(-0.9, 0.9), rng.uniform(-0.7, 0.7), 1.0]); v[i] = f / np.linalg.norm(f)
            mu[i] = 1.0 / rng.uniform(0.5, 8.0)
        else:          # a landmark somewhere around the current view, many outside it or behind it
            v[i] = T_w_cur.transform(np.array([rng.uniform(-6, 6), rng.uniform(-4, 4), rng.uniform(-1.0, 8.0)]))
    d = list(cam.dist) if cam.dist is not None else [0.0] * 4
    lines = ["%d %d %.17g %.17g %.17g %.17g %d %.17g %.17g %.17g %.17g" % (cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy,
                                                                          1 if cam.dist is not None else 0, *d),
             " ".join("%.17g" % x for x in T_cur.as7()), " ".join("%.17g" % x for x in T_ref.as7()), str(n)]
    lines += ["%d %.17g %.17g %.17g %.17g" % (0 if not kind[i] else 1, v[i, 0], v[i, 1], v[i, 2], mu[i]) for i in range(n)]
    fin, fout = tmp_path / "in.txt", tmp_path / "out.txt"
    fin.write_text("\n".join(lines) + "\n")
    r = subprocess.run([os.path.join(root, "tests", "cpp", "host_candidates_cpu"), str(fin), str(fout)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    host = np.loadtxt(str(fout))
    ncam = nd.Cam.of(cam)
    Tc, Tr = nd.Tf.from7(T_cur.as7()), nd.Tf.from7(T_ref.as7())
    n_vis = 0
    for i in range(n):
        ok, px = nd.get_candidate(ncam, Tc, Tr, v[i] if not kind[i] else None, v[i], mu[i])
        if ok != bool(host[i, 0]):   # a pixel within rounding of an integer boundary may fall on either side
            assert min(abs(px[0] - round(px[0])), abs(px[1] - round(px[1]))) < 1e-9, i
        elif ok:
            assert np.abs(host[i, 1:3] - px).max() < 1e-9
            n_vis += 1
    assert 300 < n_vis < n - 300


def test_candidate_sort_with_packed_keys_is_the_reference_sort():
    """reprojector_utils::sortCandidatesByReprojStats compares one 128-bit key per candidate instead of three fields
    (svo_hip_host.cpp).  The result must be the vector the reference's std::sort call leaves -- for ties too, whose order
    is introsort's and depends on every comparison's outcome: tests/cpp/host_sort_cpu sorts 400 adversarial lists (runs of
    equal candidates, -0.0 / +0.0, denormals, infinities, INT_MIN / INT_MAX counts, NaN scores) both ways."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "tests", "cpp"), "host_sort_cpu"])
    r = subprocess.run([os.path.join(root, "tests", "cpp", "host_sort_cpu")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok 400 "), r.stdout + r.stderr


def test_upgrade_seeds_to_features_host_half():
    """svo_hip::upgradeSeedsToFeatures (round 6; FrameHandlerBase::upgradeSeedsToFeatures, frame_handler_base.cpp:828-920) on a
    hand-made keyframe / frame pair: points at the seeds' positions, types, observations, track ids, cleared seed references, the
    list of upgraded edgelets, two features on one seed, a feature with a landmark of its own; removeObservationsOf
    (Map::removeKeyframe).  tests/cpp/host_upgrade_cpu, no GPU call."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "tests", "cpp"), "host_upgrade_cpu"])
    r = subprocess.run([os.path.join(root, "tests", "cpp", "host_upgrade_cpu")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
