"""GPU parity for a-9 (KLT) and a-10..a-14 (warp / ZMSSD / align / matcher / depth
filter): HIP kernels through the C ABI vs the CPU oracle on the same seeded inputs.

Bars: integer outputs (status, match result codes, search levels, feature types,
success flags) exact; KLT positions bit-identical (its float part is evaluated in
the same order on both sides and the per-pixel part is integer); matcher sub-pixel
positions <= 1e-4 px; depth-filter state relative 1e-9 (libm vs device libm)."""
import numpy as np
import pytest

from svo_pro_universal_amd import _capi as capi, frontend as fe, synth

import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["eight_lanes_per_unit", "one_lane_per_unit", "packed", "one_wave_per_unit"], autouse=True)
def matcher_geometry(request, gpu_ctx):
    """The matcher kernels exist in three geometries with bit-identical results: eight lanes per unit (what a small
    batch gets), one lane per unit, and the packed geometry of large batches (one lane per unit for geometry / warp /
    scan, the refinements as jobs of a workgroup-wide queue: update_seeds_packed_kernel, and since round 3
    match_packed_kernel for the direct matcher and the plain epipolar match), and one wave per unit with the epipolar
    scan 64 steps at a time (round 3: what the stereo seam's 500-step searches get; the direct matcher, which has no
    scan, runs eight lanes per unit there).  Every test of this file runs through all of them."""
    import os
    old = os.environ.get("SVOH_MATCHER_G8")
    os.environ["SVOH_MATCHER_G8"] = {"eight_lanes_per_unit": "1", "one_lane_per_unit": "0", "packed": "2", "one_wave_per_unit": "3"}[request.param]
    gpu_ctx.reload_knobs()
    yield request.param
    if old is None:
        os.environ.pop("SVOH_MATCHER_G8", None)
        gpu_ctx.reload_knobs()
    else:
        os.environ["SVOH_MATCHER_G8"] = old
        gpu_ctx.reload_knobs()


def scene_and_frames(gpu_ctx, orc, seed, cam=None, **kw):
    sc = synth.make_align_scene(seed, n_features=10, cam=cam, rot_deg=(0.5, 1.5), trans_m=(0.05, 0.15), **kw)
    ref = orc.create_img_pyramid(sc.img_ref, 5)
    cur = orc.create_img_pyramid(sc.img_cur, 5)
    fr = gpu_ctx.build_pyramid(sc.img_ref, 5)
    fc = gpu_ctx.build_pyramid(sc.img_cur, 5)
    return sc, ref, cur, fr, fc


@pytest.mark.parametrize("cam_kind", ["pinhole", "radtan"])
def test_klt_bit_identical(gpu_ctx, oracle_lib, cam_kind):
    orc = oracle_lib
    cam = synth.Camera.test_camera() if cam_kind == "pinhole" else synth.Camera.euroc_like()
    sc, ref, cur, fr, fc = scene_and_frames(gpu_ctx, orc, 51, cam)
    tr = synth.make_track_set(sc, 400, margin=6)   # some tracks start next to the border
    for kw in (dict(), dict(min_level=2), dict(max_iter=3), dict(patch_sizes=[8, 8, 16, 16, 8]),
               dict(min_update_squared=1e-6)):
        opt = capi.default_klt_options(**kw)
        po, so = orc.klt_track_batch(opt, ref, cur, tr["px_ref"], tr["px_cur_init"])
        pg, sg = gpu_ctx.klt_track_batch(opt, fr, fc, tr["px_ref"], tr["px_cur_init"])
        assert np.array_equal(so, sg)
        assert np.array_equal(po, pg), np.abs(po - pg).max()
        if not kw:
            ok = so == 1
            err = np.linalg.norm((pg - tr["px_true"]).reshape(-1, 2), axis=1)
            assert ok.mean() > 0.9 and np.median(err[ok]) < 0.1


def test_klt_per_track_reference_frames_and_edge_cases(gpu_ctx, oracle_lib):
    orc = oracle_lib
    sc, ref, cur, fr, fc = scene_and_frames(gpu_ctx, orc, 52)
    sc2, ref2, cur2, fr2, fc2 = scene_and_frames(gpu_ctx, orc, 53)
    tr = synth.make_track_set(sc, 64)
    # half of the tracks take their template from another frame (klt_template_is_first_observation)
    refs_o = [ref if i % 2 else ref2 for i in range(64)]
    refs_g = [fr if i % 2 else fr2 for i in range(64)]
    opt = capi.default_klt_options()
    # start far away / outside the image / at the border
    px0 = tr["px_cur_init"].copy().reshape(-1, 2)
    px0[0] = [-50.0, 10.0]; px0[1] = [5000.0, 5000.0]; px0[2] = [0.0, 0.0]; px0[3] += 80.0
    po, so = orc.klt_track_batch(opt, refs_o, cur, tr["px_ref"], px0.ravel())
    pg, sg = gpu_ctx.klt_track_batch(opt, refs_g, fc, tr["px_ref"], px0.ravel())
    assert np.array_equal(so, sg) and np.array_equal(po, pg)
    assert so[1] == 0
    # empty batch
    pg, sg = gpu_ctx.klt_track_batch(opt, [], fc, np.zeros(0, np.int32), np.zeros(0))
    assert pg.size == 0 and sg.size == 0
    with pytest.raises(fe.SvohError) as e:
        gpu_ctx.klt_track_batch(capi.default_klt_options(patch_sizes=[12, 16, 16, 8, 8]), fr, fc, tr["px_ref"], tr["px_cur_init"])
    assert e.value.code == -5


def views(gpu_ctx, orc, sc, ref, cur, fr, fc, mu_range):
    ov_r = orc.make_frame_view(ref, sc.cam, sc.T_ref_f_w, mu_range, 1)
    ov_c = orc.make_frame_view(cur, sc.cam, sc.T_cur_f_w_gt, 0.0, 2)
    gv_r = fe.make_frame_view(fr, sc.cam, sc.T_ref_f_w, mu_range, 1)
    gv_c = fe.make_frame_view(fc, sc.cam, sc.T_cur_f_w_gt, 0.0, 2)
    return ov_r, ov_c, gv_r, gv_c


@pytest.mark.parametrize("cam_kind", ["pinhole", "radtan"])
@pytest.mark.parametrize("sphere", [0, 1])
def test_update_seeds_parity(gpu_ctx, oracle_lib, cam_kind, sphere):
    orc = oracle_lib
    cam = synth.Camera.test_camera() if cam_kind == "pinhole" else synth.Camera.euroc_like()
    sc, ref, cur, fr, fc = scene_and_frames(gpu_ctx, orc, 61, cam)
    sd = synth.make_seed_set(sc, 3000, margin=12)
    sd["type"][::17] = capi.FT_MAPPOINT_SEED
    sd["type"][5::41] = capi.FT_CORNER        # not a seed: skipped
    sd["type"][7::53] = capi.FT_OUTLIER
    sd["type"][9::59] = capi.FT_EDGELET_SEED_CONVERGED
    ov_r, ov_c, gv_r, gv_c = views(gpu_ctx, orc, sc, ref, cur, fr, fc, sd["mu_range"])
    for mkw, dkw in ((dict(), dict()), (dict(affine_est_gain=1, max_epi_search_steps=500), dict(check_convergence=1)),
                     (dict(subpix_refinement=0, affine_est_offset=0), dict(use_vogiatzis_update=0, check_visibility=0))):
        mopt = capi.default_matcher_options(scan_on_unit_sphere=sphere, **mkw)
        dopt = capi.default_depth_filter_options(sc.cam, **dkw)
        fbo, ko = orc.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
        fbg, kg = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
        state_o, state_g = sd["state"], sd["state"]
        for rnd in range(3):  # several updates, feeding the state back like successive frames do
            nso, state_o, so, mro = orc.update_seeds_batch(mopt, dopt, [ov_r], ov_c, fbo, state_o)
            nsg, state_g, sg, mrg = gpu_ctx.update_seeds_batch(mopt, dopt, [gv_r], gv_c, fbg, state_g)
            assert nso == nsg and np.array_equal(so, sg)
            assert np.array_equal(mro, mrg), np.nonzero(mro != mrg)
            assert np.array_equal(ko["type"], kg["type"])
            assert np.allclose(state_g, state_o, rtol=1e-9, atol=0)
        assert nso > 1500
        ok = so == 1
        e = np.abs(1 / state_g.reshape(-1, 4)[ok, 0] - sd["true_depth"][ok]) / sd["true_depth"][ok]
        assert np.median(e) < 0.03


def test_seed_edge_cases(gpu_ctx, oracle_lib):
    orc = oracle_lib
    sc, ref, cur, fr, fc = scene_and_frames(gpu_ctx, orc, 62)
    sd = synth.make_seed_set(sc, 200, margin=2)  # seeds right at the image border
    ov_r, ov_c, gv_r, gv_c = views(gpu_ctx, orc, sc, ref, cur, fr, fc, sd["mu_range"])
    mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(sc.cam)
    # same frame id -> every update is refused, state untouched
    gv_same = fe.make_frame_view(fc, sc.cam, sc.T_cur_f_w_gt, 0.0, 1)
    fbg, kg = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
    ns, st, succ, mr = gpu_ctx.update_seeds_batch(mopt, dopt, [gv_r], gv_same, fbg, sd["state"])
    assert ns == 0 and np.array_equal(st, sd["state"]) and np.all(mr == capi.MATCH_NOT_RUN)
    # border seeds + huge uncertainty + zero-translation (warp NaN path) against the oracle
    state = sd["state"].copy().reshape(-1, 4)
    state[::3, 1] *= 100.0
    for T_cur in (sc.T_cur_f_w_gt, sc.T_ref_f_w):
        ov_c2 = orc.make_frame_view(cur, sc.cam, T_cur, 0.0, 2)
        gv_c2 = fe.make_frame_view(fc, sc.cam, T_cur, 0.0, 2)
        fbo, ko = orc.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
        fbg, kg = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
        nso, sto, so, mro = orc.update_seeds_batch(mopt, dopt, [ov_r], ov_c2, fbo, state.ravel())
        nsg, stg, sg, mrg = gpu_ctx.update_seeds_batch(mopt, dopt, [gv_r], gv_c2, fbg, state.ravel())
        assert nso == nsg and np.array_equal(mro, mrg) and np.array_equal(ko["type"], kg["type"])
        assert np.allclose(stg, sto, rtol=1e-9, atol=0, equal_nan=True)
    # empty batch
    fbe, ke = fe.make_feature_batch(np.zeros(0, np.int32), np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0, np.int32),
                                    np.zeros(0, np.uint8))
    ns, st, succ, mr = gpu_ctx.update_seeds_batch(mopt, dopt, [gv_r], gv_c, fbe, np.zeros(0))
    assert ns == 0


@pytest.mark.parametrize("cam_kind", ["pinhole", "radtan"])
def test_match_direct_parity(gpu_ctx, oracle_lib, cam_kind):
    orc = oracle_lib
    cam = synth.Camera.test_camera() if cam_kind == "pinhole" else synth.Camera.euroc_like()
    sc, ref, cur, fr, fc = scene_and_frames(gpu_ctx, orc, 63, cam)
    sd = synth.make_seed_set(sc, 2000, margin=3, levels=(0, 1, 2, 3))
    ov_r, ov_c, gv_r, gv_c = views(gpu_ctx, orc, sc, ref, cur, fr, fc, sd["mu_range"])
    x = sd["f"].reshape(-1, 3).T * sd["true_depth"]
    px_true = sc.cam.project(sc.T_w_cur.inverse().transform(sc.T_w_ref.transform(x)))
    px_init = np.ascontiguousarray((px_true + np.random.RandomState(1).uniform(-2.0, 2.0, px_true.shape)).T).ravel()
    px_init[:20] += 40.0  # some hopeless starts -> alignment failures / too far
    ftype = np.where(sd["type"] == 0, capi.FT_EDGELET, capi.FT_CORNER)
    for mkw in (dict(), dict(affine_est_gain=1), dict(affine_est_offset=0, align_max_iter=3)):
        mopt = capi.default_matcher_options(**mkw)
        fbo, ko = orc.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], ftype)
        fbg, kg = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], ftype)
        oo = orc.match_direct_batch(mopt, [ov_r], ov_c, fbo, sd["true_depth"], px_init)
        og = gpu_ctx.match_direct_batch(mopt, [gv_r], gv_c, fbg, sd["true_depth"], px_init)
        assert np.array_equal(oo["result"], og["result"])
        assert np.array_equal(oo["search_level"], og["search_level"])
        assert np.abs(oo["px_cur"] - og["px_cur"]).max() <= 1e-4
        ok = oo["result"] == 0
        assert ok.sum() > 1000 and len(set(oo["result"])) >= 3  # successes and several failure kinds
        assert np.allclose(oo["A"], og["A"], rtol=1e-12, atol=1e-14)
        assert np.abs(oo["f_cur"] - og["f_cur"])[np.repeat(ok, 3)].max() < 1e-6
        assert np.allclose(oo["h_inv"], og["h_inv"], rtol=1e-6)
        e = np.linalg.norm(og["px_cur"].reshape(-1, 2)[ok] - px_true.T[ok], axis=1)
        assert np.median(e) < 0.3


@pytest.mark.parametrize("cam_kind", ["pinhole", "radtan"])
def test_match_direct_pixelwise_warp_parity(gpu_ctx, oracle_lib, cam_kind):
    """Matcher::Options::use_affine_warp_ == false: findMatchDirect with warp::warpPixelwise (matcher.cpp:67-81,
    patch_warp.cpp:158-230) through svoh_match_direct_batch_pixelwise against the oracle: result codes and search
    levels exact, matched pixels <= 1e-4 (the patch itself is exact or the codes and float32 positions would move)."""
    orc = oracle_lib
    cam = synth.Camera.test_camera() if cam_kind == "pinhole" else synth.Camera.euroc_like()
    sc, ref, cur, fr, fc = scene_and_frames(gpu_ctx, orc, 67, cam)
    sd = synth.make_seed_set(sc, 1500, margin=3, levels=(0, 1, 2, 3))
    ov_r, ov_c, gv_r, gv_c = views(gpu_ctx, orc, sc, ref, cur, fr, fc, sd["mu_range"])
    x = sd["f"].reshape(-1, 3).T * sd["true_depth"]
    lm = np.ascontiguousarray(sc.T_w_ref.transform(x).T)
    lm[::97] += 50.0          # landmarks far off their pixel: patches that leave the reference image (kFailWarp) or match nothing
    px_true = sc.cam.project(sc.T_w_cur.inverse().transform(sc.T_w_ref.transform(x)))
    px_init = np.ascontiguousarray((px_true + np.random.RandomState(3).uniform(-2.0, 2.0, px_true.shape)).T).ravel()
    ftype = np.where(sd["type"] == 0, capi.FT_EDGELET, capi.FT_CORNER)
    for mkw in (dict(), dict(affine_est_gain=1)):
        mopt = capi.default_matcher_options(**mkw)
        fbo, ko = orc.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], ftype)
        fbg, kg = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], ftype)
        oo = orc.match_direct_batch(mopt, [ov_r], ov_c, fbo, sd["true_depth"], px_init, landmark_xyz=lm)
        og = gpu_ctx.match_direct_batch(mopt, [gv_r], gv_c, fbg, sd["true_depth"], px_init, landmark_xyz=lm)
        assert np.array_equal(oo["result"], og["result"]), np.nonzero(oo["result"] != og["result"])[0][:10]
        assert np.array_equal(oo["search_level"], og["search_level"])
        assert np.abs(oo["px_cur"] - og["px_cur"]).max() <= 1e-4
        ok = oo["result"] == 0
        assert ok.sum() > 700 and len(set(oo["result"])) >= 3
        assert np.allclose(oo["A"], og["A"], rtol=1e-12, atol=1e-14)
        assert np.abs(oo["f_cur"] - og["f_cur"])[np.repeat(ok, 3)].max() < 1e-6
        e = np.linalg.norm(og["px_cur"].reshape(-1, 2)[ok] - px_true.T[ok], axis=1)
        assert np.median(e) < 0.3
    # the affine entry is untouched by the new argument, and the pixelwise entry refuses a NULL landmark array
    import ctypes as C
    rv = (capi.svoh_frame_view * 1)(gv_r)
    rc = gpu_ctx.lib.svoh_match_direct_batch_pixelwise(gpu_ctx.h, C.byref(mopt), 1, rv, C.byref(gv_c), C.byref(fbg), None, None,
                                                       None, None, None, None, None, None)
    assert rc != 0


@pytest.mark.parametrize("n_features,cam_kind,sphere", [(120, "pinhole", 1), (3000, "radtan", 1), (3000, "pinhole", 0)])
def test_epipolar_match_batch_parity_stereo_seam(gpu_ctx, oracle_lib, n_features, cam_kind, sphere):
    """Row *J: n x Matcher::findEpipolarMatchDirect(frame0, frame1, T_f1f0, ftr, mean / min / max inverse
    depth, depth) with max_epi_search_steps = 500 and align_1d = isEdgelet(type), the call of
    StereoTriangulation::compute (stereo_triangulation.cpp:92-104), against the oracle: result codes, search levels
    exact, depth relative 1e-9, matched pixel <= 1e-4."""
    orc = oracle_lib
    cam = synth.Camera.test_camera() if cam_kind == "pinhole" else synth.Camera.euroc_like()
    sc, ref, cur, fr, fc = scene_and_frames(gpu_ctx, orc, 66, cam)     # "left" = ref, "right" = cur (5-15 cm baseline)
    sd = synth.make_seed_set(sc, n_features, margin=6, levels=(0, 1, 2))
    ftype = np.where(sd["type"] == 0, capi.FT_EDGELET, capi.FT_CORNER).astype(np.uint8)   # detector output, not seeds
    ov_r, ov_c, gv_r, gv_c = views(gpu_ctx, orc, sc, ref, cur, fr, fc, 0.0)
    T_f1f0 = (sc.T_cur_f_w_gt * sc.T_ref_f_w.inverse()).as7()
    d_mean = float(np.median(sd["true_depth"]))
    d_inv = [1.0 / d_mean, 1.0 / (0.3 * d_mean), 1.0 / (15.0 * d_mean)]   # StereoTriangulationOptions: mean / min / max
    mopt = capi.default_matcher_options(max_epi_search_steps=500, subpix_refinement=1, scan_on_unit_sphere=sphere)
    fbo, ko = orc.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], ftype)
    fbg, kg = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], ftype)
    oo = orc.epipolar_match_batch(mopt, [ov_r], ov_c, fbo, d_inv_common=d_inv, T_cur_ref=[T_f1f0])
    og = gpu_ctx.epipolar_match_batch(mopt, [gv_r], gv_c, fbg, d_inv_common=d_inv, T_cur_ref=[T_f1f0])
    assert np.array_equal(oo["result"], og["result"]), np.nonzero(oo["result"] != og["result"])
    assert np.array_equal(oo["search_level"], og["search_level"])
    ok = oo["result"] == capi.MATCH_SUCCESS
    assert ok.mean() > 0.5 and len(set(oo["result"])) >= 3
    assert np.allclose(og["depth"][ok], oo["depth"][ok], rtol=1e-9, atol=0)
    assert np.abs(oo["px_cur"] - og["px_cur"])[np.repeat(ok, 2)].max() <= 1e-4
    assert np.abs(oo["f_cur"] - og["f_cur"])[np.repeat(ok, 3)].max() < 1e-6
    assert np.allclose(oo["A"], og["A"], rtol=1e-12, atol=1e-14)
    e = np.abs(og["depth"][ok] - sd["true_depth"][ok]) / sd["true_depth"][ok]
    assert np.median(e) < 0.03                              # the triangulated depth is the scene's
    # per-feature depth ranges (3 x n) and the pose-derived transform (NULL) give the same answers here
    og2 = gpu_ctx.epipolar_match_batch(mopt, [gv_r], gv_c, fbg, d_inv=np.tile(d_inv, n_features))
    assert np.array_equal(og2["result"], og["result"]) and np.allclose(og2["depth"], og["depth"], rtol=1e-9, atol=0)


def test_golden_klt_seeds_fixture(gpu_ctx):
    """HIP path vs the committed fixture (no oracle call)."""
    import os
    z = np.load(os.path.join(os.path.dirname(helpers.GOLDEN), "klt_seeds_small.npz"))
    c = z["cam"]
    cam = synth.Camera(int(c[0]), int(c[1]), c[2], c[3], c[4], c[5], dist=list(c[6:10]))
    T_ref, T_cur = synth.SE3.from7(z["T_ref_f_w"]), synth.SE3.from7(z["T_cur_f_w"])
    fr = gpu_ctx.build_pyramid(z["img_ref"], 4); fc = gpu_ctx.build_pyramid(z["img_cur"], 4)
    kopt = capi.default_klt_options(max_level=3, patch_sizes=[16, 16, 8, 8])
    p, s = gpu_ctx.klt_track_batch(kopt, fr, fc, z["klt_px_ref"], z["klt_px_init"])
    assert np.array_equal(s, z["klt_status"]) and np.array_equal(p, z["klt_px_out"])
    mopt = capi.default_matcher_options()
    dopt = capi.default_depth_filter_options(px_error_angle=float(z["seed_px_error_angle"][0]))
    rv = fe.make_frame_view(fr, cam, T_ref, float(z["seed_mu_range"][0]), 1)
    cv = fe.make_frame_view(fc, cam, T_cur, 0.0, 2)
    n = z["seed_level"].size
    fb, keep = fe.make_feature_batch(np.zeros(n, np.int32), z["seed_px"], z["seed_f"], z["seed_grad"], z["seed_level"], z["seed_type_in"])
    ns, st, succ, mr = gpu_ctx.update_seeds_batch(mopt, dopt, [rv], cv, fb, z["seed_state_in"])
    assert np.array_equal(succ, z["seed_success"]) and np.array_equal(mr, z["seed_match_result"])
    assert np.array_equal(keep["type"], z["seed_type_out"]) and np.allclose(st, z["seed_state_out"], rtol=1e-9, atol=0)
    fb2, keep2 = fe.make_feature_batch(np.zeros(80, np.int32), z["seed_px"][:160], z["seed_f"][:240], z["seed_grad"][:160],
                                       z["seed_level"][:80], z["direct_type"])
    o = gpu_ctx.match_direct_batch(mopt, [rv], cv, fb2, z["direct_depth"], z["direct_px_init"])
    assert np.array_equal(o["result"], z["direct_result"]) and np.array_equal(o["search_level"], z["direct_search_level"])
    assert np.abs(o["px_cur"] - z["direct_px_out"]).max() < 1e-4


def test_golden_stereo_fixture(gpu_ctx):
    """HIP path vs tests/golden/stereo_small.npz (no oracle call): the stereo seam's epipolar matches."""
    import os
    z = np.load(os.path.join(os.path.dirname(helpers.GOLDEN), "klt_seeds_small.npz"))
    s = np.load(os.path.join(os.path.dirname(helpers.GOLDEN), "stereo_small.npz"))
    c = z["cam"]
    cam = synth.Camera(int(c[0]), int(c[1]), c[2], c[3], c[4], c[5], dist=list(c[6:10]))
    T_ref, T_cur = synth.SE3.from7(z["T_ref_f_w"]), synth.SE3.from7(z["T_cur_f_w"])
    fr = gpu_ctx.build_pyramid(z["img_ref"], 4); fc = gpu_ctx.build_pyramid(z["img_cur"], 4)
    n = int(s["n"][0])
    mopt = capi.default_matcher_options(max_epi_search_steps=500, subpix_refinement=1, scan_on_unit_sphere=1)
    fb, keep = fe.make_feature_batch(np.zeros(n, np.int32), z["seed_px"][:2 * n], z["seed_f"][:3 * n], z["seed_grad"][:2 * n],
                                     z["seed_level"][:n], s["type"])
    o = gpu_ctx.epipolar_match_batch(mopt, [fe.make_frame_view(fr, cam, T_ref, 0.0, 1)], fe.make_frame_view(fc, cam, T_cur, 0.0, 2), fb,
                                     d_inv_common=list(s["d_inv"]), T_cur_ref=[s["T_f1f0"]])
    assert np.array_equal(o["result"], s["result"]) and np.array_equal(o["search_level"], s["search_level"])
    ok = s["result"] == 0
    assert np.allclose(o["depth"][ok], s["depth"][ok], rtol=1e-9, atol=0)
    assert np.abs(o["px_cur"] - s["px_cur"])[np.repeat(ok, 2)].max() <= 1e-4


def test_euroc_752x480_geometry_all_paths(gpu_ctx, oracle_lib):
    """EuRoC's real geometry (752x480, radtan): the pyramid mixes the SSE2 and the scalar
    halfSample rule (376 % 16 != 0) and level widths become odd (47); every path must agree with the oracle."""
    orc = oracle_lib
    cam = synth.Camera(752, 480, 458.654, 457.296, 367.215, 248.375, dist=[-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05])
    sc = synth.make_align_scene(71, n_features=600, cam=cam, border_features=80, rot_deg=(0.3, 1.0), trans_m=(0.04, 0.1))
    ref = orc.create_img_pyramid(sc.img_ref, 5); cur = orc.create_img_pyramid(sc.img_cur, 5)
    assert [l.shape[1] for l in ref] == [752, 376, 188, 94, 47]
    fr, lv = gpu_ctx.build_pyramid(sc.img_ref, 5, return_levels=True)
    fc, lvc = gpu_ctx.build_pyramid(sc.img_cur, 5, return_levels=True)
    for a, b in zip(lv + lvc, ref + cur):
        assert np.array_equal(a, b)
    # sparse image alignment with the frame handler's levels (4..2) and with all levels
    for kw in (dict(min_level=2), dict(min_level=0, robustification=1)):
        opt = capi.default_align_options(**kw)
        n, ro, _ = orc.sparse_align_run(opt, orc.problem_from_scenes([(sc, ref, cur)]))
        pbs, keep = fe.make_align_problems([[(sc, fr, fc)]])
        rg = gpu_ctx.sparse_align(opt, pbs)[0]
        assert rg.n_fts_to_track == n and list(rg.iters) == list(ro.iters) and list(rg.n_meas) == list(ro.n_meas)
        assert helpers.se3_max_abs_diff(rg.T_icur_iref, ro.T_icur_iref) < 1e-8
    # KLT
    tr = synth.make_track_set(sc, 300, margin=10)
    kopt = capi.default_klt_options()
    po, so = orc.klt_track_batch(kopt, ref, cur, tr["px_ref"], tr["px_cur_init"])
    pg, sg = gpu_ctx.klt_track_batch(kopt, fr, fc, tr["px_ref"], tr["px_cur_init"])
    assert np.array_equal(so, sg) and np.array_equal(po, pg) and so.mean() > 0.8
    # seeds
    sd = synth.make_seed_set(sc, 1500, margin=10)
    ov_r, ov_c, gv_r, gv_c = views(gpu_ctx, orc, sc, ref, cur, fr, fc, sd["mu_range"])
    mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(cam)
    fbo, ko = orc.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
    fbg, kg = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
    nso, sto, so2, mro = orc.update_seeds_batch(mopt, dopt, [ov_r], ov_c, fbo, sd["state"])
    nsg, stg, sg2, mrg = gpu_ctx.update_seeds_batch(mopt, dopt, [gv_r], gv_c, fbg, sd["state"])
    assert nso == nsg and np.array_equal(mro, mrg) and np.array_equal(ko["type"], kg["type"])
    assert np.allclose(stg, sto, rtol=1e-9, atol=0) and nso > 700


def test_multi_stream_batch_equals_per_frame_calls(gpu_ctx, oracle_lib):
    """cur_frame_idx batching (several camera streams in one launch) gives exactly the per-frame results."""
    orc = oracle_lib
    packs = [scene_and_frames(gpu_ctx, orc, 80 + k) for k in range(3)]
    mopt = capi.default_matcher_options()
    seeds = [synth.make_seed_set(p[0], 400, seed=k) for k, p in enumerate(packs)]
    dopt = capi.default_depth_filter_options(packs[0][0].cam)
    singles = []
    for k, (sc, ref, cur, fr, fc) in enumerate(packs):
        sd = seeds[k]
        rv = fe.make_frame_view(fr, sc.cam, sc.T_ref_f_w, sd["mu_range"], 2 * k)
        cv = fe.make_frame_view(fc, sc.cam, sc.T_cur_f_w_gt, 0.0, 2 * k + 1)
        fb, kk = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
        singles.append(gpu_ctx.update_seeds_batch(mopt, dopt, [rv], cv, fb, sd["state"]) + (kk["type"].copy(),))
    import ctypes as C
    rvs = [fe.make_frame_view(p[3], p[0].cam, p[0].T_ref_f_w, seeds[k]["mu_range"], 2 * k) for k, p in enumerate(packs)]
    cvs = (capi.svoh_frame_view * 3)(*[fe.make_frame_view(p[4], p[0].cam, p[0].T_cur_f_w_gt, 0.0, 2 * k + 1) for k, p in enumerate(packs)])
    idx = np.repeat(np.arange(3, dtype=np.int32), 400)
    cat = lambda key: np.concatenate([s[key] for s in seeds])
    fb, kk = fe.make_feature_batch(idx, cat("px"), cat("f"), cat("grad"), cat("level"), cat("type"))
    fb.cur_frame_idx = idx.ctypes.data
    fb.n_cur_frames = 3
    st = cat("state").copy(); succ = np.zeros(1200, np.uint8); mr = np.zeros(1200, np.int32); ns = C.c_int32()
    rv_arr = (capi.svoh_frame_view * 3)(*rvs)
    gpu_ctx._check(gpu_ctx.lib.svoh_update_seeds_batch(gpu_ctx.h, C.byref(mopt), C.byref(dopt), 3, rv_arr, cvs, C.byref(fb),
                                                       st.ctypes.data, succ.ctypes.data, mr.ctypes.data, C.byref(ns)))
    assert ns.value == sum(s[0] for s in singles)
    for k, s in enumerate(singles):
        sl = slice(400 * k, 400 * (k + 1))
        assert np.array_equal(st[4 * 400 * k:4 * 400 * (k + 1)], s[1]) and np.array_equal(succ[sl], s[2])
        assert np.array_equal(mr[sl], s[3]) and np.array_equal(kk["type"][sl], s[4])
    # KLT: per-track current frames
    trs = [synth.make_track_set(p[0], 50, seed=k) for k, p in enumerate(packs)]
    kopt = capi.default_klt_options()
    rf = (capi.svoh_frame_t * 150)(*[packs[i // 50][3] for i in range(150)])
    cf = (capi.svoh_frame_t * 150)(*[packs[i // 50][4] for i in range(150)])
    pr = np.concatenate([t["px_ref"] for t in trs]); p0 = np.concatenate([t["px_cur_init"] for t in trs])
    out = p0.copy(); stt = np.zeros(150, np.uint8)
    gpu_ctx._check(gpu_ctx.lib.svoh_klt_track_multi(gpu_ctx.h, C.byref(kopt), 150, rf, cf, pr.ctypes.data, out.ctypes.data, stt.ctypes.data))
    for k, p in enumerate(packs):
        pg, sg = gpu_ctx.klt_track_batch(kopt, p[3], p[4], trs[k]["px_ref"], trs[k]["px_cur_init"])
        assert np.array_equal(pg, out[100 * k:100 * (k + 1)]) and np.array_equal(sg, stt[50 * k:50 * (k + 1)])


def test_device_resident_batches_equal_host_batches(gpu_ctx, oracle_lib):
    """mem_space = SVOH_MEM_DEVICE (arrays used in place, stream-ordered) gives bit-identical results to the
    staged host-pointer calls, for the seed update, the direct matcher and the indexed KLT entry; out-of-range
    indices in a device batch mark the unit instead of faulting."""
    import torch
    dev = torch.device("cuda", 0)
    orc = oracle_lib
    packs = [scene_and_frames(gpu_ctx, orc, 90 + k) for k in range(2)]
    mopt = capi.default_matcher_options()
    dopt = capi.default_depth_filter_options(packs[0][0].cam)
    seeds = [synth.make_seed_set(p[0], 500, seed=10 + k) for k, p in enumerate(packs)]
    rvs = [fe.make_frame_view(p[3], p[0].cam, p[0].T_ref_f_w, seeds[k]["mu_range"], 2 * k) for k, p in enumerate(packs)]
    cvs = [fe.make_frame_view(p[4], p[0].cam, p[0].T_cur_f_w_gt, 0.0, 2 * k + 1) for k, p in enumerate(packs)]
    n = 1000
    idx = np.repeat(np.arange(2, dtype=np.int32), 500)
    cat = lambda key: np.concatenate([s[key] for s in seeds])
    # host reference (multi-stream host batch)
    fb, kk = fe.make_feature_batch(idx, cat("px"), cat("f"), cat("grad"), cat("level"), cat("type"))
    fb.cur_frame_idx = idx.ctypes.data
    fb.n_cur_frames = 2
    import ctypes as C
    st = cat("state").copy(); succ = np.zeros(n, np.uint8); mr = np.zeros(n, np.int32); ns = C.c_int32()
    gpu_ctx._check(gpu_ctx.lib.svoh_update_seeds_batch(
        gpu_ctx.h, C.byref(mopt), C.byref(dopt), 2, (capi.svoh_frame_view * 2)(*rvs), (capi.svoh_frame_view * 2)(*cvs),
        C.byref(fb), st.ctypes.data, succ.ctypes.data, mr.ctypes.data, C.byref(ns)))
    # device batch; two extra features with bad indices at the end
    bad_idx = np.concatenate([idx, np.array([7, 0], np.int32)])
    bad_lvl = np.concatenate([cat("level").astype(np.int32), np.array([0, 99], np.int32)])
    pad = lambda a, k: np.concatenate([a, a[:k * 2]])
    t = dict(idx=torch.from_numpy(bad_idx).to(dev), px=torch.from_numpy(pad(cat("px"), 2)).to(dev),
             f=torch.from_numpy(pad(cat("f"), 3)).to(dev), grad=torch.from_numpy(pad(cat("grad"), 2)).to(dev),
             level=torch.from_numpy(bad_lvl).to(dev), type=torch.from_numpy(pad(cat("type").astype(np.uint8), 1)).to(dev),
             state=torch.from_numpy(pad(cat("state"), 4)).to(dev), succ=torch.full((n + 2,), 9, dtype=torch.uint8, device=dev),
             mr=torch.zeros(n + 2, dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    fbd = fe.make_feature_batch_device(n + 2, t["idx"].data_ptr(), t["px"].data_ptr(), t["f"].data_ptr(),
                                       t["grad"].data_ptr(), t["level"].data_ptr(), t["type"].data_ptr(),
                                       cur_frame_idx=t["idx"].data_ptr(), n_cur_frames=2)
    cnt = gpu_ctx.update_seeds_device(mopt, dopt, rvs, cvs, fbd, t["state"].data_ptr(), t["succ"].data_ptr(),
                                      t["mr"].data_ptr(), want_count=True)
    gpu_ctx.synchronize()
    assert cnt == ns.value
    assert np.array_equal(t["state"].cpu().numpy()[:4 * n], st)
    assert np.array_equal(t["succ"].cpu().numpy()[:n], succ) and np.array_equal(t["mr"].cpu().numpy()[:n], mr)
    assert np.array_equal(t["type"].cpu().numpy()[:n], kk["type"])
    assert t["succ"].cpu().numpy()[n:].tolist() == [0, 0]
    assert t["mr"].cpu().numpy()[n:].tolist() == [capi.MATCH_NOT_RUN] * 2
    # without the count the call does not synchronise; results are stream-ordered
    t["state"].copy_(torch.from_numpy(pad(cat("state"), 4)).to(dev)); t["type"].copy_(torch.from_numpy(pad(cat("type").astype(np.uint8), 1)).to(dev))
    torch.cuda.synchronize()
    assert gpu_ctx.update_seeds_device(mopt, dopt, rvs, cvs, fbd, t["state"].data_ptr(), t["succ"].data_ptr()) is None
    gpu_ctx.synchronize()
    assert np.array_equal(t["state"].cpu().numpy()[:4 * n], st)

    # direct matcher
    sc, ref, cur, fr, fc = packs[0]
    ms = synth.make_seed_set(sc, 300, seed=3)
    depth = np.ascontiguousarray(ms["true_depth"], np.float64)
    types = np.where(ms["type"] == capi.FT_EDGELET_SEED, capi.FT_EDGELET, capi.FT_CORNER).astype(np.uint8)
    x = ms["f"].reshape(-1, 3).T * ms["true_depth"]
    px_true = sc.cam.project(sc.T_w_cur.inverse().transform(sc.T_w_ref.transform(x)))
    px0 = np.ascontiguousarray((px_true + np.random.RandomState(2).uniform(-2.0, 2.0, px_true.shape)).T).ravel()
    fbh, _k = fe.make_feature_batch(ms["ref_frame_idx"], ms["px"], ms["f"], ms["grad"], ms["level"], types)
    host = gpu_ctx.match_direct_batch(mopt, [rvs[0]], cvs[0], fbh, depth, px0)
    d = dict(idx=torch.from_numpy(ms["ref_frame_idx"].astype(np.int32)).to(dev), px=torch.from_numpy(ms["px"]).to(dev),
             f=torch.from_numpy(ms["f"]).to(dev), grad=torch.from_numpy(ms["grad"]).to(dev),
             level=torch.from_numpy(ms["level"].astype(np.int32)).to(dev), type=torch.from_numpy(types).to(dev),
             depth=torch.from_numpy(depth).to(dev), pxc=torch.from_numpy(px0).to(dev),
             res=torch.zeros(300, dtype=torch.int32, device=dev), fcur=torch.zeros(900, dtype=torch.float64, device=dev),
             A=torch.zeros(1200, dtype=torch.float64, device=dev))
    torch.cuda.synchronize()
    fbm = fe.make_feature_batch_device(300, d["idx"].data_ptr(), d["px"].data_ptr(), d["f"].data_ptr(), d["grad"].data_ptr(),
                                       d["level"].data_ptr(), d["type"].data_ptr())
    gpu_ctx.match_direct_device(mopt, [rvs[0]], cvs[0], fbm, d["depth"].data_ptr(), d["pxc"].data_ptr(), d["res"].data_ptr(),
                                f_cur=d["fcur"].data_ptr(), A_cur_ref=d["A"].data_ptr())
    gpu_ctx.synchronize()
    assert np.array_equal(d["res"].cpu().numpy(), host["result"]) and np.array_equal(d["pxc"].cpu().numpy(), host["px_cur"])
    assert np.array_equal(d["fcur"].cpu().numpy(), host["f_cur"]) and np.array_equal(d["A"].cpu().numpy(), host["A"])
    # the pixelwise warp's entry with device-resident arrays == with host arrays
    lm = np.ascontiguousarray(sc.T_w_ref.transform(x).T)
    host_pw = gpu_ctx.match_direct_batch(mopt, [rvs[0]], cvs[0], fbh, depth, px0, landmark_xyz=lm)
    d["pxc"].copy_(torch.from_numpy(px0).to(dev)); d["res"].zero_(); d["fcur"].zero_(); d["A"].zero_()
    d["lm"] = torch.from_numpy(lm).to(dev)
    torch.cuda.synchronize()
    gpu_ctx.match_direct_device(mopt, [rvs[0]], cvs[0], fbm, d["depth"].data_ptr(), d["pxc"].data_ptr(), d["res"].data_ptr(),
                                f_cur=d["fcur"].data_ptr(), A_cur_ref=d["A"].data_ptr(), landmark_xyz=d["lm"].data_ptr())
    gpu_ctx.synchronize()
    assert np.array_equal(d["res"].cpu().numpy(), host_pw["result"]) and np.array_equal(d["pxc"].cpu().numpy(), host_pw["px_cur"])
    assert np.array_equal(d["fcur"].cpu().numpy(), host_pw["f_cur"]) and (host_pw["result"] == 0).sum() > 100
    assert not np.array_equal(host_pw["px_cur"], host["px_cur"])        # it is the other warp

    # indexed KLT: host mode == multi, device mode == host mode, bad index -> status 0
    trs = [synth.make_track_set(p[0], 60, seed=20 + k) for k, p in enumerate(packs)]
    kopt = capi.default_klt_options()
    frames = [packs[0][3], packs[0][4], packs[1][3], packs[1][4]]
    ridx = np.repeat(np.array([0, 2], np.int32), 60); cidx = ridx + 1
    pr = np.concatenate([tr["px_ref"] for tr in trs]).astype(np.int32); p0 = np.concatenate([tr["px_cur_init"] for tr in trs])
    outs, sts = [], []
    for k, p in enumerate(packs):
        pg, sg = gpu_ctx.klt_track_batch(kopt, p[3], p[4], trs[k]["px_ref"], trs[k]["px_cur_init"])
        outs.append(pg); sts.append(sg)
    want_px, want_st = np.concatenate(outs), np.concatenate(sts)
    hp = p0.copy(); hs = np.zeros(120, np.uint8)
    gpu_ctx.klt_track_indexed(kopt, frames, 120, ridx.ctypes.data, cidx.ctypes.data, pr.ctypes.data, hp.ctypes.data,
                              hs.ctypes.data, mem_space=capi.SVOH_MEM_HOST)
    assert np.array_equal(hp, want_px) and np.array_equal(hs, want_st)
    ridx_bad = np.concatenate([ridx, np.array([4], np.int32)]); cidx_bad = np.concatenate([cidx, np.array([1], np.int32)])
    k_t = dict(r=torch.from_numpy(ridx_bad).to(dev), c=torch.from_numpy(cidx_bad).to(dev),
               pr=torch.from_numpy(np.concatenate([pr, pr[:2]])).to(dev), pc=torch.from_numpy(np.concatenate([p0, p0[:2]])).to(dev),
               st=torch.full((121,), 7, dtype=torch.uint8, device=dev))
    torch.cuda.synchronize()
    gpu_ctx.klt_track_indexed(kopt, frames, 121, k_t["r"].data_ptr(), k_t["c"].data_ptr(), k_t["pr"].data_ptr(),
                              k_t["pc"].data_ptr(), k_t["st"].data_ptr())
    gpu_ctx.synchronize()
    assert np.array_equal(k_t["pc"].cpu().numpy()[:240], want_px) and np.array_equal(k_t["st"].cpu().numpy()[:120], want_st)
    assert int(k_t["st"][120]) == 0
    # host-mode validation still fails the call
    with pytest.raises(fe.SvohError):
        gpu_ctx.klt_track_indexed(kopt, frames, 121, ridx_bad.ctypes.data, cidx_bad.ctypes.data, pr.ctypes.data,
                                  hp.ctypes.data, hs.ctypes.data, mem_space=capi.SVOH_MEM_HOST)


def test_benchmark_size_batch_is_identical_in_every_geometry(gpu_ctx):
    """BASELINE's C4 size through size-independent properties: 24 (keyframe, frame) pairs x 3000 seeds = 72 000 seeds in
    ONE launch (beyond the eight-lane threshold, so the library itself picks the packed geometry with spatial binning)
    must give bit for bit what the one-lane and the eight-lane kernels give, with and without binning, and what 24
    separate launches give (no seed depends on its neighbours, its place in the batch or the processing order)."""
    import os
    B, NS = 24, 3000
    views_r, views_c, parts = [], [], []
    for b in range(B):
        sc = synth.make_align_scene(900 + b, n_features=8, rot_deg=(0.3, 1.0), trans_m=(0.05, 0.15))
        fr, fc = gpu_ctx.build_pyramid(sc.img_ref, 5), gpu_ctx.build_pyramid(sc.img_cur, 5)
        sd = synth.make_seed_set(sc, NS, seed=b)
        views_r.append(fe.make_frame_view(fr, sc.cam, sc.T_ref_f_w, sd["mu_range"], 2 * b))
        views_c.append(fe.make_frame_view(fc, sc.cam, sc.T_cur_f_w_gt, 0.0, 2 * b + 1))
        parts.append((sc, sd))
    cat = lambda k: np.concatenate([p[1][k] for p in parts])
    idx = np.repeat(np.arange(B, dtype=np.int32), NS)
    mopt = capi.default_matcher_options()
    dopt = capi.default_depth_filter_options(parts[0][0].cam)

    def run():
        fb, keep = fe.make_feature_batch(idx, cat("px"), cat("f"), cat("grad"), cat("level"), cat("type"))
        fb.cur_frame_idx = idx.ctypes.data
        fb.n_cur_frames = B
        ns, st, succ, mr = gpu_ctx.update_seeds_batch(mopt, dopt, views_r, views_c, fb, cat("state"))
        return ns, st, succ, mr, keep["type"].copy()

    old = {k: os.environ.get(k) for k in ("SVOH_MATCHER_G8", "SVOH_SEED_BINNING")}
    try:
        out = {}
        for name, g8, binning in (("auto", None, None), ("one_lane", "0", None), ("eight_lanes", "1", None), ("packed_unbinned", "2", "0")):
            for k, v in (("SVOH_MATCHER_G8", g8), ("SVOH_SEED_BINNING", binning)):
                if v is None:
                    os.environ.pop(k, None)
                    gpu_ctx.reload_knobs()
                else:
                    os.environ[k] = v
                    gpu_ctx.reload_knobs()
            out[name] = run()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
                gpu_ctx.reload_knobs()
            else:
                os.environ[k] = v
                gpu_ctx.reload_knobs()
    ref = out["one_lane"]
    assert ref[0] > 0.5 * B * NS
    for name, o in out.items():
        assert o[0] == ref[0], name
        for a, b in zip(o[1:], ref[1:]):
            assert np.array_equal(a, b), name
    # one pair alone (a small batch: eight lanes per seed) gives the batch's rows
    b = 7
    sd = parts[b][1]
    fb1, keep1 = fe.make_feature_batch(np.zeros(NS, np.int32), sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
    ns1, st1, succ1, mr1 = gpu_ctx.update_seeds_batch(mopt, dopt, [views_r[b]], views_c[b], fb1, sd["state"])
    sl = slice(b * NS, (b + 1) * NS)
    assert np.array_equal(st1, ref[1][4 * b * NS:4 * (b + 1) * NS]) and np.array_equal(succ1, ref[2][sl]) and np.array_equal(mr1, ref[3][sl])


def test_direct_and_epipolar_outputs_identical_in_every_geometry(gpu_ctx):
    """Every output array of svoh_match_direct_batch and svoh_epipolar_match_batch -- result codes, pixels, bearing
    vectors, search levels, h_inv, warp matrices, depths; successes and every kind of failure -- is the same bit for bit
    in the three geometries (the packed kernels of round 3 included)."""
    import os
    cam = synth.Camera.euroc_like()
    sc = synth.make_align_scene(71, n_features=10, cam=cam, rot_deg=(0.5, 1.5), trans_m=(0.05, 0.15))
    fr, fc = gpu_ctx.build_pyramid(sc.img_ref, 5), gpu_ctx.build_pyramid(sc.img_cur, 5)
    sd = synth.make_seed_set(sc, 3000, margin=3, levels=(0, 1, 2, 3))
    rv = fe.make_frame_view(fr, sc.cam, sc.T_ref_f_w, float(sd["mu_range"]), 1)
    cv = fe.make_frame_view(fc, sc.cam, sc.T_cur_f_w_gt, 0.0, 2)
    x = sd["f"].reshape(-1, 3).T * sd["true_depth"]
    px_true = sc.cam.project(sc.T_w_cur.inverse().transform(sc.T_w_ref.transform(x)))
    px_init = np.ascontiguousarray((px_true + np.random.RandomState(1).uniform(-2.0, 2.0, px_true.shape)).T).ravel()
    px_init[:40] += 40.0
    ftype = np.where(sd["type"] == 0, capi.FT_EDGELET, capi.FT_CORNER).astype(np.uint8)
    dm = float(np.median(sd["true_depth"]))
    old = os.environ.get("SVOH_MATCHER_G8")
    outs = {}
    try:
        for g in ("1", "0", "2", "3"):
            os.environ["SVOH_MATCHER_G8"] = g
            gpu_ctx.reload_knobs()
            for mkw in (dict(), dict(affine_est_gain=1, scan_on_unit_sphere=0)):
                mopt = capi.default_matcher_options(max_epi_search_steps=500, **mkw)
                fb, kk = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], ftype)
                d = gpu_ctx.match_direct_batch(mopt, [rv], cv, fb, sd["true_depth"], px_init)
                e = gpu_ctx.epipolar_match_batch(mopt, [rv], cv, fb, d_inv_common=[1 / dm, 3 / dm, 0.05 / dm])
                outs[(g, tuple(sorted(mkw)))] = (d, e)
    finally:
        if old is None:
            os.environ.pop("SVOH_MATCHER_G8", None)
        else:
            os.environ["SVOH_MATCHER_G8"] = old
        gpu_ctx.reload_knobs()
    for key in [k for k in outs if k[0] == "1"]:
        for g in ("0", "2", "3"):
            for a, b in zip(outs[key], outs[(g, key[1])]):
                for name in a:
                    assert np.array_equal(a[name], b[name]), (g, key, name, int(np.sum(a[name] != b[name])))
        d, e = outs[key]
        assert len(set(d["result"].tolist())) >= 3 and len(set(e["result"].tolist())) >= 4   # successes and several failure kinds
