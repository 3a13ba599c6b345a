"""Inputs the reference would crash on (float -> int overflow, NaN indices: undefined behaviour in
sparse_img_align.cpp:217-225, patch_warp.cpp:128-145) must not fault a GPU: wild units are rejected or fail
like any other invisible / unmatched unit, the call returns, and the context keeps working.  No oracle here:
the CPU restatement shares the reference's undefined behaviour on these inputs."""
import numpy as np
import pytest

from svo_pro_universal_amd import _capi as capi, frontend as fe, synth

pytestmark = pytest.mark.gpu

WILD = [1e30, -1e30, np.inf, -np.inf, np.nan, 2.0 ** 31, -2.0 ** 31 - 1.0, 1e9]


def test_sparse_align_wild_features(gpu_ctx):
    sc = synth.make_align_scene(301, n_features=64)
    fr, fc = gpu_ctx.build_pyramid(sc.img_ref, 5), gpu_ctx.build_pyramid(sc.img_cur, 5)
    opt = capi.default_align_options()
    def run(scene, rf, cf, **kw):
        problems, keep = fe.make_align_problems([[(scene, rf, cf)]], **kw)   # `keep` owns the host arrays
        return gpu_ctx.sparse_align(opt, problems)[0]

    base = run(sc, fr, fc)
    assert base.status == 0 and base.n_fts_to_track == 64
    # wild pixel coordinates: rejected by the a-3 selection test
    for k, w in enumerate(WILD):
        sc.px = sc.px.copy()
        sc.px[2 * k] = w
        sc.px[2 * k + 1] = WILD[(k + 3) % len(WILD)]
    sc.px[2 * 20] = np.nan                               # NaN x with a valid y
    r = run(sc, fr, fc)
    assert r.n_fts_to_track <= 64 - len(WILD) + 2      # NaN px passes the reference's selection test; the rest do not
    assert r.status in (0, 2)
    # wild 3-D positions / bearing vectors: every comparison with NaN is false in the reference
    sc2 = synth.make_align_scene(302, n_features=64)
    f2r, f2c = gpu_ctx.build_pyramid(sc2.img_ref, 5), gpu_ctx.build_pyramid(sc2.img_cur, 5)
    sc2.pos_world = sc2.pos_world.copy(); sc2.f = sc2.f.copy()
    sc2.pos_world[0:3] = np.nan; sc2.pos_world[3:6] = 1e300; sc2.pos_world[6:9] = -1e300
    sc2.f[9:12] = np.nan; sc2.f[12:15] = 0.0; sc2.f[15:18] = np.inf
    r2 = run(sc2, f2r, f2c)
    assert r2.status in (0, 2)
    # wild initial pose and prior
    T_bad = synth.SE3((np.nan, 0.0, 0.0, 0.0), (1e300, -1e300, np.inf))
    r3 = run(sc, fr, fc, T_init=T_bad)
    assert r3.status in (0, 2)
    # the context is still healthy and deterministic
    sc_ok = synth.make_align_scene(301, n_features=64)
    again = run(sc_ok, fr, fc)
    assert np.array_equal(fe.se3_to_numpy(again.T_icur_iref), fe.se3_to_numpy(base.T_icur_iref))


def test_klt_wild_tracks(gpu_ctx):
    sc = synth.make_align_scene(303, n_features=8)
    fr, fc = gpu_ctx.build_pyramid(sc.img_ref, 5), gpu_ctx.build_pyramid(sc.img_cur, 5)
    tr = synth.make_track_set(sc, 64)
    px_ref = tr["px_ref"].copy().astype(np.int32); px0 = tr["px_cur_init"].copy()
    ints = [2 ** 31 - 1, -2 ** 31, 10 ** 9, -10 ** 9, 0, -1, 640, 480]
    for k, w in enumerate(WILD):
        px0[2 * k] = w
        px0[2 * k + 1] = WILD[(k + 5) % len(WILD)]
        px_ref[2 * (k + 8)] = ints[k]
        px_ref[2 * (k + 8) + 1] = ints[(k + 3) % len(ints)]
    out, st = gpu_ctx.klt_track_batch(capi.default_klt_options(), fr, fc, px_ref, px0)
    assert not st[:8].any()                      # wild start positions never converge
    good, st_good = gpu_ctx.klt_track_batch(capi.default_klt_options(), fr, fc, tr["px_ref"], tr["px_cur_init"])
    assert np.array_equal(out[2 * 16:], good[2 * 16:]) and np.array_equal(st[16:], st_good[16:])   # the others are untouched


def test_matcher_and_seeds_wild_units(gpu_ctx):
    sc = synth.make_align_scene(304, n_features=8, rot_deg=(0.5, 1.5), trans_m=(0.05, 0.15))
    fr, fc = gpu_ctx.build_pyramid(sc.img_ref, 5), gpu_ctx.build_pyramid(sc.img_cur, 5)
    sd = synth.make_seed_set(sc, 256)
    rv = fe.make_frame_view(fr, sc.cam, sc.T_ref_f_w, sd["mu_range"], 1)
    cv = fe.make_frame_view(fc, sc.cam, sc.T_cur_f_w_gt, 0.0, 2)
    mopt = capi.default_matcher_options()
    ref = {}
    for check_vis in (1, 0):
        dopt = capi.default_depth_filter_options(sc.cam, check_visibility=check_vis)
        fb, kk = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
        ref[check_vis] = gpu_ctx.update_seeds_batch(mopt, dopt, [rv], cv, fb, sd["state"])
    px, f, grad, state = sd["px"].copy(), sd["f"].copy(), sd["grad"].copy(), sd["state"].copy()
    for k, w in enumerate(WILD):
        px[2 * k] = w; px[2 * k + 1] = WILD[(k + 2) % len(WILD)]
        f[3 * (k + 8):3 * (k + 8) + 3] = [w, WILD[(k + 1) % len(WILD)], 1.0]
        grad[2 * (k + 16):2 * (k + 16) + 2] = [w, 0.0]
        state[4 * (k + 24):4 * (k + 24) + 4] = [w, 1.0, 10.0, 10.0]          # wild inverse depth
        state[4 * (k + 32):4 * (k + 32) + 4] = [0.5, w, 10.0, 10.0]          # wild variance
        state[4 * (k + 40):4 * (k + 40) + 4] = [0.0, 0.0, 0.0, 0.0]
    for check_vis in (1, 0):
        dopt = capi.default_depth_filter_options(sc.cam, check_visibility=check_vis)
        fb, kk = fe.make_feature_batch(sd["ref_frame_idx"], px, f, grad, sd["level"], sd["type"])
        ns, st, succ, mr = gpu_ctx.update_seeds_batch(mopt, dopt, [rv], cv, fb, state)
        # units 48.. were left alone and behave exactly as before
        assert np.array_equal(st[4 * 48:], ref[check_vis][1][4 * 48:]) and np.array_equal(succ[48:], ref[check_vis][2][48:])
        assert not succ[:8].any()                # wild pixels never succeed
    # direct matcher with wild depth / start positions
    depth = sd["true_depth"].copy(); px_cur = sd["px"].copy()
    for k, w in enumerate(WILD):
        depth[k] = w
        px_cur[2 * (k + 8)] = w; px_cur[2 * (k + 8) + 1] = WILD[(k + 4) % len(WILD)]
    ftype = np.where(sd["type"] == 0, capi.FT_EDGELET, capi.FT_CORNER)
    fb, kk = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], ftype)
    out = gpu_ctx.match_direct_batch(mopt, [rv], cv, fb, depth, px_cur)
    assert (out["result"][8:16] != 0).all()       # wild starts cannot succeed
    # a camera that does not describe the frame is refused on the host, not discovered by a fault
    import copy
    cam_bad = copy.copy(sc.cam); cam_bad.width = sc.cam.width + 64
    with pytest.raises(fe.SvohError):
        fb, kk = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
        gpu_ctx.update_seeds_batch(mopt, capi.default_depth_filter_options(sc.cam), [fe.make_frame_view(fr, cam_bad, sc.T_ref_f_w, 1.0, 1)],
                                   cv, fb, sd["state"])
    # a current frame with fewer levels than the reference frame is refused too
    f_small = gpu_ctx.build_pyramid(sc.img_cur, 3)
    with pytest.raises(fe.SvohError):
        fb, kk = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
        gpu_ctx.update_seeds_batch(mopt, capi.default_depth_filter_options(sc.cam), [rv],
                                   fe.make_frame_view(f_small, sc.cam, sc.T_cur_f_w_gt, 0.0, 2), fb, sd["state"])
