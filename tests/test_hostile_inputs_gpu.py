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


def test_point_optimizer_wild_landmarks(gpu_ctx):
    """NaN / huge positions, bearing vectors and poses in svoh_optimize_points_batch: the call returns, wild
    landmarks end wherever their arithmetic leads, the well-formed ones next to them are untouched by it."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import pose_helpers as ph
    sc = ph.make_structure_scene(310, n_points=128, n_views=4, degenerate=False)
    good, it_good = gpu_ctx.optimize_points(sc["views"], sc["obs_begin"], sc["obs_view"], sc["obs_f"], sc["pos0"], n_iter=5)
    pos = sc["pos0"].copy()
    obs_f = sc["obs_f"].copy()
    views = [np.array(v, dtype=np.float64) for v in sc["views"]]
    for k, w in enumerate(WILD):
        pos[k] = [w, WILD[(k + 1) % len(WILD)], WILD[(k + 2) % len(WILD)]]
    wild_pts = list(range(len(WILD)))
    for k in (20, 21, 22):                       # wild bearing vectors on three landmarks
        obs_f[sc["obs_begin"][k]] = [np.nan, 1e300, 0.0]
        wild_pts.append(k)
    out, iters = gpu_ctx.optimize_points(views, sc["obs_begin"], sc["obs_view"], obs_f, pos, n_iter=5)
    clean = np.ones(128, bool)
    clean[wild_pts] = False
    assert np.array_equal(out[clean], good[clean]) and np.array_equal(iters[clean], it_good[clean])
    assert (iters >= 0).all() and (iters <= 5).all()
    # a NaN pose poisons exactly the landmarks that are observed from it
    views_bad = [v.copy() for v in views]
    views_bad[2][0] = np.nan
    out2, it2 = gpu_ctx.optimize_points(views_bad, sc["obs_begin"], sc["obs_view"], sc["obs_f"], sc["pos0"], n_iter=5)
    sees_bad = np.array([2 in sc["obs_view"][sc["obs_begin"][i]:sc["obs_begin"][i + 1]] for i in range(128)])
    assert np.array_equal(out2[~sees_bad], good[~sees_bad])
    # the context still works
    again, _ = gpu_ctx.optimize_points(sc["views"], sc["obs_begin"], sc["obs_view"], sc["obs_f"], sc["pos0"], n_iter=5)
    assert np.array_equal(again, good)


def test_split_alignment_wild_state_and_sums(gpu_ctx):
    """The patch-split entries with NaN / huge states and sums: no fault, the solver reports the failure
    (status 2, state rolled back) like the resident kernel does, and the context keeps working."""
    import ctypes as C
    import torch
    sc = synth.make_align_scene(311, n_features=200)
    fr, fc = gpu_ctx.build_pyramid(sc.img_ref, 5), gpu_ctx.build_pyramid(sc.img_cur, 5)
    opt = capi.default_align_options()
    problems, keep = fe.make_align_problems([[(sc, fr, fc)]])
    dev = torch.device("cuda", 0)
    d_state = torch.zeros(C.sizeof(capi.svoh_align_gn_state) // 8, dtype=torch.float64, device=dev)
    d_sums = torch.zeros(capi.SVOH_ALIGN_SUMS_DOUBLES, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.split_init(problems[0], d_state.data_ptr())
    gpu_ctx.partial_sums(opt, problems[0], 4, d_state.data_ptr(), d_sums.data_ptr())
    gpu_ctx.synchronize()
    ref_sums = d_sums.clone()
    # NaN in the summed gradient: the update reports a stopped solver and keeps the state (a NaN on the diagonal
    # of H alone would not: like Eigen's LDLT the solver treats an unusable pivot as zero)
    d_sums[64:70] = float("nan")
    torch.cuda.synchronize()
    st = gpu_ctx.gn_update(opt, problems[0], 4, 0, d_sums.data_ptr(), d_state.data_ptr())
    assert st.status == 2 and st.stop == 1 and st.level_done == 1
    assert [st.T_icur_iref.q[k] for k in range(4)] == [problems[0].T_icur_iref.q[k] for k in range(4)]
    # a wild state: the evaluation sees nothing (every comparison with NaN fails) and returns
    d_state.fill_(float("nan"))
    torch.cuda.synchronize()
    gpu_ctx.partial_sums(opt, problems[0], 4, d_state.data_ptr(), d_sums.data_ptr(), 3)
    gpu_ctx.synchronize()
    assert int(d_sums[73].item()) == 0
    # healthy again
    gpu_ctx.split_init(problems[0], d_state.data_ptr())
    gpu_ctx.partial_sums(opt, problems[0], 4, d_state.data_ptr(), d_sums.data_ptr())
    gpu_ctx.synchronize()
    assert torch.equal(d_sums, ref_sums)


@pytest.mark.parametrize("geometry", ["0", "1", "2"])
def test_seed_update_wild_units_in_every_geometry(gpu_ctx, geometry, monkeypatch):
    """The same wild seeds through the one-lane, the eight-lane and the packed geometry (whose binning pass turns the
    reference pixel into a tile index: NaN / inf / 1e30 must land in some tile, not index out of the histogram)."""
    monkeypatch.setenv("SVOH_MATCHER_G8", geometry)
    gpu_ctx.reload_knobs()
    sc = synth.make_align_scene(306, n_features=8, rot_deg=(0.5, 1.5), trans_m=(0.05, 0.15))
    fr, fc = gpu_ctx.build_pyramid(sc.img_ref, 5), gpu_ctx.build_pyramid(sc.img_cur, 5)
    sd = synth.make_seed_set(sc, 700)
    rv = fe.make_frame_view(fr, sc.cam, sc.T_ref_f_w, sd["mu_range"], 1)
    cv = fe.make_frame_view(fc, sc.cam, sc.T_cur_f_w_gt, 0.0, 2)
    mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(sc.cam)
    fb, kk = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
    ref = gpu_ctx.update_seeds_batch(mopt, dopt, [rv], cv, fb, sd["state"])
    px, state = sd["px"].copy(), sd["state"].copy()
    for k, w in enumerate(WILD):
        px[2 * k] = w; px[2 * k + 1] = WILD[(k + 2) % len(WILD)]
        state[4 * (k + 24):4 * (k + 24) + 4] = [w, 1.0, 10.0, 10.0]
        state[4 * (k + 32):4 * (k + 32) + 4] = [0.5, w, 10.0, 10.0]
    level = sd["level"].copy()
    fb, kk = fe.make_feature_batch(sd["ref_frame_idx"], px, sd["f"], sd["grad"], level, sd["type"])
    ns, st, succ, mr = gpu_ctx.update_seeds_batch(mopt, dopt, [rv], cv, fb, state)
    assert np.array_equal(st[4 * 48:], ref[1][4 * 48:]) and np.array_equal(succ[48:], ref[2][48:]) and np.array_equal(mr[48:], ref[3][48:])
    assert not succ[:8].any()


def test_new_entries_refuse_bad_arguments(gpu_ctx):
    """svoh_epipolar_match_batch, the deferred matcher section, svoh_sparse_align_fetch_all, svoh_context_stats and the
    packed pose call return error codes (never fault, never leave the context unusable) on misuse."""
    import ctypes as C
    lib, h = gpu_ctx.lib, gpu_ctx.h
    sc = synth.make_align_scene(307, n_features=8, rot_deg=(0.5, 1.5), trans_m=(0.05, 0.15))
    fr, fc = gpu_ctx.build_pyramid(sc.img_ref, 5), gpu_ctx.build_pyramid(sc.img_cur, 5)
    sd = synth.make_seed_set(sc, 64)
    rv = fe.make_frame_view(fr, sc.cam, sc.T_ref_f_w, 0.0, 1)
    cv = fe.make_frame_view(fc, sc.cam, sc.T_cur_f_w_gt, 0.0, 2)
    mopt = capi.default_matcher_options(max_epi_search_steps=500)
    ftype = np.where(sd["type"] == 0, capi.FT_EDGELET, capi.FT_CORNER).astype(np.uint8)
    fb, kk = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], ftype)
    dm = float(np.median(sd["true_depth"]))
    good = gpu_ctx.epipolar_match_batch(mopt, [rv], cv, fb, d_inv_common=[1 / dm, 3 / dm, 0.05 / dm])
    assert (good["result"] == 0).sum() > 20
    # no depth range at all / missing required outputs
    o = capi.svoh_epipolar_match_outputs()
    assert lib.svoh_epipolar_match_batch(h, C.byref(mopt), 1, C.byref(rv), C.byref(cv), None, C.byref(fb), None, None, C.byref(o)) != 0
    dc = (C.c_double * 3)(1 / dm, 3 / dm, 0.05 / dm)
    assert lib.svoh_epipolar_match_batch(h, C.byref(mopt), 1, C.byref(rv), C.byref(cv), None, C.byref(fb), dc, None, C.byref(o)) != 0
    # wild depth ranges per feature: the units fail or succeed, the call returns, the rest is unaffected
    d_inv = np.tile([1 / dm, 3 / dm, 0.05 / dm], 64)
    for k, w in enumerate(WILD):
        d_inv[3 * k:3 * k + 3] = [w, WILD[(k + 1) % len(WILD)], WILD[(k + 3) % len(WILD)]]
    wild = gpu_ctx.epipolar_match_batch(mopt, [rv], cv, fb, d_inv=d_inv)
    assert np.array_equal(wild["result"][8:], good["result"][8:]) and np.allclose(wild["depth"][8:], good["depth"][8:], rtol=1e-12, atol=0)
    # index out of range in a host batch is refused
    bad_idx = sd["ref_frame_idx"].copy(); bad_idx[5] = 3
    fbb, kb = fe.make_feature_batch(bad_idx, sd["px"], sd["f"], sd["grad"], sd["level"], ftype)
    with pytest.raises(fe.SvohError):
        gpu_ctx.epipolar_match_batch(mopt, [rv], cv, fbb, d_inv_common=[1 / dm, 3 / dm, 0.05 / dm])
    # deferred section misuse
    assert lib.svoh_matcher_collect(h) != 0                 # nothing open
    assert lib.svoh_matcher_begin_deferred(h) == 0
    assert lib.svoh_matcher_begin_deferred(h) != 0          # already open
    assert lib.svoh_matcher_collect(h) == 0                 # an empty section is fine
    # fetch_all with nothing queued, stats with a NULL output
    res = (capi.svoh_align_result * 1)()
    assert lib.svoh_sparse_align_fetch_all(h, 1, res) != 0
    assert lib.svoh_context_stats(h, None) != 0
    # knob / timing calls without a context
    assert lib.svoh_reload_knobs(None) != 0 and lib.svoh_set_kernel_timing(None, 1) != 0
    # a knob with a value the launch code does not know is ignored, not obeyed
    import os
    os.environ["SVOH_MATCHER_G8"] = "77"; os.environ["SVOH_KLT_BLOCK"] = "-5"
    try:
        gpu_ctx.reload_knobs()
        odd = gpu_ctx.epipolar_match_batch(mopt, [rv], cv, fb, d_inv_common=[1 / dm, 3 / dm, 0.05 / dm])
        assert np.array_equal(odd["result"], good["result"]) and np.array_equal(odd["depth"], good["depth"])
    finally:
        os.environ.pop("SVOH_MATCHER_G8"); os.environ.pop("SVOH_KLT_BLOCK")
        gpu_ctx.reload_knobs()
    # the context still works
    again = gpu_ctx.epipolar_match_batch(mopt, [rv], cv, fb, d_inv_common=[1 / dm, 3 / dm, 0.05 / dm])
    assert np.array_equal(again["result"], good["result"]) and np.array_equal(again["depth"], good["depth"])


def test_deferred_section_survives_other_calls(gpu_ctx):
    """Between svoh_matcher_begin_deferred and svoh_matcher_collect the two queued host batches stage through buffers
    of their own: an epipolar batch (larger than anything staged before, so the shared scratch is re-allocated), a
    detector call and a device-resident seed batch made inside the section leave them intact (round-2 advisor finding:
    they used to share the pinned / device scratch of the queued direct batch)."""
    import ctypes as C
    import torch
    lib, h = gpu_ctx.lib, gpu_ctx.h
    sc = synth.make_align_scene(311, n_features=8, rot_deg=(0.5, 1.5), trans_m=(0.05, 0.15))
    fr, fc = gpu_ctx.build_pyramid(sc.img_ref, 5), gpu_ctx.build_pyramid(sc.img_cur, 5)
    sd = synth.make_seed_set(sc, 300)
    big = synth.make_seed_set(sc, 6000, seed=5)
    rv = fe.make_frame_view(fr, sc.cam, sc.T_ref_f_w, float(sd["mu_range"]), 1)
    cv = fe.make_frame_view(fc, sc.cam, sc.T_cur_f_w_gt, 0.0, 2)
    mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(sc.cam)
    ftype = np.where(sd["type"] == 0, capi.FT_EDGELET, capi.FT_CORNER).astype(np.uint8)
    x = sd["f"].reshape(-1, 3).T * sd["true_depth"]
    px_true = sc.cam.project(sc.T_w_cur.inverse().transform(sc.T_w_ref.transform(x)))
    px_init = np.ascontiguousarray((px_true + 1.0).T).ravel()
    fb, kk = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], ftype)
    fbs, ks = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
    # the answers, call by call
    want_d = gpu_ctx.match_direct_batch(mopt, [rv], cv, fb, sd["true_depth"], px_init)
    want_s = gpu_ctx.update_seeds_batch(mopt, dopt, [rv], cv, fbs, sd["state"])
    bt = np.where(big["type"] == 0, capi.FT_EDGELET, capi.FT_CORNER).astype(np.uint8)
    fbb, kb = fe.make_feature_batch(big["ref_frame_idx"], big["px"], big["f"], big["grad"], big["level"], bt)
    dm = float(np.median(big["true_depth"]))
    mopt500 = capi.default_matcher_options(max_epi_search_steps=500)
    want_e = gpu_ctx.epipolar_match_batch(mopt500, [rv], cv, fbb, d_inv_common=[1 / dm, 3 / dm, 0.05 / dm])
    # the same inside ONE deferred section, with the other calls in between
    n = fb.n
    rva = (capi.svoh_frame_view * 1)(rv)
    got_d = dict(px_cur=px_init.copy(), result=np.zeros(n, np.int32), f_cur=np.zeros(3 * n), search_level=np.zeros(n, np.int32),
                 h_inv=np.zeros(n), A=np.zeros(4 * n))
    depth = np.ascontiguousarray(sd["true_depth"], np.float64)
    st = np.ascontiguousarray(sd["state"], np.float64).copy()
    succ, mr, ns = np.zeros(n, np.uint8), np.zeros(n, np.int32), C.c_int32()
    assert lib.svoh_matcher_begin_deferred(h) == 0
    try:
        assert lib.svoh_match_direct_batch(h, C.byref(mopt), 1, rva, C.byref(cv), C.byref(fb), depth.ctypes.data,
                                           got_d["px_cur"].ctypes.data, got_d["result"].ctypes.data, got_d["f_cur"].ctypes.data,
                                           got_d["search_level"].ctypes.data, got_d["h_inv"].ctypes.data, got_d["A"].ctypes.data) == 0
        assert lib.svoh_update_seeds_batch(h, C.byref(mopt), C.byref(dopt), 1, rva, C.byref(cv), C.byref(fbs), st.ctypes.data,
                                           succ.ctypes.data, mr.ctypes.data, C.byref(ns)) == 0
        # a second host batch of a kind that is queued is refused
        assert lib.svoh_match_direct_batch(h, C.byref(mopt), 1, rva, C.byref(cv), C.byref(fb), depth.ctypes.data,
                                           got_d["px_cur"].ctypes.data, got_d["result"].ctypes.data, None, None, None, None) != 0
        got_e = gpu_ctx.epipolar_match_batch(mopt500, [rv], cv, fbb, d_inv_common=[1 / dm, 3 / dm, 0.05 / dm])
        gpu_ctx.detect_features(capi.default_detector_options(), fr, sc.cam.width, sc.cam.height)
        # a device-resident seed batch runs at once (stream order) and stages its views through the shared scratch
        dev = torch.device("cuda", 0)
        t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in
             dict(idx=sd["ref_frame_idx"].astype(np.int32), px=sd["px"], f=sd["f"], grad=sd["grad"], level=sd["level"].astype(np.int32),
                  type=sd["type"].astype(np.uint8), state=sd["state"].astype(np.float64)).items()}
        d_succ = torch.zeros(n, dtype=torch.uint8, device=dev)
        fbd = fe.make_feature_batch_device(n, t["idx"].data_ptr(), t["px"].data_ptr(), t["f"].data_ptr(), t["grad"].data_ptr(),
                                           t["level"].data_ptr(), t["type"].data_ptr())
        gpu_ctx.update_seeds_device(mopt, dopt, [rv], [cv], fbd, t["state"].data_ptr(), d_succ.data_ptr())
    finally:
        assert lib.svoh_matcher_collect(h) == 0
    gpu_ctx.synchronize()
    for k in want_d:
        assert np.array_equal(want_d[k], got_d[k]), k
    assert ns.value == want_s[0] and np.array_equal(st, want_s[1]) and np.array_equal(succ, want_s[2]) and np.array_equal(mr, want_s[3])
    for k in want_e:
        assert np.array_equal(want_e[k], got_e[k]), k
    assert np.array_equal(t["state"].cpu().numpy(), want_s[1]) and np.array_equal(d_succ.cpu().numpy(), want_s[2])


def test_fetch_and_kernel_time_never_hand_out_an_older_launch(gpu_ctx):
    """Round-2 advisor findings: svoh_sparse_align_fetch after an evaluation (which delivers nothing) must fail instead
    of copying an older launch's results, and svoh_sparse_align_last_kernel_ms must fail for a launch made with kernel
    timing off instead of returning the time of the last timed one."""
    import ctypes as C
    lib, h = gpu_ctx.lib, gpu_ctx.h
    sc = synth.make_align_scene(313, n_features=120)
    fr, fc = gpu_ctx.build_pyramid(sc.img_ref, 5), gpu_ctx.build_pyramid(sc.img_cur, 5)
    pbs, keep = fe.make_align_problems([[(sc, fr, fc)]])
    opt = capi.default_align_options(min_level=2)
    first = gpu_ctx.sparse_align(opt, pbs)[0]
    ms = C.c_float()
    assert lib.svoh_sparse_align_last_kernel_ms(h, C.byref(ms)) == 0 and ms.value > 0
    gpu_ctx.sparse_align_evaluate(opt, pbs[0], 2)
    res = (capi.svoh_align_result * 1)()
    assert lib.svoh_sparse_align_fetch(h, 1, res) != 0            # the evaluation queued no result
    gpu_ctx.set_kernel_timing(False)
    try:
        again = gpu_ctx.sparse_align(opt, pbs)[0]
        assert lib.svoh_sparse_align_last_kernel_ms(h, C.byref(ms)) != 0   # that launch was not timed
    finally:
        gpu_ctx.set_kernel_timing(True)
    assert list(again.iters) == list(first.iters)
    gpu_ctx.sparse_align(opt, pbs)
    assert lib.svoh_sparse_align_last_kernel_ms(h, C.byref(ms)) == 0


def test_deferred_seed_batch_takes_its_frame_pose_late(gpu_ctx):
    """svoh_matcher_deferred_set_cur_frame (round 4): a seed batch queued in a deferred section at a WRONG pose of the
    current frame, the view replaced before the flush -- the results are those of the blocking call at the right pose, bit
    for bit.  Misuse is refused: no section, no queued seed batch, another frame, after the batch has been sent off."""
    import ctypes as C
    lib, h = gpu_ctx.lib, gpu_ctx.h
    lib.svoh_matcher_deferred_set_cur_frame.argtypes = [C.c_void_p, C.POINTER(capi.svoh_frame_view)]
    sc = synth.make_align_scene(411, n_features=8, rot_deg=(0.5, 1.5), trans_m=(0.05, 0.15))
    fr, fc = gpu_ctx.build_pyramid(sc.img_ref, 5), gpu_ctx.build_pyramid(sc.img_cur, 5)
    sd = synth.make_seed_set(sc, 700)
    rv = fe.make_frame_view(fr, sc.cam, sc.T_ref_f_w, float(sd["mu_range"]), 1)
    cv = fe.make_frame_view(fc, sc.cam, sc.T_cur_f_w_gt, 0.0, 2)
    off = synth.SE3(sc.T_cur_f_w_gt.q, np.array(sc.T_cur_f_w_gt.t) + [0.4, -0.1, 0.3])
    cv_wrong = fe.make_frame_view(fc, sc.cam, off, 0.0, 2)
    other = fe.make_frame_view(fr, sc.cam, sc.T_cur_f_w_gt, 0.0, 2)      # another frame handle
    mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(sc.cam)
    fbs, ks = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
    want = gpu_ctx.update_seeds_batch(mopt, dopt, [rv], cv, fbs, sd["state"])
    wrong = gpu_ctx.update_seeds_batch(mopt, dopt, [rv], cv_wrong, fbs, sd["state"])
    assert not np.array_equal(want[1], wrong[1])                          # the pose matters
    n = fbs.n
    rva = (capi.svoh_frame_view * 1)(rv)
    assert lib.svoh_matcher_deferred_set_cur_frame(h, C.byref(cv)) != 0   # no section
    assert lib.svoh_matcher_begin_deferred(h) == 0
    st = np.ascontiguousarray(sd["state"], np.float64).copy()
    succ, mr, ns = np.zeros(n, np.uint8), np.zeros(n, np.int32), C.c_int32()
    try:
        assert lib.svoh_matcher_deferred_set_cur_frame(h, C.byref(cv)) != 0   # no seed batch queued yet
        assert lib.svoh_update_seeds_batch(h, C.byref(mopt), C.byref(dopt), 1, rva, C.byref(cv_wrong), C.byref(fbs), st.ctypes.data,
                                           succ.ctypes.data, mr.ctypes.data, C.byref(ns)) == 0
        assert lib.svoh_matcher_deferred_set_cur_frame(h, None) != 0
        assert lib.svoh_matcher_deferred_set_cur_frame(h, C.byref(other)) != 0   # not the batch's current frame
        assert lib.svoh_matcher_deferred_set_cur_frame(h, C.byref(cv)) == 0
        assert lib.svoh_matcher_flush(h) == 0
        assert lib.svoh_matcher_deferred_set_cur_frame(h, C.byref(cv_wrong)) != 0   # sent off already
    finally:
        assert lib.svoh_matcher_collect(h) == 0
    assert ns.value == want[0] and np.array_equal(st, want[1]) and np.array_equal(succ, want[2]) and np.array_equal(mr, want[3])
