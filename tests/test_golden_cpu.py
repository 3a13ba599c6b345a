"""The oracle reproduces the committed golden fixtures (guards the checker
against drift; the GPU box re-checks the HIP path against the same file)."""
import numpy as np
import pytest

from svo_pro_universal_amd import _capi as capi, frontend as fe, synth

import helpers


@pytest.mark.parametrize("tag", ["pinhole", "radtan"])
def test_oracle_reproduces_golden(oracle_lib, tag):
    orc = oracle_lib
    z = np.load(helpers.GOLDEN)
    sc = helpers.scene_from_golden(z, tag)
    ref = orc.create_img_pyramid(sc.img_ref, 4)
    cur = orc.create_img_pyramid(sc.img_cur, 4)
    assert np.array_equal(ref[3], z[tag + "/ref_level3"]) and np.array_equal(cur[3], z[tag + "/cur_level3"])
    for name, kw in helpers.GOLDEN_OPTION_SETS.items():
        opt = capi.default_align_options(**kw)
        pb = orc.problem_from_scenes([(sc, ref, cur)])
        q = "%s/%s/" % (tag, name)
        for level in range(opt.min_level, opt.max_level + 1):
            H, g, chi2, nm, vis = orc.sparse_align_evaluate(opt, pb, level)
            assert np.array_equal(vis, z[q + "vis%d" % level]) and nm == int(z[q + "chi2_nmeas%d" % level][1])
            assert np.abs(H - z[q + "H%d" % level]).max() <= 1e-12 * np.abs(H).max()
            assert np.abs(g - z[q + "g%d" % level]).max() <= 1e-12 * np.abs(g).max()
        n, res, _ = orc.sparse_align_run(opt, pb)
        assert [n, res.status, res.n_patch_iters] == list(z[q + "run_misc"])
        assert list(res.iters) == list(z[q + "run_iters"]) and list(res.n_meas) == list(z[q + "run_nmeas"])
        assert helpers.se3_vec_diff(z[q + "run_T"], res.T_icur_iref) < 1e-12
        assert np.abs(np.array([res.alpha, res.beta]) - z[q + "run_ab"]).max() < 1e-10


def _golden2_inputs(z):
    from svo_pro_universal_amd import synth
    c = z["cam"]
    cam = synth.Camera(int(c[0]), int(c[1]), c[2], c[3], c[4], c[5], dist=list(c[6:10]))
    return cam, synth.SE3.from7(z["T_ref_f_w"]), synth.SE3.from7(z["T_cur_f_w"])


def test_oracle_reproduces_golden_klt_seeds(oracle_lib):
    import os
    orc = oracle_lib
    z = np.load(os.path.join(os.path.dirname(helpers.GOLDEN), "klt_seeds_small.npz"))
    cam, T_ref, T_cur = _golden2_inputs(z)
    ref = orc.create_img_pyramid(z["img_ref"], 4); cur = orc.create_img_pyramid(z["img_cur"], 4)
    kopt = capi.default_klt_options(max_level=3, patch_sizes=[16, 16, 8, 8])
    p, s = orc.klt_track_batch(kopt, ref, cur, z["klt_px_ref"], z["klt_px_init"])
    assert np.array_equal(s, z["klt_status"]) and np.array_equal(p, z["klt_px_out"])
    mopt = capi.default_matcher_options()
    dopt = capi.default_depth_filter_options(px_error_angle=float(z["seed_px_error_angle"][0]))
    rv = orc.make_frame_view(ref, cam, T_ref, float(z["seed_mu_range"][0]), 1)
    cv = orc.make_frame_view(cur, cam, T_cur, 0.0, 2)
    n = z["seed_level"].size
    fb, keep = orc.make_feature_batch(np.zeros(n, np.int32), z["seed_px"], z["seed_f"], z["seed_grad"], z["seed_level"], z["seed_type_in"])
    ns, st, succ, mr = orc.update_seeds_batch(mopt, dopt, [rv], cv, fb, z["seed_state_in"])
    assert np.array_equal(succ, z["seed_success"]) and np.array_equal(mr, z["seed_match_result"])
    assert np.array_equal(keep["type"], z["seed_type_out"]) and np.allclose(st, z["seed_state_out"], rtol=1e-12, atol=0)
    fb2, keep2 = orc.make_feature_batch(np.zeros(80, np.int32), z["seed_px"][:160], z["seed_f"][:240], z["seed_grad"][:160],
                                        z["seed_level"][:80], z["direct_type"])
    o = orc.match_direct_batch(mopt, [rv], cv, fb2, z["direct_depth"], z["direct_px_init"])
    assert np.array_equal(o["result"], z["direct_result"]) and np.array_equal(o["search_level"], z["direct_search_level"])
    assert np.abs(o["px_cur"] - z["direct_px_out"]).max() < 1e-9


def _golden3(z):
    c = z["cam"]
    cam = synth.Camera(int(c[0]), int(c[1]), c[2], c[3], c[4], c[5], dist=list(c[6:10]))
    cams = []
    for k in range(2):
        cams.append(dict(cam=cam, T_cam_imu=synth.SE3.from7(z["pose%d_T_cam_imu" % k]), px=z["pose%d_px" % k], f=z["pose%d_f" % k],
                         grad=z["pose%d_grad" % k], level=z["pose%d_level" % k], type=z["pose%d_type" % k],
                         xyz_world=z["pose%d_xyz_world" % k], usable=z["pose%d_usable" % k]))
    popt = capi.default_pose_options(outlier_threshold=float(z["pose_outlier_threshold"][0]))
    return cam, cams, popt


def check_golden3(z, d, r, keep):
    """shared by the CPU (oracle) and GPU (HIP) golden tests"""
    assert np.array_equal(d["px"], z["det_px"]) and np.array_equal(d["score"], z["det_score"])
    assert np.array_equal(d["level"], z["det_level"]) and np.array_equal(d["type"], z["det_type"])
    assert np.abs(d["grad"] - z["det_grad"]).max() < 1e-12
    assert list(z["pose_counts"]) == [r.n_meas, r.n_deleted_edges, r.n_deleted_corners, r.iters, r.status]
    assert r.measurement_sigma == float(z["pose_sigma"][0])
    assert np.abs(fe.se3_to_numpy(r.T_imu_world) - z["pose_T_out"]).max() < 1e-9
    for k in range(2):
        n = len(z["pose%d_level" % k])
        assert np.array_equal(keep[k]["outlier"][:n], z["pose%d_outlier" % k])
        assert np.allclose(keep[k]["final_error"][:n], z["pose%d_final_error" % k], rtol=1e-9, atol=1e-15)


def test_oracle_reproduces_golden_detect_pose(oracle_lib):
    import os
    z = np.load(os.path.join(os.path.dirname(helpers.GOLDEN), "detect_pose_small.npz"))
    cam, cams, popt = _golden3(z)
    levels = oracle_lib.create_img_pyramid(z["img"], 4)
    d = oracle_lib.detect_features(capi.default_detector_options(cell_size=20), levels, z["det_occupancy"])
    pb, keep = fe.make_pose_problem(cams, synth.SE3.from7(z["pose_T_init"]))
    check_golden3(z, d, oracle_lib.optimize_pose(popt, pb), keep)


def test_oracle_reproduces_golden_structure(oracle_lib):
    """Point::optimize fixture (tests/golden/structure_small.npz, made by make_golden_structure.py)."""
    import os
    z = np.load(os.path.join(os.path.dirname(helpers.GOLDEN), "structure_small.npz"))
    for sphere in (0, 1):
        p, it = oracle_lib.optimize_points(list(z["views"]), z["obs_begin"], z["obs_view"], z["obs_f"], z["pos0"], n_iter=5,
                                           using_bearing_vector=bool(sphere))
        assert np.array_equal(it, z["iters_%d" % sphere])
        assert np.array_equal(p, z["pos_out_%d" % sphere])     # same code, same machine arithmetic: bit for bit


def test_oracle_reproduces_the_stereo_seam_fixture(oracle_lib):
    """tests/golden/stereo_small.npz (make_golden_stereo.py): 100 x Matcher::findEpipolarMatchDirect as
    StereoTriangulation::compute calls it, on the frame pair of klt_seeds_small.npz."""
    import os
    orc = oracle_lib
    z = np.load(os.path.join(os.path.dirname(helpers.GOLDEN), "klt_seeds_small.npz"))
    s = np.load(os.path.join(os.path.dirname(helpers.GOLDEN), "stereo_small.npz"))
    cam, T_ref, T_cur = _golden2_inputs(z)
    ref = orc.create_img_pyramid(z["img_ref"], 4); cur = orc.create_img_pyramid(z["img_cur"], 4)
    n = int(s["n"][0])
    mopt = capi.default_matcher_options(max_epi_search_steps=500, subpix_refinement=1, scan_on_unit_sphere=1)
    fb, keep = orc.make_feature_batch(np.zeros(n, np.int32), z["seed_px"][:2 * n], z["seed_f"][:3 * n], z["seed_grad"][:2 * n],
                                      z["seed_level"][:n], s["type"])
    o = orc.epipolar_match_batch(mopt, [orc.make_frame_view(ref, cam, T_ref, 0.0, 1)], orc.make_frame_view(cur, cam, T_cur, 0.0, 2), fb,
                                 d_inv_common=list(s["d_inv"]), T_cur_ref=[s["T_f1f0"]])
    assert np.array_equal(o["result"], s["result"]) and np.array_equal(o["search_level"], s["search_level"])
    assert np.array_equal(o["depth"], s["depth"]) and np.array_equal(o["px_cur"], s["px_cur"])
    assert (s["result"] == 0).sum() > 60 and len(set(s["result"])) >= 3
