"""The oracle reproduces the committed golden fixtures (guards the checker
against drift; the GPU box re-checks the HIP path against the same file)."""
import numpy as np
import pytest

from svo_pro_universal_amd import _capi as capi

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
