"""Exact boundary values of tests/test_boundaries_cpu.py through the C ABI (gotcha 1)."""
import numpy as np
import pytest

from svo_pro_universal_amd import _capi as capi, frontend as fe

import helpers
import test_boundaries_cpu as tb

pytestmark = pytest.mark.gpu


def test_a3_selection_boundaries(gpu_ctx, oracle_lib):
    sc = tb.boundary_scene()
    fr, fc = gpu_ctx.build_pyramid(sc.img_ref, 5), gpu_ctx.build_pyramid(sc.img_cur, 5)
    gpb, keep = fe.make_align_problems([[(sc, fr, fc)]])
    opt = capi.default_align_options()
    H, g, chi2, nm, vis = gpu_ctx.sparse_align_evaluate(opt, gpb[0], 4)
    assert vis.size == sum(tb.A3_KEPT)                       # one visibility byte per selected feature
    res = gpu_ctx.sparse_align(opt, gpb)[0]
    assert res.n_fts_to_track == sum(tb.A3_KEPT)
    # the crafted pixels do not belong to the landmarks behind them, so the optimisation itself is meaningless here;
    # the first evaluation is still the same on both sides
    ref, cur = helpers.scene_pyramids(oracle_lib, sc)
    Ho, go, c2o, nmo, viso = oracle_lib.sparse_align_evaluate(opt, oracle_lib.problem_from_scenes([(sc, ref, cur)]), 4)
    assert nmo == nm and np.array_equal(viso, vis)
    assert np.abs(H - Ho).max() <= 1e-10 * np.abs(Ho).max() and np.abs(g - go).max() <= 1e-10 * np.abs(go).max()


def test_klt_template_and_current_boundaries(gpu_ctx, oracle_lib):
    sc = helpers.small_scene(62, n=10)
    ref = oracle_lib.create_img_pyramid(sc.img_ref, 5)
    fr = gpu_ctx.build_pyramid(sc.img_ref, 5)
    opt = capi.default_klt_options(max_level=0, min_level=0)
    px_ref, px_cur = tb.klt_boundary_tracks()
    n = px_ref.size // 2
    po, so = oracle_lib.klt_track_batch(opt, [ref] * n, ref, px_ref, px_cur)
    pg, sg = gpu_ctx.klt_track_batch(opt, [fr] * n, fr, px_ref, px_cur)
    assert np.array_equal(sg, so) and np.array_equal(pg, po)
    assert list(sg[:4]) == [int(k) for k in tb.KLT_REF_OK] and sg[8] == 0
