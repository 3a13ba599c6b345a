"""First GPU bring-up: pyramid, evaluate and full-run parity vs the oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from svo_pro_universal_amd import synth, _capi as capi, frontend as fe
from oracle import oracle as orc

ctx = fe.Context(0)
for P in (4, 8):
    for seed in (0, 1):
        cam = synth.Camera.test_camera() if seed == 0 else synth.Camera.euroc_like()
        sc = synth.make_align_scene(seed, n_features=2000, patch_size=P, cam=cam, border_features=100,
                                    invalid_fraction=0.05)
        ref = orc.create_img_pyramid(sc.img_ref, 5)
        cur = orc.create_img_pyramid(sc.img_cur, 5)
        fr, lv = ctx.build_pyramid(sc.img_ref, 5, return_levels=True)
        fc = ctx.build_pyramid(sc.img_cur, 5)
        for l in range(5):
            assert np.array_equal(lv[l], ref[l]), ("pyramid mismatch", l)
            assert np.array_equal(ctx.download_level(fc, l), cur[l])
        for illum in (0, 1):
            for robust in (0, 1):
                opt = capi.default_align_options(min_level=0, patch_size=P, estimate_illumination_gain=illum,
                                                 estimate_illumination_offset=illum, robustification=robust)
                opb = orc.problem_from_scenes([(sc, ref, cur)])
                pbs, keep = fe.make_align_problems([[(sc, fr, fc)]])
                for level in (4, 2, 0):
                    Ho, go, c2o, nmo, viso = orc.sparse_align_evaluate(opt, opb, level)
                    Hg, gg, c2g, nmg, visg = ctx.sparse_align_evaluate(opt, pbs[0], level)
                    relH = np.abs(Hg - Ho).max() / np.abs(Ho).max()
                    relg = np.abs(gg - go).max() / np.abs(go).max()
                    ok = np.array_equal(viso, visg) and nmo == nmg
                    print("P%d seed%d illum%d rob%d L%d: relH %.2e relg %.2e chi2 %.6f/%.6f n_meas %d/%d vis_equal %s nsel %d"
                          % (P, seed, illum, robust, level, relH, relg, c2g, c2o, nmg, nmo, ok, len(visg)))
                n, ro, tr = orc.sparse_align_run(opt, opb)
                t = time.time()
                rg = ctx.sparse_align(opt, pbs)[0]
                dt = time.time() - t
                To, Tg = orc.from_se3(ro.T_icur_iref), orc.from_se3(rg.T_icur_iref)
                print("  run: n %d/%d status %d/%d iters %s/%s dT(gpu,orc)=%s err_gt=%s ab=(%g,%g)/(%g,%g) %.1f ms"
                      % (rg.n_fts_to_track, n, rg.status, ro.status, list(rg.iters)[:5], list(ro.iters)[:5],
                         synth.se3_error(Tg, To), synth.se3_error(Tg, sc.T_icur_iref_gt), rg.alpha, rg.beta, ro.alpha, ro.beta, dt * 1e3))
print("DONE")
