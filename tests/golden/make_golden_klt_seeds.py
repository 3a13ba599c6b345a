#!/usr/bin/env python3
"""Generates tests/golden/klt_seeds_small.npz from the CPU oracle: a 320x240 frame
pair with 60 KLT tracks, 120 depth-filter seeds and 80 direct-match candidates and
the oracle's outputs.  (The reference holds no vectors for these functions.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from svo_pro_universal_amd import _capi as capi, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    orc.build()
    cam = synth.Camera(320, 240, 195.2, 228.6, 156.3, 124.2, dist=[-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05])
    sc = synth.make_align_scene(33, n_features=10, cam=cam, max_level=3, rot_deg=(0.4, 0.9), trans_m=(0.06, 0.12))
    ref = orc.create_img_pyramid(sc.img_ref, 4)
    cur = orc.create_img_pyramid(sc.img_cur, 4)
    out = dict(img_ref=sc.img_ref, img_cur=sc.img_cur,
               cam=np.array([cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy] + cam.dist),
               T_ref_f_w=sc.T_ref_f_w.as7(), T_cur_f_w=sc.T_cur_f_w_gt.as7())
    tr = synth.make_track_set(sc, 60, margin=8)
    kopt = capi.default_klt_options(max_level=3, patch_sizes=[16, 16, 8, 8])
    p, s = orc.klt_track_batch(kopt, ref, cur, tr["px_ref"], tr["px_cur_init"])
    out.update(klt_px_ref=tr["px_ref"], klt_px_init=tr["px_cur_init"], klt_px_out=p, klt_status=s)
    sd = synth.make_seed_set(sc, 120, margin=10, levels=(0, 1))
    sd["type"][::11] = capi.FT_MAPPOINT_SEED
    mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(cam)
    rv = orc.make_frame_view(ref, cam, sc.T_ref_f_w, sd["mu_range"], 1)
    cv = orc.make_frame_view(cur, cam, sc.T_cur_f_w_gt, 0.0, 2)
    fb, keep = orc.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
    ns, st, succ, mr = orc.update_seeds_batch(mopt, dopt, [rv], cv, fb, sd["state"])
    out.update(seed_px=sd["px"], seed_f=sd["f"], seed_grad=sd["grad"], seed_level=sd["level"], seed_type_in=sd["type"],
               seed_state_in=sd["state"], seed_mu_range=np.array([sd["mu_range"]]), seed_px_error_angle=np.array([dopt.px_error_angle]),
               seed_state_out=st, seed_success=succ, seed_match_result=mr, seed_type_out=keep["type"])
    x = sd["f"].reshape(-1, 3).T * sd["true_depth"]
    px_true = cam.project(sc.T_w_cur.inverse().transform(sc.T_w_ref.transform(x)))
    px_init = np.ascontiguousarray((px_true + np.random.RandomState(2).uniform(-1.5, 1.5, px_true.shape)).T).ravel()[:160]
    ftype = np.where(sd["type"][:80] == 0, capi.FT_EDGELET, capi.FT_CORNER).astype(np.uint8)
    fb2, keep2 = orc.make_feature_batch(sd["ref_frame_idx"][:80], sd["px"][:160], sd["f"][:240], sd["grad"][:160], sd["level"][:80], ftype)
    o = orc.match_direct_batch(mopt, [rv], cv, fb2, sd["true_depth"][:80], px_init)
    out.update(direct_type=ftype, direct_depth=sd["true_depth"][:80], direct_px_init=px_init, direct_px_out=o["px_cur"],
               direct_result=o["result"], direct_search_level=o["search_level"], direct_A=o["A"], direct_f_cur=o["f_cur"])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "klt_seeds_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; klt ok", s.mean(), "seeds ok", ns, "direct ok", (o["result"] == 0).sum())


if __name__ == "__main__":
    main()
