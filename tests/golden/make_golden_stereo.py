#!/usr/bin/env python3
"""Generates tests/golden/stereo_small.npz from the CPU oracle: 100 detector-type features of the klt_seeds_small
frame pair (its images, camera and poses are reused, not stored again) matched along their epipolar lines the way
StereoTriangulation::compute does (Matcher::findEpipolarMatchDirect with an explicit T_f1f0, 500 steps, align_1d =
isEdgelet) and the oracle's outputs.  (The reference holds no vectors for this function.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from svo_pro_universal_amd import _capi as capi, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def inputs():
    z = np.load(os.path.join(HERE, "klt_seeds_small.npz"))
    c = z["cam"]
    cam = synth.Camera(int(c[0]), int(c[1]), c[2], c[3], c[4], c[5], dist=list(c[6:10]))
    return z, cam, synth.SE3.from7(z["T_ref_f_w"]), synth.SE3.from7(z["T_cur_f_w"])


def main():
    orc.build()
    z, cam, T_ref, T_cur = inputs()
    ref = orc.create_img_pyramid(z["img_ref"], 4)
    cur = orc.create_img_pyramid(z["img_cur"], 4)
    n = 100
    ftype = np.where(z["seed_type_in"][:n] == capi.FT_EDGELET_SEED, capi.FT_EDGELET, capi.FT_CORNER).astype(np.uint8)
    T_f1f0 = (T_cur * T_ref.inverse()).as7()
    d_inv = np.array([1.0 / 3.0, 1.0 / 0.8, 1.0 / 40.0])
    mopt = capi.default_matcher_options(max_epi_search_steps=500, subpix_refinement=1, scan_on_unit_sphere=1)
    rv = orc.make_frame_view(ref, cam, T_ref, 0.0, 1)
    cv = orc.make_frame_view(cur, cam, T_cur, 0.0, 2)
    fb, keep = orc.make_feature_batch(np.zeros(n, np.int32), z["seed_px"][:2 * n], z["seed_f"][:3 * n], z["seed_grad"][:2 * n],
                                      z["seed_level"][:n], ftype)
    o = orc.epipolar_match_batch(mopt, [rv], cv, fb, d_inv_common=list(d_inv), T_cur_ref=[T_f1f0])
    path = os.path.join(HERE, "stereo_small.npz")
    np.savez_compressed(path, n=np.array([n]), type=ftype, T_f1f0=T_f1f0, d_inv=d_inv, result=o["result"], depth=o["depth"],
                        px_cur=o["px_cur"], f_cur=o["f_cur"], search_level=o["search_level"], A=o["A"])
    print("wrote", path, os.path.getsize(path), "bytes; results", np.bincount(o["result"]))


if __name__ == "__main__":
    main()
