#!/usr/bin/env python3
"""Generates tests/golden/align_small.npz from the CPU oracle (strict IEEE build).

The reference ships no golden vectors for this path and cannot be built here
(SURVEY.md 8c), so these fixtures pin the ORACLE's behaviour over time (and the
GPU path against it on the GPU box, where /root/reference does not exist).
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from svo_pro_universal_amd import _capi as capi, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

OPTION_SETS = {
    "plain": dict(max_level=3, min_level=0),
    "handler_levels": dict(max_level=3, min_level=2),
    "illum_robust": dict(max_level=3, min_level=0, estimate_illumination_gain=1, estimate_illumination_offset=1,
                         robustification=1),
    "distjac": dict(max_level=3, min_level=1, use_distortion_jacobian=1),
}


def se3v(s):
    return np.array([s.q[0], s.q[1], s.q[2], s.q[3], s.t[0], s.t[1], s.t[2]])


def main():
    orc.build()
    out = {}
    for tag, cam in (("pinhole", synth.Camera(320, 240, 160.0, 160.0, 160.0, 120.0)),
                     ("radtan", synth.Camera(320, 240, 195.2, 228.6, 156.3, 124.2,
                                             dist=[-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05]))):
        sc = synth.make_align_scene(21, n_features=150, patch_size=4, cam=cam, max_level=3, border_features=30,
                                    invalid_fraction=0.08, gain=1.04, offset=2.0)
        ref = orc.create_img_pyramid(sc.img_ref, 4)
        cur = orc.create_img_pyramid(sc.img_cur, 4)
        p = tag + "/"
        out[p + "img_ref"], out[p + "img_cur"] = sc.img_ref, sc.img_cur
        out[p + "ref_level3"], out[p + "cur_level3"] = ref[3], cur[3]
        out[p + "cam"] = np.array([cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy] + (cam.dist or [0, 0, 0, 0])
                                  + [0 if cam.dist is None else 1], dtype=np.float64)
        out[p + "px"], out[p + "f"], out[p + "pos_world"], out[p + "flags"] = sc.px, sc.f, sc.pos_world, sc.flags
        out[p + "T_cam_imu"], out[p + "T_imu_cam"] = sc.T_cam_imu.as7(), sc.T_imu_cam.as7()
        out[p + "ref_pos"] = sc.ref_pos
        out[p + "T_gt"] = sc.T_icur_iref_gt.as7()
        for name, kw in OPTION_SETS.items():
            opt = capi.default_align_options(**kw)
            pb = orc.problem_from_scenes([(sc, ref, cur)])
            q = p + name + "/"
            for level in range(opt.min_level, opt.max_level + 1):
                H, g, chi2, nm, vis = orc.sparse_align_evaluate(opt, pb, level)
                out[q + "H%d" % level], out[q + "g%d" % level] = H, g
                out[q + "chi2_nmeas%d" % level] = np.array([chi2, nm])
                out[q + "vis%d" % level] = vis
            n, res, tr = orc.sparse_align_run(opt, pb, trace_capacity=64)
            out[q + "run_T"] = se3v(res.T_icur_iref)
            out[q + "run_ab"] = np.array([res.alpha, res.beta])
            out[q + "run_iters"] = np.array(list(res.iters))
            out[q + "run_nmeas"] = np.array(list(res.n_meas))
            out[q + "run_misc"] = np.array([n, res.status, res.n_patch_iters])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "align_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
