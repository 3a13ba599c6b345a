#!/usr/bin/env python3
"""Generates tests/golden/detect_pose_small.npz from the CPU oracle: a 320x240 image with the detector's
features (FAST + edgelets, a partly occupied grid) and one 2-camera pose-optimisation problem with its result.
(The reference holds no vectors for these functions.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from svo_pro_universal_amd import _capi as capi, frontend as fe, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402
import pose_helpers as ph  # noqa: E402


def main():
    orc.build()
    cam = synth.Camera(320, 240, 195.2, 228.6, 156.3, 124.2, dist=[-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05])
    sc = synth.make_align_scene(34, n_features=10, cam=cam, max_level=3)
    levels = orc.create_img_pyramid(sc.img_ref, 4)
    opt = capi.default_detector_options(cell_size=20)
    n_cells = 16 * 12
    occ = (np.arange(n_cells) % 7 == 0).astype(np.uint8)
    d = orc.detect_features(opt, levels, occ)
    out = dict(img=sc.img_ref, det_occupancy=occ, det_px=d["px"], det_score=d["score"], det_level=d["level"], det_grad=d["grad"],
               det_type=d["type"])
    ps = ph.make_pose_scene(35, n=90, cam=cam, n_cams=2)
    popt = capi.default_pose_options(cam)
    pb, keep = fe.make_pose_problem(ps["cams"], ps["T_imu_world_init"])
    r = orc.optimize_pose(popt, pb)
    out.update(cam=np.array([cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy] + cam.dist),
               pose_T_init=ps["T_imu_world_init"].as7(), pose_outlier_threshold=np.array([popt.outlier_threshold]),
               pose_T_out=fe.se3_to_numpy(r.T_imu_world), pose_sigma=np.array([r.measurement_sigma]),
               pose_counts=np.array([r.n_meas, r.n_deleted_edges, r.n_deleted_corners, r.iters, r.status], np.int32))
    for c, (cd, k) in enumerate(zip(ps["cams"], keep)):
        out["pose%d_T_cam_imu" % c] = cd["T_cam_imu"].as7()
        for key in ("px", "f", "grad", "level", "type", "xyz_world", "usable"):
            out["pose%d_%s" % (c, key)] = np.asarray(cd[key])
        out["pose%d_outlier" % c] = k["outlier"][:len(cd["level"])].copy()
        out["pose%d_final_error" % c] = k["final_error"][:len(cd["level"])].copy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "detect_pose_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(d["score"]), "features,", int((d["type"] == capi.FT_EDGELET).sum()),
          "edgelets; pose iters", r.iters, "outliers", r.n_deleted_edges + r.n_deleted_corners)


if __name__ == "__main__":
    main()
