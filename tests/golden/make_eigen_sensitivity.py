#!/usr/bin/env python3
"""How much of the hot path's output depends on third-party arithmetic the reference links but does not contain
("bound the unpinned third-party arithmetic"):

  * Eigen's Matrix4f::inverse() in align2D (feature_alignment.cpp:275): generic cofactor path (what the oracle and the
    kernels restate) vs the vectorised block formula an x86 build of the reference executes (oracle mode 1);
  * the rounding inside Eigen's LDLT (mini_least_squares_solver.hpp:258): sequential inner sums vs two-lane packet
    partial sums (oracle mode 1).

Runs the oracle in both modes over the benchmark's synthetic workloads and writes the COUNT of outputs that change
to tests/golden/eigen_sensitivity.json.  `--subset` computes the small part the CPU test re-derives."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from svo_pro_universal_amd import _capi as capi, synth  # noqa: E402


def set_modes(inv4, ldlt):
    lib = orc.load()
    lib.orc_set_third_party_modes(int(inv4), int(ldlt))


def seeds_study(n_pairs, n_seeds, gain):
    """C4-synth (bench.py --workload seeds): updateSeed over n_pairs x n_seeds, inverse mode 0 vs 1."""
    tot = dict(seeds=0, result_code_changed=0, success_changed=0, type_changed=0, refined_2d=0, max_rel_state_diff_same_code=0.0)
    mopt = capi.default_matcher_options(affine_est_gain=gain)
    for b in range(n_pairs):
        sc = synth.make_align_scene(1000003 * 0 + b, n_features=8, rot_deg=(0.3, 1.0), trans_m=(0.05, 0.15))
        ref = orc.create_img_pyramid(sc.img_ref, 5)
        cur = orc.create_img_pyramid(sc.img_cur, 5)
        sd = synth.make_seed_set(sc, n_seeds, seed=b)
        dopt = capi.default_depth_filter_options(sc.cam)
        vr = orc.make_frame_view(ref, sc.cam, sc.T_ref_f_w, sd["mu_range"], 2 * b)
        vc = orc.make_frame_view(cur, sc.cam, sc.T_cur_f_w_gt, 0.0, 2 * b + 1)
        out = []
        for mode in (0, 1):
            set_modes(mode, 0)
            fb, keep = orc.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
            ns, st, succ, mr = orc.update_seeds_batch(mopt, dopt, [vr], vc, fb, sd["state"])
            out.append((st.reshape(-1, 4), succ, mr, keep["type"].copy()))
        set_modes(0, 0)
        (s0, u0, m0, t0), (s1, u1, m1, t1) = out
        tot["seeds"] += n_seeds
        tot["result_code_changed"] += int((m0 != m1).sum())
        tot["success_changed"] += int((u0 != u1).sum())
        tot["type_changed"] += int((t0 != t1).sum())
        tot["refined_2d"] += int(((sd["type"] != capi.FT_EDGELET_SEED) & (m0 != capi.MATCH_NOT_RUN)).sum())
        same = (m0 == m1) & (u0 == 1)
        if same.any():
            rel = np.abs(s0[same] - s1[same]) / np.maximum(np.abs(s0[same]), 1e-300)
            tot["max_rel_state_diff_same_code"] = max(tot["max_rel_state_diff_same_code"], float(rel.max()))
    return tot


def direct_study(n_scenes, n_feat):
    """findMatchDirect (reprojector path), inverse mode 0 vs 1."""
    tot = dict(features=0, result_code_changed=0, max_px_diff_same_code=0.0)
    mopt = capi.default_matcher_options()
    for b in range(n_scenes):
        sc = synth.make_align_scene(63 + b, n_features=8, rot_deg=(0.5, 1.5), trans_m=(0.05, 0.15))
        ref = orc.create_img_pyramid(sc.img_ref, 5)
        cur = orc.create_img_pyramid(sc.img_cur, 5)
        sd = synth.make_seed_set(sc, n_feat, margin=3, levels=(0, 1, 2, 3))
        x = sd["f"].reshape(-1, 3).T * sd["true_depth"]
        px_true = sc.cam.project(sc.T_w_cur.inverse().transform(sc.T_w_ref.transform(x)))
        px_init = np.ascontiguousarray((px_true + np.random.RandomState(1).uniform(-2.0, 2.0, px_true.shape)).T).ravel()
        ftype = np.where(sd["type"] == 0, capi.FT_EDGELET, capi.FT_CORNER)
        vr = orc.make_frame_view(ref, sc.cam, sc.T_ref_f_w, 0.0, 1)
        vc = orc.make_frame_view(cur, sc.cam, sc.T_cur_f_w_gt, 0.0, 2)
        out = []
        for mode in (0, 1):
            set_modes(mode, 0)
            fb, keep = orc.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], ftype)
            out.append(orc.match_direct_batch(mopt, [vr], vc, fb, sd["true_depth"], px_init))
        set_modes(0, 0)
        o0, o1 = out
        tot["features"] += n_feat
        tot["result_code_changed"] += int((o0["result"] != o1["result"]).sum())
        same = np.repeat((o0["result"] == o1["result"]) & (o0["result"] == 0), 2)
        if same.any():
            tot["max_px_diff_same_code"] = max(tot["max_px_diff_same_code"], float(np.abs(o0["px_cur"] - o1["px_cur"])[same].max()))
    return tot


def ldlt_study(n_scenes, n_feat):
    """SparseImgAlign::run (C2), LDLT inner-sum order 0 vs 1."""
    tot = dict(problems=0, iteration_counts_changed=0, status_changed=0, max_pose_diff=0.0)
    opt = capi.default_align_options()
    for b in range(n_scenes):
        sc = synth.make_align_scene(b, n_features=n_feat)
        ref = orc.create_img_pyramid(sc.img_ref, 5)
        cur = orc.create_img_pyramid(sc.img_cur, 5)
        pb = orc.problem_from_scenes([(sc, ref, cur)])
        res = []
        for mode in (0, 1):
            set_modes(0, mode)
            n, r, _ = orc.sparse_align_run(opt, pb)
            res.append((list(r.iters), r.status, np.array(list(r.T_icur_iref.q) + list(r.T_icur_iref.t))))
        set_modes(0, 0)
        tot["problems"] += 1
        tot["iteration_counts_changed"] += int(res[0][0] != res[1][0])
        tot["status_changed"] += int(res[0][1] != res[1][1])
        tot["max_pose_diff"] = max(tot["max_pose_diff"], float(np.abs(res[0][2] - res[1][2]).max()))
    return tot


def subset():
    return {"seeds_offset_only": seeds_study(2, 3000, 0), "match_direct": direct_study(1, 2000), "ldlt_align": ldlt_study(3, 600)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--subset", action="store_true")
    args = ap.parse_args()
    orc.build()
    out = {"what": __doc__.strip().split("\n\n")[0],
           "modes": "inverse4: 0 = Eigen generic cofactor path, 1 = Eigen 3.4 vectorised InverseSize4 block formula; "
                    "ldlt: 0 = sequential inner sums, 1 = two-lane packet partial sums",
           "subset": subset()}
    if not args.subset:
        out["seeds_offset_only_64x3000"] = seeds_study(64, 3000, 0)      # the benchmark's C4-synth size (192 000 seeds)
        out["seeds_offset_and_gain_16x3000"] = seeds_study(16, 3000, 1)  # 4x4 system fully populated
        out["match_direct_8x2000"] = direct_study(8, 2000)
        out["ldlt_align_32x2000"] = ldlt_study(32, 2000)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "eigen_sensitivity.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
