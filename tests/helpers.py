"""Shared helpers for the parity tests: build the same problem for the oracle
(host pyramids) and for the product (device frame handles)."""
import numpy as np

from svo_pro_universal_amd import _capi as capi, synth


def scene_pyramids(orc, sc, n_levels=5, rounding=capi.SVOH_HALFSAMPLE_REFERENCE):
    return (orc.create_img_pyramid(sc.img_ref, n_levels, rounding),
            orc.create_img_pyramid(sc.img_cur, n_levels, rounding))


def se3_max_abs_diff(a, b):
    """a, b: svoh_se3; quaternion sign-insensitive max abs difference"""
    qa = np.array([a.q[i] for i in range(4)]); qb = np.array([b.q[i] for i in range(4)])
    ta = np.array([a.t[i] for i in range(3)]); tb = np.array([b.t[i] for i in range(3)])
    dq = min(np.abs(qa - qb).max(), np.abs(qa + qb).max())
    return max(dq, np.abs(ta - tb).max())


def make_prior(T_prior, lambda_rot, lambda_trans, alpha=0.0, beta=0.0, lambda_alpha=0.0, lambda_beta=0.0):
    from oracle import oracle as orc
    p = capi.svoh_align_prior()
    p.have_prior = 1
    p.T_prior = orc.to_se3(T_prior)
    p.alpha_prior, p.beta_prior = alpha, beta
    p.lambda_rot, p.lambda_trans = lambda_rot, lambda_trans
    p.lambda_alpha, p.lambda_beta = lambda_alpha, lambda_beta
    return p


def small_scene(seed, n=300, P=4, cam=None, **kw):
    return synth.make_align_scene(seed, n_features=n, patch_size=P, cam=cam, **kw)


# ---------------------------------------------------------------------------
# golden fixtures (tests/golden/align_small.npz, made by tests/golden/make_golden.py)
# ---------------------------------------------------------------------------
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "align_small.npz")

GOLDEN_OPTION_SETS = {
    "plain": dict(max_level=3, min_level=0),
    "handler_levels": dict(max_level=3, min_level=2),
    "illum_robust": dict(max_level=3, min_level=0, estimate_illumination_gain=1, estimate_illumination_offset=1,
                         robustification=1),
    "distjac": dict(max_level=3, min_level=1, use_distortion_jacobian=1),
}


class GoldenScene(object):
    pass


def scene_from_golden(z, tag):
    p = tag + "/"
    sc = GoldenScene()
    c = z[p + "cam"]
    sc.cam = synth.Camera(int(c[0]), int(c[1]), c[2], c[3], c[4], c[5], dist=list(c[6:10]) if c[10] else None)
    sc.img_ref, sc.img_cur = z[p + "img_ref"], z[p + "img_cur"]
    sc.px, sc.f, sc.pos_world, sc.flags = z[p + "px"], z[p + "f"], z[p + "pos_world"], z[p + "flags"]
    sc.n_features = int(sc.flags.size)
    sc.T_cam_imu, sc.T_imu_cam = synth.SE3.from7(z[p + "T_cam_imu"]), synth.SE3.from7(z[p + "T_imu_cam"])
    sc.ref_pos = z[p + "ref_pos"]
    sc.T_icur_iref_init = synth.SE3()
    sc.T_icur_iref_gt = synth.SE3.from7(z[p + "T_gt"])
    return sc


def se3_vec_diff(v, s):
    """golden 7-vector vs svoh_se3, quaternion-sign insensitive"""
    q = np.array([s.q[i] for i in range(4)]); t = np.array([s.t[i] for i in range(3)])
    return max(min(np.abs(v[:4] - q).max(), np.abs(v[:4] + q).max()), np.abs(v[4:] - t).max())
