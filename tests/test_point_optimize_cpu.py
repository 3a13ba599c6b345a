"""Oracle restatement of Point::optimize (src/svo_common/src/point.cpp:216-325) against properties that need no
reference run: an independent numpy Gauss-Newton, convergence to the ground truth on noise-free data, the
"fewer than two observations" and roll-back rules.  PARITY UNPINNED (the reference has no vectors for it)."""
import numpy as np

from oracle import oracle as orc
from svo_pro_universal_amd import synth

import pose_helpers as ph


def numpy_point_gn(views, fs, pos, n_iter, sphere):
    """Plain numpy version of the same iteration (np.linalg.solve instead of the pivoted LDLT)."""
    pos = pos.copy()
    old, chi2 = pos.copy(), 0.0
    for it in range(n_iter):
        A, b, new_chi2 = np.zeros((3, 3)), np.zeros(3), 0.0
        for T, f in zip(views, fs):
            R = T.R()
            p = T.transform(pos)
            if sphere:
                n = np.linalg.norm(p)
                J = -(np.eye(3) * (p @ p) - np.outer(p, p)) / n ** 3 @ R
                e = f - p / n
            else:
                J = -np.array([[1 / p[2], 0, -p[0] / p[2] ** 2], [0, 1 / p[2], -p[1] / p[2] ** 2]]) @ R
                e = f[:2] / f[2] - p[:2] / p[2]
            A += J.T @ J
            b -= J.T @ e
            new_chi2 += e @ e
        dp = np.linalg.solve(A, b)
        if it > 0 and new_chi2 > chi2:
            return old
        old, pos, chi2 = pos.copy(), pos + dp, new_chi2
        if np.abs(dp).max() <= 1e-10:
            break
    return pos


def test_against_numpy_gauss_newton_both_error_models():
    sc = ph.make_structure_scene(3, n_points=60, degenerate=False)
    views = [synth.SE3.from7(v) for v in sc["views"]]
    for sphere in (False, True):
        out, iters = orc.optimize_points(sc["views"], sc["obs_begin"], sc["obs_view"], sc["obs_f"], sc["pos0"], n_iter=8,
                                         using_bearing_vector=sphere)
        for i in range(60):
            o0, o1 = sc["obs_begin"][i], sc["obs_begin"][i + 1]
            ref = numpy_point_gn([views[v] for v in sc["obs_view"][o0:o1]], sc["obs_f"][o0:o1], sc["pos0"][i], 8, sphere)
            # near convergence chi2 moves in its last bits, so the "error grew -> roll back" rule may fire one
            # iteration apart in the two implementations: they agree to the size of that last step
            assert np.abs(out[i] - ref).max() < 1e-6, (sphere, i)
        assert iters.min() >= 2 and iters.max() <= 8


def test_noise_free_observations_recover_the_landmark():
    sc = ph.make_structure_scene(4, n_points=80, noise=0.0, degenerate=False)
    out, iters = orc.optimize_points(sc["views"], sc["obs_begin"], sc["obs_view"], sc["obs_f"], sc["pos0"], n_iter=15)
    assert np.abs(out - sc["pos_gt"]).max() < 1e-6
    # it cannot get better than the truth: started there, the first step is ~0 and the loop leaves at once
    out2, it2 = orc.optimize_points(sc["views"], sc["obs_begin"], sc["obs_view"], sc["obs_f"], sc["pos_gt"], n_iter=15)
    assert np.abs(out2 - sc["pos_gt"]).max() < 1e-9 and it2.max() <= 2


def test_degenerate_landmarks():
    sc = ph.make_structure_scene(5, n_points=120)
    out, iters = orc.optimize_points(sc["views"], sc["obs_begin"], sc["obs_view"], sc["obs_f"], sc["pos0"], n_iter=5)
    nobs = np.diff(sc["obs_begin"])
    lone = nobs < 2
    assert lone.any() and np.array_equal(out[lone], sc["pos0"][lone]) and (iters[lone] == 0).all()
    assert np.isfinite(out[~lone]).all()           # rank-deficient and behind-the-camera starts stay finite
    good = ~lone
    good[[11, 23]] = False
    e0 = np.linalg.norm(sc["pos0"][good] - sc["pos_gt"][good], axis=1)
    e1 = np.linalg.norm(out[good] - sc["pos_gt"][good], axis=1)
    assert np.median(e1) < 0.2 * np.median(e0)
    # n_iter = 0: nothing happens
    out0, it0 = orc.optimize_points(sc["views"], sc["obs_begin"], sc["obs_view"], sc["obs_f"], sc["pos0"], n_iter=0)
    assert np.array_equal(out0, sc["pos0"]) and (it0 == 0).all()
