"""Second reading ("second opinion") of SparseImgAlign's Gauss-Newton LOOP, in NumPy, on top of the per-iteration
normal equations of np_restatement.evaluate.  Written from the reference's files, not from oracle/svo_oracle.c:

  SparseImgAlign::run                       src/svo_img_align/src/sparse_img_align.cpp:34-117
      the state {T_icur_iref, alpha, beta}, one optimize() per pyramid level from max_level down to min_level
  MiniLeastSquaresSolver::optimizeGaussNewton
                                            src/vikit/vikit_solver/include/vikit/solver/implementation/mini_least_squares_solver.hpp:42-107
      old_state = state at the start of the level; per iteration: evaluateError, applyPrior (if a prior is set), solve,
      roll back and leave the level when the solve failed -- or when an EARLIER level's did: stop_ is cleared by reset()
      only (:240-250) -- else update, and leave the level when max |dx| < eps.  stop_when_error_increases is false
      (mini_least_squares_solver.h:40), so chi2 never steers the loop.
  solveDefaultImpl                          :253-262   dx = H.ldlt().solve(g); failure iff dx[0] is NaN
  SparseImgAlignBase::update                src/svo_img_align/src/sparse_img_align_base.cpp:64-75
  SparseImgAlignBase::applyPrior            :77-107    I_prior_ is rebuilt at iteration 0 of EVERY level (iter_ restarts
                                                       with each optimize()), from that iteration's Hessian
  setWeightedPrior                          :44-62

Machinery that differs from the C oracle on purpose: numpy.linalg.solve on the active block instead of a restated
Eigen LDLT (the rows / columns of parameters that are not estimated are exactly zero; Eigen's LDLT<Lower> leaves their
pivots at zero and its D^-1 step returns zero there), rotation matrices inside evaluate, the exp / log of
np_restatement_pose.  Agreement with the oracle is therefore expected to ~1e-9 on the pose, and exactly on everything
discrete: evaluateError calls per level, measurements per level, status.
"""
import numpy as np

import np_restatement as n0
from np_restatement_direct import Tf
from np_restatement_pose import tf_exp, tf_log


def solve_ldlt_semantics(H, g):
    """H.ldlt().solve(g) for the matrices this loop produces: symmetric positive semi-definite, with exactly-zero rows
    and columns for the parameters that are not estimated.  Those components of the solution are zero."""
    active = np.array([bool(np.any(H[i] != 0.0)) for i in range(H.shape[0])])
    dx = np.zeros(H.shape[0])
    if active.any():
        idx = np.nonzero(active)[0]
        dx[idx] = np.linalg.solve(H[np.ix_(idx, idx)], g[idx])
    return dx


def update(state, dx):
    """sparse_img_align_base.cpp:64-75"""
    T, alpha, beta = state
    Tn = T * tf_exp(-dx[:6])
    an = (alpha - dx[6]) / (1.0 + dx[6])
    bn = (beta - dx[7]) / (1.0 + dx[6])
    Tn = Tf(Tn.q / np.sqrt(float(Tn.q @ Tn.q)), Tn.t)   # toImplementation().normalize()
    return (Tn, an, bn)


def run(cams, max_level, min_level, P, T_init, alpha_init=0.0, beta_init=0.0, max_iter=10, eps=0.0005,
        est_alpha=False, est_beta=False, robust=False, weight_scale=10.0, prior=None):
    """cams: list of (scene, ref_levels, cur_levels), one per camera of the bundle (the normal equations of the cameras
    are summed: sparse_img_align.cpp:138-154).  prior: None or a dict {T, alpha, beta, lambda_rot, lambda_trans,
    lambda_alpha, lambda_beta} (setWeightedPrior).  Returns a dict: T (Tf), alpha, beta, iters / n_meas per level
    (evaluateError calls, residuals of the last one), status (0 ok, 2 the solve failed: state rolled back)."""
    state = (Tf(np.array(T_init.q, np.float64), np.array(T_init.t, np.float64)), float(alpha_init), float(beta_init))
    iters, n_meas = {}, {}
    stop = False                                     # MiniLeastSquaresSolver::stop_: sticky until reset()
    status = 0
    I_prior = np.zeros(8)
    for level in range(max_level, min_level - 1, -1):
        old_state = state                            # hpp:45
        iters[level], n_meas[level] = 0, 0
        for it in range(max_iter):
            H, g, n_total = np.zeros((8, 8)), np.zeros(8), 0
            for scene, ref_levels, cur_levels in cams:
                Hc, gc, _chi2, nm, _vis = n0.evaluate(scene, ref_levels, cur_levels, level, P, state[0].q, state[0].t,
                                                      max_level=max_level, alpha=state[1], beta=state[2],
                                                      est_alpha=est_alpha, est_beta=est_beta, robust=robust,
                                                      weight_scale=weight_scale)
                H += Hc; g += gc; n_total += nm
            iters[level] += 1
            n_meas[level] = n_total
            if prior is not None:                    # applyPrior
                if it == 0:
                    I_prior = np.zeros(8)
                    I_prior[0:3] = prior["lambda_trans"] * max(abs(H[j, j]) for j in range(3))
                    I_prior[3:6] = prior["lambda_rot"] * max(abs(H[j, j]) for j in range(3, 6))
                    I_prior[6] = prior["lambda_alpha"] * H[6, 6]
                    I_prior[7] = prior["lambda_beta"] * H[7, 7]
                H = H + np.diag(I_prior)
                g = g.copy()
                g[:6] += I_prior[:6] * tf_log(prior["T"].inverse() * state[0])
                g[6] += I_prior[6] * (prior["alpha"] - state[1])
                g[7] += I_prior[7] * (prior["beta"] - state[2])
            dx = solve_ldlt_semantics(H, g)
            if np.isnan(dx[0]):
                stop = True
            if stop:                                 # roll back, leave the level (and every later level after one evaluation)
                state = old_state
                status = 2
                break
            new_state = update(state, dx)
            old_state = state
            state = new_state
            if np.max(np.abs(dx)) < eps:
                break
    return dict(T=state[0], alpha=state[1], beta=state[2], iters=iters, n_meas=n_meas, status=status)
