"""Patch-split Gauss-Newton driver (second row).

One alignment problem, N patches split over G participants (GPUs / ranks).  Every participant holds the same
frames and its share of the features; per iteration each computes the normal equations of its share at the common
state, the 74 doubles are all-reduced (RCCL over xGMI), and every participant applies the identical update, so
all of them walk through the same states -- the loop below is MiniLeastSquaresSolver::optimizeGaussNewton
(src/vikit/vikit_solver/include/vikit/solver/implementation/mini_least_squares_solver.hpp:42-107) driven by
SparseImgAlign::run's level loop (src/svo_img_align/src/sparse_img_align.cpp:80-96) with the reduction pulled
out between evaluateError and the solve.  It is latency-bound (two launches, two host synchronisations and one
all-reduce per iteration) and pays only for very large N per participant; DESIGN.md section 6 has the numbers.

The driver is independent of how the three steps are carried out: the product binds them to the C ABI
(`bind_context`), the CPU rehearsal in tests/ binds them to the oracle.
"""
from . import _capi as capi


class SplitResult(object):
    def __init__(self):
        self.status = 0
        self.iters = [0] * capi.SVOH_MAX_LEVELS
        self.n_meas = [0] * capi.SVOH_MAX_LEVELS
        self.chi2 = [0.0] * capi.SVOH_MAX_LEVELS
        self.state = None          # the last state the update step returned
        self.n_evaluations = 0


def gauss_newton_split(max_level, min_level, max_iter, partial_sums, all_reduce, gn_update):
    """partial_sums(level): normal equations of the local share at the current state into the sums buffer;
    all_reduce(): sum that buffer over the participants, in place; gn_update(level, iter) -> state with
    .level_done, .stop, .status, .chi2, .n_meas after one solver iteration on the summed buffer."""
    res = SplitResult()
    for level in range(max_level, min_level - 1, -1):
        for it in range(max_iter):
            partial_sums(level)
            all_reduce()
            st = gn_update(level, it)
            res.n_evaluations += 1
            res.iters[level] = it + 1
            res.n_meas[level] = int(st.n_meas)
            res.chi2[level] = float(st.chi2)
            res.state = st
            res.status = int(st.status)
            if st.level_done:
                break
    return res


def bind_context(ctx, opt, problem, d_state, d_sums, all_reduce, n_workgroups=0):
    """Run the split iteration through the C ABI.  d_state / d_sums: device addresses of a
    svoh_align_gn_state and of SVOH_ALIGN_SUMS_DOUBLES doubles (caller-owned, e.g. torch tensors);
    all_reduce(): sums d_sums over the participants (it is called after the context stream was drained and
    must itself return with the sum complete)."""
    ctx.split_init(problem, d_state)

    def partial(level):
        ctx.partial_sums(opt, problem, level, d_state, d_sums, n_workgroups)
        ctx.synchronize()

    def update(level, it):
        return ctx.gn_update(opt, problem, level, it, d_sums, d_state)

    return gauss_newton_split(opt.max_level, opt.min_level, opt.max_iter, partial, all_reduce, update)
