"""The patch-split Gauss-Newton loop (svo_pro_universal_amd/split_align.py) over a real process group on CPU:
two gloo ranks, each with half of the features, the three steps bound to the oracle (tests may; the product
binds them to the C ABI -- tests/test_split_align_gpu.py).  The ranks must walk through identical states and end
where the oracle's unsplit run ends."""
import copy
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _State(object):
    pass


def oracle_steps(orc, opt, sc, lo, hi, levels, all_reduce_np):
    """partial / reduce / update closures over one share [lo, hi) of the scene's features."""
    from svo_pro_universal_amd import _capi as capi
    lib = orc.load()
    share = copy.copy(sc)
    share.px, share.f = sc.px[2 * lo:2 * hi].copy(), sc.f[3 * lo:3 * hi].copy()
    share.pos_world, share.flags = sc.pos_world[3 * lo:3 * hi].copy(), sc.flags[lo:hi].copy()
    share.n_features = hi - lo
    pb = orc.problem_from_scenes([(share,) + levels])
    sums = np.zeros(74)
    st = _State()
    st.T, st.alpha, st.beta = pb.c.T_icur_iref, 0.0, 0.0
    st.level_done, st.stop, st.status, st.chi2, st.n_meas = 0, 0, 0, 0.0, 0

    def partial(level):
        pb.c.T_icur_iref, pb.c.alpha_init, pb.c.beta_init = st.T, st.alpha, st.beta
        H, g, chi2, nm, _ = orc.sparse_align_evaluate(opt, pb, level)
        sums[:64] = H.T.reshape(-1)   # column-major like the C ABI (H is symmetric anyway)
        sums[64:72] = g
        sums[72], sums[73] = chi2 * nm, nm

    def reduce_():
        all_reduce_np(sums)

    def update(level, it):
        # MiniLeastSquaresSolver::optimizeGaussNewton body, mini_least_squares_solver.hpp:61-101, without prior
        H = np.ascontiguousarray(sums[:64].reshape(8, 8).T)
        g = sums[64:72].copy()
        dx = np.zeros(8)
        ok = lib.orc_ldlt_solve(8, H.ctypes.data, g.ctypes.data, dx.ctypes.data)
        st.n_meas = int(sums[73])
        st.chi2 = sums[72] / sums[73] if sums[73] else float("nan")
        st.level_done = 0
        if not ok:
            st.stop, st.status, st.level_done = 1, 2, 1
            return st
        E, Tn = capi.svoh_se3(), capi.svoh_se3()
        mdx = (-dx[:6]).copy()
        lib.orc_se3_exp(mdx.ctypes.data_as(C.POINTER(C.c_double)), C.byref(E))
        lib.orc_se3_mul(C.byref(st.T), C.byref(E), C.byref(Tn))
        q = np.array([Tn.q[k] for k in range(4)])
        q /= np.linalg.norm(q)
        for k in range(4):
            Tn.q[k] = q[k]
        st.T = Tn
        if np.abs(dx).max() < opt.eps:
            st.level_done = 1
        return st

    return partial, reduce_, update, st


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    from svo_pro_universal_amd import _capi as capi, dist_utils as du, split_align, synth
    from oracle import oracle as orc
    import helpers
    dist = du.init("gloo", rank, world)
    sc = helpers.small_scene(81, n=160, border_features=10)     # the same scene on every rank
    levels = helpers.scene_pyramids(orc, sc, 5)
    opt = capi.default_align_options(min_level=2)
    lo, hi = du.shard_range(sc.n_features, rank, world)

    def all_reduce_np(a):
        t = torch.from_numpy(a)
        dist.all_reduce(t)          # SUM, in place (shares memory with the numpy array)

    partial, reduce_, update, st = oracle_steps(orc, opt, sc, lo, hi, levels, all_reduce_np)
    res = split_align.gauss_newton_split(opt.max_level, opt.min_level, opt.max_iter, partial, reduce_, update)
    T = [st.T.q[k] for k in range(4)] + [st.T.t[k] for k in range(3)]
    q.put((rank, lo, hi, res.iters, res.n_meas, T, res.n_evaluations))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_patch_split_matches_the_unsplit_oracle():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from svo_pro_universal_amd import _capi as capi
    from oracle import oracle as orc
    import helpers
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, it0, nm0, T0, ne0), (r1, lo1, hi1, it1, nm1, T1, ne1) = out
    assert lo0 == 0 and hi0 == lo1 and hi1 == 170
    assert it0 == it1 and nm0 == nm1 and ne0 == ne1
    assert T0 == T1                                   # identical states on both ranks, bit for bit
    sc = helpers.small_scene(81, n=160, border_features=10)
    opt = capi.default_align_options(min_level=2)
    pb = orc.problem_from_scenes([(sc,) + helpers.scene_pyramids(orc, sc, 5)])
    n, whole, _ = orc.sparse_align_run(opt, pb)
    assert it0 == list(whole.iters) and nm0 == list(whole.n_meas)
    Tw = [whole.T_icur_iref.q[k] for k in range(4)] + [whole.T_icur_iref.t[k] for k in range(3)]
    assert np.abs(np.array(T0) - np.array(Tw)).max() < 1e-9


def test_driver_loop_bookkeeping():
    """Level order, iteration cap and early exit of the loop itself, no numerics."""
    from svo_pro_universal_amd import split_align
    calls = []

    class S(object):
        level_done, stop, status, chi2, n_meas = 0, 0, 0, 1.0, 16

    def update(level, it):
        s = S()
        s.level_done = 1 if (level == 3 and it == 1) else 0
        calls.append(("u", level, it))
        return s

    res = split_align.gauss_newton_split(4, 2, 3, lambda l: calls.append(("p", l)), lambda: calls.append(("r",)),
                                         update)
    assert res.iters[2:5] == [3, 2, 3] and res.n_evaluations == 8
    assert calls[:3] == [("p", 4), ("r",), ("u", 4, 0)]
    assert [c for c in calls if c[0] == "u"][3:5] == [("u", 3, 0), ("u", 3, 1)]
