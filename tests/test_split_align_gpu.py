"""Patch-split Gauss-Newton (second row) through the C ABI on one GPU.

One participant: the split iteration runs the same device code on the same sums as the resident kernel, so the
states are identical (asserted to 1e-13).  Several participants are rehearsed in ONE process: the shares'
partial sums are added in the order a ring would (any order is a legal all-reduce) and the result has to agree
with the unsplit run within the summation-order tolerance 1e-9, with the same iteration counts.
"""
import copy
import ctypes as C

import numpy as np
import pytest
import torch

from svo_pro_universal_amd import _capi as capi, frontend as fe, split_align, synth

import helpers

pytestmark = pytest.mark.gpu


def share_of(sc, lo, hi):
    s = copy.copy(sc)
    s.px, s.f = sc.px[2 * lo:2 * hi].copy(), sc.f[3 * lo:3 * hi].copy()   # flat interleaved arrays
    s.pos_world, s.flags = sc.pos_world[3 * lo:3 * hi].copy(), sc.flags[lo:hi].copy()
    s.n_features = hi - lo
    return s


def run_split(gpu_ctx, opt, cams_per_share, n_workgroups=1, **mk):
    """cams_per_share: list (one entry per participant) of lists of (scene, ref_frame, cur_frame)."""
    dev = torch.device("cuda", 0)
    problems, keeps = [], []
    for cams in cams_per_share:
        pb, keep = fe.make_align_problems([cams], **mk)
        problems.append(pb)
        keeps.append(keep)
    d_state = torch.zeros(C.sizeof(capi.svoh_align_gn_state) // 8, dtype=torch.float64, device=dev)
    d_part = [torch.zeros(capi.SVOH_ALIGN_SUMS_DOUBLES, dtype=torch.float64, device=dev) for _ in problems]
    d_sums = torch.zeros(capi.SVOH_ALIGN_SUMS_DOUBLES, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.split_init(problems[0][0], d_state.data_ptr())

    def partial(level):
        for pb, buf in zip(problems, d_part):
            gpu_ctx.partial_sums(opt, pb[0], level, d_state.data_ptr(), buf.data_ptr(), n_workgroups)
        gpu_ctx.synchronize()

    def reduce_():
        d_sums.copy_(torch.stack(d_part).sum(0) if len(d_part) > 1 else d_part[0])
        torch.cuda.synchronize()

    def update(level, it):
        return gpu_ctx.gn_update(opt, problems[0][0], level, it, d_sums.data_ptr(), d_state.data_ptr())

    return split_align.gauss_newton_split(opt.max_level, opt.min_level, opt.max_iter, partial, reduce_, update)


def frames_of(gpu_ctx, sc, n_levels=5):
    return gpu_ctx.build_pyramid(sc.img_ref, n_levels), gpu_ctx.build_pyramid(sc.img_cur, n_levels)


@pytest.mark.parametrize("illum", [0, 1])
def test_one_participant_walks_the_resident_kernels_states(gpu_ctx, illum):
    sc = helpers.small_scene(71, n=500, gain=1.03 if illum else 1.0, offset=2.0 if illum else 0.0, border_features=40)
    fr, fc = frames_of(gpu_ctx, sc)
    Tp = synth.SE3(synth.quat_from_axis_angle([0.3, -1, 0.2], 0.004), [0.003, -0.002, 0.001])
    prior = helpers.make_prior(Tp, 0.5, 0.7, alpha=0.01, beta=-0.5, lambda_alpha=0.2 * illum, lambda_beta=0.2 * illum)
    for mk in (dict(), dict(prior=prior)):
        opt = capi.default_align_options(min_level=1, estimate_illumination_gain=illum,
                                         estimate_illumination_offset=illum)
        gpb, keep = fe.make_align_problems([[(sc, fr, fc)]], **mk)
        whole = gpu_ctx.sparse_align(opt, gpb)[0]
        res = run_split(gpu_ctx, opt, [[(sc, fr, fc)]], **mk)
        assert res.iters == list(whole.iters) and res.n_meas == list(whole.n_meas)
        assert res.status == whole.status == 0
        assert helpers.se3_max_abs_diff(res.state.T_icur_iref, whole.T_icur_iref) < 1e-13
        assert abs(res.state.alpha - whole.alpha) < 1e-13 and abs(res.state.beta - whole.beta) < 1e-11
        for l in range(capi.SVOH_MAX_LEVELS):
            if whole.n_meas[l]:
                assert abs(res.chi2[l] - whole.chi2[l]) <= 1e-12 * abs(whole.chi2[l])


@pytest.mark.parametrize("shares", [2, 3])
def test_shares_sum_to_the_unsplit_problem(gpu_ctx, shares):
    sc = helpers.small_scene(72, n=600, border_features=50, invalid_fraction=0.05)
    fr, fc = frames_of(gpu_ctx, sc)
    opt = capi.default_align_options(min_level=0)
    gpb, keep = fe.make_align_problems([[(sc, fr, fc)]])
    whole = gpu_ctx.sparse_align(opt, gpb)[0]
    cuts = np.linspace(0, sc.n_features, shares + 1).astype(int)
    parts = [[(share_of(sc, cuts[k], cuts[k + 1]), fr, fc)] for k in range(shares)]
    res = run_split(gpu_ctx, opt, parts)
    assert res.iters == list(whole.iters) and res.n_meas == list(whole.n_meas)
    assert helpers.se3_max_abs_diff(res.state.T_icur_iref, whole.T_icur_iref) < 1e-9
    # and the truth is recovered
    err = synth.se3_error(synth.SE3.from7(fe.se3_to_numpy(res.state.T_icur_iref)), sc.T_icur_iref_gt)
    assert err[0] < 1e-3 and err[1] < 5e-3


@pytest.mark.parametrize("n_workgroups", [0, 2, 7, 64])
def test_workgroups_of_one_gpu_share_a_problem(gpu_ctx, n_workgroups):
    """The same split inside one GPU: the participant's features spread over several workgroups (0 = automatic),
    host and device feature arrays, one and two cameras."""
    a = helpers.small_scene(76, n=900, border_features=60, invalid_fraction=0.05)
    b = synth.make_align_scene(76, n_features=300, cam=synth.Camera.euroc_like(), border_features=10)
    fa, fb = frames_of(gpu_ctx, a), frames_of(gpu_ctx, b)
    opt = capi.default_align_options(min_level=1)
    for cams in ([(a,) + fa], [(a,) + fa, (b,) + fb]):
        gpb, keep = fe.make_align_problems([cams])
        whole = gpu_ctx.sparse_align(opt, gpb)[0]
        res = run_split(gpu_ctx, opt, [cams], n_workgroups=n_workgroups)
        assert res.iters == list(whole.iters) and res.n_meas == list(whole.n_meas)
        assert helpers.se3_max_abs_diff(res.state.T_icur_iref, whole.T_icur_iref) < 1e-9
    # features resident in HBM
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(x).to(dev) for x in (a.px, a.f, a.pos_world, a.flags)]
    dp = dict(px=t[0].data_ptr(), f=t[1].data_ptr(), pos_world=t[2].data_ptr(), flags=t[3].data_ptr())
    torch.cuda.synchronize()
    gpb, keep = fe.make_align_problems([[(a,) + fa]])
    whole = gpu_ctx.sparse_align(opt, gpb)[0]
    res = run_split(gpu_ctx, opt, [[(a,) + fa + (dp,)]], n_workgroups=n_workgroups)
    assert res.iters == list(whole.iters)
    assert helpers.se3_max_abs_diff(res.state.T_icur_iref, whole.T_icur_iref) < 1e-9


def test_stereo_one_camera_per_participant(gpu_ctx):
    """SURVEY.md 8(e), third row: one camera of the bundle per participant, same 74-scalar sum."""
    a = helpers.small_scene(73, n=250, border_features=30)
    b = synth.make_align_scene(73, n_features=220, cam=synth.Camera.euroc_like(), border_features=10)
    fa, fb = frames_of(gpu_ctx, a), frames_of(gpu_ctx, b)
    opt = capi.default_align_options(min_level=1)
    gpb, keep = fe.make_align_problems([[(a,) + fa, (b,) + fb]])
    whole = gpu_ctx.sparse_align(opt, gpb)[0]
    res = run_split(gpu_ctx, opt, [[(a,) + fa], [(b,) + fb]])
    assert res.iters == list(whole.iters) and res.n_meas == list(whole.n_meas)
    assert helpers.se3_max_abs_diff(res.state.T_icur_iref, whole.T_icur_iref) < 1e-9


def test_empty_share_and_nothing_visible(gpu_ctx):
    sc = helpers.small_scene(74, n=200)
    fr, fc = frames_of(gpu_ctx, sc)
    opt = capi.default_align_options(min_level=2)
    gpb, keep = fe.make_align_problems([[(sc, fr, fc)]])
    whole = gpu_ctx.sparse_align(opt, gpb)[0]
    empty = share_of(sc, 0, 0)
    res = run_split(gpu_ctx, opt, [[(sc, fr, fc)], [(empty, fr, fc)]])
    assert res.iters == list(whole.iters)
    assert helpers.se3_max_abs_diff(res.state.T_icur_iref, whole.T_icur_iref) < 1e-13
    # no selected feature anywhere: zero sums, zero step, every level leaves after one evaluation
    none = copy.copy(sc)
    none.flags = np.zeros_like(sc.flags)
    res = run_split(gpu_ctx, opt, [[(none, fr, fc)]])
    assert res.iters[2:5] == [1, 1, 1] and res.n_meas[2:5] == [0, 0, 0]
    assert helpers.se3_max_abs_diff(res.state.T_icur_iref, gpb[0].T_icur_iref) == 0.0


def test_argument_errors(gpu_ctx):
    sc = helpers.small_scene(75, n=50)
    fr, fc = frames_of(gpu_ctx, sc)
    opt = capi.default_align_options()
    gpb, keep = fe.make_align_problems([[(sc, fr, fc)]])
    buf = torch.zeros(80, dtype=torch.float64, device="cuda:0")
    with pytest.raises(fe.SvohError):
        gpu_ctx.partial_sums(opt, gpb[0], 7, buf.data_ptr(), buf.data_ptr())
    with pytest.raises(fe.SvohError):
        gpu_ctx.partial_sums(opt, gpb[0], 2, 0, buf.data_ptr())
    with pytest.raises(fe.SvohError):
        gpu_ctx.gn_update(opt, gpb[0], 2, -1, buf.data_ptr(), buf.data_ptr())
    with pytest.raises(fe.SvohError):
        gpu_ctx.partial_sums(opt, gpb[0], 2, buf.data_ptr(), buf.data_ptr(), n_workgroups=-3)
    # more workgroups than features: empty shares add zeros
    gpu_ctx.split_init(gpb[0], buf.data_ptr())
    out1 = torch.zeros(74, dtype=torch.float64, device="cuda:0")
    out2 = torch.zeros(74, dtype=torch.float64, device="cuda:0")
    gpu_ctx.partial_sums(opt, gpb[0], 2, buf.data_ptr(), out1.data_ptr(), n_workgroups=1)
    gpu_ctx.partial_sums(opt, gpb[0], 2, buf.data_ptr(), out2.data_ptr(), n_workgroups=200)
    gpu_ctx.synchronize()
    a, b = out1.cpu().numpy(), out2.cpu().numpy()
    assert a[73] == b[73] > 0 and np.abs(a - b).max() <= 1e-12 * np.abs(a).max()


@pytest.mark.parametrize("tag", ["pinhole", "radtan"])
def test_partial_sums_against_the_committed_fixture(gpu_ctx, tag):
    """tests/golden/align_small.npz holds H, g, n per level of the golden scenes: the 74-double block of
    svoh_sparse_align_partial_sums must carry the same numbers, with one workgroup and with several."""
    z = np.load(helpers.GOLDEN)
    sc = helpers.scene_from_golden(z, tag)
    fr, fc = gpu_ctx.build_pyramid(sc.img_ref, 4), gpu_ctx.build_pyramid(sc.img_cur, 4)
    opt = capi.default_align_options(**helpers.GOLDEN_OPTION_SETS["plain"])
    gpb, keep = fe.make_align_problems([[(sc, fr, fc)]])
    dev = torch.device("cuda", 0)
    d_state = torch.zeros(C.sizeof(capi.svoh_align_gn_state) // 8, dtype=torch.float64, device=dev)
    d_sums = torch.zeros(capi.SVOH_ALIGN_SUMS_DOUBLES, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.split_init(gpb[0], d_state.data_ptr())
    q = "%s/plain/" % tag
    for level in range(opt.min_level, opt.max_level + 1):
        for wg in (1, 5):
            gpu_ctx.partial_sums(opt, gpb[0], level, d_state.data_ptr(), d_sums.data_ptr(), wg)
            gpu_ctx.synchronize()
            s = d_sums.cpu().numpy()
            H, g = s[:64].reshape(8, 8).T, s[64:72]
            Hz, gz = z[q + "H%d" % level], z[q + "g%d" % level]
            assert int(s[73]) == int(z[q + "chi2_nmeas%d" % level][1])
            assert np.abs(H - Hz).max() <= 1e-10 * np.abs(Hz).max() and np.abs(g - gz).max() <= 1e-10 * np.abs(gz).max()
            chi2 = z[q + "chi2_nmeas%d" % level][0]
            assert abs(s[72] / s[73] - chi2) <= 1e-4 * abs(chi2)     # the reference (and the oracle) sum chi2 in float
