"""Round 5: the C-ABI entries a lock-step driver of many camera streams stands on (include/svo_hip.h) -- keyed alignment
launches, batches staged in place, the candidate projection of many frames in one launch, the detector for many frames,
images at separate (page-locked) addresses, prefetch on a second stream.  Each must give EXACTLY what the one-at-a-time
entry gives (bit for bit: the byte-identical trajectories of tests/test_mini_frontend_gpu.py rest on it), and refuse
misuse with an error instead of reading or writing where it should not."""
import ctypes as C

import numpy as np
import pytest

from svo_pro_universal_amd import _capi as capi, frontend as fe, synth

pytestmark = pytest.mark.gpu


def result_bits(r):
    return (r.status, r.n_fts_to_track, tuple(r.T_icur_iref.q), tuple(r.T_icur_iref.t), r.alpha, r.beta, tuple(r.iters), tuple(r.n_meas), tuple(r.chi2), r.n_patch_iters)


def test_keyed_launch_gives_every_problem_the_result_of_its_own_launch(gpu_ctx):
    """svoh_sparse_align_geometry_key / _enqueue_keyed: problems of 60 ... 900 features pick four different geometries when
    launched alone (lanes per patch, 256-thread latency build, 512 threads, cluster of workgroups); launched together by key
    every one must come back with the bits of its own launch, and a candidate projection queued behind several keyed
    launches must find every problem's pose on the device."""
    ctx = gpu_ctx
    opt = capi.default_align_options(max_level=4, min_level=2)
    sizes = [60, 100, 180, 180, 300, 400, 600, 720, 900, 180, 600]
    scenes, frames, pbs = [], [], []
    for i, n in enumerate(sizes):
        sc = synth.make_align_scene(300 + i, n_features=n, patch_size=4)
        fr, fc = ctx.build_pyramid(sc.img_ref, 5), ctx.build_pyramid(sc.img_cur, 5)
        p, keep = fe.make_align_problems([[(sc, fr, fc)]])
        scenes.append((sc, keep)); frames += [fr, fc]; pbs.append(p[0])
    alone = [result_bits(ctx.sparse_align(opt, (capi.svoh_align_problem * 1)(p))[0]) for p in pbs]
    keys = []
    for p in pbs:
        k = C.c_int32()
        ctx._check(ctx.lib.svoh_sparse_align_geometry_key(ctx.h, C.byref(opt), C.byref(p), C.byref(k)))
        keys.append(k.value)
    assert len(set(keys)) >= 3, keys
    order, groups = [], []
    for k in dict.fromkeys(keys):
        members = [i for i, kk in enumerate(keys) if kk == k]
        groups.append((k, members)); order += members
    for k, members in groups:
        arr = (capi.svoh_align_problem * len(members))(*[pbs[i] for i in members])
        ctx._check(ctx.lib.svoh_sparse_align_enqueue_keyed(ctx.h, C.byref(opt), len(members), arr, k))
    # behind them: every problem's candidate projection composed from ITS result on the device
    res = ctx.sparse_align_fetch_all(len(pbs))
    for pos, i in enumerate(order):
        assert result_bits(res[pos]) == alone[i], (sizes[i], keys[i])
    # a key of another kind of problem is refused only when it is no key at all
    bad = C.c_int32(12345)
    arr = (capi.svoh_align_problem * 1)(pbs[0])
    assert ctx.lib.svoh_sparse_align_enqueue_keyed(ctx.h, C.byref(opt), 1, arr, bad) != 0
    for f in frames:
        ctx.release_frame(f)


def test_shared_geometry_classes_one_key_below_the_cluster_threshold(gpu_ctx):
    """svoh_set_align_geometry_classes(ctx, 1) (round 6): every problem below 512 patches gets ONE key -- mono problems of 40 ... 500 patches,
    and the two-camera bundles among themselves -- so a lock-step round of streams of different sizes is one launch; the setting also governs a
    launch of a single problem, so alone == in the shared launch, bit for bit; larger problems keep the cluster rule of their size; against the
    default classes the results agree to summation order (poses 1e-12, iteration counts exact) and differ in bits somewhere."""
    ctx = gpu_ctx
    opt = capi.default_align_options(max_level=4, min_level=2)
    sizes = [40, 60, 100, 128, 180, 256, 257, 300, 400, 500, 600, 900]
    keep_all, frames, pbs = [], [], []
    for i, n in enumerate(sizes):
        sc = synth.make_align_scene(1300 + i, n_features=n, patch_size=4)
        fr, fc = ctx.build_pyramid(sc.img_ref, 5), ctx.build_pyramid(sc.img_cur, 5)
        p, keep = fe.make_align_problems([[(sc, fr, fc)]])
        keep_all.append((sc, keep)); frames += [fr, fc]; pbs.append(p[0])
    # two rig bundles of different sizes (8 parameters)
    opt8 = capi.default_align_options(max_level=4, min_level=2, estimate_illumination_gain=1, estimate_illumination_offset=1)
    rigs = []
    for i, n in enumerate((90, 170)):
        scs = [synth.make_align_scene(1400 + 2 * i + c, n_features=n, patch_size=4, gain=1.03, offset=2.0) for c in range(2)]
        fr = [(ctx.build_pyramid(sc.img_ref, 5), ctx.build_pyramid(sc.img_cur, 5)) for sc in scs]
        p, keep = fe.make_align_problems([[(sc, a, b) for sc, (a, b) in zip(scs, fr)]])
        keep_all.append((scs, keep)); frames += [h for ab in fr for h in ab]; rigs.append(p[0])

    def key_of(o, p):
        k = C.c_int32()
        ctx._check(ctx.lib.svoh_sparse_align_geometry_key(ctx.h, C.byref(o), C.byref(p), C.byref(k)))
        return k.value

    default_keys = [key_of(opt, p) for p in pbs]
    default_alone = [ctx.sparse_align(opt, (capi.svoh_align_problem * 1)(p))[0] for p in pbs]
    default_bits = [result_bits(r) for r in default_alone]
    assert len(set(default_keys[:10])) >= 3
    try:
        ctx.set_align_geometry_classes(True)
        keys = [key_of(opt, p) for p in pbs]
        assert len(set(keys[:10])) == 1 and keys[10] == default_keys[10] and keys[11] == default_keys[11], [hex(k) for k in keys]
        rig_keys = [key_of(opt8, p) for p in rigs]
        assert len(set(rig_keys)) == 1 and rig_keys[0] != keys[0]
        alone = [ctx.sparse_align(opt, (capi.svoh_align_problem * 1)(p))[0] for p in pbs]
        arr = (capi.svoh_align_problem * 10)(*pbs[:10])
        ctx._check(ctx.lib.svoh_sparse_align_enqueue_keyed(ctx.h, C.byref(opt), 10, arr, keys[0]))
        together = ctx.sparse_align_fetch_all(10)
        for i in range(10):
            assert result_bits(together[i]) == result_bits(alone[i]), sizes[i]
        rig_alone = [ctx.sparse_align(opt8, (capi.svoh_align_problem * 1)(p))[0] for p in rigs]
        arr = (capi.svoh_align_problem * 2)(*rigs)
        ctx._check(ctx.lib.svoh_sparse_align_enqueue_keyed(ctx.h, C.byref(opt8), 2, arr, rig_keys[0]))
        rig_together = ctx.sparse_align_fetch_all(2)
        for i in range(2):
            assert result_bits(rig_together[i]) == result_bits(rig_alone[i])
        # against the default classes: the same alignment up to the order of the sums
        differs = 0
        for i, (a, d) in enumerate(zip(alone, default_alone)):
            assert tuple(a.iters) == tuple(d.iters) and tuple(a.n_meas) == tuple(d.n_meas) and a.n_fts_to_track == d.n_fts_to_track, sizes[i]
            assert np.allclose(list(a.T_icur_iref.q) + list(a.T_icur_iref.t), list(d.T_icur_iref.q) + list(d.T_icur_iref.t), rtol=0, atol=1e-12), sizes[i]
            differs += result_bits(a) != default_bits[i]
        assert differs >= 1, "shared classes changed no problem's bits: the default classes were one class already?"
        assert ctx.lib.svoh_set_align_geometry_classes(ctx.h, 2) != 0
    finally:
        ctx.set_align_geometry_classes(False)
    assert [key_of(opt, p) for p in pbs] == default_keys
    for f in frames:
        ctx.release_frame(f)


def test_keyed_cluster_launch_of_more_problems_than_arrival_counters(gpu_ctx):
    """70 - 90 problems of 512 ... 874 patches under ONE cluster key (the round after a keyframe of a large
    lock-step group) -- with three workgroups per problem a 256-CU device takes 85 per launch, the arrival counters hold 64:
    the keyed call must cut its launches at 64, every problem must come back with the bits of its own launch, and a plain
    cluster launch of more than 64 problems is refused."""
    ctx = gpu_ctx
    opt = capi.default_align_options(max_level=4, min_level=2)
    scenes = [synth.make_align_scene(700 + i, n_features=n, patch_size=4) for i, n in enumerate((520, 600, 700, 860))]
    frames, pbs = [], []
    for sc in scenes:
        fr, fc = ctx.build_pyramid(sc.img_ref, 5), ctx.build_pyramid(sc.img_cur, 5)
        p, keep = fe.make_align_problems([[(sc, fr, fc)]])
        frames += [fr, fc]; pbs.append((p[0], keep))
    keys = []
    for p, _ in pbs:
        k = C.c_int32()
        ctx._check(ctx.lib.svoh_sparse_align_geometry_key(ctx.h, C.byref(opt), C.byref(p), C.byref(k)))
        keys.append(k.value)
    assert len(set(keys)) == 1 and (keys[0] & 0xff) == 3, [hex(k) for k in keys]    # three workgroups per problem
    alone = [result_bits(ctx.sparse_align(opt, (capi.svoh_align_problem * 1)(p))[0]) for p, _ in pbs]
    for n in (65, 86, 130):
        arr = (capi.svoh_align_problem * n)(*[pbs[i % 4][0] for i in range(n)])
        ctx._check(ctx.lib.svoh_sparse_align_enqueue_keyed(ctx.h, C.byref(opt), n, arr, keys[0]))
        res = ctx.sparse_align_fetch_all(n)
        for i in range(n):
            assert result_bits(res[i]) == alone[i % 4], (n, i)
    for f in frames:
        ctx.release_frame(f)


def test_failed_enqueue_between_two_queued_launches_leaves_their_staging_alone(gpu_ctx):
    """the staging block of an alignment launch is chosen before the launch can still fail; a failing
    call between two queued launches must not make the next one overwrite the block the first is still uploading from."""
    ctx = gpu_ctx
    opt = capi.default_align_options(max_level=4, min_level=1)
    sc = synth.make_align_scene(41, n_features=2000, patch_size=4)
    fr, fc = ctx.build_pyramid(sc.img_ref, 5), ctx.build_pyramid(sc.img_cur, 5)
    pbs, keep = fe.make_align_problems([[(sc, fr, fc)]] * 64)
    ref = result_bits(ctx.sparse_align(opt, pbs)[0])
    bad, keep2 = fe.make_align_problems([[(sc, fr, fc)]])
    bad[0].cams[0].cur_frame = 987654321   # unknown handle: fails after the staging block was picked
    for _ in range(3):
        ctx.sparse_align_enqueue(opt, pbs)
        assert ctx.lib.svoh_sparse_align_enqueue(ctx.h, C.byref(opt), 1, bad) != 0
        ctx.sparse_align_enqueue(opt, pbs)
        res = ctx.sparse_align_fetch_all(128)
        assert all(result_bits(r) == ref for r in res)
    ctx.release_frame(fr); ctx.release_frame(fc)


def pinned_copy(ctx, arrays):
    total = sum(a.size for a in arrays)
    p = C.c_void_p()
    ctx._check(ctx.lib.svoh_host_alloc(ctx.h, C.c_size_t(total + 4096), C.byref(p)))
    ptrs, off = [], 0
    for a in arrays:
        C.memmove(p.value + off, np.ascontiguousarray(a).ctypes.data, a.size)
        ptrs.append(p.value + off); off += a.size
    return p, ptrs


@pytest.mark.parametrize("mem", ["pageable", "pinned", "pinned+prefetch"])
def test_pyramids_of_images_at_separate_addresses(gpu_ctx, mem):
    """svoh_build_pyramid_multi (+ _prefetch on the second stream, + fence): every level of every image equals the
    single-image builder's, for the 752-wide mixed-rule case too."""
    ctx = gpu_ctx
    rng = np.random.RandomState(5)
    for (w, h) in ((640, 480), (752, 480)):
        imgs = [rng.randint(0, 256, (h, w)).astype(np.uint8) for _ in range(5)]
        want = []
        for im in imgs:
            f, lv = ctx.build_pyramid(im, 5, return_levels=True)
            want.append(lv); ctx.release_frame(f)
        block = None
        if mem == "pageable":
            keep = [np.ascontiguousarray(im) for im in imgs]
            ptrs = [a.ctypes.data for a in keep]
            space = capi.SVOH_MEM_HOST
        else:
            block, ptrs = pinned_copy(ctx, imgs)
            space = capi.SVOH_MEM_HOST_PINNED
        arr = (C.c_void_p * len(imgs))(*ptrs)
        out = (capi.svoh_frame_t * len(imgs))()
        fn = ctx.lib.svoh_build_pyramid_multi_prefetch if mem.endswith("prefetch") else ctx.lib.svoh_build_pyramid_multi
        ctx._check(fn(ctx.h, arr, len(imgs), w, h, w, space, 5, capi.SVOH_HALFSAMPLE_REFERENCE, out))
        if mem.endswith("prefetch"):
            ctx._check(ctx.lib.svoh_prefetch_fence(ctx.h))
        for i in range(len(imgs)):
            for l in range(5):
                assert np.array_equal(ctx.download_level(out[i], l), want[i][l]), (mem, w, i, l)
            ctx.release_frame(out[i])
        if block is not None:
            ctx._check(ctx.lib.svoh_host_free(ctx.h, block))
    # more page-locked images than one gather kernel takes
    many = (C.c_void_p * 300)(*([1] * 300))
    out = (capi.svoh_frame_t * 300)()
    assert ctx.lib.svoh_build_pyramid_multi(ctx.h, many, 300, 64, 64, 64, capi.SVOH_MEM_HOST_PINNED, 3, 0, out) != 0


def test_detector_for_many_frames_equals_the_detector_for_each(gpu_ctx):
    """svoh_detect_cells_batch + svoh_detect_fill_features against svoh_detect_features, frame by frame: corners and
    edgelets, occupied cells, a feature budget that cuts the list -- every output array equal."""
    ctx = gpu_ctx
    cam = synth.Camera.euroc_like(752, 480)
    frames, occs = [], []
    rng = np.random.RandomState(3)
    opt = capi.svoh_detector_options()
    opt.cell_size, opt.max_level, opt.min_level, opt.border, opt.detect_edgelets = 30, 2, 0, 8, 1
    opt.threshold_primary, opt.threshold_secondary = 10.0, 100.0
    n_cells = int(np.ceil(752 / 30)) * int(np.ceil(480 / 30))
    for i in range(6):
        sc = synth.make_align_scene(700 + i, n_features=8, cam=cam)
        frames.append(ctx.build_pyramid(sc.img_ref if i % 2 else sc.img_cur, 5))
        occs.append((rng.uniform(size=n_cells) < (0.0, 0.3, 0.9)[i % 3]).astype(np.uint8))
    for budget in (n_cells, 150, 37):
        single = [ctx.detect_features(opt, frames[i], 752, 480, occupancy=occs[i], max_n_features=budget) for i in range(len(frames))]
        n = len(frames)
        fr = (capi.svoh_frame_t * n)(*frames)
        occ = np.ascontiguousarray(np.stack(occs))
        ck, ek, ang = np.zeros((n, n_cells), np.uint64), np.zeros((n, n_cells), np.uint64), np.zeros((n, n_cells), np.float32)
        ctx._check(ctx.lib.svoh_detect_cells_batch(ctx.h, n, fr, C.byref(opt), occ.ctypes.data, ck.ctypes.data, ek.ctypes.data, ang.ctypes.data))
        for i in range(n):
            px = np.zeros(2 * n_cells); score = np.zeros(n_cells); level = np.zeros(n_cells, np.int32); grad = np.zeros(2 * n_cells); typ = np.zeros(n_cells, np.uint8)
            m = C.c_int32()
            rc = ctx.lib.svoh_detect_fill_features(C.byref(opt), 752, 480, ck[i].ctypes.data, ek[i].ctypes.data, ang[i].ctypes.data, min(budget, n_cells), px.ctypes.data,
                                                   score.ctypes.data, level.ctypes.data, grad.ctypes.data, typ.ctypes.data, C.byref(m))
            assert rc == 0
            m = m.value
            s = single[i]
            assert m == len(s["score"]) and m > 0
            assert np.array_equal(px[:2 * m].reshape(-1, 2), s["px"]) and np.array_equal(score[:m], s["score"]) and np.array_equal(level[:m], s["level"])
            assert np.array_equal(grad[:2 * m].reshape(-1, 2), s["grad"]) and np.array_equal(typ[:m], s["type"])
    for f in frames:
        ctx.release_frame(f)


def seed_scene(ctx, seed, cam):
    sc = synth.make_align_scene(seed, n_features=10, cam=cam, rot_deg=(0.5, 1.5), trans_m=(0.05, 0.15))
    return sc, ctx.build_pyramid(sc.img_ref, 5), ctx.build_pyramid(sc.img_cur, 5)


def test_batches_staged_in_place_equal_the_host_array_batches(gpu_ctx):
    """svoh_matcher_stage + SVOH_MEM_STAGED: a direct batch and a seed batch over THREE (reference, current) frame pairs,
    filled into the context's page-locked blocks, against the same units through the host-array calls one pair at a time.
    Every output -- result codes, pixels, bearing vectors, search levels, A, states, types, success flags -- bit for bit."""
    ctx = gpu_ctx
    cam = synth.Camera.euroc_like()
    mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(cam)
    pairs = [seed_scene(ctx, 810 + i, cam) for i in range(3)]
    sets = [synth.make_seed_set(sc, 700 + 100 * i, seed=i, margin=12) for i, (sc, fr, fc) in enumerate(pairs)]
    for sd in sets:
        sd["type"][::13] = capi.FT_EDGELET_SEED_CONVERGED   # not updated by the seed update; matched by the direct matcher all the same
    # one at a time, host arrays
    want_seeds, want_direct = [], []
    for (sc, fr, fc), sd in zip(pairs, sets):
        rv = fe.make_frame_view(fr, cam, sc.T_ref_f_w, sd["mu_range"], 0)
        cv = fe.make_frame_view(fc, cam, sc.T_cur_f_w_gt, 0.0, 1)
        fb, keep = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
        ns, st, succ, mr = ctx.update_seeds_batch(mopt, dopt, [rv], cv, fb, sd["state"])
        want_seeds.append((ns, st, succ, mr, keep["type"].copy()))
        fb2, keep2 = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
        depth = sd["true_depth"] * 1.02
        # the projection estimate the reprojector would hand in: the true pixel, a little off
        x = sd["f"].reshape(-1, 3).T * sd["true_depth"]
        px_true = sc.cam.project(sc.T_w_cur.inverse().transform(sc.T_w_ref.transform(x)))
        px0 = np.ascontiguousarray((px_true + np.random.RandomState(1).uniform(-1.5, 1.5, px_true.shape)).T).ravel()
        want_direct.append((ctx.match_direct_batch(mopt, [rv], cv, fb2, depth, px0), depth, px0))
    # all three pairs in one staged direct batch + one staged seed batch
    n_each = [sd["level"].size for sd in sets]
    n = sum(n_each)
    refs = (capi.svoh_frame_view * 3)(*[fe.make_frame_view(fr, cam, sc.T_ref_f_w, sd["mu_range"], 0) for (sc, fr, fc), sd in zip(pairs, sets)])
    curs = (capi.svoh_frame_view * 3)(*[fe.make_frame_view(fc, cam, sc.T_cur_f_w_gt, 0.0, 1) for (sc, fr, fc) in pairs])
    ctx._check(ctx.lib.svoh_matcher_begin_deferred(ctx.h))
    ds, ss = capi.svoh_matcher_stage_t(), capi.svoh_matcher_stage_t()
    ctx._check(ctx.lib.svoh_matcher_stage(ctx.h, 0, n, 8, 1, C.byref(ds)))
    ctx._check(ctx.lib.svoh_matcher_stage(ctx.h, 1, n, 8, 1, C.byref(ss)))

    def view(ptr, dtype, count):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(count * np.dtype(dtype).itemsize,)).view(dtype)
    off = 0
    for k, sd in enumerate(sets):
        m = n_each[k]
        for g in (ds, ss):
            view(g.ref_frame_idx, np.int32, n)[off:off + m] = k
            view(g.cur_frame_idx, np.int32, n)[off:off + m] = k
            view(g.px, np.float64, 2 * n)[2 * off:2 * (off + m)] = sd["px"]
            view(g.f, np.float64, 3 * n)[3 * off:3 * (off + m)] = sd["f"]
            view(g.grad, np.float64, 2 * n)[2 * off:2 * (off + m)] = sd["grad"]
            view(g.level, np.int32, n)[off:off + m] = sd["level"]
            view(g.type, np.uint8, n)[off:off + m] = sd["type"]
        view(ds.depth, np.float64, n)[off:off + m] = want_direct[k][1]
        view(ds.px_cur, np.float64, 2 * n)[2 * off:2 * (off + m)] = want_direct[k][2]
        view(ss.state, np.float64, 4 * n)[4 * off:4 * (off + m)] = sd["state"]
        off += m

    def batch(g):
        fb = capi.svoh_feature_batch()
        fb.n, fb.mem_space, fb.n_cur_frames = n, capi.SVOH_MEM_STAGED, 3
        for k in ("ref_frame_idx", "cur_frame_idx", "px", "f", "grad", "level", "type"):
            setattr(fb, k, getattr(g, k))
        return fb
    fbd, fbs = batch(ds), batch(ss)
    ctx._check(ctx.lib.svoh_match_direct_batch(ctx.h, C.byref(mopt), 3, refs, curs, C.byref(fbd), ds.depth, ds.px_cur, ds.result, ds.f_cur, ds.search_level,
                                               ds.h_inv, ds.A_cur_ref))
    outs = capi.svoh_seed_match_outputs(ss.px_cur, ss.f_cur, ss.search_level, ss.A_cur_ref)
    ns_total = C.c_int32()
    ctx._check(ctx.lib.svoh_update_seeds_batch_ex(ctx.h, C.byref(mopt), C.byref(dopt), 3, refs, curs, C.byref(fbs), ss.state, ss.success, ss.result,
                                                  C.byref(ns_total), C.byref(outs)))
    # a second batch of a kind, a batch with foreign arrays: refused, and the section stays usable
    assert ctx.lib.svoh_matcher_stage(ctx.h, 1, n, 8, 1, C.byref(capi.svoh_matcher_stage_t())) != 0
    ctx._check(ctx.lib.svoh_matcher_flush(ctx.h))
    ctx._check(ctx.lib.svoh_matcher_collect(ctx.h))
    assert ns_total.value == sum(w[0] for w in want_seeds)
    off = 0
    for k in range(3):
        m = n_each[k]
        ns, st, succ, mr, typ = want_seeds[k]
        assert np.array_equal(view(ss.state, np.float64, 4 * n)[4 * off:4 * (off + m)], st)
        assert np.array_equal(view(ss.success, np.uint8, n)[off:off + m], succ) and np.array_equal(view(ss.result, np.int32, n)[off:off + m], mr)
        assert np.array_equal(view(ss.type, np.uint8, n)[off:off + m], typ)
        d = want_direct[k][0]
        assert np.array_equal(view(ds.result, np.int32, n)[off:off + m], d["result"])
        assert np.array_equal(view(ds.px_cur, np.float64, 2 * n)[2 * off:2 * (off + m)], d["px_cur"])
        assert np.array_equal(view(ds.f_cur, np.float64, 3 * n)[3 * off:3 * (off + m)], d["f_cur"])
        assert np.array_equal(view(ds.search_level, np.int32, n)[off:off + m], d["search_level"])
        assert np.array_equal(view(ds.A_cur_ref, np.float64, 4 * n)[4 * off:4 * (off + m)], d["A"])
        assert (d["result"] == capi.MATCH_SUCCESS).mean() > 0.5
        off += m
    # misuse: staging outside a section; a staged batch with arrays that are not the staged ones; more frames than staged for
    assert ctx.lib.svoh_matcher_stage(ctx.h, 0, 10, 4, 1, C.byref(capi.svoh_matcher_stage_t())) != 0
    ctx._check(ctx.lib.svoh_matcher_begin_deferred(ctx.h))
    g = capi.svoh_matcher_stage_t()
    ctx._check(ctx.lib.svoh_matcher_stage(ctx.h, 1, 64, 2, 0, C.byref(g)))
    assert not g.px_cur and not g.A_cur_ref   # staged without the match outputs
    fb = capi.svoh_feature_batch()
    fb.n, fb.mem_space, fb.n_cur_frames = 64, capi.SVOH_MEM_STAGED, 1
    for k in ("ref_frame_idx", "cur_frame_idx", "px", "f", "grad", "level", "type"):
        setattr(fb, k, getattr(g, k))
    foreign = np.zeros(4 * 64)
    assert ctx.lib.svoh_update_seeds_batch(ctx.h, C.byref(mopt), C.byref(dopt), 1, refs, curs, C.byref(fb), foreign.ctypes.data, g.success, g.result, None) != 0
    assert ctx.lib.svoh_update_seeds_batch(ctx.h, C.byref(mopt), C.byref(dopt), 3, refs, curs, C.byref(fb), g.state, g.success, g.result, None) != 0   # 3 + 1 views > 2
    ctx._check(ctx.lib.svoh_matcher_collect(ctx.h))
    for sc, fr, fc in pairs:
        ctx.release_frame(fr); ctx.release_frame(fc)


def _view(ptr, dtype, count):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(count * np.dtype(dtype).itemsize,)).view(dtype)


def _rigid_mul(a7, b7):
    """svoh::mul(Rigid, Rigid) of csrc/svoh_math.h, operation by operation (IEEE doubles, no contraction): [qw qx qy qz tx ty tz]."""
    aw, ax, ay, az, atx, aty, atz = [float(v) for v in a7]
    bw, bx, by, bz, btx, bty, btz = [float(v) for v in b7]
    w = aw * bw - ax * bx - ay * by - az * bz
    x = aw * bx + ax * bw + ay * bz - az * by
    y = aw * by + ay * bw + az * bx - ax * bz
    z = aw * bz + az * bw + ax * by - ay * bx
    n2 = w * w + x * x + y * y + z * z
    if abs(n2 - 1.0) > 1.0e-4:
        n = float(np.sqrt(n2))
        w, x, y, z = w / n, x / n, y / n, z / n
    ux = ay * btz - az * bty; uy = az * btx - ax * btz; uz = ax * bty - ay * btx
    ux += ux; uy += uy; uz += uz
    rx = btx + aw * ux + (ay * uz - az * uy)
    ry = bty + aw * uy + (az * ux - ax * uz)
    rz = btz + aw * uz + (ax * uy - ay * ux)
    return np.array([w, x, y, z, atx + rx, aty + ry, atz + rz])


def _rigid_transform(T7, p):
    """svoh::transform(Rigid, Vec3) of csrc/svoh_math.h, operation by operation."""
    qw, qx, qy, qz, tx, ty, tz = [float(v) for v in T7]
    vx, vy, vz = [float(v) for v in p]
    ux = qy * vz - qz * vy; uy = qz * vx - qx * vz; uz = qx * vy - qy * vx
    ux += ux; uy += uy; uz += uz
    rx = vx + qw * ux + (qy * uz - qz * uy)
    ry = vy + qw * uy + (qz * ux - qx * uz)
    rz = vz + qw * uz + (qx * uy - qy * ux)
    return [rx + tx, ry + ty, rz + tz]


def _rigid_inverse(T7):
    """svoh::inverse(Rigid) of csrc/svoh_math.h, operation by operation: conjugate; translation = -(rotation by conj(q) / |q|^2)."""
    qw, qx, qy, qz, tx, ty, tz = [float(v) for v in T7]
    n2 = qw * qw + qx * qx + qy * qy + qz * qz
    iw, ix, iy, iz = qw / n2, -qx / n2, -qy / n2, -qz / n2
    r = _rigid_transform([iw, ix, iy, iz, 0.0, 0.0, 0.0], [tx, ty, tz])
    # (transform adds a zero translation: x + 0.0 is x)
    return [qw, -qx, -qy, -qz, -r[0], -r[1], -r[2]]


def _upload_features(ctx, sets):
    """svoh_features_upload for a list of seed sets (make_seed_set dictionaries) in ONE call; returns the handles."""
    m = len(sets)
    keep = [[np.ascontiguousarray(sd[k], dt) for sd in sets] for k, dt in (("px", np.float64), ("f", np.float64), ("grad", np.float64), ("level", np.int32))]
    n = (C.c_int32 * m)(*[a.size for a in keep[3]])
    ptrs = [(C.c_void_p * m)(*[a.ctypes.data for a in col]) for col in keep]
    out = (C.c_uint64 * m)()
    ctx._check(ctx.lib.svoh_features_upload(ctx.h, m, n, ptrs[0], ptrs[1], ptrs[2], ptrs[3], out))
    return [int(h) for h in out]


def test_batches_that_name_features_by_index_equal_the_batches_with_their_columns(gpu_ctx):
    """svoh_features_upload + SVOH_STAGE_RESIDENT_COLUMNS: the constant columns of three reference frames uploaded once; a staged
    direct batch and a staged seed batch over a shuffled SUBSET of their features, named by (reference frame, index), against the
    same units staged with their px / f / grad / level.  Every output bit for bit; an index outside a set marks its unit NOT_RUN and
    leaves the others alone; the misuse cases fail with an error."""
    ctx = gpu_ctx
    cam = synth.Camera.euroc_like()
    mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(cam)
    pairs = [seed_scene(ctx, 910 + i, cam) for i in range(3)]
    sets = [synth.make_seed_set(sc, 500 + 150 * i, seed=10 + i, margin=12) for i, (sc, fr, fc) in enumerate(pairs)]
    for sd in sets:
        sd["type"][::7] = capi.FT_EDGELET_SEED_CONVERGED
    handles = _upload_features(ctx, sets)
    assert len(set(handles)) == 3 and all(handles)
    rng = np.random.RandomState(5)
    # units: (frame k, feature j), shuffled across the frames, two thirds of all features
    units = [(k, j) for k, sd in enumerate(sets) for j in range(sd["level"].size)]
    rng.shuffle(units)
    units = units[:2 * len(units) // 3]
    n = len(units)
    uk = np.array([u[0] for u in units], np.int32); uj = np.array([u[1] for u in units], np.int32)

    def col(name, width, dtype):
        return np.concatenate([np.asarray(sets[k][name], dtype).reshape(-1, width)[j] for k, j in units]).astype(dtype)
    depth = np.array([sets[k]["true_depth"][j] * 1.02 for k, j in units])
    px0 = np.empty((n, 2))
    for k, (sc, fr, fc) in enumerate(pairs):
        sd = sets[k]
        x = sd["f"].reshape(-1, 3).T * sd["true_depth"]
        pxt = sc.cam.project(sc.T_w_cur.inverse().transform(sc.T_w_ref.transform(x))).T
        sel = uk == k
        px0[sel] = pxt[uj[sel]] + rng.uniform(-1.5, 1.5, (int(sel.sum()), 2))
    state = col("state", 4, np.float64)
    typ = col("type", 1, np.uint8)

    def run(resident, bad_unit=None):
        refs = (capi.svoh_frame_view * 3)(*[fe.make_frame_view(fr, cam, sc.T_ref_f_w, sd["mu_range"], 0) for (sc, fr, fc), sd in zip(pairs, sets)])
        if resident:
            for k in range(3):
                refs[k].features = handles[k]
        curs = (capi.svoh_frame_view * 3)(*[fe.make_frame_view(fc, cam, sc.T_cur_f_w_gt, 0.0, 1) for (sc, fr, fc) in pairs])
        flags = capi.SVOH_STAGE_MATCH_OUTPUTS | (capi.SVOH_STAGE_RESIDENT_COLUMNS if resident else 0)
        ctx._check(ctx.lib.svoh_matcher_begin_deferred(ctx.h))
        ds, ss = capi.svoh_matcher_stage_t(), capi.svoh_matcher_stage_t()
        ctx._check(ctx.lib.svoh_matcher_stage(ctx.h, 0, n, 8, flags, C.byref(ds)))
        ctx._check(ctx.lib.svoh_matcher_stage(ctx.h, 1, n, 8, flags, C.byref(ss)))
        for g in (ds, ss):
            _view(g.ref_frame_idx, np.int32, n)[:] = uk
            _view(g.cur_frame_idx, np.int32, n)[:] = uk
            _view(g.type, np.uint8, n)[:] = typ
            if resident:
                assert not g.px and not g.f and not g.grad and not g.level
                fi = _view(g.feature_index, np.int32, n)
                fi[:] = uj
                if bad_unit is not None:
                    fi[bad_unit] = 10 ** 6
            else:
                assert not g.feature_index
                _view(g.px, np.float64, 2 * n)[:] = col("px", 2, np.float64)
                _view(g.f, np.float64, 3 * n)[:] = col("f", 3, np.float64)
                _view(g.grad, np.float64, 2 * n)[:] = col("grad", 2, np.float64)
                _view(g.level, np.int32, n)[:] = col("level", 1, np.int32)
        _view(ds.depth, np.float64, n)[:] = depth
        _view(ds.px_cur, np.float64, 2 * n)[:] = px0.ravel()
        _view(ss.state, np.float64, 4 * n)[:] = state

        def batch(g):
            fb = capi.svoh_feature_batch()
            fb.n, fb.mem_space, fb.n_cur_frames = n, capi.SVOH_MEM_STAGED, 3
            for k in ("ref_frame_idx", "cur_frame_idx", "px", "f", "grad", "level", "type", "feature_index"):
                setattr(fb, k, getattr(g, k))
            return fb
        fbd, fbs = batch(ds), batch(ss)
        ctx._check(ctx.lib.svoh_match_direct_batch(ctx.h, C.byref(mopt), 3, refs, curs, C.byref(fbd), ds.depth, ds.px_cur, ds.result, ds.f_cur, ds.search_level,
                                                   ds.h_inv, ds.A_cur_ref))
        outs = capi.svoh_seed_match_outputs(ss.px_cur, ss.f_cur, ss.search_level, ss.A_cur_ref)
        ctx._check(ctx.lib.svoh_update_seeds_batch_ex(ctx.h, C.byref(mopt), C.byref(dopt), 3, refs, curs, C.byref(fbs), ss.state, ss.success, ss.result, None, C.byref(outs)))
        ctx._check(ctx.lib.svoh_matcher_collect(ctx.h))
        got = {}
        for name, g in (("d", ds), ("s", ss)):
            got[name] = dict(result=_view(g.result, np.int32, n).copy(), px_cur=_view(g.px_cur, np.float64, 2 * n).copy(), f_cur=_view(g.f_cur, np.float64, 3 * n).copy(),
                             search_level=_view(g.search_level, np.int32, n).copy(), A=_view(g.A_cur_ref, np.float64, 4 * n).copy(), type=_view(g.type, np.uint8, n).copy())
        got["s"]["state"] = _view(ss.state, np.float64, 4 * n).copy(); got["s"]["success"] = _view(ss.success, np.uint8, n).copy()
        return got

    want, got = run(False), run(True)
    assert (want["d"]["result"] == capi.MATCH_SUCCESS).mean() > 0.4 and want["s"]["success"].mean() > 0.3
    for kind in ("d", "s"):
        for name, a in want[kind].items():
            assert np.array_equal(a, got[kind][name]), (kind, name)
    # an index outside its set: that unit is not run, every other unit is what it was
    bad = n // 3
    got_bad = run(True, bad_unit=bad)
    assert got_bad["d"]["result"][bad] == capi.MATCH_NOT_RUN and got_bad["s"]["result"][bad] == capi.MATCH_NOT_RUN and got_bad["s"]["success"][bad] == 0
    others = np.arange(n) != bad
    assert np.array_equal(got_bad["d"]["result"][others], want["d"]["result"][others]) and np.array_equal(got_bad["s"]["state"].reshape(-1, 4)[others], want["s"]["state"].reshape(-1, 4)[others])
    # misuse: a reference frame without columns; an unknown handle; feature_index in a host-array batch
    refs = (capi.svoh_frame_view * 3)(*[fe.make_frame_view(fr, cam, sc.T_ref_f_w, sd["mu_range"], 0) for (sc, fr, fc), sd in zip(pairs, sets)])
    curs = (capi.svoh_frame_view * 3)(*[fe.make_frame_view(fc, cam, sc.T_cur_f_w_gt, 0.0, 1) for (sc, fr, fc) in pairs])
    for which_bad in ("missing", "unknown"):
        ctx._check(ctx.lib.svoh_matcher_begin_deferred(ctx.h))
        g = capi.svoh_matcher_stage_t()
        ctx._check(ctx.lib.svoh_matcher_stage(ctx.h, 1, 16, 8, capi.SVOH_STAGE_RESIDENT_COLUMNS, C.byref(g)))
        for k in range(3):
            refs[k].features = handles[k]
        refs[1].features = 0 if which_bad == "missing" else 987654321
        fb = capi.svoh_feature_batch()
        fb.n, fb.mem_space, fb.n_cur_frames = 16, capi.SVOH_MEM_STAGED, 3
        for k in ("ref_frame_idx", "cur_frame_idx", "type", "feature_index"):
            setattr(fb, k, getattr(g, k))
        assert ctx.lib.svoh_update_seeds_batch(ctx.h, C.byref(mopt), C.byref(dopt), 3, refs, curs, C.byref(fb), g.state, g.success, g.result, None) != 0
        ctx._check(ctx.lib.svoh_matcher_collect(ctx.h))
    sd = sets[0]
    fb, keep = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
    idx = np.zeros(sd["level"].size, np.int32)
    fb.feature_index = idx.ctypes.data
    st = sd["state"].copy(); succ = np.zeros(sd["level"].size, np.uint8)
    assert ctx.lib.svoh_update_seeds_batch(ctx.h, C.byref(mopt), C.byref(dopt), 1, refs, curs, C.byref(fb), st.ctypes.data, succ.ctypes.data, None, None) != 0
    for h in handles:
        ctx._check(ctx.lib.svoh_features_release(ctx.h, h))
    assert ctx.lib.svoh_features_release(ctx.h, handles[0]) != 0   # released already
    for sc, fr, fc in pairs:
        ctx.release_frame(fr); ctx.release_frame(fc)


@pytest.mark.parametrize("case", ["eight lanes per unit", "packed (forced)", "packed (52 000 units)"])
def test_whole_sets_seed_batch_equals_the_batch_of_named_units(gpu_ctx, monkeypatch, case):
    """SVOH_BATCH_WHOLE_SETS (round 6): a staged seed batch whose units ARE the reference frames' resident features, frame after
    frame -- what DepthFilter::updateSeeds' loop is (depth_filter.cpp:200-251) -- against the same batch with every unit named by
    (reference frame, feature index).  Small batches: the prologue fills in what the per-unit kernels read.  Large batches (the
    packed geometry): the kernel walks the tile-ordered copies of the columns that svoh_features_upload made once per keyframe
    -- no counting sort, no record scatter, no un-sort per frame.  Every output bit for bit; the unit arrays the caller no longer
    fills in are poisoned here to show they are not read; sets that do not add up to n are refused."""
    ctx = gpu_ctx
    cam = synth.Camera.euroc_like()
    mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(cam)
    big = case == "packed (52 000 units)"
    if case == "packed (forced)":
        monkeypatch.setenv("SVOH_MATCHER_G8", "2")
        ctx.reload_knobs()
    pairs = [seed_scene(ctx, 930 + i, cam) for i in range(3)]
    n_sets, per = (20, 2600) if big else (5, 400)
    sets = [synth.make_seed_set(pairs[k % 3][0], per + 37 * (k % 4), seed=40 + k, margin=12) for k in range(n_sets)]
    for sd in sets:
        sd["type"][::9] = capi.FT_EDGELET_SEED_CONVERGED
    handles = _upload_features(ctx, sets)
    n = sum(sd["level"].size for sd in sets)
    assert (n > 49152) == big
    uk = np.concatenate([np.full(sd["level"].size, k, np.int32) for k, sd in enumerate(sets)])
    uj = np.concatenate([np.arange(sd["level"].size, dtype=np.int32) for sd in sets])
    ck = (uk % 3).astype(np.int32)                      # every set into the current frame of its own scene
    state = np.concatenate([sd["state"] for sd in sets]); typ = np.concatenate([sd["type"] for sd in sets])

    def run(layout):
        refs = (capi.svoh_frame_view * n_sets)(*[fe.make_frame_view(pairs[k % 3][1], cam, pairs[k % 3][0].T_ref_f_w, sd["mu_range"], 0) for k, sd in enumerate(sets)])
        for k in range(n_sets):
            refs[k].features = handles[k]
        curs = (capi.svoh_frame_view * 3)(*[fe.make_frame_view(fc, cam, sc.T_cur_f_w_gt, 0.0, 1) for (sc, fr, fc) in pairs])
        ctx._check(ctx.lib.svoh_matcher_begin_deferred(ctx.h))
        g = capi.svoh_matcher_stage_t()
        ctx._check(ctx.lib.svoh_matcher_stage(ctx.h, 1, n, n_sets + 4, capi.SVOH_STAGE_MATCH_OUTPUTS | capi.SVOH_STAGE_RESIDENT_COLUMNS, C.byref(g)))
        _view(g.ref_frame_idx, np.int32, n)[:] = uk if layout == capi.SVOH_BATCH_UNITS else -7
        _view(g.feature_index, np.int32, n)[:] = uj if layout == capi.SVOH_BATCH_UNITS else 10 ** 7
        _view(g.cur_frame_idx, np.int32, n)[:] = ck
        _view(g.type, np.uint8, n)[:] = typ
        _view(g.state, np.float64, 4 * n)[:] = state
        fb = capi.svoh_feature_batch()
        fb.n, fb.mem_space, fb.n_cur_frames, fb.layout = n, capi.SVOH_MEM_STAGED, 3, layout
        for k in ("ref_frame_idx", "cur_frame_idx", "type", "feature_index"):
            setattr(fb, k, getattr(g, k))
        outs = capi.svoh_seed_match_outputs(g.px_cur, g.f_cur, g.search_level, g.A_cur_ref)
        ctx._check(ctx.lib.svoh_update_seeds_batch_ex(ctx.h, C.byref(mopt), C.byref(dopt), n_sets, refs, curs, C.byref(fb), g.state, g.success, g.result, None, C.byref(outs)))
        ctx._check(ctx.lib.svoh_matcher_collect(ctx.h))
        return dict(result=_view(g.result, np.int32, n).copy(), px_cur=_view(g.px_cur, np.float64, 2 * n).copy(), f_cur=_view(g.f_cur, np.float64, 3 * n).copy(),
                    search_level=_view(g.search_level, np.int32, n).copy(), A=_view(g.A_cur_ref, np.float64, 4 * n).copy(), type=_view(g.type, np.uint8, n).copy(),
                    state=_view(g.state, np.float64, 4 * n).copy(), success=_view(g.success, np.uint8, n).copy())

    want, got = run(capi.SVOH_BATCH_UNITS), run(capi.SVOH_BATCH_WHOLE_SETS)
    assert want["success"].mean() > 0.3 and len(set(want["result"])) >= 3
    for name, a in want.items():
        assert np.array_equal(a, got[name]), (case, name, int((a != got[name]).sum()))
    # sets that do not add up to n: refused (the section stays usable)
    refs = (capi.svoh_frame_view * n_sets)(*[fe.make_frame_view(pairs[k % 3][1], cam, pairs[k % 3][0].T_ref_f_w, sd["mu_range"], 0) for k, sd in enumerate(sets)])
    for k in range(n_sets):
        refs[k].features = handles[k]
    curs = (capi.svoh_frame_view * 3)(*[fe.make_frame_view(fc, cam, sc.T_cur_f_w_gt, 0.0, 1) for (sc, fr, fc) in pairs])
    ctx._check(ctx.lib.svoh_matcher_begin_deferred(ctx.h))
    g = capi.svoh_matcher_stage_t()
    ctx._check(ctx.lib.svoh_matcher_stage(ctx.h, 1, n - 5, n_sets + 4, capi.SVOH_STAGE_RESIDENT_COLUMNS, C.byref(g)))
    fb = capi.svoh_feature_batch()
    fb.n, fb.mem_space, fb.n_cur_frames, fb.layout = n - 5, capi.SVOH_MEM_STAGED, 3, capi.SVOH_BATCH_WHOLE_SETS
    for k in ("ref_frame_idx", "cur_frame_idx", "type", "feature_index"):
        setattr(fb, k, getattr(g, k))
    assert ctx.lib.svoh_update_seeds_batch(ctx.h, C.byref(mopt), C.byref(dopt), n_sets, refs, curs, C.byref(fb), g.state, g.success, g.result, None) != 0
    ctx._check(ctx.lib.svoh_matcher_collect(ctx.h))
    # ... and the layout is refused where it has no meaning: a direct batch, a host-array batch
    sd = sets[0]
    fbh, keep = fe.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
    fbh.layout = capi.SVOH_BATCH_WHOLE_SETS
    st = sd["state"].copy(); succ = np.zeros(sd["level"].size, np.uint8)
    assert ctx.lib.svoh_update_seeds_batch(ctx.h, C.byref(mopt), C.byref(dopt), 1, refs, curs, C.byref(fbh), st.ctypes.data, succ.ctypes.data, None, None) != 0
    for h in handles:
        ctx._check(ctx.lib.svoh_features_release(ctx.h, h))
    for sc, fr, fc in pairs:
        ctx.release_frame(fr); ctx.release_frame(fc)


def test_seed_batch_behind_the_pose_kernel_takes_its_poses_on_the_device(gpu_ctx):
    """svoh_frame_view::pose_result_index_plus1: a staged seed batch over three current frames, queued and SENT OFF from the hook of
    svoh_optimize_pose_batch_hook -- two of the frames take their pose from the pose batch in flight (T_cam_imu x the optimised
    T_imu_world, composed on the device), one keeps the pose it was given -- against the same batch queued after the host has
    applied the poses.  States, types, success flags, result codes bit for bit.  Outside the hook the field is refused."""
    import pose_helpers as ph
    ctx = gpu_ctx
    cam = synth.Camera.euroc_like()
    mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(cam)
    pairs = [seed_scene(ctx, 940 + i, cam) for i in range(3)]
    sets = [synth.make_seed_set(sc, 400 + 100 * i, seed=20 + i, margin=12) for i, (sc, fr, fc) in enumerate(pairs)]
    n_each = [sd["level"].size for sd in sets]
    n = sum(n_each)
    # two pose problems whose optimum nobody knows in advance (the batch must really read the device's result): the current
    # frames 0 and 2 are cameras of rigs 1 and 0
    pose_of_cur = {0: 1, 2: 0}   # current frame -> pose problem
    cur_of_pose = {p: k for k, p in pose_of_cur.items()}
    scenes = [ph.make_pose_scene(300 + i, n=160, cam=cam, T_cam0_world=pairs[cur_of_pose[i]][0].T_cur_f_w_gt) for i in range(2)]
    popt = capi.default_pose_options(cam)
    pbs, keeps = zip(*[fe.make_pose_problem(sc["cams"], sc["T_imu_world_init"]) for sc in scenes])
    res_host = ctx.optimize_pose(popt, list(pbs))
    T_cam_imu = {k: scenes[p]["cams"][0]["T_cam_imu"] for k, p in pose_of_cur.items()}

    def fill(g):
        off = 0
        for k, sd in enumerate(sets):
            m = n_each[k]
            _view(g.ref_frame_idx, np.int32, n)[off:off + m] = k
            _view(g.cur_frame_idx, np.int32, n)[off:off + m] = k
            _view(g.px, np.float64, 2 * n)[2 * off:2 * (off + m)] = sd["px"]
            _view(g.f, np.float64, 3 * n)[3 * off:3 * (off + m)] = sd["f"]
            _view(g.grad, np.float64, 2 * n)[2 * off:2 * (off + m)] = sd["grad"]
            _view(g.level, np.int32, n)[off:off + m] = sd["level"]
            _view(g.type, np.uint8, n)[off:off + m] = sd["type"]
            _view(g.state, np.float64, 4 * n)[4 * off:4 * (off + m)] = sd["state"]
            off += m

    def batch(g):
        fb = capi.svoh_feature_batch()
        fb.n, fb.mem_space, fb.n_cur_frames = n, capi.SVOH_MEM_STAGED, 3
        for k in ("ref_frame_idx", "cur_frame_idx", "px", "f", "grad", "level", "type"):
            setattr(fb, k, getattr(g, k))
        return fb
    refs = (capi.svoh_frame_view * 3)(*[fe.make_frame_view(fr, cam, sc.T_ref_f_w, sd["mu_range"], 0) for (sc, fr, fc), sd in zip(pairs, sets)])

    def outputs(g):
        return dict(state=_view(g.state, np.float64, 4 * n).copy(), type=_view(g.type, np.uint8, n).copy(), success=_view(g.success, np.uint8, n).copy(),
                    result=_view(g.result, np.int32, n).copy())
    # (a) the poses applied by the host: T_f_w = T_cam_imu * T_imu_world(result)
    curs = (capi.svoh_frame_view * 3)(*[fe.make_frame_view(fc, cam, sc.T_cur_f_w_gt, 0.0, 1) for (sc, fr, fc) in pairs])
    for k, p in pose_of_cur.items():
        curs[k].T_f_w = fe._se3(_rigid_mul(T_cam_imu[k].as7(), fe.se3_to_numpy(res_host[p].T_imu_world)))
    ctx._check(ctx.lib.svoh_matcher_begin_deferred(ctx.h))
    g = capi.svoh_matcher_stage_t()
    ctx._check(ctx.lib.svoh_matcher_stage(ctx.h, 1, n, 8, 0, C.byref(g)))
    fill(g)
    fb = batch(g)
    ctx._check(ctx.lib.svoh_update_seeds_batch(ctx.h, C.byref(mopt), C.byref(dopt), 3, refs, curs, C.byref(fb), g.state, g.success, g.result, None))
    ctx._check(ctx.lib.svoh_matcher_collect(ctx.h))
    want = outputs(g)
    assert want["success"].mean() > 0.2
    # (b) queued from the pose call's hook, poses taken on the device
    curs_dev = (capi.svoh_frame_view * 3)(*[fe.make_frame_view(fc, cam, sc.T_cur_f_w_gt, 0.0, 1) for (sc, fr, fc) in pairs])
    for k, p in pose_of_cur.items():
        curs_dev[k].T_f_w = fe._se3(T_cam_imu[k])
        curs_dev[k].pose_result_index_plus1 = p + 1
    ctx._check(ctx.lib.svoh_matcher_begin_deferred(ctx.h))
    g2 = capi.svoh_matcher_stage_t()
    ctx._check(ctx.lib.svoh_matcher_stage(ctx.h, 1, n, 8, 0, C.byref(g2)))
    fill(g2)
    fb2 = batch(g2)
    # outside the hook: refused, the staged block stays usable
    assert ctx.lib.svoh_update_seeds_batch(ctx.h, C.byref(mopt), C.byref(dopt), 3, refs, curs_dev, C.byref(fb2), g2.state, g2.success, g2.result, None) != 0
    hook_rc = []

    @C.CFUNCTYPE(None, C.c_void_p)
    def hook(_user):
        hook_rc.append(ctx.lib.svoh_update_seeds_batch(ctx.h, C.byref(mopt), C.byref(dopt), 3, refs, curs_dev, C.byref(fb2), g2.state, g2.success, g2.result, None))
        hook_rc.append(ctx.lib.svoh_matcher_flush(ctx.h))
    arr = (capi.svoh_pose_problem * 2)(*pbs)
    res = (capi.svoh_pose_result * 2)()
    ctx.lib.svoh_optimize_pose_batch_hook.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    ctx._check(ctx.lib.svoh_optimize_pose_batch_hook(ctx.h, C.byref(popt), 2, arr, res, hook, None))
    assert hook_rc == [0, 0], hook_rc
    ctx._check(ctx.lib.svoh_matcher_collect(ctx.h))
    for p in range(2):
        assert np.array_equal(fe.se3_to_numpy(res[p].T_imu_world), fe.se3_to_numpy(res_host[p].T_imu_world))
    got = outputs(g2)
    for name in want:
        assert np.array_equal(want[name], got[name]), name
    # an index beyond the batch in flight: refused inside the hook as well
    curs_dev[0].pose_result_index_plus1 = 3
    ctx._check(ctx.lib.svoh_matcher_begin_deferred(ctx.h))
    g3 = capi.svoh_matcher_stage_t()
    ctx._check(ctx.lib.svoh_matcher_stage(ctx.h, 1, n, 8, 0, C.byref(g3)))
    fill(g3)
    fb3 = batch(g3)
    rc3 = []

    @C.CFUNCTYPE(None, C.c_void_p)
    def hook3(_user):
        rc3.append(ctx.lib.svoh_update_seeds_batch(ctx.h, C.byref(mopt), C.byref(dopt), 3, refs, curs_dev, C.byref(fb3), g3.state, g3.success, g3.result, None))
    ctx._check(ctx.lib.svoh_optimize_pose_batch_hook(ctx.h, C.byref(popt), 2, arr, res, hook3, None))
    assert rc3 and rc3[0] != 0
    ctx._check(ctx.lib.svoh_matcher_collect(ctx.h))
    for sc, fr, fc in pairs:
        ctx.release_frame(fr); ctx.release_frame(fc)


def test_candidate_projections_of_many_frames_equal_the_single_call(gpu_ctx):
    """svoh_project_candidates_stage / _enqueue_staged / _wait with four jobs (own camera pose, own keyframe table, one of
    them without points of kind 1) against svoh_project_candidates job by job: pixels and verdicts equal."""
    ctx = gpu_ctx
    cam = synth.Camera.euroc_like(752, 480)
    rng = np.random.RandomState(11)
    jobs = []
    for j in range(4):
        n, n_kf = (900, 1500, 40, 2500)[j], (3, 5, 1, 4)[j]
        def rand_T(s):
            q = np.array([1.0, 0, 0, 0]) + rng.normal(0, s, 4)
            return synth.SE3(q / np.linalg.norm(q), rng.normal(0, s, 3))
        T_f_w = rand_T(0.03)
        T_kf = [rand_T(0.1) for _ in range(n_kf)]
        kind = (rng.uniform(size=n) < (0.6 if j != 2 else 0.0)).astype(np.uint8)
        kf = rng.randint(0, n_kf, n).astype(np.int32)
        kf[5::97] = n_kf + 3   # a bad keyframe index: not visible, not a fault
        v = rng.normal(0, 1, (n, 3)); v[:, 2] = np.abs(v[:, 2]) + 1.0
        v[kind == 1] /= np.linalg.norm(v[kind == 1], axis=1, keepdims=True)
        mu = rng.uniform(0.1, 1.0, n)
        jobs.append(dict(T=T_f_w, T_kf=T_kf, kind=kind, kf=kf, v=np.ascontiguousarray(v.ravel()), mu=mu, n=n, n_kf=n_kf))
    want = []
    for jb in jobs:
        Tk = (capi.svoh_se3 * jb["n_kf"])(*[fe._se3(T.inverse()) for T in jb["T_kf"]])
        px = np.zeros(2 * jb["n"]); vis = np.zeros(jb["n"], np.uint8)
        c, T = fe._camera(cam), fe._se3(jb["T"])
        ctx._check(ctx.lib.svoh_project_candidates(ctx.h, C.byref(c), C.byref(T), jb["n_kf"], Tk, jb["n"], jb["kind"].ctypes.data, jb["kf"].ctypes.data,
                                                   jb["v"].ctypes.data, jb["mu"].ctypes.data, px.ctypes.data, vis.ctypes.data))
        want.append((px, vis))
    n_total, kf_total = sum(j["n"] for j in jobs), sum(j["n_kf"] for j in jobs)
    cs = capi.svoh_candidate_stage_t()
    ctx._check(ctx.lib.svoh_project_candidates_stage(ctx.h, len(jobs), kf_total, n_total, C.byref(cs)))

    def view(ptr, dtype, count):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(count * np.dtype(dtype).itemsize,)).view(dtype)
    jarr = C.cast(cs.jobs, C.POINTER(capi.svoh_candidate_job))
    karr = C.cast(cs.T_world_kf, C.POINTER(capi.svoh_se3))
    off = koff = 0
    for j, jb in enumerate(jobs):
        jj = capi.svoh_candidate_job()
        jj.cam, jj.T_f_w_or_T_cam_imu, jj.align_result_index = fe._camera(cam), fe._se3(jb["T"]), -1
        jj.kf_begin, jj.n_kf, jj.point_begin, jj.n_points = koff, jb["n_kf"], off, jb["n"]
        jarr[j] = jj
        for k, T in enumerate(jb["T_kf"]):
            karr[koff + k] = fe._se3(T.inverse())
        m = jb["n"]
        view(cs.job, np.int32, n_total)[off:off + m] = j
        view(cs.kind, np.uint8, n_total)[off:off + m] = jb["kind"]
        view(cs.kf, np.int32, n_total)[off:off + m] = jb["kf"]
        view(cs.v, np.float64, 3 * n_total)[3 * off:3 * (off + m)] = jb["v"]
        view(cs.mu, np.float64, n_total)[off:off + m] = jb["mu"]
        off += m; koff += jb["n_kf"]
    assert ctx.lib.svoh_project_candidates_wait(ctx.h) != 0   # nothing in flight yet
    ctx._check(ctx.lib.svoh_project_candidates_enqueue_staged(ctx.h))
    assert ctx.lib.svoh_project_candidates_stage(ctx.h, 1, 1, 1, C.byref(capi.svoh_candidate_stage_t())) != 0   # in flight: wait first
    ctx._check(ctx.lib.svoh_project_candidates_wait(ctx.h))
    off = 0
    for j, jb in enumerate(jobs):
        m = jb["n"]
        assert np.array_equal(view(cs.px, np.float64, 2 * n_total)[2 * off:2 * (off + m)], want[j][0])
        assert np.array_equal(view(cs.visible, np.uint8, n_total)[off:off + m], want[j][1])
        assert 0.02 < want[j][1].mean() < 0.98
        off += m
    # a job whose ranges leave the staged arrays is refused before anything is launched
    ctx._check(ctx.lib.svoh_project_candidates_stage(ctx.h, 1, 1, 10, C.byref(cs)))
    jj = capi.svoh_candidate_job()
    jj.cam, jj.align_result_index, jj.n_kf, jj.point_begin, jj.n_points = fe._camera(cam), -1, 1, 5, 10
    C.cast(cs.jobs, C.POINTER(capi.svoh_candidate_job))[0] = jj
    assert ctx.lib.svoh_project_candidates_enqueue_staged(ctx.h) != 0


@pytest.mark.parametrize("with_landmarks", [False, True])
def test_candidate_projection_over_resident_columns_equals_the_per_point_form(gpu_ctx, with_landmarks):
    """svoh_project_candidates_stage_ranges: three jobs, each over the features of a few keyframes whose columns are resident
    (svoh_features_upload) -- per point only kind and mu are staged, a seed's bearing vector is its keyframe's f column -- against
    svoh_project_candidates with the same points written out.  With landmarks among the points their positions (v) go up too."""
    ctx = gpu_ctx
    cam = synth.Camera.euroc_like(752, 480)
    rng = np.random.RandomState(23)

    def rand_T(s):
        q = np.array([1.0, 0, 0, 0]) + rng.normal(0, s, 4)
        return synth.SE3(q / np.linalg.norm(q), rng.normal(0, s, 3))
    jobs = []
    for j in range(3):
        kfs = []
        for k in range((2, 4, 1)[j]):
            n = int(rng.randint(150, 700))
            f = rng.normal(0, 0.4, (n, 3)); f[:, 2] = 1.0
            f /= np.linalg.norm(f, axis=1, keepdims=True)
            kind = np.ones(n, np.uint8)
            v = f.copy()
            if with_landmarks:
                lm = rng.uniform(size=n) < 0.3
                kind[lm] = 0
                v[lm] = rng.normal(0, 1, (int(lm.sum()), 3)) + [0, 0, 4.0]
            kfs.append(dict(T=rand_T(0.1), n=n, f=np.ascontiguousarray(f.ravel()), kind=kind, v=np.ascontiguousarray(v.ravel()), mu=rng.uniform(0.1, 1.0, n),
                            px=rng.uniform(0, 700, 2 * n), grad=rng.normal(size=2 * n), level=rng.randint(0, 3, n).astype(np.int32)))
        jobs.append(dict(T=rand_T(0.03), kfs=kfs))
    all_kfs = [kf for jb in jobs for kf in jb["kfs"]]
    handles = _upload_features(ctx, all_kfs)
    # per job through the blocking single-job call, points written out
    want = []
    for jb in jobs:
        n_kf = len(jb["kfs"])
        Tk = (capi.svoh_se3 * n_kf)(*[fe._se3(kf["T"].inverse()) for kf in jb["kfs"]])
        kind = np.concatenate([kf["kind"] for kf in jb["kfs"]]); v = np.concatenate([kf["v"] for kf in jb["kfs"]]); mu = np.concatenate([kf["mu"] for kf in jb["kfs"]])
        kfi = np.concatenate([np.full(kf["n"], k, np.int32) for k, kf in enumerate(jb["kfs"])])
        n = kind.size
        px = np.zeros(2 * n); vis = np.zeros(n, np.uint8)
        c, T = fe._camera(cam), fe._se3(jb["T"])
        ctx._check(ctx.lib.svoh_project_candidates(ctx.h, C.byref(c), C.byref(T), n_kf, Tk, n, kind.ctypes.data, kfi.ctypes.data, v.ctypes.data, mu.ctypes.data,
                                                   px.ctypes.data, vis.ctypes.data))
        want.append((px, vis))
        assert 0.02 < vis.mean() < 0.98
    n_total, kf_total = sum(kf["n"] for kf in all_kfs), len(all_kfs)
    cs = capi.svoh_candidate_stage_t()
    ctx._check(ctx.lib.svoh_project_candidates_stage_ranges(ctx.h, len(jobs), kf_total, n_total, C.byref(cs)))
    assert not cs.job and not cs.kf and cs.ranges
    jarr = C.cast(cs.jobs, C.POINTER(capi.svoh_candidate_job))
    karr = C.cast(cs.T_world_kf, C.POINTER(capi.svoh_se3))
    rarr = C.cast(cs.ranges, C.POINTER(capi.svoh_candidate_range))
    _view(cs.v, np.float64, 3 * n_total)[:] = np.nan   # what is not a landmark's must not be read
    off = koff = 0
    for j, jb in enumerate(jobs):
        jj = capi.svoh_candidate_job()
        jj.cam, jj.T_f_w_or_T_cam_imu, jj.align_result_index = fe._camera(cam), fe._se3(jb["T"]), -1
        jj.kf_begin, jj.n_kf, jj.point_begin, jj.n_points = koff, len(jb["kfs"]), off, sum(kf["n"] for kf in jb["kfs"])
        jarr[j] = jj
        for k, kf in enumerate(jb["kfs"]):
            karr[koff + k] = fe._se3(kf["T"].inverse())
            r = capi.svoh_candidate_range()
            r.features, r.point_begin, r.n_points, r.job = handles[koff + k], off, kf["n"], j
            rarr[koff + k] = r
            m = kf["n"]
            _view(cs.kind, np.uint8, n_total)[off:off + m] = kf["kind"]
            _view(cs.mu, np.float64, n_total)[off:off + m] = kf["mu"]
            lm = kf["kind"] == 0
            if lm.any():
                _view(cs.v, np.float64, 3 * n_total).reshape(-1, 3)[off:off + m][lm] = kf["v"].reshape(-1, 3)[lm]
            off += m
        koff += len(jb["kfs"])
    ctx._check(ctx.lib.svoh_project_candidates_enqueue_staged(ctx.h))
    ctx._check(ctx.lib.svoh_project_candidates_wait(ctx.h))
    off = 0
    for j, jb in enumerate(jobs):
        m = sum(kf["n"] for kf in jb["kfs"])
        assert np.array_equal(_view(cs.px, np.float64, 2 * n_total)[2 * off:2 * (off + m)], want[j][0])
        assert np.array_equal(_view(cs.visible, np.uint8, n_total)[off:off + m], want[j][1])
        off += m
    # ranges that do not lie back to back, an unknown handle: refused before anything is launched
    for bad in ("gap", "handle"):
        ctx._check(ctx.lib.svoh_project_candidates_stage_ranges(ctx.h, 1, 2, 20, C.byref(cs)))
        jj = capi.svoh_candidate_job()
        jj.cam, jj.align_result_index, jj.n_kf, jj.point_begin, jj.n_points = fe._camera(cam), -1, 2, 0, 20
        C.cast(cs.jobs, C.POINTER(capi.svoh_candidate_job))[0] = jj
        rarr = C.cast(cs.ranges, C.POINTER(capi.svoh_candidate_range))
        for k in range(2):
            r = capi.svoh_candidate_range()
            r.features, r.point_begin, r.n_points, r.job = handles[k], 10 * k + (1 if bad == "gap" and k == 1 else 0), 10, 0
            if bad == "handle" and k == 1:
                r.features = 123456789
            rarr[k] = r
        _view(cs.kind, np.uint8, 20)[:] = 1
        assert ctx.lib.svoh_project_candidates_enqueue_staged(ctx.h) != 0
    for h in handles:
        ctx._check(ctx.lib.svoh_features_release(ctx.h, h))


def test_alignment_points_taken_from_the_seed_batch_in_flight(gpu_ctx):
    """svoh_align_camera::pos_seed_unit: two alignment problems whose features are seeds of their reference frames.  A staged
    depth-filter batch over those seeds is sent off and NOT collected; the alignment queued behind it names its points by unit --
    the device computes T_world_keyframe x (f / mu) with the inverse depth the update leaves.  Against the same alignment given the
    positions computed here from the collected states: every field of the result, bit for bit.  A second batch staged in between
    (the block laid out anew) makes the entry refuse the field."""
    import copy
    ctx = gpu_ctx
    cam = synth.Camera.euroc_like()
    mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(cam)
    pairs = [seed_scene(ctx, 960 + i, cam) for i in range(2)]
    sets = [synth.make_seed_set(sc, 220 + 60 * i, seed=30 + i, margin=14) for i, (sc, fr, fc) in enumerate(pairs)]
    n_each = [sd["level"].size for sd in sets]
    n = sum(n_each)
    refs = (capi.svoh_frame_view * 2)(*[fe.make_frame_view(fr, cam, sc.T_ref_f_w, sd["mu_range"], 0) for (sc, fr, fc), sd in zip(pairs, sets)])
    curs = (capi.svoh_frame_view * 2)(*[fe.make_frame_view(fc, cam, sc.T_cur_f_w_gt, 0.0, 1) for (sc, fr, fc) in pairs])

    def stage_and_send():
        ctx._check(ctx.lib.svoh_matcher_begin_deferred(ctx.h))
        g = capi.svoh_matcher_stage_t()
        ctx._check(ctx.lib.svoh_matcher_stage(ctx.h, 1, n, 8, 0, C.byref(g)))
        off = 0
        for k, sd in enumerate(sets):
            m = n_each[k]
            _view(g.ref_frame_idx, np.int32, n)[off:off + m] = k
            _view(g.cur_frame_idx, np.int32, n)[off:off + m] = k
            _view(g.px, np.float64, 2 * n)[2 * off:2 * (off + m)] = sd["px"]
            _view(g.f, np.float64, 3 * n)[3 * off:3 * (off + m)] = sd["f"]
            _view(g.grad, np.float64, 2 * n)[2 * off:2 * (off + m)] = sd["grad"]
            _view(g.level, np.int32, n)[off:off + m] = sd["level"]
            _view(g.type, np.uint8, n)[off:off + m] = sd["type"]
            _view(g.state, np.float64, 4 * n)[4 * off:4 * (off + m)] = sd["state"]
            off += m
        fb = capi.svoh_feature_batch()
        fb.n, fb.mem_space, fb.n_cur_frames = n, capi.SVOH_MEM_STAGED, 2
        for k in ("ref_frame_idx", "cur_frame_idx", "px", "f", "grad", "level", "type"):
            setattr(fb, k, getattr(g, k))
        ctx._check(ctx.lib.svoh_update_seeds_batch(ctx.h, C.byref(mopt), C.byref(dopt), 2, refs, curs, C.byref(fb), g.state, g.success, g.result, None))
        ctx._check(ctx.lib.svoh_matcher_flush(ctx.h))
        return g

    def align_scene(k, pos_world):
        sc = copy.copy(pairs[k][0])
        sd = sets[k]
        sc.n_features = n_each[k]
        sc.px, sc.f, sc.pos_world = sd["px"], sd["f"], pos_world
        sc.flags = np.ones(n_each[k], np.uint8)
        return sc
    opt = capi.default_align_options(max_level=4, min_level=2)
    # (a) the update sent off, the alignment queued behind it with its points named by unit, garbage where the positions would be
    g = stage_and_send()
    units = [np.arange(sum(n_each[:k]), sum(n_each[:k + 1]), dtype=np.int32) for k in range(2)]
    units[1][::9] = -1                                   # some points keep the host's position ...
    pos_dummy = [np.full(3 * m, 1.0e30) for m in n_each]
    problems, keep = fe.make_align_problems([[(align_scene(k, pos_dummy[k]), pairs[k][1], pairs[k][2])] for k in range(2)])
    ctx._check(ctx.lib.svoh_matcher_collect(ctx.h))      # (needed here to know the positions of the kept points; the block stands)
    state_after = _view(g.state, np.float64, 4 * n).copy().reshape(-1, 4)
    pos_host = []
    for k, (sc, fr, fc) in enumerate(pairs):
        lo = sum(n_each[:k])
        mu = state_after[lo:lo + n_each[k], 0]
        T_w_kf = _rigid_inverse(fe.se3_to_numpy(refs[k].T_f_w))
        f = sets[k]["f"].reshape(-1, 3)
        p = np.empty((n_each[k], 3))
        for i in range(n_each[k]):
            depth = 1.0 / float(mu[i])
            p[i] = _rigid_transform(T_w_kf, [float(f[i, 0]) * depth, float(f[i, 1]) * depth, float(f[i, 2]) * depth])
        pos_host.append(np.ascontiguousarray(p.ravel()))
    kept = units[1] < 0
    pos_dummy[1].reshape(-1, 3)[kept] = pos_host[1].reshape(-1, 3)[kept]   # ... which must then be right on the host side
    problems, keep = fe.make_align_problems([[(align_scene(k, pos_dummy[k]), pairs[k][1], pairs[k][2])] for k in range(2)])
    for k in range(2):
        problems[k].cams[0].pos_seed_unit = units[k].ctypes.data
    got = ctx.sparse_align(opt, problems)
    # (b) the positions computed here
    problems_b, keep_b = fe.make_align_problems([[(align_scene(k, pos_host[k]), pairs[k][1], pairs[k][2])] for k in range(2)])
    want = ctx.sparse_align(opt, problems_b)
    for k in range(2):
        assert result_bits(got[k]) == result_bits(want[k]), k
        assert want[k].status == 0 and want[k].n_fts_to_track > 50
    # the update still in flight (not collected): the same again, the alignment queued right behind the flush
    g = stage_and_send()
    got2 = ctx.sparse_align(opt, problems)
    ctx._check(ctx.lib.svoh_matcher_collect(ctx.h))
    assert np.array_equal(_view(g.state, np.float64, 4 * n).reshape(-1, 4), state_after)
    for k in range(2):
        assert result_bits(got2[k]) == result_bits(want[k]), k
    # the seed block staged again since: refused before anything is queued
    ctx._check(ctx.lib.svoh_matcher_begin_deferred(ctx.h))
    ctx._check(ctx.lib.svoh_matcher_stage(ctx.h, 1, 16, 4, 0, C.byref(capi.svoh_matcher_stage_t())))
    res = (capi.svoh_align_result * 2)()
    assert ctx.lib.svoh_sparse_align_batch(ctx.h, C.byref(opt), 2, problems, res) != 0
    ctx._check(ctx.lib.svoh_matcher_collect(ctx.h))
    for sc, fr, fc in pairs:
        ctx.release_frame(fr); ctx.release_frame(fc)


def test_candidate_projection_reads_inverse_depths_from_the_seed_batch_in_flight(gpu_ctx):
    """svoh_candidate_stage_t::mu_unit (ranges form): the points of two keyframes are the seeds of a depth-filter batch that has
    been sent off; the projection queued behind it names each seed's unit instead of carrying its inverse depth.  Against the
    projection given the inverse depths the update left: pixels and verdicts bit for bit.  Without a seed batch on the context
    the entry refuses."""
    ctx = gpu_ctx
    cam = synth.Camera.euroc_like()
    mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(cam)
    pairs = [seed_scene(ctx, 980 + i, cam) for i in range(2)]
    sets = [synth.make_seed_set(sc, 300 + 80 * i, seed=40 + i, margin=14) for i, (sc, fr, fc) in enumerate(pairs)]
    n_each = [sd["level"].size for sd in sets]
    n = sum(n_each)
    handles = _upload_features(ctx, sets)
    refs = (capi.svoh_frame_view * 2)(*[fe.make_frame_view(fr, cam, sc.T_ref_f_w, sd["mu_range"], 0) for (sc, fr, fc), sd in zip(pairs, sets)])
    curs = (capi.svoh_frame_view * 2)(*[fe.make_frame_view(fc, cam, sc.T_cur_f_w_gt, 0.0, 1) for (sc, fr, fc) in pairs])

    def project(mu_per_set, units_per_set, with_units):
        cs = capi.svoh_candidate_stage_t()
        ctx._check(ctx.lib.svoh_project_candidates_stage_ranges(ctx.h, 1, 2, n, C.byref(cs)))
        jj = capi.svoh_candidate_job()
        jj.cam, jj.T_f_w_or_T_cam_imu, jj.align_result_index = fe._camera(cam), fe._se3(pairs[0][0].T_cur_f_w_gt), -1
        jj.kf_begin, jj.n_kf, jj.point_begin, jj.n_points = 0, 2, 0, n
        C.cast(cs.jobs, C.POINTER(capi.svoh_candidate_job))[0] = jj
        karr = C.cast(cs.T_world_kf, C.POINTER(capi.svoh_se3))
        rarr = C.cast(cs.ranges, C.POINTER(capi.svoh_candidate_range))
        assert np.all(_view(cs.mu_unit, np.int32, n) == -1)   # preset
        off = 0
        for k, (sc, fr, fc) in enumerate(pairs):
            karr[k] = fe._se3(_rigid_inverse(fe.se3_to_numpy(refs[k].T_f_w)))
            r = capi.svoh_candidate_range()
            r.features, r.point_begin, r.n_points, r.job = handles[k], off, n_each[k], 0
            rarr[k] = r
            _view(cs.kind, np.uint8, n)[off:off + n_each[k]] = 1
            _view(cs.mu, np.float64, n)[off:off + n_each[k]] = mu_per_set[k]
            if units_per_set is not None:
                _view(cs.mu_unit, np.int32, n)[off:off + n_each[k]] = units_per_set[k]
            off += n_each[k]
        rc = (ctx.lib.svoh_project_candidates_enqueue_staged_units if with_units else ctx.lib.svoh_project_candidates_enqueue_staged)(ctx.h)
        if rc != 0:
            return rc, None, None
        ctx._check(ctx.lib.svoh_project_candidates_wait(ctx.h))
        return 0, _view(cs.px, np.float64, 2 * n).copy(), _view(cs.visible, np.uint8, n).copy()
    # no seed batch has been sent off on this context yet (a fresh one): refused
    fresh = fe.Context(0)
    try:
        assert fresh.lib.svoh_project_candidates_stage_ranges(fresh.h, 1, 1, 4, C.byref(capi.svoh_candidate_stage_t())) == 0
        assert fresh.lib.svoh_project_candidates_enqueue_staged_units(fresh.h) != 0
    finally:
        fresh.close()
    # the update, sent off
    ctx._check(ctx.lib.svoh_matcher_begin_deferred(ctx.h))
    g = capi.svoh_matcher_stage_t()
    ctx._check(ctx.lib.svoh_matcher_stage(ctx.h, 1, n, 8, 0, C.byref(g)))
    off = 0
    for k, sd in enumerate(sets):
        m = n_each[k]
        _view(g.ref_frame_idx, np.int32, n)[off:off + m] = k
        _view(g.cur_frame_idx, np.int32, n)[off:off + m] = k
        for name, width, dt in (("px", 2, np.float64), ("f", 3, np.float64), ("grad", 2, np.float64), ("level", 1, np.int32), ("type", 1, np.uint8), ("state", 4, np.float64)):
            _view(getattr(g, name), dt, width * n)[width * off:width * (off + m)] = sd[name]
        off += m
    fb = capi.svoh_feature_batch()
    fb.n, fb.mem_space, fb.n_cur_frames = n, capi.SVOH_MEM_STAGED, 2
    for k in ("ref_frame_idx", "cur_frame_idx", "px", "f", "grad", "level", "type"):
        setattr(fb, k, getattr(g, k))
    ctx._check(ctx.lib.svoh_update_seeds_batch(ctx.h, C.byref(mopt), C.byref(dopt), 2, refs, curs, C.byref(fb), g.state, g.success, g.result, None))
    ctx._check(ctx.lib.svoh_matcher_flush(ctx.h))
    # the projection behind it, by unit (every ninth point keeps a staged inverse depth: the old one, as a driver would have it)
    units = [np.arange(sum(n_each[:k]), sum(n_each[:k + 1]), dtype=np.int32) for k in range(2)]
    old_mu = [sd["state"].reshape(-1, 4)[:, 0].copy() for sd in sets]
    for k in range(2):
        units[k][::9] = -1
    rc, px_u, vis_u = project(old_mu, units, True)
    assert rc == 0
    ctx._check(ctx.lib.svoh_matcher_collect(ctx.h))
    new_mu = _view(g.state, np.float64, 4 * n).reshape(-1, 4)[:, 0].copy()
    assert (new_mu != np.concatenate(old_mu)).mean() > 0.3   # the update did move the seeds
    want_mu = []
    for k in range(2):
        lo = sum(n_each[:k])
        m = new_mu[lo:lo + n_each[k]].copy()
        m[units[k] < 0] = old_mu[k][units[k] < 0]
        want_mu.append(m)
    rc, px_w, vis_w = project(want_mu, None, False)
    assert rc == 0
    assert np.array_equal(px_u, px_w) and np.array_equal(vis_u, vis_w) and 0.05 < vis_w.mean() < 1.0
    for h in handles:
        ctx._check(ctx.lib.svoh_features_release(ctx.h, h))
    for sc, fr, fc in pairs:
        ctx.release_frame(fr); ctx.release_frame(fc)
