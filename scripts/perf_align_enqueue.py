#!/usr/bin/env python3
"""Host time of one keyed alignment enqueue (svoh_sparse_align_enqueue_keyed) for small problems: what a lock-step round pays per launch geometry."""
import ctypes as C, sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa
from svo_pro_universal_amd import _capi as capi, frontend as fe, synth
ctx = fe.Context(0, kernel_timing=False)
opt = capi.default_align_options(max_level=4, min_level=2)
cam = synth.Camera.euroc_like(752, 480)
for nfeat, nprob in ((120, 3), (180, 3), (240, 2), (540, 1)):
    scs = [synth.make_align_scene(900 + i, n_features=nfeat, patch_size=4, cam=cam) for i in range(nprob)]
    frames = [(ctx.build_pyramid(sc.img_ref, 5), ctx.build_pyramid(sc.img_cur, 5)) for sc in scs]
    pbs, keep = fe.make_align_problems([[(sc, fr, fc)] for sc, (fr, fc) in zip(scs, frames)])
    k = C.c_int32()
    ctx._check(ctx.lib.svoh_sparse_align_geometry_key(ctx.h, C.byref(opt), C.byref(pbs[0]), C.byref(k)))
    for rep in range(3):
        t_enq = t_key = t_fetch = 0.0
        N = 300
        for _ in range(N):
            t0 = time.perf_counter()
            for p in range(nprob):
                ctx.lib.svoh_sparse_align_geometry_key(ctx.h, C.byref(opt), C.byref(pbs[p]), C.byref(k))
            t1 = time.perf_counter()
            ctx._check(ctx.lib.svoh_sparse_align_enqueue_keyed(ctx.h, C.byref(opt), nprob, pbs, k.value))
            t2 = time.perf_counter()
            ctx.sparse_align_fetch_all(nprob)
            t3 = time.perf_counter()
            t_key += t1 - t0; t_enq += t2 - t1; t_fetch += t3 - t2
    print("%d problems x %d features (key %#x): geometry keys %.1f us, enqueue %.1f us, wait + fetch %.1f us" % (nprob, nfeat, k.value, 1e6 * t_key / N, 1e6 * t_enq / N, 1e6 * t_fetch / N))

# a two-camera bundle with gain / offset and a rotation prior (the stereo front end's problem), host arrays
opt8 = capi.default_align_options(max_level=4, min_level=2, estimate_illumination_gain=1, estimate_illumination_offset=1)
for nfeat, nprob in ((170, 1), (170, 8)):
    items, priors, keep_all = [], [], []
    for i in range(nprob):
        scs = [synth.make_align_scene(950 + 2 * i + c, n_features=nfeat, patch_size=4, cam=cam, gain=1.03, offset=2.0) for c in range(2)]
        frames = [(ctx.build_pyramid(sc.img_ref, 5), ctx.build_pyramid(sc.img_cur, 5)) for sc in scs]
        items.append([(sc, fr, fc) for sc, (fr, fc) in zip(scs, frames)])
        pr = capi.svoh_align_prior(); pr.have_prior = 1; pr.T_prior = fe._se3(scs[0].T_icur_iref_gt); pr.lambda_rot = 0.5
        priors.append(pr)
    pbs, keep = fe.make_align_problems(items, prior=priors)
    k = C.c_int32()
    ctx._check(ctx.lib.svoh_sparse_align_geometry_key(ctx.h, C.byref(opt8), C.byref(pbs[0]), C.byref(k)))
    for rep in range(3):
        t_enq = t_fetch = 0.0
        N = 300
        for _ in range(N):
            t1 = time.perf_counter()
            ctx._check(ctx.lib.svoh_sparse_align_enqueue_keyed(ctx.h, C.byref(opt8), nprob, pbs, k.value))
            t2 = time.perf_counter()
            res = ctx.sparse_align_fetch_all(nprob)
            t3 = time.perf_counter()
            t_enq += t2 - t1; t_fetch += t3 - t2
    print("%d bundles of 2 x %d features, 8 parameters (key %#x): enqueue %.1f us, wait + fetch %.1f us, iterations %s" % (nprob, nfeat, k.value, 1e6 * t_enq / N, 1e6 * t_fetch / N, list(res[0].iters)[:5]))
