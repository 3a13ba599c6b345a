"""Rows geometry (P lanes per patch) against the lane-per-patch geometries: kernel time of n problems of N patches.
SVOH_LIB picks the library; usage: python scripts/perf_rows.py [quick]"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svo_pro_universal_amd import _capi as capi, frontend as fe
import bench
ctx = fe.Context(0)
ms = ctypes.c_float()
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
cases = [(180, 4, (1, 8, 64)), (180, 8, (1, 8)), (100, 4, (1,)), (60, 8, (1,)), (700, 4, (1,)), (2000, 4, (1, 16))]
if quick: cases = cases[:2]
geoms = [("lane/patch 256", dict(SVOH_ALIGN_ROWS="0", SVOH_ALIGN_THREADS="256")),
         ("lane/patch 512", dict(SVOH_ALIGN_ROWS="0", SVOH_ALIGN_THREADS="512")),
         ("2 lanes/patch", dict(SVOH_ALIGN_ROWS="2", SVOH_ALIGN_THREADS="512")),
         ("4 lanes/patch", dict(SVOH_ALIGN_ROWS="4", SVOH_ALIGN_THREADS="512")),
         ("8 lanes/patch", dict(SVOH_ALIGN_ROWS="8", SVOH_ALIGN_THREADS="512")),
         ("default", dict())]
for N, P, ns in cases:
    problems, scenes, imgs, keep = bench.build_problems(ctx, torch.device("cuda", 0), 0, max(ns), N, P, 4)
    for n in ns:
        for kw in (dict(min_level=2), dict(min_level=0)):
            opt = capi.default_align_options(patch_size=P, **kw)
            pbs = (capi.svoh_align_problem * n)(*[problems[i] for i in range(n)])
            ref = None
            for tag, env in geoms:
                for k in ("SVOH_ALIGN_ROWS", "SVOH_ALIGN_THREADS", "SVOH_ALIGN_CLUSTER"): os.environ.pop(k, None)
                os.environ.update(env)
                if tag != "default": os.environ["SVOH_ALIGN_CLUSTER"] = "0"
                ctx.reload_knobs()
                ts = []
                for i in range(25):
                    res = ctx.sparse_align(opt, pbs)
                    ctx.lib.svoh_sparse_align_last_kernel_ms(ctx.h, ctypes.byref(ms))
                    if i >= 5: ts.append(ms.value)
                T = np.array([list(r.T_icur_iref.q) + list(r.T_icur_iref.t) for r in res])
                it = [list(r.iters)[:5] for r in res]
                if ref is None: ref = (T, it)
                dev = np.abs(T - ref[0]).max()
                print("N=%4d P=%d n=%2d levels 4..%d %-15s kernel median %.4f ms (min %.4f)  iters %s  |T - T_first| %.1e %s" % (
                    N, P, n, kw["min_level"], tag, np.median(ts), np.min(ts), it[0], dev, "" if it == ref[1] else "ITERS DIFFER"), flush=True)
