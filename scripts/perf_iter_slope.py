"""Kernel time of ONE small alignment problem against the number of Gauss-Newton iterations (eps = 0: every level runs
max_iter iterations): slope = cost of an iteration, intercept = what a launch pays before and after them."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svo_pro_universal_amd import _capi as capi, frontend as fe
import bench
ctx = fe.Context(0)
ms = ctypes.c_float()
N = int(os.environ.get("N", "180")); P = int(os.environ.get("P", "4"))
problems, scenes, imgs, keep = bench.build_problems(ctx, torch.device("cuda", 0), 0, 1, N, P, 4)
pbs = (capi.svoh_align_problem * 1)(problems[0])
for levels in ((4, 2), (2, 2)):
    xs, ys = [], []
    for mi in (1, 2, 4, 8, 16):
        opt = capi.default_align_options(patch_size=P, max_level=levels[0], min_level=levels[1], max_iter=mi, eps=0.0)
        ts = []
        for i in range(30):
            res = ctx.sparse_align(opt, pbs)
            ctx.lib.svoh_sparse_align_last_kernel_ms(ctx.h, ctypes.byref(ms))
            if i >= 5: ts.append(ms.value)
        it = sum(res[0].iters)
        xs.append(it); ys.append(np.median(ts))
        print("levels %d..%d max_iter %2d: %2d iterations, kernel median %.4f ms" % (levels[0], levels[1], mi, it, np.median(ts)), flush=True)
    a, b = np.polyfit(xs, ys, 1)
    print("  -> %.2f us per iteration, %.1f us outside the iterations (%s)" % (a * 1e3, b * 1e3, os.environ.get("SVOH_LIB", "product")[-20:]), flush=True)
