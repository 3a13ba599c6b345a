"""Kernel time of ONE alignment problem of C3 size (180 patches, levels 4..2) and of a small batch; SVOH_LIB picks the library."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svo_pro_universal_amd import _capi as capi, frontend as fe
import bench
ctx = fe.Context(0)
problems, scenes, imgs, keep = bench.build_problems(ctx, torch.device("cuda", 0), 0, 8, 180, 4, 4)
ms = ctypes.c_float()
for n in (1, 8):
    for kw in (dict(min_level=2), dict(min_level=0)):
        opt = capi.default_align_options(patch_size=4, **kw)
        pbs = (capi.svoh_align_problem * n)(*[problems[i] for i in range(n)])
        ts = []
        for i in range(30):
            res = ctx.sparse_align(opt, pbs)
            ctx.lib.svoh_sparse_align_last_kernel_ms(ctx.h, ctypes.byref(ms))
            if i >= 5: ts.append(ms.value)
        print("%s n=%d levels 4..%d: kernel median %.4f ms (min %.4f), iters %s" % (os.environ.get("SVOH_LIB", "product")[-14:], n, kw["min_level"], np.median(ts), np.min(ts), list(res[0].iters)[:5]), flush=True)
