"""Single-frame latency (ms/frame) of the alignment kernel for small batches."""
import sys, os, ctypes, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svo_pro_universal_amd import _capi as capi, frontend as fe
import bench
ctx = fe.Context(0)
problems, scenes, imgs, keep = bench.build_problems(ctx, torch.device("cuda", 0), 0, 16, 2000, 4, 4)
ms = ctypes.c_float()
for n in (1, 2, 8, 16):
    for nt in (256, 512, 1024):
        for levels in ((4, 0), (4, 2)):
            os.environ["SVOH_ALIGN_THREADS"] = str(nt)
            ctx.reload_knobs()
            opt = capi.default_align_options(max_level=levels[0], min_level=levels[1])
            pbs = (capi.svoh_align_problem * n)(*[problems[i] for i in range(n)])
            ks, ws = [], []
            for i in range(6):
                t0 = time.perf_counter()
                res = ctx.sparse_align(opt, pbs)
                w = time.perf_counter() - t0
                ctx.lib.svoh_sparse_align_last_kernel_ms(ctx.h, ctypes.byref(ms))
                if i: ks.append(ms.value); ws.append(w * 1e3)
            print("n=%2d nt=%4d levels %d..%d: kernel %.3f ms, wall (host ptr path) %.3f ms, iters %s" % (n, nt, levels[0], levels[1], np.mean(ks), np.mean(ws), list(res[0].iters)[:5]), flush=True)
