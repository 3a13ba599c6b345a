"""Single-problem latency of the alignment with and without cluster mode."""
import sys, os, ctypes, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svo_pro_universal_amd import _capi as capi, frontend as fe
import bench
ctx = fe.Context(0)
ms = ctypes.c_float()
for N in [int(v) for v in os.environ.get("NS", "180,700,1000,2000,4000,8000,20000").split(",")]:
    problems, scenes, imgs, keep = bench.build_problems(ctx, torch.device("cuda", 0), 0, 1, N, 4, 4)
    for minl in (0, 2):
        opt = capi.default_align_options(min_level=minl)
        row = []
        for g in (os.environ.get("GS", "0,2,4,8,16,32").split(",") + [None]):
            if g is None: os.environ.pop("SVOH_ALIGN_CLUSTER", None)
            else: os.environ["SVOH_ALIGN_CLUSTER"] = g
            ctx.reload_knobs()
            ts = []; t0 = None
            for i in range(int(os.environ.get("REPS", "8"))):
                a = time.perf_counter(); res = ctx.sparse_align(opt, problems); b = time.perf_counter()
                ctx.lib.svoh_sparse_align_last_kernel_ms(ctx.h, ctypes.byref(ms))
                if i >= 2: ts.append((ms.value, (b - a) * 1e3))
            row.append("%s: %.3f/%.3f" % (g or "auto", np.median([t[0] for t in ts]), np.median([t[1] for t in ts])))
        print("N=%d levels 4..%d status %d iters %s  kernel/call ms  %s" % (N, minl, res[0].status, list(res[0].iters)[:5], "  ".join(row)), flush=True)
