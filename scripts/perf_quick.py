import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svo_pro_universal_amd import _capi as capi, frontend as fe
import bench
B = int(os.environ.get("B", "1024")); P = int(os.environ.get("P", "4"))
ctx = fe.Context(0)
problems, scenes, imgs, keep = bench.build_problems(ctx, torch.device("cuda", 0), 0, B, 2000, P, 4)
ms = ctypes.c_float()
def run(tag, reps=5, **kw):
    opt = capi.default_align_options(patch_size=P, **kw)
    ts = []
    for i in range(reps + 1):
        res = ctx.sparse_align(opt, problems)
        ctx.lib.svoh_sparse_align_last_kernel_ms(ctx.h, ctypes.byref(ms))
        if i: ts.append(ms.value)
    pit = sum(r.n_patch_iters for r in res)
    print("%s %-14s kernel %.3f ms  %.2f Gpi/s" % (os.environ.get("SVOH_LIB", "product")[-16:], tag, np.mean(ts), pit / np.mean(ts) / 1e6), flush=True)
run("4..0", min_level=0)
run("4..2", min_level=2)
run("0 only", max_level=0, min_level=0)
if os.environ.get("ILLUM", "1") == "1":
    run("4..2 illum", min_level=2, estimate_illumination_gain=1, estimate_illumination_offset=1)
    run("4..0 illum", min_level=0, estimate_illumination_gain=1, estimate_illumination_offset=1)
if os.environ.get("ROBUST", "0") == "1":
    run("4..0 robust", min_level=0, robustification=1)
    run("4..0 robust+illum", min_level=0, robustification=1, estimate_illumination_gain=1, estimate_illumination_offset=1)
