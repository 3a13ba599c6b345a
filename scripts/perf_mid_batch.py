"""Mid-size batches: 512-thread workgroups (one per CU) vs the 256-thread batch geometry."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svo_pro_universal_amd import _capi as capi, frontend as fe
import bench
ctx = fe.Context(0)
ms = ctypes.c_float()
opt = capi.default_align_options(min_level=0)
os.environ["SVOH_ALIGN_CLUSTER"] = "0"
ctx.reload_knobs()
for B in (24, 48, 96, 128, 192, 256, 384, 512):
    problems, scenes, imgs, keep = bench.build_problems(ctx, torch.device("cuda", 0), 0, B, 2000, 4, 4)
    row = []
    for nt in ("512", "256"):
        os.environ["SVOH_ALIGN_THREADS"] = nt
        ctx.reload_knobs()
        ts = []
        for i in range(6):
            ctx.sparse_align(opt, problems)
            ctx.lib.svoh_sparse_align_last_kernel_ms(ctx.h, ctypes.byref(ms))
            if i >= 2: ts.append(ms.value)
        row.append("nt=%s %.3f ms" % (nt, np.median(ts)))
    print("B=%d  %s" % (B, "   ".join(row)), flush=True)
