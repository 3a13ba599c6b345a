"""Small batches of large problems: one workgroup per problem vs a cluster of workgroups per problem."""
import sys, os, ctypes, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svo_pro_universal_amd import _capi as capi, frontend as fe
import bench
ctx = fe.Context(0)
ms = ctypes.c_float()
opt = capi.default_align_options(min_level=0)
for B, N in ((2, 2000), (8, 2000), (16, 2000), (32, 2000), (64, 2000), (128, 2000), (32, 700), (64, 4000)):
    problems, scenes, imgs, keep = bench.build_problems(ctx, torch.device("cuda", 0), 0, B, N, 4, 4)
    row = []
    for g in ("0", None):
        if g is None: os.environ.pop("SVOH_ALIGN_CLUSTER", None)
        else: os.environ["SVOH_ALIGN_CLUSTER"] = g
        ctx.reload_knobs()
        ts = []
        for i in range(8):
            res = ctx.sparse_align(opt, problems)
            ctx.lib.svoh_sparse_align_last_kernel_ms(ctx.h, ctypes.byref(ms))
            if i >= 2: ts.append(ms.value)
        row.append("%s %.3f ms" % ("one workgroup each" if g else "clustered", np.median(ts)))
    print("B=%d N=%d  %s" % (B, N, "   ".join(row)), flush=True)
