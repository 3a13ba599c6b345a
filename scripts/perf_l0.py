"""Level 0 alone (the level read from global memory), 1024 x 2000 patches: for counter passes."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svo_pro_universal_amd import _capi as capi, frontend as fe
import bench
P = int(os.environ.get("P", "4"))
ctx = fe.Context(0)
problems, scenes, imgs, keep = bench.build_problems(ctx, torch.device("cuda", 0), 0, 1024, 2000, P, 4)
ms = ctypes.c_float()
opt = capi.default_align_options(patch_size=P, max_level=int(os.environ.get("LVL", "0")), min_level=int(os.environ.get("LVL", "0")))
ts = []
for i in range(4):
    res = ctx.sparse_align(opt, problems)
    ctx.lib.svoh_sparse_align_last_kernel_ms(ctx.h, ctypes.byref(ms)); ts.append(ms.value)
print("level", os.environ.get("LVL", "0"), "kernel ms", ts, "iters", list(res[0].iters)[:5], "patch-iters", sum(r.n_patch_iters for r in res))
