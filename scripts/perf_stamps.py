import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svo_pro_universal_amd import _capi as capi, frontend as fe
import bench
B = int(os.environ.get("B", "512")); P = int(os.environ.get("P", "4")); N = int(os.environ.get("N", "2000"))
ctx = fe.Context(0)
problems, scenes, imgs, keep = bench.build_problems(ctx, torch.device("cuda", 0), 0, B, N, P, 4)
for kw in (dict(min_level=0), dict(min_level=2), dict(max_level=0, min_level=0)):
    opt = capi.default_align_options(patch_size=P, **kw)
    print(kw, flush=True)
    for i in range(2):
        res = ctx.sparse_align(opt, problems)
    print("iters of problem 0:", list(res[0].iters), "sum", sum(res[0].iters), flush=True)
