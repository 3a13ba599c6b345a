"""Minimal driver for rocprofv3: build B problems, run the alignment kernel a few times."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svo_pro_universal_amd import _capi as capi, frontend as fe
import bench
B = int(os.environ.get("B", "1024")); P = int(os.environ.get("P", "4")); REPS = int(os.environ.get("REPS", "3"))
ctx = fe.Context(0)
problems, scenes, imgs, keep = bench.build_problems(ctx, torch.device("cuda", 0), 0, B, 2000, P, 4)
opt = capi.default_align_options(patch_size=P, min_level=int(os.environ.get("MINL", "0")), max_level=int(os.environ.get("MAXL", "4")))
for i in range(REPS):
    res = ctx.sparse_align(opt, problems)
print("patch_iters", sum(r.n_patch_iters for r in res))
