"""In-process A/B timing of the alignment kernel under different geometries."""
import sys, os, time, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svo_pro_universal_amd import _capi as capi, frontend as fe, synth
import bench

B = int(os.environ.get("B", "1024"))
P = int(os.environ.get("P", "4"))
dev = torch.device("cuda", 0)
ctx = fe.Context(0)
problems, scenes, imgs, keep = bench.build_problems(ctx, dev, 0, B, 2000, P, 4)
ms = ctypes.c_float()

def run(tag, reps=5, **kw):
    env = {k: str(v) for k, v in kw.pop("env", {}).items()}
    for k in ("SVOH_ALIGN_THREADS", "SVOH_ALIGN_LDS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx.reload_knobs()
    opt = capi.default_align_options(patch_size=P, **kw)
    ts = []
    for i in range(reps + 1):
        res = ctx.sparse_align(opt, problems)
        ctx.lib.svoh_sparse_align_last_kernel_ms(ctx.h, ctypes.byref(ms))
        if i: ts.append(ms.value)
    pit = sum(r.n_patch_iters for r in res)
    its = np.array([list(r.iters)[:5] for r in res]).mean(0)
    print("%-34s kernel %.3f ms (min %.3f)  patch_iters %.2fM  %.2f Gpi/s  iters/level %s" %
          (tag, np.mean(ts), np.min(ts), pit / 1e6, pit / np.mean(ts) / 1e6, np.round(its, 2)), flush=True)

run("default 4..0", min_level=0)
run("levels 4..2", min_level=2)
run("levels 4..1", min_level=1)
run("level 4 only", min_level=4)
run("level 2 only", max_level=2, min_level=2)
run("level 1 only", max_level=1, min_level=1)
run("level 0 only", max_level=0, min_level=0)
run("4..0 no LDS", min_level=0, env=dict(SVOH_ALIGN_LDS=0))
run("4..0 nt512", min_level=0, env=dict(SVOH_ALIGN_THREADS=512))
run("4..0 nt1024", min_level=0, env=dict(SVOH_ALIGN_THREADS=1024))
run("4..0 nt512 lds38400", min_level=0, env=dict(SVOH_ALIGN_THREADS=512, SVOH_ALIGN_LDS=38400))
run("4..0 robust", min_level=0, robustification=1)
run("4..0 illum", min_level=0, estimate_illumination_gain=1, estimate_illumination_offset=1)
