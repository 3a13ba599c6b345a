"""Kernel time of ONE stereo bundle alignment as the stereo harness issues it: two cameras, 4x4 patches, gain + offset
estimated, rotation prior, levels 4..2 -- against the same patches as one camera, and without the illumination terms /
the prior.  N = patches per camera (default 125)."""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from svo_pro_universal_amd import _capi as capi, frontend as fe, synth
import helpers
N = int(os.environ.get("N", "125"))
ctx = fe.Context(0)
cam = synth.Camera.euroc_like()
a = synth.make_align_scene(141, n_features=N, cam=cam, gain=1.03, offset=2.0)
b = synth.make_align_scene(141, n_features=N, cam=cam, gain=1.03, offset=2.0)
c = synth.make_align_scene(141, n_features=2 * N, cam=cam, gain=1.03, offset=2.0)
Tp = synth.SE3(synth.quat_from_axis_angle([0.2, -1, 0.3], 0.003), [0.0, 0.0, 0.0])
prior = helpers.make_prior(Tp, 0.5, 0.0)
ms = ctypes.c_float()
def frames(sc): return ctx.build_pyramid(sc.img_ref, 5), ctx.build_pyramid(sc.img_cur, 5)
fa, fb, fc = frames(a), frames(b), frames(c)
def run(tag, items, prior, **kw):
    pbs, keep = fe.make_align_problems([items], prior=prior)
    opt = capi.default_align_options(min_level=2, **kw)
    ts = []
    for i in range(8):
        res = ctx.sparse_align(opt, pbs)
        ctx.lib.svoh_sparse_align_last_kernel_ms(ctx.h, ctypes.byref(ms))
        if i >= 2: ts.append(ms.value)
    r = res[0]
    print("%-58s kernel %.4f ms  evaluateError calls %s  patch-iterations %d" % (tag, np.median(ts), [r.iters[l] for l in (4, 3, 2)], r.n_patch_iters), flush=True)
il = dict(estimate_illumination_gain=1, estimate_illumination_offset=1)
run("2 cams x %d, gain+offset, rotation prior (the harness)" % N, [(a, fa[0], fa[1]), (b, fb[0], fb[1])], prior, **il)
run("2 cams x %d, gain+offset" % N, [(a, fa[0], fa[1]), (b, fb[0], fb[1])], None, **il)
run("2 cams x %d, pose only" % N, [(a, fa[0], fa[1]), (b, fb[0], fb[1])], None)
run("1 cam x %d, gain+offset, rotation prior" % (2 * N), [(c, fc[0], fc[1])], prior, **il)
run("1 cam x %d, pose only" % (2 * N), [(c, fc[0], fc[1])], None)
run("1 cam x %d, pose only" % N, [(a, fa[0], fa[1])], None)
