import sys, os, ctypes, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svo_pro_universal_amd import _capi as capi, frontend as fe, synth
import bench
B = int(os.environ.get("B", "256")); NT = 400
ctx = fe.Context(0)
dev = torch.device("cuda", 0)
cam, scenes, imgs, frames = bench.render_pairs(ctx, dev, 0, B, 4, rot_deg=(0.5, 1.5), trans_m=(0.05, 0.15))
tracks = [synth.make_track_set(sc, NT, seed=i) for i, sc in enumerate(scenes)]
px_ref = np.concatenate([t["px_ref"] for t in tracks]); px0 = np.concatenate([t["px_cur_init"] for t in tracks])
n = B * NT
rf = (capi.svoh_frame_t * n)(*[frames[2 * (i // NT)] for i in range(n)])
cf = (capi.svoh_frame_t * n)(*[frames[2 * (i // NT) + 1] for i in range(n)])
status = np.zeros(n, np.uint8)
def run(tag, **kw):
    opt = capi.default_klt_options(**kw)
    ts = []
    for i in range(4):
        out = px0.copy()
        t0 = time.perf_counter()
        ctx._check(ctx.lib.svoh_klt_track_multi(ctx.h, ctypes.byref(opt), n, rf, cf, px_ref.ctypes.data, out.ctypes.data, status.ctypes.data))
        wall = time.perf_counter() - t0
        if i: ts.append((bench.misc_kernel_ms(ctx), wall * 1e3))
    print("%-28s kernel %.3f ms  wall %.3f ms  ok %.3f counters %s" % (tag, np.mean([t[0] for t in ts]), np.mean([t[1] for t in ts]), status.mean(), bench.misc_counters(ctx)[:4]), flush=True)
for blk in (64, 128, 256):
    os.environ["SVOH_KLT_BLOCK"] = str(blk)
    ctx.reload_knobs()
    run("block %d" % blk)
run("max_iter 1", max_iter=1)
run("levels 4..4", min_level=4)
run("level 0 only", max_level=0)
