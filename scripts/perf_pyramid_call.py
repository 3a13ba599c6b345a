"""Host cost of a frame's life at the C ABI: svoh_build_pyramid (host image, 5 levels) + svoh_release_frame of the
frame before it, as a per-frame front end issues them.  Prints the call's own time (what the caller's thread spends
inside the two calls) and the time until the pyramid is complete on the device.  SVOH_LIB picks the library (A/B)."""
import sys, os, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svo_pro_universal_amd import frontend as fe

W, H, L = int(os.environ.get("W", "752")), int(os.environ.get("H", "480")), 5
ctx = fe.Context(0)
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(8)]


def loop(n, sync):
    prev = ctx.build_pyramid(imgs[0], L)
    ctx.synchronize()
    t_build, t_rel, t_sync = [], [], []
    for k in range(n):
        t0 = time.perf_counter()
        cur = ctx.build_pyramid(imgs[k % 8], L)
        t1 = time.perf_counter()
        if sync:
            ctx.synchronize()
        t2 = time.perf_counter()
        ctx.release_frame(prev)
        t3 = time.perf_counter()
        prev = cur
        t_build.append(t1 - t0); t_sync.append(t2 - t1); t_rel.append(t3 - t2)
    ctx.synchronize()
    ctx.release_frame(prev)
    med = lambda v: 1e6 * float(np.median(v[n // 4:]))
    return med(t_build), med(t_sync), med(t_rel)


for sync in (True, False):
    loop(50, sync)
    b, s, r = loop(400, sync)
    print("%-16s %dx%d  build call %.1f us  %s release call %.1f us  (medians of 300 frames)" % (
        os.environ.get("SVOH_LIB", "product")[-16:], W, H, b, ("wait for the device %.1f us " % s) if sync else "(no wait)              ", r), flush=True)
