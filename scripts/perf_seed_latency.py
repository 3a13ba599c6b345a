"""Kernel time and per-seed work tails of the seed update at C3 size (3 keyframes x 540 seeds) and smaller."""
import sys, os, ctypes, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svo_pro_universal_amd import _capi as capi, frontend as fe, synth
import bench
ctx = fe.Context(0)
dev = torch.device("cuda", 0)
cam, scenes, imgs, frames = bench.render_pairs(ctx, dev, 0, 1, 4, rot_deg=(0.3, 1.0), trans_m=(0.03, 0.10))
sc = scenes[0]
mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(cam)
for n in (64, 540, 1620, 6000):
    sd = synth.make_seed_set(sc, n, seed=2)
    ref_views = [fe.make_frame_view(frames[0], cam, sc.T_ref_f_w, sd["mu_range"], 0)]
    cur_view = fe.make_frame_view(frames[1], cam, sc.T_cur_f_w_gt, 0.0, 1)
    idx = np.zeros(n, np.int32)
    ks, cs, calls = [], None, []
    for i in range(8):
        fb, kk = fe.make_feature_batch(idx, sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
        t0 = time.perf_counter()
        ns, st, succ, mr = ctx.update_seeds_batch(mopt, dopt, ref_views, cur_view, fb, sd["state"])
        t1 = time.perf_counter()
        if i >= 2:
            ks.append(bench.misc_kernel_ms(ctx)); calls.append((t1 - t0) * 1e3)
        cs = bench.misc_counters(ctx)
    print("n=%d kernel %.3f ms call %.3f ms  warps %d zmssd %d align_its %d  tails: align>=5 %d >=10 %d  zmssd>=20 %d >=50 %d"
          % (n, np.median(ks), np.median(calls), cs[0], cs[1], cs[2], cs[4], cs[5], cs[6], cs[7]), flush=True)
