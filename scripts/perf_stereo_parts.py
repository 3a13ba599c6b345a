"""Where the stereo seam's kernel time goes: the same batch with the refinement off / one pass, and with a short scan."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svo_pro_universal_amd import _capi as capi, frontend as fe, synth
import bench
ctx = fe.Context(0, kernel_timing=True)
dev = torch.device("cuda", 0)
B, NF = 64, 120
cam, scenes, imgs, frames = bench.render_pairs(ctx, dev, 0, B, 4, rot_deg=(0.0, 0.2), trans_m=(0.08, 0.12))
feats = [synth.make_seed_set(sc, NF, seed=i, margin=6, levels=(0, 1, 2)) for i, sc in enumerate(scenes)]
ref_views = [fe.make_frame_view(frames[2 * i], cam, sc.T_ref_f_w, 0.0, 2 * i) for i, sc in enumerate(scenes)]
cur_views = [fe.make_frame_view(frames[2 * i + 1], cam, sc.T_cur_f_w_gt, 0.0, 2 * i + 1) for i, sc in enumerate(scenes)]
idx = np.repeat(np.arange(B, dtype=np.int32), NF)
cat = lambda k: np.concatenate([f[k] for f in feats])
ftype = np.where(cat("type") == 0, capi.FT_EDGELET, capi.FT_CORNER).astype(np.uint8)
fb, keep = fe.make_feature_batch(idx, cat("px"), cat("f"), cat("grad"), cat("level"), ftype)
fb.cur_frame_idx = idx.ctypes.data
fb.n_cur_frames = B
d_inv = np.concatenate([np.tile([1.0 / np.median(f["true_depth"]), 1.0 / (0.3 * np.median(f["true_depth"])),
                                 1.0 / (15.0 * np.median(f["true_depth"]))], NF) for f in feats])
for name, kw in (("default", dict()), ("no refinement", dict(subpix_refinement=0)), ("one refinement pass", dict(align_max_iter=1)),
                 ("scan <= 64 steps", dict(max_epi_search_steps=64)), ("scan <= 64, no refinement", dict(max_epi_search_steps=64, subpix_refinement=0)),
                 ("unit plane", dict(scan_on_unit_sphere=0))):
    o = dict(max_epi_search_steps=500, subpix_refinement=1, scan_on_unit_sphere=1); o.update(kw)
    mopt = capi.default_matcher_options(**o)
    ts = []
    for i in range(12):
        out = ctx.epipolar_match_batch(mopt, ref_views, cur_views, fb, d_inv=d_inv)
        if i >= 2: ts.append(bench.misc_kernel_ms(ctx))
    cnt = bench.misc_counters(ctx)
    print("%-28s kernel %.4f ms  success %.2f  counters (warps, zmssd, align iters, ok) %s" % (name, np.median(ts), (out["result"] == 0).mean(), cnt[:4]), flush=True)
