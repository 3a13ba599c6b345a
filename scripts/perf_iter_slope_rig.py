"""Kernel time of ONE alignment problem against the number of Gauss-Newton iterations (eps = 0), for one camera / a two-camera rig and with / without
the illumination terms: what an iteration of the stereo front end's bundle costs, and which part of it the rig and the two extra parameters are.
N=<patches per camera> (170), P=<patch size> (4)."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svo_pro_universal_amd import _capi as capi, frontend as fe, synth
ctx = fe.Context(0)
dev = torch.device("cuda", 0)
ms = ctypes.c_float()
N = int(os.environ.get("N", "170")); P = int(os.environ.get("P", "4"))
cams = [synth.Camera.euroc_like(752, 480), synth.Camera.euroc_like(752, 480)]
scenes = [synth.make_align_scene(7, n_features=N, patch_size=P, cam=cam, max_level=4, render_images=False, gain=1.03, offset=2.0) for cam in cams]
frames, keep_imgs = [], []
for cam, sc in zip(cams, scenes):
    imgs = synth.render_batch_torch(cam, [sc.T_w_ref, sc.T_w_cur], [sc.plane] * 2, [sc.tex] * 2, dev, gains=[1.0, 1.03], offsets=[0.0, 2.0])
    torch.cuda.synchronize()
    frames.append(ctx.build_pyramid_batch_device(imgs.data_ptr(), cam.width * cam.height, 2, cam.width, cam.height, cam.width, 5))
    keep_imgs.append(imgs)
ctx.synchronize()
px = torch.from_numpy(np.concatenate([s.px for s in scenes])).to(dev)
f = torch.from_numpy(np.concatenate([s.f for s in scenes])).to(dev)
pw = torch.from_numpy(np.concatenate([s.pos_world for s in scenes])).to(dev)
fl = torch.from_numpy(np.concatenate([s.flags for s in scenes])).to(dev)


def problem(n_cams):
    cams_i, off = [], 0
    for c in range(n_cams):
        sc = scenes[c]
        dp = dict(px=px.data_ptr() + 16 * off, f=f.data_ptr() + 24 * off, pos_world=pw.data_ptr() + 24 * off, flags=fl.data_ptr() + off)
        cams_i.append((sc, frames[c][0], frames[c][1], dp))
        off += sc.n_features
    pr = capi.svoh_align_prior()
    pr.have_prior = 1
    pr.T_prior = fe._se3(scenes[0].T_icur_iref_gt)
    pr.lambda_rot = 0.5
    return fe.make_align_problems([cams_i], prior=[pr])


for n_cams in (1, 2):
    problems, keep = problem(n_cams)
    for illum in (0, 1):
        xs, ys = [], []
        for mi in (1, 2, 4, 8, 10, 16):
            opt = capi.default_align_options(patch_size=P, max_level=4, min_level=2, max_iter=mi, eps=0.0, estimate_illumination_gain=illum, estimate_illumination_offset=illum)
            ts = []
            for i in range(30):
                res = ctx.sparse_align(opt, problems)
                ctx.lib.svoh_sparse_align_last_kernel_ms(ctx.h, ctypes.byref(ms))
                if i >= 5:
                    ts.append(ms.value)
            xs.append(sum(res[0].iters)); ys.append(np.median(ts))
        a, b = np.polyfit(xs, ys, 1)
        print("%d camera(s) x %d patches %dx%d, illumination %s: %.2f us per iteration, %.1f us outside the iterations; 3 levels x 10 iterations: %.4f ms"
              % (n_cams, N, P, P, "on " if illum else "off", a * 1e3, b * 1e3, ys[4]), flush=True)
# with the reference's own stopping rule (eps 5e-4): how many iterations a level takes when gain and offset are estimated
for illum in (0, 1):
    problems, keep = problem(2)
    opt = capi.default_align_options(patch_size=P, max_level=4, min_level=2, estimate_illumination_gain=illum, estimate_illumination_offset=illum)
    res = ctx.sparse_align(opt, problems)
    print("2 cameras, illumination %s, eps 5e-4: iterations per level (0..4) %s" % ("on " if illum else "off", list(res[0].iters)[:5]), flush=True)
