"""Kernel time of ONE pose bundle against its size: features per camera x cameras (the stereo harness' bundle is
2 x 160).  Prints kernel ms, iterations and measurements."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from svo_pro_universal_amd import _capi as capi, frontend as fe
import bench, pose_helpers as ph
ctx = fe.Context(0)
for n_cams, n in ((1, 90), (1, 180), (1, 250), (1, 320), (1, 500), (2, 90), (2, 160), (2, 250)):
    sc = ph.make_pose_scene(700 + n + n_cams, n=n, n_cams=n_cams)
    opt = capi.default_pose_options(sc["cam"])
    pb, keep = fe.make_pose_problem(sc["cams"], sc["T_imu_world_init"])
    ks = []
    for i in range(8):
        res = ctx.optimize_pose(opt, [pb])
        if i >= 2: ks.append(bench.misc_kernel_ms(ctx))
    print("%d cam x %3d features: kernel %.4f ms  iters %d  n_meas %d" % (n_cams, n, np.median(ks), res[0].iters, res[0].n_meas), flush=True)
