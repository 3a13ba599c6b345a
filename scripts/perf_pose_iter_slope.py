import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from svo_pro_universal_amd import _capi as capi, frontend as fe
import bench, pose_helpers as ph
ctx = fe.Context(0)
scenes = [ph.make_pose_scene(500 + i, n=180) for i in range(16)]
built = [fe.make_pose_problem(sc["cams"], sc["T_imu_world_init"]) for sc in scenes]
for B in (1, 11):
    probs = [built[i][0] for i in range(B)]
    for mi in (1, 2, 4, 8, 10):
        opt = capi.default_pose_options(scenes[0]["cam"])
        opt.max_iter = mi
        ks, cs = [], []
        for i in range(12):
            t0 = time.perf_counter(); res = ctx.optimize_pose(opt, probs); t1 = time.perf_counter()
            if i >= 2: ks.append(bench.misc_kernel_ms(ctx)); cs.append((t1 - t0) * 1e3)
        print("B=%d max_iter %d kernel %.4f ms call %.4f ms iters %s" % (B, mi, np.median(ks), np.median(cs), [r.iters for r in res][:4]), flush=True)
