"""Host cost of one seed-update call (blocking, and split into the parts of a deferred section) at per-frame sizes."""
import sys, os, time, ctypes as C
import numpy as np
import torch  # first: libsvo_hip must share torch's HIP runtime
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from svo_pro_universal_amd import _capi as capi, frontend as fe
from svo_pro_universal_amd import synth

ctx = fe.Context(0, kernel_timing=False)
n_kf, per_kf = 3, int(os.environ.get("PER_KF", "300"))
sc = synth.make_align_scene(11, n_features=8, cam=synth.Camera.euroc_like(752, 480), rot_deg=(0.3, 0.8), trans_m=(0.05, 0.12))
fr, fc = ctx.build_pyramid(sc.img_ref, 5), ctx.build_pyramid(sc.img_cur, 5)
sd = synth.make_seed_set(sc, per_kf * n_kf)
rv = [fe.make_frame_view(fr, sc.cam, sc.T_ref_f_w, sd["mu_range"], 1 + k) for k in range(n_kf)]
cv = fe.make_frame_view(fc, sc.cam, sc.T_cur_f_w_gt, 0.0, 99)
idx = (np.arange(per_kf * n_kf) % n_kf).astype(np.int32)
fb, keep = fe.make_feature_batch(idx, sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(sc.cam)
n = fb.n
rva = (capi.svoh_frame_view * n_kf)(*rv)
state0 = np.ascontiguousarray(sd["state"], np.float64)
type0 = keep["type"].copy()
succ = np.zeros(n, np.uint8); mr = np.zeros(n, np.int32); ns = C.c_int32()
lib, h = ctx.lib, ctx.h
def call():
    st = state0.copy(); keep["type"][:] = type0
    t0 = time.perf_counter()
    rc = lib.svoh_update_seeds_batch(h, C.byref(mopt), C.byref(dopt), n_kf, rva, C.byref(cv), C.byref(fb), st.ctypes.data, succ.ctypes.data, mr.ctypes.data, C.byref(ns))
    t1 = time.perf_counter(); assert rc == 0
    return (t1 - t0) * 1e6
def call_deferred():
    st = state0.copy(); keep["type"][:] = type0
    t0 = time.perf_counter()
    assert lib.svoh_matcher_begin_deferred(h) == 0
    t1 = time.perf_counter()
    rc = lib.svoh_update_seeds_batch(h, C.byref(mopt), C.byref(dopt), n_kf, rva, C.byref(cv), C.byref(fb), st.ctypes.data, succ.ctypes.data, mr.ctypes.data, C.byref(ns))
    t2 = time.perf_counter(); assert rc == 0
    assert lib.svoh_matcher_collect(h) == 0
    t3 = time.perf_counter()
    return (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6
for _ in range(20): call(); call_deferred()
a = np.array([call() for _ in range(200)])
ctx.set_kernel_timing(True); call(); kms = C.c_float(); lib.svoh_last_kernel_ms(h, C.byref(kms)); ctx.set_kernel_timing(False)
d = np.array([call_deferred() for _ in range(200)])
print("%d seeds of %d keyframes: blocking call median %.1f us (kernel %.1f us); deferred: begin %.1f, stage %.1f, collect %.1f us"
      % (n, n_kf, np.median(a), kms.value * 1e3, *np.median(d, axis=0)))
