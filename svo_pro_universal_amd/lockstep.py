"""ctypes plumbing over the C face of FrontendLockstep (host/svo_hip_lockstep_c.h in libsvo_hip_host.so): many camera
streams through the per-frame chain in lock step, one launch per stage.  The engine itself is C++ (host/svo_hip_lockstep.cpp);
this file only marshals pointers for bench.py and the tests.  No CPU path: the library is loaded or an error is raised."""
import ctypes as C
import os

import numpy as np

from . import _capi as capi
from . import frontend as fe

HOST_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "libsvo_hip_host.so")
_HOST = None


def load_host():
    global _HOST
    if _HOST is not None:
        return _HOST
    capi.load()   # libsvo_hip.so first: the same HIP runtime as the rest of the process (see _capi._share_hip_runtime_with_torch)
    if not os.path.exists(HOST_LIB_PATH):
        raise RuntimeError("libsvo_hip_host.so not built (%s): __graft_entry__.build()" % HOST_LIB_PATH)
    lib = C.CDLL(HOST_LIB_PATH, mode=C.RTLD_GLOBAL)
    P = C.POINTER
    lib.svohl_create.argtypes = [C.c_void_p, C.c_int, P(capi.svoh_camera), P(capi.svoh_se3), C.c_char_p, C.c_double, C.c_double, C.c_double,
                                 C.c_int, C.c_int, C.c_int, P(C.c_void_p)]
    lib.svohl_pool_create.argtypes = [C.c_int, P(C.c_void_p)]
    lib.svohl_pool_create_exclusive.argtypes = [C.c_int, P(C.c_void_p)]
    lib.svohl_pool_destroy.argtypes = [C.c_void_p]
    lib.svohl_pool_destroy.restype = None
    lib.svohl_create_shared.argtypes = [C.c_void_p, C.c_int, P(capi.svoh_camera), P(capi.svoh_se3), C.c_char_p, C.c_double, C.c_double, C.c_double,
                                        C.c_int, C.c_void_p, C.c_int, C.c_int, P(C.c_void_p)]
    lib.svohl_create_streams.argtypes = [C.c_void_p, C.c_int, P(capi.svoh_camera), P(capi.svoh_se3), P(C.c_char_p), P(C.c_double), P(C.c_int), P(C.c_int),
                                         C.c_int, C.c_void_p, C.c_int, C.c_int, P(C.c_void_p)]
    lib.svohl_create_streams_cameras.argtypes = lib.svohl_create_streams.argtypes
    lib.svohl_run_schedule.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_long, C.c_int, P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_int),
                                       C.c_void_p, C.c_void_p, P(C.c_long)]
    lib.svohl_destroy.argtypes = [C.c_void_p]
    lib.svohl_destroy.restype = None
    lib.svohl_add_images.argtypes = [C.c_void_p, P(C.c_void_p), C.c_int, C.c_void_p]
    lib.svohl_pose.argtypes = [C.c_void_p, C.c_int, P(capi.svoh_se3)]
    lib.svohl_run_sequence.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_void_p]
    lib.svohl_last_round.argtypes = [C.c_void_p, P(C.c_double), P(C.c_int)]
    lib.svohl_completed_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, P(C.c_int64), P(C.c_int)]
    lib.svohl_finish.argtypes = [C.c_void_p]
    lib.svohl_phase_times.argtypes = [C.c_void_p, C.c_int, P(C.c_double), P(C.c_int)]
    lib.svohl_phase_name.argtypes = [C.c_int]
    lib.svohl_phase_name.restype = C.c_char_p
    lib.svohl_last_error.restype = C.c_char_p
    lib.svohs_create.argtypes = [C.c_void_p, C.c_int, P(capi.svoh_camera), P(capi.svoh_se3), C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_int, P(C.c_void_p)]
    lib.svohs_destroy.argtypes = [C.c_void_p]
    lib.svohs_destroy.restype = None
    lib.svohs_run_sequence.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.svohs_pose.argtypes = [C.c_void_p, C.c_int, P(capi.svoh_se3)]
    lib.svohs_phase_times.argtypes = [C.c_void_p, P(C.c_double)]
    lib.svohs_finish.argtypes = [C.c_void_p]
    lib.svohs_last_error.restype = C.c_char_p
    _HOST = lib
    return lib


class PinnedImages(object):
    """n_copies copies (one per camera stream) of a sequence of equally sized u8 images in page-locked memory of a context
    (svoh_host_alloc).  One copy per stream, because the device reads the images in place: streams that shared a buffer
    would be served from the device's caches after the first one instead of crossing PCIe, as the images of different
    cameras must."""

    def __init__(self, ctx, images, n_copies=1):
        self.ctx = ctx
        self.n = len(images)
        self.bytes = int(images[0].size)
        self.copies = int(n_copies)
        self.stride = self.bytes * self.n
        p = C.c_void_p()
        ctx._check(ctx.lib.svoh_host_alloc(ctx.h, C.c_size_t(self.stride * self.copies), C.byref(p)))
        self.ptr = p.value
        for k, im in enumerate(images):
            a = np.ascontiguousarray(im, dtype=np.uint8)
            assert a.size == self.bytes
            for c in range(self.copies):
                C.memmove(self.ptr + c * self.stride + k * self.bytes, a.ctypes.data, self.bytes)

    def address(self, k, copy=0):
        return self.ptr + copy * self.stride + k * self.bytes

    def free(self):
        if self.ptr:
            self.ctx.lib.svoh_host_free(self.ctx.h, C.c_void_p(self.ptr))
            self.ptr = None


class SharedPool(object):
    """Worker threads that several Lockstep engines draw on (host/svo_hip_pool.h: SharedPool)."""

    def __init__(self, n_workers, exclusive=False):
        """exclusive: ONE pool of n_workers threads (a calling group's thread counts as one) taken by the groups in turns, a phase at a
        time (ExclusivePool) instead of workers that serve several groups' phases side by side."""
        self.lib = load_host()
        h = C.c_void_p()
        if (self.lib.svohl_pool_create_exclusive if exclusive else self.lib.svohl_pool_create)(int(n_workers), C.byref(h)) != 0:
            raise RuntimeError(self.lib.svohl_last_error().decode())
        self.h = h

    def close(self):
        if self.h:
            self.lib.svohl_pool_destroy(self.h)
            self.h = None


class Lockstep(object):
    """One lock-step group of n_streams streams on the context `ctx` (frontend.Context).  pool: a SharedPool (the engine
    then has no threads of its own; seed = its first stream's index among all groups)."""

    def __init__(self, ctx, n_streams, cam, T_B_C7, params_yaml, depth_min, depth_mean, depth_max, kf_every=8, n_workers=1, images_pinned=True,
                 pool=None, seed=0, per_stream=None):
        """per_stream (round 6): a list of n_streams dicts(params_yaml, kf_every, min_tracked, depth=(min, mean, max)) -- streams that
        differ (svohl_create_streams); missing keys take the common arguments.  If any dict has "cam" (and optionally "T_B_C7"), every stream
        gets a camera of its own (svohl_create_streams_cameras: same image size; streams without the key take `cam`; stream 0's is the engine's)."""
        self.lib = load_host()
        self.ctx = ctx
        self.n = int(n_streams)
        h = C.c_void_p()
        c = fe._camera(cam)
        T = fe._se3(np.asarray(T_B_C7, dtype=np.float64))
        if per_stream is not None:
            assert len(per_stream) == self.n
            yamls = (C.c_char_p * self.n)(*[(d.get("params_yaml", params_yaml) or "").encode() or None for d in per_stream])
            depth = (C.c_double * (3 * self.n))(*[float(v) for d in per_stream for v in d.get("depth", (depth_min, depth_mean, depth_max))])
            kfe = (C.c_int * self.n)(*[int(d.get("kf_every", kf_every)) for d in per_stream])
            mtr = (C.c_int * self.n)(*[int(d.get("min_tracked", 60)) for d in per_stream])
            if any("cam" in d for d in per_stream):
                cams = (capi.svoh_camera * self.n)(*[fe._camera(d.get("cam", cam)) for d in per_stream])
                Ts = (capi.svoh_se3 * self.n)(*[fe._se3(np.asarray(d.get("T_B_C7", T_B_C7), dtype=np.float64)) for d in per_stream])
                rc = self.lib.svohl_create_streams_cameras(ctx.h, self.n, cams, Ts, yamls, depth, kfe, mtr, int(n_workers), pool.h if pool is not None else None,
                                                           int(seed), 1 if images_pinned else 0, C.byref(h))
            else:
                rc = self.lib.svohl_create_streams(ctx.h, self.n, C.byref(c), C.byref(T), yamls, depth, kfe, mtr, int(n_workers), pool.h if pool is not None else None,
                                                   int(seed), 1 if images_pinned else 0, C.byref(h))
        elif pool is not None:
            rc = self.lib.svohl_create_shared(ctx.h, self.n, C.byref(c), C.byref(T), params_yaml.encode() if params_yaml else None, float(depth_min),
                                              float(depth_mean), float(depth_max), int(kf_every), pool.h, int(seed), 1 if images_pinned else 0, C.byref(h))
        else:
            rc = self.lib.svohl_create(ctx.h, self.n, C.byref(c), C.byref(T), params_yaml.encode() if params_yaml else None, float(depth_min),
                                       float(depth_mean), float(depth_max), int(kf_every), int(n_workers), 1 if images_pinned else 0, C.byref(h))
        if rc != 0:
            raise fe.SvohError(rc, self.lib.svohl_last_error().decode())
        self.h = h
        self._ptrs = (C.c_void_p * self.n)()

    def _check(self, rc):
        if rc != 0:
            raise fe.SvohError(rc, self.lib.svohl_last_error().decode())

    def add_images(self, addresses, pitch, T_f_w_first=None):
        """addresses: n_streams host addresses (ints) of the streams' level-0 images."""
        for i, a in enumerate(addresses):
            self._ptrs[i] = a
        first = None
        if T_f_w_first is not None:
            arr = (capi.svoh_se3 * self.n)()
            for i, T in enumerate(T_f_w_first):
                arr[i] = fe._se3(T)
            first = C.cast(arr, C.c_void_p)
        self._check(self.lib.svohl_add_images(self.h, self._ptrs, int(pitch), first))

    def run_sequence(self, pinned, pitch, k_first, n_rounds, T_f_w_first=None):
        """n_rounds rounds inside ONE foreign call (the interpreter's lock is released throughout): every stream gets image
        frame_of(k) of the PinnedImages sequence, k = k_first ..; returns the rounds' stage times, (n_rounds, 7) ms."""
        first = None
        if T_f_w_first is not None:
            arr = (capi.svoh_se3 * self.n)()
            for i, T in enumerate(T_f_w_first):
                arr[i] = fe._se3(T)
            first = C.cast(arr, C.c_void_p)
        out = np.zeros((max(1, n_rounds), 7))
        assert pinned.copies >= self.n
        # (diagnostic: SVOH_LOCKSTEP_SHARED_IMAGES=1 lets every stream read the first copy -- the device then serves the images
        # from its caches, which says what the images' way over PCIe costs a run; never a number to report)
        stride = 0 if os.environ.get("SVOH_LOCKSTEP_SHARED_IMAGES") == "1" else pinned.stride
        self._check(self.lib.svohl_run_sequence(self.h, C.c_void_p(pinned.ptr), pinned.bytes, stride, pinned.n, int(pitch), int(k_first), int(n_rounds), first,
                                                out.ctypes.data))
        return out[:n_rounds]

    def run_schedule(self, pinned, pitch, k_first, n_rounds, schedule, T_f_w_first=None):
        """svohl_run_schedule: schedule = list of (start, step, every, phase) per stream; returns (stage times (n_rounds, 7) ms, frames taken)."""
        first = None
        if T_f_w_first is not None:
            arr = (capi.svoh_se3 * self.n)()
            for i, T in enumerate(T_f_w_first):
                arr[i] = fe._se3(T)
            first = C.cast(arr, C.c_void_p)
        out = np.zeros((max(1, n_rounds), 7))
        cols = [(C.c_int * self.n)(*[int(sc[j]) for sc in schedule]) for j in range(4)]
        done = C.c_long(0)
        assert pinned.copies >= self.n and len(schedule) == self.n
        self._check(self.lib.svohl_run_schedule(self.h, C.c_void_p(pinned.ptr), pinned.bytes, pinned.stride, pinned.n, int(pitch), int(k_first), int(n_rounds),
                                                cols[0], cols[1], cols[2], cols[3], first, out.ctypes.data, C.byref(done)))
        return out[:n_rounds], done.value

    def pose(self, s):
        T = capi.svoh_se3()
        self._check(self.lib.svohl_pose(self.h, int(s), C.byref(T)))
        return fe.se3_to_numpy(T)

    def last_round(self):
        t = (C.c_double * 7)()
        n = C.c_int()
        self._check(self.lib.svohl_last_round(self.h, t, C.byref(n)))
        return dict(zip(("pyramid", "align", "reproject", "pose", "seeds", "keyframe", "total"), list(t))), n.value

    def completed_rows(self, s, max_rows=4096):
        rows = (C.c_int64 * (7 * max_rows))()
        n = C.c_int()
        self._check(self.lib.svohl_completed_rows(self.h, int(s), max_rows, rows, C.byref(n)))
        return np.array(rows[:7 * n.value], dtype=np.int64).reshape(-1, 7)

    def finish(self):
        self._check(self.lib.svohl_finish(self.h))

    def phase_times(self):
        """{phase: ms summed over all rounds so far} of the group thread (FrontendLockstep::phaseTimes)."""
        ms = (C.c_double * 32)()
        n = C.c_int()
        self._check(self.lib.svohl_phase_times(self.h, 32, ms, C.byref(n)))
        return {self.lib.svohl_phase_name(k).decode(): ms[k] for k in range(n.value)}

    def close(self):
        if self.h:
            self.lib.svohl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LockstepStereo(object):
    """One lock-step group of n_streams STEREO streams on `ctx` (host/svo_hip_lockstep_stereo.h through its C face, svohs_*)."""
    PHASES = ("pyramids", "finish seeds", "align", "reproject", "pose", "structure", "keyframes", "seed updates")

    def __init__(self, ctx, n_streams, cams, T_B_C7s, params_yaml, kf_every=8, lambda_rot=0.5, n_workers=1, images_pinned=True):
        self.lib = load_host()
        self.ctx = ctx
        self.n = int(n_streams)
        cc = (capi.svoh_camera * 2)(fe._camera(cams[0]), fe._camera(cams[1]))
        TT = (capi.svoh_se3 * 2)(fe._se3(np.asarray(T_B_C7s[0], dtype=np.float64)), fe._se3(np.asarray(T_B_C7s[1], dtype=np.float64)))
        h = C.c_void_p()
        rc = self.lib.svohs_create(ctx.h, self.n, cc, TT, params_yaml.encode() if params_yaml else None, int(kf_every), float(lambda_rot), int(n_workers),
                                   1 if images_pinned else 0, C.byref(h))
        if rc != 0:
            raise fe.SvohError(rc, self.lib.svohs_last_error().decode())
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise fe.SvohError(rc, self.lib.svohs_last_error().decode())

    def run_sequence(self, pinned, pitch, k_first, n_rounds, T_imu_world_first=None, prior_forward=None):
        """pinned: PinnedImages of the interleaved pairs (left 0, right 0, left 1, ...), one copy per stream; returns the rounds' times (ms)."""
        first = None
        if T_imu_world_first is not None:
            arr = (capi.svoh_se3 * self.n)()
            for i, T in enumerate(T_imu_world_first):
                arr[i] = fe._se3(T)
            first = C.cast(arr, C.c_void_p)
        pr = None if prior_forward is None else np.ascontiguousarray(prior_forward, np.float64)
        out = np.zeros(max(1, n_rounds))
        assert pinned.copies >= self.n and pinned.n % 2 == 0
        self._check(self.lib.svohs_run_sequence(self.h, C.c_void_p(pinned.ptr), pinned.bytes, pinned.stride, pinned.n // 2, int(pitch), int(k_first), int(n_rounds), first,
                                                None if pr is None else pr.ctypes.data, out.ctypes.data))
        return out[:n_rounds]

    def pose(self, s):
        T = capi.svoh_se3()
        self._check(self.lib.svohs_pose(self.h, int(s), C.byref(T)))
        return fe.se3_to_numpy(T)

    def phase_times(self):
        ms = (C.c_double * 8)()
        self._check(self.lib.svohs_phase_times(self.h, ms))
        return dict(zip(self.PHASES, list(ms)))

    def finish(self):
        self._check(self.lib.svohs_finish(self.h))

    def close(self):
        if self.h:
            self.lib.svohs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
