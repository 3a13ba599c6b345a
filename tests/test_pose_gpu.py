"""GPU parity for the pose optimiser: svoh_optimize_pose_batch through the C ABI vs the
CPU oracle.  Bars: measurement sigma (a float median) exact, iteration counts / outlier flags / counters exact,
pose <= 1e-9 (fp64 sums in a different order), per-feature final errors rel 1e-9."""
import numpy as np
import pytest
import torch  # before the first svoh call of the process: one HIP runtime for both (see _capi._share_hip_runtime_with_torch)

from svo_pro_universal_amd import _capi as capi, frontend as fe, synth

import pose_helpers as ph

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["four_waves_per_bundle", "one_wave_per_bundle", "eight_waves_per_bundle"], autouse=True)
def pose_geometry(request, gpu_ctx):
    """The pose kernel has three geometries (pose.hip, pose_optimize_kernel<256> for batches up to one bundle per compute
    unit, <64> beyond, <512> for a few bundles of more than 256 features -- not instantiated for the bearing-vector
    error, which then runs with 256): every test of this file runs through all of them, against the same bars."""
    import os
    old = os.environ.get("SVOH_POSE_THREADS")
    os.environ["SVOH_POSE_THREADS"] = {"four_waves_per_bundle": "256", "one_wave_per_bundle": "64", "eight_waves_per_bundle": "512"}[request.param]
    gpu_ctx.reload_knobs()
    yield request.param
    if old is None:
        os.environ.pop("SVOH_POSE_THREADS", None)
        gpu_ctx.reload_knobs()
    else:
        os.environ["SVOH_POSE_THREADS"] = old
        gpu_ctx.reload_knobs()


def check(rg, ro, keep_g, keep_o, cams):
    assert rg.status == ro.status and rg.iters == ro.iters and rg.n_meas == ro.n_meas
    assert rg.measurement_sigma == ro.measurement_sigma and rg.reproj_error_before == ro.reproj_error_before
    assert np.abs(fe.se3_to_numpy(rg.T_imu_world) - fe.se3_to_numpy(ro.T_imu_world)).max() < 1e-9
    assert rg.n_deleted_edges == ro.n_deleted_edges and rg.n_deleted_corners == ro.n_deleted_corners
    assert rg.reproj_error_after == pytest.approx(ro.reproj_error_after, rel=1e-9)
    for kg, ko, c in zip(keep_g, keep_o, cams):
        n = len(c["level"])
        assert np.array_equal(kg["outlier"][:n], ko["outlier"][:n])
        assert np.allclose(kg["final_error"][:n], ko["final_error"][:n], rtol=1e-9, atol=1e-15)


@pytest.mark.parametrize("error_type", [capi.POSE_ERR_UNIT_PLANE, capi.POSE_ERR_BEARING_DIFF, capi.POSE_ERR_IMAGE_PLANE])
@pytest.mark.parametrize("n_cams", [1, 2])
def test_pose_parity(gpu_ctx, oracle_lib, error_type, n_cams):
    for seed, cam in ((31, synth.Camera.euroc_like(752, 480)), (32, synth.Camera.test_camera())):
        sc = ph.make_pose_scene(seed + n_cams, n=300, cam=cam, n_cams=n_cams)
        opt = capi.default_pose_options(sc["cam"], error_type=error_type)
        pbg, kg = fe.make_pose_problem(sc["cams"], sc["T_imu_world_init"])
        pbo, ko = fe.make_pose_problem(sc["cams"], sc["T_imu_world_init"])
        rg = gpu_ctx.optimize_pose(opt, [pbg])[0]
        ro = oracle_lib.optimize_pose(opt, pbo)
        check(rg, ro, kg, ko, sc["cams"])
        e1 = synth.se3_error(synth.SE3.from7(fe.se3_to_numpy(rg.T_imu_world)), sc["T_imu_world_gt"])
        e0 = synth.se3_error(sc["T_imu_world_init"], sc["T_imu_world_gt"])
        assert e1[0] < 0.1 * e0[0] and e1[1] < 0.1 * e0[1]


def test_pose_prior_iteration_cap_and_degenerate(gpu_ctx, oracle_lib):
    sc = ph.make_pose_scene(41, n=120)
    q_init = sc["T_imu_world_init"].as7()[:4]
    for kw in (dict(have_rotation_prior=1, prior_lambda=50.0, R_prior=q_init), dict(max_iter=2), dict(eps=1e-3),
               dict(outlier_threshold=1e-9), dict(outlier_threshold=1e9)):
        opt = capi.default_pose_options(sc["cam"], **kw)
        pbg, kg = fe.make_pose_problem(sc["cams"], sc["T_imu_world_init"])
        pbo, ko = fe.make_pose_problem(sc["cams"], sc["T_imu_world_init"])
        check(gpu_ctx.optimize_pose(opt, [pbg])[0], oracle_lib.optimize_pose(opt, pbo), kg, ko, sc["cams"])
    # nothing usable / no feature at all / a single measurement (rank-deficient system: the solver stops)
    opt = capi.default_pose_options(sc["cam"])
    for mode in ("none_usable", "empty", "single"):
        cams = [dict(c) for c in sc["cams"]]
        for c in cams:
            c["usable"] = c["usable"].copy()
            if mode == "none_usable":
                c["usable"][:] = 0
            elif mode == "single":
                c["usable"][:] = 0; c["usable"][5] = 1
            else:
                for k in ("px", "f", "grad", "xyz_world"):
                    c[k] = c[k][:0]
                c["level"] = c["level"][:0]; c["type"] = c["type"][:0]; c["usable"] = c["usable"][:0]
        pbg, kg = fe.make_pose_problem(cams, sc["T_imu_world_init"])
        pbo, ko = fe.make_pose_problem(cams, sc["T_imu_world_init"])
        rg, ro = gpu_ctx.optimize_pose(opt, [pbg])[0], oracle_lib.optimize_pose(opt, pbo)
        assert rg.status == ro.status and rg.n_meas == ro.n_meas
        if mode != "single":
            assert rg.status == 1 and np.array_equal(fe.se3_to_numpy(rg.T_imu_world), sc["T_imu_world_init"].as7())


def test_pose_batch_equals_singles(gpu_ctx):
    scenes = [ph.make_pose_scene(50 + i, n=40 + 37 * i, n_cams=1 + (i % 2)) for i in range(9)]
    opt = capi.default_pose_options(scenes[0]["cam"])
    built = [fe.make_pose_problem(sc["cams"], sc["T_imu_world_init"]) for sc in scenes]
    batch = gpu_ctx.optimize_pose(opt, [b[0] for b in built])
    flags = [[k["outlier"].copy() for k in b[1]] for b in built]
    for i, sc in enumerate(scenes):
        pb, keep = fe.make_pose_problem(sc["cams"], sc["T_imu_world_init"])
        r = gpu_ctx.optimize_pose(opt, [pb])[0]
        assert np.array_equal(fe.se3_to_numpy(r.T_imu_world), fe.se3_to_numpy(batch[i].T_imu_world))
        assert r.iters == batch[i].iters and r.n_deleted_corners == batch[i].n_deleted_corners
        assert all(np.array_equal(a, k["outlier"]) for a, k in zip(flags[i], keep))
    with pytest.raises(fe.SvohError):
        gpu_ctx.optimize_pose(capi.default_pose_options(scenes[0]["cam"], error_type=7), [built[0][0]])


def test_pose_large_batch_staged_by_threads(gpu_ctx):
    """A batch above 65 536 features is staged into the pinned block by several host threads (pose.hip,
    kPoseParallelStagingFeatures): every bundle, including its per-feature outputs, must come out as when it is
    submitted alone.  Ragged sizes and one- / two-camera bundles, so that a wrong offset cannot cancel."""
    scenes = [ph.make_pose_scene(700 + i, n=120 + 41 * i, n_cams=1 + (i % 2)) for i in range(6)]
    opt = capi.default_pose_options(scenes[0]["cam"])
    singles = []
    for sc in scenes:
        pb, keep = fe.make_pose_problem(sc["cams"], sc["T_imu_world_init"])
        r = gpu_ctx.optimize_pose(opt, [pb])[0]
        singles.append((fe.se3_to_numpy(r.T_imu_world).copy(), r.iters, r.n_meas,
                        [k["outlier"].copy() for k in keep], [k["final_error"].copy() for k in keep]))
    n_feat = [sum(len(c["level"]) for c in sc["cams"]) for sc in scenes]
    B = 66000 // min(n_feat) + 6
    built = [fe.make_pose_problem(scenes[i % 6]["cams"], scenes[i % 6]["T_imu_world_init"]) for i in range(B)]
    assert sum(n_feat[i % 6] for i in range(B)) >= 65536
    batch = gpu_ctx.optimize_pose(opt, [b[0] for b in built])
    for i in range(B):
        T, iters, n_meas, out, ferr = singles[i % 6]
        assert batch[i].iters == iters and batch[i].n_meas == n_meas
        # a batch this large runs one wave per bundle unless the fixture forces four: summation order may differ
        assert np.abs(fe.se3_to_numpy(batch[i].T_imu_world) - T).max() < 1e-9
        for k, o, e in zip(built[i][1], out, ferr):
            assert np.array_equal(k["outlier"], o)
            assert np.allclose(k["final_error"], e, rtol=1e-9, atol=1e-15)


def test_packed_device_arrays_equal_the_host_staged_call(gpu_ctx):
    """svoh_optimize_pose_batch_packed: the per-feature arrays of all bundles concatenated on the device and used in
    place (nothing but descriptors and per-bundle results crosses PCIe) -- bit-identical to the host-staged call."""
    import ctypes as C
    scs = [ph.make_pose_scene(60 + k, n=60 + 37 * k, n_cams=1 + k % 2) for k in range(5)]
    opt = capi.default_pose_options(scs[0]["cam"])
    built = [fe.make_pose_problem(sc["cams"], sc["T_imu_world_init"]) for sc in scs]
    want = gpu_ctx.optimize_pose(opt, [b[0] for b in built])
    cat = {k: [] for k in ("px", "f", "grad", "level", "type", "xyz_world", "usable")}
    for pb, keep in built:
        for a in keep:
            for k in cat:
                cat[k].append(a[k].ravel())
    dev = torch.device("cuda", 0)
    t = {k: torch.from_numpy(np.concatenate(v)).to(dev) for k, v in cat.items()}
    n_total = t["level"].numel()
    t["outlier"] = torch.zeros(n_total, dtype=torch.uint8, device=dev)
    t["final_error"] = torch.zeros(n_total, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    arr = capi.svoh_pose_packed_arrays()
    arr.n_features_total = n_total
    for k in ("px", "f", "grad", "level", "type", "xyz_world", "usable", "outlier", "final_error"):
        setattr(arr, k, t[k].data_ptr())
    pbs = (capi.svoh_pose_problem * len(built))(*[b[0] for b in built])
    res = (capi.svoh_pose_result * len(built))()
    gpu_ctx._check(gpu_ctx.lib.svoh_optimize_pose_batch_packed(gpu_ctx.h, C.byref(opt), len(built), pbs, C.byref(arr), res))
    out_g, err_g = t["outlier"].cpu().numpy(), t["final_error"].cpu().numpy()
    off = 0
    for r, w, (pb, keep) in zip(res, want, built):
        assert r.status == w.status and r.iters == w.iters and r.n_meas == w.n_meas
        assert np.array_equal(fe.se3_to_numpy(r.T_imu_world), fe.se3_to_numpy(w.T_imu_world))
        assert r.measurement_sigma == w.measurement_sigma and r.reproj_error_after == w.reproj_error_after
        for a in keep:
            n = a["level"].size
            assert np.array_equal(out_g[off:off + n], a["outlier"][:n]) and np.array_equal(err_g[off:off + n], a["final_error"][:n])
            off += n
    assert off == n_total
    arr.n_features_total = n_total + 1          # a size that does not match the problems is refused
    assert gpu_ctx.lib.svoh_optimize_pose_batch_packed(gpu_ctx.h, C.byref(opt), len(built), pbs, C.byref(arr), res) != 0
