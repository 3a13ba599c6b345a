#!/usr/bin/env python3
"""bench.py -- aligned patches/s of the MI355X-native SVO sparse image alignment.

Workload (BASELINE.json configs[1], SURVEY.md 8(d) "C2"): SparseImgAlign only,
synthetic 640x480 frames, 2000 patches x 4x4, 5 pyramid levels (4..0), SE3 6-DoF,
<= 10 Gauss-Newton iterations per level, eps 5e-4.  One "step" = one call of
svoh_sparse_align_batch() over B independent (reference, current) frame pairs
whose pyramids and feature arrays are already resident in HBM (B defaults to
1024: ~0.9 GB of pyramids, far beyond the 256 MiB Infinity Cache).

Multi-GPU: the path shards by independent frame pairs; every rank aligns its own
B pairs, no data-path collective; scaling is "weak".

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch  # first: libsvo_hip must share torch's HIP runtime (same SONAME)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from svo_pro_universal_amd import _capi as capi  # noqa: E402
from svo_pro_universal_amd import dist_utils as du  # noqa: E402
from svo_pro_universal_amd import frontend as fe  # noqa: E402
from svo_pro_universal_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(P, D, patch_iters, n_sel_levels):
    """B_iter(P,D) per patch-iteration + per-level precompute per patch."""
    b_iter = (P + 1) ** 2 + 12 + 4 * P * P + 4 * D * P * P
    b_pre = ((P + 3) ** 2 + 8 + 48) + 4 * P * P * (1 + D)
    return b_iter * patch_iters + b_pre * n_sel_levels


FP64_PEAK_TFLOPS = 63.8   # measured: tools/svoh_microbench, independent v_fma_f64 on every SIMD (profiles/r02_fetch_calibration.json)
FP64_VENDOR_PEAK_TFLOPS = 78.6   # the vendor's vector fp64 figure for the part (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz)


def pmc_summary(workload_key):
    """The committed rocprofv3 counter summary of this very workload (profiles/rNN_<tag>_pmc_summary.json, written by
    scripts/profile_round.sh + scripts/pmc_summary.py from separate --pmc FETCH_SIZE / WRITE_SIZE / SQ passes), newest
    round first; None when there is none -- the counters cannot be read from inside the timed process."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload_key") == workload_key and d.get("traffic_bytes_per_step"):
            d["_file"] = os.path.basename(f)
            best = d
    return best


def roofline(kernel, kernel_ms, alg_bytes, workload_key, **extra):
    """roofline object of the bench line.  `achieved` / `frac` are PHYSICAL: HBM-side bytes per step from the PMC
    passes of this workload (2 x FETCH_SIZE + WRITE_SIZE, see scripts/pmc_summary.py) over the kernel time measured
    live, against the 8 TB/s peak.  The SURVEY 8(d) algorithmic figure (bytes the cache-everything formulation would
    move) is kept beside it as algorithmic_*: it exceeds 1 for kernels that recompute instead of re-reading."""
    t = kernel_ms * 1e-3
    alg = alg_bytes / t / 1e9
    summ = pmc_summary(workload_key)
    r = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": kernel,
         "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_achieved": alg, "algorithmic_frac": alg / HBM_PEAK_GBS}
    if summ:
        traffic = float(summ["traffic_bytes_per_step"])
        prof_ms = summ.get("kernel_ms_per_step_rocprof") or (summ.get("bench_under_rocprof") or {}).get("kernel_ms")
        # `frac` / `achieved`: the committed profile's traffic over the committed profile's OWN kernel time -- one box, one
        # run, reproducible from profiles/ alone.  `frac_live` / `achieved_live`: the same traffic over the kernel time measured
        # in THIS run (the counters cannot be read inside the timed process; kernel times move a few per cent box to box).
        t_ref = (prof_ms or kernel_ms) * 1e-3
        r.update({"achieved": traffic / t_ref / 1e9, "frac": traffic / t_ref / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                  "frac_basis": "HBM-side traffic (PMC) / the kernel time of the same profile run" if prof_ms else "HBM-side traffic (PMC) / live kernel time",
                  "achieved_live": traffic / t / 1e9, "frac_live": traffic / t / 1e9 / HBM_PEAK_GBS,
                  "traffic_source": "profiles/" + summ["_file"], "traffic_live": False, "kernel_ms_live": kernel_ms})
        if prof_ms:
            r["kernel_ms_when_profiled"] = prof_ms
            r["traffic_stale"] = bool(abs(prof_ms - kernel_ms) > 0.25 * kernel_ms)   # kernel changed since the PMC passes?
        cs = summ.get("compute_side")
        if cs:
            r["compute_side"] = {k: cs[k] for k in cs if k != "counters_mean_per_dispatch"}
            c = cs.get("counters_mean_per_dispatch", {})
            if "SQ_INSTS_VALU" in c:
                r["compute_side"]["valu_wave_instructions_per_launch"] = c["SQ_INSTS_VALU"]
            # fp64 arithmetic against BOTH peaks: the one measured on these boxes with independent v_fma_f64 on every
            # SIMD (tools/svoh_microbench: 63.8 TFLOP/s) and the vendor figure (78.6); the flop count is the profiled
            # launch's, the time the live one
            flop = cs.get("fp64_flop_per_dispatch_upper_bound")
            if flop:
                tf = flop / t / 1e12
                r["compute_side"]["fp64_tflops_live"] = tf
                r["compute_side"]["fp64_fraction_of_measured_peak_63p8_live"] = tf / FP64_PEAK_TFLOPS
                r["compute_side"]["fp64_fraction_of_vendor_peak_78p6_live"] = tf / FP64_VENDOR_PEAK_TFLOPS
            # which resource is nearest its ceiling (`what_binds`; `bound` stays the roofline of SURVEY 8(d) that `peak` / `frac` refer to, "hbm"):
            # the three candidates with their fractions
            cand = {"hbm": r["frac_live"]}
            if flop:
                cand["fp64 VALU issue"] = r["compute_side"]["fp64_fraction_of_measured_peak_63p8_live"]
            busy, waiting = cs.get("valu_busy_fraction_of_simd_time"), cs.get("SQ_WAIT_ANY_share_of_wave_cycles")
            r["ceilings"] = dict(cand, valu_busy=busy, waves_waiting_share=waiting)
            top = max(cand, key=lambda k: cand[k])
            if busy is not None and waiting is not None and waiting >= 0.40 and busy < 0.60:
                # no throughput resource past 60 %: the waves spend their time waiting on their own dependent chains
                r["what_binds"] = "latency (waves wait %.0f %% of their cycles; VALU busy %.0f %%, %s %.2f)" % (
                    100 * waiting, 100 * busy, top, cand[top])
            else:
                r["what_binds"] = "%s (%.2f of its ceiling%s)" % (top, cand[top], "; VALU busy %.0f %%" % (100 * busy) if busy is not None else "")
    else:
        r.update({"achieved": alg, "frac": alg / HBM_PEAK_GBS, "traffic": None,
                  "frac_basis": "algorithmic bytes (no PMC summary committed for this workload key)"})
    pm = measured_copy_bandwidth()
    if pm:   # the measured copy bandwidth next to the vendor figure, and the fraction against it
        r["peak_measured"] = pm["copy_GBs"]
        r["peak_measured_how"] = pm["how"]
        r["frac_of_peak_measured"] = r["achieved"] / pm["copy_GBs"]
    r.update(extra)
    return r


_COPY_BW = None


def measured_copy_bandwidth():
    """Device-to-device copy of 1 GiB (read + write = 2 GiB of HBM traffic) on this box, best of five, timed with events on
    the copy's own stream: the measured HBM bandwidth to quote next to the 8 TB/s vendor peak.  ~10 ms, once per run,
    outside every timed region.  None when no GPU."""
    global _COPY_BW
    if _COPY_BW is not None:
        return _COPY_BW or None
    try:
        if not torch.cuda.is_available():
            _COPY_BW = {}
            return None
        n = 1 << 30
        a = torch.empty(n, dtype=torch.uint8, device="cuda").fill_(3)
        b = torch.empty_like(a)
        best = None
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            b.copy_(a)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None or ms < best else best
        del a, b
        torch.cuda.empty_cache()
        _COPY_BW = {"copy_GBs": 2.0 * n / (best * 1e-3) / 1e9, "how": "torch D2D copy_ of 1 GiB, (read + write) bytes / best of 6, this run"}
    except Exception:   # noqa: BLE001 -- a diagnostic, never a reason to fail the bench
        _COPY_BW = {}
    return _COPY_BW or None


def build_problems(ctx, dev, rank, B, N, P, max_level, reuse=None):
    """B synthetic frame pairs rendered on the GPU; returns problems + keepalives.  reuse = (imgs, frames) of an
    earlier call with the same B / seeds: the scenes' images do not depend on the patch size, only the features do."""
    cam = synth.Camera.test_camera()
    scenes = [synth.make_align_scene(du.problem_seed(rank, i), n_features=N, patch_size=P, cam=cam,
                                     max_level=max_level, render_images=False) for i in range(B)]
    if reuse is not None:
        imgs, frames = reuse
    else:
        poses, planes, texs = [], [], []
        for sc in scenes:  # image 2i = reference frame, 2i+1 = current frame of pair i
            poses += [sc.T_w_ref, sc.T_w_cur]
            planes += [sc.plane, sc.plane]
            texs += [sc.tex, sc.tex]
        imgs = synth.render_batch_torch(cam, poses, planes, texs, dev)
        torch.cuda.synchronize()
        frames = ctx.build_pyramid_batch_device(imgs.data_ptr(), cam.width * cam.height, 2 * B, cam.width, cam.height,
                                                cam.width, max_level + 1)
        ctx.synchronize()
    # feature arrays resident in HBM (one buffer per kind)
    px = torch.from_numpy(np.concatenate([s.px for s in scenes])).to(dev)
    f = torch.from_numpy(np.concatenate([s.f for s in scenes])).to(dev)
    pw = torch.from_numpy(np.concatenate([s.pos_world for s in scenes])).to(dev)
    fl = torch.from_numpy(np.concatenate([s.flags for s in scenes])).to(dev)
    items = []
    off = 0
    for i, sc in enumerate(scenes):
        n = sc.n_features
        dp = dict(px=px.data_ptr() + 16 * off, f=f.data_ptr() + 24 * off, pos_world=pw.data_ptr() + 24 * off,
                  flags=fl.data_ptr() + off)
        items.append([(sc, frames[2 * i], frames[2 * i + 1], dp)])
        off += n
    problems, keep = fe.make_align_problems(items)
    return problems, scenes, imgs, (px, f, pw, fl, keep, frames)


def cpu_baseline(scenes, imgs, opt, max_level, budget_s=15.0):
    """Time the CPU restatement (oracle, -O3 -march=native build, 1 thread) on a
    bounded sample of the same problems: kind "port"."""
    from oracle import oracle as orc  # test infrastructure: used only as the timed CPU baseline
    orc.build(fast=True)
    done_patches, n_done, t_total = 0, 0, 0.0
    for i, sc in enumerate(scenes):
        ref = orc.create_img_pyramid(imgs[2 * i].cpu().numpy(), max_level + 1, fast=True)
        cur = orc.create_img_pyramid(imgs[2 * i + 1].cpu().numpy(), max_level + 1, fast=True)
        pb = orc.problem_from_scenes([(sc, ref, cur)])
        t0 = time.perf_counter()
        n, res, _ = orc.sparse_align_run(opt, pb, fast=True)
        t_total += time.perf_counter() - t0
        done_patches += n
        n_done += 1
        if t_total > budget_s:
            break
    base = {"value": done_patches / t_total, "unit": "aligned patches/s", "cores": 1, "kind": "port",
            "sample": "%d of the benchmark's frame pairs (%d patches), SparseImgAlign::run restatement "
                      "(oracle/svo_oracle.c, gcc -O3 -march=native, fp64, single thread), %.1f s of CPU time"
                      % (n_done, done_patches, t_total),
            "ms_per_frame": 1e3 * t_total / n_done}
    # The same port with one frame pair per thread (the reference's img-align itself is single-threaded, so this is the
    # multi-stream CPU alternative, not the reference), twice: on the host-core SHARE of one GPU (16 threads on the
    # benchmark boxes: gpu_host_share) and on every CPU the process may run on (all_cpus).  On a box that hands out a CPU
    # *quota* rather than a CPU set (cgroup cpu.max), the second leg cannot use more CPU time than the quota whatever its
    # thread count: the quota is printed beside it, and whole_host_extrapolated scales the measured per-thread rate of
    # the share leg to the machine's physical cores -- an estimate of what the port would do with the whole host.
    def parallel_leg(cores):
        from concurrent.futures import ThreadPoolExecutor
        n_par = min(len(scenes), max(cores * 2, n_done))
        prepared = []
        for i in range(n_par):
            ref = orc.create_img_pyramid(imgs[2 * i].cpu().numpy(), max_level + 1, fast=True)
            cur = orc.create_img_pyramid(imgs[2 * i + 1].cpu().numpy(), max_level + 1, fast=True)
            prepared.append((orc.problem_from_scenes([(scenes[i], ref, cur)]), ref, cur))
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:
            counts = list(ex.map(lambda p: orc.sparse_align_run(opt, p[0], fast=True)[0], prepared))
        t_par = time.perf_counter() - t0
        return {"value": sum(counts) / t_par, "unit": "aligned patches/s", "cores": cores,
                "sample": "%d frame pairs, one per thread, %.1f s wall" % (n_par, t_par)}
    n_avail = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        share = min(n_avail, int(os.environ.get("SVOH_BENCH_CPU_THREADS", "16")))
        base["gpu_host_share"] = parallel_leg(share)
        base["all_cores"] = dict(base["gpu_host_share"], note="kept under its old name: the 16-thread host share of one GPU, NOT the whole host (see all_cpus)")
        if n_avail > share:
            base["all_cpus"] = parallel_leg(n_avail)
            base["all_cpus"]["cpu_quota_of_this_container"] = quota
        else:
            base["all_cpus"] = dict(base["gpu_host_share"], note="the process may run on no more CPUs than the share leg used")
        phys = None
        try:
            cores_seen = set()
            for c in range(os.cpu_count() or 0):
                base_dir = "/sys/devices/system/cpu/cpu%d/topology/" % c
                cores_seen.add((open(base_dir + "physical_package_id").read().strip(), open(base_dir + "core_id").read().strip()))
            phys = len(cores_seen) or None
        except Exception:
            pass
        if phys:
            per_thread = base["gpu_host_share"]["value"] / base["gpu_host_share"]["cores"]
            base["whole_host_extrapolated"] = {"value": per_thread * phys, "unit": "aligned patches/s", "physical_cores": phys,
                                               "note": "per-thread rate of the gpu_host_share leg x physical cores of the machine: an estimate, not a measurement"}
    except Exception as e:  # the baseline leg must not take the benchmark down
        base["gpu_host_share"] = {"error": str(e)}
    try:   # the host the baseline ran on (core count and CPU model stated)
        model = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown")
        base["host"] = {"cpu_model": model, "logical_cpus": os.cpu_count(), "cpus_available_to_this_process": len(os.sched_getaffinity(0))}
    except Exception:
        pass
    return base


def timed_steps(ctx, dist, world, dev, step_fn, steps, warmup):
    """W untimed + K timed steps between barriers; returns (elapsed_s, mean kernel ms, last result)."""
    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    out = None
    for _ in range(warmup):
        out, _ = step_fn()
    barrier()
    ksum = 0.0
    t0 = time.perf_counter()
    have_kms = True
    for _ in range(steps):
        out, kms = step_fn()
        if kms is None:
            have_kms = False
        else:
            ksum += kms
    barrier()
    elapsed = time.perf_counter() - t0
    # stream-ordered (device-resident) steps do not wait for their kernel: read the last launch's events afterwards
    return elapsed, (ksum / steps if have_kms else misc_kernel_ms(ctx)), out


def misc_kernel_ms(ctx):
    ms = ctypes.c_float()
    ctx.lib.svoh_last_kernel_ms(ctx.h, ctypes.byref(ms))
    return ms.value


def misc_counters(ctx):
    c = (ctypes.c_uint64 * 8)()
    ctx.lib.svoh_last_kernel_counters(ctx.h, c)
    return [int(x) for x in c]


def render_pairs(ctx, dev, rank, B, max_level, **scene_kw):
    cam = synth.Camera.test_camera()
    scenes = [synth.make_align_scene(du.problem_seed(rank, i), n_features=8, cam=cam, max_level=max_level,
                                     render_images=False, **scene_kw) for i in range(B)]
    poses, planes, texs = [], [], []
    for sc in scenes:
        poses += [sc.T_w_ref, sc.T_w_cur]
        planes += [sc.plane, sc.plane]
        texs += [sc.tex, sc.tex]
    imgs = synth.render_batch_torch(cam, poses, planes, texs, dev)
    torch.cuda.synchronize()
    frames = ctx.build_pyramid_batch_device(imgs.data_ptr(), cam.width * cam.height, 2 * B, cam.width, cam.height,
                                            cam.width, max_level + 1)
    ctx.synchronize()
    return cam, scenes, imgs, frames


def bench_klt(args, ctx, dist, rank, world, dev, comm_dev=None):
    """KLT-synth: 400 tracks per frame pair, patches {16,16,16,8,8}, <=30 it, B pairs per step."""
    B = args.problems or 256
    NT = 400
    cam, scenes, imgs, frames = render_pairs(ctx, dev, rank, B, 4, rot_deg=(0.5, 1.5), trans_m=(0.05, 0.15))
    opt = capi.default_klt_options()
    tracks = [synth.make_track_set(sc, NT, seed=i) for i, sc in enumerate(scenes)]
    px_ref = np.concatenate([t["px_ref"] for t in tracks]); px0 = np.concatenate([t["px_cur_init"] for t in tracks])
    n = B * NT
    rf = (capi.svoh_frame_t * n)(*[frames[2 * (i // NT)] for i in range(n)])
    cf = (capi.svoh_frame_t * n)(*[frames[2 * (i // NT) + 1] for i in range(n)])
    status = np.zeros(n, np.uint8)

    def step_host():
        out = px0.copy()
        ctx._check(ctx.lib.svoh_klt_track_multi(ctx.h, ctypes.byref(opt), n, rf, cf, px_ref.ctypes.data, out.ctypes.data,
                                                status.ctypes.data))
        return (out, status.copy()), misc_kernel_ms(ctx)

    # value: per-track arrays resident in HBM, used in place through the frame-table entry
    ext = torch.cuda.ExternalStream(ctx.stream(), device=dev)
    t_ridx = torch.from_numpy(np.repeat(2 * np.arange(B, dtype=np.int32), NT)).to(dev)
    t_cidx = t_ridx + 1
    t_pxr = torch.from_numpy(px_ref.astype(np.int32)).to(dev)
    t_px0 = torch.from_numpy(px0).to(dev)
    t_px = t_px0.clone()
    t_st = torch.zeros(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step():
        with torch.cuda.stream(ext):
            t_px.copy_(t_px0)
        ctx.klt_track_indexed(opt, list(frames), n, t_ridx.data_ptr(), t_cidx.data_ptr(), t_pxr.data_ptr(),
                              t_px.data_ptr(), t_st.data_ptr())
        return None, None

    h_elapsed, _hk, (out, st) = timed_steps(ctx, dist, world, dev, step_host, max(2, args.steps // 4), 1)
    host_rate = n * max(2, args.steps // 4) / h_elapsed
    elapsed, kms, _ = timed_steps(ctx, dist, world, dev, step, args.steps, args.warmup)
    assert np.array_equal(t_px.cpu().numpy(), out) and np.array_equal(t_st.cpu().numpy(), st)
    cnt = misc_counters(ctx)
    # per track-iteration (P+1)^2 + P^2 + 4 P^2 bytes; template build (P+2)^2 read per level
    alg = cnt[0] * (17 * 17 + 5 * 256) + cnt[1] * (9 * 9 + 5 * 64) + cnt[2] * 18 * 18 + cnt[3] * 10 * 10
    elapsed, total = du.combine(dist, world, elapsed, n, comm_dev)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        orc.build(fast=True)
        t_cpu, n_cpu = 0.0, 0
        for i, sc in enumerate(scenes):
            ref = orc.create_img_pyramid(imgs[2 * i].cpu().numpy(), 5, fast=True)
            cur = orc.create_img_pyramid(imgs[2 * i + 1].cpu().numpy(), 5, fast=True)
            t0 = time.perf_counter()
            po, so = orc.klt_track_batch(opt, ref, cur, tracks[i]["px_ref"], tracks[i]["px_cur_init"], fast=True)
            t_cpu += time.perf_counter() - t0
            n_cpu += NT
            assert np.array_equal(so, st[i * NT:(i + 1) * NT]) and np.array_equal(po, out[2 * i * NT:2 * (i + 1) * NT])
            if t_cpu > 10.0:
                break
        cpu = {"value": n_cpu / t_cpu, "unit": "tracks/s", "cores": 1, "kind": "port",
               "sample": "%d tracks of the benchmark (oracle alignPyr2D, gcc -O3 -march=native, 1 thread, %.1f s); "
                         "GPU results bit-identical on the sample" % (n_cpu, t_cpu)}
    if rank != 0:
        return None
    ok = st == 1
    return {"metric": "KLT tracks/s (alignPyr2D, 400 tracks/frame, patches {16,16,16,8,8}, <=30 it)",
            "value": total * args.steps / elapsed, "unit": "tracks/s", "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_frame": 1e3 * elapsed / args.steps / B, "dtype": "i32+f32",
            "config": {"workload": "KLT-synth: %d frame pairs x %d tracks per GPU per step, 640x480, levels 4..0, track arrays resident in HBM" % (B, NT),
                       "frame_pairs_per_gpu": B, "tracks_per_frame": NT},
            "kernel_ms": kms, "converged_fraction": float(ok.mean()),
            "host_staged_tracks_per_s": host_rate,  # same work through svoh_klt_track_multi with host arrays (PCIe-inclusive)
            "roofline": roofline("klt_track_kernel", kms, alg, "klt:default" if not args.problems else "klt:B%d" % B, counters=cnt[:4]),
            "cpu_baseline": cpu}


def bench_seeds(args, ctx, dist, rank, world, dev, comm_dev=None):
    """C4-synth: 3000 seeds per keyframe, 8x8 patches, epipolar ZMSSD scan (<=100 steps) +
    align1D/2D + Vogiatzis update; B (keyframe, frame) pairs per step."""
    B = args.problems or 64
    NS = 3000
    cam, scenes, imgs, frames = render_pairs(ctx, dev, rank, B, 4, rot_deg=(0.3, 1.0), trans_m=(0.05, 0.15))
    mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(cam)
    seeds = [synth.make_seed_set(sc, NS, seed=i) for i, sc in enumerate(scenes)]
    ref_views = [fe.make_frame_view(frames[2 * i], cam, sc.T_ref_f_w, seeds[i]["mu_range"], 2 * i) for i, sc in enumerate(scenes)]
    cur_views = (capi.svoh_frame_view * B)(*[fe.make_frame_view(frames[2 * i + 1], cam, sc.T_cur_f_w_gt, 0.0, 2 * i + 1)
                                             for i, sc in enumerate(scenes)])
    idx = np.repeat(np.arange(B, dtype=np.int32), NS)
    cat = lambda k: np.concatenate([s[k] for s in seeds])
    fb, keep = fe.make_feature_batch(idx, cat("px"), cat("f"), cat("grad"), cat("level"), cat("type"))
    fb.cur_frame_idx = idx.ctypes.data
    fb.n_cur_frames = B
    state0 = cat("state")
    type0 = keep["type"].copy()
    n = B * NS
    rv = (capi.svoh_frame_view * B)(*ref_views)
    success = np.zeros(n, np.uint8); mr = np.zeros(n, np.int32); ns = ctypes.c_int32()

    def step_host():
        st = state0.copy()
        keep["type"][:] = type0
        ctx._check(ctx.lib.svoh_update_seeds_batch(ctx.h, ctypes.byref(mopt), ctypes.byref(dopt), B, rv, cur_views,
                                                   ctypes.byref(fb), st.ctypes.data, success.ctypes.data, mr.ctypes.data,
                                                   ctypes.byref(ns)))
        return (st, success.copy(), mr.copy()), misc_kernel_ms(ctx)

    # value: seed arrays resident in HBM, updated in place (mem_space = SVOH_MEM_DEVICE)
    ext = torch.cuda.ExternalStream(ctx.stream(), device=dev)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in
         dict(idx=idx, px=cat("px"), f=cat("f"), grad=cat("grad"), level=cat("level").astype(np.int32),
              type0=type0, state0=state0).items()}
    t["type"] = t["type0"].clone(); t["state"] = t["state0"].clone()
    t["succ"] = torch.zeros(n, dtype=torch.uint8, device=dev); t["mr"] = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    fbd = fe.make_feature_batch_device(n, t["idx"].data_ptr(), t["px"].data_ptr(), t["f"].data_ptr(), t["grad"].data_ptr(),
                                       t["level"].data_ptr(), t["type"].data_ptr(), cur_frame_idx=t["idx"].data_ptr(),
                                       n_cur_frames=B)
    cur_list = [cur_views[i] for i in range(B)]

    def step_units():
        with torch.cuda.stream(ext):
            t["state"].copy_(t["state0"]); t["type"].copy_(t["type0"])
        ctx.update_seeds_device(mopt, dopt, ref_views, cur_list, fbd, t["state"].data_ptr(), t["succ"].data_ptr(),
                                t["mr"].data_ptr())
        return None, None

    # the keyframes' constant columns resident on the device (svoh_features_upload, once per keyframe's life); the batch = every seed
    # of every keyframe (SVOH_BATCH_WHOLE_SETS): processed in the tile order computed at upload, no count / scan / scatter / un-sort
    m = len(seeds)
    cols = [[np.ascontiguousarray(sd[k], dt) for sd in seeds] for k, dt in (("px", np.float64), ("f", np.float64), ("grad", np.float64), ("level", np.int32))]
    handles = (ctypes.c_uint64 * m)()
    ctx._check(ctx.lib.svoh_features_upload(ctx.h, m, (ctypes.c_int32 * m)(*[a.size for a in cols[3]]), *[(ctypes.c_void_p * m)(*[a.ctypes.data for a in c]) for c in cols], handles))
    ref_views_ws = [fe.make_frame_view(frames[2 * i], cam, sc.T_ref_f_w, seeds[i]["mu_range"], 2 * i) for i, sc in enumerate(scenes)]
    for i in range(m):
        ref_views_ws[i].features = handles[i]
    fbw = fe.make_feature_batch_device(n, None, None, None, None, None, t["type"].data_ptr(), cur_frame_idx=t["idx"].data_ptr(), n_cur_frames=B)
    fbw.layout = capi.SVOH_BATCH_WHOLE_SETS

    def step_whole_sets():
        with torch.cuda.stream(ext):
            t["state"].copy_(t["state0"]); t["type"].copy_(t["type0"])
        ctx.update_seeds_device(mopt, dopt, ref_views_ws, cur_list, fbw, t["state"].data_ptr(), t["succ"].data_ptr(), t["mr"].data_ptr())
        return None, None
    step, other = (step_whole_sets, step_units) if args.whole_sets else (step_units, step_whole_sets)

    h_steps = max(2, args.steps // 4)
    h_elapsed, _hk, (st, succ, mres) = timed_steps(ctx, dist, world, dev, step_host, 1 if args.no_secondary else h_steps, 0 if args.no_secondary else 1)
    host_rate = None if args.no_secondary else n * h_steps / h_elapsed   # (a profiling run keeps ONE call of this leg: it is what the timed leg's results are checked against)
    # the leg that is not the timed one: a few steps, for its kernel time and the equality of the two
    kms_other = None
    if not args.no_secondary:   # (profiling runs pass --no-secondary: one leg's kernels only)
        _e2, kms_other, _ = timed_steps(ctx, dist, world, dev, other, max(3, args.steps // 2), 1)
        assert np.array_equal(t["state"].cpu().numpy(), st) and np.array_equal(t["mr"].cpu().numpy(), mres) and np.array_equal(t["succ"].cpu().numpy(), succ)
    elapsed, kms, _ = timed_steps(ctx, dist, world, dev, step, args.steps, args.warmup)
    assert np.array_equal(t["state"].cpu().numpy(), st) and np.array_equal(t["mr"].cpu().numpy(), mres) and np.array_equal(t["succ"].cpu().numpy(), succ)
    cnt = misc_counters(ctx)
    # warp <= 11x11 B, scan 64+64 B per ZMSSD, align 81 B per iteration, state 32 B in + out
    alg = cnt[0] * 121 + cnt[1] * 128 + cnt[2] * 81 + n * 32 + cnt[3] * 32
    elapsed, total = du.combine(dist, world, elapsed, n, comm_dev)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        orc.build(fast=True)
        t_cpu, n_cpu = 0.0, 0
        for i, sc in enumerate(scenes):
            ref = orc.create_img_pyramid(imgs[2 * i].cpu().numpy(), 5, fast=True)
            cur = orc.create_img_pyramid(imgs[2 * i + 1].cpu().numpy(), 5, fast=True)
            ov_r = orc.make_frame_view(ref, cam, sc.T_ref_f_w, seeds[i]["mu_range"], 2 * i)
            ov_c = orc.make_frame_view(cur, cam, sc.T_cur_f_w_gt, 0.0, 2 * i + 1)
            sd = seeds[i]
            fbo, ko = orc.make_feature_batch(sd["ref_frame_idx"], sd["px"], sd["f"], sd["grad"], sd["level"], sd["type"])
            t0 = time.perf_counter()
            nso, sto, so, mro = orc.update_seeds_batch(mopt, dopt, [ov_r], ov_c, fbo, sd["state"], fast=True)
            t_cpu += time.perf_counter() - t0
            n_cpu += NS
            sl = slice(i * NS, (i + 1) * NS)
            assert (mro != mres[sl]).mean() < 1e-3  # -O3 -march=native may contract FMAs: rare knife-edge flips allowed
            if t_cpu > 10.0:
                break
        cpu = {"value": n_cpu / t_cpu, "unit": "seed updates/s", "cores": 1, "kind": "port",
               "sample": "%d seeds of the benchmark (oracle updateSeed, gcc -O3 -march=native, 1 thread, %.1f s)" % (n_cpu, t_cpu)}
    if rank != 0:
        return None
    return {"metric": "depth-filter seed updates/s (updateSeed: epipolar ZMSSD scan + align + Vogiatzis)",
            "value": total * args.steps / elapsed, "unit": "seed updates/s", "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_frame": 1e3 * elapsed / args.steps / B, "dtype": "i32+f32+f64",
            "config": {"workload": "C4-synth: %d (keyframe, frame) pairs x %d seeds per GPU per step, 640x480, 8x8 patches, "
                                   "<=100 epipolar steps, seed arrays resident in HBM%s" % (B, NS, "; the keyframes' columns resident (svoh_features_upload), "
                                   "the batch = every seed of every keyframe in the upload's tile order (SVOH_BATCH_WHOLE_SETS)" if args.whole_sets else ""),
                       "frame_pairs_per_gpu": B, "seeds_per_keyframe": NS},
            "kernel_ms": kms, "success_fraction": float(succ.mean()),
            # both forms give the same bits (asserted above): per-unit columns in the caller's device arrays, binned per step
            # (count + scatter + packed kernel + un-sort), and whole resident sets (the packed kernel alone)
            "kernel_ms_units_binned_per_step": kms_other if args.whole_sets else kms,
            "kernel_ms_whole_resident_sets": kms if args.whole_sets else kms_other,
            "host_staged_seed_updates_per_s": host_rate,  # same work with host arrays staged per call (PCIe-inclusive)
            "roofline": roofline("update_seeds_packed_kernel alone, over the resident sets' tile-ordered columns" if args.whole_sets else
                                 "update_seeds_packed_kernel (+ seed_bin_count / scan / scatter, seed_unsort: the whole step)", kms,
                                 alg, ("seeds-ws:default" if args.whole_sets else "seeds:default") if not args.problems else "seeds:B%d" % B, counters=cnt[:4],
                                 unit_tails={"align_iters_ge5": cnt[4], "align_iters_ge10": cnt[5], "zmssd_ge20": cnt[6],
                                             "zmssd_ge50": cnt[7]}),
            "cpu_baseline": cpu}


def bench_stereo(args, ctx, dist, rank, world, dev, comm_dev=None):
    """The stereo keyframe seam (StereoTriangulation::compute, stereo_triangulation.cpp:60-140): per stereo pair 120
    features of the left frame searched along the epipolar line in the right frame (Matcher defaults: unit sphere,
    max_epi_search_steps = 500), depth triangulated.  B pairs per step through svoh_epipolar_match_batch (host arrays)."""
    B = args.problems or 64
    NF = 120
    cam, scenes, imgs, frames = render_pairs(ctx, dev, rank, B, 4, rot_deg=(0.0, 0.2), trans_m=(0.08, 0.12))
    mopt = capi.default_matcher_options(max_epi_search_steps=500, subpix_refinement=1, scan_on_unit_sphere=1)
    feats = [synth.make_seed_set(sc, NF, seed=i, margin=6, levels=(0, 1, 2)) for i, sc in enumerate(scenes)]
    ref_views = [fe.make_frame_view(frames[2 * i], cam, sc.T_ref_f_w, 0.0, 2 * i) for i, sc in enumerate(scenes)]
    cur_views = [fe.make_frame_view(frames[2 * i + 1], cam, sc.T_cur_f_w_gt, 0.0, 2 * i + 1) for i, sc in enumerate(scenes)]
    idx = np.repeat(np.arange(B, dtype=np.int32), NF)
    cat = lambda k: np.concatenate([f[k] for f in feats])
    ftype = np.where(cat("type") == 0, capi.FT_EDGELET, capi.FT_CORNER).astype(np.uint8)   # detector output, not seeds
    fb, keep = fe.make_feature_batch(idx, cat("px"), cat("f"), cat("grad"), cat("level"), ftype)
    fb.cur_frame_idx = idx.ctypes.data
    fb.n_cur_frames = B
    n = B * NF
    d_inv = np.concatenate([np.tile([1.0 / np.median(f["true_depth"]), 1.0 / (0.3 * np.median(f["true_depth"])),
                                     1.0 / (15.0 * np.median(f["true_depth"]))], NF) for f in feats])

    def step():
        out = ctx.epipolar_match_batch(mopt, ref_views, cur_views, fb, d_inv=d_inv)
        return out, misc_kernel_ms(ctx)

    elapsed, kms, out = timed_steps(ctx, dist, world, dev, step, args.steps, args.warmup)
    ok = out["result"] == capi.MATCH_SUCCESS
    td = cat("true_depth")
    depth_err = float(np.median(np.abs(out["depth"][ok] - td[ok]) / td[ok])) if ok.any() else None
    cnt = misc_counters(ctx)
    alg = cnt[0] * 121 + cnt[1] * 128 + cnt[2] * 81 + n * 8
    elapsed, total = du.combine(dist, world, elapsed, n, comm_dev)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc  # test infrastructure: the timed CPU baseline only
        orc.build(fast=True)
        t_cpu, n_cpu = 0.0, 0
        for i, sc in enumerate(scenes):
            ref = orc.create_img_pyramid(imgs[2 * i].cpu().numpy(), 5, fast=True)
            cur = orc.create_img_pyramid(imgs[2 * i + 1].cpu().numpy(), 5, fast=True)
            ov_r = orc.make_frame_view(ref, cam, sc.T_ref_f_w, 0.0, 2 * i)
            ov_c = orc.make_frame_view(cur, cam, sc.T_cur_f_w_gt, 0.0, 2 * i + 1)
            f = feats[i]
            sl = slice(i * NF, (i + 1) * NF)
            fbo, ko = orc.make_feature_batch(f["ref_frame_idx"], f["px"], f["f"], f["grad"], f["level"], ftype[sl])
            for rep in range(40):   # a pair's 120 features take ~0.2 ms: repeated for a stable time
                t0 = time.perf_counter()
                oo = orc.epipolar_match_batch(mopt, [ov_r], ov_c, fbo, d_inv=d_inv[3 * i * NF:3 * (i + 1) * NF], fast=True)
                t_cpu += time.perf_counter() - t0
                n_cpu += NF
            assert (oo["result"] != out["result"][sl]).mean() < 1e-2   # -O3 -march=native may contract FMAs: rare knife-edge flips
            if t_cpu > 10.0:
                break
        cpu = {"value": n_cpu / t_cpu, "unit": "features/s", "cores": 1, "kind": "port",
               "sample": "%d feature searches: every pair of the benchmark 40 times (oracle findEpipolarMatchDirect, gcc -O3 -march=native, 1 thread, %.1f s)" % (n_cpu, t_cpu)}
    if rank != 0:
        return None
    return {"metric": "stereo-triangulated features/s (findEpipolarMatchDirect, <=500 epipolar steps, 120 features per stereo pair)",
            "value": total / elapsed, "unit": "features/s", "ms_per_step": 1e3 * elapsed / args.steps, "dtype": "u8+i32+f32+f64",
            "config": {"workload": "stereo seam: %d stereo pairs x %d features per GPU per step, 640x480, unit-sphere scan, host arrays in / out" % (B, NF),
                       "pairs_per_gpu": B, "features_per_pair": NF},
            "kernel_ms": kms, "success_fraction": float(ok.mean()), "median_depth_error": depth_err,
            "roofline": roofline("epipolar_match_kernel", kms, alg, "stereo:default", counters=cnt[:4]),
            "cpu_baseline": cpu}


def bench_frame(args, ctx, dist, rank, world, dev, comm_dev=None):
    """C3-synth: the per-frame hot path at the EuRoC mono sizes of SURVEY.md Appendix A, one frame at a time
    (latency, not throughput): 752x480 radtan camera; 5-level pyramid of the new image (host image in);
    SparseImgAlign 180 features, 4x4, levels 4..2 (host arrays in, pose out); KLT 180 tracks {16,16,16,8,8};
    depth-filter update of 3 keyframes x 540 seeds against the new frame (host arrays in/out).  Every stage is
    one blocking C-ABI call, as the frame handler would make it."""
    cam = synth.Camera.euroc_like(752, 480)
    NF, NS, NKF = 180, 540, 3
    sc = synth.make_align_scene(du.problem_seed(rank, 7), n_features=NF, patch_size=4, cam=cam, max_level=4,
                                rot_deg=(0.3, 1.0), trans_m=(0.03, 0.10))
    opt = capi.default_align_options(max_level=4, min_level=2, patch_size=4)
    kopt = capi.default_klt_options()
    mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(cam)
    f_ref = ctx.build_pyramid(sc.img_ref, 5)
    tracks = synth.make_track_set(sc, NF, seed=1)
    seeds = synth.make_seed_set(sc, NS * NKF, seed=2)
    kf_idx = np.repeat(np.arange(NKF, dtype=np.int32), NS)
    stages = {k: [] for k in ("pyramid", "align", "klt", "seeds", "total")}
    last = {}

    def one_frame():
        t0 = time.perf_counter()
        f_cur = ctx.build_pyramid(sc.img_cur, 5)
        t1 = time.perf_counter()
        problems, keep = fe.make_align_problems([[(sc, f_ref, f_cur, None)]])
        res = ctx.sparse_align(opt, problems)
        t2 = time.perf_counter()
        pk, sk = ctx.klt_track_batch(kopt, f_ref, f_cur, tracks["px_ref"], tracks["px_cur_init"])
        t3 = time.perf_counter()
        ref_views = [fe.make_frame_view(f_ref, cam, sc.T_ref_f_w, seeds["mu_range"], k) for k in range(NKF)]
        cur_view = fe.make_frame_view(f_cur, cam, sc.T_cur_f_w_gt, 0.0, 100)
        fb, kk = fe.make_feature_batch(kf_idx, seeds["px"], seeds["f"], seeds["grad"], seeds["level"], seeds["type"])
        ns, st, succ, mr = ctx.update_seeds_batch(mopt, dopt, ref_views, cur_view, fb, seeds["state"])
        t4 = time.perf_counter()
        ctx.release_frame(f_cur)
        last.update(res=res[0], klt=(pk, sk), seeds=(ns, st, succ, mr))
        return t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0

    for _ in range(3):
        one_frame()   # with the event pairs on: the alignment kernel's device time for the roofline line below
    kms = ctypes.c_float()
    ctx.lib.svoh_sparse_align_last_kernel_ms(ctx.h, ctypes.byref(kms))
    ctx.set_kernel_timing(False)   # the library's default: the stage times are those of a deployment
    for _ in range(args.warmup):
        one_frame()
    if world > 1:
        dist.barrier()
    t_begin = time.perf_counter()
    for _ in range(args.steps):
        ts = one_frame()
        for k, v in zip(("pyramid", "align", "klt", "seeds", "total"), ts):
            stages[k].append(1e3 * v)
    ctx.synchronize()
    elapsed = time.perf_counter() - t_begin
    ctx.set_kernel_timing(True)
    elapsed, total_frames = du.combine(dist, world, elapsed, args.steps, comm_dev)
    med = {k: float(np.median(v)) for k, v in stages.items()}
    err = synth.se3_error(synth.SE3.from7(fe.se3_to_numpy(last["res"].T_icur_iref)), sc.T_icur_iref_gt)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc  # test infrastructure: the timed CPU baseline only
        orc.build(fast=True)
        ref = orc.create_img_pyramid(sc.img_ref, 5, fast=True)
        c = {k: [] for k in stages}
        t_cpu0 = time.perf_counter()
        while time.perf_counter() - t_cpu0 < 10.0 and len(c["total"]) < 200:
            t0 = time.perf_counter()
            cur = orc.create_img_pyramid(sc.img_cur, 5, fast=True)
            t1 = time.perf_counter()
            pb = orc.problem_from_scenes([(sc, ref, cur)])
            n_o, res_o, _ = orc.sparse_align_run(opt, pb, fast=True)
            t2 = time.perf_counter()
            po, so = orc.klt_track_batch(kopt, ref, cur, tracks["px_ref"], tracks["px_cur_init"], fast=True)
            t3 = time.perf_counter()
            ov_r = [orc.make_frame_view(ref, cam, sc.T_ref_f_w, seeds["mu_range"], k) for k in range(NKF)]
            ov_c = orc.make_frame_view(cur, cam, sc.T_cur_f_w_gt, 0.0, 100)
            fbo, ko = orc.make_feature_batch(kf_idx, seeds["px"], seeds["f"], seeds["grad"], seeds["level"], seeds["type"])
            nso, sto, so2, mro = orc.update_seeds_batch(mopt, dopt, ov_r, ov_c, fbo, seeds["state"], fast=True)
            t4 = time.perf_counter()
            for k, v in zip(("pyramid", "align", "klt", "seeds", "total"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)):
                c[k].append(1e3 * v)
        cmed = {k: float(np.median(v)) for k, v in c.items()}
        assert np.array_equal(so, last["klt"][1]) and np.array_equal(po, last["klt"][0])
        cpu = {"value": 1e3 / cmed["total"], "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "%d repetitions of the same frame through the oracle (gcc -O3 -march=native, 1 thread)" % len(c["total"]),
               "stage_ms_median": cmed}
    if rank != 0:
        return None
    alg = algorithmic_bytes(4, 6, last["res"].n_patch_iters, last["res"].n_fts_to_track * 3)
    return {"metric": "frames/s, one frame at a time (pyramid + SparseImgAlign + KLT + depth-filter update, EuRoC mono sizes)",
            "value": total_frames / elapsed, "unit": "frames/s", "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_frame": med["total"], "dtype": "u8+i32+f32+f64",
            "config": {"workload": "C3-synth: 752x480 radtan, 5-level pyramid, align 180 feat 4x4 levels 4..2, KLT 180 tracks, "
                                   "3 keyframes x 540 seeds; blocking C-ABI calls with host arrays, one frame at a time",
                       "features": NF, "seeds": NS * NKF},
            "stage_ms_median": med, "align_pose_err_vs_gt": {"rot_rad": err[0], "trans_m": err[1]},
            "seed_successes": int(last["seeds"][0]), "klt_converged": int(last["klt"][1].sum()),
            "roofline": roofline("sparse_align_kernel<4,*,false> (single problem: latency-bound by design)", kms.value, alg,
                                 "frame:default"),
            "cpu_baseline": cpu}


def bench_frame_streams(args, ctx, dist, rank, world, dev, comm_dev=None):
    """BASELINE config 5's "batched multi-sequence" on ONE GPU per rank: S camera streams through the whole per-frame chain
    of FrameHandlerMono::processFrame (frame_handler_mono.cpp:120-158) in LOCK STEP -- sparse alignment + candidate
    projection, reprojection (direct + seed matcher batches), pose optimisation, depth-filter update, detector and seed
    initialisation at keyframes -- every stage ONE launch for all streams of a group (host/svo_hip_lockstep.h), the
    streams' host work on W threads per group, G groups side by side (one group's host phases against the other's device
    phases).  A step = one round = one frame of every stream.  The streams of a rank replay one rendered EuRoC-layout
    sequence (752x480 radtan, forwards then backwards, so that the run never has to restart); images in page-locked
    memory, every stream reading ITS OWN copy, so that every image crosses PCIe inside the timed region.  Per-stream results are those of the
    single-stream chain byte for byte (tests/test_mini_frontend_gpu.py::test_lockstep_streams_reproduce_the_single_stream)."""
    import threading
    from svo_pro_universal_amd import lockstep as ls
    cam = synth.Camera.euroc_like(752, 480)
    n_frames = 40
    sc = synth.make_align_scene(du.problem_seed(rank, 160), n_features=8, cam=cam, rot_deg=(0.3, 0.5), trans_m=(0.015, 0.025))
    stepT = sc.T_w_ref.inverse() * sc.T_w_cur
    poses = [sc.T_w_ref]
    for _ in range(1, n_frames):
        poses.append(poses[-1] * stepT)
    images = [synth.render(cam, T, sc.plane, sc.tex) for T in poses]
    depth = float(np.mean(sc.depth))
    params = ("max_fts: 180\ngrid_size: 30\nn_pyr_levels: 3\ndetector_threshold_secondary: 100\nuse_threaded_depthfilter: False\n"
              "img_align_max_level: 4\nimg_align_min_level: 2\n")
    # host threads of this rank: its own slice of the CPUs when main() pinned the rank to one (pin_rank_to_its_cores has
    # divided by the world size already); an equal share of what the process sees when it did not
    n_slice = len(os.sched_getaffinity(0))
    n_host = _CPUS_BEFORE_PIN or n_slice
    budget = int(os.environ.get("SVOH_BENCH_HOST_THREADS", "0")) or max(1, min(16, n_slice if _PINNED else n_slice // max(1, world)))

    def frame_of(k):   # 0 1 .. n-1 n-2 .. 1 0 1 ..: the camera walks the path forth and back
        k %= 2 * (n_frames - 1)
        return k if k < n_frames else 2 * (n_frames - 1) - k

    def mix_of(s):   # stream s of a rank's --stream-mix: (start, step, every, phase), keyframe period, feature budget
        step = (1, -1, 1, -1, 2, 1, -1, -2)[s % 8]
        return ((7 * s) % n_frames, step, 2 if s % 8 == 5 else 1, 0), (8, 6, 5, 7, 9, 4)[s % 6], (180, 120, 240)[s % 3]

    def camera_of(s):   # ... and every fourth stream of the mix a calibration of its own (same sensor size: LockstepStreamOptions::own_camera)
        if s % 4 != 3:
            return cam
        return synth.Camera(cam.width, cam.height, cam.fx * 1.01, cam.fy * 0.992, cam.cx + 1.5, cam.cy - 2.0, dist=[0.9 * d for d in cam.dist] if cam.dist is not None else None)

    def image_of(sched, k):   # the image of the stream's frame in round k (svohl_run_schedule's rule), or None
        start, step, every, phase = sched
        if k < phase or (k - phase) % every:
            return None
        m = (start + step * ((k - phase) // every)) % (2 * (n_frames - 1))
        return m if m < n_frames else 2 * (n_frames - 1) - m

    def run(S, G, W, n_warm, n_steps, mix=False, shared_classes=False):
        G = max(1, min(G, S))
        ctxs = [ctx] + [fe.Context(dev.index if dev.index is not None else 0, kernel_timing=False) for _ in range(G - 1)]
        ctx.set_kernel_timing(False)
        for c in ctxs:   # svoh_set_align_geometry_classes: one launch geometry for every alignment problem below 512 patches
            c.set_align_geometry_classes(shared_classes)
        ranges = [(S * g // G, S * (g + 1) // G) for g in range(G)]
        pins = [ls.PinnedImages(c, images, hi - lo) for c, (lo, hi) in zip(ctxs, ranges)]   # every stream its own copy of the sequence
        shared = None   # (a pool shared between the groups was built and raced in round 5: no gain, profiles/r05_shared_pool_ab.txt; every group has its own)
        per = [None] * G
        if mix:
            per = [[dict(params_yaml=params.replace("max_fts: 180", "max_fts: %d" % mix_of(s)[2]), kf_every=mix_of(s)[1], cam=camera_of(s)) for s in range(lo, hi)] for lo, hi in ranges]
        engines = [ls.Lockstep(c, hi - lo, cam, np.array([1.0, 0, 0, 0, 0, 0, 0]), params, 0.5 * depth, depth, 2.0 * depth, 8, W, True, pool=shared, seed=lo, per_stream=ps)
                   for c, (lo, hi), ps in zip(ctxs, ranges, per)]
        first = poses[0].inverse().as7()
        frames_taken = [0] * G
        total = n_warm + n_steps
        gate = threading.Barrier(G + 1)
        times = [[] for _ in range(G)]
        warm_phases = [{} for _ in range(G)]
        errors = []

        def loop(g):
            try:
                e, pin, n = engines[g], pins[g], ranges[g][1] - ranges[g][0]
                # (a group's rounds run inside ONE foreign call: with the loop in Python, the groups' threads spend more
                # time handing the interpreter lock to each other than in their rounds)
                if mix:
                    sched = [mix_of(s)[0] for s in range(ranges[g][0], ranges[g][1])]
                    firsts = [poses[sc[0]].inverse().as7() for sc in sched]
                    e.run_schedule(pin, cam.width, 0, n_warm, sched, firsts)
                else:
                    e.run_sequence(pin, cam.width, 0, n_warm, [first] * n)
                warm_phases[g] = e.phase_times()
                gate.wait()   # the timed region starts for every group at once
                if mix:
                    times[g], frames_taken[g] = e.run_schedule(pin, cam.width, n_warm, n_steps, sched, firsts)
                else:
                    times[g] = e.run_sequence(pin, cam.width, n_warm, n_steps)
                    frames_taken[g] = n * n_steps
                e.finish()
                gate.wait()
            except Exception as ex:   # noqa: BLE001 -- reported by the caller
                errors.append(ex)
                gate.abort()
        th = [threading.Thread(target=loop, args=(g,)) for g in range(G)]
        for t in th:
            t.start()
        try:
            gate.wait()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            gate.wait()
            elapsed = time.perf_counter() - t0
        except threading.BrokenBarrierError:
            elapsed = float("nan")
        for t in th:
            t.join()
        if errors:
            raise errors[0]
        # every stream saw the same images: every stream must be at the same pose, to the bit (a monocular chain's pose
        # against the rendered one is right up to the scale its depth prior gave it: tests/test_mini_frontend_gpu.py has the ATE)
        p0 = engines[0].pose(0)
        agree = all(np.array_equal(e.pose(s), p0) for e in engines for s in range(e.n))
        gt = poses[frame_of(total - 1)].inverse()
        err = synth.se3_error(synth.SE3.from7(p0), gt)
        mix_info = None
        if mix:   # every stream against the rendered pose of ITS last image (unscaled, as below)
            errs, distinct = [], set()
            for e, (lo, hi) in zip(engines, ranges):
                for i, s in enumerate(range(lo, hi)):
                    sched = mix_of(s)[0]
                    last = max(k for k in range(total) if image_of(sched, k) is not None)
                    T = e.pose(i)
                    distinct.add(T.tobytes())
                    errs.append(synth.se3_error(synth.SE3.from7(T), poses[image_of(sched, last)].inverse()))
            err = (max(v[0] for v in errs), max(v[1] for v in errs))
            mix_info = {"distinct_final_poses": len(distinct), "keyframe_periods": sorted({mix_of(s)[1] for s in range(S)}), "max_fts": sorted({mix_of(s)[2] for s in range(S)}),
                        "steps_over_the_sequence": sorted({mix_of(s)[0][1] for s in range(S)}), "streams_at_half_rate": sum(1 for s in range(S) if mix_of(s)[0][2] == 2),
                        "pose_error_is": "the maximum over the streams"}
        rows = engines[0].completed_rows(0)
        stage = dict(zip(("pyramid", "align", "reproject", "pose", "seeds", "keyframe", "total"), [float(v) for v in np.median(times[0], axis=0)])) if len(times[0]) else {}
        calls = engines[0].last_round()[1]
        # where group 0's thread spent the timed rounds: waits for the device against host phases
        phases = {k: (v - warm_phases[0].get(k, 0.0)) / max(1, n_steps) for k, v in engines[0].phase_times().items()}
        for e in engines:
            e.close()
        if shared is not None:
            shared.close()
        for p in pins:
            p.free()
        for c in ctxs[1:]:
            c.close()
        return {"streams": S, "groups": G, "host_threads_per_group": W, "host_threads": G * W, "workers_shared_between_groups": shared is not None,
                "frames_per_s": sum(frames_taken) / elapsed, "ms_per_round": 1e3 * elapsed / n_steps, "frames_in_the_timed_rounds": int(sum(frames_taken)),
                "stream_mix": mix_info,
                "round_stage_ms_median_group0": stage, "round_phase_ms_mean_group0": {k: round(v, 4) for k, v in phases.items()},
                "device_waits_ms_per_round_group0": round(sum(phases.get(k, 0.0) for k in ("seed wait", "align wait", "match wait", "pose call", "detect wait")), 4),
                "device_calls_per_round_per_group": calls,
                "all_streams_at_the_same_pose": bool(agree), "rounds_run": total,
                "pose_vs_rendered_pose_unscaled": {"rot_rad": float(err[0]), "trans_m": float(err[1]), "path_m_per_traverse": float(np.linalg.norm(np.asarray(stepT.t))) * (n_frames - 1)},
                "features_per_frame_median": float(np.median(rows[1:, 3])) if len(rows) > 1 else None}, elapsed

    def shape(S):   # groups and threads per group out of the rank's host-thread budget
        G = 1 if S < 8 else min(4 if S >= 24 else 2, budget)   # (measured shapes: profiles/r05_lockstep_steps_ab.txt)
        return G, max(1, min(budget // G, -(-S // G)))   # (no more threads than a group has streams)

    S = args.streams
    G, W = (args.stream_groups, args.stream_workers) if args.stream_groups and args.stream_workers else shape(S)
    # streams that differ run with SHARED alignment classes (a round's alignment is then one or two launches instead of a launch per size class;
    # every stream still reproduces its single-stream run under the same setting: tests/test_mini_frontend_gpu.py); --own-align-classes: the default classes
    shared_classes = bool(args.stream_mix and not args.own_align_classes)
    main_run, elapsed = run(S, G, W, max(3, args.warmup), args.steps, mix=args.stream_mix, shared_classes=shared_classes)
    main_run["shared_alignment_classes"] = shared_classes
    elapsed, total_frames = du.combine(dist, world, elapsed, main_run["frames_in_the_timed_rounds"], comm_dev)
    same_streams = own_classes = None
    if args.stream_mix and rank == 0 and world == 1:   # the identical-streams number beside it: the best case of the grouping, speculation and detector batching
        r0, _ = run(S, G, W, max(3, args.warmup), args.steps)
        same_streams = {k: r0[k] for k in ("streams", "groups", "host_threads_per_group", "frames_per_s", "ms_per_round", "all_streams_at_the_same_pose", "device_waits_ms_per_round_group0")}
        if shared_classes:   # ... and the mix with every problem in the geometry that is fastest for it alone (a launch per size class)
            r1, _ = run(S, G, W, max(3, args.warmup), args.steps, mix=True, shared_classes=False)
            own_classes = {k: r1[k] for k in ("frames_per_s", "ms_per_round", "device_waits_ms_per_round_group0")}
            own_classes["align_ms_per_round_group0"] = {k: r1["round_phase_ms_mean_group0"][k] for k in ("align launch", "align wait")}
    sweep = []
    if rank == 0 and world == 1 and not args.no_secondary and not args.stream_mix:
        for s2 in (1, 8, 32, 64):
            if s2 == S:
                sweep.append({k: main_run[k] for k in ("streams", "groups", "host_threads_per_group", "frames_per_s", "ms_per_round")})
                continue
            g2, w2 = shape(s2)
            r2, _ = run(s2, g2, w2, max(3, args.warmup), args.steps)   # (as long as the main run: the map of a short run is still filling)
            sweep.append({k: r2[k] for k in ("streams", "groups", "host_threads_per_group", "frames_per_s", "ms_per_round")})
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_frame_stages(cam)
    if rank != 0:
        return None
    return {"metric": "frames/s, S camera streams in lock step through the whole per-frame chain (align + reproject + pose + depth filter + keyframes), one launch per stage",
            "value": total_frames / elapsed, "unit": "frames/s", "ms_per_step": 1e3 * elapsed / args.steps, "ms_per_frame": 1e3 * elapsed * world / max(1, total_frames),
            "dtype": "u8+i32+f32+f64",
            "config": {"workload": "C5-synth per GPU: %d streams x 752x480 radtan in %d lock-step group(s) with %d host thread(s) each; per stream and frame: 5-level "
                                   "pyramid, align <=180..720 patches 4x4 levels 4..2, ~100 direct + ~900 seed matcher units, pose over <=180 features, "
                                   "depth-filter update of <=5 keyframes (~1600 seeds), a keyframe every 8 frames; a step = one frame of every stream; "
                                   "the streams replay one rendered sequence forth and back%s" % (S, G, W, " -- EVERY STREAM ITS OWN WALK, keyframe period and feature budget, every fourth its own camera calibration (--stream-mix)" if args.stream_mix else ""),
                       "streams": S, "groups": G, "host_threads_per_group": W, "host_cpus_visible": n_host},
            "lockstep": main_run, "identical_streams_beside_it": same_streams, "mix_with_own_alignment_classes_beside_it": own_classes, "streams_sweep": sweep,
            "roofline": {"bound": "latency", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                         "note": "a chain of small latency-bound launches by design: the kernels' own rooflines are those of --workload align / seeds / pose / klt"},
            "cpu_baseline": cpu}


def cpu_frame_stages(cam):
    """The device stages of one stream's frame through the oracle, one thread: 5-level pyramid, SparseImgAlign of 180 features
    (levels 4..2), 100 direct matches, 900 + 1620 seed updates, the pose optimiser over 180 features.  Host bookkeeping
    (candidate walk, sort, replay) is the same code on both sides and not part of it."""
    from oracle import oracle as orc  # test infrastructure: the timed CPU baseline only
    orc.build(fast=True)
    sc = synth.make_align_scene(7, n_features=180, patch_size=4, cam=cam, max_level=4, rot_deg=(0.3, 1.0), trans_m=(0.03, 0.10))
    opt = capi.default_align_options(max_level=4, min_level=2, patch_size=4)
    mopt, dopt = capi.default_matcher_options(), capi.default_depth_filter_options(cam)
    NS = 900 + 1620
    seeds = synth.make_seed_set(sc, NS, seed=2)
    kf_idx = np.zeros(NS, dtype=np.int32)
    ref = orc.create_img_pyramid(sc.img_ref, 5, fast=True)
    c = []
    t_cpu0 = time.perf_counter()
    while time.perf_counter() - t_cpu0 < 10.0 and len(c) < 100:
        t0 = time.perf_counter()
        cur = orc.create_img_pyramid(sc.img_cur, 5, fast=True)
        pb = orc.problem_from_scenes([(sc, ref, cur)])
        orc.sparse_align_run(opt, pb, fast=True)
        ov_r = [orc.make_frame_view(ref, cam, sc.T_ref_f_w, seeds["mu_range"], 0)]
        ov_c = orc.make_frame_view(cur, cam, sc.T_cur_f_w_gt, 0.0, 100)
        fbo, ko = orc.make_feature_batch(kf_idx, seeds["px"], seeds["f"], seeds["grad"], seeds["level"], seeds["type"])
        orc.update_seeds_batch(mopt, dopt, ov_r, ov_c, fbo, seeds["state"], fast=True)
        c.append(time.perf_counter() - t0)
    med = float(np.median(c))
    return {"value": 1.0 / med, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d repetitions of one stream's frame through the oracle (pyramid + align 180 features + %d seed updates; gcc -O3 -march=native, 1 thread; "
                      "the direct matches and the pose optimiser, ~5 %% of the frame, are left out)" % (len(c), NS)}


def bench_frame_stereo_streams(args, ctx, dist, rank, world, dev, comm_dev=None):
    """BASELINE config 3 x config 5 on one GPU: S STEREO streams in lock step (FrontendLockstepStereo, host/svo_hip_lockstep_stereo.h) -- per pair
    and stream: two 5-level pyramids, the two-camera alignment bundle (pose + illumination gain / offset, IMU rotation prior), both cameras' reprojection,
    the rig's pose optimisation, structure optimisation, both depth-filter updates, stereo triangulation + new seeds at keyframes.  A step = one round =
    one pair of every stream.  The streams of a rank replay one rendered stereo sequence (752x480 radtan, baseline 0.11 m, drifting gain and offset) forth
    and back, every stream reading ITS OWN page-locked copy.  Per-stream results are those of the single-stream stereo chain byte for byte
    (tests/test_mini_stereo_gpu.py::test_stereo_streams_in_lock_step_reproduce_their_single_stream_runs)."""
    import threading
    from svo_pro_universal_amd import lockstep as ls
    cam = synth.Camera.euroc_like(752, 480)
    n_pairs, base_m = 30, 0.11
    sc = synth.make_align_scene(du.problem_seed(rank, 171), n_features=8, cam=cam, rot_deg=(0.3, 0.5), trans_m=(0.015, 0.025))
    stepT = sc.T_w_ref.inverse() * sc.T_w_cur
    poses = [sc.T_w_ref]
    for _ in range(1, n_pairs):
        poses.append(poses[-1] * stepT)
    T_B_C = [synth.SE3(), synth.SE3((1.0, 0.0, 0.0, 0.0), (base_m, 0.0, 0.0))]
    images = []
    for k, T in enumerate(poses):
        gain, offset = 1.0 + 0.08 * np.sin(k / 4.0), 6.0 * np.cos(k / 5.0)
        for c in range(2):
            images.append(synth.render(cam, T * T_B_C[c], sc.plane, sc.tex, gain=gain, offset=offset))
    rng = np.random.RandomState(3)
    prior = np.zeros((n_pairs, 4)); prior[0] = (1, 0, 0, 0)
    for k in range(1, n_pairs):   # what a gyroscope integration would hand over: R_imu(k)_imu(k-1), slightly off
        noise = synth.SE3(synth.quat_from_axis_angle(rng.normal(size=3), 2e-4), (0, 0, 0))
        prior[k] = (noise * (poses[k].inverse() * poses[k - 1])).q
    params = ("max_fts: 160\ngrid_size: 35\nn_pyr_levels: 3\ndetector_threshold_secondary: 100\nuse_threaded_depthfilter: False\n"
              "img_align_max_level: 4\nimg_align_min_level: 2\n")
    n_slice = len(os.sched_getaffinity(0))
    n_host = _CPUS_BEFORE_PIN or n_slice
    budget = int(os.environ.get("SVOH_BENCH_HOST_THREADS", "0")) or max(1, min(16, n_slice if _PINNED else n_slice // max(1, world)))
    S = args.streams
    G = args.stream_groups or (1 if S < 8 else min(8 if S >= 32 else 4, budget))
    W = args.stream_workers or max(1, min(budget // G, -(-S // G)))
    G = max(1, min(G, S))
    ctxs = [ctx] + [fe.Context(dev.index if dev.index is not None else 0, kernel_timing=False) for _ in range(G - 1)]
    ctx.set_kernel_timing(False)
    ranges = [(S * g // G, S * (g + 1) // G) for g in range(G)]
    pins = [ls.PinnedImages(c, images, hi - lo) for c, (lo, hi) in zip(ctxs, ranges)]
    T7 = [np.array(list(T.q) + list(T.t)) for T in T_B_C]
    engines = [ls.LockstepStereo(c, hi - lo, [cam, cam], T7, params, 8, 0.5, W, True) for c, (lo, hi) in zip(ctxs, ranges)]
    first = poses[0].inverse().as7()
    n_warm, n_steps = max(3, args.warmup), args.steps
    gate = threading.Barrier(G + 1)
    errors, times = [], [None] * G

    def loop(g):
        try:
            e, pin, n = engines[g], pins[g], ranges[g][1] - ranges[g][0]
            e.run_sequence(pin, cam.width, 0, n_warm, [first] * n, prior)
            gate.wait()
            times[g] = e.run_sequence(pin, cam.width, n_warm, n_steps, None, prior)
            e.finish()
            gate.wait()
        except Exception as ex:   # noqa: BLE001 -- reported by the caller
            errors.append(ex)
            gate.abort()
    warm_phases = [e.phase_times() for e in engines]
    th = [threading.Thread(target=loop, args=(g,)) for g in range(G)]
    for t in th:
        t.start()
    try:
        gate.wait()
        warm_phases = [e.phase_times() for e in engines]
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        gate.wait()
        elapsed = time.perf_counter() - t0
    except threading.BrokenBarrierError:
        elapsed = float("nan")
    for t in th:
        t.join()
    if errors:
        raise errors[0]
    p0 = engines[0].pose(0)
    agree = all(np.array_equal(e.pose(s), p0) for e in engines for s in range(e.n))
    total = n_warm + n_steps
    m = (total - 1) % (2 * (n_pairs - 1))
    last_pair = m if m < n_pairs else 2 * (n_pairs - 1) - m
    err = synth.se3_error(synth.SE3.from7(p0), poses[last_pair].inverse())
    phases = {k: (v - warm_phases[0].get(k, 0.0)) / max(1, n_steps) for k, v in engines[0].phase_times().items()}
    for e in engines:
        e.close()
    for p in pins:
        p.free()
    for c in ctxs[1:]:
        c.close()
    elapsed, total_pairs = du.combine(dist, world, elapsed, S * n_steps, comm_dev)
    if rank != 0:
        return None
    return {"metric": "stereo frame pairs/s, S stereo streams in lock step through the whole per-pair chain (bundle align + 2 x reproject + rig pose + structure + 2 x depth filter + keyframes)",
            "value": total_pairs / elapsed, "unit": "pairs/s", "ms_per_step": 1e3 * elapsed / n_steps, "ms_per_frame": 1e3 * elapsed * world / max(1, total_pairs), "dtype": "u8+i32+f32+f64",
            "config": {"workload": "C3 x C5-synth per GPU: %d stereo streams x 2 x 752x480 radtan (baseline 0.11 m, IMU rotation prior, illumination gain + offset estimated) in %d "
                                   "lock-step group(s) with %d host thread(s) each; a step = one pair of every stream; the streams replay one rendered 30-pair sequence forth and back" % (S, G, W),
                       "streams": S, "groups": G, "host_threads_per_group": W, "host_cpus_visible": n_host},
            "lockstep_stereo": {"round_ms_median_group0": float(np.median(times[0])) if times[0] is not None and len(times[0]) else None,
                                "round_phase_ms_mean_group0": {k: round(v, 4) for k, v in phases.items()}, "all_streams_at_the_same_pose": bool(agree),
                                "pose_vs_rendered_pose_metric_scale": {"rot_rad": float(err[0]), "trans_m": float(err[1])}, "rounds_run": total},
            "roofline": {"bound": "latency", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                         "note": "a chain of small latency-bound launches by design: the kernels' own rooflines are those of --workload align-c4 / seeds / pose / stereo"},
            "cpu_baseline": None}


def bench_frame_stereo(args, ctx, dist, rank, world, dev, comm_dev=None):
    """C4-synth, one stereo frame pair at a time (latency): BASELINE config 3 / 4 at the sizes of SURVEY.md Appendix A's
    stereo column.  Per pair: two 5-level pyramids (host images in); SparseImgAlign of the BUNDLE -- two cameras x 160
    features, 4x4, levels 4..2, pose + illumination gain and offset (8 parameters), rotation prior lambda 0.5
    (euroc_stereo_imu.yaml) -- one call; depth-filter update of 3 keyframes x 360 seeds per camera against that camera's
    new frame (two calls).  The keyframe-only stereo seam (120 features, 500 epipolar steps) is timed beside the frame.
    Every stage is a blocking C-ABI call with host arrays, as FrameHandlerStereo would make it."""
    cam = synth.Camera.euroc_like(752, 480)
    NF, NS, NKF, NTRI = 160, 360, 3, 120
    seed = du.problem_seed(rank, 9)
    scs = [synth.make_align_scene(seed, n_features=NF, patch_size=4, cam=cam, max_level=4, rot_deg=(0.3, 1.0), trans_m=(0.03, 0.10),
                                  gain=1.03, offset=2.0) for _ in range(2)]
    opt = capi.default_align_options(max_level=4, min_level=2, patch_size=4, estimate_illumination_gain=1, estimate_illumination_offset=1)
    mopt, dopt = capi.default_matcher_options(affine_est_gain=1), capi.default_depth_filter_options(cam)
    mtri = capi.default_matcher_options(max_epi_search_steps=500, subpix_refinement=1, scan_on_unit_sphere=1)
    f_ref = [ctx.build_pyramid(sc.img_ref, 5) for sc in scs]
    seeds = [synth.make_seed_set(sc, NS * NKF, seed=2 + c) for c, sc in enumerate(scs)]
    tri = synth.make_seed_set(scs[0], NTRI, seed=11, margin=6, levels=(0, 1, 2))
    tri_type = np.where(tri["type"] == 0, capi.FT_EDGELET, capi.FT_CORNER).astype(np.uint8)
    dm = float(np.median(tri["true_depth"]))
    kf_idx = np.repeat(np.arange(NKF, dtype=np.int32), NS)
    pr = capi.svoh_align_prior()
    pr.have_prior = 1
    pr.T_prior = fe._se3(scs[0].T_icur_iref_gt)
    pr.lambda_rot = 0.5
    names = ("pyramids", "align_bundle", "seeds", "total")
    stages = {k: [] for k in names}
    last = {}

    def one_pair():
        t0 = time.perf_counter()
        f_cur = [ctx.build_pyramid(sc.img_cur, 5) for sc in scs]
        t1 = time.perf_counter()
        problems, keep = fe.make_align_problems([[(scs[0], f_ref[0], f_cur[0], None), (scs[1], f_ref[1], f_cur[1], None)]], prior=[pr])
        res = ctx.sparse_align(opt, problems)
        t2 = time.perf_counter()
        ns = 0
        for c, sc in enumerate(scs):
            ref_views = [fe.make_frame_view(f_ref[c], cam, sc.T_ref_f_w, seeds[c]["mu_range"], k) for k in range(NKF)]
            cur_view = fe.make_frame_view(f_cur[c], cam, sc.T_cur_f_w_gt, 0.0, 100 + c)
            fb, kk = fe.make_feature_batch(kf_idx, seeds[c]["px"], seeds[c]["f"], seeds[c]["grad"], seeds[c]["level"], seeds[c]["type"])
            n1, st, succ, mr = ctx.update_seeds_batch(mopt, dopt, ref_views, cur_view, fb, seeds[c]["state"])
            ns += n1
        t3 = time.perf_counter()
        last.update(res=res[0], ns=ns, f_cur=f_cur)
        return t1 - t0, t2 - t1, t3 - t2, t3 - t0

    def seam():   # the left frame's new features into the right frame (here: the pair's other image), 500 steps
        rv = fe.make_frame_view(f_ref[0], cam, scs[0].T_ref_f_w, 0.0, 1)
        cv = fe.make_frame_view(last["f_cur"][0], cam, scs[0].T_cur_f_w_gt, 0.0, 2)
        fb, kk = fe.make_feature_batch(tri["ref_frame_idx"], tri["px"], tri["f"], tri["grad"], tri["level"], tri_type)
        t0 = time.perf_counter()
        out = ctx.epipolar_match_batch(mtri, [rv], cv, fb, d_inv_common=[1 / dm, 3 / dm, 0.05 / dm])
        return time.perf_counter() - t0, out

    def release():
        for f in last.get("f_cur", []):
            ctx.release_frame(f)

    for _ in range(3):
        one_pair(); release()
    kms = ctypes.c_float()
    ctx.lib.svoh_sparse_align_last_kernel_ms(ctx.h, ctypes.byref(kms))
    ctx.set_kernel_timing(False)
    for _ in range(args.warmup):
        one_pair(); release()
    if world > 1:
        dist.barrier()
    seam_ms = []
    t_begin = time.perf_counter()
    t_seam = 0.0
    for _ in range(args.steps):
        ts = one_pair()
        for k, v in zip(names, ts):
            stages[k].append(1e3 * v)
        tsm, seam_out = seam()
        t_seam += tsm
        seam_ms.append(1e3 * tsm)
        release()
    ctx.synchronize()
    elapsed = time.perf_counter() - t_begin - t_seam
    ctx.set_kernel_timing(True)
    elapsed, total_pairs = du.combine(dist, world, elapsed, args.steps, comm_dev)
    med = {k: float(np.median(v)) for k, v in stages.items()}
    err = synth.se3_error(synth.SE3.from7(fe.se3_to_numpy(last["res"].T_icur_iref)), scs[0].T_icur_iref_gt)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc  # test infrastructure: the timed CPU baseline only
        orc.build(fast=True)
        refs = [orc.create_img_pyramid(sc.img_ref, 5, fast=True) for sc in scs]
        c_tot = []
        t_cpu0 = time.perf_counter()
        while time.perf_counter() - t_cpu0 < 10.0 and len(c_tot) < 100:
            t0 = time.perf_counter()
            curs = [orc.create_img_pyramid(sc.img_cur, 5, fast=True) for sc in scs]
            pb = orc.problem_from_scenes([(scs[0], refs[0], curs[0]), (scs[1], refs[1], curs[1])], prior=pr)
            n_o, res_o, _ = orc.sparse_align_run(opt, pb, fast=True)
            for c, sc in enumerate(scs):
                ov_r = [orc.make_frame_view(refs[c], cam, sc.T_ref_f_w, seeds[c]["mu_range"], k) for k in range(NKF)]
                ov_c = orc.make_frame_view(curs[c], cam, sc.T_cur_f_w_gt, 0.0, 100 + c)
                fbo, ko = orc.make_feature_batch(kf_idx, seeds[c]["px"], seeds[c]["f"], seeds[c]["grad"], seeds[c]["level"], seeds[c]["type"])
                orc.update_seeds_batch(mopt, dopt, ov_r, ov_c, fbo, seeds[c]["state"], fast=True)
            c_tot.append(1e3 * (time.perf_counter() - t0))
        assert list(res_o.iters) == list(last["res"].iters)
        cpu = {"value": 1e3 / float(np.median(c_tot)), "unit": "frame pairs/s", "cores": 1, "kind": "port",
               "sample": "%d repetitions of the same pair through the oracle (gcc -O3 -march=native, 1 thread)" % len(c_tot),
               "ms_per_pair_median": float(np.median(c_tot))}
    if rank != 0:
        return None
    alg = algorithmic_bytes(4, 8, last["res"].n_patch_iters, last["res"].n_fts_to_track * 3)
    return {"metric": "stereo frame pairs/s, one pair at a time (2 pyramids + bundle SparseImgAlign with illumination terms and rotation prior "
                      "+ depth-filter update per camera, EuRoC stereo sizes)",
            "value": total_pairs / elapsed, "unit": "frame pairs/s", "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_frame": med["total"], "dtype": "u8+i32+f32+f64",
            "config": {"workload": "C4-synth per pair: 2 x 752x480 radtan, 5-level pyramids, bundle alignment 2 x %d features 4x4 levels 4..2 with gain + offset "
                                   "and rotation prior, 2 x 3 keyframes x %d seeds; blocking C-ABI calls with host arrays" % (NF, NS),
                       "features_per_camera": NF, "seeds_per_camera": NS * NKF, "cameras": 2},
            "stage_ms_median": med, "stereo_seam_ms_median": float(np.median(seam_ms)),
            "stereo_seam_successes": int((seam_out["result"] == capi.MATCH_SUCCESS).sum()),
            "align_pose_err_vs_gt": {"rot_rad": err[0], "trans_m": err[1]}, "align_alpha_beta": [last["res"].alpha, last["res"].beta],
            "seed_successes": int(last["ns"]),
            "roofline": roofline("sparse_align_kernel<4,*,true> (single stereo bundle: latency-bound by design)", kms.value, alg, "frame-stereo:default"),
            "cpu_baseline": cpu}


def bench_detect(args, ctx, dist, rank, world, dev, comm_dev=None):
    """Keyframe feature detection: FastGradDetector::detect on a 752x480 5-level pyramid
    (FAST-10 on levels 0..2, best corner per 30-px cell, edgelets on level 1 in the free cells), one keyframe
    per call with the pyramid resident in HBM, results on the host."""
    cam = synth.Camera.euroc_like(752, 480)
    B = args.problems or 16
    scenes = [synth.make_align_scene(du.problem_seed(rank, 300 + i), n_features=8, cam=cam) for i in range(B)]
    frames = [ctx.build_pyramid(sc.img_ref, 5) for sc in scenes]
    opt = capi.default_detector_options()
    last = {}

    def step():
        n = 0
        for fr in frames:
            last["d"] = ctx.detect_features(opt, fr, cam.width, cam.height)
            n += len(last["d"]["score"])
        last["n"] = n
        return None, misc_kernel_ms(ctx)

    elapsed, kms, _ = timed_steps(ctx, dist, world, dev, step, args.steps, args.warmup)
    elapsed, total = du.combine(dist, world, elapsed, B, comm_dev)
    # per keyframe: every level's image read once, its u8 score map written and read once; level 1 read once more,
    # its float magnitude map written and read once
    px = [(cam.width >> l) * (cam.height >> l) for l in range(5)]
    alg = sum(3 * px[l] for l in range(opt.min_level, opt.max_level + 1)) + 9 * px[1]
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc  # test infrastructure: the timed CPU baseline only
        orc.build(fast=True)
        t_cpu, n_cpu = 0.0, 0
        for sc in scenes:
            levels = orc.create_img_pyramid(sc.img_ref, 5, fast=True)
            t0 = time.perf_counter()
            d = orc.detect_features(opt, levels, fast=True)
            t_cpu += time.perf_counter() - t0
            n_cpu += 1
            if t_cpu > 10.0:
                break
        assert np.array_equal(d["px"], last["d"]["px"]) if n_cpu == B else True
        cpu = {"value": n_cpu / t_cpu, "unit": "keyframes/s", "cores": 1, "kind": "port",
               "sample": "%d keyframes of the benchmark (oracle detector: definitional FAST-10, gcc -O3 -march=native, "
                         "1 thread, %.1f s; the reference uses the SSE2 decision-tree detector, faster than this port)" % (n_cpu, t_cpu)}
    if rank != 0:
        return None
    return {"metric": "keyframes/s (FastGradDetector::detect, 752x480, FAST-10 levels 0..2 + edgelets, 30-px grid)",
            "value": total * args.steps / elapsed, "unit": "keyframes/s", "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_frame": 1e3 * elapsed / args.steps / B, "dtype": "u8+i32+f32",
            "config": {"workload": "detector: %d keyframes per step, 752x480, pyramid resident in HBM, one blocking call each" % B,
                       "keyframes_per_step": B},
            "kernel_ms": kms, "features_per_keyframe": last["n"] / float(B),
            "roofline": roofline("fast_score/select + edge_score/select/angle kernels of one keyframe (latency-bound: 9 launches)",
                                 kms, alg, "detect:default"),
            "cpu_baseline": cpu}


def bench_pose(args, ctx, dist, rank, world, dev, comm_dev=None):
    """PoseOptimizer::run for B frame bundles of 180 features per call (multi-stream batch),
    unit-plane error, 10 % gross outliers; host arrays in, poses + outlier flags out."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
    import pose_helpers as ph
    B = args.problems or 2048
    n_unique = 64
    scenes = [ph.make_pose_scene(du.problem_seed(rank, 500 + i), n=180) for i in range(n_unique)]
    opt = capi.default_pose_options(scenes[0]["cam"])
    built = [fe.make_pose_problem(sc["cams"], sc["T_imu_world_init"]) for sc in scenes]
    problems = [built[i % n_unique][0] for i in range(B)]
    arr = (capi.svoh_pose_problem * B)(*problems)
    res = (capi.svoh_pose_result * B)()
    n_meas = [0]

    def step():
        ctx._check(ctx.lib.svoh_optimize_pose_batch(ctx.h, ctypes.byref(opt), B, arr, res))
        return None, misc_kernel_ms(ctx)

    # host arrays staged per call first (PCIe-inclusive), then value: the per-feature arrays resident in HBM, used in place
    h_steps = max(2, args.steps // 4)
    h_elapsed, _hk, _ = timed_steps(ctx, dist, world, dev, step, h_steps, 1)
    host_rate = B * h_steps / h_elapsed
    cat = {k: [] for k in ("px", "f", "grad", "level", "type", "xyz_world", "usable")}
    for i in range(B):
        for a in built[i % n_unique][1]:
            for k in cat:
                cat[k].append(a[k].ravel())
    t = {k: torch.from_numpy(np.concatenate(v)).to(dev) for k, v in cat.items()}
    n_total = t["level"].numel()
    t["outlier"] = torch.zeros(n_total, dtype=torch.uint8, device=dev)
    t["final_error"] = torch.zeros(n_total, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    parr = capi.svoh_pose_packed_arrays()
    parr.n_features_total = n_total
    for k in ("px", "f", "grad", "level", "type", "xyz_world", "usable", "outlier", "final_error"):
        setattr(parr, k, t[k].data_ptr())
    res_h = [(r.iters, r.n_meas, fe.se3_to_numpy(r.T_imu_world).copy()) for r in res]

    def step_dev():
        ctx._check(ctx.lib.svoh_optimize_pose_batch_packed(ctx.h, ctypes.byref(opt), B, arr, ctypes.byref(parr), res))
        return None, misc_kernel_ms(ctx)

    elapsed, kms, _ = timed_steps(ctx, dist, world, dev, step_dev, args.steps, args.warmup)
    assert all(r.iters == w[0] and r.n_meas == w[1] and np.array_equal(fe.se3_to_numpy(r.T_imu_world), w[2]) for r, w in zip(res, res_h))
    elapsed, total = du.combine(dist, world, elapsed, B, comm_dev)
    iters = sum(r.iters for r in res)
    meas = sum(r.n_meas for r in res)
    # per measurement and iteration: px 16 + f 24 + grad 16 + xyz 24 + level 4 + type 1 + usable 1 bytes
    alg = 86 * sum(r.n_meas * (r.iters + 2) for r in res)
    t1 = time.perf_counter()
    for _ in range(50):
        ctx.optimize_pose(opt, [problems[0]])
    single_ms = (time.perf_counter() - t1) / 50 * 1e3
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc  # test infrastructure: the timed CPU baseline only
        orc.build(fast=True)
        t_cpu, n_cpu = 0.0, 0
        while t_cpu < 5.0 and n_cpu < B:
            t0 = time.perf_counter()
            ro = orc.optimize_pose(opt, problems[n_cpu], fast=True)
            t_cpu += time.perf_counter() - t0
            assert ro.iters == res[n_cpu].iters
            n_cpu += 1
        cpu = {"value": n_cpu / t_cpu, "unit": "bundles/s", "cores": 1, "kind": "port",
               "sample": "%d bundles of the benchmark (oracle PoseOptimizer::run, gcc -O3 -march=native, 1 thread, %.1f s)" % (n_cpu, t_cpu)}
    if rank != 0:
        return None
    return {"metric": "frame bundles/s (PoseOptimizer::run, 180 features, unit-plane error, Tukey + MAD)",
            "value": total * args.steps / elapsed, "unit": "bundles/s", "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_frame": 1e3 * elapsed / args.steps / B, "dtype": "f64",
            "config": {"workload": "pose: %d bundles x 180 features per call, feature arrays resident in HBM (svoh_optimize_pose_batch_packed)" % B,
                       "bundles_per_step": B},
            "host_staged_bundles_per_s": host_rate,   # the same bundles through svoh_optimize_pose_batch with host arrays (PCIe-inclusive)
            "kernel_ms": kms, "mean_iterations": iters / float(B), "measurements_per_bundle": meas / float(B),
            "single_bundle_call_ms": single_ms,
            "roofline": roofline("pose_optimize_kernel", kms, alg, "pose:default" if not args.problems else "pose:B%d" % B),
            "cpu_baseline": cpu}


def bench_align_c4(args, ctx, dist, rank, world, dev, comm_dev=None):
    """BASELINE config 4's alignment (SURVEY.md Appendix A, C4 column): a stereo bundle per problem -- two cameras
    with different intrinsics (pinhole and EuRoC radtan) behind one rig motion, `--features` patches per camera --
    illumination gain and offset estimated (8-DoF), rotation prior with lambda_rot 0.5, levels 4..2, 4x4 patches."""
    # (2048 bundles per step since the end of round 6, as the headline's 4096 pairs: 512 were ONE problem per resident workgroup -- the step ended with its
    # slowest problem: 1.25 ms per 512 bundles against 1.00 at 2048, gpurun_out of the round: profiles/r06_align_batch_size_sweep.txt)
    P, N, B = args.patch, args.features, (args.problems or 2048)
    opt = capi.default_align_options(max_level=4, min_level=2, patch_size=P, estimate_illumination_gain=1,
                                     estimate_illumination_offset=1)
    cams = [synth.Camera.test_camera(), synth.Camera.euroc_like()]
    scenes = [[synth.make_align_scene(du.problem_seed(rank, i), n_features=N, patch_size=P, cam=cam, max_level=4,
                                      render_images=False, gain=1.03, offset=2.0) for cam in cams] for i in range(B)]
    frames, imgs_keep = [], []
    for c, cam in enumerate(cams):   # one rendering + pyramid batch per camera model
        poses, planes, texs = [], [], []
        for pair in scenes:
            sc = pair[c]
            poses += [sc.T_w_ref, sc.T_w_cur]; planes += [sc.plane] * 2; texs += [sc.tex] * 2
        imgs = synth.render_batch_torch(cam, poses, planes, texs, dev, gains=[1.0, 1.03] * B, offsets=[0.0, 2.0] * B)
        torch.cuda.synchronize()
        frames.append(ctx.build_pyramid_batch_device(imgs.data_ptr(), cam.width * cam.height, 2 * B, cam.width, cam.height,
                                                     cam.width, 5))
        imgs_keep.append(imgs)
    ctx.synchronize()
    flat = [sc for pair in scenes for sc in pair]
    px = torch.from_numpy(np.concatenate([s.px for s in flat])).to(dev)
    f = torch.from_numpy(np.concatenate([s.f for s in flat])).to(dev)
    pw = torch.from_numpy(np.concatenate([s.pos_world for s in flat])).to(dev)
    fl = torch.from_numpy(np.concatenate([s.flags for s in flat])).to(dev)
    items, priors, off = [], [], 0
    for i, pair in enumerate(scenes):
        cams_i = []
        for c, sc in enumerate(pair):
            dp = dict(px=px.data_ptr() + 16 * off, f=f.data_ptr() + 24 * off, pos_world=pw.data_ptr() + 24 * off,
                      flags=fl.data_ptr() + off)
            cams_i.append((sc, frames[c][2 * i], frames[c][2 * i + 1], dp))
            off += sc.n_features
        items.append(cams_i)
        pr = capi.svoh_align_prior()
        pr.have_prior = 1
        pr.T_prior = fe._se3(pair[0].T_icur_iref_gt)    # an IMU-integrated rotation prior: the true relative pose
        pr.lambda_rot = 0.5
        priors.append(pr)
    problems, keep = fe.make_align_problems(items, prior=priors)

    def step():
        ctx.sparse_align_enqueue(opt, problems)
        return None, 0.0

    for _ in range(max(1, args.warmup)):
        res = ctx.sparse_align(opt, problems)
    elapsed, _, _ = timed_steps(ctx, dist, world, dev, step, args.steps, 0)
    res = ctx.sparse_align_fetch(len(problems))
    n_hist = ctypes.c_int()
    hist = (ctypes.c_float * 32)()
    ctx._check(ctx.lib.svoh_sparse_align_kernel_ms_history(ctx.h, min(32, args.steps), hist, ctypes.byref(n_hist)))
    kms = sum(hist[i] for i in range(n_hist.value)) / max(1, n_hist.value)
    n_sel = sum(r.n_fts_to_track for r in res)
    elapsed, total = du.combine(dist, world, elapsed, n_sel, comm_dev)
    patch_iters = sum(r.n_patch_iters for r in res)
    errs = [synth.se3_error(synth.SE3.from7(fe.se3_to_numpy(r.T_icur_iref)), pair[0].T_icur_iref_gt) for r, pair in zip(res, scenes)]
    if rank != 0:
        return None
    alg = algorithmic_bytes(P, 8, patch_iters, n_sel * 3)
    return {"metric": "aligned patches/s, stereo bundles with illumination terms and rotation prior (BASELINE config 4's alignment)",
            "value": total * args.steps / elapsed, "unit": "aligned patches/s", "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_frame": 1e3 * elapsed / args.steps / B, "dtype": "f64",
            "config": {"workload": "C4-synth alignment: %d stereo bundles per GPU per step, 2 cameras x %d patches x %dx%d, "
                                   "levels 4..2, SE3 + gain + offset (8-DoF), rotation prior 0.5, inputs resident in HBM" % (B, N, P, P),
                       "bundles_per_gpu": B, "patches_per_camera": N, "patch_size": P, "levels": [4, 2],
                       "parallelism": "bundles sharded x%d, no collective" % world},
            "patch_iterations_per_step": patch_iters, "patch_iterations_per_s": patch_iters * world * args.steps / elapsed,
            "kernel_ms": kms, "solver_failures": sum(1 for r in res if r.status != 0),
            "illumination_median": {"alpha": float(np.median([r.alpha for r in res])), "beta": float(np.median([r.beta for r in res]))},
            "pose_err_vs_gt": {"rot_rad_median": float(np.median([e[0] for e in errs])), "trans_m_median": float(np.median([e[1] for e in errs]))},
            "roofline": roofline("sparse_align_kernel<%d,256,true,false>" % P, kms, alg,
                                 "align-c4:default" if (not args.problems and N == 2000 and P == 4) else "align-c4:B%d:N%d:P%d" % (B, N, P)),
            "cpu_baseline": None}


def bench_align_split(args, ctx, dist, rank, world, dev, comm_dev=None):
    """SURVEY.md 8(e) second row / BASELINE config 5's collective: ONE frame pair, its N patches split over the
    ranks, every Gauss-Newton iteration = partial normal equations per rank -> all-reduce of 74 doubles (RCCL) ->
    identical update on every rank.  Strong scaling (the problem is fixed); latency-bound by construction, the
    per-iteration breakdown says where the crossover against the resident single-GPU kernel lies."""
    import torch.distributed as tdist
    from svo_pro_universal_amd import split_align
    P, N = args.patch, args.features
    opt = capi.default_align_options(max_level=args.max_level, min_level=args.min_level, patch_size=P)
    cam = synth.Camera.test_camera()
    sc = synth.make_align_scene(du.problem_seed(0, 900), n_features=N, patch_size=P, cam=cam, max_level=args.max_level,
                                render_images=False)   # the same scene on every rank
    imgs = synth.render_batch_torch(cam, [sc.T_w_ref, sc.T_w_cur], [sc.plane] * 2, [sc.tex] * 2, dev)
    torch.cuda.synchronize()
    frames = ctx.build_pyramid_batch_device(imgs.data_ptr(), cam.width * cam.height, 2, cam.width, cam.height,
                                            cam.width, args.max_level + 1)
    ctx.synchronize()
    px, f = torch.from_numpy(sc.px).to(dev), torch.from_numpy(sc.f).to(dev)
    pw, fl = torch.from_numpy(sc.pos_world).to(dev), torch.from_numpy(sc.flags).to(dev)

    def problem_of(lo, hi):
        import copy
        share = copy.copy(sc)
        share.n_features = hi - lo
        dp = dict(px=px.data_ptr() + 16 * lo, f=f.data_ptr() + 24 * lo, pos_world=pw.data_ptr() + 24 * lo,
                  flags=fl.data_ptr() + lo)
        return fe.make_align_problems([[(share, frames[0], frames[1], dp)]])

    lo, hi = du.shard_range(sc.n_features, rank, world)
    mine, keep_m = problem_of(lo, hi)
    whole, keep_w = problem_of(0, sc.n_features)
    d_state = torch.zeros(ctypes.sizeof(capi.svoh_align_gn_state) // 8, dtype=torch.float64, device=dev)
    d_sums = torch.zeros(capi.SVOH_ALIGN_SUMS_DOUBLES, dtype=torch.float64, device=dev)
    own_group = False
    if world == 1 and not tdist.is_initialized():
        # a one-rank RCCL group, so that the all-reduce's launch floor is part of the single-GPU number too
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29731")
        tdist.init_process_group(os.environ.get("SVOH_BENCH_BACKEND", "nccl"), rank=0, world_size=1)
        own_group = True
    t_part, t_red, t_upd = [0.0], [0.0], [0.0]

    def all_reduce():
        t0 = time.perf_counter()
        if tdist.get_backend() == "gloo":   # one-GPU rehearsal only: gloo sums on the host
            h = d_sums.cpu()
            tdist.all_reduce(h)
            d_sums.copy_(h)
        else:
            tdist.all_reduce(d_sums)
        torch.cuda.synchronize()
        t_red[0] += time.perf_counter() - t0

    def one_run():
        ctx.split_init(mine[0], d_state.data_ptr())

        def partial(level):
            t0 = time.perf_counter()
            ctx.partial_sums(opt, mine[0], level, d_state.data_ptr(), d_sums.data_ptr())
            ctx.synchronize()
            t_part[0] += time.perf_counter() - t0

        def update(level, it):
            t0 = time.perf_counter()
            st = ctx.gn_update(opt, mine[0], level, it, d_sums.data_ptr(), d_state.data_ptr())
            t_upd[0] += time.perf_counter() - t0
            return st

        return split_align.gauss_newton_split(opt.max_level, opt.min_level, opt.max_iter, partial, all_reduce, update), 0.0

    for _ in range(max(1, args.warmup)):   # also creates the communicator: not part of the per-iteration figures
        one_run()
    t_part[0] = t_red[0] = t_upd[0] = 0.0
    elapsed, _, res = timed_steps(ctx, dist, world, dev, one_run, args.steps, 0)
    n_runs = args.steps
    iters = res.n_evaluations
    n_meas_last = res.n_meas[opt.min_level]
    elapsed, _ = du.combine(dist, world, elapsed, 0, comm_dev)
    # the same problem, unsplit, in the resident kernel on this GPU (what the split has to beat)
    kms = ctypes.c_float()
    wres = None
    for _ in range(3):
        wres = ctx.sparse_align(opt, whole)[0]
    t0 = time.perf_counter()
    for _ in range(10):
        wres = ctx.sparse_align(opt, whole)[0]
    resident_call_ms = (time.perf_counter() - t0) / 10 * 1e3
    ctx.lib.svoh_sparse_align_last_kernel_ms(ctx.h, ctypes.byref(kms))
    Ts = synth.SE3.from7(fe.se3_to_numpy(res.state.T_icur_iref))
    err = synth.se3_error(Ts, sc.T_icur_iref_gt)
    dvs = synth.se3_error(Ts, synth.SE3.from7(fe.se3_to_numpy(wres.T_icur_iref)))
    if own_group:
        tdist.destroy_process_group()
    if rank != 0:
        return None
    per_it = lambda t: 1e6 * t[0] / (n_runs * iters)
    ms_run = 1e3 * elapsed / args.steps
    D = 6
    patch_iters = sum(res.iters[l] * res.n_meas[l] // (P * P) for l in range(capi.SVOH_MAX_LEVELS))
    alg = algorithmic_bytes(P, D, patch_iters, 0)
    return {"metric": "aligned patches/s, one frame pair split over the GPUs (all-reduce of the normal equations per iteration)",
            "value": wres.n_fts_to_track / (ms_run * 1e-3), "unit": "aligned patches/s", "ms_per_step": ms_run,
            "ms_per_frame": ms_run, "dtype": "f64", "scaling": "strong",
            "config": {"workload": "SparseImgAlign, ONE synthetic 640x480 frame pair, %d patches x %dx%d, levels %d..%d, "
                                   "patches split over %d rank(s), 74-double all-reduce per iteration" %
                                   (sc.n_features, P, P, args.max_level, args.min_level, world),
                       "patches": sc.n_features, "patch_size": P, "levels": [args.max_level, args.min_level],
                       "parallelism": "patch-split x%d, RCCL all-reduce 592 B per iteration" % world},
            "iterations": iters, "iterations_per_level": res.iters[:args.max_level + 1],
            "per_iteration_us": {"partial_sums": per_it(t_part), "all_reduce": per_it(t_red), "gn_update": per_it(t_upd)},
            "resident_single_gpu": {"kernel_ms": kms.value, "blocking_call_ms": resident_call_ms,
                                    "iterations": sum(wres.iters)},
            "pose_err_vs_gt": {"rot_rad": err[0], "trans_m": err[1]},
            "pose_diff_vs_resident": {"rot_rad": dvs[0], "trans_m": dvs[1]},
            "roofline": roofline("sparse_align_kernel (evaluate mode) + all-reduce + align_gn_update_kernel, whole run (per rank)",
                                 ms_run, alg / world, "align-split:N%d:W%d" % (args.features, world)),
            "cpu_baseline": None}


_JSON_FD = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner to stdout when a communicator is
    created, and libraries may add their own chatter: send everything to stderr and keep the real stdout for
    the result line."""
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)


def emit(obj):
    obj.setdefault("host_cpu_slice_rank0", _CPU_SLICE)
    sys.stdout.flush()
    os.write(_JSON_FD if _JSON_FD is not None else 1, (json.dumps(obj) + "\n").encode())


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--problems", type=int, default=0, help="frame pairs per GPU per step (default: per workload)")
    ap.add_argument("--workload", default="align",
                    choices=["align", "klt", "seeds", "frame", "detect", "pose", "align-split", "align-c4", "stereo",
                             "launch-check"],
                    help="align = the headline SparseImgAlign config (default); klt / seeds / stereo = the other "
                         "hot-path rows; frame = the whole per-frame chain at EuRoC mono sizes, one frame at a time "
                         "(latency); launch-check = the N-rank launch, barrier and MAX/SUM plumbing with an empty step")
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--patch", type=int, default=4)
    ap.add_argument("--min-level", type=int, default=0)
    ap.add_argument("--max-level", type=int, default=4)
    ap.add_argument("--own-align-classes", action="store_true", help="with --stream-mix: every alignment problem in the launch geometry that is fastest for it alone (the library's default) instead of shared classes")
    ap.add_argument("--stereo", action="store_true", help="with --workload frame: the stereo pair chain (BASELINE config 3)")
    ap.add_argument("--streams", type=int, default=0, help="with --workload frame: S camera streams per GPU through the whole chain in lock step (BASELINE config 5)")
    ap.add_argument("--stream-mix", action="store_true", help="with --streams: streams that DIFFER -- every stream its own walk over the sequence (start, direction, stride), "
                    "keyframe period 4..9, feature budget 120 / 180 / 240, one stream in eight at half the frame rate (FrontendLockstep's per-stream options)")
    ap.add_argument("--stream-groups", type=int, default=0, help="lock-step groups per GPU (default: chosen from S and the host-thread budget)")
    ap.add_argument("--stream-workers", type=int, default=0, help="host threads per lock-step group")
    ap.add_argument("--whole-sets", action="store_true", help="with --workload seeds: the timed leg is the batch over RESIDENT keyframe columns "
                    "(svoh_features_upload once; SVOH_BATCH_WHOLE_SETS: tile order from the upload, no per-frame binning)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the 8x8-patch leg that the default line carries as `secondary`")
    return ap.parse_args(argv)


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_workers(n, argv):
    """`bench.py --gpus N` started directly (no WORLD_SIZE in the environment): start N fresh worker processes
    through torch.distributed.run, one per GPU, and hand their exit code back.  This process has made no GPU call
    and makes none; it is never replaced by exec (that is fatal on the GPU boxes once a process has touched the GPU,
    and a child is just as good).  Rank 0's JSON line goes straight to the inherited stdout."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    print("bench.py: launching %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


_CPU_SLICE = None
_CPUS_BEFORE_PIN = None   # CPUs the process could run on before pin_rank_to_its_cores narrowed it
_PINNED = False


def pin_rank_to_its_cores(local_rank, world):
    """One rank per GPU on one node: rank r takes the r-th of `world` equal slices of the CPUs this process may run on
    (os.sched_setaffinity, before any GPU call and before any thread is started), so that the host-bound workloads --
    the per-frame chains, whose time is host threads waiting on small launches -- do not share cores between ranks.
    SVOH_BENCH_PIN=0 leaves the affinity alone.  Returns (and remembers for the line) the slice as "first-last (n)"."""
    global _CPU_SLICE, _CPUS_BEFORE_PIN, _PINNED
    cpus = sorted(os.sched_getaffinity(0))
    _CPUS_BEFORE_PIN = len(cpus)
    if world > 1 and os.environ.get("SVOH_BENCH_PIN", "1") != "0" and len(cpus) >= world:
        _PINNED = True
        per = len(cpus) // world
        mine = cpus[local_rank % world * per:(local_rank % world + 1) * per]
        os.sched_setaffinity(0, mine)
        cpus = mine
    _CPU_SLICE = "%d-%d (%d)" % (cpus[0], cpus[-1], len(cpus))
    return _CPU_SLICE


def bench_launch_check(args, dist, rank, world, comm_dev):
    """The N-rank plumbing alone: W+K empty steps between the same barriers, MAX of the elapsed time and SUM of the
    units over the ranks.  Needs no GPU with SVOH_BENCH_BACKEND=gloo (the CPU test of the launcher)."""
    def barrier():
        if world > 1:
            dist.barrier()
    for _ in range(args.warmup):
        pass
    barrier()
    t0 = time.perf_counter()
    units = 0
    for _ in range(args.steps):
        units += 1000 + rank
    barrier()
    elapsed, total = du.combine(dist, world, time.perf_counter() - t0, units, comm_dev)
    ranks_seen = du.ranks_in_collective(dist, world, comm_dev)
    return {"metric": "launch check (empty steps)", "value": total / max(elapsed, 1e-9), "unit": "units/s",
            "ms_per_step": 1e3 * elapsed / max(1, args.steps), "dtype": "none", "units_total": total,
            "ranks_in_collective": ranks_seen, "backend": dist.get_backend() if world > 1 else None,
            "config": {"workload": "launch-check"}}


def main(argv=None):
    args = parse_args(argv)
    rank, local_rank, world = du.env_world()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under torchrun: be the launcher (before anything touches the GPU), relay the workers' exit code
        raise SystemExit(launch_workers(args.gpus, sys.argv[1:] if argv is None else argv))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a different n_gpus than asked for"
                         % (args.gpus, world))
    claim_stdout()
    cpu_slice = pin_rank_to_its_cores(local_rank, world)

    # Rehearsal knobs for a one-GPU box (never set by the driver): all ranks on cuda:0 and gloo for the
    # barrier / MAX / SUM, because RCCL refuses two ranks on one device.
    backend = os.environ.get("SVOH_BENCH_BACKEND", "nccl")
    if os.environ.get("SVOH_BENCH_ONE_DEVICE", "0") == "1":
        local_rank = 0
    dist = None
    if args.workload == "launch-check" and backend == "gloo":     # CPU rehearsal of the launcher: no GPU call at all
        if world > 1:
            dist = du.init(backend, rank, world)
        out = bench_launch_check(args, dist, rank, world, None)
        if rank == 0:
            out.update({"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
                        "scaling": "weak", "vs_baseline": None, "data": "synthetic"})
            emit(out)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if world > 1:
        dist = du.init(backend, rank, world, torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_dev = dev if backend == "nccl" else None

    ctx = fe.Context(local_rank)
    if args.workload != "align":
        out = {"klt": bench_klt, "seeds": bench_seeds,
               "frame": (bench_frame_stereo_streams if args.stereo else bench_frame_streams) if args.streams > 0 else (bench_frame_stereo if args.stereo else bench_frame), "detect": bench_detect,
               "pose": bench_pose,
               "align-split": bench_align_split, "align-c4": bench_align_c4, "stereo": bench_stereo,
               "launch-check": lambda a, c, d, r, w, dv, cd: bench_launch_check(a, d, r, w, cd)}[args.workload](
            args, ctx, dist, rank, world, dev, comm_dev)
        ranks_seen = du.ranks_in_collective(dist, world, comm_dev)   # a collective: every rank calls it
        if rank == 0:
            out.setdefault("scaling", "weak")
            out.update({"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
                        "vs_baseline": None, "data": "synthetic"})
            out.setdefault("ranks_in_collective", ranks_seen)
            out.setdefault("backend", dist.get_backend() if world > 1 else None)
            emit(out)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    # 4096 frame pairs per step since the end of round 6 (rounds 2 - 5: 1024).  The kernel's 512 resident workgroups pull problems from a queue; 1024 problems
    # are exactly two each, so that step ends with the slowest pair and measures a tail, not the kernel's rate (profiles/r06_align_batch_size_sweep.txt: 1.17 ms
    # per 1024 problems at 1024, 1.04 at 2048, 0.94 at 4096).  The 1024-pair step stays in the line as `step_of_1024`.
    P, N, B = args.patch, args.features, (args.problems or 4096)
    out, shared = run_align(args, ctx, dist, rank, world, dev, comm_dev, P, N, B, None, not args.no_cpu_baseline and world == 1)
    # the 8x8-patch configuration north_star quotes its roofline target on, carried in the same line (same frame
    # pairs, same launch shape; its own features because the border margin depends on the patch size)
    if P == 4 and not args.no_secondary and not args.problems and N == 2000 and args.min_level == 0 and args.max_level == 4:
        sec, _ = run_align(args, ctx, dist, rank, world, dev, comm_dev, 8, N, B, shared, False, steps=max(3, args.steps // 2))
        if rank == 0:
            out["secondary"] = {k: sec[k] for k in ("value", "unit", "kernel_ms", "kernel_ms_min", "kernel_ms_max", "ms_per_step", "patch_iterations_per_step",
                                                    "patch_iterations_per_s", "solver_failures", "roofline")}
            out["secondary"]["config"] = sec["config"]
            out["secondary"]["steps"] = max(3, args.steps // 2)
    # ... and the step of rounds 2 - 5 (1024 frame pairs: two problems per resident workgroup), for comparison with their numbers
    if P == 4 and not args.no_secondary and not args.problems and N == 2000 and args.min_level == 0 and args.max_level == 4 and world == 1:
        small, small_shared = run_align(args, ctx, dist, rank, world, dev, comm_dev, 4, N, 1024, None, False, steps=args.steps)
        out["step_of_1024"] = {"frame_pairs_per_step": 1024, "kernel_ms": small["kernel_ms"], "kernel_ms_min": small["kernel_ms_min"], "kernel_ms_max": small["kernel_ms_max"],
                               "value": small["value"], "unit": small["unit"], "ms_per_step": small["ms_per_step"], "steps": args.steps,
                               "roofline": small["roofline"],
                               "note": "the benchmark's step until round 6: the launch ends with the slowest pair of problems (a tail), all workgroups start together"}
        small8, _ = run_align(args, ctx, dist, rank, world, dev, comm_dev, 8, N, 1024, small_shared, False, steps=max(3, args.steps // 2))
        out["step_of_1024"]["patch_8"] = {"kernel_ms": small8["kernel_ms"], "value": small8["value"], "ms_per_step": small8["ms_per_step"], "roofline": small8["roofline"]}
    ranks_seen = du.ranks_in_collective(dist, world, comm_dev)   # a collective: every rank calls it
    if rank == 0:
        out["ranks_in_collective"] = ranks_seen
        out["backend"] = dist.get_backend() if world > 1 else None
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_align(args, ctx, dist, rank, world, dev, comm_dev, P, N, B, shared, with_cpu, steps=None):
    """The headline workload at patch size P: W warm-up + K timed steps of one launch over B frame pairs."""
    steps = steps or args.steps
    opt = capi.default_align_options(max_level=args.max_level, min_level=args.min_level, patch_size=P)
    problems, scenes, imgs, keep = build_problems(ctx, dev, rank, B, N, P, args.max_level, reuse=shared)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    kern_ms = ctypes.c_float()
    for _ in range(args.warmup):
        res = ctx.sparse_align(opt, problems)
    barrier()
    # K steps queued back to back on the context stream: each enqueue builds and uploads its launch descriptors
    # while the previous step's kernel runs, and every step's results are copied to pinned host memory behind its
    # kernel; a fetch hands out the last step's (one every 32 steps: the library keeps 32 event pairs and a bounded
    # number of queued results).  Kernel times come from the library's per-launch HIP events.
    t0 = time.perf_counter()
    kernel_ms_sum, n_timed = 0.0, 0
    kernel_ms_all = []
    hist = (ctypes.c_float * 32)()
    n_hist = ctypes.c_int()
    done = 0
    while done < steps:
        chunk = min(32, steps - done)
        for _ in range(chunk):
            ctx.sparse_align_enqueue(opt, problems)
        res = ctx.sparse_align_fetch(len(problems))
        done += chunk
        if done >= steps:
            barrier()
            elapsed = time.perf_counter() - t0
        ctx._check(ctx.lib.svoh_sparse_align_kernel_ms_history(ctx.h, chunk, hist, ctypes.byref(n_hist)))
        kernel_ms_sum += sum(hist[i] for i in range(n_hist.value))
        kernel_ms_all += [float(hist[i]) for i in range(n_hist.value)]
        n_timed += n_hist.value
    kernel_ms_sum *= steps / float(max(1, n_timed))

    # latency of ONE frame pair of the same workload (outside the timed region): what a single camera stream sees
    one_arr = (capi.svoh_align_problem * 1)(problems[0])
    lat_call, lat_kern = [], []
    for k in range(12):
        t1 = time.perf_counter()
        ctx.sparse_align(opt, one_arr)
        t2 = time.perf_counter()
        ctx.lib.svoh_sparse_align_last_kernel_ms(ctx.h, ctypes.byref(kern_ms))
        if k >= 2:
            lat_call.append((t2 - t1) * 1e3)
            lat_kern.append(kern_ms.value)

    n_sel = sum(r.n_fts_to_track for r in res)
    # whole-job numbers: MAX of the elapsed time, SUM of the patches all ranks aligned
    elapsed, patches_total = du.combine(dist, world, elapsed, n_sel, comm_dev)
    patch_iters = sum(r.n_patch_iters for r in res)
    n_levels = args.max_level - args.min_level + 1
    n_bad = sum(1 for r in res if r.status != 0)
    # accuracy against the synthetic ground truth (informational)
    errs = [synth.se3_error(synth.SE3.from7(fe.se3_to_numpy(r.T_icur_iref)), sc.T_icur_iref_gt)
            for r, sc in zip(res, scenes)]
    value = patches_total * steps / elapsed
    out = None
    if rank == 0:
        kernel_ms = kernel_ms_sum / steps
        D = 6
        alg = algorithmic_bytes(P, D, patch_iters, n_sel * n_levels)
        out = {
            "metric": "aligned patches/sec + ms/frame, EuRoC 640x480 mono, 1/2/4/8 MI355X",
            "value": value,
            "unit": "aligned patches/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / steps,
            # batch-amortised: one step aligns B frame pairs at once; the latency ONE frame pair sees is
            # one_frame_pair_latency_ms below (and the whole per-frame chain: bench.py --workload frame)
            "ms_per_frame": 1e3 * elapsed / steps / B,
            "ms_per_frame_is": "ms_per_step / frame pairs per step (throughput of a multi-stream batch, not a latency)",
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "SparseImgAlign only: synthetic 640x480, %d patches x %dx%d, levels %d..%d, SE3 6-DoF, "
                            "GN <=10 it/level, eps 5e-4; %d independent frame pairs per GPU per step, inputs "
                            "resident in HBM; every step solves the SAME frame pairs again from the same initial pose "
                            "(%d MB of pyramids per GPU, far beyond the 256 MB Infinity Cache; %d problems per resident workgroup of the persistent kernel)"
                            % (N, P, P, args.max_level, args.min_level, B, B * 2 * 409200 // 1000000, max(1, B // 512)),
                "frame_pairs_per_gpu": B, "patches_per_frame": N, "patch_size": P,
                "levels": [args.max_level, args.min_level], "parallelism": "frame-pairs sharded x%d, no collective" % world,
            },
            "patch_iterations_per_step": patch_iters,
            "patch_iterations_per_s": patch_iters * world * steps / elapsed,
            "kernel_ms": kernel_ms,
            "kernel_ms_min": min(kernel_ms_all) if kernel_ms_all else None,
            "kernel_ms_max": max(kernel_ms_all) if kernel_ms_all else None,
            "kernel_ms_launches_timed": len(kernel_ms_all),
            # median 1e-4 m: what Gauss-Newton on bilinearly rendered 640x480 scenes resolves; the tail (max) belongs to
            # scenes with little texture under the patches -- the algorithm's answer, not the kernel's: GPU and oracle
            # agree to 1e-15 on every one of them (tests/test_sparse_align_gpu.py)
            "pose_err_vs_gt": {"rot_rad_median": float(np.median([e[0] for e in errs])),
                               "trans_m_median": float(np.median([e[1] for e in errs])),
                               "rot_rad_max": float(np.max([e[0] for e in errs])),
                               "trans_m_max": float(np.max([e[1] for e in errs]))},
            "solver_failures": n_bad,
            "one_frame_pair_latency_ms": {"blocking_call": float(np.median(lat_call)), "kernel": float(np.median(lat_kern)),
                                          "note": "one problem of the same workload, several workgroups per problem"},
            "roofline": roofline("sparse_align_kernel<%d,256,false,false>" % P, kernel_ms, alg,
                                 "align:B%d:N%d:P%d:L%d-%d" % (B, N, P, args.max_level, args.min_level),
                                 patch_iterations=patch_iters,
                                 fp64_note="fp64 vector peak measured %.1f TFLOP/s (tools/svoh_microbench)" % FP64_PEAK_TFLOPS),
        }
        if with_cpu:
            out["cpu_baseline"] = cpu_baseline(scenes, imgs, opt, args.max_level)
    return out, (imgs, keep[5])


if __name__ == "__main__":
    main()
