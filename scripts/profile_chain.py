"""rocprofv3 kernel statistics of the per-frame chain (tools/svoh_mini_frontend on the synthetic EuRoC-layout sequence of
tests/test_mini_frontend_gpu.py): which kernels a frame launches and how long they run.  Writes
gpurun_out/profiles/<round>_chain_kernel_stats.csv.  Run on the GPU box:  python scripts/profile_chain.py r02"""
import glob, os, pathlib, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_mini_frontend_gpu as t

rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
tmp = pathlib.Path(tempfile.mkdtemp(prefix="chain_", dir="/tmp"))
if os.environ.get("TOOL", "mono") == "stereo":   # tools/svoh_mini_stereo on the stereo sequence of tests/test_mini_stereo_gpu.py
    import test_mini_stereo_gpu as ts
    n_frames = 30
    cmd, out_dir, poses, stamps = ts.make_stereo_dataset(tmp, n_frames)
    cmd = cmd + [str(n_frames), "8", "0.5"]
else:
    cmd, out_dir, poses, stamps, n_frames = t.make_dataset(tmp)
# LOCKSTEP="S:W:G[:laps]": the lock-step mode of the tool (host/svo_hip_lockstep.h) instead of one stream per thread
if os.environ.get("LOCKSTEP"):
    ls_ = os.environ["LOCKSTEP"].split(":")
    cmd = cmd + [str(n_frames), "8", ls_[0], "lockstep", ls_[1], ls_[2]] + ([ls_[3]] if len(ls_) > 3 else [])
    rnd = rnd + "_lockstep_S%s_W%s_G%s" % tuple(ls_[:3])
prof = tmp / "prof"
env = dict(os.environ, TMPDIR="/tmp")
# the tool itself directly after `--` (no shell, no env wrapper: the profiler's preload initialises the GPU first)
dst = os.path.join(ROOT, "gpurun_out", "profiles")
os.makedirs(dst, exist_ok=True)
if os.environ.get("MODE", "kernel") == "hip":
    # the host side of the chain: which HIP runtime calls a frame makes and what they cost the caller's thread
    # (no counters in this run: --hip-trace is a trace domain)
    subprocess.check_call(["rocprofv3", "--hip-trace", "--stats", "--output-format", "csv", "-d", str(prof), "--"] + cmd, env=env, cwd="/tmp")
    st = glob.glob(str(prof / "**" / "*hip_api_stats.csv"), recursive=True)[0]
    shutil.copy(st, os.path.join(dst, "%s_chain_hip_api_stats.csv" % rnd))
    shutil.copy(glob.glob(str(prof / "**" / "*hip_api_trace.csv"), recursive=True)[0], os.path.join(dst, "%s_chain_hip_api_trace.csv" % rnd))
    print("frames:", n_frames)
elif os.environ.get("MODE") == "trace":
    # every dispatch and every runtime copy with its start and end on the device: scripts/device_timeline.py reads how much of
    # the time the device runs something, and how much of it side by side
    subprocess.check_call(["rocprofv3", "--kernel-trace", "--memory-copy-trace", "--output-format", "csv", "-d", str(prof), "--"] + cmd, env=env, cwd="/tmp")
    st = glob.glob(str(prof / "**" / "*kernel_trace.csv"), recursive=True)[0]
    shutil.copy(st, os.path.join(dst, "%s_chain_kernel_trace.csv" % rnd))
    mc = glob.glob(str(prof / "**" / "*memory_copy_trace.csv"), recursive=True)
    if mc:
        shutil.copy(mc[0], os.path.join(dst, "%s_chain_memory_copy_trace.csv" % rnd))
    sys.exit(0)
else:
    subprocess.check_call(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", str(prof), "--"] + cmd, env=env, cwd="/tmp")
    st = glob.glob(str(prof / "**" / "*kernel_stats.csv"), recursive=True)[0]
    shutil.copy(st, os.path.join(dst, "%s_chain_kernel_stats.csv" % rnd))
print(open(st).read()[:4000])
