"""A/B of the per-frame chain (tools/svoh_mini_frontend on the synthetic sequence of tests/test_mini_frontend_gpu.py) between
the libraries in the tree and those in another directory (default build/head: libsvo_hip.so + libsvo_hip_host.so of an
earlier commit), interleaved on one box.  Prints the median / mean of ms_frame (the caller's clock) per run and whether
the trajectory files and the counters of frontend.csv are identical.   python scripts/ab_chain.py [other_dir] [rounds]"""
import os, pathlib, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_mini_frontend_gpu as t

# "env:NAME=VALUE" instead of a directory: "other" = the tree's libraries with that environment variable set
other_env = None
if len(sys.argv) > 1 and sys.argv[1].startswith("env:"):
    other_env = sys.argv[1][4:].split("=", 1)
other = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 and not other_env else os.path.join(ROOT, "build", "head"))
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tmp = pathlib.Path(tempfile.mkdtemp(prefix="abchain_", dir="/tmp"))
cmd, out_dir, poses, stamps, n_frames = t.make_dataset(tmp)
res = {"tree": [], "other": []}
files = {}
for r in range(rounds):
    for name in ("other", "tree") if r % 2 else ("tree", "other"):
        env = dict(os.environ)
        if name == "other" and other_env:
            env[other_env[0]] = other_env[1]
        elif name == "other":
            env["LD_LIBRARY_PATH"] = other + ":" + env.get("LD_LIBRARY_PATH", "")
        p = subprocess.run(cmd, capture_output=True, text=True, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
        fc = np.loadtxt(str(out_dir / "frontend.csv"), delimiter=",", skiprows=1)
        res[name].append((float(np.median(fc[3:, 13])), float(fc[3:, 13].mean()), np.median(fc[3:, 7:13], axis=0)))
        files[name] = (open(str(out_dir / "trajectory.txt")).read(), fc[:, :7].copy())
for name in ("other", "tree"):
    print("%-5s ms_frame median per run: %s   mean per run: %s" % (name, " ".join("%.3f" % m for m, _, _ in res[name]), " ".join("%.3f" % m for _, m, _ in res[name])))
    st = np.median(np.array([s for _, _, s in res[name]]), axis=0)
    print("      stage medians: pyramid %.3f align %.3f reproject %.3f pose %.3f seeds %.3f keyframe+release %.3f" % tuple(st))
print("trajectory identical:", files["tree"][0] == files["other"][0], " counters identical:", bool(np.array_equal(files["tree"][1], files["other"][1])))
