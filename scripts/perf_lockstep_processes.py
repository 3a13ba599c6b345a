#!/usr/bin/env python3
"""One process with G lock-step groups against G processes with one group each (svoh_mini_frontend ... lockstep), same total streams and host
threads: does the HIP runtime's submission path, which the groups of one process share, hold the groups back?  Steady-state frames/s summed."""
import os, re, subprocess, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_mini_frontend_gpu as t

tmp = pathlib.Path(tempfile.mkdtemp())
cmd, out_dir, poses, stamps, n_frames = t.make_dataset(tmp)
S, G, W, LAPS = 32, 4, 4, 6
rx = re.compile(r"([0-9]+) frames/s in steady state")


def run_one(n_streams, groups, workers, out, env=None):
    os.makedirs(out, exist_ok=True)
    c = list(cmd); c[4] = out
    return subprocess.Popen(c + [str(n_frames), "8", str(n_streams), "lockstep", str(workers), str(groups), str(LAPS)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                            env=dict(os.environ, **(env or {})))


for rep in range(2):
    p = run_one(S, G, W, str(tmp / "one"))
    o, e = p.communicate()
    one = float(rx.search(o).group(1)) if rx.search(o) else float("nan")
    ps = [run_one(S // G, 1, W, str(tmp / ("p%d" % i))) for i in range(G)]
    rates = []
    for q in ps:
        o, e = q.communicate()
        rates.append(float(rx.search(o).group(1)) if rx.search(o) else float("nan"))
    print("run %d: one process, %d groups x %d threads: %.0f frames/s | %d processes of one group x %d threads: %.0f frames/s in total %s" % (rep, G, W, one, G, W, sum(rates), rates))
