#!/usr/bin/env python3
"""A/B of one switch of the mono lock-step engine on ONE box: svoh_mini_frontend ... lockstep, 32 streams in 4 groups x 4 threads, 12 laps of the
40-frame sequence, the two settings in turns.   usage: perf_lockstep_ab.py ENV_NAME [repeats] [mix]   (the tool reads ENV_NAME=0 / 1)"""
import os, re, statistics, subprocess, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_mini_frontend_gpu as t
name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
tmp = pathlib.Path(tempfile.mkdtemp())
cmd, out_dir, poses, stamps, n_frames = t.make_dataset(tmp)
for S, W, G in ((32, 4, 4), (8, 4, 1)):
    rates = {"0": [], "1": []}
    for r in range(reps):
        for v in ("0", "1"):
            p = subprocess.run(cmd + [str(n_frames), "8", str(S), "lockstep", str(W), str(G), "12"], capture_output=True, text=True, env=dict(os.environ, **{name: v}))
            m = re.search(r"(\d+) frames/s in steady state", p.stdout)
            if not m:
                print(p.stdout[-2000:], p.stderr[-2000:]); sys.exit(1)
            rates[v].append(int(m.group(1)))
    print("%s, %d streams in %d group(s) x %d threads:  =0: median %d frames/s %s   =1: median %d frames/s %s" % (name, S, G, W, statistics.median(rates["0"]), rates["0"], statistics.median(rates["1"]), rates["1"]))
