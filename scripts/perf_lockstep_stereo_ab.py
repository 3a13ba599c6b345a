#!/usr/bin/env python3
"""A/B of one switch of the stereo lock-step engine on ONE box: svoh_mini_stereo ... 32 streams, 4 groups x 4 threads, the two settings in turns.
usage: perf_lockstep_stereo_ab.py ENV_NAME [repeats]   (the tool reads ENV_NAME=0 / 1)"""
import os, re, statistics, subprocess, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_mini_stereo_gpu as t
name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tmp = pathlib.Path(tempfile.mkdtemp())
cmd_a, out_dir, _, _ = t.make_stereo_dataset(tmp, 30, seed=171, ds="dsA")
cmd_b, _, _, _ = t.make_stereo_dataset(tmp, 30, seed=377, ds="dsB")
roots = "%s:%s" % (tmp / "dsA", tmp / "dsB")
rates = {"0": [], "1": []}
for S, W, G in ((32, 4, 4), (8, 4, 1)):
    for r in range(reps):
        for v in ("0", "1"):
            p = subprocess.run(cmd_a + ["30", "8", "0.5", str(S), str(W), str(G)], capture_output=True, text=True, env=dict(os.environ, SVOH_MINI_STEREO_ROOTS=roots, **{name: v}))
            m = re.search(r"(\d+) pairs/s in steady state", p.stdout)
            rates[v].append(int(m.group(1)))
    print("%s, %d streams in %d group(s) x %d threads:  =0: median %d pairs/s %s   =1: median %d pairs/s %s" % (name, S, G, W, statistics.median(rates["0"]), rates["0"], statistics.median(rates["1"]), rates["1"]))
    rates = {"0": [], "1": []}
