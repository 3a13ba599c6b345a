#!/usr/bin/env python3
"""pairs/s of S stereo streams in lock step (svoh_mini_stereo ... <n_streams> <n_workers>) over two rendered sequences, against one stream alone."""
import os, re, subprocess, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_mini_stereo_gpu as t
tmp = pathlib.Path(tempfile.mkdtemp())
cmd_a, out_dir, _, _ = t.make_stereo_dataset(tmp, 30, seed=171, ds="dsA")
cmd_b, _, _, _ = t.make_stereo_dataset(tmp, 30, seed=377, ds="dsB")
r = subprocess.run(cmd_a + ["30", "8", "0.5"], capture_output=True, text=True)
print(r.stdout.strip())
roots = "%s:%s" % (tmp / "dsA", tmp / "dsB")
for S, W, G in ((1, 1, 1), (8, 4, 1), (16, 8, 1), (32, 16, 1), (32, 4, 4), (32, 2, 8), (64, 4, 4), (64, 2, 8)):
    r = subprocess.run(cmd_a + ["30", "8", "0.5", str(S), str(W), str(G)], capture_output=True, text=True, env=dict(os.environ, SVOH_MINI_STEREO_ROOTS=roots, SVOH_LOCKSTEP_TIMING="1"))
    print(r.stdout.strip().splitlines()[-1]); print("   ", [l for l in r.stderr.splitlines() if "lockstep stereo" in l][-2:])
