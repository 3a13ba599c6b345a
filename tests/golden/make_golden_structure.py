#!/usr/bin/env python3
"""Generates tests/golden/structure_small.npz from the CPU oracle: 60 landmarks with their observations (a
rank-deficient one, one starting behind a camera and a few with fewer than two observations among them) and the
result of Point::optimize for both error models.  (The reference holds no vectors for this function.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc  # noqa: E402
import pose_helpers as ph  # noqa: E402


def main():
    orc.build()
    sc = ph.make_structure_scene(21, n_points=60, n_views=4)
    out = dict(views=np.array(sc["views"]), obs_begin=sc["obs_begin"], obs_view=sc["obs_view"], obs_f=sc["obs_f"], pos0=sc["pos0"])
    for sphere in (0, 1):
        p, it = orc.optimize_points(sc["views"], sc["obs_begin"], sc["obs_view"], sc["obs_f"], sc["pos0"], n_iter=5,
                                    using_bearing_vector=bool(sphere))
        out["pos_out_%d" % sphere], out["iters_%d" % sphere] = p, it
    path = os.path.join(ROOT, "tests", "golden", "structure_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
