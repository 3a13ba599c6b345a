#!/usr/bin/env python3
"""Absolute trajectory error between two TUM-format trajectories ("timestamp tx ty tz qx qy qz qw"):
associate by timestamp, align with the closed-form similarity / rigid transform (Umeyama), report the RMSE of
the translational residuals.  The counterpart of the reference's trajectory-evaluation scripts
(src/vikit/vikit_py, extra/*/scripts).

  python scripts/ate.py estimate.txt groundtruth.txt [--scale] [--max-dt 0.01]
"""
import argparse
import sys

import numpy as np


def load_tum(path):
    rows = []
    for line in open(path):
        line = line.strip()
        if not line or line[0] == "#":
            continue
        v = line.replace(",", " ").split()
        rows.append([float(x) for x in v[:8]])
    a = np.asarray(rows, dtype=np.float64).reshape(-1, 8)
    return a[np.argsort(a[:, 0])]


def associate(t_est, t_gt, max_dt):
    """Greedy nearest-timestamp matching, each ground-truth stamp used once."""
    pairs, j = [], 0
    for i, t in enumerate(t_est):
        while j + 1 < len(t_gt) and abs(t_gt[j + 1] - t) <= abs(t_gt[j] - t):
            j += 1
        if len(t_gt) and abs(t_gt[j] - t) <= max_dt:
            pairs.append((i, j))
    used, out = set(), []
    for i, j in pairs:
        if j not in used:
            used.add(j)
            out.append((i, j))
    return out


def umeyama(src, dst, with_scale):
    """R, t, s minimising sum |dst - (s R src + t)|^2 (Umeyama 1991)."""
    mu_s, mu_d = src.mean(0), dst.mean(0)
    xs, xd = src - mu_s, dst - mu_d
    cov = xd.T @ xs / len(src)
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    s = float(np.trace(np.diag(D) @ S) / (xs ** 2).sum() * len(src)) if with_scale else 1.0
    t = mu_d - s * R @ mu_s
    return R, t, s


def ate(est, gt, with_scale=False, max_dt=0.01):
    pairs = associate(est[:, 0], gt[:, 0], max_dt)
    if len(pairs) < 3:
        raise ValueError("fewer than 3 associated poses")
    pe = est[[i for i, _ in pairs], 1:4]
    pg = gt[[j for _, j in pairs], 1:4]
    R, t, s = umeyama(pe, pg, with_scale)
    res = pg - (s * (R @ pe.T).T + t)
    err = np.linalg.norm(res, axis=1)
    return dict(rmse=float(np.sqrt((err ** 2).mean())), mean=float(err.mean()), median=float(np.median(err)),
                max=float(err.max()), n=len(pairs), scale=s, R=R, t=t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("estimate"); ap.add_argument("groundtruth")
    ap.add_argument("--scale", action="store_true", help="also estimate a scale factor (monocular)")
    ap.add_argument("--max-dt", type=float, default=0.01)
    a = ap.parse_args()
    r = ate(load_tum(a.estimate), load_tum(a.groundtruth), a.scale, a.max_dt)
    print("ATE rmse %.6f m  mean %.6f  median %.6f  max %.6f  (%d poses, scale %.6f)" %
          (r["rmse"], r["mean"], r["median"], r["max"], r["n"], r["scale"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
