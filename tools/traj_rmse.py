#!/usr/bin/env python3
"""Absolute trajectory error of a VIO run against ground truth — the tool behind BASELINE.json's "trajectory RMSE within 1 mm of
the reference" clause.

  tools/traj_rmse.py estimate ground_truth [--ref other_estimate] [--align se3|yaw|none] [--takeoff msckf_2_takeoff.txt]

estimate / --ref:   a TUM trajectory ("t x y z qx qy qz qw", what examples/larvio_euroc --tum writes), or the reference's own
                    log msckf_2_state.txt ("t-takeoff qw qx qy qz vx vy vz px py pz …", larvio.cpp:446-453; give --takeoff or the
                    file msckf_2_takeoff.txt is looked up next to it)
ground_truth:       EuRoC state_groundtruth_estimate0/data.csv (ns, p, q_wxyz, …) or a TUM file

The estimate is associated to the ground truth by linear interpolation at its own stamps, aligned with the closed-form
least-squares rigid transform (Horn/Umeyama without scale; "yaw" restricts the rotation to the gravity axis, the 4 unobservable
degrees of freedom of VIO), and the RMSE of the remaining position differences is printed.  With --ref the same is done for the
second estimate and the difference of the two RMSEs is reported in millimetres.
"""
import argparse
import os
import sys

import numpy as np


def load_trajectory(path, takeoff=None):
    """-> (t[n] seconds, p[n,3])"""
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line[0] in "#%":
                continue
            rows.append([float(x) for x in line.replace(",", " ").split()])
    if not rows:
        raise ValueError(f"{path}: no samples")
    width = min(len(r) for r in rows)
    a = np.array([r[:width] for r in rows], np.float64)
    if width >= 24:                                             # msckf_2_state.txt
        if takeoff is None:
            cand = os.path.join(os.path.dirname(os.path.abspath(path)), "msckf_2_takeoff.txt")
            takeoff = cand if os.path.exists(cand) else None
        t0 = float(open(takeoff).read().split()[0]) if isinstance(takeoff, str) else float(takeoff or 0.0)
        return a[:, 0] + t0, a[:, 8:11]
    t = a[:, 0]
    if np.median(t) > 1e12:                                     # nanosecond stamps (EuRoC csv)
        t = t * 1e-9
    return t, a[:, 1:4]


def associate(t_est, p_est, t_gt, p_gt):
    keep = (t_est >= t_gt[0]) & (t_est <= t_gt[-1])
    te, pe = t_est[keep], p_est[keep]
    pg = np.stack([np.interp(te, t_gt, p_gt[:, k]) for k in range(3)], axis=1)
    return te, pe, pg


def align(pe, pg, mode="se3"):
    """R, t minimising sum |R pe + t - pg|^2"""
    if mode == "none":
        return np.eye(3), np.zeros(3)
    ce, cg = pe.mean(0), pg.mean(0)
    W = (pg - cg).T @ (pe - ce)
    if mode == "yaw":
        th = np.arctan2(W[1, 0] - W[0, 1], W[0, 0] + W[1, 1])
        R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    else:
        U, _, Vt = np.linalg.svd(W)
        D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
        R = U @ D @ Vt
    return R, cg - R @ ce


def ate_rmse(t_est, p_est, t_gt, p_gt, mode="se3"):
    te, pe, pg = associate(t_est, p_est, t_gt, p_gt)
    if len(te) < 3:
        raise ValueError("fewer than 3 associated samples")
    R, t = align(pe, pg, mode)
    d = (pe @ R.T + t) - pg
    return float(np.sqrt((d * d).sum(1).mean())), len(te)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("estimate"); ap.add_argument("ground_truth")
    ap.add_argument("--ref"); ap.add_argument("--align", default="se3", choices=["se3", "yaw", "none"])
    ap.add_argument("--takeoff")
    a = ap.parse_args(argv)
    t_gt, p_gt = load_trajectory(a.ground_truth)
    t_e, p_e = load_trajectory(a.estimate, a.takeoff)
    rmse, n = ate_rmse(t_e, p_e, t_gt, p_gt, a.align)
    print(f"estimate  : ATE RMSE {rmse:.6f} m over {n} poses ({a.align} alignment)")
    if a.ref:
        t_r, p_r = load_trajectory(a.ref, a.takeoff)
        rmse_r, n_r = ate_rmse(t_r, p_r, t_gt, p_gt, a.align)
        print(f"reference : ATE RMSE {rmse_r:.6f} m over {n_r} poses")
        print(f"difference: {1e3 * (rmse - rmse_r):+.3f} mm")
    return 0


if __name__ == "__main__":
    sys.exit(main())
