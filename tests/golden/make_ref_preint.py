"""Generates tests/golden/ref_preint.npz from the REFERENCE's own IMU pre-integration - /root/reference/include/Initializer/
ImuPreintegration.h (IntegrationBase: midPointIntegration :62-142, propagate :144-167, repropagate :48-61) compiled in place into
oracle/_ref/liblvref_preint.so (oracle/Makefile target `ref`; Eigen served by oracle/ref_shim/lvref_eigen.hpp).  The outputs stored here
are NOT the oracle's: they are what the reference's text computes.  Needs /root/reference; run from the repo root:
    python tests/golden/make_ref_preint.py
Cases (seeded): 20 sample streams of 5-60 samples at 200 Hz with jittered intervals, hand-held rates (up to ~1.5 rad/s) and specific
forces around gravity, linearised about a zero accelerometer bias (as the initialiser does) and a random gyro bias; half of them are
re-propagated about another gyro bias (solveGyroscopeBias' use)."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lvref  # noqa: E402


def cases(seed, n_cases):
    rng = np.random.default_rng(seed)
    for k in range(n_cases):
        n = int(rng.integers(5, 61))
        t = np.cumsum(np.full(n + 1, 0.005) + rng.uniform(-2e-4, 2e-4, n + 1))
        w0 = rng.normal(0, 0.6, 3); gyr = w0 + 0.4 * np.sin(np.outer(t, rng.uniform(2, 9, 3)) + rng.uniform(0, 6, 3)) + rng.normal(0, 0.004, (n + 1, 3))
        acc = np.array([0, 0, 9.81]) + 1.5 * np.sin(np.outer(t, rng.uniform(1, 7, 3)) + rng.uniform(0, 6, 3)) + rng.normal(0, 0.08, (n + 1, 3))
        bg = rng.normal(0, 0.01, 3)
        yield dict(acc0=acc[0], gyr0=gyr[0], ba=np.zeros(3), bg=bg, dt=np.diff(t), acc=acc[1:], gyr=gyr[1:],
                   rebias=(np.zeros(3), bg + rng.normal(0, 0.005, 3)) if k % 2 else None)


def main():
    N, S = 20, 60
    head = np.zeros((N, 12)); ns = np.zeros(N, np.int32); samples = np.zeros((N, S, 7)); reb = np.zeros((N, 7)); out = np.zeros((N, 56))
    for k, c in enumerate(cases(20260925, N)):
        n = len(c["dt"]); ns[k] = n
        head[k] = np.concatenate([c["acc0"], c["gyr0"], c["ba"], c["bg"]])
        samples[k, :n, 0] = c["dt"]; samples[k, :n, 1:4] = c["acc"]; samples[k, :n, 4:7] = c["gyr"]
        if c["rebias"] is not None:
            reb[k] = np.concatenate([[1.0], c["rebias"][0], c["rebias"][1]])
        r = lvref.preintegrate(c["acc0"], c["gyr0"], c["ba"], c["bg"], c["dt"], c["acc"], c["gyr"], c["rebias"])
        out[k] = np.concatenate([r["dp"], r["dq"], r["dv"], [r["sum_dt"]], r["dq_dbg"].ravel(), r["dp_dbg"].ravel(), r["dv_dbg"].ravel(), r["dp_dba"].ravel(), r["dv_dba"].ravel()])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_preint.npz"), head=head, n=ns, samples=samples, rebias=reb, out=out)
    print("ref_preint.npz written: %d streams, %d samples" % (N, int(ns.sum())))


if __name__ == "__main__":
    main()
