"""Generates tests/golden/ref_static.npz from the REFERENCE's own static initialiser - /root/reference/src/StaticInitializer.cpp
(tryIncInit :12-58, initializeGravityAndBias :61-109, assignInitialState :112-145) compiled in place into oracle/_ref/liblvref_static.so
(oracle/Makefile target `ref`; Eigen served by oracle/ref_shim/lvref_eigen.hpp).  The outputs stored here are NOT the oracle's.
Needs /root/reference; run from the repo root:
    python tests/golden/make_ref_static.py
Cases (seeded): message streams at 10 Hz with 30-60 features, a platform at rest for 0.4-1.9 s (feature jitter 0.0002) that then moves
(displacements well above zupt_max_feature_dis), features that come and go (fewer than 20 common ones resets the counter), tilted IMUs
(gravity direction random), a gyro bias, IMU noise; static_duration 1.0 s at 10 Hz (static_Num 10) and 0.5 s (5)."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scipy.spatial.transform import Rotation  # noqa: E402

THRESH = 7e-4                                             # zupt_max_feature_dis (config/euroc.yaml)


def cases(seed, n_cases):
    rng = np.random.default_rng(seed)
    for k in range(n_cases):
        static_num = 10 if k % 2 == 0 else 5
        rest = rng.uniform(0.4, 1.9)
        n_msgs = 24
        ts = 10.0 + 0.1 * np.arange(n_msgs) + rng.uniform(0, 0.004)
        nf = int(rng.integers(30, 61))
        base = rng.uniform(-0.4, 0.4, (nf, 2))
        Rg = Rotation.from_rotvec(rng.normal(0, 0.5, 3)).as_matrix()
        bias = rng.normal(0, 0.02, 3)
        t_imu = 9.9 + 0.005 * np.arange(int((ts[-1] - 9.9) / 0.005) + 30)
        imu7 = np.column_stack([t_imu, bias + rng.normal(0, 0.004, (len(t_imu), 3)), (Rg @ np.array([0, 0, 9.81])) + rng.normal(0, 0.08, (len(t_imu), 3))])
        msgs = []
        ids0 = np.arange(1000 * k, 1000 * k + nf)
        for i, t in enumerate(ts):
            moved = max(0.0, t - (ts[0] + rest))
            uv = base + rng.normal(0, 2e-4, base.shape) + moved * np.array([0.08, -0.05])
            ids = ids0.copy()
            if k % 3 == 2 and i == 4:                    # a frame that shares fewer than 20 features with its predecessor: the counter starts over
                ids = ids + 500
            msgs.append((float(t), ids, uv))
        yield dict(static_num=static_num, msgs=msgs, imu7=imu7)


def run_reference(c):
    from oracle import lvref
    si = lvref.RefStaticInitializer(THRESH, c["static_num"])
    lo = 0
    for i, (t, ids, uv) in enumerate(c["msgs"]):
        hi = int(np.searchsorted(c["imu7"][:, 0], t + 0.05))
        r = si.try_init(t, ids, uv, c["imu7"][lo:hi])
        if r is not None:
            return i, r
    return -1, None


def main():
    N = 8
    rec = dict(static_num=[], ts=[], ids=[], uv=[], nf=[], imu7=[], n_imu=[], msg=[], out=[])
    for c in cases(20260925, N):
        i, r = run_reference(c)
        nf = len(c["msgs"][0][1])
        ids = np.zeros((24, 60), np.int64); uv = np.zeros((24, 60, 2))
        for j, (t, a, b) in enumerate(c["msgs"]):
            ids[j, :nf] = a; uv[j, :nf] = b
        imu = np.zeros((560, 7)); imu[:len(c["imu7"])] = c["imu7"]
        rec["static_num"].append(c["static_num"]); rec["ts"].append([m[0] for m in c["msgs"]]); rec["ids"].append(ids); rec["uv"].append(uv); rec["nf"].append(nf)
        rec["imu7"].append(imu); rec["n_imu"].append(len(c["imu7"])); rec["msg"].append(i)
        rec["out"].append(np.concatenate([[r["t"]], r["q"], r["bg"], [r["erased"]]]) if r is not None else np.zeros(9))
        print("stream %d: static_Num %d -> initialised at message %d" % (len(rec["msg"]) - 1, c["static_num"], i))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_static.npz"), **{k: np.array(v) for k, v in rec.items()})


if __name__ == "__main__":
    main()
