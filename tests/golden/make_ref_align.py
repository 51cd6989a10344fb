"""Generates tests/golden/ref_align.npz from the REFERENCE's own visual-inertial alignment - /root/reference/src/initial_alignment.cpp
(solveGyroscopeBias :10-46, LinearAlignment :131-201, RefineGravity :65-128 with TangentBasis :49-62) and the pre-integration it reads
(include/Initializer/ImuPreintegration.h), compiled in place into oracle/_ref/liblvref_align.so (oracle/Makefile target `ref`; Eigen and
boost::shared_ptr served by oracle/ref_shim/).  The outputs stored here are NOT the oracle's: they are what the reference's text computes.
Needs /root/reference; run from the repo root:
    python tests/golden/make_ref_align.py
Cases (seeded): windows of 11 frames 0.1 s apart cut from the synthetic trajectory at three speeds (tests/feature_sim.py's IMU stream with
its noise and a gyro bias), frame rotations = true attitudes with 0.2 mrad of noise in an arbitrary gravity-tilted reference frame, frame
positions = true camera positions in that frame divided by an arbitrary structure-from-motion scale, with 0.2 mm of noise."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scipy.spatial.transform import Rotation  # noqa: E402


def cases(seed, n_cases):
    from larvio_amd import synthetic as S
    from tests import feature_sim as F
    rng = np.random.default_rng(seed)
    T_ci = np.asarray(S.EUROC["T_cam_imu"], float); R_b2c = T_ci[:3, :3]; t_c_b = -R_b2c.T @ T_ci[:3, 3]        # camera position in the body frame
    for k in range(n_cases):
        speed = (2.0, 3.0, 4.0)[k % 3]
        tr = S.Trajectory(speed=speed)
        t0 = 3.0 + 0.37 * k
        sim = F.simulate(100 + k, t0=t0, t1=t0 + 1.4, sigma=0.0, imu_noise=1.0, traj=tr, fresh_ids=True)
        imu = sim["imu"]
        ts = np.array([m[0] for m in sim["msgs"]][:11])
        Rc0w = Rotation.from_rotvec(rng.normal(0, 0.6, 3)).as_matrix()
        scale = rng.uniform(0.3, 3.0)
        R = np.array([Rc0w @ tr.R_wb(t) @ Rotation.from_rotvec(rng.normal(0, 2e-4, 3)).as_matrix() for t in ts])
        T = np.array([Rc0w @ (tr.p_wb(t) + tr.R_wb(t) @ t_c_b) for t in ts]) / scale + rng.normal(0, 2e-4, (11, 3))
        bias = rng.normal(0, 0.01, 3)
        heads = [None]; streams = [None]
        for j in range(1, 11):
            sel = np.flatnonzero((imu["t"] > ts[j - 1]) & (imu["t"] <= ts[j]))
            first = sel[0] - 1
            heads.append((imu["acc"][first].copy(), imu["gyro"][first] + bias))
            tt = imu["t"][np.concatenate([[first], sel])]
            streams.append(np.column_stack([np.diff(tt), imu["acc"][sel], imu["gyro"][sel] + bias]))
        yield dict(t=ts, R=R, T=T, heads=heads, streams=streams, bg0=np.zeros(3), tic=t_c_b, scale=scale, g_c0=Rc0w @ np.array([0, 0, 9.81]), bias=bias)


def pack(c):
    n_s = np.array([0] + [len(s) for s in c["streams"][1:]], np.int32)
    hd = np.zeros((11, 6)); hd[1:] = [np.concatenate(h) for h in c["heads"][1:]]
    sm = np.concatenate(c["streams"][1:])
    return n_s, hd, sm


def main():
    from oracle import lvref
    N = 9
    rec = dict(t=[], R=[], T=[], n_s=[], head=[], samples=[], tic=[], ok=[], bg=[], g=[], x=[])
    for c in cases(20260925, N):
        r = lvref.visual_imu_alignment(c["t"], c["R"], c["T"], c["heads"], c["streams"], c["bg0"], c["tic"])
        n_s, hd, sm = pack(c)
        pad = np.zeros((11 * 24, 7)); pad[:len(sm)] = sm
        rec["t"].append(c["t"]); rec["R"].append(c["R"]); rec["T"].append(c["T"]); rec["n_s"].append(n_s); rec["head"].append(hd); rec["samples"].append(pad)
        rec["tic"].append(c["tic"]); rec["ok"].append(int(r["ok"])); rec["bg"].append(r["bg"]); rec["g"].append(r["g"]); rec["x"].append(r["x"] if len(r["x"]) == 36 else np.zeros(36))
        print("window at t = %.2f: ok %d, |g| %.4f, scale %.4f (true %.4f), bg - bias %.1e" % (c["t"][0], r["ok"], np.linalg.norm(r["g"]), r["x"][-1] if len(r["x"]) else -1, c["scale"],
                                                                                           np.abs(r["bg"] - c["bias"]).max()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_align.npz"), **{k: np.array(v) for k, v in rec.items()})
    print("ref_align.npz written: %d windows, %d aligned" % (N, int(np.sum(rec["ok"]))))


if __name__ == "__main__":
    main()
