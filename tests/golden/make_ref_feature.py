"""Generates tests/golden/ref_feature.npz from the REFERENCE's own Feature code - /root/reference/include/larvio/feature.hpp (checkMotion
:334-381; initializePosition :383-552, initializePosition_AssignAnchor :554-721, initializeInvParamPosition :723-890 with cost / jacobian /
generateInitialGuess :252-332) compiled in place into oracle/_ref/liblvref_feature.so (oracle/Makefile target `ref`; Eigen served by the
stand-in oracle/ref_shim/lvref_eigen.hpp).  The outputs stored here are NOT the oracle's: they are what the reference's text computes.
Needs /root/reference; run from the repo root:
    python tests/golden/make_ref_feature.py
Cases (seeded): 3-9 views along a hand-held path, landmarks at 1.5-6 m and (one in seven) at 20-60 m where the parallax is small,
observation noise 0.002 (one in five: 0.02, so that some solutions fail the reprojection test), all three variants, fresh features and
(one in four) features that already hold a position (which the third variant has to ignore); checkMotion with and without the current frame, two thresholds."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lvref  # noqa: E402
from scipy.spatial.transform import Rotation  # noqa: E402


def cases(seed, n_cases):
    rng = np.random.default_rng(seed)
    for trial in range(n_cases):
        n = int(rng.integers(3, 10))
        ids = np.arange(100, 100 + n).astype(np.int64)
        q = np.array([Rotation.from_rotvec(rng.normal(0, 0.05, 3) + np.array([0, 0.02 * i, 0])).as_quat() for i in range(n)])      # camera-to-world, [x y z w]
        pc = np.array([np.array([0.08 * i, 0.01 * i, 0.0]) * rng.uniform(0.2, 1.5) + rng.normal(0, 0.005, 3) for i in range(n)])
        X = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(1.5, 6)])
        if trial % 7 == 0:
            X[2] = rng.uniform(20, 60)
        if trial % 23 == 0:
            X[2] = -X[2]                                   # behind the cameras: the validity test has to say no
        sig = 0.002 if trial % 5 else 0.02
        uv = np.array([(lambda pcam: pcam[:2] / pcam[2])(Rotation.from_quat(q[i]).as_matrix().T @ (X - pc[i])) + rng.normal(0, sig, 2) for i in range(n)])
        mode = trial % 3
        curr = int(ids[-1]) if mode != 1 else -1
        is_init = trial % 4 == 0                           # (initializeInvParamPosition ignores it: it always starts from the two-view guess, :766)
        pos_in = X + rng.normal(0, 0.05, 3) if is_init else np.zeros(3)
        yield dict(ids=ids, q=q, pc=pc, uv=uv, mode=mode, curr=curr, is_init=is_init, pos_in=pos_in, thr=0.2 if trial % 2 else 0.05)


def main():
    N = 240; V = 9
    ids = np.zeros((N, V), np.int64); q = np.zeros((N, V, 4)); pc = np.zeros((N, V, 3)); uv = np.zeros((N, V, 2)); nv = np.zeros(N, np.int32)
    mode = np.zeros(N, np.int32); curr = np.zeros(N, np.int64); is_init = np.zeros(N, np.int32); pos_in = np.zeros((N, 3)); thr = np.zeros(N)
    ok = np.zeros(N, np.int32); out = np.zeros((N, 15)); motion = np.zeros((N, 2), np.int32)
    for k, c in enumerate(cases(20260925, N)):
        n = len(c["ids"]); nv[k] = n
        ids[k, :n] = c["ids"]; q[k, :n] = c["q"]; pc[k, :n] = c["pc"]; uv[k, :n] = c["uv"]
        mode[k] = c["mode"]; curr[k] = c["curr"]; is_init[k] = c["is_init"]; pos_in[k] = c["pos_in"]; thr[k] = c["thr"]
        o, m = lvref.feature_initialize(c["mode"], c["ids"], c["q"], c["pc"], c["ids"], c["uv"], c["curr"], c["is_init"], c["pos_in"])
        ok[k] = o
        out[k] = np.concatenate([m["position"], m["position_fej"], [m["inv_depth"]], m["obs_anchor"], [m["id_anchor"]], m["inv_param"], [m["is_initialized"]]])
        for t in (0, 1):
            motion[k, t] = lvref.feature_check_motion(c["ids"], c["q"], c["pc"], c["ids"], c["uv"], t, c["thr"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_feature.npz"), n_views=nv, ids=ids, q_cam=q, p_cam=pc, uv=uv, mode=mode, curr_id=curr,
                        is_initialized=is_init, position_in=pos_in, threshold=thr, ok=ok, out=out, motion=motion)
    print("ref_feature.npz written: %d cases, %d valid solutions, checkMotion true in %d of %d" % (N, int(ok.sum()), int(motion.sum()), motion.size))


if __name__ == "__main__":
    main()
