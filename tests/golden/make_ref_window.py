"""Generates tests/golden/ref_window.npz from the REFERENCE's own window bookkeeping - /root/reference/src/feature_manager.cpp
(addFeatureCheckParallax :45-97, getCorresponding :100-120, removeBack :203-220) compiled in place into oracle/_ref/liblvref_fm.so
(oracle/Makefile target `ref`; Eigen served by oracle/ref_shim/lvref_eigen.hpp).  The outputs stored here are NOT the oracle's.
Needs /root/reference; run from the repo root:
    python tests/golden/make_ref_window.py
Streams (seeded): 24 messages of 25-45 tracks with lifetimes of 2-20 messages (ids appear, live and disappear; some ids exceed 2^31 to
exercise the reference's `int feature_id`), pixel motion with per-track velocities, a camera-IMU time offset; the window fills (11
frames) and then slides once per message, as when every initialisation attempt fails."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
WINDOW = 10


def streams(seed, n_streams):
    rng = np.random.default_rng(seed)
    for s in range(n_streams):
        td = float(rng.normal(0, 0.005))
        live = {}; next_id = 10 + (2 ** 31 + 5 if s % 3 == 2 else 0)
        msgs = []
        for m in range(24):
            for i in list(live):
                live[i][0] -= 1
                if live[i][0] <= 0:
                    del live[i]
            want = int(rng.integers(25, 46))
            while len(live) < want:
                live[next_id] = [int(rng.integers(2, 21)), rng.uniform(-0.5, 0.5, 2), rng.normal(0, 0.3, 2)]; next_id += int(rng.integers(1, 4))
            ids = np.array(sorted(live), np.int64)
            uvv = np.array([np.concatenate([live[i][1] + 0.1 * m * live[i][2] + rng.normal(0, 1e-4, 2), live[i][2]]) for i in ids])
            msgs.append((ids, uvv))
        yield dict(td=td, msgs=msgs)


def run_reference(c):
    """-> (answers of addFeatureCheckParallax per message, {(message, l): correspondences (k, 4)}, feature counts after each slide)"""
    from oracle import lvref
    fm = lvref.RefFeatureManager(); fc = 0; ans = []; cor = {}; cnt = []
    for m, (ids, uvv) in enumerate(c["msgs"]):
        ans.append(fm.add(fc, ids, uvv, c["td"]))
        if fc == WINDOW:
            for l in range(WINDOW):
                cor[(m, l)] = fm.corresponding(l, WINDOW)
            fm.remove_back(); cnt.append(fm.feature_count())
        else:
            fc += 1
    return ans, cor, cnt


def main():
    N = 6
    rec = dict(td=[], ids=[], uvv=[], n=[], ans=[], ncor=[], cor=[], cnt=[])
    for c in streams(20260925, N):
        ans, cor, cnt = run_reference(c)
        ids = np.zeros((24, 45), np.int64); uvv = np.zeros((24, 45, 4)); n = np.zeros(24, np.int32)
        for m, (a, b) in enumerate(c["msgs"]):
            n[m] = len(a); ids[m, :len(a)] = a; uvv[m, :len(a)] = b
        ncor = np.zeros((24, WINDOW), np.int32); cc = np.zeros((24, WINDOW, 45, 4))
        for (m, l), v in cor.items():
            ncor[m, l] = len(v); cc[m, l, :len(v)] = v
        rec["td"].append(c["td"]); rec["ids"].append(ids); rec["uvv"].append(uvv); rec["n"].append(n); rec["ans"].append(np.array(ans, np.int32)); rec["ncor"].append(ncor)
        rec["cor"].append(cc); rec["cnt"].append(np.array(cnt, np.int32))
        print("stream: addFeatureCheckParallax said 'oldest' %d of %d times; %d slides, %d..%d correspondences with the newest frame" % (sum(ans), len(ans), len(cnt), ncor[10:].min(), ncor[10:].max()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_window.npz"), **{k: np.array(v) for k, v in rec.items()})


if __name__ == "__main__":
    main()
