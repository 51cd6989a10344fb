"""Generates tests/golden/*.npz from the CPU oracle (the reference has no fixtures of its own; the fixtures written by
the reference compiled in place are the ref_*.npz files, made by make_ref_*.py).  The fixtures freeze the oracle's arithmetic so that any later
edit that changes a byte is caught by tests/test_oracle_frontend.py::test_golden_fixtures and by the GPU parity tests.
Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lvo, lvo_be  # noqa: E402
from scipy import ndimage  # noqa: E402

rng = np.random.default_rng(20260924)
base = ndimage.zoom(rng.uniform(0, 255, (40, 52)), 4, order=3)[:150, :200]
img0 = np.clip(base + rng.normal(0, 4, base.shape), 0, 255).astype(np.uint8)
img1 = np.clip(ndimage.shift(base, (0.8, -1.7), order=3, mode="reflect") + rng.normal(0, 4, base.shape), 0, 255).astype(np.uint8)
p0 = lvo.LkPyramid(lvo.clahe(img0)); p1 = lvo.LkPyramid(lvo.clahe(img1))
pts = p0.good_features(40, 0.01, 10.0)
out, st, it = lvo.lk_track(p0, p1, pts, pts)
e, b = p0.orb_prepare()
d, a = lvo.orb_describe(e, b, pts)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_frontend_stages import _two_view  # noqa: E402
x1, x2 = _two_view(120, 0.25, 77)
ok, mask, iters = lvo.ransac_fundamental(x1, x2)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frontend_small.npz"), img0=img0, img1=img1, clahe0=lvo.clahe(img0),
                    lvl2=p0.image(2), der1=p0.deriv(1), corners=pts, lk_status=st, lk_pts=out, desc=d, x1=x1, x2=x2,
                    ransac_mask=mask, ransac_iters=iters)
print("frontend_small.npz written:", len(pts), "corners,", int(st.sum()), "tracked,", int(mask.sum()), "ransac inliers")

# ---- back-end: a short simulated run (tests/feature_sim.py), inputs AND the oracle's outputs, so that the regression check does not
# depend on the random generator that made the inputs
from tests import feature_sim as F  # noqa: E402
sim = F.simulate(11, t0=2.0, t1=3.9, max_feat=48, sw_size=8, estimate_td=1, estimate_extrin=1)
ekf = lvo_be.Ekf(sim["cfg"])
trace = []
n_upd = F.drive(ekf, sim, lambda ts: trace.append(np.concatenate([[ekf.state()["t"]], ekf.state()["q"], ekf.state()["p"], ekf.state()["v"], [ekf.dim]])))
s = ekf.state(); ids, idp, pos = ekf.features()
cfg_keys = sorted(k for k in sim["cfg"] if k not in ("intrinsics", "T_cam_imu"))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "backend_sim.npz"),
                    cfg_keys=np.array(cfg_keys), cfg_vals=np.array([float(sim["cfg"][k]) for k in cfg_keys]),
                    intrinsics=np.array(sim["cfg"]["intrinsics"], np.float64), T_cam_imu=np.asarray(sim["cfg"]["T_cam_imu"], np.float64),
                    imu=sim["imu"], init=np.concatenate([[sim["init"][0]]] + [np.asarray(x, np.float64) for x in sim["init"][1:]]),
                    msg_ts=np.array([m[0] for m in sim["msgs"]]), msg_len=np.array([len(m[1]) for m in sim["msgs"]]),
                    msg_obs=np.concatenate([m[1] for m in sim["msgs"]]),
                    trace=np.array(trace), cov=ekf.cov(), clone_ids=ekf.clones()["id"], feat_ids=ids, feat_idp=idp,
                    bg=s["bg"], ba=s["ba"], R_b2c=s["R_b2c"], t_c_b=s["t_c_b"], td=s["td"],
                    counters=np.array([ekf.counters()[k] for k in ("hybrid", "msckf", "zupt", "gated_in", "gated_out", "map")]))
print("backend_sim.npz written:", n_upd, "updates, dim", ekf.dim, ekf.counters())
