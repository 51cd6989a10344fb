"""Generates tests/golden/ref_larvio.npz from the REFERENCE'S OWN FILTER - /root/reference/src/larvio.cpp (LarVio::processFeatures) with
src/FlexibleInitializer.cpp, src/StaticInitializer.cpp and src/feature_manager.cpp compiled in place into oracle/_ref/liblvref_larvio.so
(oracle/Makefile target `ref`; Eigen / OpenCV / boost served by oracle/ref_shim2/).  The outputs stored here are NOT the oracle's.
Needs /root/reference; run from the repo root:
    python tests/golden/make_ref_larvio.py
Streams:  A - the inputs already stored in tests/golden/backend_sim.npz (19 updates, td and extrinsics estimated, 8-clone window, start
from a handed-in state: only the reference's outputs are added here);  B - a start at rest with config/euroc.yaml's initial covariances
(the reference's StaticInitializer fires after one second of motionless features, zero-velocity updates, then flight; 40 tracks,
12-clone window, td and extrinsics estimated), inputs and outputs stored.
Per processFeatures call: its return value and the number of IMU samples it erased; per update: the 30 state numbers (time, q, v, p,
b_g, b_a, R_imu_cam0, t_cam0_imu, td), the state dimension, the number of in-state features, P[:15, :15], trace(P) and |P|_F; at the
end: the whole covariance, the in-state feature ids in state order, the clone times."""
import os
import sys
import tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lvref  # noqa: E402
from tests import feature_sim as F  # noqa: E402
from tests.test_oracle_backend import _load_backend_golden  # noqa: E402
from tests.test_oracle_ref_larvio import state_row, EUROC_COV  # noqa: E402


def run_reference(cfg, msgs, imu, init):
    ref = lvref.RefLarVio(cfg, tempfile.mkdtemp())
    if init is not None:
        ref.set_state(*init)
    lo = 0; rows, p15, pn, used_all, ok_all = [], [], [], [], []
    for ts, m in msgs:
        hi = int(np.searchsorted(imu["t"], ts + 0.05, side="left"))
        ok, used = ref.process(ts, m, imu[lo:hi]); lo += used
        used_all.append(used); ok_all.append(int(ok))
        if not ok:
            continue
        P = ref.cov(); ids = ref.features()[0]
        rows.append(state_row(ref.state(), P.shape[0], len(ids))); p15.append(P[:15, :15].reshape(-1)); pn.append([np.trace(P), np.linalg.norm(P)])
    return dict(state=np.array(rows), p15=np.array(p15), pnorm=np.array(pn), used=np.array(used_all), ok=np.array(ok_all), cov=ref.cov(),
                feat_ids=ref.features()[0], clone_t=ref.clones()["time"])


out = {}
z, cfg, init, msgs = _load_backend_golden()
a = run_reference(cfg, msgs, z["imu"], init)
out.update({"a_" + k: v for k, v in a.items()})

sim = F.simulate(3, t0=0.1, t1=3.9, max_feat=40, sw_size=12, if_zupt_valid=1, estimate_td=1, estimate_extrin=1, **EUROC_COV)
b = run_reference(sim["cfg"], sim["msgs"], sim["imu"], None)
out.update({"b_" + k: v for k, v in b.items()})
cfg_keys = sorted(k for k in sim["cfg"] if k not in ("intrinsics", "T_cam_imu"))
hi = int(np.searchsorted(sim["imu"]["t"], sim["msgs"][-1][0] + 0.06))
out.update(b_cfg_keys=np.array(cfg_keys), b_cfg_vals=np.array([float(sim["cfg"][k]) for k in cfg_keys]),
           b_intrinsics=np.array(sim["cfg"]["intrinsics"], np.float64), b_T_cam_imu=np.asarray(sim["cfg"]["T_cam_imu"], np.float64),
           b_imu=sim["imu"][:hi], b_msg_ts=np.array([m[0] for m in sim["msgs"]]), b_msg_len=np.array([len(m[1]) for m in sim["msgs"]]),
           b_msg_obs=np.concatenate([m[1] for m in sim["msgs"]]))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_larvio.npz"), **out)
print("ref_larvio.npz written: A", len(a["state"]), "updates, dim", a["cov"].shape[0], "| B", len(b["state"]), "updates of", len(sim["msgs"]), "messages, dim",
      b["cov"].shape[0], "first update at", b["state"][0][0])
