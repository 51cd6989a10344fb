"""Generates tests/golden/ref_dyninit.npz from the REFERENCE'S OWN moving-start initialiser - /root/reference/src/DynamicInitializer.cpp,
initial_sfm.cpp, solve_5pts.cpp, initial_alignment.cpp, feature_manager.cpp compiled in place into oracle/_ref/liblvref_dyninit.so (oracle/Makefile target
`ref`; the minimisers behind cv::solvePnP and Ceres are stand-ins, see oracle/ref_shim4/).  The outputs stored here are NOT the oracle's.
Needs /root/reference; run from the repo root:
    python tests/golden/make_ref_dyninit.py
One recorded start (tests/feature_sim.py, 60 tracks, 0.14 px observation noise, IMU noise, 4 m/s trajectory): the messages and IMU samples,
and what DynamicInitializer::assignInitialState handed over."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lvref  # noqa: E402
from larvio_amd import synthetic as S  # noqa: E402
from tests import feature_sim as F  # noqa: E402

sim = F.simulate(12, t0=3.5, t1=4.8, sigma=3e-4, imu_noise=1.0, max_feat=60, traj=S.Trajectory(speed=4.0), fresh_ids=True)
T = np.asarray(S.EUROC["T_cam_imu"], float); R_b2c = T[:3, :3]; t_c_b = -R_b2c.T @ T[:3, 3]
r = lvref.dynamic_init(sim["msgs"], sim["imu"], R_b2c, t_c_b)
assert r is not None
hi = int(np.searchsorted(sim["imu"]["t"], sim["msgs"][-1][0] + 0.06))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_dyninit.npz"), R_b2c=R_b2c, t_c_b=t_c_b, imu=sim["imu"][:hi],
                    msg_ts=np.array([m[0] for m in sim["msgs"]]), msg_len=np.array([len(m[1]) for m in sim["msgs"]]), msg_obs=np.concatenate([m[1] for m in sim["msgs"]]),
                    message=r["message"], state_time=r["state_time"], erase=r["erase"], q=r["q"], v=r["v"], bg=r["bg"], g=r["g"])
print("ref_dyninit.npz written: success at message", r["message"], "state time", r["state_time"], "erase", r["erase"], "v", r["v"])
