"""Feature-level simulator for back-end tests (test infrastructure): a landmark cloud seen by the camera of the synthetic
trajectory (larvio_amd.synthetic.Trajectory), turned into the feature messages a front-end would publish — normalised, undistorted
observations with additive noise, persistent ids, finite-difference velocities — plus the matching IMU stream.  No images, no
front-end: ground truth for the filter that does not pass through any code under test."""
import numpy as np

from larvio_amd import synthetic as S
from larvio_amd._lib import OBS


from larvio_amd.synthetic import R2q, qmul, LandmarkCloud, INIT_COV, simulate_features as simulate   # noqa: F401  (the generator lives with the other synthetic inputs)


def drive(ekf, sim, on_update=None, set_state=True):
    """feed a simulated run to an object with the oracle's set_state and process; returns the number of updates.
    set_state=False leaves the start to the filter's own (static) initializer."""
    imu = sim["imu"]; lo = 0; n = 0
    if set_state:
        ekf.set_state(*sim["init"])
    for ts, m in sim["msgs"]:
        hi = int(np.searchsorted(imu["t"], ts + 0.05, side="left"))
        upd, used = ekf.process(ts, m, imu[lo:hi]); lo += used
        if upd:
            n += 1
            if on_update:
                on_update(ts)
    return n


def errors(state, tr):
    """(position, orientation [rad, small-angle], velocity) error of a state dict against the trajectory"""
    t = state["t"]
    dq = qmul(state["q"], R2q(tr.R_wb(t)) * np.array([-1, -1, -1, 1]))
    return state["p"] - tr.p_wb(t), 2 * dq[:3] * np.sign(dq[3]), state["v"] - tr.vel(t)
