"""Feature-level simulator for back-end tests (test infrastructure): a landmark cloud seen by the camera of the synthetic
trajectory (larvio_amd.synthetic.Trajectory), turned into the feature messages a front-end would publish — normalised, undistorted
observations with additive noise, persistent ids, finite-difference velocities — plus the matching IMU stream.  No images, no
front-end: ground truth for the filter that does not pass through any code under test."""
import numpy as np

from larvio_amd import synthetic as S
from larvio_amd._lib import OBS


def R2q(R):
    t = np.trace(R); s = np.sqrt(t + 1) * 2
    return np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])


def qmul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


class LandmarkCloud:
    def __init__(self, tr, t0, t1, seed, n_per_batch=120, fov=(0.75, 0.48)):
        rng = np.random.default_rng([seed, 3]); self.tr, self.fov = tr, fov
        pts = []
        for t in np.arange(t0 - 0.5, t1 + 0.5, 0.5):          # landmarks placed in front of the camera along the whole path
            R_wc, p_wc = tr.cam_pose(t)
            u = rng.uniform(-fov[0], fov[0], n_per_batch); v = rng.uniform(-fov[1], fov[1], n_per_batch); z = rng.uniform(2.0, 9.0, n_per_batch)
            pts.append(np.stack([u * z, v * z, z], 1) @ R_wc.T + p_wc)
        self.pts = np.concatenate(pts)

    def project(self, t):
        R_wc, p_wc = self.tr.cam_pose(t)
        pc = (self.pts - p_wc) @ R_wc
        z = pc[:, 2]; ok = z > 0.5
        uv = np.full((len(pc), 2), np.nan); uv[ok] = pc[ok, :2] / z[ok, None]
        return uv, ok & (np.abs(uv[:, 0]) < self.fov[0]) & (np.abs(uv[:, 1]) < self.fov[1])


INIT_COV = dict(initial_covariance_orientation=1e-6, initial_covariance_velocity=1e-4, initial_covariance_position=1e-6,
                initial_covariance_gyro_bias=1e-8, initial_covariance_acc_bias=1e-6)


def simulate(seed, t0=2.0, t1=8.0, sigma=3e-4, imu_noise=1.0, max_feat=150, perturb=True, n_per_batch=120, **cfg_over):
    """-> dict(cfg, imu, init=(t, q, p, v, bg, ba, gyro_old, acc_old), msgs=[(ts, OBS array)], traj)
    sigma: observation noise in normalised image units (3e-4 ~ 0.14 px at f = 458: what sub-pixel LK delivers);
    imu_noise: scale on the simulator's IMU noise densities (1.0 = synthetic.IMU_NOISE_*).  The initial state is the truth plus a
    draw from the (small) initial covariance the configuration states."""
    tr = S.Trajectory(); seq = S.imu_only_sequence(seed=seed, noise_scale=imu_noise)
    rng = np.random.default_rng([seed, 5])
    cloud = LandmarkCloud(tr, t0, t1, seed, n_per_batch=n_per_batch)
    over = dict(sw_size=20, estimate_td=0, estimate_extrin=0, if_zupt_valid=0, **INIT_COV)
    over.update(cfg_over)
    cfg = S.backend_config(**over)
    imu = seq.imu_array(int(round(t0 * 200)) - 2, int(t1 * 200) + 40)
    ki = int(np.searchsorted(imu["t"], t0, side="right")) - 1; ti = imu["t"][ki]
    q = R2q(tr.R_wb(ti)); p = tr.p_wb(ti).copy(); v = tr.vel(ti).copy()
    if perturb:
        dth = rng.normal(0, np.sqrt(cfg["initial_covariance_orientation"]), 3)
        q = qmul(np.concatenate([0.5 * dth, [1.0]]), q); q /= np.linalg.norm(q)
        p += rng.normal(0, np.sqrt(cfg["initial_covariance_position"]), 3); v += rng.normal(0, np.sqrt(cfg["initial_covariance_velocity"]), 3)
    init = (ti, q, p, v, np.zeros(3), np.zeros(3), imu["gyro"][ki].copy(), imu["acc"][ki].copy())
    tracked, prev_uv, msgs = {}, None, []
    for i in range(int(round((t1 - t0) * 20)) + 1):            # camera frames at 20 Hz, a message every other frame
        ts = t0 + i * 0.05
        uv, vis = cloud.project(ts)
        uvn = uv + rng.normal(0, sigma, uv.shape)
        for j in list(tracked):
            if not vis[j]:
                del tracked[j]
        if len(tracked) < max_feat:
            cand = np.flatnonzero(vis); rng.shuffle(cand)
            for j in cand:
                if len(tracked) >= max_feat:
                    break
                tracked.setdefault(int(j), True)
        if i % 2 == 0 and prev_uv is not None:
            ids = sorted(tracked)
            m = np.zeros(len(ids), OBS)
            for r, j in enumerate(ids):
                pv = prev_uv[j] if np.isfinite(prev_uv[j]).all() else uvn[j]
                m[r] = (j, uvn[j, 0], uvn[j, 1], -1.0, -1.0, (uvn[j, 0] - pv[0]) / 0.05, (uvn[j, 1] - pv[1]) / 0.05, 0.0, 0.0)
            msgs.append((ts, m))
        prev_uv = uvn
    return dict(cfg=cfg, imu=imu, init=init, msgs=msgs, traj=tr, landmarks=cloud.pts)      # feature id = row of `landmarks`


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
