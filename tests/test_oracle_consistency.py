"""The oracle's filter against ground truth that does not pass through any code under test: simulated feature messages
(tests/feature_sim.py) from a landmark cloud along the synthetic trajectory.  Pins the whole back-end restatement — propagation,
augmentation, MSCKF and EKF-SLAM updates, re-anchoring and pruning over 60 updates — by (a) staying on the truth when nothing is
noisy, (b) keeping its own covariance honest when things are (Monte-Carlo NEES), (c) showing that the NEES statistic does react to a
filter whose assumptions are broken (so (b) is not vacuous)."""
import numpy as np

from oracle import lvo_be
from tests import feature_sim as F


def _run(seed, **kw):
    sim = F.simulate(seed, **kw)
    ekf = lvo_be.Ekf(sim["cfg"])
    rec = []

    def on_update(ts):
        s, P = ekf.state(), ekf.cov()
        ep, eth, ev = F.errors(s, sim["traj"])
        rec.append((np.linalg.norm(ep), np.linalg.norm(eth), np.linalg.norm(ev), ep @ np.linalg.solve(P[6:9, 6:9], ep),
                    eth @ np.linalg.solve(P[0:3, 0:3], eth), ev @ np.linalg.solve(P[3:6, 3:6], ev), np.sqrt(np.trace(P[6:9, 6:9]))))
        assert np.array_equal(P, P.T) and np.linalg.eigvalsh(P).min() > -1e-9
    n = F.drive(ekf, sim, on_update)
    return np.array(rec), ekf.counters(), n


def test_noise_free_run_stays_on_the_truth():
    """exact IMU, exact observations, start on the truth: 6 s and 60 updates later the estimate is still there (what remains is the
    RK4 / finite-difference error of the simulator and the filter's linearisation) and the chi-square gate rejected nothing"""
    rec, c, n = _run(1, sigma=0.0, imu_noise=0.0, perturb=False)
    assert n == 60 and c["gated_out"] == 0 and c["gated_in"] > 3000 and c["hybrid"] >= 50 and c["msckf"] >= 10
    assert rec[:, 0].max() < 0.01 and rec[:, 1].max() < 1e-3 and rec[:, 2].max() < 0.01, rec.max(0)


def test_in_state_landmarks_stay_on_the_true_points_through_re_anchoring():
    """noise-free again, looking at the EKF-SLAM half: after every update each in-state feature's world position (anchor pose x
    inverse depth, re-anchored whenever its anchor clone is pruned — updateFeatureCov_1didp, larvio.cpp:3125-3293; with a 10-clone
    window that is every few updates) must be the landmark the simulator projected.  A freshly initialised distant point carries the
    depth error of its short baseline (measured: up to 1.5 % at 11 m, shrinking to millimetres), so the checks are: the bearing from
    the current camera is right to a milliradian, the depth to a few percent, and the error moves smoothly from one update to the
    next (measured: at most 4.5 cm per update, the drift of a low-parallax point's depth) — a wrong re-anchoring transform would show
    as a jump of decimetres and break the depth bound."""
    sim = F.simulate(2, sigma=0.0, imu_noise=0.0, perturb=False, sw_size=10)
    ekf = lvo_be.Ekf(sim["cfg"])
    last, stats = {}, dict(bearing=0.0, rel=0.0, jump=0.0, n=0, seen=set())

    def on_update(ts):
        ids, idp, pos = ekf.features()
        if not len(ids):
            return
        assert (idp > 0).all()
        _, p_wc = sim["traj"].cam_pose(ekf.state()["t"])
        true = sim["landmarks"][ids]
        depth = np.linalg.norm(true - p_wc, axis=1)
        err = np.linalg.norm(pos - true, axis=1)
        b1 = (pos - p_wc) / np.linalg.norm(pos - p_wc, axis=1)[:, None]; b2 = (true - p_wc) / depth[:, None]
        stats["bearing"] = max(stats["bearing"], float(np.arccos(np.clip((b1 * b2).sum(1), -1, 1)).max()))
        stats["rel"] = max(stats["rel"], float((err / depth).max()))
        now = {}
        for i, e in zip(ids.tolist(), err.tolist()):
            if i in last:
                stats["jump"] = max(stats["jump"], abs(e - last[i]))
            now[i] = e
        last.clear(); last.update(now)
        stats["n"] += len(ids); stats["seen"].update(ids.tolist())
    n = F.drive(ekf, sim, on_update)
    print("in-state landmarks: bearing %.2e rad, depth %.2f %%, largest step %.4f m; %d feature-updates, %d distinct features"
          % (stats["bearing"], 100 * stats["rel"], stats["jump"], stats["n"], len(stats["seen"])))
    assert n == 60 and len(stats["seen"]) >= 30 and stats["n"] > 1000
    assert stats["bearing"] < 2.5e-3 and stats["rel"] < 0.03 and stats["jump"] < 0.07, stats


def test_monte_carlo_consistency_in_the_reference_regime():
    """IMU noise at the simulator's densities, observation noise of a sub-pixel tracker, the filter configured as config/euroc.yaml
    is (its sigmas are several times the actual ones, and its gate is the 5 % quantile): errors stay at the centimetre level and the
    covariance is conservative — mean NEES below the number of degrees of freedom for position, orientation and velocity."""
    nees, worst = [], 0.0
    for seed in range(1, 9):
        rec, c, n = _run(seed)
        assert n == 60 and c["gated_in"] > 10 * c["gated_out"]
        worst = max(worst, rec[:, 0].max())
        nees.append(rec[:, 3:6].mean(0))
        assert 0.02 < rec[-1, 6] < 0.5                       # the unobservable global position: sigma grows, slowly
    nees = np.array(nees)
    print("max position error %.3f m; mean NEES (p, theta, v) %s; per seed p: %s" % (worst, nees.mean(0).round(2), nees[:, 0].round(2)))
    assert worst < 0.10
    assert (nees.mean(0) < 3.0).all() and (nees < 6.0).all()


def test_nees_reacts_to_a_filter_whose_noise_model_is_wrong():
    """IMU noise 300 x what the filter is told: the estimate wanders by metres and NEES goes to the hundreds"""
    rec, c, n = _run(2, imu_noise=300.0)
    assert rec[:, 3].mean() > 30 and rec[:, 0].max() > 1.0


def _q2R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_static_initializer_and_zupt_from_rest():
    """no set_state: the filter has to start itself.  The trajectory rests for 1.2 s; the static initializer (StaticInitializer.cpp)
    needs static_duration = 1 s of motionless features, then takes gravity from the mean accelerometer reading and the gyro bias from
    the mean rate; zero-velocity updates fire while the platform still rests.  Its world frame is gravity-aligned with an arbitrary
    heading and origin, so what is compared with the truth is what is observable: the direction of gravity, the gyro bias, the
    length and the vertical component of the path flown afterwards."""
    ref_cov = dict(initial_covariance_orientation=4e-4, initial_covariance_velocity=0.25, initial_covariance_position=1.0,
                   initial_covariance_gyro_bias=4e-4, initial_covariance_acc_bias=0.01)                    # config/euroc.yaml's values
    sim = F.simulate(3, t0=0.1, t1=4.6, if_zupt_valid=1, estimate_td=1, estimate_extrin=1, **ref_cov)
    ekf = lvo_be.Ekf(sim["cfg"]); rec = []
    assert not ekf.initialized
    n = F.drive(ekf, sim, lambda ts: rec.append((ts, ekf.state())), set_state=False)
    c = ekf.counters()
    assert ekf.initialized and n >= 30 and c["zupt"] >= 1
    ts0, s0 = rec[0]; ts1, s1 = rec[-1]
    assert 1.05 <= ts0 <= 1.25                                         # one second of rest after the first message at 0.2 s
    assert np.array_equal(s0["p"], np.zeros(3)) and np.abs(s0["bg"]).max() < 5e-4
    tr = sim["traj"]
    for ts, s in rec[::5]:
        up = tr.R_wb(s["t"]) @ _q2R(s["q"]).T @ np.array([0, 0, 1.0])   # the estimate's vertical, seen from the true world
        assert np.hypot(up[0], up[1]) < 1e-2, (ts, up)               # initial sigma is 0.02 rad; measured: 0.4 mrad at start, <= 4.5 mrad in flight
    d_est = s1["p"] - s0["p"]; d_true = tr.p_wb(s1["t"]) - tr.p_wb(s0["t"])
    assert abs(np.linalg.norm(d_est) / np.linalg.norm(d_true) - 1) < 0.05 and abs(d_est[2] - d_true[2]) < 0.03, (d_est, d_true)
