"""The moving-start ("dynamic") initialiser behind lvk_ekf_process (larvio_amd/csrc/be_init.h; DynamicInitializer.cpp /
FlexibleInitializer.cpp:11-25 in the reference): no set_state, no rest at the start - the filter has to find gravity, velocity, metric
scale and the gyroscope bias from a window of 11 feature messages and the IMU samples between them, then run on.

There is no bit-level parity target for this row (the reference takes RANSAC / recoverPose / solvePnP from OpenCV and the bundle
adjustment from Ceres; be_init.h restates the published algorithms, and its header says which).  What is checked is what the reference's
initialiser is FOR: the state it hands to the filter against ground truth (tests/feature_sim.py: a landmark cloud seen from the
synthetic trajectory, observation noise 0.14 px, IMU noise at the configuration's densities), and that the filter it starts then tracks
the truth.  Its world frame is gravity-aligned with an arbitrary heading and origin, so positions are compared after the one
yaw + translation that aligns the first state.  tests/host/init_check.hip pins the building blocks on closed-form cases (CPU suite)."""
import numpy as np
import pytest

from tests import feature_sim as F

pytestmark = pytest.mark.gpu

REF_COV = dict(initial_covariance_orientation=4e-4, initial_covariance_velocity=0.25, initial_covariance_position=1.0,
               initial_covariance_gyro_bias=4e-4, initial_covariance_acc_bias=0.01)                        # config/euroc.yaml's values


def _q2R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _run(gpu_ctx, seed, speed, t0=3.5, t1=8.0, sigma=3e-4, imu_noise=1.0, **cfg):
    import larvio_amd
    from larvio_amd import synthetic as S
    tr = S.Trajectory(speed=speed)                         # the default trajectory's path, `speed` times faster; past its ramp at t0
    sim = F.simulate(seed, t0=t0, t1=t1, sigma=sigma, imu_noise=imu_noise, traj=tr, fresh_ids=True, **dict(REF_COV, **cfg))
    ekf = larvio_amd.LarVio(sim["cfg"], gpu_ctx); assert ekf.initialize()
    imu = sim["imu"]; lo = 0; rec = []; first = None
    for k, (ts, m) in enumerate(sim["msgs"]):
        hi = int(np.searchsorted(imu["t"], ts + 0.05, side="left"))
        upd, rest = ekf.processFeatures((ts, m), imu[lo:hi]); lo = hi - len(rest)
        if upd:
            if first is None:
                first = k
            rec.append((ts, ekf.state()))
        else:
            assert first is None and not ekf.initialized
    c = ekf.counters(); take_off = ekf.take_off_stamp
    ekf.close()
    return sim, tr, first, rec, c, take_off


@pytest.mark.parametrize("seed,speed", [(1, 2.0), (2, 4.0), (3, 3.0)])
def test_moving_start_state_against_ground_truth_and_the_run_after_it(gpu_ctx, seed, speed):
    sim, tr, first, rec, c, take_off = _run(gpu_ctx, seed, speed)
    assert first == 10, first                                 # WINDOW_SIZE + 1 = 11 messages (feature_manager.h:24), the first try succeeds
    ts0, s0 = rec[0]
    assert abs(take_off - ts0) <= 0.0026                      # state time = the last IMU sample within imu_img_timeTh of the message
    worst_up = worst_v = 0.0
    for ts, s in rec:
        Re, Rt = _q2R(s["q"]), tr.R_wb(s["t"])
        worst_up = max(worst_up, np.abs(Re[2] - Rt[2]).max())                       # gravity direction in the body frame
        worst_v = max(worst_v, np.abs(Re.T @ s["v"] - Rt.T @ tr.vel(s["t"])).max())   # body-frame velocity: heading-free
    # aligned path: the yaw + translation that maps the first state onto the truth
    Ra = tr.R_wb(s0["t"]) @ _q2R(s0["q"]).T
    assert abs(Ra[2, 2] - 1) < 1e-4                           # ... is a rotation about the vertical
    pa = tr.p_wb(s0["t"]) - Ra @ s0["p"]
    err = [np.linalg.norm(Ra @ s["p"] + pa - tr.p_wb(s["t"])) for ts, s in rec]
    path = sum(np.linalg.norm(tr.p_wb(rec[i + 1][1]["t"]) - tr.p_wb(rec[i][1]["t"])) for i in range(len(rec) - 1))
    print("moving start seed %d speed %.0f: first update at message %d, %d updates, |v0| %.2f m/s, gravity %.2e, body velocity %.3f m/s, "
          "position (aligned) max %.3f m over %.2f m of path, bg %.1e" % (seed, speed, first, len(rec), np.linalg.norm(tr.vel(s0["t"])), worst_up, worst_v, max(err), path,
                                                                         np.abs(rec[-1][1]["bg"]).max()))
    assert len(rec) >= 30 and c["hybrid"] + c["msckf"] >= 20
    # measured: 1.3e-3 .. 8.3e-3 and 0.017 .. 0.063 m/s.  (With an 8-point refit of the relative pose on the inliers the velocity was within
    # 0.036 m/s - but cv::findFundamentalMat returns the RANSAC's best 7-point model as it is, the bundle adjustment holds the newest
    # frame's translation to it, and the reference lives with that; so does this.)
    assert worst_up < 1e-2 and worst_v < 0.1
    assert max(err) < 0.015 * path                            # measured: 6 .. 32 mm over 3.1 .. 5.3 m


def test_neither_initialiser_starts_on_too_few_tracks(gpu_ctx):
    """18 tracks per message: relativePose wants more than 20 correspondences between some frame and the newest
    (DynamicInitializer.cpp:336-337) and the static initialiser at least 20 common features between consecutive messages
    (StaticInitializer.cpp:52-56) - the window keeps sliding, no update, no state, no IMU sample consumed."""
    import larvio_amd
    from larvio_amd import synthetic as S
    sim = F.simulate(4, t0=3.5, t1=6.0, max_feat=18, traj=S.Trajectory(speed=3.0), fresh_ids=True, **REF_COV)
    ekf = larvio_amd.LarVio(sim["cfg"], gpu_ctx); assert ekf.initialize()
    imu = sim["imu"]
    for ts, m in sim["msgs"]:
        assert len(m) <= 18
        hi = int(np.searchsorted(imu["t"], ts + 0.05, side="left"))
        upd, rest = ekf.processFeatures((ts, m), imu[:hi])
        assert not upd and not ekf.initialized and len(rest) == hi
    ekf.close()
