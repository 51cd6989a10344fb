"""Pins the back-end oracle (oracle/be_*.c) against an INDEPENDENT numpy/scipy float64 implementation of the same
algebra (SVD null space, dense QR, literal K = P H^T S^-1, (I-KH)P) and against closed-form properties.  The reference's
own implementation ships no vectors; since round 5 it is compiled in place against stand-in headers and the oracle is
held to it directly (tests/test_oracle_ref_larvio.py) - these tests remain the independent-algebra pins."""
import numpy as np
import pytest
import scipy.linalg as sla
from scipy.stats import chi2
from oracle import lvo_be


def _rot(rv):
    th = np.linalg.norm(rv)
    if th < 1e-12:
        return np.eye(3)
    k = rv / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _R2q(R):
    t = np.trace(R)
    s = np.sqrt(t + 1) * 2
    return np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])


def _scene(seed, M=6, n_clones=10):
    rng = np.random.default_rng(seed)
    R_b2c = _rot(rng.normal(0, 0.02, 3)) @ np.array([[0, 1.0, 0], [-1, 0, 0], [0, 0, 1]])
    t_c_b = rng.normal(0, 0.05, 3)
    clones = np.zeros(n_clones, lvo_be.CLONE)
    for i in range(n_clones):
        R = _rot(rng.normal(0, 0.08, 3)); p = np.array([0.1 * i, 0.02 * i * i * 0.1, 0.0]) + rng.normal(0, 0.02, 3)
        clones[i]["id"] = 100 + i; clones[i]["q"] = _R2q(R); clones[i]["p"] = p; clones[i]["p_fej"] = p + rng.normal(0, 1e-3, 3)
        clones[i]["R_b2c"] = R_b2c.ravel(); clones[i]["t_c_b"] = t_c_b
        R_c2w = R @ R_b2c.T
        clones[i]["q_cam"] = _R2q(R_c2w); clones[i]["p_cam"] = p + R @ t_c_b
    # a landmark in front of the cameras (camera z = body x for this R_b2c... pick it from the first camera)
    R0 = _rot(np.zeros(3))
    Rc0 = np.array(clones[0]["q_cam"])
    from_q = lambda q: np.array([[1 - 2 * (q[1] ** 2 + q[2] ** 2), 2 * (q[0] * q[1] - q[3] * q[2]), 2 * (q[0] * q[2] + q[3] * q[1])],
                                 [2 * (q[0] * q[1] + q[3] * q[2]), 1 - 2 * (q[0] ** 2 + q[2] ** 2), 2 * (q[1] * q[2] - q[3] * q[0])],
                                 [2 * (q[0] * q[2] - q[3] * q[1]), 2 * (q[1] * q[2] + q[3] * q[0]), 1 - 2 * (q[0] ** 2 + q[1] ** 2)]])
    p_w = clones[0]["p_cam"] + from_q(Rc0) @ np.array([0.3, -0.2, 4.0])
    ranks = np.sort(rng.choice(n_clones, M, replace=False)).astype(np.int32)
    obs = np.zeros((M, 2)); vel = rng.normal(0, 0.05, (M, 2))
    for j, r in enumerate(ranks):
        pc = from_q(clones[r]["q_cam"]).T @ (p_w - clones[r]["p_cam"])
        obs[j] = pc[:2] / pc[2] + rng.normal(0, 0.002, 2)
    return clones, ranks, obs, vel, p_w, from_q


def test_chi2_table_is_the_lower_5_percent_quantile():
    for dof in (1, 2, 9, 57, 99):
        assert lvo_be.chi2_table(dof) == pytest.approx(chi2.ppf(0.05, dof), rel=1e-14)
    assert lvo_be.chi2_table(0) == 0.0 and lvo_be.chi2_table(100) == 0.0       # larvio.cpp:353-357 fills dof 1..99 only


def test_triangulation_recovers_the_landmark():
    clones, ranks, obs, vel, p_w, from_q = _scene(1, M=7)
    poses = np.zeros(len(ranks), lvo_be.POSE)
    for j, r in enumerate(ranks):
        poses[j]["R"] = from_q(clones[r]["q_cam"]).ravel(); poses[j]["t"] = clones[r]["p_cam"]
    ok, pos, sol, idp, oa = lvo_be.triangulate(poses, obs)
    assert ok
    assert np.linalg.norm(pos - p_w) < 0.15                      # 2e-3 observation noise at ~4 m depth
    # consistency of the outputs: position = R_last (alpha, beta, 1)/rho + t_last ; obs_anchor = (alpha, beta, 1)
    Rl = poses[-1]["R"].reshape(3, 3)
    assert np.allclose(pos, Rl @ (np.array([sol[0], sol[1], 1.0]) / sol[2]) + poses[-1]["t"], atol=1e-12)
    assert np.allclose(oa, [sol[0], sol[1], 1.0], atol=1e-12) and idp == pytest.approx(sol[2], rel=1e-12)
    # warm start from the solution converges to (numerically) the same point
    ok2, pos2, *_ = lvo_be.triangulate(poses, obs, use_position=True, position_in=pos)
    assert ok2 and np.linalg.norm(pos2 - pos) < 1e-5
    # a landmark behind the cameras is rejected
    ok3, *_ = lvo_be.triangulate(poses, -obs)
    assert not ok3


def _numeric_msckf(clones, ranks, obs, vel, p_w, from_q, N, leg=22, td=True):
    """independent: numeric differentiation of the measurement function w.r.t. the error state (no FEJ)"""
    M = len(ranks)
    Hx = np.zeros((2 * M, N)); Hf = np.zeros((2 * M, 3)); r = np.zeros(2 * M)

    def h(R_b2w, p_b, R_b2c, t_c_b, pw):
        pc = R_b2c @ R_b2w.T @ (pw - (p_b + R_b2w @ t_c_b))
        return pc[:2] / pc[2]
    eps = 1e-6
    for j, rk in enumerate(ranks):
        c = clones[rk]
        R = from_q(c["q"]); p = np.array(c["p"]); Rbc = c["R_b2c"].reshape(3, 3); tcb = np.array(c["t_c_b"])
        z0 = h(R, p, Rbc, tcb, p_w)
        r[2 * j:2 * j + 2] = obs[j] - z0
        for k in range(3):
            d = np.zeros(3); d[k] = eps
            Hx[2 * j:2 * j + 2, leg + 6 * rk + k] = (h(_rot(d) @ R, p, Rbc, tcb, p_w) - h(_rot(-d) @ R, p, Rbc, tcb, p_w)) / (2 * eps)
            Hx[2 * j:2 * j + 2, leg + 6 * rk + 3 + k] = (h(R, p + d, Rbc, tcb, p_w) - h(R, p - d, Rbc, tcb, p_w)) / (2 * eps)
            # extrinsic rotation error: R_b2c <- R_b2c * R(dq)^T (larvio.cpp:1488-1490)
            Hx[2 * j:2 * j + 2, 15 + k] = (h(R, p, Rbc @ _rot(d).T, tcb, p_w) - h(R, p, Rbc @ _rot(-d).T, tcb, p_w)) / (2 * eps)
            Hx[2 * j:2 * j + 2, 18 + k] = (h(R, p, Rbc, tcb + d, p_w) - h(R, p, Rbc, tcb - d, p_w)) / (2 * eps)
            Hf[2 * j:2 * j + 2, k] = (h(R, p, Rbc, tcb, p_w + d) - h(R, p, Rbc, tcb, p_w - d)) / (2 * eps)
        if td:
            Hx[2 * j:2 * j + 2, 21] = vel[j]
    return Hx, Hf, r


def test_msckf_jacobian_and_nullspace_projection_vs_numeric():
    clones, ranks, obs, vel, p_w, from_q = _scene(2, M=6, n_clones=10)
    N = 22 + 6 * 10 + 3
    H, r = lvo_be.msckf_feature_jacobian(clones, ranks, obs, vel, p_w, N, if_fej=0)
    assert H.shape == (9, N)
    Hx, Hf, rr = _numeric_msckf(clones, ranks, obs, vel, p_w, from_q, N)
    A = sla.null_space(Hf.T)                                   # 12 x 9, the reference's JacobiSVD full-U tail (larvio.cpp:973-976)
    Hn, rn = A.T @ Hx, A.T @ rr
    # both are orthonormal projections onto the same 9-dim subspace: compare basis-invariant quantities
    assert np.allclose(H.T @ H, Hn.T @ Hn, atol=2e-6 * np.abs(Hn.T @ Hn).max())
    assert np.allclose(H.T @ r, Hn.T @ rn, atol=2e-6 * np.abs(Hn.T @ rn).max() + 1e-9)
    rng = np.random.default_rng(3)
    B = rng.normal(0, 1, (N, N)); P = B @ B.T * 1e-4 + np.eye(N) * 1e-6
    g = lvo_be.gating_gamma(H, r, P, 0.008 ** 2)
    gn = rn @ np.linalg.solve(Hn @ P @ Hn.T + 0.008 ** 2 * np.eye(9), rn)
    assert g == pytest.approx(gn, rel=1e-5)


def test_qr_compress_preserves_the_information():
    rng = np.random.default_rng(4)
    H = rng.normal(0, 1, (300, 82)); H[:, :15] = 0.0            # the IMU columns of MSCKF rows are zero (rank deficient)
    r = rng.normal(0, 1, 300)
    R, rc = lvo_be.qr_compress(H, r)
    assert R.shape == (82, 82) and rc.shape == (82,)
    assert np.allclose(np.tril(R, -1), 0, atol=1e-12)
    assert np.allclose(R.T @ R, H.T @ H, atol=1e-10 * 300)
    assert np.allclose(R.T @ rc, H.T @ r, atol=1e-10 * 300)


def test_update_matches_literal_numpy_kalman_update():
    rng = np.random.default_rng(5)
    N, m = 118, 140
    B = rng.normal(0, 1, (N, N)); P = B @ B.T * 1e-3 + np.diag(rng.uniform(1e-8, 1e-2, N))
    H = rng.normal(0, 1, (m, N)) * (rng.uniform(0, 1, (m, N)) < 0.2); H[:, :15] = 0
    r = rng.normal(0, 0.01, m); s2 = 0.008 ** 2
    dx, Pn = lvo_be.ekf_update(P, H, r, s2)
    S = H @ P @ H.T + s2 * np.eye(m)
    K = np.linalg.solve(S, H @ P).T                             # larvio.cpp:1456-1457
    dx_np = K @ r
    P_np = (np.eye(N) - K @ H) @ P; P_np = (P_np + P_np.T) / 2  # :1578-1594
    assert np.allclose(dx, dx_np, rtol=1e-9, atol=1e-14)
    assert np.abs(Pn - P_np).max() <= 1e-9 * np.abs(P_np).max()
    assert np.array_equal(Pn, Pn.T)
    assert np.linalg.eigvalsh(Pn).min() > -1e-12


def test_end_to_end_vio_tracks_ground_truth():
    """sequence level: oracle front-end + oracle back-end on the synthetic sequence, initialised from ground truth,
    must stay within a few centimetres over 2.5 s (sanity of the whole restatement, incl. propagation and pruning)"""
    from oracle import lvo
    from larvio_amd import synthetic as S
    from tests.conftest import synth_frames
    frames = synth_frames(40, 50)
    seq = S.imu_only_sequence()
    fe = lvo.Frontend(S.frontend_config(max_features_num=150))
    be = lvo_be.Ekf(S.backend_config(sw_size=20))
    k_first = int(frames[0][0] * 200) - 2
    imu_all = seq.imu_array(k_first, k_first + 200 * 4)
    ptr, inited, errs = 0, False, []
    for ts, img in frames:
        buf = imu_all[ptr:int(np.searchsorted(imu_all["t"], ts + 0.05))]
        have, msg = fe.process(img, ts, buf)
        if not have:
            continue
        if not inited:
            k = int(np.searchsorted(imu_all["t"], ts, side="right")) - 1
            t0 = imu_all["t"][k]; tr = seq.traj
            be.set_state(t0, _R2q(tr.R_wb(t0)), tr.p_wb(t0), tr.vel(t0), np.zeros(3), np.zeros(3), imu_all["gyro"][k], imu_all["acc"][k])
            inited = True
        ok, used = be.process(ts, msg, buf); ptr += used
        s = be.state()
        errs.append(np.linalg.norm(s["p"] - seq.traj.p_wb(s["t"])))
        P = be.cov()
        assert np.array_equal(P, P.T) and np.linalg.eigvalsh(P).min() > -1e-9
    c = be.counters()
    assert c["hybrid"] >= 10 and c["msckf"] >= 1 and c["gated_in"] > 5 * c["gated_out"]
    assert be.dim > 22 + 6 * 18
    assert max(errs) < 0.08, max(errs)


def _one_imu_step(cfg, imx, bg, ba, n=1):
    """oracle filter after exactly n IMU samples from a ground-truth state (no features: pure propagation + one clone)"""
    from oracle import lvo
    from larvio_amd import synthetic as S
    seq = S.imu_only_sequence()
    imu = seq.imu_array(600, 600 + n + 2)
    tr = seq.traj; t0 = imu["t"][0]
    e = lvo_be.Ekf(cfg)
    e.set_state(t0, _R2q(tr.R_wb(t0)), tr.p_wb(t0), tr.vel(t0), bg, ba, imu["gyro"][0], imu["acc"][0])
    e.set_imu_intrinsics(imx)
    e.process(float(imu["t"][n]), np.zeros(0, lvo.OBS), imu[1:n + 1])
    return e


def _qmul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


def test_imu_intrinsics_transition_blocks_vs_finite_differences():
    """calib_imu_instrinsic = 1 (config 3, LEG_DIM 46): the 24 intrinsic columns and the bias columns of Phi (larvio.cpp:3475-3800)
    against finite differences of the state predictor.  Phi[0:9, c] is read from the cross-covariance after ONE IMU step from the
    diagonal initial covariance (rows c of Phi are identity rows, so P'[0:9, c] = Phi[0:9, c] P0[c, c]).  LARVIO's blocks are
    Simpson/RK-weighted approximations, so the agreement is to a few percent, block by block, with the right signs."""
    from larvio_amd import synthetic as S
    cfg = S.backend_config(sw_size=10, calib_imu_instrinsic=1, if_fej=0)
    base = np.zeros(24); base[3:6] = 1; base[21:24] = 1
    base += np.random.default_rng(0).normal(0, 0.02, 24)              # non-trivial Tg, As, Ma
    bg0, ba0 = np.array([0.01, -0.02, 0.005]), np.array([0.05, 0.02, -0.03])
    e0 = _one_imu_step(cfg, base, bg0, ba0)
    assert e0.dim == 46 + 6
    s0, P = e0.state(), e0.cov()
    P0 = np.zeros(46); P0[9:12] = cfg["initial_covariance_gyro_bias"]; P0[12:15] = cfg["initial_covariance_acc_bias"]; P0[22:46] = 1e-4

    def err_state(s1):
        qi = s0["q"] * np.array([-1, -1, -1, 1]); dq = _qmul(s1["q"], qi)
        return np.concatenate([2 * dq[:3] * np.sign(dq[3]), s1["v"] - s0["v"], s1["p"] - s0["p"]])

    eps = 1e-6
    for c in list(range(9, 15)) + list(range(22, 46)):
        phi_c = P[0:9, c] / P0[c]
        imx, bg, ba = base.copy(), bg0.copy(), ba0.copy()
        if c < 12: bg[c - 9] += eps
        elif c < 15: ba[c - 12] += eps
        else: imx[c - 22] += eps
        num = err_state(_one_imu_step(cfg, imx, bg, ba).state()) / eps
        g0 = 3 * ((c - 22) // 3) + 22 if c >= 22 else (9 if c < 12 else 12)          # first column of this parameter group
        for blk in (slice(0, 3), slice(3, 6), slice(6, 9)):
            scale = np.abs(P[blk, g0:g0 + 3] / P0[c]).max()                            # size of the whole 3x3 block
            assert np.abs(phi_c[blk] - num[blk]).max() <= 0.08 * scale + 1e-13, (c, blk, phi_c[blk], num[blk])


def test_calibrating_filter_tracks_ground_truth_and_keeps_intrinsics_near_identity():
    """whole oracle VIO with calib_imu_instrinsic = 1 on the synthetic sequence (perfect IMU intrinsics): position error stays at the
    centimetre level and the 24 calibration states stay within a few 1e-3 of identity"""
    from oracle import lvo
    from larvio_amd import synthetic as S
    from tests.conftest import synth_frames
    frames = synth_frames(40, 70)
    seq = S.imu_only_sequence()
    ts = [f[0] for f in frames]
    imu_all = seq.imu_array(max(int(ts[0] * 200) - 4, 0), int(ts[-1] * 200) + 40)
    ofe = lvo.Frontend(S.frontend_config(max_features_num=150))
    obe = lvo_be.Ekf(S.backend_config(sw_size=15, if_zupt_valid=0, calib_imu_instrinsic=1))
    lo = 0; n_upd = 0
    for i, (t, img) in enumerate(frames):
        hi = int(np.searchsorted(imu_all["t"], t + 0.05, side="left"))
        if i == 1:
            k = int(np.searchsorted(imu_all["t"], t, side="right")) - 1
            t0 = imu_all["t"][k]; tr = seq.traj
            obe.set_state(t0, _R2q(tr.R_wb(t0)), tr.p_wb(t0), tr.vel(t0), np.zeros(3), np.zeros(3), imu_all["gyro"][k], imu_all["acc"][k])
        buf = imu_all[lo:hi]
        have, m = ofe.process(img, t, buf)
        if have:
            ok, used = obe.process(t, m, buf); lo += used; n_upd += int(ok)
    assert n_upd >= 25 and obe.dim >= 46 + 6 * 10
    s = obe.state()
    # a sanity bound, not a parity pin: 3.5 s / 1.56 m of flight with 24 extra, barely observable calibration states.  Measured:
    # 8.9 cm with the reference's grid bookkeeping (the default since round 6; the filter is held to the reference itself in
    # test_oracle_ref_larvio.py), 6.6 cm with the pre-round-6 one - other features enter the state, neither is "the better filter".
    # What is asked: below 8 % of the distance travelled AND consistent with the filter's own position covariance (3-dof chi-square, 99.9 %).
    e = s["p"] - seq.traj.p_wb(s["t"])
    assert np.linalg.norm(e) < 0.08 * 1.56
    assert e @ np.linalg.solve(obe.cov()[6:9, 6:9], e) < 16.27
    ident = np.zeros(24); ident[3:6] = 1; ident[21:24] = 1
    assert np.abs(obe.imu_intrinsics() - ident).max() < 0.02


def _load_backend_golden():
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "backend_sim.npz"))
    cfg = {}
    for k, v in zip(z["cfg_keys"], z["cfg_vals"]):
        k = str(k)
        cfg[k] = int(v) if (k in lvo_be._CFG_INT or k in ("calib_imu_instrinsic", "feature_idp_dim", "use_schmidt", "max_features")) else float(v)
    cfg["intrinsics"] = tuple(z["intrinsics"]); cfg["T_cam_imu"] = z["T_cam_imu"]
    i = z["init"]
    init = (float(i[0]), i[1:5], i[5:8], i[8:11], i[11:14], i[14:17], i[17:20], i[20:23])
    off = np.concatenate([[0], np.cumsum(z["msg_len"])])
    msgs = [(float(t), z["msg_obs"][off[k]:off[k + 1]]) for k, t in enumerate(z["msg_ts"])]
    return z, cfg, init, msgs


def test_backend_golden_fixture():
    """tests/golden/backend_sim.npz (make_golden.py): stored inputs of a short simulated run AND what the oracle made of them — 19
    updates with hybrid, MSCKF and pruning steps, td and extrinsics estimated.  Any later edit of the oracle that moves a number
    shows here (1e-9: the same C code on another libm may differ in the last bits)."""
    z, cfg, init, msgs = _load_backend_golden()
    ekf = lvo_be.Ekf(cfg)
    ekf.set_state(*init)
    imu = z["imu"]; lo = 0; trace = []
    for ts, m in msgs:
        hi = int(np.searchsorted(imu["t"], ts + 0.05, side="left"))
        upd, used = ekf.process(ts, m, imu[lo:hi]); lo += used
        if upd:
            s = ekf.state(); trace.append(np.concatenate([[s["t"]], s["q"], s["p"], s["v"], [ekf.dim]]))
    trace = np.array(trace)
    assert trace.shape == z["trace"].shape
    assert np.allclose(trace, z["trace"], rtol=1e-9, atol=1e-12)
    s = ekf.state()
    for k in ("bg", "ba", "R_b2c", "t_c_b"):
        assert np.allclose(s[k], z[k], rtol=1e-9, atol=1e-13), k
    assert abs(s["td"] - float(z["td"])) < 1e-12
    assert np.allclose(ekf.cov(), z["cov"], rtol=1e-8, atol=1e-16)
    assert np.array_equal(ekf.clones()["id"], z["clone_ids"])
    ids, idp, _ = ekf.features()
    assert np.array_equal(ids, z["feat_ids"]) and np.allclose(idp, z["feat_idp"], rtol=1e-9)
    c = ekf.counters()
    assert [c[k] for k in ("hybrid", "msckf", "zupt", "gated_in", "gated_out", "map")] == list(z["counters"])


def test_transition_matrix_of_the_plain_model_vs_finite_differences():
    """calib_imu_instrinsic = 0 (LEG_DIM 22, the north-star configuration): after ONE IMU step from the diagonal initial covariance
    P0, the covariance must be Phi P0 Phi^T + Q with Phi = d(error state after) / d(error state before).  Phi's rows for
    (theta, v, p) are taken numerically — perturb the initial orientation (left-multiplied small rotation), velocity, position and
    the two biases, push each through the oracle's own state predictor — and the oracle's covariance block is compared with
    Phi_num P0 Phi_num^T; what may remain is Q (sigma^2 dt on the diagonal) and the second-order terms of LARVIO's closed-form blocks."""
    from oracle import lvo
    from larvio_amd import synthetic as S
    cfg = S.backend_config(sw_size=10, if_fej=0)
    seq = S.imu_only_sequence()
    imu = seq.imu_array(600, 604)
    tr = seq.traj; t0 = imu["t"][0]
    q0, p0, v0 = _R2q(tr.R_wb(t0)), tr.p_wb(t0), tr.vel(t0)
    bg0, ba0 = np.array([0.01, -0.02, 0.005]), np.array([0.05, 0.02, -0.03])

    def step(q, p, v, bg, ba):
        e = lvo_be.Ekf(cfg)
        e.set_state(t0, q, p, v, bg, ba, imu["gyro"][0], imu["acc"][0])
        e.process(float(imu["t"][1]), np.zeros(0, lvo.OBS), imu[1:2])
        return e
    e0 = step(q0, p0, v0, bg0, ba0)
    s0, P = e0.state(), e0.cov()
    assert e0.dim == 22 + 6

    def err_state(s1):
        qi = s0["q"] * np.array([-1, -1, -1, 1]); dq = _qmul(s1["q"], qi)
        return np.concatenate([2 * dq[:3] * np.sign(dq[3]), s1["v"] - s0["v"], s1["p"] - s0["p"]])
    eps = 1e-6
    Phi = np.zeros((9, 15))
    for c in range(15):
        q, p, v, bg, ba = q0.copy(), p0.copy(), v0.copy(), bg0.copy(), ba0.copy()
        if c < 3:
            d = np.zeros(4); d[c] = 0.5 * eps; d[3] = 1.0; q = _qmul(d, q0); q /= np.linalg.norm(q)
        elif c < 6: v[c - 3] += eps
        elif c < 9: p[c - 6] += eps
        elif c < 12: bg[c - 9] += eps
        else: ba[c - 12] += eps
        Phi[:, c] = err_state(step(q, p, v, bg, ba).state()) / eps
    D = np.zeros(15)
    D[0:3] = cfg["initial_covariance_orientation"]; D[3:6] = cfg["initial_covariance_velocity"]; D[6:9] = cfg["initial_covariance_position"]
    D[9:12] = cfg["initial_covariance_gyro_bias"]; D[12:15] = cfg["initial_covariance_acc_bias"]
    want = Phi @ np.diag(D) @ Phi.T
    got = P[0:9, 0:9]
    dt = float(imu["t"][1] - t0)
    q_bound = 4 * max(cfg["noise_gyro"], cfg["noise_acc"]) ** 2 * dt
    scale = np.sqrt(np.outer(np.diag(want), np.diag(want)))
    assert np.abs(got - want).max() <= q_bound + 2e-3 * scale.max(), (np.abs(got - want).max(), q_bound)
    assert (np.abs(got - want) <= q_bound + 2e-2 * scale).all()
    # and the cross-covariance with the biases gives Phi's bias columns directly (their own rows are identity rows)
    for c in range(9, 15):
        col = P[0:9, c] / D[c]
        assert np.abs(col - Phi[:, c]).max() <= 0.05 * np.abs(Phi[:, c]).max() + 1e-12, (c, col, Phi[:, c])
    # stateAugmentation (larvio.cpp:752-798): the clone is the IMU pose, J selects (theta, p): its covariance rows are copies
    sel = [0, 1, 2, 6, 7, 8]
    assert np.array_equal(P[22:28, 0:22], P[sel, 0:22]) and np.array_equal(P[22:28, 22:28], P[np.ix_(sel, sel)])


# ---------------------------------------------------------------------------------------------------------------------
# In-state (1-D inverse depth) features: the measurement Jacobian and the re-anchoring Jacobian against numeric differentiation
# of the geometry they linearise.  Error-state conventions as in the MSCKF pin above: rotation error multiplies from the left
# (R <- R(d) R), positions add, the extrinsic rotation error acts as R_b2c <- R_b2c R(d)^T (larvio.cpp:1476-1575).


def _camera(R_b2w, p_b, R_b2c, t_c_b):
    """camera-to-world rotation and camera position of a clone (larvio.cpp:1529-1541)"""
    return R_b2w @ R_b2c.T, p_b + R_b2w @ t_c_b


def _idp_point(R_a, p_a, R_b2c, t_c_b, f_an, rho):
    """world position of a feature with bearing f_an = (alpha, beta, 1) and inverse depth rho in the camera of anchor clone a"""
    Rc, tc = _camera(R_a, p_a, R_b2c, t_c_b)
    return Rc @ (f_an / rho) + tc


def _project(R_k, p_k, R_b2c, t_c_b, pw):
    Rc, tc = _camera(R_k, p_k, R_b2c, t_c_b)
    pc = Rc.T @ (pw - tc)
    return pc[:2] / pc[2]


def _two_clones(seed):
    clones, ranks, obs, vel, p_w, from_q = _scene(seed, M=6, n_clones=10)
    a, k = clones[2].copy(), clones[7].copy()
    R_b2c = a["R_b2c"].reshape(3, 3).copy(); t_c_b = np.array(a["t_c_b"])
    Ra, pa, Rk, pk = from_q(a["q"]), np.array(a["p"]), from_q(k["q"]), np.array(k["p"])
    Rca, tca = _camera(Ra, pa, R_b2c, t_c_b)
    pc = Rca.T @ (p_w - tca)
    f_an = np.array([pc[0] / pc[2], pc[1] / pc[2], 1.0]); rho = 1.0 / pc[2]
    return a, k, R_b2c, t_c_b, Ra, pa, Rk, pk, f_an, rho, p_w


def test_in_state_feature_jacobian_vs_numeric():
    """measurementJacobian_ekf_1didp (larvio.cpp:1117-1244): H_f (inverse depth), H_a (anchor clone), H_x (observing clone), H_e (extrinsics)"""
    for seed in (4, 5, 6):
        a, k, R_b2c, t_c_b, Ra, pa, Rk, pk, f_an, rho, p_w = _two_clones(seed)
        z = _project(Rk, pk, R_b2c, t_c_b, p_w) + np.array([0.003, -0.002])
        ok, Hf, Ha, Hx, He, r = lvo_be.ekf1d_obs_jacobian(k, a, p_w, rho, f_an[:2], z)
        assert ok

        def h(dth_a=np.zeros(3), dp_a=np.zeros(3), dth_k=np.zeros(3), dp_k=np.zeros(3), dth_e=np.zeros(3), dp_e=np.zeros(3), drho=0.0):
            Rbc = R_b2c @ _rot(dth_e).T; tcb = t_c_b + dp_e
            pw = _idp_point(_rot(dth_a) @ Ra, pa + dp_a, Rbc, tcb, f_an, rho + drho)
            return _project(_rot(dth_k) @ Rk, pk + dp_k, Rbc, tcb, pw)
        assert np.allclose(r, z - h(), atol=1e-12)
        eps = 1e-6
        num = lambda name, i: (h(**{name: np.eye(3)[i] * eps}) - h(**{name: -np.eye(3)[i] * eps})) / (2 * eps)
        nHa = np.column_stack([num("dth_a", i) for i in range(3)] + [num("dp_a", i) for i in range(3)])
        nHx = np.column_stack([num("dth_k", i) for i in range(3)] + [num("dp_k", i) for i in range(3)])
        nHe = np.column_stack([num("dth_e", i) for i in range(3)] + [num("dp_e", i) for i in range(3)])
        nHf = (h(drho=eps * rho) - h(drho=-eps * rho)) / (2 * eps * rho)
        scale = max(np.abs(nHx).max(), 1.0)
        assert np.allclose(Hf, nHf, atol=2e-6 * max(np.abs(nHf).max(), 1.0)), (seed, Hf, nHf)
        assert np.allclose(Ha, nHa, atol=2e-6 * scale), (seed, Ha - nHa)
        assert np.allclose(Hx, nHx, atol=2e-6 * scale), (seed, Hx - nHx)
        assert np.allclose(He, nHe, atol=2e-6 * scale), (seed, He - nHe)
    # the anchor's own observation carries no information in 1-D mode (larvio.cpp:1199-1207): flagged, residual still returned
    ok, *_rest, r = lvo_be.ekf1d_obs_jacobian(a, a, p_w, rho, f_an[:2], f_an[:2])
    assert not ok and np.allclose(r, 0, atol=1e-12)


def test_reanchoring_jacobian_vs_numeric():
    """updateFeatureCov_1didp (larvio.cpp:3125-3293): the new inverse depth as a function of the old one, both anchor clones and the extrinsics"""
    for seed in (7, 8, 9):
        a, k, R_b2c, t_c_b, Ra, pa, Rk, pk, f_an, rho, p_w = _two_clones(seed)
        Rck, tck = _camera(Rk, pk, R_b2c, t_c_b)
        rho_new0 = 1.0 / (Rck.T @ (p_w - tck))[2]
        o, n = a.copy(), k.copy()
        J = lvo_be.reanchor_row(o, n, R_b2c, t_c_b, p_w, rho_new0)

        def g(dth_o=np.zeros(3), dp_o=np.zeros(3), dth_n=np.zeros(3), dp_n=np.zeros(3), dth_e=np.zeros(3), dp_e=np.zeros(3), drho=0.0):
            Rbc = R_b2c @ _rot(dth_e).T; tcb = t_c_b + dp_e
            pw = _idp_point(_rot(dth_o) @ Ra, pa + dp_o, Rbc, tcb, f_an, rho + drho)
            Rc, tc = _camera(_rot(dth_n) @ Rk, pk + dp_n, Rbc, tcb)
            return 1.0 / (Rc.T @ (pw - tc))[2]
        assert g() == pytest.approx(rho_new0, rel=1e-12)
        eps = 1e-6
        num = lambda name, i: (g(**{name: np.eye(3)[i] * eps}) - g(**{name: -np.eye(3)[i] * eps})) / (2 * eps)
        nJ = np.array([(g(drho=eps * rho) - g(drho=-eps * rho)) / (2 * eps * rho)]
                      + [num("dth_o", i) for i in range(3)] + [num("dp_o", i) for i in range(3)]
                      + [num("dth_n", i) for i in range(3)] + [num("dp_n", i) for i in range(3)]
                      + [num("dth_e", i) for i in range(3)] + [num("dp_e", i) for i in range(3)])
        assert np.allclose(J, nJ, atol=3e-6 * max(np.abs(nJ).max(), 1.0)), (seed, J - nJ)


def test_delayed_initialisation_equals_a_diffuse_prior_update():
    """measurementUpdate_hybrid with new in-state features (larvio.cpp:1605-1862, delayed initialisation :1821-1854): update with the
    rows that do not involve the new features, then dx_new = -HH dx + H2^-1 r1, P_new,old = -HH P, P_new,new = HH P HH^T + sigma^2 H2^-2.
    Independent statement of the same estimate: ONE standard EKF update on the state augmented by the new features under a diffuse
    (variance -> infinity) prior, with all rows at once."""
    rng = np.random.default_rng(12)
    N, m, n_acc, sigma2 = 40, 25, 3, 0.01 ** 2
    B = rng.normal(0, 1, (N, N)); P = B @ B.T * 1e-3 + np.eye(N) * 1e-5
    Ho = rng.normal(0, 1, (m, N)); ro = rng.normal(0, 0.02, m)
    H1 = rng.normal(0, 1, (n_acc, N)); H2 = rng.uniform(0.5, 3.0, n_acc) * rng.choice([-1, 1], n_acc); r1 = rng.normal(0, 0.02, n_acc)
    Pn, dx = lvo_be.hybrid_update_with_new(P, Ho, ro, H1, H2, r1, sigma2)
    kappa = 1e9
    Pa = np.zeros((N + n_acc, N + n_acc)); Pa[:N, :N] = P; Pa[N:, N:] = np.eye(n_acc) * kappa
    Ha = np.zeros((m + n_acc, N + n_acc)); Ha[:m, :N] = Ho; Ha[m:, :N] = H1; Ha[m:, N:] = np.diag(H2)
    ra = np.concatenate([ro, r1])
    S = Ha @ Pa @ Ha.T + sigma2 * np.eye(m + n_acc)
    K = Pa @ Ha.T @ np.linalg.inv(S)
    dxa = K @ ra
    Pp = (np.eye(N + n_acc) - K @ Ha) @ Pa
    Pp = (Pp + Pp.T) / 2
    assert np.allclose(dx, dxa, atol=1e-6 * np.abs(dxa).max())
    assert np.allclose(Pn, Pp, atol=1e-6 * np.abs(Pp).max())
    assert np.allclose(Pn, Pn.T, atol=0)
