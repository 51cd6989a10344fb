"""GPU parity of the back-end: FP64-MFMA algebra against numpy, the stage-level update / compression against the oracle,
and the device-resident LarVio against the oracle's, update by update (state and covariance within 1e-5 relative —
the tolerance BASELINE.json's north_star states)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL = 1e-5


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("shape", [(232, 262, 232), (17, 5, 33), (64, 64, 64), (300, 1, 7), (216, 217, 110), (160, 160, 216), (440, 434, 433), (33, 47, 192), (33, 47, 193), (20, 20, 257)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False)])
def test_dgemm_mfma_f64(gpu_ctx, shape, ta, tb):
    from larvio_amd import larvio as lv
    M, N, K = shape
    rng = np.random.default_rng(M * 1000 + N)
    A = rng.normal(0, 1, (K, M) if ta else (M, K)); B = rng.normal(0, 1, (N, K) if tb else (K, N)); C0 = rng.normal(0, 1, (M, N))
    ref = 0.7 * (A.T if ta else A) @ (B.T if tb else B) - 0.3 * C0
    out = lv.dgemm(gpu_ctx, A, B, ta, tb, alpha=0.7, beta=-0.3, Cin=C0)
    assert np.abs(out - ref).max() < 1e-12 * K
    # an asymmetric B with A = I catches a row/column swap in the f64 accumulator map
    if not ta and not tb and M == K:
        out = lv.dgemm(gpu_ctx, np.eye(M), B)
        assert np.array_equal(out, B)


def _update_problem(seed, N, m):
    rng = np.random.default_rng(seed)
    Bm = rng.normal(0, 1, (N, N)); P = Bm @ Bm.T * 1e-3 + np.diag(rng.uniform(1e-8, 1e-2, N))
    H = rng.normal(0, 1, (m, N)) * (rng.uniform(0, 1, (m, N)) < 0.2); H[:, :15] = 0
    r = rng.normal(0, 0.01, m)
    return P, H, r


@pytest.mark.parametrize("N,m", [(232, 262), (118, 40), (46, 9), (232, 530), (250, 1), (232, 160), (232, 161), (232, 192), (240, 320), (330, 321), (460, 450), (470, 485)])
def test_ekf_update_matches_oracle(gpu_ctx, N, m):
    from oracle import lvo_be
    from larvio_amd import larvio as lv
    P, H, r = _update_problem(N + m, N, m)
    dx_o, P_o = lvo_be.ekf_update(P, H, r, 0.008 ** 2)
    dx_g, P_g = lv.ekf_update(gpu_ctx, P, H, r, 0.008 ** 2)
    assert _rel(dx_g, dx_o) < 1e-9
    assert _rel(P_g, P_o) < 1e-10
    assert np.array_equal(P_g, P_g.T)                              # P - W^T W is symmetric by construction
    assert np.linalg.eigvalsh(P_g).min() > -1e-12


@pytest.mark.parametrize("N,m,bad", [(120, 40, 7), (232, 150, 149), (232, 200, 170), (232, 330, 5)])
def test_ekf_update_reports_a_non_positive_definite_innovation(gpu_ctx, N, m, bad):
    """k_chol_fused writes the first non-positive pivot into its report words; the stage-level call returns LVK_ERR_NUMERIC (5)
    instead of factoring on with pivot := 1 (what nobody used to read), and the same context goes on working afterwards.
    The indefinite S comes from an indefinite P: one direction with a large negative variance, seen by measurement row `bad`."""
    from larvio_amd import larvio as lv
    from larvio_amd._lib import LvkError
    P, H, r = _update_problem(N + m, N, m)
    H[:, 30] = 0.0; H[bad, :] = 0.0; H[bad, 30] = 1.0       # only row `bad` sees direction 30, and it sees nothing else
    P[30, :] = 0.0; P[:, 30] = 0.0; P[30, 30] = -1.0
    with pytest.raises(LvkError) as ei:
        lv.ekf_update(gpu_ctx, P, H, r, 0.008 ** 2)
    assert "not positive definite" in str(ei.value), str(ei.value)
    # rows before `bad` do not see the negative direction: the reported pivot is exactly that row
    assert f"pivot {bad})" in str(ei.value), str(ei.value)
    P2, H2, r2 = _update_problem(1, N, m)
    dx, Pn = lv.ekf_update(gpu_ctx, P2, H2, r2, 0.008 ** 2)
    assert np.isfinite(dx).all() and np.isfinite(Pn).all()


@pytest.mark.parametrize("rows,cols", [(300, 202), (700, 202), (2500, 82), (18000, 120), (150, 202), (3000, 442), (3536, 452), (9000, 300), (1500, 500),
                                       (513, 31), (8192, 64), (8193, 64), (40000, 33), (600, 599)])
def test_qr_compression_preserves_information(gpu_ctx, rows, cols):
    """lvk_ekf_compress_qr (be_qr_dense.hip: CAQR, NB = 32 up to 8192 rows, NB = 16 above; one and two levels per panel; panels and
    chunks with ragged ends; rank-deficient leading columns) against numpy's H^T H and H^T r"""
    from larvio_amd import larvio as lv
    rng = np.random.default_rng(rows)
    H = rng.normal(0, 1, (rows, cols)); H[:, :15] = 0.0
    if cols > 100:
        H[:, 70] = 0.0; H[:, 90] = H[:, 80] * 2.0                  # an interior zero column and an exactly dependent one
    r = rng.normal(0, 1, rows)
    R, rc = lv.compress_qr(gpu_ctx, H, r)
    assert R.shape[0] == min(rows, cols)
    G = H.T @ H
    assert np.abs(R.T @ R - G).max() < 1e-11 * np.abs(G).max() * np.sqrt(rows)
    assert np.abs(R.T @ rc - H.T @ r).max() < 1e-11 * np.abs(H.T @ r).max() * np.sqrt(rows)
    if rows > cols:
        assert np.abs(np.tril(R, -1)).max() < 1e-12 * np.abs(R).max()


@pytest.mark.parametrize("case", ["steady_A", "burst_5", "steady_5", "ragged"])
def test_structure_aware_qr_preserves_information(gpu_ctx, case):
    """lvk_ekf_compress_qr_groups (k_qr_sparse: LDS-resident Householder TSQR over row groups with known column sets) against numpy's
    H^T H and H^T r on the shapes of tests/test_qr_plan.py, including the 17,000-row burst of configs[4]; then the dense finish."""
    from larvio_amd import larvio as lv
    from tests.test_qr_plan import _msckf_like
    if case == "steady_A":
        N, groups, H, r = _msckf_like(1, 25, 30, n_state_feat=30)
    elif case == "burst_5":
        N, groups, H, r = _msckf_like(2, 1900, 60, n_state_feat=60, burst=True)
    elif case == "steady_5":
        N, groups, H, r = _msckf_like(3, 330, 60, n_state_feat=60)
    else:                                                                    # single-row groups, a rank-deficient node, an all-zero group
        N, groups, H, r = _msckf_like(6, 120, 24, track=3)
        H[10:13] = 0.0; r[10:13] = 0.0
        H[40:49, 15:22] = 0.0
    G0, g0 = H.T @ H, H.T @ r
    Hc, rc = lv.compress_qr_groups(gpu_ctx, H, r, groups)
    levels, final_rows = lv.qr_plan(N, groups)
    assert len(Hc) == final_rows < len(H)
    G1, g1 = Hc.T @ Hc, Hc.T @ rc
    assert np.abs(G1 - G0).max() <= 1e-11 * np.abs(G0).max() * np.sqrt(len(H)), np.abs(G1 - G0).max() / np.abs(G0).max()
    assert np.abs(g1 - g0).max() <= 1e-11 * np.abs(g0).max() * np.sqrt(len(H))
    Hd, rd = lv.compress_qr(gpu_ctx, Hc, rc)                                 # dense finish (a no-op when rows <= cols)
    assert len(Hd) <= N and np.abs(Hd.T @ Hd - G0).max() <= 1e-10 * np.abs(G0).max() * np.sqrt(len(H))
    print(case, len(H), "->", len(Hc), "->", len(Hd), "rel", np.abs(G1 - G0).max() / np.abs(G0).max())


@pytest.mark.parametrize("n_rows,n_live", [(190, 1), (400, 1), (190, 2), (64, 1), (700, 3)])
def test_structure_aware_qr_when_the_gate_rejected_almost_every_row(gpu_ctx, n_rows, n_live):
    """The pruning update of a filter that is rejecting nearly everything (a poor moving start: whole-program fuzz, second and third
    profile, "pivot 0 of 19 rows"): hundreds of one-row blocks over the SAME 19 columns (extrinsics, td, the two clones that leave),
    all zeroed by the gate but one or two.  The node's matrix then has rank 1 or 2; every further Householder step works on what the
    step before left of an exactly cancelled column - 1e-17, 1e-34, ... of the entries - and by the 18th column the sum of squares is a
    denormal number, where 2 / |v|^2 overflowed (and the fast reciprocal square root of the register kernel returned garbage): NaN
    in all 19 rows handed to the update, reported as a non-positive pivot.  Asked: finite rows that carry the same information."""
    from larvio_amd import larvio as lv
    rng = np.random.default_rng(n_rows + n_live)
    N = 94; cols = list(range(15, 22)) + list(range(22 + 6 * 3, 22 + 6 * 5))          # 7 + 12 = 19 columns
    H = np.zeros((n_rows, N)); r = np.zeros(n_rows)
    for i in rng.choice(n_rows, n_live, replace=False):
        H[i, cols] = rng.normal(0, 15, len(cols)); r[i] = rng.normal(0, 0.01)
    groups = [(1, cols)] * n_rows
    G0, g0 = H.T @ H, H.T @ r
    Hc, rc = lv.compress_qr_groups(gpu_ctx, H, r, groups)
    assert np.isfinite(Hc).all() and np.isfinite(rc).all(), "NaN rows out of the compression"
    assert len(Hc) < len(H) and np.abs(Hc.T @ Hc - G0).max() <= 1e-11 * np.abs(G0).max() and np.abs(Hc.T @ rc - g0).max() <= 1e-11 * np.abs(g0).max()
    Hd, rd = lv.compress_qr(gpu_ctx, np.vstack([Hc, np.zeros((N, N))]), np.concatenate([rc, np.zeros(N)]))     # and the dense kernel on the same rank-1 system
    assert np.isfinite(Hd).all() and np.abs(Hd.T @ Hd - G0).max() <= 1e-11 * np.abs(G0).max()


def _R2q(R):
    t = np.trace(R); s = np.sqrt(t + 1) * 2
    return np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])


def _messages(first, count, max_features=150):
    """feature messages of the ORACLE front-end on the synthetic sequence (the back-end parity input)"""
    from oracle import lvo
    from larvio_amd import synthetic as S
    from tests.conftest import synth_frames
    frames = synth_frames(first, count)
    seq = S.imu_only_sequence()
    fe = lvo.Frontend(S.frontend_config(max_features_num=max_features))
    k_first = max(int(frames[0][0] * 200) - 2, 0)
    imu_all = seq.imu_array(k_first, k_first + 200 * (count // 20 + 2))
    msgs = []
    for ts, img in frames:
        buf = imu_all[:int(np.searchsorted(imu_all["t"], ts + 0.05))]
        have, msg = fe.process(img, ts, buf[-60:])
        if have:
            msgs.append((ts, msg))
    return msgs, imu_all, seq


def _run_pair(gpu_ctx, msgs, imu_all, seq, cfg, init_from_gt=True, init_args=None, on_update=None):
    from oracle import lvo_be
    import larvio_amd
    ora = lvo_be.Ekf(cfg)
    gpu = larvio_amd.LarVio(cfg, gpu_ctx)
    assert gpu.initialize()
    buf_o = imu_all.copy(); buf_g = imu_all.copy()
    inited = not init_from_gt
    if init_args is not None:
        ora.set_state(*init_args); gpu.set_state(*init_args); inited = True
    n_upd, worst_x, worst_P = 0, 0.0, 0.0
    for ts, msg in msgs:
        bo = buf_o[:int(np.searchsorted(buf_o["t"], ts + 0.05))]
        bg = buf_g[:int(np.searchsorted(buf_g["t"], ts + 0.05))]
        if not inited:
            k = int(np.searchsorted(imu_all["t"], ts, side="right")) - 1
            t0 = imu_all["t"][k]; tr = seq.traj
            args = (t0, _R2q(tr.R_wb(t0)), tr.p_wb(t0), tr.vel(t0), np.zeros(3), np.zeros(3), imu_all["gyro"][k], imu_all["acc"][k])
            ora.set_state(*args); gpu.set_state(*args)
            inited = True
        ok_o, used_o = ora.process(ts, msg, bo)
        ok_g, rest = gpu.processFeatures((ts, msg), bg)
        used_g = len(bg) - len(rest)
        assert ok_g == ok_o and used_g == used_o
        buf_o = buf_o[used_o:]; buf_g = buf_g[used_g:]
        if not ok_o:
            continue
        n_upd += 1
        assert gpu.dim == ora.dim
        so, sg = ora.state(), gpu.state()
        for k in ("q", "v", "p", "bg", "ba", "R_b2c", "t_c_b"):
            worst_x = max(worst_x, _rel(np.asarray(sg[k]), np.asarray(so[k])))
        assert abs(sg["td"] - so["td"]) <= REL * max(abs(so["td"]), 1e-3)
        Pi = gpu.cov_imu(9)                              # getPpose / getPvel's block, from the update's host-side mirror: read BEFORE anything moves the matrix
        Po, Pg = ora.cov(), gpu.cov()
        assert np.array_equal(Pi, Pg[:9, :9])            # ... and bit for bit the covariance's own leading block
        worst_P = max(worst_P, _rel(Pg, Po))
        co, cg = ora.clones(), gpu.clones()
        assert np.array_equal(cg["id"], co["id"])
        assert _rel(cg["p"], co["p"]) < REL and _rel(cg["q"], co["q"]) < REL
        io, do_, po = ora.features(); ig, dg, pg = gpu.features()
        assert np.array_equal(ig, io)
        if len(io):
            assert _rel(dg, do_) < REL and _rel(pg, po) < REL
        assert worst_x < REL and worst_P < REL, (n_upd, worst_x, worst_P)
        if on_update is not None:
            on_update(sg, Pg, ig)                         # (the HIP filter's state, covariance and in-state feature ids after this update)
    cg, co = gpu.counters(), ora.counters()
    for k in ("hybrid", "msckf", "zupt", "gated_in", "gated_out", "map"):
        assert cg[k] == co[k], (k, cg, co)
    gpu.close()
    return n_upd, worst_x, worst_P, co, ora


def test_backend_sequence_parity_hybrid(gpu_ctx):
    """EuRoC-shaped hybrid MSCKF / 1-D EKF-SLAM, sw_size 20: window fills, prunes, re-anchors; state and P within 1e-5"""
    from larvio_amd import synthetic as S
    msgs, imu_all, seq = _messages(40, 90)
    cfg = S.backend_config(sw_size=20, if_zupt_valid=0)
    n_upd, wx, wP, c, ora = _run_pair(gpu_ctx, msgs, imu_all, seq, cfg)
    assert n_upd >= 40 and c["hybrid"] >= 30 and c["msckf"] >= 5
    assert len(ora.features()[0]) >= 5                               # in-state (EKF-SLAM) features exist
    print("hybrid parity: updates", n_upd, "max rel state", wx, "max rel P", wP, c)


def test_backend_sequence_parity_pure_msckf_and_static_init(gpu_ctx):
    """configs[0] flavour: max_features_in_one_grid 0 (pure MSCKF), start at rest: static initializer + ZUPT updates"""
    from larvio_amd import synthetic as S
    msgs, imu_all, seq = _messages(0, 70)
    cfg = S.backend_config(sw_size=12, max_features_in_one_grid=0)
    n_upd, wx, wP, c, ora = _run_pair(gpu_ctx, msgs, imu_all, seq, cfg, init_from_gt=False)
    assert n_upd >= 15 and c["zupt"] >= 1 and c["hybrid"] + c["msckf"] >= 5
    assert ora.dim == 22 + 6 * len(ora.clones())


def test_triangulation_stage_bit_exact(gpu_ctx):
    """lvk_triangulate vs the oracle: sums are taken in view order on both sides, so the results are the same bits"""
    from oracle import lvo_be
    from larvio_amd import larvio as lv
    from tests.test_oracle_backend import _scene
    for seed, M in ((1, 7), (5, 3), (9, 12), (11, 2)):
        clones, ranks, obs, vel, p_w, from_q = _scene(seed, M=M, n_clones=14)
        poses = np.zeros(len(ranks), lvo_be.POSE)
        for j, r in enumerate(ranks):
            poses[j]["R"] = from_q(clones[r]["q_cam"]).ravel(); poses[j]["t"] = clones[r]["p_cam"]
        for o in (obs, -obs):                                            # -obs: behind the cameras, must be rejected on both sides
            ok_o, pos_o, sol_o, idp_o, oa_o = lvo_be.triangulate(poses, o)
            ok_g, pos_g, sol_g, idp_g, oa_g = lv.triangulate(gpu_ctx, poses, o)
            assert ok_g == ok_o
            if ok_o:                                                      # a rejected feature's outputs are never read (larvio.cpp:1915-1920)
                assert np.array_equal(pos_g, pos_o) and np.array_equal(sol_g, sol_o) and idp_g == idp_o and np.array_equal(oa_g, oa_o)
        ok_o, pos_o, *_ = lvo_be.triangulate(poses, obs, use_position=True, position_in=p_w)
        ok_g, pos_g, *_ = lv.triangulate(gpu_ctx, poses, obs, use_position=True, position_in=p_w)
        assert ok_g == ok_o and np.array_equal(pos_g, pos_o)


def test_gate_and_stack_stage_matches_oracle(gpu_ctx):
    """lvk_ekf_gate_and_stack: per-feature Jacobian rows, null-space projection and chi-square gate vs the oracle's
    featureJacobian_msckf / gatingTest.  Both use the same Householder sequence: rows agree to 1e-10; gate decisions identical."""
    from oracle import lvo_be
    from larvio_amd import larvio as lv
    from tests.test_oracle_backend import _scene
    n_clones = 12; N = 22 + 6 * n_clones + 4
    rng = np.random.default_rng(77)
    Bm = rng.normal(0, 1, (N, N)); P = Bm @ Bm.T * 2e-6 + np.eye(N) * 1e-7
    sigma2 = 0.008 ** 2
    clones = None; feats = []; ranks_all = []; obs_all = []; vel_all = []; ref = []
    for k, (seed, M) in enumerate(((2, 6), (3, 3), (4, 12), (6, 2), (8, 9))):
        c, ranks, obs, vel, p_w, _ = _scene(seed, M=M, n_clones=n_clones)
        if clones is None:
            clones = c
        else:                                                            # same window for every feature: re-project into the first scene's clones
            from_q = _
            for j, r in enumerate(ranks):
                pc = from_q(clones[r]["q_cam"]).T @ (p_w - clones[r]["p_cam"])
                obs[j] = pc[:2] / pc[2] + rng.normal(0, 0.004 * (1 + 3 * (k == 2)), 2)
        feats.append((p_w, M, len(ranks_all)))
        ranks_all += list(ranks); obs_all += list(obs); vel_all += list(vel)
        H, r = lvo_be.msckf_feature_jacobian(clones, ranks, obs, vel, p_w, N)
        g = lvo_be.gating_gamma(H, r, P, sigma2)
        ref.append((H, r, g, g < lvo_be.chi2_table(2 * M - 3)))
    Hg, rg, gamma, acc = lv.gate_and_stack(gpu_ctx, clones, feats, ranks_all, np.array(obs_all), np.array(vel_all), P, sigma2)
    assert list(acc) == [bool(x[3]) for x in ref]
    assert any(acc) and not all(acc)                                     # the batch exercises both outcomes
    for k, x in enumerate(ref):
        assert gamma[k] == pytest.approx(x[2], rel=1e-8)
    Ho = np.vstack([x[0] for x in ref if x[3]]); ro = np.concatenate([x[1] for x in ref if x[3]])
    assert Hg.shape == Ho.shape
    assert np.abs(Hg - Ho).max() < 1e-10 * max(np.abs(Ho).max(), 1) and np.abs(rg - ro).max() < 1e-10


def test_backend_sequence_parity_imu_intrinsics_calibration(gpu_ctx):
    """config 3 flavour: calib_imu_instrinsic 1 => LEG_DIM 46 (the 24 IMU-intrinsic states in the filter), online extrinsics + td.
    State, covariance and the calibration states within 1e-5 relative of the oracle after every update."""
    from larvio_amd import synthetic as S
    msgs, imu_all, seq = _messages(40, 80)
    cfg = S.backend_config(sw_size=16, if_zupt_valid=0, calib_imu_instrinsic=1)
    import larvio_amd
    from oracle import lvo_be
    n_upd, wx, wP, c, ora = _run_pair(gpu_ctx, msgs, imu_all, seq, cfg)
    assert n_upd >= 35 and ora.dim >= 46 + 6 * 14
    # the calibration states themselves: replay on a fresh pair and compare them at the end
    gpu = larvio_amd.LarVio(cfg, gpu_ctx); assert gpu.initialize()
    ora2 = lvo_be.Ekf(cfg)
    buf = imu_all.copy(); first = True
    for ts, msg in msgs:
        b = buf[:int(np.searchsorted(buf["t"], ts + 0.05))]
        if first:
            k = int(np.searchsorted(imu_all["t"], ts, side="right")) - 1
            t0 = imu_all["t"][k]; tr = seq.traj
            a = (t0, _R2q(tr.R_wb(t0)), tr.p_wb(t0), tr.vel(t0), np.zeros(3), np.zeros(3), imu_all["gyro"][k], imu_all["acc"][k])
            ora2.set_state(*a); gpu.set_state(*a); first = False
        ok_o, used = ora2.process(ts, msg, b); gpu.processFeatures((ts, msg), b)
        buf = buf[used:]
    xo, xg = ora2.imu_intrinsics(), gpu.imu_intrinsics()
    assert np.abs(xo - np.array([0, 0, 0, 1, 1, 1] + [0] * 15 + [1, 1, 1])).max() > 1e-7        # they did move
    assert np.abs(xg - xo).max() < 1e-5 * max(np.abs(xo).max(), 1.0)
    gpu.close()
    print("calibration parity: updates", n_upd, "max rel state", wx, "max rel P", wP)


def test_sharded_update_equals_unsharded_on_device(gpu_ctx):
    """SURVEY 8e on ONE GPU: the per-rank work of the sharded MSCKF update (each "rank" builds and gates the rows of a contiguous
    feature range with lvk_ekf_gate_and_stack, reduces them to its packed n x n triangle with lvk_ekf_compress_qr) followed by
    the rank-ordered stack -> QR -> lvk_ekf_update gives the update of the unsharded rows.  The all-gather of the packed
    triangles between the two halves is exercised by tests/test_sharding_gloo.py; here the exchange is a concatenation."""
    from larvio_amd import larvio as lv
    from larvio_amd import sharding as sh
    from tests.test_oracle_backend import _scene
    n_clones = 14; n = 22 + 6 * n_clones; N = n + 6                       # six in-state feature columns the MSCKF rows do not touch
    rng = np.random.default_rng(5)
    Bm = rng.normal(0, 1, (N, N)); P = Bm @ Bm.T * 2e-6 + np.eye(N) * 1e-7
    sigma2 = 0.008 ** 2
    clones = None; feats = []; ranks_all = []; obs_all = []; vel_all = []
    for k in range(60):                                                   # 60 features x ~15 rows: taller than wide (900 > 112)
        c, ranks, obs, vel, p_w, from_q = _scene(100 + k, M=int(rng.integers(6, 12)), n_clones=n_clones)
        if clones is None:
            clones = c
        p_w = clones[0]["p_cam"] + from_q(clones[0]["q_cam"]) @ np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1, 1), rng.uniform(3, 7)])
        for j, r in enumerate(ranks):
            pc = from_q(clones[r]["q_cam"]).T @ (p_w - clones[r]["p_cam"])
            obs[j] = pc[:2] / pc[2] + rng.normal(0, 0.0008, 2)
        feats.append((p_w, len(ranks), len(ranks_all)))
        ranks_all += list(ranks); obs_all += list(obs); vel_all += list(vel)
    obs_all = np.array(obs_all); vel_all = np.array(vel_all)
    # unsharded: all accepted rows -> (compress) -> update
    H, r, gamma, acc = lv.gate_and_stack(gpu_ctx, clones, feats, ranks_all, obs_all, vel_all, P, sigma2)
    assert acc.sum() >= 40 and H.shape[0] > N
    Hc, rc = lv.compress_qr(gpu_ctx, H, r)
    dx_ref, P_ref = lv.ekf_update(gpu_ctx, P, Hc, rc, sigma2)
    # sharded over 2 and over 4 "ranks"
    for world in (2, 4):
        blocks = []
        for lo, hi in sh.shard_ranges([2 * f[1] - 3 for f in feats], world):
            Hs, rs, _, _ = lv.gate_and_stack(gpu_ctx, clones, feats[lo:hi], ranks_all, obs_all, vel_all, P, sigma2)
            Rs, qs = lv.compress_qr(gpu_ctx, Hs, rs)                       # rank-local TSQR: <= N rows
            Rg, qg = sh.unpack_upper(sh.pack_upper(Rs, qs), N)             # the wire format (packed upper triangle + rhs)
            blocks.append((Rg, qg))
        Hst = np.vstack([b[0] for b in blocks]); rst = np.concatenate([b[1] for b in blocks])
        Hf, rf = lv.compress_qr(gpu_ctx, Hst, rst)
        dx, Pn = lv.ekf_update(gpu_ctx, P, Hf, rf, sigma2)
        assert _rel(dx, dx_ref) < 1e-8 and _rel(Pn, P_ref) < 1e-9, (world, _rel(dx, dx_ref), _rel(Pn, P_ref))


@pytest.mark.parametrize("seed,kw", [(4, dict()), (5, dict(sigma=2e-3, imu_noise=30.0, sw_size=12, max_feat=60)),
                                     (6, dict(calib_imu_instrinsic=1, estimate_td=1, estimate_extrin=1, max_feat=220))])
def test_backend_parity_on_simulated_feature_messages(gpu_ctx, seed, kw):
    """tests/feature_sim.py: landmark cloud + trajectory -> feature messages, no front-end involved.  A different input
    distribution from the tracker's (every feature lives until it leaves the field of view, a full budget in every message, no
    initial-frame observations), 6 s = 60 updates; the second case is the stressed one (noisy IMU, 2/3 of the features gated out,
    short window), the third calibrates IMU intrinsics, extrinsics and time offset with a larger budget."""
    from tests import feature_sim as F
    sim = F.simulate(seed, **kw)

    class _Seq:
        traj = sim["traj"]
    n_upd, wx, wP, c, ora = _run_pair(gpu_ctx, sim["msgs"], sim["imu"], _Seq, sim["cfg"], init_args=sim["init"])
    assert n_upd == 60 and c["hybrid"] >= 20
    print("simulated features", seed, kw, "worst rel state", wx, "cov", wP, c)


def test_backend_parity_config5_depth(gpu_ctx):
    """BASELINE.json configs[4] at real depth, without rendering 1080p frames: 2000 features per message, a 60-clone window (N up to
    22 + 360 + 60), 72 updates.  Every sixth message all features of a generation reach max_track_len together: ~18,000 raw rows
     >> N columns, so the QR compression (larvio.cpp:1430-1445) runs at its largest shape, then the window fills and the
    pruning update + re-anchoring run at 60 clones.  State and covariance within 1e-5 after every update; gate counters identical."""
    import os
    from oracle import lvo
    from tests import feature_sim as F
    sim = F.simulate(8, t0=2.0, t1=9.2, max_feat=2000, n_per_batch=500, sw_size=60, max_features_in_one_grid=2, estimate_td=1, estimate_extrin=1,
                     max_features=2000)

    class _Seq:
        traj = sim["traj"]
    assert min(len(m) for _, m in sim["msgs"][2:]) >= 1800
    lvo.set_threads(min(32, os.cpu_count() or 1))
    try:
        n_upd, wx, wP, c, ora = _run_pair(gpu_ctx, sim["msgs"], sim["imu"], _Seq, sim["cfg"], init_args=sim["init"])
    finally:
        lvo.set_threads(1)
    assert n_upd >= 70 and len(ora.clones()) >= 58 and ora.dim >= 22 + 6 * 58 + 30 and c["msckf"] >= 3, (n_upd, ora.dim, c)
    print("config-5 depth: updates", n_upd, "N", ora.dim, "worst rel state", wx, "cov", wP, c)


def test_gate_quirk_for_dof_100_and_more(gpu_ctx):
    """larvio.cpp:353-357 fills chi_squared_test_table for dof 1..99 only; gatingTest (:1873) reads table[dof] through std::map's
    operator[], i.e. 0.0 for dof >= 100, so a feature with >= 52 observations can never pass.  With max_track_len 58 in a 60-clone
    window the long tracks hit exactly that: both sides must reject them (and keep accepting the short ones)."""
    from tests import feature_sim as F
    sim = F.simulate(9, t0=2.0, t1=9.4, max_feat=40, sw_size=60, max_track_len=58, max_features_in_one_grid=0)

    class _Seq:
        traj = sim["traj"]
    n_upd, wx, wP, c, ora = _run_pair(gpu_ctx, sim["msgs"], sim["imu"], _Seq, sim["cfg"], init_args=sim["init"])
    assert n_upd >= 70 and c["gated_out"] >= 5 and c["gated_in"] >= 20, c
    print("dof >= 100 quirk: updates", n_upd, "worst rel", wx, wP, c)


def test_take_lost_features_hands_out_every_lost_in_state_feature_once(gpu_ctx):
    """lvk_ekf_take_lost_features = getStableMapPointPositions (larvio.cpp:2717-2722, filled at :3342): every in-state feature that
    leaves the state because its track was lost shows up exactly once, with the last world position the filter held for it, and the
    list is cleared by the read."""
    import larvio_amd
    from larvio_amd import synthetic as S
    msgs, imu_all, seq = _messages(40, 110)
    cfg = S.backend_config(sw_size=20, if_zupt_valid=0)
    gpu = larvio_amd.LarVio(cfg, gpu_ctx); assert gpu.initialize()
    buf = imu_all.copy(); first = True
    prev = {}; lost_expected = {}; got = {}
    for ts, msg in msgs:
        b = buf[:int(np.searchsorted(buf["t"], ts + 0.05))]
        if first:
            k = int(np.searchsorted(imu_all["t"], ts, side="right")) - 1
            t0 = imu_all["t"][k]; tr = seq.traj
            gpu.set_state(t0, _R2q(tr.R_wb(t0)), tr.p_wb(t0), tr.vel(t0), np.zeros(3), np.zeros(3), imu_all["gyro"][k], imu_all["acc"][k]); first = False
        upd, rest = gpu.processFeatures((ts, msg), b)
        buf = buf[len(b) - len(rest):]
        ids, _, pos = gpu.features()
        now = {int(i): p for i, p in zip(ids, pos)}
        tracked = set(int(x) for x in msg["id"])
        for i in prev:
            if i not in now and i not in tracked:
                lost_expected[i] = prev[i]                                   # dropped because the tracker lost it (rmLostFeaturesCov)
        prev = now
        if len(lost_expected) and len(got) < len(lost_expected) and (len(lost_expected) % 2 == 0):
            li, lp = gpu.stable_map_points()
            for i, p in zip(li, lp):
                assert int(i) not in got
                got[int(i)] = p
            assert len(gpu.stable_map_points()[0]) == 0                      # cleared by the read
    li, lp = gpu.stable_map_points()
    for i, p in zip(li, lp):
        assert int(i) not in got
        got[int(i)] = p
    assert len(lost_expected) >= 3 and set(got) == set(lost_expected), (sorted(got), sorted(lost_expected))
    for i in got:                                                            # the position at the last update before the loss
        assert np.abs(got[i] - lost_expected[i]).max() < 1e-9
    gpu.close()


def test_backend_accepts_empty_feature_messages(gpu_ctx):
    """messages with zero features between normal ones (the ABI allows them): pure propagation + augmentation + pruning"""
    from larvio_amd import synthetic as S
    msgs, imu_all, seq = _messages(40, 60)
    import numpy as _np
    from oracle import lvo
    thin = [(ts, m if (i % 3) else _np.zeros(0, lvo.OBS)) for i, (ts, m) in enumerate(msgs)]
    cfg = S.backend_config(sw_size=12, if_zupt_valid=0)
    n_upd, wx, wP, c, ora = _run_pair(gpu_ctx, thin, imu_all, seq, cfg)
    assert n_upd >= 25


def test_border_features_follow_the_references_grid_map(gpu_ctx, monkeypatch):
    """grid codes beyond the image bounds (ref larvio.cpp:1969-1975, 3351-3370; larvio.h:383: grid_map is a std::map, such a code gets a
    cell of its own that never clears).  The HIP filter's DEFAULT against the oracle's default (which agrees with the reference's own
    filter on this stream: tests/test_oracle_ref_larvio.py::test_features_beyond_the_image_bounds_get_cells_of_their_own) on a moving
    start whose border features make the old and the reference's bookkeeping part (100 rendered frames, 200 tracks); then the opt-out
    (LVK_GRID_REFERENCE=0 / reference_grid = 0) on both sides, and the two modes must NOT agree with each other on this stream."""
    from tests.test_oracle_ref_larvio import _tracker_stream
    monkeypatch.delenv("LVK_GRID_REFERENCE", raising=False)
    sim = _tracker_stream(30, 100, 200, 15, sw_size=30, max_features_in_one_grid=1)

    class _Seq:
        traj = None
    hist = {"ref": [], "old": []}
    n_upd, wx, wP, c, ora = _run_pair(gpu_ctx, sim["msgs"], sim["imu"], _Seq, sim["cfg"], init_args=sim["init"],
                                      on_update=lambda s_, P_, ids: hist["ref"].append(tuple(int(i) for i in ids)))
    assert n_upd >= 45 and c["hybrid"] >= 40
    print("reference grid (default): updates", n_upd, "worst rel state", wx, "cov", wP, c)
    monkeypatch.setenv("LVK_GRID_REFERENCE", "0")
    n2, wx2, wP2, c2, ora2 = _run_pair(gpu_ctx, sim["msgs"], sim["imu"], _Seq, dict(sim["cfg"], reference_grid=0), init_args=sim["init"],
                                       on_update=lambda s_, P_, ids: hist["old"].append(tuple(int(i) for i in ids)))
    print("legacy grid (opt-out): updates", n2, "worst rel state", wx2, "cov", wP2, c2)
    assert n2 >= 45
    assert hist["ref"] != hist["old"]                     # the stream does tell the two bookkeepings apart (in-state feature ids differ)
