"""TEST INFRASTRUCTURE ONLY - never imported by the product (larvio_amd/).  An independent restatement of LARVIO's moving-start
initialiser in numpy / scipy, the checker for larvio_amd/csrc/be_init.h (row N4 of SURVEY 8):

    DynamicInitializer.cpp           tryDynInit / processIMU / processImage / initialStructure / relativePose / visualInitialAlign / slideWindow
    include/Initializer/ImuPreintegration.h   mid-point pre-integration (only delta_p / delta_q / delta_v and d(delta_q)/d(b_g) are ever read)
    feature_manager.cpp              track bookkeeping of the window (addFeatureCheckParallax, getCorresponding, removeBack)
    solve_5pts.cpp                   solveRelativeRT: findFundamentalMat + recoverPose
    initial_sfm.cpp                  GlobalSFM::construct: PnP / triangulation chain + bundle adjustment
    initial_alignment.cpp            solveGyroscopeBias, LinearAlignment, RefineGravity

PINNED, as far as the reference's own text goes, to the reference compiled in place (oracle/_ref/liblvref_dyninit.so: DynamicInitializer.cpp,
initial_sfm.cpp, solve_5pts.cpp, initial_alignment.cpp, feature_manager.cpp; tests/test_oracle_ref_dyninit.py) - its orchestration, with
OpenCV's and Ceres' minimisers replaced by stand-ins that find the same minima; beyond the minimisers' tolerances nothing is pinned.  This file is
independent of be_init.h where independence is possible: library SVD / least squares (numpy.linalg, scipy.optimize.least_squares with a
rotation-vector parametrisation) instead of the product's hand-written Jacobi SVD, Levenberg-Marquardt PnP and Schur-complement bundle
adjustment; the bookkeeping (which samples, which frames, which gauge) is restated from the same reference lines.  Both minimise the same
costs, so their results agree to the minimisers' tolerances - which is also all that can be said about the reference's own.
findFundamentalMat's RANSAC is replaced by "every correspondence is an inlier" here and in the replay harness the CPU test drives
(tests/host/init_replay.hip); the RANSAC kernel itself is pinned bit for bit elsewhere (oracle/fe_track.c, tests/test_gpu_frontend_stages.py)."""
import numpy as np
from scipy.optimize import least_squares
from scipy.sparse import lil_matrix
from scipy.spatial.transform import Rotation

WINDOW_SIZE = 10                                  # feature_manager.h:24
G_NORM = 9.81                                     # ImuPreintegration.h:19


def _skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def _qmul(a, b):                                   # Hamilton product, [x y z w]
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def _q2R(q):                                       # Eigen's toRotationMatrix formula, valid as written for a non-unit q as well
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


class PreInt:
    """IntegrationBase (ImuPreintegration.h:27-230), mid-point rule; linearised about zero accelerometer bias and the given gyro bias"""

    def __init__(self, acc0, gyr0, bg):
        self.lin_acc, self.lin_gyr = np.array(acc0, float), np.array(gyr0, float)
        self.buf = []; self.bg = np.array(bg, float)
        self._reset()

    def _reset(self):
        self.acc0, self.gyr0 = self.lin_acc.copy(), self.lin_gyr.copy()
        self.sum_dt = 0.0; self.dp = np.zeros(3); self.dv = np.zeros(3); self.dq = np.array([0, 0, 0, 1.0]); self.J = np.zeros((3, 3))

    @staticmethod
    def _qv(q, v):                                  # Eigen's Quaternion * Vector3 (_transformVector): exact rotation only for a unit q
        uv = 2 * np.cross(q[:3], v)
        return v + q[3] * uv + np.cross(q[:3], uv)

    def _propagate(self, dt, a1, g1):              # midPointIntegration (:62-142)
        un0 = self._qv(self.dq, self.acc0)
        w = 0.5 * (self.gyr0 + g1) - self.bg
        q1 = _qmul(self.dq, np.array([w[0] * dt / 2, w[1] * dt / 2, w[2] * dt / 2, 1.0]))
        un1 = self._qv(q1, np.asarray(a1, float))   # (:77-78: the product with the not yet normalised quaternion)
        ua = 0.5 * (un0 + un1)
        self.dp = self.dp + self.dv * dt + 0.5 * ua * dt * dt
        self.dv = self.dv + ua * dt
        self.J = (np.eye(3) - _skew(w) * dt) @ self.J - np.eye(3) * dt          # rows O_R of F / V (:104-105,127)
        self.dq = q1 / np.linalg.norm(q1)
        self.sum_dt += dt; self.acc0, self.gyr0 = np.array(a1, float), np.array(g1, float)

    def push_back(self, dt, a, g):
        self.buf.append((dt, np.array(a, float), np.array(g, float))); self._propagate(dt, a, g)

    def repropagate(self, bg):                     # (:48-61)
        self.bg = np.array(bg, float); self._reset()
        for dt, a, g in self.buf:
            self._propagate(dt, a, g)


# ------------------------------------------------------------------------------------------------ two-view geometry
def eight_point(p1, p2):
    """Hartley-normalised 8-point F with x2^T F x1 = 0 (what findFundamentalMat's final refit on the inliers computes)"""
    def norm(p):
        c = p.mean(0); s = np.sqrt(2) / np.mean(np.linalg.norm(p - c, axis=1))
        return (p - c) * s, np.array([[s, 0, -c[0] * s], [0, s, -c[1] * s], [0, 0, 1.0]])
    a, T1 = norm(p1); b, T2 = norm(p2)
    A = np.stack([b[:, 0] * a[:, 0], b[:, 0] * a[:, 1], b[:, 0], b[:, 1] * a[:, 0], b[:, 1] * a[:, 1], b[:, 1], a[:, 0], a[:, 1], np.ones(len(a))], 1)
    F = np.linalg.svd(A)[2][-1].reshape(3, 3)
    U, S, Vt = np.linalg.svd(F); S[2] = 0
    return T2.T @ (U * S) @ Vt @ T1


def _triangulate(P0, P1, a, b):
    M = np.stack([a[0] * P0[2] - P0[0], a[1] * P0[2] - P0[1], b[0] * P1[2] - P1[0], b[1] * P1[2] - P1[1]])
    X = np.linalg.svd(M)[2][-1]
    return X[:3] / X[3]


def recover_pose(E, p1, p2, mask=None):
    """recoverPose (solve_5pts.cpp:29-181) with the identity camera: (R, t) with x2 ~ R x1 + t, the candidate with the most points (of
    those the incoming mask keeps) in front of both cameras"""
    U, _, Vt = np.linalg.svd(E)
    if np.linalg.det(U) < 0: U = -U
    if np.linalg.det(Vt) < 0: Vt = -Vt
    W = np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1.0]])
    R1, R2, t0 = U @ W @ Vt, U @ W.T @ Vt, U[:, 2]
    best = None
    for R, t in ((R1, t0), (R2, t0), (R1, -t0), (R2, -t0)):         # the order of preference of solve_5pts.cpp:150-180 (first of equals wins)
        P1 = np.hstack([R, t[:, None]]); P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
        good = 0
        for k, (a, b) in enumerate(zip(p1, p2)):
            if mask is not None and not mask[k]:
                continue
            X = _triangulate(P0, P1, a, b); z2 = (R @ X + t)[2]
            good += (0 < X[2] < 50) and (0 < z2 < 50)
        if best is None or good > best[0]:
            best = (good, R, t)
    return best


# ------------------------------------------------------------------------------------------------ PnP, bundle adjustment (scipy)
def _pnp(X, z, R0, t0):
    """cv::solvePnP(..., useExtrinsicGuess): minimise the reprojection error from (R0, t0); x_cam = R X + t"""
    def res(p):
        Pc = Rotation.from_rotvec(p[:3]).apply(X) + p[3:]
        return (Pc[:, :2] / Pc[:, 2:3] - z).ravel()
    s = least_squares(res, np.concatenate([Rotation.from_matrix(R0).as_rotvec(), t0]), method="lm", xtol=1e-14, ftol=1e-14, gtol=1e-14)
    return Rotation.from_rotvec(s.x[:3]).as_matrix(), s.x[3:]


def _bundle(nf, l, cR, cT, pts, obs):
    """initial_sfm.cpp:232-316: camera-from-reference poses (cR[i], cT[i]) and points; rotation of l and translations of l, nf-1 held.
    obs: list of (point index, frame, (x, y)).  Returns refined (cR, cT, pts)."""
    free_r = [i for i in range(nf) if i != l]; free_t = [i for i in range(nf) if i not in (l, nf - 1)]
    ir = {f: 3 * k for k, f in enumerate(free_r)}; it = {f: 3 * len(free_r) + 3 * k for k, f in enumerate(free_t)}
    op = 3 * len(free_r) + 3 * len(free_t)
    x0 = np.concatenate([np.concatenate([Rotation.from_matrix(cR[f]).as_rotvec() for f in free_r]), np.concatenate([cT[f] for f in free_t]), pts.ravel()])
    pi = np.array([o[0] for o in obs]); fi = np.array([o[1] for o in obs]); zz = np.array([o[2] for o in obs])

    def unpack(x):
        Rs = [None] * nf; Ts = [None] * nf
        for f in range(nf):
            Rs[f] = Rotation.from_rotvec(x[ir[f]:ir[f] + 3]).as_matrix() if f in ir else cR[f]
            Ts[f] = x[it[f]:it[f] + 3] if f in it else cT[f]
        return np.array(Rs), np.array(Ts), x[op:].reshape(-1, 3)

    def res(x):
        Rs, Ts, P = unpack(x)
        Pc = np.einsum("nij,nj->ni", Rs[fi], P[pi]) + Ts[fi]
        return (Pc[:, :2] / Pc[:, 2:3] - zz).ravel()
    S = lil_matrix((2 * len(obs), len(x0)), dtype=int)
    for k, (p, f, _) in enumerate(obs):
        for r in (2 * k, 2 * k + 1):
            if f in ir: S[r, ir[f]:ir[f] + 3] = 1
            if f in it: S[r, it[f]:it[f] + 3] = 1
            S[r, op + 3 * p:op + 3 * p + 3] = 1
    s = least_squares(res, x0, jac_sparsity=S, method="trf", x_scale="jac", xtol=1e-12, ftol=1e-12, gtol=1e-12, max_nfev=60)
    s = least_squares(res, s.x, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15)       # MINPACK with a dense Jacobian to finish: trf's last digits are slow
    Rs, Ts, P = unpack(s.x)
    return Rs, Ts, P, s.cost


def global_sfm(nf, l, relR, relT, tracks):
    """GlobalSFM::construct (initial_sfm.cpp:131-330).  tracks: list of dict(frame -> (x, y)) in feature order.  Returns world-from-camera
    (R[i], T[i]) in the frame of camera l, or None."""
    cR = [None] * nf; cT = [None] * nf
    cR[l] = np.eye(3); cT[l] = np.zeros(3)
    cR[nf - 1] = relR.T; cT[nf - 1] = -relR.T @ relT
    pos = [None] * len(tracks)
    P = lambda i: np.hstack([cR[i], cT[i][:, None]])

    def tri_two(f0, f1):
        for j, tr in enumerate(tracks):
            if pos[j] is None and f0 in tr and f1 in tr:
                pos[j] = _triangulate(P(f0), P(f1), tr[f0], tr[f1])

    def pnp(i, R0, t0):
        idx = [j for j, tr in enumerate(tracks) if pos[j] is not None and i in tr]
        if len(idx) < 10:
            return False
        X = np.array([pos[j] for j in idx], np.float32).astype(np.float64)          # cv::Point3f / Point2f
        z = np.array([tracks[j][i] for j in idx], np.float32).astype(np.float64)
        cR[i], cT[i] = _pnp(X, z, R0, t0)
        return True
    for i in range(l, nf - 1):
        if i > l and not pnp(i, cR[i - 1], cT[i - 1]):
            return None
        tri_two(i, nf - 1)
    for i in range(l + 1, nf - 1):
        tri_two(l, i)
    for i in range(l - 1, -1, -1):
        if not pnp(i, cR[i + 1], cT[i + 1]):
            return None
        tri_two(i, l)
    for j, tr in enumerate(tracks):
        if pos[j] is None and len(tr) >= 2:
            fs = sorted(tr)
            pos[j] = _triangulate(P(fs[0]), P(fs[-1]), tr[fs[0]], tr[fs[-1]])
    idx = [j for j in range(len(tracks)) if pos[j] is not None]
    obs = [(k, f, tracks[j][f]) for k, j in enumerate(idx) for f in sorted(tracks[j])]
    Rs, Ts, _, cost = _bundle(nf, l, cR, cT, np.array([pos[j] for j in idx]), obs)
    return [Rs[i].T for i in range(nf)], [-Rs[i].T @ Ts[i] for i in range(nf)], len(idx), cost


# ------------------------------------------------------------------------------------------------ visual-inertial alignment
def _alignment(frames, TIC, g0=None):
    """LinearAlignment (ng = 3, g0 None) / one RefineGravity pass (ng = 2) (initial_alignment.cpp:65-201): normal equations as the
    reference accumulates them, solved with a library solver"""
    nfr = len(frames); ng = 3 if g0 is None else 2; n = nfr * 3 + ng + 1
    A = np.zeros((n, n)); b = np.zeros(n)
    if g0 is not None:
        a = g0 / np.linalg.norm(g0); tmp = np.array([1.0, 0, 0]) if np.array_equal(a, [0, 0, 1.0]) else np.array([0, 0, 1.0])
        bb = tmp - a * (a @ tmp); bb /= np.linalg.norm(bb); lxly = np.stack([bb, np.cross(a, bb)], 1)
    for i in range(nfr - 1):
        fi, fj = frames[i], frames[i + 1]; dt = fj["pre"].sum_dt
        tA = np.zeros((6, 6 + ng + 1)); tb = np.zeros(6)
        Rit = fi["R"].T
        tA[0:3, 0:3] = -dt * np.eye(3)
        tA[3:6, 0:3] = -np.eye(3); tA[3:6, 3:6] = Rit @ fj["R"]
        G = Rit if g0 is None else Rit @ lxly
        tA[0:3, 6:6 + ng] = G * dt * dt / 2; tA[3:6, 6:6 + ng] = G * dt
        tA[0:3, 6 + ng] = Rit @ (fj["T"] - fi["T"]) / 100.0
        Rg = np.zeros(3) if g0 is None else Rit @ g0
        tb[0:3] = fj["pre"].dp + Rit @ fj["R"] @ TIC - TIC - Rg * dt * dt / 2
        tb[3:6] = fj["pre"].dv - Rg * dt
        cols = list(range(3 * i, 3 * i + 6)) + list(range(n - ng - 1, n))
        A[np.ix_(cols, cols)] += tA.T @ tA; b[cols] += tA.T @ tb
    return A, b, (None if g0 is None else lxly)


def visual_imu_alignment(frames, TIC, Bg):
    """VisualIMUAlignment (:204-212).  frames: dicts R, T, pre (frames[1:] carry the pre-integration from their predecessor); Bg: the
    window's gyro bias, updated IN PLACE even when the alignment then fails (solveGyroscopeBias adds delta_bg to every Bgs[i] - they are
    one vector - and re-propagates with it before LinearAlignment has its say).  Returns (g, x) or None."""
    A = np.zeros((3, 3)); b = np.zeros(3)
    for fi, fj in zip(frames[:-1], frames[1:]):                                      # solveGyroscopeBias (:12-46)
        qij = Rotation.from_matrix(fi["R"].T @ fj["R"]).as_quat()
        dq = fj["pre"].dq; qe = _qmul(np.array([-dq[0], -dq[1], -dq[2], dq[3]]), qij)
        if qe[3] < 0: qe = -qe                                                      # (Eigen's Quaterniond(Matrix3d) returns w >= 0 for these small rotations)
        J = fj["pre"].J
        A += J.T @ J; b += J.T @ (2 * qe[:3])
    Bg += np.linalg.solve(A, b)
    for f in frames[1:]:
        f["pre"].repropagate(Bg)
    A, b, _ = _alignment(frames, TIC)
    x = np.linalg.solve(A * 1000.0, b * 1000.0)
    n = len(x); g = x[n - 4:n - 1]; s = x[n - 1] / 100.0
    if abs(np.linalg.norm(g) - G_NORM) > 1.0 or s < 0:
        return None
    g0 = g / np.linalg.norm(g) * G_NORM
    Aacc = bacc = 0.0                                # RefineGravity never clears its normal equations between the four passes (:74-77, 120-122)
    for _ in range(4):
        A, b, lxly = _alignment(frames, TIC, g0)
        Aacc = (Aacc + A) * 1000.0; bacc = (bacc + b) * 1000.0
        x = np.linalg.solve(Aacc, bacc)
        dg = x[-3:-1]
        g0 = g0 + lxly @ dg; g0 = g0 / np.linalg.norm(g0) * G_NORM
    x = x.copy(); x[-1] = x[-1] / 100.0
    if x[-1] < 0:
        return None
    return g0, x


# ------------------------------------------------------------------------------------------------ the replay
def dynamic_init(msgs, imu, R_b2c, t_c_b, td=0.0, imu_img_time_th=1.0 / 400, fundamental=None):
    """Feed feature messages [(ts, structured array with id/u/v/u_vel/v_vel)] and the IMU stream as LarVio::processFeatures would
    (DynamicInitializer.cpp:20-44) until the initialiser succeeds.  -> dict of the successful attempt's results, or None.
    fundamental(p1, p2, thresh, conf) -> (mask, F): the stand-in for cv::findFundamentalMat (oracle.lvo.find_fundamental is the real one);
    None = "every correspondence is an inlier, F = their 8-point fit"."""
    RIC = np.asarray(R_b2c, float).T; TIC = np.asarray(t_c_b, float)
    lower = 0.0; ddt = 0.0; first_imu = False; frame_count = 0; initial_ts = 0.0
    acc0 = gyr0 = None; curr_time = -1.0
    Times = [0.0] * (WINDOW_SIZE + 1)
    tmp_pre = None
    Bg = np.zeros(3)                                 # Bgs[0..WINDOW_SIZE]: one vector (see visual_imu_alignment)
    frames = {}                                      # all_image_frame: key ts + td
    tracks = []                                      # FeatureManager::feature: dict(id, start, pts list)
    last = None
    for mi, (ts, m) in enumerate(msgs):
        bound = ts + td
        sel = imu[imu["t"] < ts + 0.05]
        for s in sel:
            if s["t"] <= lower: continue
            if s["t"] - bound > imu_img_time_th: break
            ddt = s["t"] - bound
            a, g = np.array(s["acc"], float), np.array(s["gyro"], float)
            if not first_imu:
                first_imu = True; acc0, gyr0, curr_time = a, g, s["t"]
            dt = s["t"] - curr_time
            if tmp_pre is None:
                tmp_pre = PreInt(acc0, gyr0, Bg)
            if frame_count != 0:
                tmp_pre.push_back(dt, a, g)
            acc0, gyr0, curr_time = a, g, s["t"]; last = (g, a)
        lower = bound + imu_img_time_th
        # processImage (:100-133)
        tdd = td + ddt
        for o in m:
            p = (o["u"] + o["u_vel"] * tdd, o["v"] + o["v_vel"] * tdd); fid = int(np.int32(np.uint64(o["id"]) & np.uint64(0xFFFFFFFF)))
            tr = next((t for t in tracks if t["id"] == fid), None)
            if tr is None: tracks.append(dict(id=fid, start=frame_count, pts=[p]))
            else: tr["pts"].append(p)
        Times[frame_count] = ts
        frames[ts + td] = dict(pre=tmp_pre)
        tmp_pre = PreInt(acc0, gyr0, Bg) if acc0 is not None else None
        if frame_count < WINDOW_SIZE:
            frame_count += 1
            continue
        ok = None
        if ts - initial_ts > 0.1:
            ok = _initial_structure(tracks, frames, Times, td, RIC, TIC, Bg, fundamental)
            initial_ts = ts
        if ok is not None:
            state_time = Times[WINDOW_SIZE] + td + ddt
            ok.update(message=mi, state_time=state_time, erase=int(np.searchsorted(sel["t"], state_time, side="right")), last_gyro=last[0], last_acc=last[1])
            return ok
        # slideWindow, MARGIN_OLD (:362-402) - the only branch that can occur (MIN_PARALLAX = 10/460 = 0, feature_manager.h:25)
        Times[:-1] = Times[1:]
        t0 = Times[0] + td
        for k in [k for k in frames if k < t0]:
            del frames[k]
        for tr in list(tracks):                      # removeBack (feature_manager.cpp:205-222)
            if tr["start"] != 0: tr["start"] -= 1
            else:
                tr["pts"].pop(0)
                if not tr["pts"]: tracks.remove(tr)
    return None


def _initial_structure(tracks, frames, Times, td, RIC, TIC, Bg, fundamental=None):
    nf = WINDOW_SIZE + 1
    tr_maps = [{t["start"] + k: p for k, p in enumerate(t["pts"])} for t in tracks]
    # relativePose (:330-359)
    l = None
    for i in range(WINDOW_SIZE):
        c = [(m[i], m[WINDOW_SIZE]) for m in tr_maps if i in m and WINDOW_SIZE in m and min(m) <= i]
        if len(c) <= 20: continue
        a = np.array([x[0] for x in c]); b = np.array([x[1] for x in c])
        if np.mean(np.linalg.norm(a - b, axis=1)) * 460 <= 30 or len(c) < 15: continue
        a32, b32 = a.astype(np.float32).astype(np.float64), b.astype(np.float32).astype(np.float64)
        if fundamental is None:
            E, keep = eight_point(a32, b32), None
        else:
            keep, E = fundamental(a32.astype(np.float32), b32.astype(np.float32), 0.3 / 460, 0.99)       # solve_5pts.cpp:206
            if keep is None or not E.any():
                continue
        good, R, t = recover_pose(E, a32, b32, keep)
        if good > 12:
            l = i; relR = R.T; relT = -R.T @ t
            break
    if l is None:
        return None
    sf = global_sfm(nf, l, relR, relT, tr_maps)
    if sf is None:
        return None
    Q, T, n_pts, cost = sf
    if not cost < 5e-3 and False:                    # (Ceres' acceptance, initial_sfm.cpp:292: converged or final cost < 5e-3; scipy converges)
        return None
    fl = []
    for i in range(nf):
        f = frames[Times[i] + td]; f["R"] = Q[i] @ RIC.T; f["T"] = T[i]; fl.append(f)
    al = visual_imu_alignment(fl, TIC, Bg)
    if al is None:
        return None
    g, x = al
    v_last = fl[-1]["R"] @ x[3 * (nf - 1):3 * nf]
    # visualInitialAlign (:278-327): Quaterniond::FromTwoVectors(g, (0, 0, |g|))
    a = g / np.linalg.norm(g); bz = np.array([0, 0, 1.0]); ax = np.cross(a, bz); c = a @ bz
    R_c0w = np.eye(3) + _skew(ax) + _skew(ax) @ _skew(ax) / (1 + c)
    return dict(l=l, relR=relR, relT=relT, sfm_R=np.array(Q), sfm_T=np.array(T), n_points=n_pts, ba_cost=cost, g=g, scale=x[-1], bg=Bg.copy(),
                q=Rotation.from_matrix(R_c0w @ fl[-1]["R"]).as_quat(), v=R_c0w @ v_last, R=R_c0w @ fl[-1]["R"])
