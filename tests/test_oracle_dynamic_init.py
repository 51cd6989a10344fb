"""Row N4 on the CPU: the product's moving-start initialiser (larvio_amd/csrc/be_init.h, compiled into a replay harness -
tests/host/init_replay.hip, host code only) against the independent numpy / scipy restatement oracle/dyn_init.py on recorded starts
(tests/feature_sim.py: no code under test produces the inputs), and both against ground truth.

The two share the reference's bookkeeping (DynamicInitializer.cpp etc., cited in both) and nothing else: the product's 3x3 Jacobi SVD,
Levenberg-Marquardt PnP and Schur-complement bundle adjustment stand against numpy.linalg.svd and scipy.optimize.least_squares on a
rotation-vector parametrisation.  They have to arrive at the same minimum: intermediate results (frame l, relative pose, the window's
structure-from-motion poses, gyro bias, gravity, metric scale) and the state handed to the filter agree to 1e-13 on noise-free input and
to 1e-8 with observation / IMU noise (what is left is where each minimiser stops).  findFundamentalMat's RANSAC is "keep everything" on
both sides in the "keep" cases; the "real" cases of every recorded start run cv::findFundamentalMat's restatement (oracle/liblvo.so, bit for
bit the product kernel's mask and matrix) on both sides, and the GPU suite drives lvk_ekf_process with the kernel itself and compares
its lvk_ekf_init_report with this restatement fed the kernel's own answers (tests/test_gpu_dynamic_init.py)."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from tests import feature_sim as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def replay(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("init") / "init_replay")
    src = os.path.join(ROOT, "tests", "host", "init_replay.hip")
    cxx = shutil.which("g++") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = ["-O2", "-std=c++17", "-ffp-contract=off", "-w", "-x", "c++"] if cxx.endswith("g++") else ["-O2", "-std=c++17", "-ffp-contract=off", "-w", "-x", "hip", "--offload-arch=gfx950"]
    subprocess.check_call([cxx] + flags + [src, "-o", exe, "-ldl"])
    return exe


def _record(path, sim, R_b2c, t_c_b):
    with open(path, "w") as f:
        f.write(" ".join("%.17g" % x for x in R_b2c.ravel()) + "\n" + " ".join("%.17g" % x for x in t_c_b) + "\n%.17g\n" % (1.0 / 400))
        imu = sim["imu"]; f.write("%d\n" % len(imu))
        for s in imu:
            f.write("%.17g %s %s\n" % (s["t"], " ".join("%.17g" % x for x in s["gyro"]), " ".join("%.17g" % x for x in s["acc"])))
        f.write("%d\n" % len(sim["msgs"]))
        for ts, m in sim["msgs"]:
            f.write("%.17g %d\n" % (ts, len(m)))
            for o in m:
                f.write("%d %.17g %.17g %.17g %.17g\n" % (o["id"], o["u"], o["v"], o["u_vel"], o["v_vel"]))


def _ang(A, B):
    return float(np.linalg.norm(Rotation.from_matrix(np.asarray(A).T @ np.asarray(B)).as_rotvec()))


def _sides(replay, rec, sim, R_b2c, t_c_b, ransac):
    """product (replay harness) and independent restatement on one recorded start; ransac = "keep" (every correspondence an inlier, 8-point
    fit: the stand-in) or "real" (cv::findFundamentalMat's restatement oracle/liblvo.so on both sides - mask and matrix bit for bit the
    product kernel's, tests/test_gpu_frontend_stages.py - so the 7-point model hand-off of solve_5pts.cpp:206 is exercised end to end)"""
    from oracle import dyn_init as D, lvo
    if ransac == "keep":
        P = json.loads(subprocess.run([replay, rec], capture_output=True, text=True, check=True, timeout=120).stdout)
        return P, D.dynamic_init(sim["msgs"], sim["imu"], R_b2c, t_c_b)
    lvo.lib()                                                                       # builds oracle/liblvo.so if need be
    so = os.path.join(ROOT, "oracle", "liblvo.so")
    P = json.loads(subprocess.run([replay, rec, so], capture_output=True, text=True, check=True, timeout=120).stdout)
    return P, D.dynamic_init(sim["msgs"], sim["imu"], R_b2c, t_c_b, fundamental=lambda a, b, th, cf: lvo.find_fundamental(a, b, th, cf))


@pytest.mark.parametrize("ransac", ["keep", "real"])
@pytest.mark.parametrize("seed,speed,sigma,imu_noise,tol", [(1, 2.0, 0.0, 0.0, 1e-9), (2, 4.0, 3e-4, 1.0, 1e-6), (3, 3.0, 6e-4, 2.0, 1e-6)])
def test_product_initialiser_against_the_independent_restatement_and_the_truth(replay, tmp_path, seed, speed, sigma, imu_noise, tol, ransac):
    from larvio_amd import synthetic as S
    tr = S.Trajectory(speed=speed)
    sim = F.simulate(seed, t0=3.5, t1=5.2, sigma=sigma, imu_noise=imu_noise, traj=tr, fresh_ids=True)
    T = np.asarray(S.EUROC["T_cam_imu"], float); R_b2c = T[:3, :3]; t_c_b = -R_b2c.T @ T[:3, 3]
    rec = str(tmp_path / "start.txt"); _record(rec, sim, R_b2c, t_c_b)
    P, O = _sides(replay, rec, sim, R_b2c, t_c_b, ransac)
    assert O is not None and P["message"] == O["message"] == 10                     # the 11th message, first attempt
    assert P["l"] == O["l"] and P["n_points"] == O["n_points"] and P["erase"] == O["erase"] and P["state_time"] == O["state_time"]
    # relative pose of (l, newest): same linear algebra, library SVD against Jacobi
    assert _ang(np.reshape(P["relR"], (3, 3)), O["relR"]) < 1e-9 and np.abs(np.array(P["relT"]) - O["relT"]).max() < 1e-9
    # the window's structure from motion: two different minimisers, one minimum (gauge: rotation of l, positions of l and the newest held)
    PR = np.reshape(P["sfm_R"], (-1, 3, 3)); PT = np.reshape(P["sfm_T"], (-1, 3))
    worst_R = max(_ang(PR[i], O["sfm_R"][i]) for i in range(11)); worst_T = float(np.abs(PT - O["sfm_T"]).max())
    d_bg = float(np.abs(np.array(P["bg"]) - O["bg"]).max()); d_g = float(np.abs(np.array(P["g"]) - O["g"]).max()); d_s = abs(P["scale"] / O["scale"] - 1)
    d_q = _ang(Rotation.from_quat(P["q"]).as_matrix(), O["R"]); d_v = float(np.abs(np.array(P["v"]) - O["v"]).max())
    print("seed %d sigma %.0e: sfm R %.1e T %.1e | bg %.1e g %.1e scale %.1e | attitude %.1e velocity %.1e (bundle cost %.2e)" % (seed, sigma, worst_R, worst_T, d_bg, d_g, d_s, d_q, d_v, O["ba_cost"]))
    assert worst_R < tol and worst_T < tol and d_bg < tol and d_g < 10 * tol and d_s < tol and d_q < tol and d_v < 10 * tol      # (g in m/s^2: 1e-4 of 9.81)
    # ... and that minimum is the truth up to the noise: gravity direction and velocity in the body frame (heading-free)
    ts = P["state_time"]; Re = Rotation.from_quat(P["q"]).as_matrix(); Rt = tr.R_wb(ts)
    e_up = float(np.abs(Re[2] - Rt[2]).max()); e_v = float(np.abs(Re.T @ np.array(P["v"]) - Rt.T @ tr.vel(ts)).max())
    bound_up, bound_v = (1e-5, 1e-4) if sigma == 0 else (4e-3 * sigma / 3e-4, 3e-2 * sigma / 3e-4)      # measured at 0.14 / 0.28 px: 0.6e-3 / 6.5e-3 and 2 / 54 mm/s
    assert e_up < bound_up and e_v < bound_v, (e_up, e_v)
    # metric scale: the structure-from-motion baseline (l, newest) against the true one
    base_true = np.linalg.norm(tr.cam_pose(sim["msgs"][10][0])[1] - tr.cam_pose(sim["msgs"][P["l"]][0])[1])
    assert abs(P["scale"] * np.linalg.norm(PT[10] - PT[P["l"]]) / base_true - 1) < (1e-4 if sigma == 0 else 3e-2 * sigma / 3e-4)      # metric scale from one second of motion: measured 0.2 % at 0.14 px, 4.6 % at 0.28 px and twice the IMU noise


@pytest.mark.parametrize("ransac", ["keep", "real"])
def test_window_slides_until_the_platform_moves(replay, tmp_path, ransac):
    """A start from rest seen by the DYNAMIC initialiser alone (in the product the static one would fire first): no parallax, so
    relativePose refuses and the window slides message after message (slideWindow / removeBack, DynamicInitializer.cpp:362-402,
    feature_manager.cpp:205-222); when the platform has moved enough the first attempt that gets through must be the same one on both
    sides, with the same results."""
    from larvio_amd import synthetic as S
    tr = S.Trajectory(speed=3.0)                                                    # rests until 1.2 s, then ramps up
    sim = F.simulate(5, t0=0.3, t1=3.6, sigma=3e-4, imu_noise=1.0, traj=tr, fresh_ids=True)
    T = np.asarray(S.EUROC["T_cam_imu"], float); R_b2c = T[:3, :3]; t_c_b = -R_b2c.T @ T[:3, 3]
    rec = str(tmp_path / "start.txt"); _record(rec, sim, R_b2c, t_c_b)
    P, O = _sides(replay, rec, sim, R_b2c, t_c_b, ransac)
    assert O is not None and P["message"] == O["message"] and P["message"] > 12, (P["message"], None if O is None else O["message"])
    assert P["l"] == O["l"] and P["n_points"] == O["n_points"] and P["erase"] == O["erase"] and P["state_time"] == O["state_time"]
    PT = np.reshape(P["sfm_T"], (-1, 3))
    d = dict(sfm_T=float(np.abs(PT - O["sfm_T"]).max()), bg=float(np.abs(np.array(P["bg"]) - O["bg"]).max()), scale=float(abs(P["scale"] / O["scale"] - 1)),
             attitude=_ang(Rotation.from_quat(P["q"]).as_matrix(), O["R"]), v=float(np.abs(np.array(P["v"]) - O["v"]).max()))
    print("product against oracle:", {k: "%.1e" % v for k, v in d.items()})
    # the window still holds frames from the rest: their poses sit in a flat valley of the bundle adjustment (no baseline), where the two
    # minimisers stop a few 1e-5 apart; what is handed to the filter agrees much better than that
    assert d["sfm_T"] < 2e-4 and d["bg"] < 1e-5 and d["scale"] < 2e-3 and d["attitude"] < 1e-4 and d["v"] < 2e-3
    ts = P["state_time"]; Re = Rotation.from_quat(P["q"]).as_matrix(); Rt = tr.R_wb(ts)
    print("first successful attempt at message %d (t = %.2f s), l = %d; gravity direction %.1e, body velocity %.3f m/s off the truth" %
          (P["message"], ts, P["l"], np.abs(Re[2] - Rt[2]).max(), np.abs(Re.T @ np.array(P["v"]) - Rt.T @ tr.vel(ts)).max()))
    assert np.abs(Re[2] - Rt[2]).max() < 2e-2 and np.abs(Re.T @ np.array(P["v"]) - Rt.T @ tr.vel(ts)).max() < 0.1


def test_with_the_real_ransac_stage_and_mismatched_tracks(replay, tmp_path):
    """The whole initialiser with cv::findFundamentalMat's restatement in the loop (oracle/liblvo.so on both sides - the harness loads it at
    run time; its mask and matrix are bit for bit the product kernel's) on a start whose tracks contain mismatches: 12 % of the
    observations of the newest window frame are displaced by 3-12 px.  The RANSAC mask keeps them out of the relative pose (they still
    enter the structure from motion, as in the reference); product and independent restatement must agree."""
    from larvio_amd import synthetic as S
    from oracle import dyn_init as D, lvo
    tr = S.Trajectory(speed=3.0)
    sim = F.simulate(7, t0=3.5, t1=5.2, sigma=3e-4, imu_noise=1.0, traj=tr, fresh_ids=True)
    rng = np.random.default_rng(11)
    ts10, m10 = sim["msgs"][10]; m10 = m10.copy()
    bad = rng.random(len(m10)) < 0.12
    ang = rng.uniform(0, 2 * np.pi, len(m10)); mag = rng.uniform(3, 12, len(m10)) / 460
    m10["u"] += np.where(bad, mag * np.cos(ang), 0); m10["v"] += np.where(bad, mag * np.sin(ang), 0)
    sim["msgs"][10] = (ts10, m10)
    T = np.asarray(S.EUROC["T_cam_imu"], float); R_b2c = T[:3, :3]; t_c_b = -R_b2c.T @ T[:3, 3]
    rec = str(tmp_path / "start.txt"); _record(rec, sim, R_b2c, t_c_b)
    lvo.lib()                                                                       # builds oracle/liblvo.so if need be
    so = os.path.join(ROOT, "oracle", "liblvo.so")
    P = json.loads(subprocess.run([replay, rec, so], capture_output=True, text=True, check=True, timeout=120).stdout)
    O = D.dynamic_init(sim["msgs"], sim["imu"], R_b2c, t_c_b, fundamental=lambda a, b, th, cf: lvo.find_fundamental(a, b, th, cf))
    assert O is not None and P["message"] == O["message"] == 10 and P["l"] == O["l"] and P["n_points"] == O["n_points"]
    assert _ang(np.reshape(P["relR"], (3, 3)), O["relR"]) < 1e-9 and np.abs(np.array(P["relT"]) - O["relT"]).max() < 1e-9
    PT = np.reshape(P["sfm_T"], (-1, 3))
    d = dict(sfm_T=float(np.abs(PT - O["sfm_T"]).max()), bg=float(np.abs(np.array(P["bg"]) - O["bg"]).max()), scale=float(abs(P["scale"] / O["scale"] - 1)),
             attitude=_ang(Rotation.from_quat(P["q"]).as_matrix(), O["R"]), v=float(np.abs(np.array(P["v"]) - O["v"]).max()))
    print("product against oracle with the RANSAC stage in the loop:", {k: "%.1e" % v for k, v in d.items()}, "| %d of %d observations displaced" % (bad.sum(), len(m10)))
    assert max(d.values()) < 1e-5
    # the relative pose came from inliers only: within a few degrees of the true baseline direction despite the mismatches
    l = P["l"]; c_l, c_n = tr.cam_pose(sim["msgs"][l][0]), tr.cam_pose(sim["msgs"][10][0])
    base = c_l[0].T @ (c_n[1] - c_l[1]); base /= np.linalg.norm(base)
    relT = np.array(P["relT"]); cosang = float(base @ relT / np.linalg.norm(relT))
    assert cosang > np.cos(np.radians(8)), np.degrees(np.arccos(cosang))


def test_preintegration_restatements_agree(replay):
    """the oracle's PreInt against a direct quadrature of the same constant-rate motion (closed form), so that the two sides of the test
    above do not merely share a formula"""
    from oracle import dyn_init as D
    w = np.array([0.3, -0.2, 0.5]); a = np.array([0.2, -0.1, 9.6])
    p = D.PreInt(a, w, np.zeros(3))
    for _ in range(400):
        p.push_back(0.0025, a, w)
    Rw = Rotation.from_rotvec(w * 1.0).as_matrix()
    assert np.linalg.norm(Rotation.from_matrix(Rw.T @ D._q2R(p.dq)).as_rotvec()) < 1e-6
    # delta_v = int_0^1 R(t) a dt with R(t) = exp([w] t): Rodrigues integrated in closed form
    th = np.linalg.norm(w); K = D._skew(w / th)
    I1 = np.eye(3) + (1 - np.cos(th)) / th * K + (th - np.sin(th)) / th * K @ K
    assert np.abs(p.dv - I1 @ a).max() < 1e-5
    # d(delta_q)/d(b_g) against a finite difference of repropagate()
    q0 = p.dq.copy(); J = p.J.copy()
    for k in range(3):
        bg = np.zeros(3); bg[k] = 1e-6
        p.repropagate(bg)
        d = D._qmul(np.array([-q0[0], -q0[1], -q0[2], q0[3]]), p.dq)
        assert np.abs(2 * d[:3] / 1e-6 - J[:, k]).max() < 2e-3                      # first order in w dt, as the reference's recursion
