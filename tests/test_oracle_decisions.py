"""Decision-trace pins for the host logic of the back-end.  The product (backend.hip) and the oracle (oracle/be_filter.c) spell the
pointer-chasing decisions of LarVio the same way, so a shared misreading of the reference could not be caught by comparing the two.
Here each decision is restated a THIRD time, independently and directly from /root/reference/src/larvio.cpp (cited per function,
written against the reference's text, not against be_filter.c), and replayed on a trace the oracle records while it runs
(LVO_TRACE=<file>: the inputs each decision saw and what it decided):

  findRedundantImuStates   larvio.cpp:2259-2307   which two clones leave the window
  getNewAnchorId           larvio.cpp:3412-3472   the new anchor of a feature whose anchor clone is removed
  removeLostFeatures       larvio.cpp:1915-2005   triage of the features not in the state (invalid / MSCKF / new EKF / untouched)
  updateGridMap            larvio.cpp:3351-3370   occupancy of the augmentation grid

A disagreement means the oracle (and with it, most likely, the product) misreads the reference."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def quat_xyzw_to_R(q):
    """Eigen::Quaterniond(w, x, y, z).toRotationMatrix() for the reference's [x y z w] storage (larvio.cpp:2273-2274)"""
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def angle_axis_angle(R):
    """Eigen::AngleAxisd(R).angle(): through the quaternion of R, angle = 2 atan2(|vec|, |w|) in [0, pi]"""
    c = (np.trace(R) - 1.0) / 2.0
    s = 0.5 * np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return float(np.arctan2(s, c))                        # same angle as the quaternion form for a proper rotation


def find_redundant(clones, tracking_rate, rot_th, trans_th, rate_th):
    """larvio.cpp:2259-2307.  clones: list of (id, q_cam, p_cam) in window order."""
    n = len(clones)
    key = n - 4                                           # :2263-2265: four steps back from end()
    it = key + 1                                          # :2266-2267
    first = 0                                             # :2268
    key_R = quat_xyzw_to_R(clones[key][1]); key_p = clones[key][2]
    rm = []
    for _ in range(2):                                    # :2279
        cid, q, p = clones[it]
        Rt = quat_xyzw_to_R(q).T                          # :2283-2284 (.transpose())
        distance = np.linalg.norm(p - key_p)              # :2286
        angle = angle_axis_angle(Rt @ key_R)              # :2287-2288
        if angle < rot_th and distance < trans_th and tracking_rate > rate_th:   # :2290-2292
            rm.append(cid); it += 1                       # :2293-2294
        else:
            rm.append(clones[first][0]); first += 1       # :2296-2297
            it -= 2                                       # :2298-2299
    return sorted(rm)                                     # :2304


def new_anchor_id(p_w, rm_ids, clones):
    """larvio.cpp:3412-3472.  clones: list of (id, q_cam, p_cam, observed, z)."""
    size = len(clones)
    if size <= 2:                                         # :3425-3429
        return clones[-1][0]
    best, best_dis = None, 99999.0                        # :3434-3436
    for cid, q, p, observed, z in clones[:size - 2]:      # :3437 looplen = size - 2, from the oldest clone
        if not observed:                                  # :3438-3441
            continue
        if cid in rm_ids:                                 # :3442-3443
            continue
        p_new = quat_xyzw_to_R(q).T @ (p_w - p)           # :3445-3449 (R_c2w.inverse())
        dis = np.hypot(p_new[0] / p_new[2] - z[0], p_new[1] / p_new[2] - z[1])   # :3450-3452
        if best_dis > dis:                                # :3453-3457
            best_dis, best = dis, cid
    return best if best is not None else clones[-1][0]    # :3464-3471


def grid_counts(rows, cols, x_min, y_min, gw, gh, feats):
    """larvio.cpp:3351-3370: one entry per in-state feature at its current observation"""
    cnt = [0] * (rows * cols)
    if rows * cols == 0:
        return cnt
    for x, y in feats:
        code = int((y - y_min) / gh) * cols + int((x - x_min) / gw)      # static_cast<int> truncates towards zero, as int() does
        if 0 <= code < rows * cols:
            cnt[code] += 1
    return cnt


def triage(header, feats, grid, geom):
    """larvio.cpp:1926-2005 for the features NOT in the state, in map order.  feats: dicts with the recorded inputs; returns
    (category, is_initialized_after) per feature and consumes motion / triangulation outcomes exactly where the reference
    evaluates checkMotion / initializePosition / initializeInvParamPosition (-1 in the trace = never evaluated there)."""
    least_obs, max_track_len, max_features, cells, n_feature_states, since_zupt = header
    rows, cols, x_min, y_min, gw, gh = geom
    grid = list(grid)
    n_new = 0
    out = []
    for f in feats:
        init = bool(f["init0"])

        def need(key):
            assert f[key] != -1, (f["id"], key, "the reference evaluates this, the trace did not")
            return bool(f[key])
        used = set()
        if not f["tracked"]:                                                        # :1934
            if f["n_obs"] < least_obs:                                              # :1935-1938
                out.append((1, init, used)); continue
            if not init:                                                            # :1941
                used.add("motion")
                if not need("motion"):                                              # :1942-1944
                    out.append((1, init, used)); continue
                used.add("tri")
                if not need("tri"):                                                 # :1946-1949
                    out.append((1, init, used)); continue
                init = True
            out.append((2, init, used)); continue                                   # :1953-1955
        if not (f["n_obs"] >= max_track_len):                                       # :1958-1963
            out.append((0, init, used)); continue
        code = int((f["y"] - y_min) / gh) * cols + int((f["x"] - x_min) / gw)       # :1966-1970
        occupied = grid[code] if 0 <= code < cells else 0
        if occupied < max_features and since_zupt > 5 and (n_feature_states + n_new) < max_features * cells:   # :1971-1972
            if not f["ekf0"]:                                                       # :1974
                init = False                                                        # :1975
                used.add("motion")
                if need("motion"):                                                  # :1976
                    used.add("tri")
                    init = need("tri")                                              # :1977 (initializeInvParamPosition sets is_initialized on success)
            if not init:                                                            # :1980-1981
                out.append((4, init, used)); continue
            n_new += 1                                                              # :1986
            if 0 <= code < cells:
                grid[code] += 1                                                     # :1987
            out.append((3, init, used)); continue
        if not init:                                                                # :1990
            used.add("motion")
            if need("motion"):                                                      # :1991
                used.add("tri")
                init = need("tri")                                                  # :1992
        if not init:                                                                # :1994-1995
            out.append((4, init, used)); continue
        out.append((2, init, used))                                                 # :1996-1998
    return out


def _record_trace(tmp_path, seed, **kw):
    """run the oracle over simulated feature messages in a subprocess with LVO_TRACE set (the file is opened at create time)"""
    path = str(tmp_path / ("trace_%d.txt" % seed))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from oracle import lvo_be\nfrom tests import feature_sim as F\n"
            "sim = F.simulate(%d, **%r)\nora = lvo_be.Ekf(sim['cfg'])\nn = F.drive(ora, sim)\nprint(n, ora.counters())\ndel ora\n" % (ROOT, seed, kw))
    env = dict(os.environ, LVO_TRACE=path)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr
    return open(path).read().splitlines(), r.stdout


@pytest.mark.parametrize("seed,kw", [(4, dict(t1=9.0, sw_size=20, max_feat=150)),
                                     (5, dict(t1=8.0, sigma=2e-3, imu_noise=30.0, sw_size=12, max_feat=60)),
                                     (12, dict(t1=7.0, sw_size=16, max_feat=300, n_per_batch=200, max_features_in_one_grid=2, max_track_len=9))])
def test_decisions_replayed_from_the_reference_text(tmp_path, seed, kw):
    lines, out = _record_trace(tmp_path, seed, **kw)
    n = dict(red=0, anchor=0, grid=0, tri=0, feats=0, new=0, msckf=0, invalid=0, failed=0)
    grid_feats = []; last_grid = None; geom = None
    tri_header = None; tri_feats = []
    for ln in lines:
        w = ln.split()
        if w[0] == "REDUNDANT":
            nc = int(w[1]); rate, rot_th, trans_th, rate_th = map(float, w[2:6])
            v = w[6:6 + 8 * nc]
            clones = [(int(v[8 * i]), np.array(v[8 * i + 1:8 * i + 5], float), np.array(v[8 * i + 5:8 * i + 8], float)) for i in range(nc)]
            got = [int(w[-2]), int(w[-1])]
            assert find_redundant(clones, rate, rot_th, trans_th, rate_th) == got, ln[:200]
            n["red"] += 1
        elif w[0] == "ANCHOR":
            p_w = np.array(w[2:5], float); nrm = int(w[5]); rm = [int(x) for x in w[6:6 + nrm]]
            size = int(w[6 + nrm]); v = w[7 + nrm:7 + nrm + 11 * size]
            clones = [(int(v[11 * i]), np.array(v[11 * i + 1:11 * i + 5], float), np.array(v[11 * i + 5:11 * i + 8], float), int(v[11 * i + 8]) == 1,
                       np.array(v[11 * i + 9:11 * i + 11], float)) for i in range(size)]
            assert new_anchor_id(p_w, rm, clones) == int(w[-1]), ln[:200]
            n["anchor"] += 1
        elif w[0] == "GRIDF":
            grid_feats.append((float(w[2]), float(w[3])))
        elif w[0] == "GRID":
            rows, cols = int(w[1]), int(w[2]); geom = (rows, cols) + tuple(map(float, w[3:7]))
            last_grid = [int(x) for x in w[7:]]
            assert grid_counts(*geom, grid_feats) == last_grid
            grid_feats = []; n["grid"] += 1
        elif w[0] == "TRIAGE":
            tri_header = (int(w[1]), int(w[2]), int(w[3]), int(w[4]), int(w[5]), float(w[6])); tri_feats = []
        elif w[0] == "TRI":
            tri_feats.append(dict(id=int(w[1]), tracked=int(w[2]), n_obs=int(w[3]), init0=int(w[4]), ekf0=int(w[5]), x=float(w[6]), y=float(w[7]),
                                  motion=int(w[8]), tri=int(w[9]), cat=int(w[10]), init1=int(w[11])))
        elif w[0] == "TRIEND":
            res = triage(tri_header, tri_feats, last_grid if last_grid is not None else [0] * max(tri_header[3], 1), geom or (0, 0, 0, 0, 1, 1))
            for f, (cat, init, used) in zip(tri_feats, res):
                assert cat == f["cat"] and int(init) == f["init1"], (f, cat, init)
                # the trace must not hold an outcome the reference would never have computed at that point
                assert (f["motion"] != -1) == ("motion" in used) and (f["tri"] != -1) == ("tri" in used), (f, used)
                n["feats"] += 1; n["new"] += cat == 3; n["msckf"] += cat == 2; n["invalid"] += cat == 1; n["failed"] += cat == 4
            n["tri"] += 1
    print(seed, n, out.strip())
    assert n["tri"] >= 40 and n["grid"] >= 40 and n["feats"] > 2000 and n["msckf"] > 100 and n["new"] >= 5
    if seed != 5:
        assert n["red"] >= 10 and n["anchor"] >= 3          # the window filled and in-state features had to be re-anchored
    if seed == 5:
        assert n["invalid"] >= 3
