"""LarVio — host-side mirror of larvio::LarVio over the C ABI (include/lvk_c.h, back-end section).

Same surface as the reference class (include/larvio/larvio.h:37-90): construct with the configuration,
``initialize()``, then ``processFeatures(msg, imu_buffer)`` per feature message; getters for pose / velocity /
covariance / sliding window / map points.  The arithmetic runs in liblvk_hip.so on the GPU; like the reference,
``processFeatures`` erases the IMU samples it consumed from the caller's buffer (larvio.cpp:511-512).
"""
import ctypes as C
import os
import numpy as np
from ._lib import lib, _p, Context, LvkError, IMU, OBS

CLONE = np.dtype([("id", np.int64), ("time", np.float64), ("dt", np.float64), ("q", np.float64, 4), ("p", np.float64, 3),
                  ("p_fej", np.float64, 3), ("R_b2c", np.float64, 9), ("t_c_b", np.float64, 3), ("q_cam", np.float64, 4),
                  ("p_cam", np.float64, 3)])

_CFG_INT = ["if_fej", "estimate_extrin", "estimate_td", "if_zupt_valid", "sw_size", "max_track_len", "least_observation_number",
            "max_features_in_one_grid", "aug_grid_rows", "aug_grid_cols", "width", "height"]
_CFG_DBL = ["td", "pub_frequency", "imu_rate", "noise_gyro", "noise_acc", "noise_gyro_bias", "noise_acc_bias", "noise_feature",
            "initial_covariance_orientation", "initial_covariance_velocity", "initial_covariance_position",
            "initial_covariance_gyro_bias", "initial_covariance_acc_bias", "initial_covariance_extrin_rot",
            "initial_covariance_extrin_trans", "rotation_threshold", "translation_threshold", "tracking_rate_threshold",
            "feature_translation_threshold", "zupt_max_feature_dis", "zupt_noise_v", "zupt_noise_p", "zupt_noise_q", "static_duration"]


class EkfConfig(C.Structure):
    _fields_ = ([(k, C.c_int) for k in _CFG_INT] + [("intrinsics", C.c_double * 4), ("T_cam_imu", C.c_double * 16)] +
                [(k, C.c_double) for k in _CFG_DBL] +
                [("feature_idp_dim", C.c_int), ("use_schmidt", C.c_int), ("calib_imu_instrinsic", C.c_int), ("max_features", C.c_int),
                 ("legacy_grid", C.c_int), ("reserved0", C.c_int)])


_sig_done = False


class InitReport(C.Structure):
    """lvk_init_report (include/lvk_c.h)"""
    _fields_ = [("valid", C.c_int), ("message", C.c_int), ("attempts", C.c_int), ("ransac_calls", C.c_int), ("l", C.c_int), ("n_points", C.c_int), ("erase", C.c_int), ("pad", C.c_int),
                ("state_time", C.c_double), ("scale", C.c_double), ("rel_R", C.c_double * 9), ("rel_T", C.c_double * 3), ("sfm_R", C.c_double * 99), ("sfm_T", C.c_double * 33),
                ("bg", C.c_double * 3), ("g", C.c_double * 3), ("q", C.c_double * 4), ("v", C.c_double * 3)]


def _L():
    global _sig_done
    L = lib()
    if not _sig_done:
        vp, i, d = C.c_void_p, C.c_int, C.c_double
        pi = C.POINTER(C.c_int)
        L.lvk_ekf_compress_qr.argtypes = [vp, vp, i, i, i, vp, pi]; L.lvk_ekf_compress_qr.restype = i
        L.lvk_ekf_update.argtypes = [vp, vp, i, i, vp, i, i, vp, d, vp]; L.lvk_ekf_update.restype = i
        L.lvk_ekf_compress_qr_groups.argtypes = [vp, vp, i, i, i, vp, i, vp, vp, vp, pi]; L.lvk_ekf_compress_qr_groups.restype = i
        L.lvk_ekf_qr_plan.argtypes = [i, i, vp, vp, vp, vp, i, vp, i, vp, vp, i, pi]; L.lvk_ekf_qr_plan.restype = i
        L.lvk_dgemm.argtypes = [vp, i, i, i, i, i, d, vp, i, vp, i, d, vp, i]; L.lvk_dgemm.restype = i
        L.lvk_ekf_create.argtypes = [vp, C.POINTER(EkfConfig), C.POINTER(vp)]; L.lvk_ekf_create.restype = i
        L.lvk_ekf_destroy.argtypes = [vp]; L.lvk_ekf_destroy.restype = None
        L.lvk_ekf_process.argtypes = [vp, d, vp, i, vp, i, pi, pi]; L.lvk_ekf_process.restype = i
        L.lvk_ekf_set_state.argtypes = [vp, d, vp, vp, vp, vp, vp, vp, vp]; L.lvk_ekf_set_state.restype = i
        L.lvk_ekf_dim.argtypes = [vp]; L.lvk_ekf_dim.restype = i
        L.lvk_ekf_is_initialized.argtypes = [vp]; L.lvk_ekf_is_initialized.restype = i
        L.lvk_ekf_take_off_stamp.argtypes = [vp]; L.lvk_ekf_take_off_stamp.restype = C.c_double
        L.lvk_ekf_take_lost_features.argtypes = [vp, vp, vp, i]; L.lvk_ekf_take_lost_features.restype = i
        L.lvk_ekf_get_state.argtypes = [vp, vp]; L.lvk_ekf_get_state.restype = i
        L.lvk_ekf_get_cov.argtypes = [vp, vp]; L.lvk_ekf_get_cov.restype = i
        L.lvk_ekf_get_cov_imu.argtypes = [vp, i, vp]; L.lvk_ekf_get_cov_imu.restype = i
        L.lvk_ekf_get_imu_intrinsics.argtypes = [vp, vp]; L.lvk_ekf_get_imu_intrinsics.restype = i
        L.lvk_ekf_set_imu_intrinsics.argtypes = [vp, vp]; L.lvk_ekf_set_imu_intrinsics.restype = i
        L.lvk_ekf_get_clones.argtypes = [vp, vp, i]; L.lvk_ekf_get_clones.restype = i
        L.lvk_ekf_get_features.argtypes = [vp, vp, vp, vp, i]; L.lvk_ekf_get_features.restype = i
        L.lvk_ekf_counters.argtypes = [vp, vp]; L.lvk_ekf_counters.restype = None
        L.lvk_ekf_init_report.argtypes = [vp, vp]; L.lvk_ekf_init_report.restype = C.c_int
        L.lvk_ekf_profile.argtypes = [vp, i, vp]; L.lvk_ekf_profile.restype = i
        L.lvk_ekf_set_shard.argtypes = [vp, i, i, vp, vp]; L.lvk_ekf_set_shard.restype = i
        L.lvk_ekf_shard_stats.argtypes = [vp, vp]; L.lvk_ekf_shard_stats.restype = None
        L.lvk_triangulate.argtypes = [vp, vp, vp, i, i, vp, pi, vp, vp, vp, vp]; L.lvk_triangulate.restype = i
        L.lvk_ekf_gate_and_stack.argtypes = [vp, vp, i, vp, i, vp, vp, vp, vp, i, i, i, d, vp, vp, i, pi, vp, vp]; L.lvk_ekf_gate_and_stack.restype = i
        _sig_done = True
    return L


# ---------------------------------------------------------------- stage-level wrappers (numpy in / numpy out)
POSE = np.dtype([("R", np.float64, 9), ("t", np.float64, 3)])
MSCKF_FEATURE = np.dtype([("p_w", np.float64, 3), ("n_obs", np.int32), ("obs_off", np.int32)])


def triangulate(ctx, poses, obs, use_position=False, position_in=None):
    """Feature::initializePosition for one feature (feature.hpp:383-552).  Returns (ok, position, solution, inv_depth, obs_anchor)."""
    poses = np.ascontiguousarray(poses, POSE); obs = np.ascontiguousarray(obs, np.float64)
    pin = np.ascontiguousarray(position_in if position_in is not None else np.zeros(3), np.float64)
    ok = C.c_int(0); pos = np.zeros(3); sol = np.zeros(3); idp = np.zeros(1); oa = np.zeros(3)
    ctx.check(_L().lvk_triangulate(ctx.h, _p(poses), _p(obs), len(poses), int(use_position), _p(pin), C.byref(ok), _p(pos), _p(sol), _p(idp), _p(oa)))
    return bool(ok.value), pos, sol, float(idp[0]), oa


def gate_and_stack(ctx, clones, feats, clone_rank, obs, obs_vel, P, sigma2, if_fej=1, estimate_td=1):
    """featureJacobian_msckf + gatingTest + stacking for a batch of MSCKF features.  feats: list of (p_w, n_obs, obs_off).
    Returns (H (rows x N), r, gamma[n_feats], accept[n_feats])."""
    clones = np.ascontiguousarray(clones, CLONE)
    fa = np.zeros(len(feats), MSCKF_FEATURE)
    for k, (pw, n, off) in enumerate(feats):
        fa[k]["p_w"] = pw; fa[k]["n_obs"] = n; fa[k]["obs_off"] = off
    cr = np.ascontiguousarray(clone_rank, np.int32); obs = np.ascontiguousarray(obs, np.float64); ov = np.ascontiguousarray(obs_vel, np.float64)
    P = np.ascontiguousarray(P, np.float64); N = P.shape[0]
    cap = int(sum(2 * n - 3 for _, n, _ in feats)) + 1
    H = np.zeros((cap, N)); r = np.zeros(cap); rows = C.c_int(0); gamma = np.zeros(len(feats)); acc = np.zeros(len(feats), np.int32)
    ctx.check(_L().lvk_ekf_gate_and_stack(ctx.h, _p(clones), len(clones), _p(fa), len(fa), _p(cr), _p(obs), _p(ov), _p(P), N, int(if_fej), int(estimate_td),
                                          float(sigma2), _p(H), _p(r), cap, C.byref(rows), _p(gamma), _p(acc)))
    return H[:rows.value].copy(), r[:rows.value].copy(), gamma, acc.astype(bool)


def dgemm(ctx, A, B, transa=False, transb=False, alpha=1.0, beta=0.0, Cin=None):
    A = np.ascontiguousarray(A, np.float64); B = np.ascontiguousarray(B, np.float64)
    M = A.shape[1] if transa else A.shape[0]; K = A.shape[0] if transa else A.shape[1]; N = B.shape[0] if transb else B.shape[1]
    Cm = np.zeros((M, N)) if Cin is None else np.array(Cin, np.float64, order="C")
    dA, dB, dC = ctx.to_device(A), ctx.to_device(B), ctx.to_device(Cm)
    ctx.check(_L().lvk_dgemm(ctx.h, int(transa), int(transb), M, N, K, alpha, _p(dA), A.shape[1], _p(dB), B.shape[1], beta, _p(dC), N))
    return ctx.to_host(dC, np.float64, (M, N))


def ekf_update(ctx, P, H, r, sigma2):
    P = np.array(P, np.float64, order="C"); H = np.ascontiguousarray(H, np.float64); r = np.ascontiguousarray(r, np.float64)
    n, m = P.shape[0], H.shape[0]
    dP, dH, dr, ddx = ctx.to_device(P), ctx.to_device(H), ctx.to_device(r), ctx.alloc(8 * n)
    ctx.check(_L().lvk_ekf_update(ctx.h, _p(dP), n, n, _p(dH), H.shape[1], m, _p(dr), sigma2, _p(ddx)))
    return ctx.to_host(ddx, np.float64, (n,)), ctx.to_host(dP, np.float64, (n, n))


def compress_qr(ctx, H, r):
    H = np.array(H, np.float64, order="C"); r = np.array(r, np.float64)
    rows, cols = H.shape
    dH, dr = ctx.to_device(H), ctx.to_device(r)
    out = C.c_int(0)
    ctx.check(_L().lvk_ekf_compress_qr(ctx.h, _p(dH), cols, rows, cols, _p(dr), C.byref(out)))
    k = out.value
    return ctx.to_host(dH, np.float64, (rows, cols))[:k].copy(), ctx.to_host(dr, np.float64, (rows,))[:k].copy()


def _group_arrays(groups):
    rows = np.ascontiguousarray([g[0] for g in groups], np.int32)
    off = np.zeros(len(groups) + 1, np.int32)
    for k, g in enumerate(groups):
        off[k + 1] = off[k] + len(g[1])
    cols = np.ascontiguousarray(np.concatenate([np.asarray(g[1], np.int32) for g in groups]) if groups else np.zeros(0, np.int32), np.int32)
    return rows, off, cols


def qr_plan(N, groups):
    """lvk_ekf_qr_plan (host only): groups = [(rows, ascending column list), ...] in stacking order ->
    (levels, final_rows), levels = [dict(blocks=[dict(in_start, in_rows, out_start, out_rows, ncols, col_off, copy)], cols=int array)]"""
    rows, off, cols = _group_arrays(groups)
    cap_b, cap_c, cap_l = 4 * len(groups) + 16, 8 * (len(cols) + 16), 16
    blocks = np.zeros((cap_b, 8), np.int32); bcols = np.zeros(cap_c, np.int32); lb = np.zeros(cap_l, np.int32); lc = np.zeros(cap_l, np.int32)
    fin = C.c_int(0)
    n = _L().lvk_ekf_qr_plan(int(N), len(groups), _p(rows), _p(off), _p(cols), _p(blocks), cap_b, _p(bcols), cap_c, _p(lb), _p(lc), cap_l, C.byref(fin))
    if n < 0:
        raise LvkError("lvk_ekf_qr_plan failed")
    levels = []; ob = oc = 0
    keys = ("in_start", "in_rows", "out_start", "out_rows", "ncols", "col_off", "copy")
    for l in range(n):
        levels.append(dict(blocks=[dict(zip(keys, map(int, b[:7]))) for b in blocks[ob:ob + lb[l]]], cols=bcols[oc:oc + lc[l]].copy()))
        ob += lb[l]; oc += lc[l]
    return levels, fin.value


def compress_qr_groups(ctx, H, r, groups):
    """lvk_ekf_compress_qr_groups on host arrays: H (rows x cols), r, groups = [(rows, ascending column list), ...] -> (H', r')"""
    H = np.ascontiguousarray(H, np.float64); r = np.ascontiguousarray(r, np.float64)
    rows, cols = H.shape
    gr, off, gc = _group_arrays(groups)
    dH = ctx.to_device(H); dr = ctx.to_device(r)
    out = C.c_int(0)
    ctx.check(_L().lvk_ekf_compress_qr_groups(ctx.h, _p(dH), cols, rows, cols, _p(dr), len(groups), _p(gr), _p(off), _p(gc), C.byref(out)))
    k = out.value
    return ctx.to_host(dH, np.float64, (rows, cols))[:k].copy(), ctx.to_host(dr, np.float64, (rows,))[:k].copy()


def make_ekf_config(config):
    """lvk_ekf_config from the dict of LarVio parameters (larvio_amd.synthetic.backend_config)"""
    c = EkfConfig()
    for k in _CFG_INT + _CFG_DBL:
        setattr(c, k, config[k])
    c.intrinsics = (C.c_double * 4)(*config["intrinsics"])
    c.T_cam_imu = (C.c_double * 16)(*np.asarray(config["T_cam_imu"], np.float64).reshape(16))
    c.feature_idp_dim = config.get("feature_idp_dim", 1); c.use_schmidt = config.get("use_schmidt", 0)
    c.calib_imu_instrinsic = config.get("calib_imu_instrinsic", 0); c.max_features = config.get("max_features", 0)
    c.legacy_grid = config.get("legacy_grid", 0)          # 0 = the reference's grid_map bookkeeping (lvk_c.h)
    return c


class LarVio:
    def __init__(self, config, ctx=None):
        """config: dict with the keys LarVio::loadParameters reads (larvio.cpp:58-311; see synthetic.backend_config), or the path
        of a LARVIO configuration file (the reference's constructor argument)."""
        if isinstance(config, (str, os.PathLike)):
            from .config import load_config
            config = load_config(config)[1]
        self.config = dict(config)
        self.ctx = ctx
        self._h = None

    def initialize(self):
        if self.ctx is None:
            self.ctx = Context()
        c = make_ekf_config(self.config)
        h = C.c_void_p()
        st = _L().lvk_ekf_create(self.ctx.h, C.byref(c), C.byref(h))
        if st != 0:
            print("lvk_ekf_create failed:", lib().lvk_last_error(self.ctx.h).decode())
            return False
        self._h = h
        self.ctx.adopt(self)
        return True

    def processFeatures(self, msg, imu_msg_buffer):
        """msg: MonoCameraMeasurement (or (ts, features)); imu_msg_buffer: structured IMU array.
        Returns (bool, remaining_imu_buffer) — the reference mutates the caller's vector in place."""
        ts, feats = (msg.timeStampToSec, msg.features) if hasattr(msg, "features") else msg
        feats = np.ascontiguousarray(feats, OBS); imu = np.ascontiguousarray(imu_msg_buffer, IMU)
        used, upd = C.c_int(0), C.c_int(0)
        self.ctx.check(_L().lvk_ekf_process(self._h, float(ts), _p(feats), len(feats), _p(imu), len(imu), C.byref(used), C.byref(upd)))
        return bool(upd.value), imu[used.value:]

    def processFeaturesAsync(self, msg, imu_msg_buffer):
        """lvk_ekf_process_async: same arguments and return value as processFeatures, but the update itself runs on the filter's
        worker thread and stream; any getter (or the next update, or wait()) waits for it."""
        ts, feats = (msg.timeStampToSec, msg.features) if hasattr(msg, "features") else msg
        feats = np.ascontiguousarray(feats, OBS); imu = np.ascontiguousarray(imu_msg_buffer, IMU)
        used, upd = C.c_int(0), C.c_int(0)
        L = _L()
        L.lvk_ekf_process_async.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]; L.lvk_ekf_process_async.restype = C.c_int
        self.ctx.check(L.lvk_ekf_process_async(self._h, float(ts), _p(feats), len(feats), _p(imu), len(imu), C.byref(used), C.byref(upd)))
        return bool(upd.value), imu[used.value:]

    def wait(self):
        """lvk_ekf_wait: block until the deferred update is done; raises if it failed; -> whether it updated"""
        L = _L()
        L.lvk_ekf_wait.argtypes = [C.c_void_p, C.c_void_p]; L.lvk_ekf_wait.restype = C.c_int
        upd = C.c_int(0)
        self.ctx.check(L.lvk_ekf_wait(self._h, C.byref(upd)))
        return bool(upd.value)

    def set_state(self, t, q, p, v, bg, ba, gyro_old, acc_old):
        a = [np.ascontiguousarray(x, np.float64) for x in (q, p, v, bg, ba, gyro_old, acc_old)]
        self.ctx.check(_L().lvk_ekf_set_state(self._h, float(t), *[_p(x) for x in a]))

    @property
    def dim(self):
        return _L().lvk_ekf_dim(self._h)

    @property
    def initialized(self):
        return bool(_L().lvk_ekf_is_initialized(self._h))

    @property
    def take_off_stamp(self):
        """larvio.cpp:380 — state time at which the initializer succeeded"""
        return float(_L().lvk_ekf_take_off_stamp(self._h))

    def state(self):
        o = np.zeros(30); self.ctx.check(_L().lvk_ekf_get_state(self._h, _p(o)))
        return dict(t=o[0], q=o[1:5].copy(), v=o[5:8].copy(), p=o[8:11].copy(), bg=o[11:14].copy(), ba=o[14:17].copy(),
                    R_b2c=o[17:26].reshape(3, 3).copy(), t_c_b=o[26:29].copy(), td=o[29])

    def stable_map_points(self):
        """getStableMapPointPositions (larvio.cpp:2717-2722): (ids, positions) of in-state features lost since the last call"""
        ids = np.zeros(4096, np.int64); pos = np.zeros((4096, 3))
        n = _L().lvk_ekf_take_lost_features(self._h, _p(ids), _p(pos), 4096)
        return ids[:n].copy(), pos[:n].copy()

    def set_shard(self, rank, world, fn, user, keepalive=None):
        """lvk_ekf_set_shard: this filter does the per-feature device work of rank `rank` of `world`; fn/user = the all-gather
        (larvio_amd.sharding.RcclShard(...).args() or HostExchange(...).args())."""
        self._shard_keep = keepalive
        self.ctx.check(_L().lvk_ekf_set_shard(self._h, int(rank), int(world), fn, user))

    def shard_stats(self):
        o = np.zeros(8, np.int64); _L().lvk_ekf_shard_stats(self._h, _p(o))
        return dict(exchanges=int(o[0]), bytes_sent=int(o[1]), sharded_updates=int(o[2]), rows_stacked=int(o[3]),
                    qr_updates=int(o[4]), qr_levels=int(o[5]), qr_rows_in=int(o[6]), qr_rows_out=int(o[7]))

    def profile(self, enable=True):
        """HIP-event time of the H P GEMM since the last call: dict(ms, flops, launches); enables/disables the bracket"""
        o = np.zeros(3); self.ctx.check(_L().lvk_ekf_profile(self._h, int(enable), _p(o)))
        return dict(ms=o[0], flops=o[1], launches=int(o[2]))

    def profile_qr(self):
        """HIP-event time of the structure-aware TSQR levels (k_qr_sparse) since the last call, while profile(True) is on:
        dict(ms, flops on the structure factored, launches, rows entering the levels)"""
        L = _L(); L.lvk_ekf_profile_qr.argtypes = [C.c_void_p, C.c_void_p]; L.lvk_ekf_profile_qr.restype = C.c_int
        o = np.zeros(4); self.ctx.check(L.lvk_ekf_profile_qr(self._h, _p(o)))
        return dict(ms=o[0], flops=o[1], launches=int(o[2]), rows=o[3])

    def imu_intrinsics(self):
        """T1 T2 T3 A1 A2 A3 M1 M2 (24 numbers; state columns 22..45 when calib_imu_instrinsic = 1)"""
        o = np.zeros(24); self.ctx.check(_L().lvk_ekf_get_imu_intrinsics(self._h, _p(o))); return o

    def set_imu_intrinsics(self, v):
        v = np.ascontiguousarray(v, np.float64); assert v.shape == (24,)
        self.ctx.check(_L().lvk_ekf_set_imu_intrinsics(self._h, _p(v)))

    def cov(self):
        N = self.dim; P = np.zeros((N, N)); self.ctx.check(_L().lvk_ekf_get_cov(self._h, _p(P))); return P

    def cov_imu(self, n=9):
        """the covariance's leading n x n block (n <= 16): what getPpose / getPvel read, served without moving the matrix"""
        P = np.zeros((n, n)); self.ctx.check(_L().lvk_ekf_get_cov_imu(self._h, n, _p(P))); return P

    def clones(self):
        o = np.zeros(256, CLONE); n = _L().lvk_ekf_get_clones(self._h, _p(o), 256); return o[:n].copy()

    def features(self):
        ids = np.zeros(4096, np.int64); idp = np.zeros(4096); pos = np.zeros((4096, 3))
        n = _L().lvk_ekf_get_features(self._h, _p(ids), _p(idp), _p(pos), 4096)
        return ids[:n].copy(), idp[:n].copy(), pos[:n].copy()

    def counters(self):
        o = np.zeros(8, np.int64); _L().lvk_ekf_counters(self._h, _p(o))
        return dict(hybrid=int(o[0]), msckf=int(o[1]), last_rows=int(o[2]), zupt=int(o[3]), gated_in=int(o[4]), gated_out=int(o[5]),
                    map=int(o[6]), triangulations=int(o[7]))

    def init_report(self):
        """lvk_ekf_init_report: what the moving-start initialiser handed to the filter + the successful attempt's intermediate results
        (None until it has succeeded on this handle)"""
        r = InitReport(); rc = _L().lvk_ekf_init_report(self._h, C.byref(r))
        if rc != 0:
            raise LvkError("lvk_ekf_init_report failed (%d)" % rc)
        if not r.valid:
            return None
        a = lambda x, shape=None: np.array(x, np.float64).reshape(shape) if shape else np.array(x, np.float64)
        return dict(message=r.message, attempts=r.attempts, ransac_calls=r.ransac_calls, l=r.l, n_points=r.n_points, erase=r.erase, state_time=r.state_time, scale=r.scale,
                    relR=a(r.rel_R, (3, 3)), relT=a(r.rel_T), sfm_R=a(r.sfm_R, (11, 3, 3)), sfm_T=a(r.sfm_T, (11, 3)), bg=a(r.bg), g=a(r.g), q=a(r.q), v=a(r.v))

    # reference getters (larvio.cpp:2644-2735)
    def getTbw(self):
        s = self.state(); q = s["q"]
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = s["p"]
        return T

    def getVel(self):
        return self.state()["v"]

    def close(self):
        if self._h:
            _L().lvk_ekf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
