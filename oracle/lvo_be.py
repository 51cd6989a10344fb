"""ctypes binding of the oracle's back-end (oracle/be_*.c).  TEST INFRASTRUCTURE ONLY (see oracle/lvo.h)."""
import ctypes as C
import numpy as np
from . import lvo

POSE = np.dtype([("R", np.float64, 9), ("t", np.float64, 3)])
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
    _fields_ = ([(k, C.c_int) for k in _CFG_INT[:14]] + [("intrinsics", C.c_double * 4), ("T_cam_imu", C.c_double * 16)] +
                [(k, C.c_double) for k in _CFG_DBL] + [("calib_imu_instrinsic", C.c_int), ("reference_grid", C.c_int)])


def make_ekf_config(cfg, cls=EkfConfig):
    c = cls()
    for k in _CFG_INT + _CFG_DBL:
        setattr(c, k, cfg[k])
    if hasattr(c, "calib_imu_instrinsic"):
        c.calib_imu_instrinsic = int(cfg.get("calib_imu_instrinsic", 0))
    if hasattr(c, "reference_grid"):
        c.reference_grid = int(cfg.get("reference_grid", 1))      # see lvo.h: 1 = the reference's cells for out-of-range grid codes (the default here and in the product); 0 = the pre-round-6 bookkeeping
    c.intrinsics = (C.c_double * 4)(*cfg["intrinsics"])
    c.T_cam_imu = (C.c_double * 16)(*np.asarray(cfg["T_cam_imu"], np.float64).reshape(16))
    return c


_done = False


def _lib():
    global _done
    L = lvo.lib()
    if not _done:
        vp, i, d = C.c_void_p, C.c_int, C.c_double
        L.lvo_ekf_get_imu_intrinsics.argtypes = [vp, vp]; L.lvo_ekf_get_imu_intrinsics.restype = None
        L.lvo_ekf_set_imu_intrinsics.argtypes = [vp, vp]; L.lvo_ekf_set_imu_intrinsics.restype = None
        L.lvo_triangulate.argtypes = [vp, vp, i, i, vp, vp, vp, vp, vp]; L.lvo_triangulate.restype = i
        L.lvo_check_motion.argtypes = [vp, vp, vp, d]; L.lvo_check_motion.restype = i
        L.lvo_msckf_feature_jacobian.argtypes = [vp, vp, vp, vp, i, vp, i, i, i, i, vp, vp]; L.lvo_msckf_feature_jacobian.restype = i
        L.lvo_gating_gamma.argtypes = [vp, vp, i, i, vp, i, d]; L.lvo_gating_gamma.restype = d
        L.lvo_chi2_table.argtypes = [i]; L.lvo_chi2_table.restype = d
        L.lvo_qr_compress.argtypes = [vp, vp, i, i]
        L.lvo_ekf_update.argtypes = [vp, i, i, vp, i, vp, d, vp]
        L.lvo_ekf_create.argtypes = [C.POINTER(EkfConfig)]; L.lvo_ekf_create.restype = vp
        L.lvo_ekf_destroy.argtypes = [vp]
        L.lvo_ekf_process.argtypes = [vp, d, vp, i, vp, i, C.POINTER(i)]; L.lvo_ekf_process.restype = i
        L.lvo_ekf_set_state.argtypes = [vp, d, vp, vp, vp, vp, vp, vp, vp]
        L.lvo_ekf_dim.argtypes = [vp]; L.lvo_ekf_dim.restype = i
        L.lvo_ekf_is_initialized.argtypes = [vp]; L.lvo_ekf_is_initialized.restype = i
        L.lvo_ekf_get_state.argtypes = [vp, vp]
        L.lvo_ekf_get_cov.argtypes = [vp, vp]
        L.lvo_ekf_get_clones.argtypes = [vp, vp, i]; L.lvo_ekf_get_clones.restype = i
        L.lvo_ekf_get_features.argtypes = [vp, vp, vp, vp, i]; L.lvo_ekf_get_features.restype = i
        L.lvo_ekf_counters.argtypes = [vp, vp]
        L.lvo_ekf_static_try_init.argtypes = [vp, d, vp, i, vp, i, vp, vp]; L.lvo_ekf_static_try_init.restype = i
        L.lvo_stage_ekf1d_obs_jacobian.argtypes = [vp, vp, vp, d, vp, vp, vp, vp, vp, vp, vp]; L.lvo_stage_ekf1d_obs_jacobian.restype = i
        L.lvo_stage_hybrid_update_with_new.argtypes = [vp, i, vp, i, vp, vp, vp, vp, i, d, vp, vp]; L.lvo_stage_hybrid_update_with_new.restype = i
        L.lvo_stage_reanchor_row.argtypes = [vp, vp, vp, vp, vp, d, vp]; L.lvo_stage_reanchor_row.restype = i
        _done = True
    return L


_p = lvo._p


def triangulate(poses, obs, use_position=False, position_in=None):
    poses = np.ascontiguousarray(poses, POSE); obs = np.ascontiguousarray(obs, np.float64)
    pin = np.ascontiguousarray(position_in if position_in is not None else np.zeros(3), np.float64)
    pos = np.zeros(3); sol = np.zeros(3); idp = np.zeros(1); oa = np.zeros(3)
    ok = _lib().lvo_triangulate(_p(poses), _p(obs), len(poses), int(use_position), _p(pin), _p(pos), _p(sol), _p(idp), _p(oa))
    return bool(ok), pos, sol, float(idp[0]), oa


def msckf_feature_jacobian(clones, clone_rank, obs, obs_vel, p_w, N, leg_dim=22, if_fej=1, estimate_td=1):
    clones = np.ascontiguousarray(clones, CLONE); cr = np.ascontiguousarray(clone_rank, np.int32)
    obs = np.ascontiguousarray(obs, np.float64); ov = np.ascontiguousarray(obs_vel, np.float64); pw = np.ascontiguousarray(p_w, np.float64)
    M = len(cr); H = np.zeros((2 * M, N)); r = np.zeros(2 * M)
    k = _lib().lvo_msckf_feature_jacobian(_p(clones), _p(cr), _p(obs), _p(ov), M, _p(pw), N, leg_dim, if_fej, estimate_td, _p(H), _p(r))
    return H[:k].copy(), r[:k].copy()


def gating_gamma(H, r, P, sigma2):
    H = np.ascontiguousarray(H, np.float64); r = np.ascontiguousarray(r, np.float64); P = np.ascontiguousarray(P, np.float64)
    return _lib().lvo_gating_gamma(_p(H), _p(r), H.shape[0], H.shape[1], _p(P), P.shape[1], sigma2)


def chi2_table(dof):
    return _lib().lvo_chi2_table(dof)


def qr_compress(H, r):
    H = np.array(H, np.float64, order="C"); r = np.array(r, np.float64)
    _lib().lvo_qr_compress(_p(H), _p(r), H.shape[0], H.shape[1])
    n = min(H.shape)
    return H[:H.shape[1]] if H.shape[0] > H.shape[1] else H, r[:H.shape[1]] if H.shape[0] > H.shape[1] else r


def ekf_update(P, H, r, sigma2):
    P = np.array(P, np.float64, order="C"); H = np.ascontiguousarray(H, np.float64); r = np.ascontiguousarray(r, np.float64)
    N = P.shape[0]; dx = np.zeros(N)
    _lib().lvo_ekf_update(_p(P), N, N, _p(H), H.shape[0], _p(r), sigma2, _p(dx))
    return dx, P


def ekf1d_obs_jacobian(clone_k, clone_a, p_w, inv_depth, obs_anchor, z):
    """measurementJacobian_ekf_1didp (larvio.cpp:1117-1244), one observation, no FEJ: (is_not_anchor, Hf 2, Ha 2x6, Hx 2x6, He 2x6, r 2)"""
    k = np.ascontiguousarray(clone_k, CLONE).reshape(1); a = np.ascontiguousarray(clone_a, CLONE).reshape(1)
    pw = np.ascontiguousarray(p_w, np.float64); oa = np.ascontiguousarray(obs_anchor, np.float64); zz = np.ascontiguousarray(z, np.float64)
    Hf = np.zeros(2); Ha = np.zeros((2, 6)); Hx = np.zeros((2, 6)); He = np.zeros((2, 6)); r = np.zeros(2)
    ok = _lib().lvo_stage_ekf1d_obs_jacobian(_p(k), _p(a), _p(pw), float(inv_depth), _p(oa), _p(zz), _p(Hf), _p(Ha), _p(Hx), _p(He), _p(r))
    return bool(ok), Hf, Ha, Hx, He, r


def hybrid_update_with_new(P, Ho, ro, H1, H2, r1, sigma2):
    """measurementUpdate_hybrid incl. delayed initialisation of len(H2) new in-state features (larvio.cpp:1605-1862): (P_new, dx)"""
    P = np.array(P, np.float64, order="C"); N = P.shape[0]; n_acc = len(H2)
    Ho = np.ascontiguousarray(Ho, np.float64); ro = np.ascontiguousarray(ro, np.float64)
    H1 = np.ascontiguousarray(H1, np.float64); H2 = np.ascontiguousarray(H2, np.float64); r1 = np.ascontiguousarray(r1, np.float64)
    Po = np.zeros((N + n_acc, N + n_acc)); dx = np.zeros(N + n_acc)
    _lib().lvo_stage_hybrid_update_with_new(_p(P), N, _p(Ho), Ho.shape[0], _p(ro), _p(H1), _p(H2), _p(r1), n_acc, sigma2, _p(Po), _p(dx))
    return Po, dx


def reanchor_row(clone_old, clone_new, R_b2c, t_c_b, p_w, inv_depth_new):
    """updateFeatureCov_1didp (larvio.cpp:3125-3293), no FEJ: J = [d rho_new / d rho_old, d/d(old clone) 6, d/d(new clone) 6, d/d(extrinsics) 6]"""
    o = np.ascontiguousarray(clone_old, CLONE).reshape(1); n = np.ascontiguousarray(clone_new, CLONE).reshape(1)
    R = np.ascontiguousarray(R_b2c, np.float64); t = np.ascontiguousarray(t_c_b, np.float64); pw = np.ascontiguousarray(p_w, np.float64)
    J = np.zeros(19)
    ok = _lib().lvo_stage_reanchor_row(_p(o), _p(n), _p(R), _p(t), _p(pw), float(inv_depth_new), _p(J))
    assert ok
    return J


class Ekf:
    """the oracle's LarVio (larvio.cpp:363-461)"""

    def __init__(self, cfg):
        self._c = make_ekf_config(cfg)
        self.h = _lib().lvo_ekf_create(C.byref(self._c))

    def __del__(self):
        try:
            _lib().lvo_ekf_destroy(self.h)
        except Exception:
            pass

    def static_try_init(self, ts, feats, imu):
        """one step of the static initialiser alone (StaticInitializer::tryIncInit + assignInitialState): None, or dict(t, q, bg, erased)"""
        feats = np.ascontiguousarray(feats, lvo.OBS); imu = np.ascontiguousarray(imu, lvo.IMU)
        n = C.c_int(0); out = np.zeros(8)
        if not _lib().lvo_ekf_static_try_init(self.h, ts, _p(feats), len(feats), _p(imu), len(imu), C.byref(n), _p(out)):
            return None
        return dict(t=float(out[0]), q=out[1:5].copy(), bg=out[5:8].copy(), erased=int(n.value))

    def process(self, ts, feats, imu):
        feats = np.ascontiguousarray(feats, lvo.OBS); imu = np.ascontiguousarray(imu, lvo.IMU)
        n = C.c_int(0)
        ok = _lib().lvo_ekf_process(self.h, ts, _p(feats), len(feats), _p(imu), len(imu), C.byref(n))
        return bool(ok), n.value

    def set_state(self, t, q, p, v, bg, ba, gyro_old, acc_old):
        a = [np.ascontiguousarray(x, np.float64) for x in (q, p, v, bg, ba, gyro_old, acc_old)]
        _lib().lvo_ekf_set_state(self.h, t, *[_p(x) for x in a])

    def set_last_zupt_time(self, t):
        """the start as an initialiser leaves it (larvio.cpp:384): in-state features wait 5 s from t"""
        L = _lib(); L.lvo_ekf_set_last_zupt_time.argtypes = [C.c_void_p, C.c_double]; L.lvo_ekf_set_last_zupt_time(self.h, float(t))

    @property
    def dim(self):
        return _lib().lvo_ekf_dim(self.h)

    @property
    def initialized(self):
        return bool(_lib().lvo_ekf_is_initialized(self.h))

    def state(self):
        o = np.zeros(30); _lib().lvo_ekf_get_state(self.h, _p(o))
        return dict(t=o[0], q=o[1:5].copy(), v=o[5:8].copy(), p=o[8:11].copy(), bg=o[11:14].copy(), ba=o[14:17].copy(),
                    R_b2c=o[17:26].reshape(3, 3).copy(), t_c_b=o[26:29].copy(), td=o[29])

    def imu_intrinsics(self):
        """T1 T2 T3 A1 A2 A3 M1 M2 (24 numbers; state columns 22..45 when calib_imu_instrinsic = 1)"""
        o = np.zeros(24); _lib().lvo_ekf_get_imu_intrinsics(self.h, _p(o)); return o

    def set_imu_intrinsics(self, v):
        v = np.ascontiguousarray(v, np.float64); assert v.shape == (24,)
        _lib().lvo_ekf_set_imu_intrinsics(self.h, _p(v))

    def cov(self):
        N = self.dim; P = np.zeros((N, N)); _lib().lvo_ekf_get_cov(self.h, _p(P)); return P

    def clones(self):
        o = np.zeros(256, CLONE); n = _lib().lvo_ekf_get_clones(self.h, _p(o), 256); return o[:n].copy()

    def features(self):
        ids = np.zeros(4096, np.int64); idp = np.zeros(4096); pos = np.zeros((4096, 3))
        n = _lib().lvo_ekf_get_features(self.h, _p(ids), _p(idp), _p(pos), 4096)
        return ids[:n].copy(), idp[:n].copy(), pos[:n].copy()

    def counters(self):
        o = np.zeros(7, np.int64); _lib().lvo_ekf_counters(self.h, _p(o))
        return dict(hybrid=int(o[0]), msckf=int(o[1]), last_rows=int(o[2]), zupt=int(o[3]), gated_in=int(o[4]), gated_out=int(o[5]), map=int(o[6]))
