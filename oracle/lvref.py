"""ctypes bindings of oracle/_ref/*.so: the REFERENCE's own sources compiled where they lie under /root/reference (never copied) against
stand-in headers for the libraries this image does not have (oracle/Makefile, target `ref`; oracle/ref_shim .. ref_shim4):
    liblvref_larvio.so    src/larvio.cpp + FlexibleInitializer / StaticInitializer / feature_manager   -> RefLarVio (the whole filter)
    liblvref_imgproc.so   src/image_processor.cpp + ORBDescriptor.cpp over the oracle's OpenCV restatements -> RefImageProcessor
    liblvref_dyninit.so   src/DynamicInitializer.cpp + initial_sfm / solve_5pts / initial_alignment / feature_manager over stand-in minimisers -> dynamic_init
    liblvref_orb.so       src/ORBDescriptor.cpp                                                         -> RefOrb
    liblvref_feature.so   include/larvio/feature.hpp, math_utils.hpp                                    -> feature_initialize, feature_check_motion, math_*
    liblvref_preint.so / _align.so / _static.so / _fm.so   ImuPreintegration.h, initial_alignment.cpp, StaticInitializer.cpp, feature_manager.cpp
TEST INFRASTRUCTURE ONLY: these pin the restatements in oracle/ (and, through fixtures they wrote, the product); the product never loads them.
The libraries are built here (where /root/reference exists) by __graft_entry__.build(); on the GPU box only the prebuilt files exist."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "liblvref_orb.so")
_lib = None


def available(build=True):
    if os.path.exists(_SO):
        return True
    if build and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/liblvref_orb.so is missing and /root/reference is not here to build it from")
        L = C.CDLL(_SO)
        vp, i = C.c_void_p, C.c_int
        L.lvref_orb_create.restype = vp; L.lvref_orb_create.argtypes = [vp, i, i, i, i, i]
        L.lvref_orb_destroy.argtypes = [vp]
        L.lvref_orb_describe.restype = i; L.lvref_orb_describe.argtypes = [vp, vp, i, vp, vp]
        L.lvref_orb_hamming.restype = i; L.lvref_orb_hamming.argtypes = [vp, vp]
        L.lvref_orb_planes.restype = i; L.lvref_orb_planes.argtypes = [vp, vp, vp]
        L.lvref_orb_umax.restype = i; L.lvref_orb_umax.argtypes = [vp, vp]
        L.lvref_orb_pattern.restype = i; L.lvref_orb_pattern.argtypes = [vp, vp]
        _lib = L
    return _lib


class RefOrb:
    """larvio::ORBdescriptor(image, 2, nlevels) where `image` is the view of level 0 inside its padded LK-pyramid buffer
    (`padded`: (h + 2 pad) x (w + 2 pad) u8, as image_processor.cpp:150 hands it over); pad = 0 for a stand-alone image."""

    def __init__(self, padded, pad, nlevels=2):
        padded = np.ascontiguousarray(padded, np.uint8)
        self.h, self.w = padded.shape[0] - 2 * pad, padded.shape[1] - 2 * pad
        self._keep = padded
        self.o = lib().lvref_orb_create(padded.ctypes.data, self.w, self.h, pad, padded.shape[1], nlevels)
        if not self.o:
            raise MemoryError

    def __del__(self):
        if getattr(self, "o", None):
            lib().lvref_orb_destroy(self.o); self.o = None

    def describe(self, points):
        p = np.ascontiguousarray(points, np.float32).reshape(-1, 2); n = len(p)
        desc = np.empty((n, 32), np.uint8); ang = np.empty(n, np.float32)
        if lib().lvref_orb_describe(self.o, p.ctypes.data, n, desc.ctypes.data, ang.ctypes.data) != 0:
            raise RuntimeError("computeDescriptors returned false")
        return desc, ang

    def planes(self):
        ext = np.empty((self.h + 64, self.w + 64), np.uint8); blur = np.empty_like(ext)
        b = lib().lvref_orb_planes(self.o, ext.ctypes.data, blur.ctypes.data)
        if b != 32:
            raise RuntimeError(f"unexpected mosaic border {b}")
        return ext, blur

    def umax(self):
        u = np.zeros(16, np.int32); n = lib().lvref_orb_umax(self.o, u.ctypes.data); return u[:n]

    def pattern(self):
        p = np.zeros(1024, np.int32); n = lib().lvref_orb_pattern(self.o, p.ctypes.data); return p[:2 * n]


def hamming(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().lvref_orb_hamming(a.ctypes.data, b.ctypes.data)


# ---------------------------------------------------------------------------------------------- the reference's Feature code
_SO_F = os.path.join(_HERE, "_ref", "liblvref_feature.so")
_lib_f = None


def feature_available(build=True):
    if os.path.exists(_SO_F):
        return True
    if build and os.path.isdir("/root/reference/include/larvio"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return os.path.exists(_SO_F)


def _libf():
    global _lib_f
    if _lib_f is None:
        if not feature_available():
            raise RuntimeError("oracle/_ref/liblvref_feature.so is missing and /root/reference is not here to build it from")
        L = C.CDLL(_SO_F)
        vp, i, ll, d = C.c_void_p, C.c_int, C.c_longlong, C.c_double
        L.lvref_feature_initialize.argtypes = [i, i, vp, vp, vp, i, vp, vp, ll, i, vp, vp]; L.lvref_feature_initialize.restype = i
        L.lvref_feature_check_motion.argtypes = [i, vp, vp, vp, i, vp, vp, i, d]; L.lvref_feature_check_motion.restype = i
        _lib_f = L
    return _lib_f


def _views(state_ids, q_cam, p_cam, obs_ids, obs_uv):
    return (np.ascontiguousarray(state_ids, np.int64), np.ascontiguousarray(q_cam, np.float64).reshape(-1, 4), np.ascontiguousarray(p_cam, np.float64).reshape(-1, 3),
            np.ascontiguousarray(obs_ids, np.int64), np.ascontiguousarray(obs_uv, np.float64).reshape(-1, 2))


def feature_initialize(mode, state_ids, q_cam, p_cam, obs_ids, obs_uv, curr_id=-1, is_initialized=False, position_in=None):
    """Feature::initializePosition (mode 0, feature.hpp:383-552), initializePosition_AssignAnchor (1, :554-721) or
    initializeInvParamPosition (2, :723-890) of the compiled reference on a feature with the given observations (state id -> (u, v)) and
    camera states (id -> orientation_cam [x y z w], position_cam).  -> (ok, dict of the feature's members after the call)"""
    sid, q, p, oid, uv = _views(state_ids, q_cam, p_cam, obs_ids, obs_uv)
    pin = np.ascontiguousarray(position_in if position_in is not None else np.zeros(3), np.float64)
    out = np.zeros(15)
    ok = _libf().lvref_feature_initialize(int(mode), len(sid), sid.ctypes.data, q.ctypes.data, p.ctypes.data, len(oid), oid.ctypes.data, uv.ctypes.data,
                                          int(curr_id), int(bool(is_initialized)), pin.ctypes.data, out.ctypes.data)
    return bool(ok), dict(position=out[0:3].copy(), position_fej=out[3:6].copy(), inv_depth=float(out[6]), obs_anchor=out[7:10].copy(), id_anchor=int(out[10]),
                          inv_param=out[11:14].copy(), is_initialized=bool(out[14]))


def math_small_angle(w):
    """math_utils.hpp of the compiled reference: (skewSymmetric(w) 3x3, smallAngleQuaternion(w) [x y z w], getSmallAngleQuaternion(w) [x y z w])"""
    L = _libf()
    L.lvref_math_small_angle.argtypes = [C.c_void_p] * 4
    w = np.ascontiguousarray(w, np.float64); S = np.zeros(9); a = np.zeros(4); b = np.zeros(4)
    L.lvref_math_small_angle(w.ctypes.data, S.ctypes.data, a.ctypes.data, b.ctypes.data)
    return S.reshape(3, 3), a, b


def math_quat(q, p):
    """math_utils.hpp of the compiled reference: (quaternionToRotation(q) 3x3, rotationToQuaternion(of that) [x y z w],
    quaternionMultiplication(q, p) [x y z w] - normalised, as the reference's is)"""
    L = _libf()
    L.lvref_math_quat.argtypes = [C.c_void_p] * 5
    q = np.ascontiguousarray(q, np.float64); p = np.ascontiguousarray(p, np.float64); R = np.zeros(9); q2 = np.zeros(4); qp = np.zeros(4)
    L.lvref_math_quat(q.ctypes.data, p.ctypes.data, R.ctypes.data, q2.ctypes.data, qp.ctypes.data)
    return R.reshape(3, 3), q2, qp


def feature_check_motion(state_ids, q_cam, p_cam, obs_ids, obs_uv, if_tracked, translation_threshold):
    """Feature::checkMotion (feature.hpp:334-381) of the compiled reference"""
    sid, q, p, oid, uv = _views(state_ids, q_cam, p_cam, obs_ids, obs_uv)
    return bool(_libf().lvref_feature_check_motion(len(sid), sid.ctypes.data, q.ctypes.data, p.ctypes.data, len(oid), oid.ctypes.data, uv.ctypes.data,
                                                   int(bool(if_tracked)), float(translation_threshold)))


# ---------------------------------------------------------------------------------------------- the reference's IMU pre-integration
_SO_P = os.path.join(_HERE, "_ref", "liblvref_preint.so")
_lib_p = None


def preint_available(build=True):
    if os.path.exists(_SO_P):
        return True
    if build and os.path.isdir("/root/reference/include/Initializer"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return os.path.exists(_SO_P)


def preintegrate(acc0, gyr0, ba, bg, dt, acc, gyr, rebias=None):
    """IntegrationBase (ImuPreintegration.h:27-230) of the compiled reference: linearised at (acc0, gyr0, ba, bg), the samples pushed,
    optionally re-propagated about rebias = (ba2, bg2).  -> dict(dp, dq [x y z w], dv, sum_dt, dq_dbg, dp_dbg, dv_dbg, dp_dba, dv_dba)"""
    global _lib_p
    if _lib_p is None:
        if not preint_available():
            raise RuntimeError("oracle/_ref/liblvref_preint.so is missing and /root/reference is not here to build it from")
        _lib_p = C.CDLL(_SO_P)
        vp, i = C.c_void_p, C.c_int
        _lib_p.lvref_preintegrate.argtypes = [vp, vp, vp, vp, i, vp, vp, vp, i, vp, vp, vp]; _lib_p.lvref_preintegrate.restype = i
    f = lambda a: np.ascontiguousarray(a, np.float64)
    a0, g0, ba, bg, dt, acc, gyr = f(acc0), f(gyr0), f(ba), f(bg), f(dt), f(acc).reshape(-1, 3), f(gyr).reshape(-1, 3)
    ba2, bg2 = (f(rebias[0]), f(rebias[1])) if rebias is not None else (np.zeros(3), np.zeros(3))
    out = np.zeros(56)
    _lib_p.lvref_preintegrate(a0.ctypes.data, g0.ctypes.data, ba.ctypes.data, bg.ctypes.data, len(dt), dt.ctypes.data, acc.ctypes.data, gyr.ctypes.data,
                              int(rebias is not None), ba2.ctypes.data, bg2.ctypes.data, out.ctypes.data)
    m = lambda k: out[11 + 9 * k:20 + 9 * k].reshape(3, 3).copy()
    return dict(dp=out[0:3].copy(), dq=out[3:7].copy(), dv=out[7:10].copy(), sum_dt=float(out[10]), dq_dbg=m(0), dp_dbg=m(1), dv_dbg=m(2), dp_dba=m(3), dv_dba=m(4))


# ---------------------------------------------------------------------------------------------- the reference's visual-inertial alignment
_SO_A = os.path.join(_HERE, "_ref", "liblvref_align.so")
_lib_a = None


def align_available(build=True):
    if os.path.exists(_SO_A):
        return True
    if build and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return os.path.exists(_SO_A)


def visual_imu_alignment(t, R, T, heads, streams, bg0, tic):
    """VisualIMUAlignment (src/initial_alignment.cpp:204-212) of the compiled reference on a window: frame keys t, rotations R (n, 3, 3) and
    positions T (n, 3) as all_image_frame holds them; heads[j] = (acc0, gyr0) and streams[j] = (m, 7) samples (dt, acc, gyr) of the
    pre-integration frame j >= 1 carries (index 0 unused).  -> dict(ok, bg, g, x)"""
    global _lib_a
    if _lib_a is None:
        if not align_available():
            raise RuntimeError("oracle/_ref/liblvref_align.so is missing and /root/reference is not here to build it from")
        _lib_a = C.CDLL(_SO_A)
        vp, i = C.c_void_p, C.c_int
        _lib_a.lvref_visual_imu_alignment.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, vp, vp]; _lib_a.lvref_visual_imu_alignment.restype = i
    f = lambda a: np.ascontiguousarray(a, np.float64)
    n = len(t)
    tt, RR, TT = f(t), f(R).reshape(n, 9), f(T).reshape(n, 3)
    ns = np.array([0] + [len(streams[j]) for j in range(1, n)], np.int32)
    hd = np.zeros((n, 6)); hd[1:] = [np.concatenate(heads[j]) for j in range(1, n)]
    sm = f(np.concatenate([np.asarray(streams[j], float).reshape(-1, 7) for j in range(1, n)]))
    out = np.zeros(8 + 3 * n + 8)
    ok = _lib_a.lvref_visual_imu_alignment(n, tt.ctypes.data, RR.ctypes.data, TT.ctypes.data, ns.ctypes.data, hd.ctypes.data, sm.ctypes.data, f(bg0).ctypes.data, f(tic).ctypes.data,
                                           out.ctypes.data)
    nx = int(out[7])
    return dict(ok=bool(ok), bg=out[1:4].copy(), g=out[4:7].copy(), x=out[8:8 + nx].copy())


# ---------------------------------------------------------------------------------------------- the reference's static initialiser
_SO_S = os.path.join(_HERE, "_ref", "liblvref_static.so")
_lib_s = None


def static_available(build=True):
    if os.path.exists(_SO_S):
        return True
    if build and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return os.path.exists(_SO_S)


class RefStaticInitializer:
    """larvio::StaticInitializer (src/StaticInitializer.cpp) of the compiled reference: try_init(ts, ids, uv, imu) per message ->
    None, or dict(t, q [x y z w], bg, erased, gyro_old, acc_old) = what tryIncInit + assignInitialState hand to the filter"""

    def __init__(self, max_feature_dis, static_num, td=0.0, Ma=np.eye(3), Tg=np.eye(3), As=np.zeros((3, 3))):
        global _lib_s
        if _lib_s is None:
            if not static_available():
                raise RuntimeError("oracle/_ref/liblvref_static.so is missing and /root/reference is not here to build it from")
            _lib_s = C.CDLL(_SO_S)
            vp, i, d = C.c_void_p, C.c_int, C.c_double
            _lib_s.lvref_static_create.argtypes = [d, i, d, vp, vp, vp]; _lib_s.lvref_static_create.restype = vp
            _lib_s.lvref_static_destroy.argtypes = [vp]
            _lib_s.lvref_static_try.argtypes = [vp, d, i, vp, vp, i, vp, vp]; _lib_s.lvref_static_try.restype = i
        f = lambda a: np.ascontiguousarray(a, np.float64)
        self.h = _lib_s.lvref_static_create(float(max_feature_dis), int(static_num), float(td), f(Ma).ctypes.data, f(Tg).ctypes.data, f(As).ctypes.data)

    def __del__(self):
        if getattr(self, "h", None):
            _lib_s.lvref_static_destroy(self.h); self.h = None

    def try_init(self, ts, ids, uv, imu7):
        ids = np.ascontiguousarray(ids, np.int64); uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2); imu7 = np.ascontiguousarray(imu7, np.float64).reshape(-1, 7)
        out = np.zeros(15)
        if not _lib_s.lvref_static_try(self.h, float(ts), len(ids), ids.ctypes.data, uv.ctypes.data, len(imu7), imu7.ctypes.data, out.ctypes.data):
            return None
        return dict(t=float(out[0]), q=out[1:5].copy(), bg=out[5:8].copy(), erased=int(out[8]), gyro_old=out[9:12].copy(), acc_old=out[12:15].copy())


# ---------------------------------------------------------------------------------------------- the reference's FeatureManager (window bookkeeping)
_SO_M = os.path.join(_HERE, "_ref", "liblvref_fm.so")
_lib_m = None


def fm_available(build=True):
    if os.path.exists(_SO_M):
        return True
    if build and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return os.path.exists(_SO_M)


class RefFeatureManager:
    """larvio::FeatureManager (src/feature_manager.cpp) of the compiled reference"""

    def __init__(self):
        global _lib_m
        if _lib_m is None:
            if not fm_available():
                raise RuntimeError("oracle/_ref/liblvref_fm.so is missing and /root/reference is not here to build it from")
            _lib_m = C.CDLL(_SO_M)
            vp, i, d = C.c_void_p, C.c_int, C.c_double
            _lib_m.lvref_fm_create.restype = vp
            _lib_m.lvref_fm_destroy.argtypes = [vp]
            _lib_m.lvref_fm_add.argtypes = [vp, i, i, vp, vp, d]; _lib_m.lvref_fm_add.restype = i
            _lib_m.lvref_fm_corresponding.argtypes = [vp, i, i, vp, i]; _lib_m.lvref_fm_corresponding.restype = i
            _lib_m.lvref_fm_remove_back.argtypes = [vp]
            _lib_m.lvref_fm_feature_count.argtypes = [vp]; _lib_m.lvref_fm_feature_count.restype = i
        self.h = _lib_m.lvref_fm_create()

    def __del__(self):
        if getattr(self, "h", None):
            _lib_m.lvref_fm_destroy(self.h); self.h = None

    def add(self, frame_count, ids, uvv, td):
        """addFeatureCheckParallax(frame_count, image, td): uvv = (n, 4) u, v, u_vel, v_vel -> True = marginalise the oldest frame"""
        ids = np.ascontiguousarray(ids, np.int64); uvv = np.ascontiguousarray(uvv, np.float64).reshape(-1, 4)
        return bool(_lib_m.lvref_fm_add(self.h, int(frame_count), len(ids), ids.ctypes.data, uvv.ctypes.data, float(td)))

    def corresponding(self, l, r):
        out = np.zeros((4096, 4)); n = _lib_m.lvref_fm_corresponding(self.h, int(l), int(r), out.ctypes.data, 4096)
        return out[:n].copy()

    def remove_back(self):
        _lib_m.lvref_fm_remove_back(self.h)

    def feature_count(self):
        return _lib_m.lvref_fm_feature_count(self.h)


# ------------------------------------------------------------------------------------------------ the reference's filter itself
_SO_LARVIO = os.environ.get("LVREF_LARVIO_SO", os.path.join(_HERE, "_ref", "liblvref_larvio.so"))
_libv = None


def larvio_available(build=True):
    if os.path.exists(_SO_LARVIO):
        return True
    if build and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return os.path.exists(_SO_LARVIO)


def _libl():
    global _libv
    if _libv is None:
        if not larvio_available():
            raise RuntimeError("oracle/_ref/liblvref_larvio.so is missing and /root/reference is not here to build it from")
        L = C.CDLL(_SO_LARVIO)
        vp, i, d = C.c_void_p, C.c_int, C.c_double
        L.lvref_larvio_create.restype = vp; L.lvref_larvio_create.argtypes = [C.c_char_p]
        L.lvref_larvio_destroy.argtypes = [vp]
        L.lvref_larvio_set_state.argtypes = [vp, d] + [vp] * 7
        L.lvref_larvio_process.restype = i; L.lvref_larvio_process.argtypes = [vp, d, i, vp, i, vp, C.POINTER(i)]
        L.lvref_larvio_dim.restype = i; L.lvref_larvio_dim.argtypes = [vp]
        L.lvref_larvio_initialized.restype = i; L.lvref_larvio_initialized.argtypes = [vp]
        L.lvref_larvio_get_state.argtypes = [vp, vp]; L.lvref_larvio_get_cov.argtypes = [vp, vp]
        L.lvref_larvio_get_clones.restype = i; L.lvref_larvio_get_clones.argtypes = [vp, vp, i]
        L.lvref_larvio_get_features.restype = i; L.lvref_larvio_get_features.argtypes = [vp, vp, vp, vp, i]
        L.lvref_larvio_map_size.restype = i; L.lvref_larvio_map_size.argtypes = [vp]
        L.lvref_larvio_get_clone_cams.restype = i; L.lvref_larvio_get_clone_cams.argtypes = [vp, vp, i]
        L.lvref_larvio_chi2.restype = d; L.lvref_larvio_chi2.argtypes = [vp, i]
        L.lvref_larvio_getters.argtypes = [vp, vp, vp, vp, vp]
        _libv = L
    return _libv


def larvio_yaml(cfg, output_dir):
    """the oracle's configuration dict (larvio_amd.synthetic.backend_config) written the way the reference ships its own
    (config/euroc.yaml: `%YAML:1.0`, an `!!opencv-matrix` node for T_cam_imu) - every key LarVio::loadParameters reads
    (larvio.cpp:58-311); feature_idp_dim 1 and use_schmidt 0 are the modes this repository implements"""
    T = np.asarray(cfg["T_cam_imu"], np.float64).reshape(4, 4)
    fx, fy, cx, cy = cfg["intrinsics"]
    g = lambda k: repr(float(cfg[k]))
    lines = ["%YAML:1.0", "", 'output_dir: "%s"' % (output_dir if output_dir.endswith("/") else output_dir + "/"),
             "if_FEJ: %d" % cfg["if_fej"], "estimate_extrin: %d" % cfg["estimate_extrin"], "estimate_td: %d" % cfg["estimate_td"],
             "calib_imu_instrinsic: %d" % int(cfg.get("calib_imu_instrinsic", 0)),
             "resolution_width: %d" % cfg["width"], "resolution_height: %d" % cfg["height"],
             "intrinsics:", "   fx: %r" % float(fx), "   fy: %r" % float(fy), "   cx: %r" % float(cx), "   cy: %r" % float(cy),
             "T_cam_imu: !!opencv-matrix", "   rows: 4", "   cols: 4", "   dt: d", "   data:",
             "    [" + ",\n     ".join(", ".join(repr(float(x)) for x in row) for row in T) + "]",
             "td: " + g("td"), "pub_frequency: " + g("pub_frequency"), "sw_size: %d" % cfg["sw_size"],
             "position_std_threshold: 8.0", "rotation_threshold: " + g("rotation_threshold"), "translation_threshold: " + g("translation_threshold"),
             "tracking_rate_threshold: " + g("tracking_rate_threshold"), "feature_translation_threshold: " + g("feature_translation_threshold"),
             "noise_gyro: " + g("noise_gyro"), "noise_acc: " + g("noise_acc"), "noise_gyro_bias: " + g("noise_gyro_bias"), "noise_acc_bias: " + g("noise_acc_bias"),
             "noise_feature: " + g("noise_feature"), "zupt_noise_v: " + g("zupt_noise_v"), "zupt_noise_p: " + g("zupt_noise_p"), "zupt_noise_q: " + g("zupt_noise_q")]
    for k in ("orientation", "velocity", "position", "gyro_bias", "acc_bias", "extrin_rot", "extrin_trans"):
        lines.append("initial_covariance_%s: %s" % (k, g("initial_covariance_" + k)))
    lines += ["reset_fej_threshold: 10.11", "if_ZUPT_valid: %d" % cfg["if_zupt_valid"], "zupt_max_feature_dis: " + g("zupt_max_feature_dis"),
              "static_duration: " + g("static_duration"), "imu_rate: " + g("imu_rate"), "max_track_len: %d" % cfg["max_track_len"],
              "feature_idp_dim: %d" % int(cfg.get("feature_idp_dim", 1)), "use_schmidt: %d" % int(cfg.get("use_schmidt", 0)), "least_observation_number: %d" % cfg["least_observation_number"],
              "max_features_in_one_grid: %d" % cfg["max_features_in_one_grid"], "aug_grid_rows: %d" % cfg["aug_grid_rows"], "aug_grid_cols: %d" % cfg["aug_grid_cols"], ""]
    return "\n".join(lines)


class RefLarVio:
    """larvio::LarVio (src/larvio.cpp) of the compiled reference, driven like the oracle's lvo_be.Ekf: process(ts, feats, imu) takes the
    message (lvo.OBS records) and the driver's whole IMU view for it (the samples the previous call did not erase come first, as
    tests/feature_sim.drive hands them over) and returns (processFeatures' answer, samples erased from that view)."""

    def __init__(self, cfg, workdir):
        os.makedirs(workdir, exist_ok=True)
        self._yaml = os.path.join(workdir, "lvref_larvio.yaml")
        with open(self._yaml, "w") as f:
            f.write(larvio_yaml(cfg, workdir))
        self.h = _libl().lvref_larvio_create(self._yaml.encode())
        if not self.h:
            raise RuntimeError("the reference's LarVio::initialize() failed on " + self._yaml)
        self._held = 0                      # samples of the caller's view that already sit in the wrapper's buffer

    @classmethod
    def from_yaml(cls, path):
        """LarVio(config_file) + initialize() on a configuration file as it is (e.g. the reference's own config/euroc.yaml; its output_dir
        must exist: the reference opens two log files there)"""
        self = cls.__new__(cls); self._yaml = path; self._held = 0
        self.h = _libl().lvref_larvio_create(path.encode())
        if not self.h:
            raise RuntimeError("the reference's LarVio::initialize() failed on " + path)
        return self

    PARAMS = ("if_fej estimate_extrin estimate_td if_zupt_valid sw_size max_track_len least_observation_number max_features_in_one_grid aug_grid_rows aug_grid_cols "
              "pub_frequency imu_rate width height fx fy cx cy td noise_gyro noise_acc noise_gyro_bias noise_acc_bias noise_feature initial_covariance_orientation "
              "initial_covariance_velocity initial_covariance_position initial_covariance_gyro_bias initial_covariance_acc_bias initial_covariance_extrin_rot "
              "initial_covariance_extrin_trans rotation_threshold translation_threshold tracking_rate_threshold feature_translation_threshold zupt_max_feature_dis "
              "zupt_noise_v zupt_noise_p zupt_noise_q static_duration feature_idp_dim use_schmidt calib_imu_instrinsic").split()

    def params(self):
        """what LarVio::loadParameters (larvio.cpp:58-311) made of the file, by lvk_ekf_config's field names (+ R_imu_cam0 3x3, t_cam0_imu)"""
        L = _libl(); L.lvref_larvio_params.argtypes = [C.c_void_p, C.c_void_p]
        o = np.zeros(55); L.lvref_larvio_params(self.h, o.ctypes.data)
        d = dict(zip(self.PARAMS, o[:43].tolist())); d["R_imu_cam0"] = o[43:52].reshape(3, 3).copy(); d["t_cam0_imu"] = o[52:55].copy()
        return d

    def __del__(self):
        if getattr(self, "h", None):
            _libl().lvref_larvio_destroy(self.h); self.h = None

    def set_state(self, t, q, p, v, bg, ba, gyro_old, acc_old):
        a = [np.ascontiguousarray(x, np.float64) for x in (q, p, v, bg, ba, gyro_old, acc_old)]
        _libl().lvref_larvio_set_state(self.h, float(t), *[x.ctypes.data for x in a])

    def process(self, ts, feats, imu):
        n = len(feats); f = np.zeros((n, 9))
        for k, name in enumerate(("id", "u", "v", "u_init", "v_init", "u_vel", "v_vel", "u_init_vel", "v_init_vel")):
            f[:, k] = feats[name]
        new = imu[self._held:]                                       # the view's head is what the buffer still holds
        m = np.zeros((len(new), 7)); m[:, 0] = new["t"]; m[:, 1:4] = new["gyro"]; m[:, 4:7] = new["acc"]
        left = C.c_int(0)
        ok = _libl().lvref_larvio_process(self.h, float(ts), n, f.ctypes.data, len(m), m.ctypes.data, C.byref(left))
        used = len(imu) - left.value
        self._held = left.value
        return bool(ok), used

    @property
    def dim(self):
        return _libl().lvref_larvio_dim(self.h)

    @property
    def initialized(self):
        return bool(_libl().lvref_larvio_initialized(self.h))

    def state(self):
        o = np.zeros(30); _libl().lvref_larvio_get_state(self.h, o.ctypes.data)
        return dict(t=o[0], q=o[1:5].copy(), v=o[5:8].copy(), p=o[8:11].copy(), bg=o[11:14].copy(), ba=o[14:17].copy(),
                    R_b2c=o[17:26].reshape(3, 3).copy(), t_c_b=o[26:29].copy(), td=o[29])

    def cov(self):
        N = self.dim; P = np.zeros((N, N)); _libl().lvref_larvio_get_cov(self.h, P.ctypes.data); return P

    def set_last_zupt_time(self, t):
        L = _libl(); L.lvref_larvio_set_last_zupt_time.argtypes = [C.c_void_p, C.c_double]; L.lvref_larvio_set_last_zupt_time(self.h, float(t))

    def clones(self):
        o = np.zeros((256, 12)); n = _libl().lvref_larvio_get_clones(self.h, o.ctypes.data, 256)
        return dict(id=o[:n, 0].astype(np.int64), time=o[:n, 1].copy(), q=o[:n, 2:6].copy(), p=o[:n, 6:9].copy(), p_fej=o[:n, 9:12].copy())

    def clone_cams(self):
        """the clones' camera poses as feature.hpp reads them: ids, orientation_cam (n x 4), position_cam (n x 3)"""
        o = np.zeros((256, 8)); n = _libl().lvref_larvio_get_clone_cams(self.h, o.ctypes.data, 256)
        return o[:n, 0].astype(np.int64), o[:n, 1:5].copy(), o[:n, 5:8].copy()

    def features(self):
        ids = np.zeros(4096, np.int64); idp = np.zeros(4096); pos = np.zeros((4096, 3))
        n = _libl().lvref_larvio_get_features(self.h, ids.ctypes.data, idp.ctypes.data, pos.ctypes.data, 4096)
        return ids[:n].copy(), idp[:n].copy(), pos[:n].copy()

    def getters(self):
        """LarVio::getTbw / getVel / getPpose / getPvel (larvio.cpp:2645-2700) -> (T 4x4, v, P_pose 6x6, P_vel 3x3)"""
        T = np.zeros(16); v = np.zeros(3); Pp = np.zeros(36); Pv = np.zeros(9)
        _libl().lvref_larvio_getters(self.h, T.ctypes.data, v.ctypes.data, Pp.ctypes.data, Pv.ctypes.data)
        return T.reshape(4, 4), v, Pp.reshape(6, 6), Pv.reshape(3, 3)

    def map(self):
        """map_server: (ids, observation counts, flags: 1 is_initialized, 2 in_state, 4 ekf_feature)"""
        L = _libl(); L.lvref_larvio_map.restype = C.c_int; L.lvref_larvio_map.argtypes = [C.c_void_p] * 4 + [C.c_int]
        ids = np.zeros(8192, np.int64); n_obs = np.zeros(8192, np.int32); fl = np.zeros(8192, np.int32)
        n = L.lvref_larvio_map(self.h, ids.ctypes.data, n_obs.ctypes.data, fl.ctypes.data, 8192)
        return ids[:n].copy(), n_obs[:n].copy(), fl[:n].copy()

    def map_size(self):
        return _libl().lvref_larvio_map_size(self.h)

    def chi2(self, dof):
        return _libl().lvref_larvio_chi2(self.h, dof)


# ------------------------------------------------------------------------------------------------ the reference's front-end class
_SO_IMGPROC = os.environ.get("LVREF_IMGPROC_SO", os.path.join(_HERE, "_ref", "liblvref_imgproc.so"))
_libi = None


def imgproc_available(build=True):
    if os.path.exists(_SO_IMGPROC):
        return True
    if build and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return os.path.exists(_SO_IMGPROC)


def _libip():
    global _libi
    if _libi is None:
        if not imgproc_available():
            raise RuntimeError("oracle/_ref/liblvref_imgproc.so is missing and /root/reference is not here to build it from")
        L = C.CDLL(_SO_IMGPROC)
        vp, i, d = C.c_void_p, C.c_int, C.c_double
        L.lvref_imgproc_create.restype = vp; L.lvref_imgproc_create.argtypes = [C.c_char_p]
        L.lvref_imgproc_destroy.argtypes = [vp]
        L.lvref_imgproc_process.restype = i; L.lvref_imgproc_process.argtypes = [vp, d, vp, i, i, i, i, vp, vp, i, C.POINTER(i)]
        L.lvref_imgproc_state.restype = i; L.lvref_imgproc_state.argtypes = [vp]
        L.lvref_imgproc_tracks.restype = i; L.lvref_imgproc_tracks.argtypes = [vp, vp, vp, vp, vp, vp, i]
        L.lvref_imgproc_new_pts.restype = i; L.lvref_imgproc_new_pts.argtypes = [vp, vp, i]
        L.lvref_imgproc_counts.restype = i; L.lvref_imgproc_counts.argtypes = [vp, vp]
        _libi = L
    return _libi


def imgproc_yaml(cfg, output_dir):
    """the oracle's front-end configuration dict (larvio_amd.synthetic.frontend_config) in the reference's YAML dialect: every key
    ImageProcessor::loadParameters reads (image_processor.cpp:44-113).  T_cam_imu: rotation = R_cam_imu^T (:93), translation unused by the
    front-end."""
    fx, fy, cx, cy = cfg["intrinsics"]; k1, k2, p1, p2 = cfg["distortion"]
    T = np.eye(4); T[:3, :3] = np.asarray(cfg["R_cam_imu"], np.float64).reshape(3, 3).T
    lines = ["%YAML:1.0", "", 'output_dir: "%s"' % (output_dir if output_dir.endswith("/") else output_dir + "/"),
             'distortion_model: "%s"' % (cfg["distortion_model"] if isinstance(cfg["distortion_model"], str) else ("radtan", "equidistant")[int(cfg["distortion_model"])]), "resolution_width: %d" % cfg["width"], "resolution_height: %d" % cfg["height"],
             "intrinsics:", "   fx: %r" % float(fx), "   fy: %r" % float(fy), "   cx: %r" % float(cx), "   cy: %r" % float(cy),
             "distortion_coeffs:", "   k1: %r" % float(k1), "   k2: %r" % float(k2), "   p1: %r" % float(p1), "   p2: %r" % float(p2),
             "T_cam_imu: !!opencv-matrix", "   rows: 4", "   cols: 4", "   dt: d", "   data:",
             "    [" + ",\n     ".join(", ".join(repr(float(x)) for x in row) for row in T) + "]",
             "pyramid_levels: %d" % cfg["pyramid_levels"], "patch_size: %d" % cfg["patch_size"], "fast_threshold: 30",
             "max_iteration: %d" % cfg["max_iteration"], "track_precision: %r" % float(cfg["track_precision"]), "ransac_threshold: 1",
             "max_features_num: %d" % cfg["max_features_num"], "min_distance: %d" % cfg["min_distance"], "flag_equalize: %d" % cfg["flag_equalize"],
             "pub_frequency: %d" % cfg["pub_frequency"], "img_rate: 20", ""]
    return "\n".join(lines)


class RefImageProcessor:
    """larvio::ImageProcessor (src/image_processor.cpp) of the compiled reference, driven like the oracle's lvo.Frontend:
    process(img, ts, imu) -> (have, message as lvo.OBS records).  Behind cv::'s image algorithms stand the oracle's restatements (see
    oracle/ref_shim3/lvref_cv3.hpp): this object pins the ORCHESTRATION to the reference's text."""

    @classmethod
    def from_yaml(cls, path, cap=4096):
        """ImageProcessor(config_file) + initialize() on a configuration file as it is (e.g. the reference's own config/euroc.yaml)"""
        from . import lvo
        self = cls.__new__(cls); self._lvo = lvo; lvo.lib(); self._yaml = path; self.cap = cap
        self.h = _libip().lvref_imgproc_create(path.encode())
        if not self.h:
            raise RuntimeError("the reference's ImageProcessor::initialize() failed on " + path)
        return self

    FE_PARAMS = "width height pyramid_levels patch_size max_iteration track_precision max_features_num min_distance flag_equalize pub_frequency distortion_model".split()

    def params(self):
        """what ImageProcessor::loadParameters (image_processor.cpp:44-113) made of the file, by lvk_fe_config's field names"""
        L = _libip(); L.lvref_imgproc_params.argtypes = [C.c_void_p, C.c_void_p]
        o = np.zeros(30); L.lvref_imgproc_params(self.h, o.ctypes.data)
        d = dict(zip(self.FE_PARAMS, o[:11].tolist()))
        d.update(intrinsics=o[11:15].copy(), distortion=o[15:19].copy(), R_cam_imu=o[19:28].reshape(3, 3).copy(), ransac_threshold=float(o[28]), img_rate=float(o[29]))
        return d

    def __init__(self, cfg, workdir):
        from . import lvo
        self._lvo = lvo
        lvo.lib()                                                      # liblvo.so first: the library links against it
        os.makedirs(workdir, exist_ok=True)
        self._yaml = os.path.join(workdir, "lvref_imgproc.yaml")
        with open(self._yaml, "w") as f:
            f.write(imgproc_yaml(cfg, workdir))
        self.h = _libip().lvref_imgproc_create(self._yaml.encode())
        if not self.h:
            raise RuntimeError("the reference's ImageProcessor::initialize() failed on " + self._yaml)
        self.cap = max(4096, cfg["max_features_num"] * 4)

    def __del__(self):
        if getattr(self, "h", None):
            _libip().lvref_imgproc_destroy(self.h); self.h = None

    def process(self, img, ts, imu):
        img = np.ascontiguousarray(img, np.uint8)
        m = np.zeros((len(imu), 7)); m[:, 0] = imu["t"]; m[:, 1:4] = imu["gyro"]; m[:, 4:7] = imu["acc"]
        out = np.zeros((self.cap, 9)); n = C.c_int(0)
        have = _libip().lvref_imgproc_process(self.h, float(ts), img.ctypes.data, img.shape[1], img.shape[0], img.strides[0], len(m), m.ctypes.data,
                                              out.ctypes.data, self.cap, C.byref(n))
        msg = np.zeros(n.value, self._lvo.OBS)
        for k, name in enumerate(("id", "u", "v", "u_init", "v_init", "u_vel", "v_vel", "u_init_vel", "v_init_vel")):
            msg[name] = out[:n.value, k]
        return bool(have), msg

    @property
    def state(self):
        return _libip().lvref_imgproc_state(self.h)

    def tracks(self):
        cap = self.cap
        ids = np.zeros(cap, np.uint64); p = np.zeros((cap, 2), np.float32); life = np.zeros(cap, np.int32)
        ini = np.zeros((cap, 2), np.float32); desc = np.zeros((cap, 32), np.uint8)
        n = _libip().lvref_imgproc_tracks(self.h, ids.ctypes.data, p.ctypes.data, life.ctypes.data, ini.ctypes.data, desc.ctypes.data, cap)
        return dict(ids=ids[:n].copy(), pts=p[:n].copy(), lifetime=life[:n].copy(), init=ini[:n].copy(), desc=desc[:n].copy())

    def new_pts(self):
        p = np.zeros((self.cap, 2), np.float32)
        n = _libip().lvref_imgproc_new_pts(self.h, p.ctypes.data, self.cap)
        return p[:n].copy()


# ------------------------------------------------------------------------------------------------ the reference's moving-start initialiser
_SO_DYN = os.environ.get("LVREF_DYNINIT_SO", os.path.join(_HERE, "_ref", "liblvref_dyninit.so"))
_libd = None


def dyninit_available(build=True):
    if os.path.exists(_SO_DYN):
        return True
    if build and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return os.path.exists(_SO_DYN)


def _libdy():
    global _libd
    if _libd is None:
        if not dyninit_available():
            raise RuntimeError("oracle/_ref/liblvref_dyninit.so is missing and /root/reference is not here to build it from")
        from . import lvo
        lvo.lib()                                                      # liblvo.so first: the library links against it
        L = C.CDLL(_SO_DYN)
        vp, i, d = C.c_void_p, C.c_int, C.c_double
        L.lvref_dyninit_create.restype = vp; L.lvref_dyninit_create.argtypes = [d, vp, vp, d, d, d, d, d]
        L.lvref_dyninit_destroy.argtypes = [vp]
        L.lvref_dyninit_try.restype = i; L.lvref_dyninit_try.argtypes = [vp, d, i, vp, i, vp, vp]
        L.lvref_dyninit_frame_count.restype = i; L.lvref_dyninit_frame_count.argtypes = [vp]
        _libd = L
    return _libd


def dynamic_init(msgs, imu, R_b2c, t_c_b, td=0.0, imu_img_time_th=1.0 / 400, noise=(0.08, 4e-5, 0.004, 2e-6)):
    """larvio::DynamicInitializer of the compiled reference fed like oracle/dyn_init.dynamic_init: feature messages [(ts, OBS records)]
    and the IMU stream, as LarVio::processFeatures -> FlexibleInitializer::tryIncInit hands them over (the buffer only grows until the
    initialiser succeeds).  -> dict(message, state_time, q [x y z w], v, bg, g, erase, last_gyro, last_acc) of the first success, or None."""
    L = _libdy()
    R_c2b = np.ascontiguousarray(np.asarray(R_b2c, np.float64).T); t = np.ascontiguousarray(t_c_b, np.float64)
    h = L.lvref_dyninit_create(float(td), R_c2b.ctypes.data, t.ctypes.data, *[float(x) for x in noise], float(imu_img_time_th))
    try:
        sent = 0
        for mi, (ts, m) in enumerate(msgs):
            hi = int(np.searchsorted(imu["t"], ts + 0.05, side="left"))
            new = imu[sent:hi]; sent = hi
            a = np.zeros((len(new), 7)); a[:, 0] = new["t"]; a[:, 1:4] = new["gyro"]; a[:, 4:7] = new["acc"]
            f = np.zeros((len(m), 9))
            for k, name in enumerate(("id", "u", "v", "u_init", "v_init", "u_vel", "v_vel", "u_init_vel", "v_init_vel")):
                f[:, k] = m[name]
            out = np.zeros(21)
            if L.lvref_dyninit_try(h, float(ts), len(f), f.ctypes.data, len(a), a.ctypes.data, out.ctypes.data):
                return dict(message=mi, state_time=out[0], q=out[1:5].copy(), v=out[5:8].copy(), bg=out[8:11].copy(), last_gyro=out[11:14].copy(),
                            last_acc=out[14:17].copy(), g=out[17:20].copy(), erase=int(out[20]))
        return None
    finally:
        L.lvref_dyninit_destroy(h)
