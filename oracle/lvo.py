"""ctypes binding of the CPU oracle (oracle/liblvo.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/lvo.h).  The product package larvio_amd never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PT = np.dtype([("x", np.float32), ("y", np.float32)])
IMU = np.dtype([("t", np.float64), ("gyro", np.float64, 3), ("acc", np.float64, 3)])
OBS = np.dtype([("id", np.uint64), ("u", np.float64), ("v", np.float64), ("u_init", np.float64),
                ("v_init", np.float64), ("u_vel", np.float64), ("v_vel", np.float64),
                ("u_init_vel", np.float64), ("v_init_vel", np.float64)])
MAX_LEVELS = 8


class Pyramid(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("pad", C.c_int),
                ("w", C.c_int * MAX_LEVELS), ("h", C.c_int * MAX_LEVELS),
                ("istride", C.c_int * MAX_LEVELS), ("dstride", C.c_int * MAX_LEVELS),
                ("img", C.c_void_p * MAX_LEVELS), ("der", C.c_void_p * MAX_LEVELS)]

    def image(self, l, padded=False):
        p = self.pad
        H, W, s = self.h[l] + 2 * p, self.w[l] + 2 * p, self.istride[l]
        a = np.ctypeslib.as_array(C.cast(self.img[l], C.POINTER(C.c_uint8)), shape=(H, s))[:, :W]
        return a if padded else a[p:p + self.h[l], p:p + self.w[l]]

    def deriv(self, l, padded=False):
        p = self.pad
        H, W, s = self.h[l] + 2 * p, self.w[l] + 2 * p, self.dstride[l]
        a = np.ctypeslib.as_array(C.cast(self.der[l], C.POINTER(C.c_int16)), shape=(H, s))[:, :2 * W]
        a = a.reshape(H, W, 2)
        return a if padded else a[p:p + self.h[l], p:p + self.w[l]]


class FeConfig(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("pyramid_levels", C.c_int), ("patch_size", C.c_int),
                ("max_iteration", C.c_int), ("track_precision", C.c_double), ("max_features_num", C.c_int),
                ("min_distance", C.c_int), ("flag_equalize", C.c_int), ("pub_frequency", C.c_int),
                ("distortion_model", C.c_int), ("intrinsics", C.c_double * 4), ("distortion", C.c_double * 4),
                ("R_cam_imu", C.c_double * 9)]


def build(force=False):
    so = os.path.join(_HERE, "liblvo.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h", ".inc"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liblvo.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        vp, i, d = C.c_void_p, C.c_int, C.c_double
        L.lvo_clahe_u8.argtypes = [vp, i, i, i, vp, i, d, i, i]
        L.lvo_pyr_down_u8.argtypes = [vp, i, i, i, vp, i]
        L.lvo_scharr_deriv.argtypes = [vp, i, i, i, vp, i]
        L.lvo_pyramid_build.argtypes = [vp, i, i, i, i, i, C.POINTER(Pyramid)]
        L.lvo_pyramid_free.argtypes = [C.POINTER(Pyramid)]
        L.lvo_orb_prepare.argtypes = [C.POINTER(Pyramid), vp, vp]
        L.lvo_good_features.argtypes = [C.POINTER(Pyramid), vp, i, d, d, vp, i]
        L.lvo_good_features.restype = i
        L.lvo_min_eigen_map.argtypes = [C.POINTER(Pyramid), vp]
        L.lvo_lk_track.argtypes = [C.POINTER(Pyramid), C.POINTER(Pyramid), vp, vp, vp, i, i, d, vp]
        L.lvo_orb_describe.argtypes = [vp, vp, i, i, vp, i, vp, vp]
        L.lvo_hamming256.argtypes = [vp, vp]
        L.lvo_hamming256.restype = i
        L.lvo_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.lvo_fast_atan2.restype = C.c_float
        L.lvo_undistort_points.argtypes = [vp, i, vp, i, vp, vp, vp]
        L.lvo_find_fundamental_mask.argtypes = [vp, vp, i, d, d, vp]
        L.lvo_find_fundamental_mask.restype = i
        L.lvo_find_fundamental.argtypes = [vp, vp, i, d, d, vp, vp]
        L.lvo_find_fundamental.restype = i
        L.lvo_ransac_fundamental.argtypes = [vp, vp, i, d, d, i, vp, vp]
        L.lvo_ransac_fundamental.restype = i
        L.lvo_fundamental_7pt.argtypes = [vp, vp, vp]
        L.lvo_fundamental_7pt.restype = i
        L.lvo_predict_homography.argtypes = [vp, i, d, d, vp, vp, vp]
        L.lvo_apply_homography.argtypes = [vp, vp, i, vp]
        L.lvo_frontend_create.argtypes = [C.POINTER(FeConfig)]
        L.lvo_frontend_create.restype = vp
        L.lvo_frontend_destroy.argtypes = [vp]
        L.lvo_frontend_process.argtypes = [vp, vp, i, d, vp, i, vp, i, C.POINTER(i)]
        L.lvo_frontend_process.restype = i
        L.lvo_frontend_tracks.argtypes = [vp, vp, vp, vp, vp, vp, i]
        L.lvo_frontend_tracks.restype = i
        L.lvo_frontend_new_pts.argtypes = [vp, vp, i]
        L.lvo_frontend_new_pts.restype = i
        L.lvo_frontend_state.argtypes = [vp]
        L.lvo_frontend_state.restype = i
        L.lvo_frontend_lk_stats.argtypes = [vp, vp, vp]
        L.lvo_set_threads.argtypes = [i]; L.lvo_set_threads.restype = None
        L.lvo_get_threads.argtypes = []; L.lvo_get_threads.restype = i
        _LIB = L
    return _LIB


def set_threads(n):
    """host threads for the oracle's loops with independent iterations (bench.py's all-core cpu_baseline leg); results do not
    depend on the count.  1 = the reference's single-threaded design (the default)."""
    lib().lvo_set_threads(int(n))


def set_lk_float_accum(on):
    """sensitivity probe: LK's A11/A12/A22/b1/b2 as float32 running sums in row-major order (OpenCV's scalar LKTrackerInvoker) instead of
    the exact integer sums.  Off in every parity test."""
    lib().lvo_set_lk_float_accum(int(bool(on)))


def get_threads():
    return lib().lvo_get_threads()


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 2
    return img


def pts(a):
    """(n,2) float array -> contiguous float32 (n,2) (layout == lvo_pt2f[n])."""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1, 2))


def clahe(img, clip=3.0, tiles=(8, 8)):
    img = _u8(img); h, w = img.shape
    out = np.empty_like(img)
    lib().lvo_clahe_u8(_p(img), w, h, w, _p(out), w, clip, tiles[0], tiles[1])
    return out


def pyr_down(img):
    img = _u8(img); h, w = img.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().lvo_pyr_down_u8(_p(img), w, h, w, _p(out), out.shape[1])
    return out


def scharr(img):
    img = _u8(img); h, w = img.shape
    out = np.empty((h, w, 2), np.int16)
    lib().lvo_scharr_deriv(_p(img), w, h, w, _p(out), 2 * w)
    return out


class LkPyramid:
    def __init__(self, img, win=21, max_level=2):
        img = _u8(img); h, w = img.shape
        self.p = Pyramid()
        lib().lvo_pyramid_build(_p(img), w, h, w, win, max_level, C.byref(self.p))
        self.w, self.h = w, h

    def __del__(self):
        try:
            lib().lvo_pyramid_free(C.byref(self.p))
        except Exception:
            pass

    @property
    def n_levels(self):
        return self.p.n_levels

    def image(self, l, padded=False):
        return self.p.image(l, padded)

    def deriv(self, l, padded=False):
        return self.p.deriv(l, padded)

    def orb_prepare(self):
        ext = np.empty((self.h + 64, self.w + 64), np.uint8); blur = np.empty_like(ext)
        lib().lvo_orb_prepare(C.byref(self.p), _p(ext), _p(blur))
        return ext, blur

    def min_eigen_map(self):
        e = np.empty((self.h, self.w), np.float32)
        lib().lvo_min_eigen_map(C.byref(self.p), _p(e))
        return e

    def good_features(self, max_corners, quality=0.01, min_distance=20.0, mask=None):
        cap = max_corners if max_corners > 0 else self.w * self.h
        out = np.empty((cap, 2), np.float32)
        m = _u8(mask) if mask is not None else None
        n = lib().lvo_good_features(C.byref(self.p), _p(m), max_corners, quality, min_distance, _p(out), cap)
        return out[:n].copy()


def lk_track(prev, nxt, prev_pts, init_pts, max_iter=30, eps=0.01):
    p0 = pts(prev_pts); p1 = pts(init_pts).copy(); n = len(p0)
    st = np.empty(n, np.uint8)
    nl = min(prev.n_levels, nxt.n_levels)
    it = np.zeros((n, nl), np.int32)
    lib().lvo_lk_track(C.byref(prev.p), C.byref(nxt.p), _p(p0), _p(p1), _p(st), n, max_iter, eps, _p(it))
    return p1, st, it


def orb_describe(ext, blur, points):
    p = pts(points); n = len(p)
    h, w = ext.shape[0] - 64, ext.shape[1] - 64
    desc = np.empty((n, 32), np.uint8); ang = np.empty(n, np.float32)
    lib().lvo_orb_describe(_p(ext), _p(blur), w, h, _p(p), n, _p(desc), _p(ang))
    return desc, ang


def hamming(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().lvo_hamming256(_p(a), _p(b))


def undistort(points, intr, model, dist, new_intr):
    p = pts(points); out = np.empty_like(p)
    a = np.asarray(intr, np.float64); d = np.asarray(dist, np.float64); k = np.asarray(new_intr, np.float64)
    lib().lvo_undistort_points(_p(p), len(p), _p(a), model, _p(d), _p(k), _p(out))
    return out


def find_fundamental_mask(p1, p2, thresh=1.0, conf=0.99):
    a = pts(p1); b = pts(p2); n = len(a)
    mask = np.zeros(max(n, 1), np.uint8)
    wrote = lib().lvo_find_fundamental_mask(_p(a), _p(b), n, thresh, conf, _p(mask))
    return (mask[:n] if wrote else None)


def find_fundamental(p1, p2, thresh=1.0, conf=0.99):
    """-> (mask or None, F 3x3): what cv::findFundamentalMat(p1, p2, FM_RANSAC, thresh, conf, mask) returns (F = zeros: the empty Mat)"""
    a = pts(p1); b = pts(p2); n = len(a)
    mask = np.zeros(max(n, 1), np.uint8); F = np.zeros(9, np.float64)
    wrote = lib().lvo_find_fundamental(_p(a), _p(b), n, thresh, conf, _p(mask), _p(F))
    return (mask[:n] if wrote else None), F.reshape(3, 3)


def ransac_fundamental(p1, p2, thresh=1.0, conf=0.99, max_iters=1000):
    a = pts(p1); b = pts(p2); n = len(a)
    mask = np.zeros(n, np.uint8); it = C.c_int(0)
    ok = lib().lvo_ransac_fundamental(_p(a), _p(b), n, thresh, conf, max_iters, _p(mask), C.byref(it))
    return ok, mask, it.value


def fundamental_7pt(m1, m2):
    a = pts(m1); b = pts(m2)
    F = np.zeros((3, 9), np.float64)
    n = lib().lvo_fundamental_7pt(_p(a), _p(b), _p(F))
    return F[:n].reshape(n, 3, 3)


def predict_homography(imu, t_prev, t_curr, R_cam_imu, intr):
    imu = np.ascontiguousarray(imu, IMU)
    R = np.ascontiguousarray(R_cam_imu, np.float64); k = np.asarray(intr, np.float64)
    H = np.empty(9, np.float32)
    lib().lvo_predict_homography(_p(imu), len(imu), t_prev, t_curr, _p(R), _p(k), _p(H))
    return H.reshape(3, 3)


def apply_homography(H, points):
    p = pts(points); out = np.empty_like(p)
    Hc = np.ascontiguousarray(H, np.float32)
    lib().lvo_apply_homography(_p(Hc), _p(p), len(p), _p(out))
    return out


def make_fe_config(cfg):
    c = FeConfig()
    for k in ("width", "height", "pyramid_levels", "patch_size", "max_iteration", "track_precision",
              "max_features_num", "min_distance", "flag_equalize", "pub_frequency", "distortion_model"):
        setattr(c, k, cfg[k])
    c.intrinsics = (C.c_double * 4)(*cfg["intrinsics"])
    c.distortion = (C.c_double * 4)(*cfg["distortion"])
    c.R_cam_imu = (C.c_double * 9)(*np.asarray(cfg["R_cam_imu"], np.float64).reshape(9))
    return c


class Frontend:
    """lvo_frontend: the oracle's ImageProcessor (image_processor.cpp:130-219)."""

    def __init__(self, cfg):
        self.cfg = dict(cfg)
        self._c = make_fe_config(cfg)
        self.h = lib().lvo_frontend_create(C.byref(self._c))
        self.cap = max(4096, cfg["max_features_num"] * 4)

    def __del__(self):
        try:
            lib().lvo_frontend_destroy(self.h)
        except Exception:
            pass

    def process(self, img, ts, imu):
        img = _u8(img)
        imu = np.ascontiguousarray(imu, IMU)
        out = np.zeros(self.cap, OBS); n = C.c_int(0)
        have = lib().lvo_frontend_process(self.h, _p(img), img.shape[1], ts, _p(imu), len(imu), _p(out), self.cap, C.byref(n))
        return bool(have), out[:n.value].copy()

    def tracks(self):
        cap = self.cap
        ids = np.empty(cap, np.uint64); p = np.empty((cap, 2), np.float32); life = np.empty(cap, np.int32)
        ini = np.empty((cap, 2), np.float32); desc = np.empty((cap, 32), np.uint8)
        n = lib().lvo_frontend_tracks(self.h, _p(ids), _p(p), _p(life), _p(ini), _p(desc), cap)
        return dict(ids=ids[:n].copy(), pts=p[:n].copy(), lifetime=life[:n].copy(), init=ini[:n].copy(), desc=desc[:n].copy())

    def new_pts(self):
        p = np.empty((self.cap, 2), np.float32)
        n = lib().lvo_frontend_new_pts(self.h, _p(p), self.cap)
        return p[:n].copy()

    @property
    def state(self):
        return lib().lvo_frontend_state(self.h)

    def lk_stats(self):
        a = C.c_uint64(0); b = C.c_uint64(0)
        lib().lvo_frontend_lk_stats(self.h, C.byref(a), C.byref(b))
        return a.value, b.value
