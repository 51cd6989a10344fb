"""Stage-level host wrappers over the C ABI (numpy in / numpy out; data staged through HBM).

These mirror, call for call, the OpenCV / ORBdescriptor functions the reference's front-end invokes
(see include/lvk_c.h for the file:line of each), so the parity tests read like the reference's code.
"""
import ctypes as C
import numpy as np
from ._lib import lib, _p, Context, LvkError, IMU  # noqa: F401


def _u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 2
    return img


def _pts(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1, 2))


def clahe(ctx, img, clip=3.0, tiles=(8, 8)):
    img = _u8(img); h, w = img.shape
    d_src = ctx.to_device(img); d_dst = ctx.alloc(img.nbytes)
    ctx.check(lib().lvk_clahe_u8(ctx.h, _p(d_src), w, h, w, _p(d_dst), w, clip, tiles[0], tiles[1]))
    return ctx.to_host(d_dst, np.uint8, (h, w))


class Pyramid:
    """lvk_pyramid: cv::buildOpticalFlowPyramid(img, win, max_level, withDerivatives=True) on the GPU."""

    def __init__(self, ctx, w, h, win=21, max_level=2):
        self.ctx, self.w, self.h, self.win = ctx, w, h, win
        p = C.c_void_p()
        ctx.check(lib().lvk_pyramid_create(ctx.h, w, h, win, max_level, C.byref(p)))
        self.h_ = p

    def build(self, img, clahe=False, clip=3.0, tiles=(8, 8)):
        img = _u8(img)
        assert img.shape == (self.h, self.w)
        d = self.ctx.to_device(img)
        self.build_device(d, self.w, clahe, clip, tiles)
        self.ctx.sync()
        return self

    def build_device(self, d_img, stride, clahe=False, clip=3.0, tiles=(8, 8)):
        if clahe:
            self.ctx.check(lib().lvk_pyramid_build_clahe(self.ctx.h, self.h_, _p(d_img), stride, clip, tiles[0], tiles[1]))
        else:
            self.ctx.check(lib().lvk_pyramid_build(self.ctx.h, self.h_, _p(d_img), stride))

    @property
    def n_levels(self):
        return lib().lvk_pyramid_levels(self.h_)

    def _level(self, l):
        w, h, pad, ist, dst = (C.c_int() for _ in range(5))
        pi, pd = C.c_void_p(), C.c_void_p()
        st = lib().lvk_pyramid_level(self.h_, l, C.byref(w), C.byref(h), C.byref(pad), C.byref(ist), C.byref(dst), C.byref(pi), C.byref(pd))
        if st != 0:
            raise LvkError("bad level")
        return w.value, h.value, pad.value, ist.value, dst.value, pi.value, pd.value

    def image(self, l, padded=False):
        w, h, pad, ist, _, pi, _ = self._level(l)
        a = self.ctx.to_host(pi, np.uint8, (h + 2 * pad, ist))[:, :w + 2 * pad]
        return a if padded else a[pad:pad + h, pad:pad + w]

    def deriv(self, l, padded=False):
        w, h, pad, _, dst, _, pd = self._level(l)
        a = self.ctx.to_host(pd, np.int16, (h + 2 * pad, dst))[:, :2 * (w + 2 * pad)].reshape(h + 2 * pad, w + 2 * pad, 2)
        return a if padded else a[pad:pad + h, pad:pad + w]

    def orb_prepare(self):
        n = (self.h + 64) * (self.w + 64)
        self.d_ext = self.ctx.alloc(n); self.d_blur = self.ctx.alloc(n)
        self.ctx.check(lib().lvk_orb_prepare(self.ctx.h, self.h_, _p(self.d_ext), _p(self.d_blur)))
        return (self.ctx.to_host(self.d_ext, np.uint8, (self.h + 64, self.w + 64)),
                self.ctx.to_host(self.d_blur, np.uint8, (self.h + 64, self.w + 64)))

    def min_eigen_map(self):
        d = self.ctx.alloc(4 * self.w * self.h)
        self.ctx.check(lib().lvk_min_eigen_map(self.ctx.h, self.h_, _p(d)))
        return self.ctx.to_host(d, np.float32, (self.h, self.w))

    def good_features(self, max_corners, quality=0.01, min_distance=20.0, mask=None):
        cap = max_corners
        d_out = self.ctx.alloc(8 * cap); d_n = self.ctx.alloc(4)
        d_mask = self.ctx.to_device(_u8(mask)) if mask is not None else None
        self.ctx.check(lib().lvk_good_features(self.ctx.h, self.h_, _p(d_mask), max_corners, quality, min_distance, _p(d_out), cap, _p(d_n)))
        n = int(self.ctx.to_host(d_n, np.int32, (1,))[0])
        return self.ctx.to_host(d_out, np.float32, (cap, 2))[:n].copy()

    def close(self):
        if self.h_:
            self.ctx.sync()
            lib().lvk_pyramid_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def lk_track(ctx, prev, nxt, prev_pts, init_pts, max_iter=30, eps=0.01):
    p0 = _pts(prev_pts); p1 = _pts(init_pts); n = len(p0)
    nl = min(prev.n_levels, nxt.n_levels)
    d0 = ctx.to_device(p0); d1 = ctx.to_device(p1); ds = ctx.alloc(max(n, 1)); di = ctx.alloc(4 * max(n, 1) * nl)
    ctx.check(lib().lvk_lk_track(ctx.h, prev.h_, nxt.h_, _p(d0), _p(d1), _p(ds), n, max_iter, eps, _p(di)))
    return ctx.to_host(d1, np.float32, (n, 2)), ctx.to_host(ds, np.uint8, (n,)), ctx.to_host(di, np.int32, (n, nl))


def orb_describe(ctx, pyr, points):
    """pyr.orb_prepare() must have been called."""
    p = _pts(points); n = len(p)
    dp = ctx.to_device(p); dd = ctx.alloc(32 * max(n, 1)); da = ctx.alloc(4 * max(n, 1))
    ctx.check(lib().lvk_orb_describe(ctx.h, _p(pyr.d_ext), _p(pyr.d_blur), pyr.w, pyr.h, _p(dp), n, _p(dd), _p(da)))
    return ctx.to_host(dd, np.uint8, (n, 32)), ctx.to_host(da, np.float32, (n,))


def hamming_rows(ctx, a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8); n = len(a)
    da = ctx.to_device(a); db = ctx.to_device(b); dd = ctx.alloc(4 * max(n, 1))
    ctx.check(lib().lvk_hamming256_rows(ctx.h, _p(da), _p(db), n, _p(dd)))
    return ctx.to_host(dd, np.int32, (n,))


def undistort(ctx, points, intr, model, dist, new_intr):
    p = _pts(points); n = len(p)
    a = np.asarray(intr, np.float64); d = np.asarray(dist, np.float64); k = np.asarray(new_intr, np.float64)
    dp = ctx.to_device(p); do = ctx.alloc(8 * max(n, 1))
    ctx.check(lib().lvk_undistort_points(ctx.h, _p(dp), n, _p(a), model, _p(d), _p(k), _p(do)))
    return ctx.to_host(do, np.float32, (n, 2))


def find_fundamental_mask(ctx, p1, p2, thresh=1.0, conf=0.99):
    a = _pts(p1); b = _pts(p2); n = len(a)
    da = ctx.to_device(a); db = ctx.to_device(b); dm = ctx.alloc(max(n, 1)); di = ctx.alloc(8)
    ctx.check(lib().lvk_find_fundamental_mask(ctx.h, _p(da), _p(db), n, thresh, conf, _p(dm), _p(di)))
    info = ctx.to_host(di, np.int32, (2,))
    return (ctx.to_host(dm, np.uint8, (n,)) if info[0] else None), int(info[1])


def find_fundamental(ctx, p1, p2, thresh=1.0, conf=0.99):
    """lvk_find_fundamental: -> (mask or None, F 3x3, hypotheses drawn); F = the matrix cv::findFundamentalMat returns (zeros: the empty Mat)"""
    a = _pts(p1); b = _pts(p2); n = len(a)
    da = ctx.to_device(a); db = ctx.to_device(b); dm = ctx.alloc(max(n, 1)); di = ctx.alloc(8); dF = ctx.alloc(72)
    ctx.check(lib().lvk_find_fundamental(ctx.h, _p(da), _p(db), n, thresh, conf, _p(dm), _p(di), _p(dF)))
    info = ctx.to_host(di, np.int32, (2,))
    return (ctx.to_host(dm, np.uint8, (n,)) if info[0] else None), ctx.to_host(dF, np.float64, (3, 3)), int(info[1])


def ransac_fundamental(ctx, p1, p2, thresh=1.0, conf=0.99, max_iters=1000):
    a = _pts(p1); b = _pts(p2); n = len(a)
    da = ctx.to_device(a); db = ctx.to_device(b); dm = ctx.alloc(max(n, 1)); di = ctx.alloc(8)
    ctx.check(lib().lvk_ransac_fundamental(ctx.h, _p(da), _p(db), n, thresh, conf, max_iters, _p(dm), _p(di)))
    info = ctx.to_host(di, np.int32, (2,))
    return ctx.to_host(dm, np.uint8, (n,)), int(info[1])


def predict_homography(imu, t_prev, t_curr, R_cam_imu, intr):
    imu = np.ascontiguousarray(imu, IMU)
    R = np.ascontiguousarray(R_cam_imu, np.float64); k = np.asarray(intr, np.float64)
    H = np.empty(9, np.float32)
    st = lib().lvk_predict_homography(_p(imu), len(imu), t_prev, t_curr, _p(R), _p(k), _p(H))
    if st != 0:
        raise LvkError("lvk_predict_homography")
    return H.reshape(3, 3)
