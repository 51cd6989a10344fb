"""The driver step: one camera frame through ImageProcessor.processImage and, when it produced a message,
LarVio.processFeatures — the loop body of app/larvioMain.cpp:87-117 — as ONE C-ABI call (lvk_vio_process)."""
import ctypes as C
import numpy as np
from ._lib import lib, _p, IMU, Image, make_image

_done = False


def _L():
    global _done
    L = lib()
    if not _done:
        vp, i, d = C.c_void_p, C.c_int, C.c_double
        pi = C.POINTER(C.c_int)
        L.lvk_vio_process.argtypes = [vp, vp, C.POINTER(Image), d, vp, i, pi, pi, pi]
        L.lvk_vio_process.restype = i
        L.lvk_vio_process_deferred.argtypes = [vp, vp, C.POINTER(Image), d, vp, i, pi, pi, pi]
        L.lvk_vio_process_deferred.restype = i
        pl = C.POINTER(C.c_long)
        L.lvk_vio_pipe_create.argtypes = [vp, vp, C.POINTER(vp)]; L.lvk_vio_pipe_create.restype = i
        L.lvk_vio_pipe_destroy.argtypes = [vp]; L.lvk_vio_pipe_destroy.restype = None
        L.lvk_vio_pipe_push_imu.argtypes = [vp, vp, i]; L.lvk_vio_pipe_push_imu.restype = i
        L.lvk_vio_pipe_submit.argtypes = [vp, C.POINTER(Image), d, pi]; L.lvk_vio_pipe_submit.restype = i
        L.lvk_vio_pipe_drain.argtypes = [vp, pl, pl]; L.lvk_vio_pipe_drain.restype = i
        L.lvk_vio_pipe_stats.argtypes = [vp, C.POINTER(C.c_double), i]; L.lvk_vio_pipe_stats.restype = i
        L.lvk_vio_pipe_latency.argtypes = [vp, vp, i, pi, i]; L.lvk_vio_pipe_latency.restype = i
        L.lvk_vio_pipe_early_counts.argtypes = [vp, pl, pl]; L.lvk_vio_pipe_early_counts.restype = i
        _done = True
    return L


class VioDriver:
    """Keeps the driver's IMU buffer: samples with t < t_img + 0.05 are visible (larvioMain.cpp:98-102), the back-end erases
    what it consumed (larvio.cpp:511-512)."""

    def __init__(self, image_processor, larvio, imu_all):
        self.fe, self.be = image_processor, larvio
        self.imu = np.ascontiguousarray(imu_all, IMU)
        self.t = self.imu["t"].copy()
        self.lo = 0
        self._shape = (image_processor.config["height"], image_processor.config["width"])
        self._c = (C.c_int(0), C.c_int(0), C.c_int(0))
        self._base = self.imu.ctypes.data
        self._isz = self.imu.dtype.itemsize

    def visible_end(self, ts):
        return int(np.searchsorted(self.t, ts + 0.05, side="left"))

    def step(self, ts, hi, img=None, device_ptr=None, stride=None, shape=None):
        """hi = visible_end(ts) (precomputable).  Returns (has_msg, updated)."""
        used, has, upd = self._c
        im, keep = make_image(img, device_ptr, stride, shape if shape is not None else self._shape)
        st = _L().lvk_vio_process(self.fe._h, self.be._h, C.byref(im), float(ts), C.c_void_p(self._base + self.lo * self._isz), hi - self.lo,
                                  C.byref(used), C.byref(has), C.byref(upd))
        self.fe.ctx.check(st)
        self.lo += used.value
        return bool(has.value), bool(upd.value)


class VioDeferred(VioDriver):
    """The adapter's schedule under a blocking driver (lvk_vio_process_deferred): processImage waits for its message, processFeatures
    queues the update on the filter's worker thread and returns; LarVio getters (state(), cov(), ...) or the next step wait for it.
    Same results as VioDriver.  step() returns (has_msg, will_update)."""

    def step(self, ts, hi, img=None, device_ptr=None, stride=None, shape=None):
        used, has, upd = self._c
        im, keep = make_image(img, device_ptr, stride, shape if shape is not None else self._shape)
        st = _L().lvk_vio_process_deferred(self.fe._h, self.be._h, C.byref(im), float(ts), C.c_void_p(self._base + self.lo * self._isz), hi - self.lo,
                                           C.byref(used), C.byref(has), C.byref(upd))
        self.fe.ctx.check(st)
        self.lo += used.value
        return bool(has.value), bool(upd.value)


class VioPipeline:
    """The same loop with the filter update of frame k overlapping the front-end of frame k+1 (lvk_vio_pipe_*): the front-end
    and the filter live on different contexts (streams); results are identical to VioDriver's."""
    _close_order = 0            # closed before the front-end and the filter it points at

    def __init__(self, image_processor, larvio, imu_all):
        if image_processor.ctx is larvio.ctx:
            raise ValueError("VioPipeline needs the front-end and the filter on different Contexts")
        self.fe, self.be = image_processor, larvio
        self.imu = np.ascontiguousarray(imu_all, IMU)
        self.t = self.imu["t"].copy()
        self.pushed = 0
        self._shape = (image_processor.config["height"], image_processor.config["width"])
        self._has = C.c_int(0)
        self._base = self.imu.ctypes.data
        self._isz = self.imu.dtype.itemsize
        h = C.c_void_p()
        self.be.ctx.check(_L().lvk_vio_pipe_create(self.fe._h, self.be._h, C.byref(h)))
        self._h = h
        self.be.ctx.adopt(self); self.fe.ctx.adopt(self)

    def visible_end(self, ts):
        return int(np.searchsorted(self.t, ts + 0.05, side="left"))

    def step(self, ts, hi, img=None, device_ptr=None, stride=None, shape=None):
        """hi = visible_end(ts).  Returns has_msg; the update it triggers completes asynchronously (drain())."""
        L = _L()
        if hi > self.pushed:
            L.lvk_vio_pipe_push_imu(self._h, C.c_void_p(self._base + self.pushed * self._isz), hi - self.pushed)
            self.pushed = hi
        im, keep = make_image(img, device_ptr, stride, shape if shape is not None else self._shape)
        st = L.lvk_vio_pipe_submit(self._h, C.byref(im), float(ts), C.byref(self._has))
        if st != 0:
            self.fe.ctx.check(st); self.be.ctx.check(st)
        return bool(self._has.value)

    def drain(self):
        nu, nm = C.c_long(0), C.c_long(0)
        st = _L().lvk_vio_pipe_drain(self._h, C.byref(nu), C.byref(nm))
        if st != 0:
            self.be.ctx.check(st); self.fe.ctx.check(st)
        return nu.value, nm.value

    def stats(self, reset=False):
        """host wall microseconds since the last reset: front-end (caller thread), caller waiting, filter (worker), worker idle"""
        o = (C.c_double * 4)()
        _L().lvk_vio_pipe_stats(self._h, o, 1 if reset else 0)
        return dict(front_end_us=o[0], caller_wait_us=o[1], filter_us=o[2], worker_idle_us=o[3])

    def early_counts(self):
        """(erase counts taken early by submit, how many of them the filter's thread found different): lvk_vio_pipe_early_counts"""
        a, b = C.c_long(0), C.c_long(0)
        _L().lvk_vio_pipe_early_counts(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def latencies(self, reset=True, cap=1 << 20):
        """image-in -> state-out latency [us] of every message-carrying frame since the last reset (drain() first)"""
        out = np.empty(cap, np.float32); n = C.c_int(0)
        _L().lvk_vio_pipe_latency(self._h, _p(out), cap, C.byref(n), 1 if reset else 0)
        return out[:n.value].copy()

    def close(self):
        if self._h:
            _L().lvk_vio_pipe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
