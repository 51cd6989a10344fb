"""The driver step: one camera frame through ImageProcessor.processImage and, when it produced a message,
LarVio.processFeatures — the loop body of app/larvioMain.cpp:87-117 — as ONE C-ABI call (lvk_vio_process)."""
import ctypes as C
import numpy as np
from ._lib import lib, _p, IMU

_done = False


def _L():
    global _done
    L = lib()
    if not _done:
        vp, i, d = C.c_void_p, C.c_int, C.c_double
        pi = C.POINTER(C.c_int)
        L.lvk_vio_process.argtypes = [vp, vp, vp, i, i, d, vp, i, pi, pi, pi]
        L.lvk_vio_process.restype = i
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
        self._c = (C.c_int(0), C.c_int(0), C.c_int(0))
        self._base = self.imu.ctypes.data
        self._isz = self.imu.dtype.itemsize

    def visible_end(self, ts):
        return int(np.searchsorted(self.t, ts + 0.05, side="left"))

    def step(self, ts, hi, img=None, device_ptr=None, stride=None):
        """hi = visible_end(ts) (precomputable).  Returns (has_msg, updated)."""
        used, has, upd = self._c
        if device_ptr is not None:
            ptr, s, is_dev = C.c_void_p(device_ptr), stride, 1
        else:
            img = np.ascontiguousarray(img, np.uint8)
            ptr, s, is_dev = _p(img), img.shape[1], 0
        st = _L().lvk_vio_process(self.fe._h, self.be._h, ptr, s, is_dev, float(ts), C.c_void_p(self._base + self.lo * self._isz), hi - self.lo,
                                  C.byref(used), C.byref(has), C.byref(upd))
        self.fe.ctx.check(st)
        self.lo += used.value
        return bool(has.value), bool(upd.value)
