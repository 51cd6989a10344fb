"""ImageProcessor — host-side mirror of larvio::ImageProcessor over the C ABI.

Same surface as the reference class (include/larvio/image_processor.h:36-68): construct with the
configuration, ``initialize()``, then ``processImage(img, imu_buffer)`` per frame; the boolean it returns is
the reference's ``haveFeatures`` and the feature list is the ``MonoCameraMeasurement`` it fills
(include/larvio/feature_msg.h:15-52).  All arithmetic happens in liblvk_hip.so on the GPU.
"""
import ctypes as C
import os
import numpy as np
from ._lib import lib, _p, Context, FeConfig, LvkError, IMU, OBS, make_image


class MonoCameraMeasurement:
    """feature_msg.h:47-52"""

    def __init__(self, time_stamp, features):
        self.timeStampToSec = time_stamp
        self.features = features          # structured array, dtype OBS (id,u,v,u_init,v_init,u_vel,v_vel,u_init_vel,v_init_vel)


def make_fe_config(config):
    """lvk_fe_config from the dict of ImageProcessor parameters (larvio_amd.synthetic.frontend_config)"""
    c = FeConfig()
    for k in ("width", "height", "pyramid_levels", "patch_size", "max_iteration", "track_precision",
              "max_features_num", "min_distance", "flag_equalize", "pub_frequency", "distortion_model"):
        setattr(c, k, config[k])
    c.intrinsics = (C.c_double * 4)(*config["intrinsics"])
    c.distortion = (C.c_double * 4)(*config["distortion"])
    c.R_cam_imu = (C.c_double * 9)(*np.asarray(config["R_cam_imu"], np.float64).reshape(9))
    return c


class ImageProcessor:
    FIRST_IMAGE, SECOND_IMAGE, OTHER_IMAGES = 1, 2, 3

    def __init__(self, config, ctx=None):
        """config: dict with the keys ImageProcessor::loadParameters reads (image_processor.cpp:44-113; see
        larvio_amd.synthetic.frontend_config), or the path of a LARVIO configuration file."""
        if isinstance(config, (str, os.PathLike)):          # the reference's constructor argument: the YAML path
            from .config import load_config
            config = load_config(config)[0]
        self.config = dict(config)
        self.ctx = ctx
        self._h = None

    def initialize(self):
        if self.ctx is None:
            self.ctx = Context()
        c = make_fe_config(self.config)
        h = C.c_void_p()
        st = lib().lvk_frontend_create(self.ctx.h, C.byref(c), C.byref(h))
        if st != 0:
            print("lvk_frontend_create failed:", lib().lvk_last_error(self.ctx.h).decode())
            return False
        self._h = h
        self.ctx.adopt(self)
        self._cap = self.config["max_features_num"]
        self._out = np.zeros(self._cap, OBS)
        return True

    def processImage(self, img, imu_msg_buffer, ts=None, device_ptr=None, stride=None, shape=None):
        """img: (H,W) uint8 array (host; its shape is checked against the configured resolution, as a cv::Mat's would be) — or pass
        device_ptr/stride (and shape=(H,W) if it is not the configured one) for an image already in HBM.
        imu_msg_buffer: structured array (t, gyro[3], acc[3]).  Returns (haveFeatures, MonoCameraMeasurement|None)."""
        if self._h is None:
            raise LvkError("ImageProcessor.initialize() has not succeeded")
        imu = np.ascontiguousarray(imu_msg_buffer, IMU)
        n_out, has = C.c_int(0), C.c_int(0)
        im, keep = make_image(img, device_ptr, stride, shape if shape is not None else (self.config["height"], self.config["width"]))
        st = lib().lvk_frontend_process(self._h, C.byref(im), float(ts), _p(imu), len(imu), _p(self._out), self._cap,
                                        C.byref(n_out), C.byref(has))
        self.ctx.check(st)
        if not has.value:
            return False, None
        return True, MonoCameraMeasurement(float(ts), self._out[:n_out.value].copy())

    # ---- introspection used by the parity tests
    def tracks(self):
        cap = self._cap
        ids = np.empty(cap, np.uint64); p = np.empty((cap, 2), np.float32); life = np.empty(cap, np.int32)
        ini = np.empty((cap, 2), np.float32); desc = np.empty((cap, 32), np.uint8); n = C.c_int(0)
        self.ctx.check(lib().lvk_frontend_tracks(self._h, _p(ids), _p(p), _p(life), _p(ini), _p(desc), cap, C.byref(n)))
        n = n.value
        return dict(ids=ids[:n].copy(), pts=p[:n].copy(), lifetime=life[:n].copy(), init=ini[:n].copy(), desc=desc[:n].copy())

    def new_pts(self):
        p = np.empty((self._cap, 2), np.float32); n = C.c_int(0)
        self.ctx.check(lib().lvk_frontend_new_pts(self._h, _p(p), self._cap, C.byref(n)))
        return p[:n.value].copy()

    @property
    def state(self):
        return lib().lvk_frontend_state(self._h)

    def lk_stats(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self.ctx.check(lib().lvk_frontend_lk_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def msg_stats(self):
        """-> (messages published, features they carried in total) since the front-end was created"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self.ctx.check(lib().lvk_frontend_msg_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def profile_enable(self, stage_mask=0x1FF):
        self.ctx.check(lib().lvk_frontend_profile_enable(self._h, stage_mask))

    def profile_read(self, reset=True):
        """-> {stage_name: (gpu_ms_sum, launches)} from HIP events on the context stream"""
        from ._lib import FE_STAGES
        ms = np.zeros(FE_STAGES, np.float64); n = np.zeros(FE_STAGES, np.uint64)
        self.ctx.check(lib().lvk_frontend_profile_read(self._h, _p(ms), _p(n), 1 if reset else 0))
        return {lib().lvk_frontend_stage_name(i).decode(): (float(ms[i]), int(n[i])) for i in range(FE_STAGES)}

    def close(self):
        if self._h:
            lib().lvk_frontend_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
