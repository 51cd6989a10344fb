"""ctypes binding of liblvk_hip.so (the C ABI declared in include/lvk_c.h)."""
import ctypes as C
import weakref
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("LVK_LIB", os.path.join(_HERE, "liblvk_hip.so"))
_LIB = None

PT = np.dtype([("x", np.float32), ("y", np.float32)])
IMU = np.dtype([("t", np.float64), ("gyro", np.float64, 3), ("acc", np.float64, 3)])
OBS = np.dtype([("id", np.uint64), ("u", np.float64), ("v", np.float64), ("u_init", np.float64),
                ("v_init", np.float64), ("u_vel", np.float64), ("v_vel", np.float64),
                ("u_init_vel", np.float64), ("v_init_vel", np.float64)])

# every symbol include/lvk_c.h declares (checked by tests/test_abi.py against the header text)
ABI_SYMBOLS = [
    "lvk_runtime_env", "lvk_context_create", "lvk_context_destroy", "lvk_context_set_stream", "lvk_context_get_stream", "lvk_sync", "lvk_last_error", "lvk_version",
    "lvk_malloc", "lvk_free", "lvk_memcpy_h2d", "lvk_memcpy_d2h", "lvk_memset",
    "lvk_clahe_u8", "lvk_pyramid_create", "lvk_pyramid_destroy", "lvk_pyramid_build", "lvk_pyramid_build_clahe",
    "lvk_pyramid_levels", "lvk_pyramid_level", "lvk_orb_prepare", "lvk_min_eigen_map", "lvk_good_features",
    "lvk_lk_track", "lvk_orb_describe", "lvk_hamming256_rows", "lvk_undistort_points", "lvk_find_fundamental_mask", "lvk_find_fundamental",
    "lvk_ransac_fundamental", "lvk_predict_homography",
    "lvk_frontend_create", "lvk_frontend_destroy", "lvk_frontend_process", "lvk_frontend_tracks", "lvk_frontend_new_pts",
    "lvk_frontend_state", "lvk_frontend_lk_stats", "lvk_frontend_msg_stats", "lvk_frontend_profile_enable", "lvk_frontend_profile_read",
    "lvk_frontend_stage_name",
    "lvk_ekf_compress_qr", "lvk_ekf_compress_qr_groups", "lvk_ekf_qr_plan", "lvk_ekf_update", "lvk_dgemm", "lvk_ekf_create", "lvk_ekf_destroy", "lvk_ekf_process", "lvk_ekf_process_async", "lvk_ekf_wait", "lvk_ekf_set_state",
    "lvk_ekf_dim", "lvk_ekf_is_initialized", "lvk_ekf_take_off_stamp", "lvk_ekf_get_state", "lvk_ekf_get_imu_intrinsics", "lvk_ekf_set_imu_intrinsics", "lvk_ekf_get_cov", "lvk_ekf_get_cov_imu", "lvk_ekf_get_clones", "lvk_ekf_get_features", "lvk_ekf_take_lost_features",
    "lvk_ekf_counters", "lvk_ekf_init_report", "lvk_ekf_profile", "lvk_ekf_profile_qr", "lvk_ekf_set_shard", "lvk_ekf_shard_stats", "lvk_shard_unique_id", "lvk_shard_comm_create", "lvk_shard_comm_destroy", "lvk_shard_comm_error", "lvk_shard_rccl_path", "lvk_shard_allgather_rccl", "lvk_triangulate", "lvk_ekf_gate_and_stack", "lvk_vio_process", "lvk_vio_process_deferred", "lvk_vio_pipe_create", "lvk_vio_pipe_destroy", "lvk_vio_pipe_push_imu", "lvk_vio_pipe_submit", "lvk_vio_pipe_drain", "lvk_vio_pipe_on_update", "lvk_vio_pipe_stats", "lvk_vio_pipe_latency", "lvk_vio_pipe_early_counts",
]
FE_STAGES = 9


class LvkError(RuntimeError):
    pass


class Image(C.Structure):
    """lvk_image: pointer + the size and step a cv::Mat carries (include/lvk_c.h)"""
    _fields_ = [("data", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("stride", C.c_int), ("is_device", C.c_int)]


def make_image(img=None, device_ptr=None, stride=None, shape=None):
    """lvk_image for a host array (H, W) uint8 — or for a device pointer with an explicit (H, W) shape and row stride.
    Returns (Image, keepalive): hold the second value until the call that consumes the image has returned."""
    if device_ptr is not None:
        h, w = shape
        return Image(int(device_ptr), int(w), int(h), int(stride if stride is not None else w), 1), None
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim != 2 or img.strides[1] != 1:
        img = np.ascontiguousarray(img, np.uint8)
        if img.ndim != 2:
            raise ValueError("image must be a 2-D uint8 array")
    return Image(img.ctypes.data, img.shape[1], img.shape[0], img.strides[0], 0), img


class FeConfig(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("pyramid_levels", C.c_int), ("patch_size", C.c_int),
                ("max_iteration", C.c_int), ("track_precision", C.c_double), ("max_features_num", C.c_int),
                ("min_distance", C.c_int), ("flag_equalize", C.c_int), ("pub_frequency", C.c_int),
                ("distortion_model", C.c_int), ("intrinsics", C.c_double * 4), ("distortion", C.c_double * 4),
                ("R_cam_imu", C.c_double * 9)]


def lib():
    """Load liblvk_hip.so.  Fails loudly when the HIP extension has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise LvkError(f"{_SO} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        L = C.CDLL(_SO)
        vp, i, d, sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
        pi = C.POINTER(C.c_int)
        sig = {
            "lvk_runtime_env": ([C.c_uint, i], C.c_uint),
            "lvk_context_create": ([i, C.POINTER(vp)], i), "lvk_context_destroy": ([vp], None),
            "lvk_context_set_stream": ([vp, vp], i), "lvk_sync": ([vp], i),
            "lvk_last_error": ([vp], C.c_char_p), "lvk_version": ([], C.c_char_p),
            "lvk_malloc": ([vp, sz, C.POINTER(vp)], i), "lvk_free": ([vp, vp], i),
            "lvk_memcpy_h2d": ([vp, vp, vp, sz], i), "lvk_memcpy_d2h": ([vp, vp, vp, sz], i), "lvk_memset": ([vp, vp, i, sz], i),
            "lvk_clahe_u8": ([vp, vp, i, i, i, vp, i, d, i, i], i),
            "lvk_pyramid_create": ([vp, i, i, i, i, C.POINTER(vp)], i), "lvk_pyramid_destroy": ([vp], None),
            "lvk_pyramid_build": ([vp, vp, vp, i], i), "lvk_pyramid_build_clahe": ([vp, vp, vp, i, d, i, i], i),
            "lvk_pyramid_levels": ([vp], i),
            "lvk_pyramid_level": ([vp, i, pi, pi, pi, pi, pi, C.POINTER(vp), C.POINTER(vp)], i),
            "lvk_orb_prepare": ([vp, vp, vp, vp], i), "lvk_min_eigen_map": ([vp, vp, vp], i),
            "lvk_good_features": ([vp, vp, vp, i, d, d, vp, i, vp], i),
            "lvk_lk_track": ([vp, vp, vp, vp, vp, vp, i, i, d, vp], i),
            "lvk_orb_describe": ([vp, vp, vp, i, i, vp, i, vp, vp], i),
            "lvk_hamming256_rows": ([vp, vp, vp, i, vp], i),
            "lvk_undistort_points": ([vp, vp, i, vp, i, vp, vp, vp], i),
            "lvk_find_fundamental_mask": ([vp, vp, vp, i, d, d, vp, vp], i),
            "lvk_find_fundamental": ([vp, vp, vp, i, d, d, vp, vp, vp], i),
            "lvk_ransac_fundamental": ([vp, vp, vp, i, d, d, i, vp, vp], i),
            "lvk_predict_homography": ([vp, i, d, d, vp, vp, vp], i),
            "lvk_frontend_create": ([vp, C.POINTER(FeConfig), C.POINTER(vp)], i), "lvk_frontend_destroy": ([vp], None),
            "lvk_frontend_process": ([vp, C.POINTER(Image), d, vp, i, vp, i, pi, pi], i),
            "lvk_frontend_tracks": ([vp, vp, vp, vp, vp, vp, i, pi], i),
            "lvk_frontend_new_pts": ([vp, vp, i, pi], i), "lvk_frontend_state": ([vp], i),
            "lvk_frontend_lk_stats": ([vp, vp, vp], i),
            "lvk_frontend_msg_stats": ([vp, vp, vp], i),
            "lvk_frontend_profile_enable": ([vp, C.c_uint], i), "lvk_frontend_profile_read": ([vp, vp, vp, i], i),
            "lvk_frontend_stage_name": ([i], C.c_char_p),
        }
        for name, (args, res) in sig.items():
            f = getattr(L, name)
            f.argtypes = args
            f.restype = res
        _LIB = L
    return _LIB


def _p(a):
    if a is None:
        return None
    if isinstance(a, DeviceBuffer):
        return C.c_void_p(a.ptr)
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if isinstance(a, int):
        return C.c_void_p(a)
    if hasattr(a, "data_ptr"):          # torch tensor on the GPU
        return C.c_void_p(a.data_ptr())
    raise TypeError(type(a))


class Context:
    """lvk_context: one GPU, one stream."""

    def __init__(self, device=0, stream=None):
        L = lib()
        h = C.c_void_p()
        st = L.lvk_context_create(device, C.byref(h))
        if st != 0:
            raise LvkError(f"lvk_context_create failed with status {st}: no usable gfx950 device (there is no CPU fallback)")
        self.h = h
        self._bufs = []
        self._children = weakref.WeakSet()      # front-ends / filters created on this context: closed before it
        if stream is not None:
            self.check(L.lvk_context_set_stream(self.h, C.c_void_p(stream)))

    def check(self, st):
        if st != 0:
            raise LvkError(f"lvk status {st}: {lib().lvk_last_error(self.h).decode()}")

    def sync(self):
        self.check(lib().lvk_sync(self.h))

    @property
    def stream(self):
        """the hipStream_t (as an integer) the context's calls are ordered on"""
        L = lib()
        L.lvk_context_get_stream.argtypes = [C.c_void_p]; L.lvk_context_get_stream.restype = C.c_void_p
        return L.lvk_context_get_stream(self.h) or 0

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        b = DeviceBuffer(self, arr.nbytes)
        if arr.nbytes:
            self.check(lib().lvk_memcpy_h2d(self.h, C.c_void_p(b.ptr), _p(arr), arr.nbytes))
        return b

    def to_host(self, buf, dtype, shape):
        out = np.empty(shape, dtype)
        if out.nbytes:
            self.check(lib().lvk_memcpy_d2h(self.h, _p(out), C.c_void_p(buf.ptr if isinstance(buf, DeviceBuffer) else buf), out.nbytes))
        return out

    def adopt(self, obj):
        self._children.add(obj)

    def close(self):
        if self.h:
            for o in sorted(list(self._children), key=lambda o: getattr(o, "_close_order", 1)):
                try:
                    o.close()
                except Exception:
                    pass
            lib().lvk_sync(self.h)
            lib().lvk_context_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBuffer:
    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        p = C.c_void_p()
        ctx.check(lib().lvk_malloc(ctx.h, max(int(nbytes), 1), C.byref(p)))
        self.ptr = p.value
        self.nbytes = nbytes

    def free(self):
        if self.ptr and self.ctx.h:
            lib().lvk_free(self.ctx.h, C.c_void_p(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
