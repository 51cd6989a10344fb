"""Seeded synthetic EuRoC-shaped input: images + IMU + ground truth.

The EuRoC dataset is not available offline, so benchmarks and parity tests run on a procedural
scene: a textured box room rendered through the pinhole+radtan (or equidistant) camera of the
reference's config (config/euroc.yaml:15-41), seen from a smooth Lissajous trajectory with a
static lead-in (the reference's static initializer needs ~1 s at rest, euroc.yaml:92), with a
200 Hz IMU stream (analytic derivatives + white noise, euroc.yaml:69-72).
Master seed 20260924.  numpy only; this is input generation, not part of the timed path.
"""
import numpy as np

MASTER_SEED = 20260924
# Simulated IMU noise densities: the EuRoC sensor's (ADIS16448 datasheet class), NOT the filter's parameters —
# config/euroc.yaml:69-72 (0.004 / 0.08) are deliberately inflated tuning values; feeding noise that large makes the
# reference's 5%-lower-tail chi-square gate (larvio.cpp:353-357) reject most features.
IMU_NOISE_GYRO = 1.7e-4     # rad/s/sqrt(Hz)
IMU_NOISE_ACC = 2.0e-3      # m/s^2/sqrt(Hz)

# config/euroc.yaml:15-41 (values are data, restated)
EUROC = dict(
    width=752, height=480,
    intrinsics=(458.654, 457.296, 367.215, 248.375),
    distortion_model=0,  # radtan
    distortion=(-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05),
    T_cam_imu=np.array([[0.014865542981794, 0.999557249008346, -0.025774436697440, 0.065222909535531],
                        [-0.999880929698575, 0.014967213324719, 0.003756188357967, -0.020706385492719],
                        [0.004140296794224, 0.025715529947966, 0.999660727177902, -0.008054602460030],
                        [0, 0, 0, 1.0]]),
)


def frontend_config(cam=EUROC, max_features_num=200, pyramid_levels=2, patch_size=21, max_iteration=30,
                    track_precision=0.01, min_distance=20, flag_equalize=1, pub_frequency=10):
    """The ImageProcessor parameters (image_processor.cpp:44-113) as a plain dict."""
    R_imu_cam = np.asarray(cam["T_cam_imu"], np.float64)[:3, :3]
    return dict(width=cam["width"], height=cam["height"], pyramid_levels=pyramid_levels, patch_size=patch_size,
                max_iteration=max_iteration, track_precision=track_precision, max_features_num=max_features_num,
                min_distance=min_distance, flag_equalize=flag_equalize, pub_frequency=pub_frequency,
                distortion_model=cam["distortion_model"], intrinsics=tuple(cam["intrinsics"]),
                distortion=tuple(cam["distortion"]), R_cam_imu=R_imu_cam.T.copy())  # image_processor.cpp:93


def _smoothstep(x):
    x = np.clip(x, 0.0, 1.0)
    return x * x * x * (x * (6 * x - 15) + 10)


def _rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    if axis == 0:
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == 1:
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


class Trajectory:
    """Body (IMU) pose in a z-up world, gravity (0,0,-9.81)."""

    def __init__(self, cam=EUROC, static_s=1.2, ramp_s=2.0, amp=(1.2, 1.0, 0.4), ang=(0.25, 0.12, 0.10), speed=1.0):
        T = np.asarray(cam["T_cam_imu"], np.float64)
        self.R_cb, self.t_cb = T[:3, :3], T[:3, 3]
        R_wc0 = np.array([[0, 0, 1.0], [-1, 0, 0], [0, -1, 0]])
        self.R_wb0 = R_wc0 @ self.R_cb
        self.static_s, self.ramp_s, self.amp, self.ang, self.speed = static_s, ramp_s, amp, ang, speed
        self.g = np.array([0, 0, -9.81])

    def _env(self, t):
        return _smoothstep((t - self.static_s) / self.ramp_s)

    def p_wb(self, t):
        e = self._env(t); s = self.speed; tt = t - self.static_s
        return e * np.array([self.amp[0] * np.sin(0.31 * s * tt), self.amp[1] * np.sin(0.43 * s * tt + 0.7) - self.amp[1] * np.sin(0.7),
                             self.amp[2] * np.sin(0.59 * s * tt + 1.3) - self.amp[2] * np.sin(1.3)])

    def R_wb(self, t):
        e = self._env(t); s = self.speed; tt = t - self.static_s
        yaw = e * self.ang[0] * np.sin(0.37 * s * tt)
        pitch = e * self.ang[1] * np.sin(0.53 * s * tt + 0.4)
        roll = e * self.ang[2] * np.sin(0.29 * s * tt + 1.1)
        return _rot(2, yaw) @ _rot(1, pitch) @ _rot(0, roll) @ self.R_wb0

    def vel(self, t, h=1e-4):
        return (self.p_wb(t + h) - self.p_wb(t - h)) / (2 * h)

    def acc(self, t, h=1e-4):
        return (self.p_wb(t + h) - 2 * self.p_wb(t) + self.p_wb(t - h)) / (h * h)

    def omega_b(self, t, h=1e-5):
        R0, R1, R = self.R_wb(t - h), self.R_wb(t + h), self.R_wb(t)
        W = R.T @ (R1 - R0) / (2 * h)
        return np.array([W[2, 1] - W[1, 2], W[0, 2] - W[2, 0], W[1, 0] - W[0, 1]]) * 0.5

    def cam_pose(self, t):
        R_wb, p = self.R_wb(t), self.p_wb(t)
        R_wc = R_wb @ self.R_cb.T
        p_wc = p + R_wb @ (-self.R_cb.T @ self.t_cb)
        return R_wc, p_wc

    def imu(self, t):
        R = self.R_wb(t)
        return self.omega_b(t), R.T @ (self.acc(t) - self.g)


def _make_texture(rng, n=1536):
    """Corner-rich 8-bit texture: band-limited noise + random rectangles/discs, lightly blurred."""
    from scipy import ndimage
    tex = np.zeros((n, n), np.float32)
    for scale, amp in ((96, 38.0), (32, 26.0), (12, 16.0), (5, 9.0)):
        m = n // scale + 3
        coarse = rng.standard_normal((m, m)).astype(np.float32)
        up = ndimage.zoom(coarse, scale, order=3)[:n, :n]
        tex += amp * up
    tex += 118.0
    nshape = 900
    cx = rng.integers(0, n, nshape); cy = rng.integers(0, n, nshape)
    sz = rng.integers(6, 42, (nshape, 2)); val = rng.uniform(-85, 85, nshape); kind = rng.integers(0, 2, nshape)
    yy, xx = np.mgrid[0:n, 0:n]
    for i in range(nshape):
        x0, x1 = max(cx[i] - sz[i, 0], 0), min(cx[i] + sz[i, 0], n)
        y0, y1 = max(cy[i] - sz[i, 1], 0), min(cy[i] + sz[i, 1], n)
        if kind[i] == 0:
            tex[y0:y1, x0:x1] += val[i]
        else:
            sub = (xx[y0:y1, x0:x1] - cx[i]) ** 2 / max(sz[i, 0], 1) ** 2 + (yy[y0:y1, x0:x1] - cy[i]) ** 2 / max(sz[i, 1], 1) ** 2 <= 1
            tex[y0:y1, x0:x1] += val[i] * sub
    tex = ndimage.gaussian_filter(tex, 0.8)
    return np.clip(tex, 4, 251).astype(np.float32)


class Scene:
    """Box room x,y in [-5,5], z in [-2,2.5]; one texture per face."""

    def __init__(self, seed=MASTER_SEED, tex_px_per_m=150.0, tex_n=1536):
        ss = np.random.SeedSequence(seed)
        self.rngs = [np.random.default_rng(s) for s in ss.spawn(8)]
        self.ppm = tex_px_per_m
        # (normal, offset d with n.x = d, u axis, v axis)
        self.planes = [
            (np.array([1.0, 0, 0]), 5.0, np.array([0, 1.0, 0]), np.array([0, 0, 1.0])),
            (np.array([-1.0, 0, 0]), 5.0, np.array([0, 1.0, 0]), np.array([0, 0, 1.0])),
            (np.array([0, 1.0, 0]), 5.0, np.array([1.0, 0, 0]), np.array([0, 0, 1.0])),
            (np.array([0, -1.0, 0]), 5.0, np.array([1.0, 0, 0]), np.array([0, 0, 1.0])),
            (np.array([0, 0, 1.0]), 2.5, np.array([1.0, 0, 0]), np.array([0, 1.0, 0])),
            (np.array([0, 0, -1.0]), 2.0, np.array([1.0, 0, 0]), np.array([0, 1.0, 0])),
        ]
        self.tex = [_make_texture(self.rngs[i], tex_n) for i in range(6)]
        self.tex_n = tex_n
        self.tex_stack = np.stack(self.tex)

    def shade(self, origin, rays):
        """rays (...,3) world directions -> intensity float32 (...)."""
        shp = rays.shape[:-1]
        r = rays.reshape(-1, 3)
        N = np.stack([p[0] for p in self.planes])            # (6,3)
        d = np.array([p[1] for p in self.planes])
        denom = r @ N.T                                       # (n,6)
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (d - N @ origin)[None, :] / denom
        t = np.where((denom > 1e-9) & (t > 0), t, np.inf)
        k = np.argmin(t, axis=1)
        tk = t[np.arange(len(r)), k]
        hit = origin + tk[:, None] * r
        UA = np.stack([p[2] for p in self.planes])[k]; VA = np.stack([p[3] for p in self.planes])[k]
        n = self.tex_n
        u = np.mod((np.einsum("ij,ij->i", hit, UA) + 5.0) * self.ppm, n - 1)
        v = np.mod((np.einsum("ij,ij->i", hit, VA) + 5.0) * self.ppm, n - 1)
        iu = np.floor(u).astype(np.int64); iv = np.floor(v).astype(np.int64)
        fu = (u - iu).astype(np.float32); fv = (v - iv).astype(np.float32)
        T = self.tex_stack
        val = (T[k, iv, iu] * (1 - fu) + T[k, iv, iu + 1] * fu) * (1 - fv) + (T[k, iv + 1, iu] * (1 - fu) + T[k, iv + 1, iu + 1] * fu) * fv
        return val.reshape(shp)


def _undistort_grid(cam):
    """Normalized undistorted ray (x,y,1) for every pixel centre (vectorised fixed-point inverse)."""
    w, h = cam["width"], cam["height"]
    fx, fy, cx, cy = cam["intrinsics"]
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    xd, yd = (xs - cx) / fx, (ys - cy) / fy
    d = cam["distortion"]
    if cam["distortion_model"] == 0:
        k1, k2, p1, p2 = d
        x, y = xd.copy(), yd.copy()
        for _ in range(40):
            r2 = x * x + y * y
            ic = 1.0 / (1 + (k2 * r2 + k1) * r2)
            dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x); dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            x, y = (xd - dx) * ic, (yd - dy) * ic
    else:
        thd = np.sqrt(xd * xd + yd * yd)
        th = thd.copy()
        for _ in range(30):
            t2 = th * th
            f = th * (1 + t2 * (d[0] + t2 * (d[1] + t2 * (d[2] + t2 * d[3])))) - thd
            fp = 1 + t2 * (3 * d[0] + t2 * (5 * d[1] + t2 * (7 * d[2] + t2 * 9 * d[3])))
            th = th - f / fp
        s = np.where(thd > 1e-9, np.tan(th) / np.maximum(thd, 1e-12), 1.0)
        x, y = xd * s, yd * s
    return np.stack([x, y, np.ones_like(x)], -1)


class Sequence:
    """frame(i) -> (ts, u8 image); imu_between(t0, t1) -> structured IMU samples."""

    def __init__(self, cam=EUROC, seed=MASTER_SEED, img_rate=20.0, imu_rate=200.0, t0=0.0,
                 noise_gyro=IMU_NOISE_GYRO, noise_acc=IMU_NOISE_ACC, pixel_noise=1.5, traj=None, scene=None):
        self.cam, self.img_rate, self.imu_rate, self.t0 = cam, img_rate, imu_rate, t0
        self.traj = traj or Trajectory(cam)
        self.scene = scene or Scene(seed)
        self.rays_c = _undistort_grid(cam)
        self.seed = seed
        self.pixel_noise = pixel_noise
        # discrete-time white noise: sigma_d = sigma_c * sqrt(rate)
        self.sg, self.sa = noise_gyro * np.sqrt(imu_rate), noise_acc * np.sqrt(imu_rate)

    def frame_time(self, i):
        return self.t0 + i / self.img_rate

    def frame(self, i):
        t = self.frame_time(i)
        R_wc, p_wc = self.traj.cam_pose(t)
        rays_w = self.rays_c @ R_wc.T
        img = self.scene.shade(p_wc, rays_w)
        rng = np.random.default_rng([self.seed, 7, i])
        img = img + rng.standard_normal(img.shape).astype(np.float32) * self.pixel_noise
        return t, np.clip(np.rint(img), 0, 255).astype(np.uint8)

    def imu_index_range(self, t_lo, t_hi):
        """IMU sample indices k with t_lo <= t_k < t_hi, t_k = t0 + k/imu_rate - 0.5/imu_rate*0."""
        k0 = int(np.ceil((t_lo - self.t0) * self.imu_rate - 1e-9)); k1 = int(np.ceil((t_hi - self.t0) * self.imu_rate - 1e-9))
        return max(k0, 0), max(k1, 0)

    def imu_sample(self, k):
        t = self.t0 + k / self.imu_rate
        w, a = self.traj.imu(t)
        rng = np.random.default_rng([self.seed, 11, k])
        n = rng.standard_normal(6)
        return t, w + self.sg * n[:3], a + self.sa * n[3:]

    def imu_array(self, k0, k1):
        from numpy import zeros
        out = zeros(max(k1 - k0, 0), dtype=[("t", np.float64), ("gyro", np.float64, 3), ("acc", np.float64, 3)])
        for j, k in enumerate(range(k0, k1)):
            t, w, a = self.imu_sample(k)
            out[j] = (t, w, a)
        return out


def backend_config(cam=EUROC, sw_size=30, max_track_len=6, max_features_in_one_grid=1, **over):
    """The LarVio parameters (larvio.cpp:58-311) with config/euroc.yaml's values; sw_size 30 per BASELINE.json."""
    c = dict(if_fej=1, estimate_extrin=1, estimate_td=1, if_zupt_valid=1, sw_size=sw_size, max_track_len=max_track_len,
             least_observation_number=3, max_features_in_one_grid=max_features_in_one_grid, aug_grid_rows=5, aug_grid_cols=6,
             pub_frequency=10, imu_rate=200, width=cam["width"], height=cam["height"], intrinsics=tuple(cam["intrinsics"]),
             T_cam_imu=np.asarray(cam["T_cam_imu"], np.float64), td=0.0,
             noise_gyro=0.004, noise_acc=0.08, noise_gyro_bias=2e-6, noise_acc_bias=4e-5, noise_feature=0.008,
             initial_covariance_orientation=4e-4, initial_covariance_velocity=0.25, initial_covariance_position=1.0,
             initial_covariance_gyro_bias=4e-4, initial_covariance_acc_bias=0.01, initial_covariance_extrin_rot=3.0462e-8,
             initial_covariance_extrin_trans=9e-8, rotation_threshold=0.2618, translation_threshold=0.4, tracking_rate_threshold=0.5,
             feature_translation_threshold=-1.0, zupt_max_feature_dis=2e-3, zupt_noise_v=1e-2, zupt_noise_p=1e-2, zupt_noise_q=3.4e-2,
             static_duration=1.0)
    c.update(over)
    return c


def imu_only_sequence(seed=MASTER_SEED, cam=EUROC, imu_rate=200.0, noise_scale=1.0):
    """A Sequence without scene/ray tables: only frame times and the IMU stream (cheap to construct)."""
    seq = Sequence.__new__(Sequence)
    seq.cam = cam; seq.traj = Trajectory(cam); seq.t0 = 0.0; seq.img_rate = 20.0; seq.imu_rate = imu_rate; seq.seed = seed
    seq.sg = IMU_NOISE_GYRO * np.sqrt(imu_rate) * noise_scale; seq.sa = IMU_NOISE_ACC * np.sqrt(imu_rate) * noise_scale
    return seq



def unaligned_stamps(ts, imu_all, seed=MASTER_SEED, phase=1.7e-3, cam_jitter=3e-4, imu_jitter=5e-5):
    """A camera that is NOT on the IMU grid (every real one): image stamps shifted by a constant phase plus per-frame jitter, IMU stamps
    jittered around their grid.  The synthetic sequence puts every image stamp exactly half-way between two IMU samples' bounds, where
    the pipelined driver's early erase count (lvk_vio_pipe_submit) is always safe; with these stamps a sample falls within its 0.5 ms
    margin of the bound in roughly a fifth of the message frames, which then wait for the running update (larvio.cpp:464-512 is the
    rule both schedules implement).  The constant phase acts as a true camera-IMU time offset: with estimate_td the filter's td walks
    towards it, so the bound moves while the test runs.  Returns (ts', imu_all') - same images, same IMU values, new stamps."""
    rng = np.random.default_rng([seed, 77])
    ts2 = np.asarray(ts, np.float64) + phase + rng.uniform(-cam_jitter, cam_jitter, len(ts))
    imu2 = np.array(imu_all, copy=True)
    imu2["t"] = imu2["t"] + rng.uniform(-imu_jitter, imu_jitter, len(imu2))
    return ts2, imu2

# ------------------------------------------------------------------ the configurations BASELINE.json names
CAM_TUMVI_LIKE = dict(width=512, height=512, intrinsics=(190.978, 190.973, 254.932, 256.897), distortion_model=1,
                      distortion=(0.0034823894, 0.0007150348, -0.0020532361, 0.0002029367), T_cam_imu=EUROC["T_cam_imu"])
CAM_1080P = dict(width=1920, height=1080, intrinsics=(1100.0, 1100.0, 960.0, 540.0), distortion_model=0,
                 distortion=(-0.12, 0.03, 0.0002, -0.0001), T_cam_imu=EUROC["T_cam_imu"])


def workload(name, max_features=None, sw_size=None):
    """-> dict(cam, img_rate, fcfg, bcfg, label) for BASELINE.json's configs[1..4] ('A', '3', '4', '5').
    The tracker budget (max_features_num) is sized so that the tracker HOLDS about the track count the configuration states:
    a budget of 150 keeps ~133 alive on this sequence, 170 keeps ~152 (messages of ~151 features); 350 keeps ~300 at configs[3]."""
    if name == "A":       # configs[1]: EuRoC-shaped 752x480 @20 Hz, ~150 tracks, 30-clone window, 1-D hybrid
        cam, rate, mf, sw, fo, bo = EUROC, 20.0, 170, 30, {}, {}
        label = "configs[1]: EuRoC-shaped synthetic 752x480 @20 Hz, 1d-hybrid"
    elif name == "3":     # configs[2]: as A with online imu-cam extrinsic + td + IMU-intrinsic calibration (LEG_DIM 46)
        cam, rate, mf, sw, fo, bo = EUROC, 20.0, 170, 30, {}, dict(calib_imu_instrinsic=1)
        label = "configs[2]: 752x480 @20 Hz, 1d-hybrid + online extrinsic/td/IMU-intrinsic calibration"
    elif name == "4":     # configs[3]: TUM-VI-shaped 512x512 equidistant, ~300 tracks, ZUPT
        cam, rate, mf, sw, fo, bo = CAM_TUMVI_LIKE, 20.0, 350, 30, dict(min_distance=15), dict(if_zupt_valid=1)
        label = "configs[3]: TUM-VI-shaped synthetic 512x512 equidistant @20 Hz, 1d-hybrid + ZUPT"
    elif name == "5":     # configs[4]: 1920x1080 @60 Hz, 2000 tracks, 60-clone window (messages at 30 Hz: every other frame, as EuRoC's 20 -> 10)
        cam, rate, mf, sw, fo, bo = CAM_1080P, 60.0, 2000, 60, dict(pub_frequency=30), dict(pub_frequency=30, max_features_in_one_grid=2)
        label = "configs[4]: synthetic 1920x1080 @60 Hz, 1d-hybrid, messages at 30 Hz"
    else:
        raise ValueError("workload must be one of A, 3, 4, 5")
    nominal = {"A": 150, "3": 150, "4": 300, "5": 2000}[name]       # the track count BASELINE.json states for the configuration
    mf = max_features or mf; sw = sw_size or sw
    fcfg = frontend_config(cam=cam, max_features_num=mf, **fo)
    bcfg = backend_config(cam=cam, sw_size=sw, max_features=mf, **bo)
    return dict(name=name, cam=cam, img_rate=rate, fcfg=fcfg, bcfg=bcfg, max_features=mf, sw_size=sw, nominal_tracks=nominal,
                label="%s, tracker budget max_features_num %d (sized to HOLD the stated ~%d tracks), sw_size %d" % (label, mf, nominal, sw))


_RENDER_SEQ = None


def _render_one(i):
    return _RENDER_SEQ.frame(i)


def render_frames(first, count, cam=EUROC, seed=MASTER_SEED, img_rate=20.0, procs=None, cache_dir="/tmp"):
    """(ts[count], img[count, H, W] u8) of frames [first, first+count), rendered on `procs` forked workers (frames are independent)
    and cached on disk one file per frame, so overlapping ranges share work.  Call BEFORE anything initialises HIP in this process
    when frames are missing (the workers are forked)."""
    import os
    d = os.path.join(cache_dir, "lvk_frames_%dx%d_m%d_r%g_s%d" % (cam["width"], cam["height"], cam["distortion_model"], img_rate, seed))
    os.makedirs(d, exist_ok=True)
    idx = list(range(first, first + count))
    fn = lambda i: os.path.join(d, "f%06d.npy" % i)
    missing = [i for i in idx if not os.path.exists(fn(i))]
    if missing:
        global _RENDER_SEQ
        _RENDER_SEQ = Sequence(cam=cam, seed=seed, img_rate=img_rate)
        procs = procs or min(32, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
        if procs > 1 and len(missing) > 4:
            import multiprocessing as mp
            with mp.get_context("fork").Pool(procs) as pool:
                res = pool.map(_render_one, missing, chunksize=max(1, len(missing) // (4 * procs)))
        else:
            res = [_render_one(i) for i in missing]
        _RENDER_SEQ = None
        for i, (t, im) in zip(missing, res):
            tmp = fn(i) + ".tmp%d.npy" % os.getpid()
            np.save(tmp, im); os.replace(tmp, fn(i))
    ts = np.array([i / img_rate for i in idx])               # Sequence.frame_time with t0 = 0
    img = np.stack([np.load(fn(i)) for i in idx])
    return ts, img


# ------------------------------------------------------------------ feature-level simulator (no images, no front-end)
# A landmark cloud seen by the camera of the synthetic trajectory, turned into the feature messages a front-end would publish:
# normalised, undistorted observations with additive noise, persistent ids, finite-difference velocities - plus the matching IMU
# stream.  Ground truth for the filter that passes through no tracker; also the cheap way to load the back-end at configs[4]
# depth (2000 features per message) without rendering 1080p frames (bench.py --backend-only, tests/feature_sim.py).
_OBS = np.dtype([("id", np.uint64), ("u", np.float64), ("v", np.float64), ("u_init", np.float64), ("v_init", np.float64),
                 ("u_vel", np.float64), ("v_vel", np.float64), ("u_init_vel", np.float64), ("v_init_vel", np.float64)])


def R2q(R):
    t = np.trace(R); s = np.sqrt(t + 1) * 2
    return np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])


def qmul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


class LandmarkCloud:
    def __init__(self, tr, t0, t1, seed, n_per_batch=120, fov=(0.75, 0.48)):
        rng = np.random.default_rng([seed, 3]); self.tr, self.fov = tr, fov
        pts = []
        for t in np.arange(t0 - 0.5, t1 + 0.5, 0.5):          # landmarks placed in front of the camera along the whole path
            R_wc, p_wc = tr.cam_pose(t)
            u = rng.uniform(-fov[0], fov[0], n_per_batch); v = rng.uniform(-fov[1], fov[1], n_per_batch); z = rng.uniform(2.0, 9.0, n_per_batch)
            pts.append(np.stack([u * z, v * z, z], 1) @ R_wc.T + p_wc)
        self.pts = np.concatenate(pts)

    def project(self, t):
        R_wc, p_wc = self.tr.cam_pose(t)
        pc = (self.pts - p_wc) @ R_wc
        z = pc[:, 2]; ok = z > 0.5
        uv = np.full((len(pc), 2), np.nan); uv[ok] = pc[ok, :2] / z[ok, None]
        return uv, ok & (np.abs(uv[:, 0]) < self.fov[0]) & (np.abs(uv[:, 1]) < self.fov[1])


INIT_COV = dict(initial_covariance_orientation=1e-6, initial_covariance_velocity=1e-4, initial_covariance_position=1e-6,
                initial_covariance_gyro_bias=1e-8, initial_covariance_acc_bias=1e-6)


def simulate_features(seed, t0=2.0, t1=8.0, sigma=3e-4, imu_noise=1.0, max_feat=150, perturb=True, n_per_batch=120, traj=None, fresh_ids=False, **cfg_over):
    """-> dict(cfg, imu, init=(t, q, p, v, bg, ba, gyro_old, acc_old), msgs=[(ts, OBS array)], traj)
    sigma: observation noise in normalised image units (3e-4 ~ 0.14 px at f = 458: what sub-pixel LK delivers);
    imu_noise: scale on the simulator's IMU noise densities (1.0 = synthetic.IMU_NOISE_*).  The initial state is the truth plus a
    draw from the (small) initial covariance the configuration states.  traj: another Trajectory than the default one; fresh_ids: a
    landmark that left the view never comes back under its id (what a tracker does: a lost track's id is not reused)."""
    tr = traj if traj is not None else Trajectory(); seq = imu_only_sequence(seed=seed, noise_scale=imu_noise); seq.traj = tr
    rng = np.random.default_rng([seed, 5])
    cloud = LandmarkCloud(tr, t0, t1, seed, n_per_batch=n_per_batch)
    over = dict(sw_size=20, estimate_td=0, estimate_extrin=0, if_zupt_valid=0, **INIT_COV)
    over.update(cfg_over)
    cfg = backend_config(**over)
    imu = seq.imu_array(int(round(t0 * 200)) - 2, int(t1 * 200) + 40)
    ki = int(np.searchsorted(imu["t"], t0, side="right")) - 1; ti = imu["t"][ki]
    q = R2q(tr.R_wb(ti)); p = tr.p_wb(ti).copy(); v = tr.vel(ti).copy()
    if perturb:
        dth = rng.normal(0, np.sqrt(cfg["initial_covariance_orientation"]), 3)
        q = qmul(np.concatenate([0.5 * dth, [1.0]]), q); q /= np.linalg.norm(q)
        p += rng.normal(0, np.sqrt(cfg["initial_covariance_position"]), 3); v += rng.normal(0, np.sqrt(cfg["initial_covariance_velocity"]), 3)
    init = (ti, q, p, v, np.zeros(3), np.zeros(3), imu["gyro"][ki].copy(), imu["acc"][ki].copy())
    tracked, prev_uv, msgs, retired = {}, None, [], set()
    for i in range(int(round((t1 - t0) * 20)) + 1):            # camera frames at 20 Hz, a message every other frame
        ts = t0 + i * 0.05
        uv, vis = cloud.project(ts)
        uvn = uv + rng.normal(0, sigma, uv.shape)
        for j in list(tracked):
            if not vis[j]:
                del tracked[j]
                if fresh_ids:
                    retired.add(j)
        if len(tracked) < max_feat:
            cand = np.flatnonzero(vis); rng.shuffle(cand)
            for j in cand:
                if len(tracked) >= max_feat:
                    break
                if int(j) not in retired:
                    tracked.setdefault(int(j), True)
        if i % 2 == 0 and prev_uv is not None:
            ids = sorted(tracked)
            m = np.zeros(len(ids), _OBS)
            for r, j in enumerate(ids):
                pv = prev_uv[j] if np.isfinite(prev_uv[j]).all() else uvn[j]
                m[r] = (j, uvn[j, 0], uvn[j, 1], -1.0, -1.0, (uvn[j, 0] - pv[0]) / 0.05, (uvn[j, 1] - pv[1]) / 0.05, 0.0, 0.0)
            msgs.append((ts, m))
        prev_uv = uvn
    return dict(cfg=cfg, imu=imu, init=init, msgs=msgs, traj=tr, landmarks=cloud.pts)      # feature id = row of `landmarks`


