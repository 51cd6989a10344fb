"""Front-end edge cases at frame level, HIP against the oracle frame by frame (ids, lifetimes, points, descriptors, new corners and
the feature message bit for bit): odd image sizes, featureless frames at the start and in the middle of a run, a feature budget at
the bootstrap threshold, jumps beyond the LK capture range, unrelated noise frames, a padded device image (stride != width).
Frames are crops of one smooth random texture (cheap to make, plenty of corners), the IMU is a slow constant rotation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

def _texture(seed, h, w):
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    a = ndimage.gaussian_filter(rng.random((h, w)), 1.6)
    b = ndimage.gaussian_filter(rng.random((h, w)), 6.0)
    t = (a - a.mean()) / a.std() * 38 + (b - b.mean()) / b.std() * 30 + 120
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(160):                                  # bright and dark blobs: strong corners
        cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(3, 9)
        t[(np.abs(yy - cy) < r) & (np.abs(xx - cx) < r)] += rng.choice([-70, 70])
    return t.clip(0, 255).astype(np.uint8)


def _crops(tex, w, h, offsets):
    return [np.ascontiguousarray(tex[oy:oy + h, ox:ox + w]) for ox, oy in offsets]


def _imu(ts_last, rate=200.0, gyro=(0.0, 0.0, 0.04)):
    from larvio_amd._lib import IMU
    t = np.arange(0.5, ts_last + 0.2, 1.0 / rate)
    a = np.zeros(len(t), IMU)
    a["t"] = t; a["gyro"] = gyro; a["acc"] = (0.0, 0.0, 9.81)
    return a


def _cfg(w, h, **over):
    from larvio_amd import synthetic as S
    cam = dict(width=w, height=h, intrinsics=(0.6 * w, 0.6 * w, w / 2.0 - 0.5, h / 2.0 + 0.25), distortion_model=over.pop("distortion_model", 0),
               distortion=over.pop("distortion", (-0.05, 0.01, 0.0003, -0.0002)), T_cam_imu=S.EUROC["T_cam_imu"])
    return S.frontend_config(cam=cam, **over)


def _run(gpu_ctx, frames, cfg, device_stride=None, on_frame=None):
    from oracle import lvo
    import larvio_amd
    ora = lvo.Frontend(cfg)
    gpu = larvio_amd.ImageProcessor(cfg, gpu_ctx)
    assert gpu.initialize()
    ts_all = [1.0 + 0.05 * i for i in range(len(frames))]
    imu_all = _imu(ts_all[-1])
    states, n_msgs, n_tracks = [], 0, []
    keep = []
    for i, (ts, img) in enumerate(zip(ts_all, frames)):
        imu = imu_all[(imu_all["t"] < ts + 0.05)][-60:]
        ho, mo = ora.process(img, ts, imu)
        if device_stride:
            pad = np.full((img.shape[0], device_stride), 77, np.uint8); pad[:, :img.shape[1]] = img
            d = gpu_ctx.to_device(pad); keep.append(d)
            hg, mg = gpu.processImage(None, imu, ts=ts, device_ptr=d.ptr, stride=device_stride)
        else:
            hg, mg = gpu.processImage(img, imu, ts=ts)
        assert hg == ho, f"frame {i}: haveFeatures"
        assert gpu.state == ora.state, f"frame {i}: image_state {gpu.state} {ora.state}"
        to, tg = ora.tracks(), gpu.tracks()
        assert np.array_equal(tg["ids"], to["ids"]), f"frame {i}: ids"
        assert np.array_equal(tg["lifetime"], to["lifetime"]), f"frame {i}: lifetime"
        assert np.array_equal(tg["pts"].view(np.uint32), to["pts"].view(np.uint32)), f"frame {i}: pts"
        assert np.array_equal(tg["init"].view(np.uint32), to["init"].view(np.uint32)), f"frame {i}: init"
        assert np.array_equal(tg["desc"], to["desc"]), f"frame {i}: desc"
        assert np.array_equal(gpu.new_pts(), ora.new_pts()), f"frame {i}: new_pts"
        if ho:
            n_msgs += 1
            assert mg.features.tobytes() == mo.tobytes(), f"frame {i}: feature message"
        if on_frame is not None:
            on_frame(i, hg, mg.features.tobytes() if hg else b"", gpu.state, tg, gpu.new_pts())       # (the HIP front-end's outputs of this frame)
        states.append(ora.state); n_tracks.append(len(to["ids"]))
    assert gpu.lk_stats() == ora.lk_stats()
    gpu.close()
    return states, n_tracks, n_msgs


def _walk(n, step=(2, 1), start=(40, 30)):
    return [(start[0] + step[0] * i, start[1] + step[1] * i) for i in range(n)]


@pytest.mark.parametrize("w,h,levels", [(321, 243, 3), (250, 187, 2), (336, 200, 4)])
def test_odd_image_sizes(gpu_ctx, w, h, levels):
    tex = _texture(1, h + 120, w + 160)
    frames = _crops(tex, w, h, _walk(14))
    states, n_tracks, n_msgs = _run(gpu_ctx, frames, _cfg(w, h, max_features_num=120, min_distance=12, pyramid_levels=levels))
    assert states[-1] == 3 and n_tracks[-1] > 40 and n_msgs >= 5


def test_featureless_frames_at_the_start_and_in_the_middle(gpu_ctx):
    w, h = 320, 240
    tex = _texture(2, h + 120, w + 160)
    crops = _crops(tex, w, h, _walk(16))
    flat = np.full((h, w), 128, np.uint8)
    frames = [flat, flat, flat] + crops[:6] + [flat, flat] + crops[6:13] + [np.zeros((h, w), np.uint8)] + crops[13:]
    states, n_tracks, n_msgs = _run(gpu_ctx, frames, _cfg(w, h, max_features_num=100, min_distance=15))
    assert states[:3] == [1, 1, 1]                        # nothing to detect: initializeFirstFrame keeps waiting (image_processor.cpp:337-352)
    assert states[5] == 3 and n_tracks[5] > 30
    print("states", states, "tracks", n_tracks)


def test_feature_budget_at_the_bootstrap_threshold(gpu_ctx):
    """max_features_num 22: the first frame needs more than 20 corners to leave FIRST_IMAGE (image_processor.cpp:349)"""
    w, h = 320, 240
    tex = _texture(3, h + 120, w + 160)
    frames = _crops(tex, w, h, _walk(12))
    for budget, dist in ((22, 30), (20, 30), (8, 60)):
        states, n_tracks, n_msgs = _run(gpu_ctx, frames, _cfg(w, h, max_features_num=budget, min_distance=dist))
        print("budget", budget, "states", states, "tracks", n_tracks)
        if budget <= 20:
            assert set(states) == {1}                     # never more than 20 corners: stuck in FIRST_IMAGE, as the reference would be


def test_jumps_beyond_the_capture_range_and_unrelated_frames(gpu_ctx):
    w, h = 320, 240
    tex = _texture(4, h + 200, w + 300)
    offs = _walk(6) + [(40 + 2 * 6 + 45, 30 + 6 + 30)] + [(40 + 57 + 2 * i, 66 + i) for i in range(1, 6)]
    frames = _crops(tex, w, h, offs)
    rng = np.random.default_rng(5)
    noise = [rng.integers(0, 256, (h, w)).astype(np.uint8) for _ in range(2)]
    frames = frames[:9] + noise + frames[9:] + [_texture(6, h, w)] + _crops(tex, w, h, _walk(4, start=(100, 80)))
    states, n_tracks, n_msgs = _run(gpu_ctx, frames, _cfg(w, h, max_features_num=150, min_distance=10))
    print("states", states, "tracks", n_tracks)
    assert n_msgs >= 4


def test_padded_device_image_and_equidistant_model(gpu_ctx):
    w, h = 328, 248
    tex = _texture(7, h + 120, w + 160)
    frames = _crops(tex, w, h, _walk(12, step=(1, 2)))
    cfg = _cfg(w, h, max_features_num=90, min_distance=14, distortion_model=1, distortion=(0.003, 0.0007, -0.002, 0.0002), flag_equalize=0)
    states, n_tracks, n_msgs = _run(gpu_ctx, frames, cfg, device_stride=w + 24)
    assert states[-1] == 3 and n_tracks[-1] > 30
