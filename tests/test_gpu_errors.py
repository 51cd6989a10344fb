"""Error behaviour of the C ABI on a GPU box: refusals come back as a status + message (initialize() == False in the host
classes, as the reference's initialize() does when loadParameters fails) — never as a crash, and never as a silent fallback."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_frontend_refuses_what_it_does_not_implement(gpu_ctx, capsys):
    import larvio_amd
    from larvio_amd import synthetic as S
    good = S.frontend_config(max_features_num=100)
    for over in (dict(width=32), dict(height=0), dict(max_features_num=0), dict(max_features_num=10 ** 6), dict(distortion_model=2),
                 dict(min_distance=0), dict(pub_frequency=0), dict(patch_size=17), dict(pyramid_levels=-1), dict(max_iteration=0), dict(max_iteration=101)):
        fe = larvio_amd.ImageProcessor(dict(good, **over), gpu_ctx)
        assert fe.initialize() is False, over
        assert "lvk_frontend_create failed" in capsys.readouterr().out
        with pytest.raises(larvio_amd.LvkError):
            fe.processImage(np.zeros((480, 752), np.uint8), np.zeros(0, larvio_amd._lib.IMU), ts=1.0)
    fe = larvio_amd.ImageProcessor(good, gpu_ctx); assert fe.initialize()           # and the context is still usable afterwards
    has, msg = fe.processImage(np.zeros((480, 752), np.uint8), np.zeros(0, larvio_amd._lib.IMU), ts=1.0)
    assert has is False                                                              # no IMU yet: the first-image gate (image_processor.cpp:134-142)
    fe.close()


def test_filter_refuses_what_it_does_not_implement(gpu_ctx, capsys):
    import larvio_amd
    from larvio_amd import synthetic as S
    good = S.backend_config(sw_size=10)
    for over in (dict(feature_idp_dim=3), dict(use_schmidt=1), dict(sw_size=4), dict(sw_size=63), dict(calib_imu_instrinsic=2)):
        be = larvio_amd.LarVio(dict(good, **over), gpu_ctx)
        assert be.initialize() is False, over
    be = larvio_amd.LarVio(good, gpu_ctx); assert be.initialize()
    fe = larvio_amd.ImageProcessor(S.frontend_config(), gpu_ctx); assert fe.initialize()
    from larvio_amd.vio import VioPipeline
    with pytest.raises((larvio_amd.LvkError, ValueError)):                           # both halves on one context: no second stream to overlap on
        VioPipeline(fe, be, np.zeros(4, larvio_amd._lib.IMU))
    fe.close(); be.close()


def test_oversized_feature_message_is_an_error_not_a_crash(gpu_ctx):
    import larvio_amd
    from larvio_amd import synthetic as S
    from larvio_amd._lib import OBS
    seq = S.imu_only_sequence()
    imu = seq.imu_array(380, 460)
    be = larvio_amd.LarVio(S.backend_config(sw_size=8, max_features=16), gpu_ctx); assert be.initialize()
    t0 = imu["t"][10]; tr = seq.traj
    R = tr.R_wb(t0); s = np.sqrt(np.trace(R) + 1) * 2
    q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    be.set_state(t0, q, tr.p_wb(t0), tr.vel(t0), np.zeros(3), np.zeros(3), imu["gyro"][10], imu["acc"][10])
    rng = np.random.default_rng(1)
    raised = False
    for k in range(12):                                      # 4000 features per message into a filter sized for 16
        m = np.zeros(4000, OBS); m["id"] = np.arange(4000)
        m["u"] = rng.uniform(-0.5, 0.5, 4000) + 0.002 * k; m["v"] = rng.uniform(-0.4, 0.4, 4000); m["u_init"] = -1; m["v_init"] = -1
        ts = imu["t"][12 + 5 * k]
        try:
            be.processFeatures((ts, m), imu[imu["t"] < ts + 0.05])
        except larvio_amd.LvkError as e:
            raised = True
            assert "capacity" in str(e) or "exceeds" in str(e) or "too many" in str(e) or "exhausted" in str(e), str(e)
            break
    assert raised
    be.close()
    be2 = larvio_amd.LarVio(S.backend_config(sw_size=8), gpu_ctx); assert be2.initialize(); be2.close()      # the context survives


def test_image_of_the_wrong_size_is_refused_not_read_past(gpu_ctx):
    """A cv::Mat carries its own size; lvk_image does too.  A 512x512 (TUM-VI) image under a 752x480 (EuRoC) configuration, a stride
    smaller than the width, or a null pointer are LVK_ERR_ARG from every image entry point - before and after the first-image gate -
    and the front-end keeps working afterwards."""
    import larvio_amd
    from larvio_amd import synthetic as S
    from larvio_amd._lib import IMU, Image, lib
    from larvio_amd.vio import VioDriver, VioPipeline
    fe = larvio_amd.ImageProcessor(S.frontend_config(max_features_num=100), gpu_ctx); assert fe.initialize()
    imu = np.zeros(3, IMU); imu["t"] = [0.9, 0.95, 1.0]
    small = np.zeros((512, 512), np.uint8)
    for ts in (0.5, 1.0):                                                            # 0.5: gate still closed (no IMU sample older than the image)
        with pytest.raises(larvio_amd.LvkError, match="configured for 752x480"):
            fe.processImage(small, imu, ts=ts)
    good = np.zeros((480, 752), np.uint8)
    n_out, has = C.c_int(0), C.c_int(0)
    for bad in (Image(good.ctypes.data, 752, 480, 700, 0), Image(None, 752, 480, 752, 0)):
        st = lib().lvk_frontend_process(fe._h, C.byref(bad), 1.0, imu.ctypes.data_as(C.c_void_p), 3, fe._out.ctypes.data_as(C.c_void_p), fe._cap,
                                        C.byref(n_out), C.byref(has))
        assert st == 1, st                                                           # LVK_ERR_ARG
    be = larvio_amd.LarVio(S.backend_config(sw_size=8), gpu_ctx); assert be.initialize()
    with pytest.raises(larvio_amd.LvkError):
        VioDriver(fe, be, imu).step(1.0, 3, img=small)
    ctx2 = larvio_amd.Context(0)
    be2 = larvio_amd.LarVio(S.backend_config(sw_size=8), ctx2); assert be2.initialize()
    pipe = VioPipeline(fe, be2, imu)
    with pytest.raises(larvio_amd.LvkError):
        pipe.step(1.0, 3, img=small)
    pipe.close()
    # a strided view of a wider buffer (cv::Mat ROI) of the right size is fine
    wide = np.zeros((480, 800), np.uint8)
    has_msg, _ = fe.processImage(wide[:, 24:776], imu, ts=1.0)
    assert has_msg is False and fe.state in (1, 2)
    be.close(); be2.close(); fe.close(); ctx2.close()


def test_deferred_update_reports_its_failure_at_the_next_call(gpu_ctx):
    """lvk_ekf_process_async returns before the update runs: a failure of the queued update (here: a message far beyond the filter's
    capacity) has to come back from lvk_ekf_wait, and the handle has to stay failed (sticky) for every later call - not crash, not
    compute on with a half-updated state."""
    import larvio_amd
    from larvio_amd import synthetic as S
    from larvio_amd._lib import OBS
    seq = S.imu_only_sequence()
    imu = seq.imu_array(380, 520)
    be = larvio_amd.LarVio(S.backend_config(sw_size=8, max_features=16), gpu_ctx); assert be.initialize()
    t0 = imu["t"][10]; tr = seq.traj
    R = tr.R_wb(t0); s = np.sqrt(np.trace(R) + 1) * 2
    q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    be.set_state(t0, q, tr.p_wb(t0), tr.vel(t0), np.zeros(3), np.zeros(3), imu["gyro"][10], imu["acc"][10])
    rng = np.random.default_rng(1)
    failed_at = None
    for k in range(14):
        m = np.zeros(4000, OBS); m["id"] = np.arange(4000)
        m["u"] = rng.uniform(-0.5, 0.5, 4000) + 0.002 * k; m["v"] = rng.uniform(-0.4, 0.4, 4000); m["u_init"] = -1; m["v_init"] = -1
        ts = imu["t"][12 + 5 * k]
        try:
            upd, _ = be.processFeaturesAsync((ts, m), imu[imu["t"] < ts + 0.05])     # returns at once: the update is queued
            assert upd is True
            be.wait()                                                                 # ... and this is where its failure surfaces
        except larvio_amd.LvkError as e:
            failed_at = k
            assert "capacity" in str(e) or "exceeds" in str(e) or "too many" in str(e) or "exhausted" in str(e) or "failed state" in str(e), str(e)
            break
    assert failed_at is not None
    with pytest.raises(larvio_amd.LvkError, match="failed state"):
        be.processFeaturesAsync((imu["t"][100], np.zeros(4, OBS)), imu[:110])
    with pytest.raises(larvio_amd.LvkError, match="failed state"):
        be.processFeatures((imu["t"][100], np.zeros(4, OBS)), imu[:110])
    # the getters of a failed filter say so instead of handing out a half-updated state (lvk_ekf_get_state / lvk_ekf_get_cov)
    with pytest.raises(larvio_amd.LvkError):
        be.state()
    with pytest.raises(larvio_amd.LvkError):
        be.cov()
    be.close()
    be2 = larvio_amd.LarVio(S.backend_config(sw_size=8), gpu_ctx); assert be2.initialize(); be2.close()      # the context survives
