"""GPU parity at sequence level: the device-resident ImageProcessor against the oracle's, frame by frame.
Track ids / indices / lifetimes / descriptors bit-exact, points and the feature message bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _imu_for(seq, ts, n_hist=60):
    k0, k1 = seq.imu_index_range(-1.0, ts + 0.05)       # driver rule: samples with t < ts + 0.05 (larvioMain.cpp:98-102)
    return seq.imu_array(max(k1 - n_hist, 0), k1)


def _run_pair(gpu_ctx, frames, seq, cfg):
    from oracle import lvo
    import larvio_amd
    ora = lvo.Frontend(cfg)
    gpu = larvio_amd.ImageProcessor(cfg, gpu_ctx)
    assert gpu.initialize()
    n_msgs = 0
    for i, (ts, img) in enumerate(frames):
        imu = _imu_for(seq, ts)
        ho, mo = ora.process(img, ts, imu)
        hg, mg = gpu.processImage(img, imu, ts=ts)
        assert hg == ho, f"frame {i}: haveFeatures"
        assert gpu.state == ora.state, f"frame {i}: image_state"
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
    assert gpu.lk_stats() == ora.lk_stats()
    gpu.close()
    return n_msgs, ora


def test_sequence_parity_euroc_shape(gpu_ctx):
    from tests.conftest import synth_frames
    from larvio_amd import synthetic as S
    frames = synth_frames(40, 36)                 # t = 2.0 .. 3.75 s: bootstrap, steady tracking, re-detection
    seq = S.imu_only_sequence(S.MASTER_SEED)
    cfg = S.frontend_config(max_features_num=200)
    n_msgs, ora = _run_pair(gpu_ctx, frames, seq, cfg)
    assert n_msgs >= 15
    assert len(ora.tracks()["ids"]) > 100


def test_sequence_parity_few_features_and_no_clahe(gpu_ctx):
    """small feature budget (exercises the LMedS / <7 / ==7 paths of findFundamentalMat) and flag_equalize 0, then the same frames
    with CLAHE and a normal budget on the same context"""
    from tests.conftest import synth_frames
    from larvio_amd import synthetic as S
    frames = synth_frames(40, 26)
    seq = S.imu_only_sequence(S.MASTER_SEED)
    cfg = S.frontend_config(max_features_num=30, flag_equalize=0, min_distance=40)
    _run_pair(gpu_ctx, frames, seq, cfg)
    _run_pair(gpu_ctx, frames, seq, S.frontend_config(max_features_num=120, flag_equalize=1))
