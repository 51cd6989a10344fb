import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


_FRAME_CACHE = {}


def synth_frames(first, count, t0=0.0, cam=None):
    """Rendered synthetic frames [(ts, img)], cached on disk (rendering costs ~0.7 s/frame)."""
    from larvio_amd import synthetic as S
    cam = cam or S.EUROC
    key = (first, count, t0, cam["width"], cam["height"], cam["distortion_model"])
    if key in _FRAME_CACHE:
        return _FRAME_CACHE[key]
    path = os.path.join("/tmp", "lvk_frames_%d_%d_%g_%dx%d_%d_s%d.npz" % (first, count, t0, cam["width"], cam["height"], cam["distortion_model"], S.MASTER_SEED))
    if os.path.exists(path):
        z = np.load(path)
        out = list(zip(z["ts"].tolist(), list(z["img"])))
    else:
        seq = S.Sequence(cam=cam, t0=t0)
        out = [seq.frame(i) for i in range(first, first + count)]
        try:
            np.savez(path, ts=np.array([o[0] for o in out]), img=np.stack([o[1] for o in out]))
        except OSError:
            pass
    _FRAME_CACHE[key] = out
    return out


@pytest.fixture(scope="session")
def two_frames():
    """Two consecutive frames in the moving part of the trajectory."""
    f = synth_frames(70, 2)
    return f[0][1], f[1][1]


@pytest.fixture(scope="session")
def gpu_ctx():
    import larvio_amd
    ctx = larvio_amd.Context()       # raises without a GPU: there is no CPU fallback
    yield ctx
    ctx.close()
