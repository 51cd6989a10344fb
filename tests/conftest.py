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


def synth_frames(first, count, t0=0.0, cam=None, img_rate=20.0):
    """Rendered synthetic frames [(ts, img)], cached on disk.  Rendering (~0.25 s/frame/core) happens in a fresh subprocess that
    forks one worker per core (larvio_amd.synthetic.render_frames): this process may already have initialised HIP, after which
    forking is not safe."""
    import pickle
    import subprocess
    from larvio_amd import synthetic as S
    cam = cam or S.EUROC
    assert t0 == 0.0
    key = (first, count, cam["width"], cam["height"], cam["distortion_model"], img_rate)
    if key in _FRAME_CACHE:
        return _FRAME_CACHE[key]
    code = ("import pickle, sys; sys.path.insert(0, %r); from larvio_amd import synthetic as S; "
            "cam = pickle.loads(bytes.fromhex(%r)); S.render_frames(%d, %d, cam=cam, img_rate=%r)"
            % (ROOT, pickle.dumps(dict(cam)).hex(), first, count, img_rate))
    subprocess.check_call([sys.executable, "-c", code])
    ts, img = S.render_frames(first, count, cam=cam, img_rate=img_rate, procs=1)      # now a cache hit
    out = list(zip(ts.tolist(), list(img)))
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
