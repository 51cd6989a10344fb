"""The static initialiser against the REFERENCE ITSELF: /root/reference/src/StaticInitializer.cpp (tryIncInit :12-58,
initializeGravityAndBias :61-109, assignInitialState :112-145) compiled in place (oracle/Makefile target `ref` ->
oracle/_ref/liblvref_static.so; Eigen is the stand-in of oracle/ref_shim/lvref_eigen.hpp).  Held to the reference's own text: the
oracle's static_try_init (oracle/be_filter.c, through the stage entry lvo_ekf_static_try_init) - the message at which the initialiser
fires (consecutive static images, the 19th-largest displacement rule, the counter reset on fewer than 20 common features), the state
time (last IMU sample inside the window), the attitude from the mean specific force (Quaterniond::FromTwoVectors), the gyro bias from
the mean rate, and the number of IMU samples assignInitialState erases.  The product's static_try_init (larvio_amd/csrc/backend.hip)
is the same restatement, compared with the oracle after every update of the static-start runs of the GPU suite
(tests/test_gpu_backend.py::test_backend_sequence_parity_pure_msckf_and_static_init).
The first test runs the compiled reference live on fresh streams; the second holds the oracle to the committed outputs of the
reference (tests/golden/ref_static.npz, written by tests/golden/make_ref_static.py), which needs nothing but the file."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import lvo, lvo_be

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "ref_static.npz")


def _gen():
    spec = importlib.util.spec_from_file_location("make_ref_static", os.path.join(ROOT, "tests", "golden", "make_ref_static.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    return gen


def _oracle(static_num, msgs, imu7, thresh):
    from larvio_amd import synthetic as S
    cfg = S.backend_config(if_zupt_valid=0)
    cfg["zupt_max_feature_dis"] = thresh; cfg["static_duration"] = static_num / cfg["pub_frequency"]
    e = lvo_be.Ekf(cfg)
    imu = np.zeros(len(imu7), lvo.IMU); imu["t"] = imu7[:, 0]; imu["gyro"] = imu7[:, 1:4]; imu["acc"] = imu7[:, 4:7]
    for i, (t, ids, uv) in enumerate(msgs):
        f = np.zeros(len(ids), lvo.OBS); f["id"] = ids; f["u"] = uv[:, 0]; f["v"] = uv[:, 1]; f["u_init"] = -1; f["v_init"] = -1
        hi = int(np.searchsorted(imu["t"], t + 0.05))
        r = e.static_try_init(float(t), f, imu[:hi])
        if r is not None:
            return i, r
    return -1, None


def _same(i_o, r_o, i_r, r_r):
    assert i_o == i_r, (i_o, i_r)
    if i_r < 0:
        return 0.0
    assert r_o["t"] == r_r["t"] and r_o["erased"] == r_r["erased"]
    return float(max(np.abs(r_o["q"] - r_r["q"]).max(), np.abs(r_o["bg"] - r_r["bg"]).max()))


def test_oracle_static_initialiser_against_the_compiled_reference():
    from oracle import lvref
    if not lvref.static_available():
        pytest.skip("oracle/_ref/liblvref_static.so not built and /root/reference absent")
    gen = _gen()
    worst = 0.0; fired = 0
    for c in gen.cases(3, 18):                                                       # other streams than the committed fixture's
        i_r, r_r = gen.run_reference(c)
        i_o, r_o = _oracle(c["static_num"], c["msgs"], c["imu7"], gen.THRESH)
        worst = max(worst, _same(i_o, r_o, i_r, r_r)); fired += i_r >= 0
    print("static initialiser against the compiled reference: 18 streams, %d initialised, worst difference %.1e" % (fired, worst))
    assert fired >= 12 and worst < 1e-14


def test_oracle_static_initialiser_against_the_references_committed_outputs():
    g = np.load(GOLDEN); thresh = _gen().THRESH
    worst = 0.0; fired = 0
    for k in range(len(g["msg"])):
        nf = int(g["nf"][k])
        msgs = [(float(g["ts"][k][j]), g["ids"][k][j, :nf], g["uv"][k][j, :nf]) for j in range(24)]
        i_o, r_o = _oracle(int(g["static_num"][k]), msgs, g["imu7"][k][:int(g["n_imu"][k])], thresh)
        o = g["out"][k]
        r_r = dict(t=float(o[0]), q=o[1:5], bg=o[5:8], erased=int(o[8])) if g["msg"][k] >= 0 else None
        worst = max(worst, _same(i_o, r_o, int(g["msg"][k]), r_r)); fired += g["msg"][k] >= 0
    print("static initialiser against tests/golden/ref_static.npz: %d streams, %d initialised, worst difference %.1e" % (len(g["msg"]), fired, worst))
    assert worst < 1e-14
