"""The window bookkeeping of the moving-start initialiser against the REFERENCE ITSELF: /root/reference/src/feature_manager.cpp
(addFeatureCheckParallax :45-97, getCorresponding :100-120, removeBack :203-220) compiled in place (oracle/Makefile target `ref` ->
oracle/_ref/liblvref_fm.so; Eigen is the stand-in of oracle/ref_shim/lvref_eigen.hpp).  Held to the reference's own text: the product's
lvk_init::DynInit::add_features / corresponding / slide_window (larvio_amd/csrc/be_init.h, through the host-only harness
tests/host/window_dump.hip) on message streams whose tracks appear, live and die while the window fills and then slides once per
message: for every full window the correspondences of every frame with the newest one - how many, in which order, which coordinates
(the time-offset correction u + u_vel td included) - and the number of tracks left after every slide, exactly.  Two claims of
be_init.h's comments are checked against the real code on the way: MIN_PARALLAX = 10/460 is an integer division (feature_manager.h:25),
so addFeatureCheckParallax answers "marginalise the oldest frame" on every path; and feature ids pass through an `int`.
The first test runs the compiled reference live on fresh streams; the second holds the product to the committed outputs of the
reference (tests/golden/ref_window.npz, written by tests/golden/make_ref_window.py), which needs nothing but the file."""
import importlib.util
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "ref_window.npz")


@pytest.fixture(scope="module")
def product_window(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("window") / "window_dump")
    cxx = shutil.which("g++") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = ["-O2", "-std=c++17", "-ffp-contract=off", "-w", "-x", "c++"] if cxx.endswith("g++") else ["-O2", "-std=c++17", "-ffp-contract=off", "-w", "-x", "hip", "--offload-arch=gfx950"]
    subprocess.check_call([cxx] + flags + [os.path.join(ROOT, "tests", "host", "window_dump.hip"), "-o", exe])

    def run(td, msgs):
        path = exe + ".in"
        with open(path, "w") as f:
            f.write("%d %.17g\n" % (len(msgs), td))
            for ids, uvv in msgs:
                f.write("%d\n" % len(ids))
                for i in range(len(ids)):
                    f.write("%d %s\n" % (ids[i], " ".join("%.17g" % x for x in uvv[i])))
        cor = {}; cnt = []
        for l in subprocess.run([exe, path], capture_output=True, text=True, check=True, timeout=60).stdout.strip().splitlines():
            v = l.split(); m, i, n = int(v[0]), int(v[1]), int(v[2])
            if i < 0:
                cnt.append(n)
            else:
                cor[(m, i)] = np.array([float(x) for x in v[3:]]).reshape(n, 4)
        return cor, cnt
    return run


def _compare(cor_p, cnt_p, cor_r, cnt_r):
    assert cnt_p == list(cnt_r) and set(cor_p) == set(cor_r)
    pairs = 0
    for k in cor_r:
        assert cor_p[k].shape == cor_r[k].shape and np.array_equal(cor_p[k], cor_r[k]), k
        pairs += len(cor_r[k])
    return pairs


def test_product_window_bookkeeping_against_the_compiled_reference(product_window):
    from oracle import lvref
    if not lvref.fm_available():
        pytest.skip("oracle/_ref/liblvref_fm.so not built and /root/reference absent")
    spec = importlib.util.spec_from_file_location("make_ref_window", os.path.join(ROOT, "tests", "golden", "make_ref_window.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    pairs = 0
    for c in gen.streams(9, 8):                                                      # other streams than the committed fixture's
        ans, cor_r, cnt_r = gen.run_reference(c)
        assert all(ans)                                                              # the reference's own code: always "marginalise the oldest"
        cor_p, cnt_p = product_window(c["td"], c["msgs"])
        pairs += _compare(cor_p, cnt_p, cor_r, cnt_r)
    print("window bookkeeping against the compiled reference: 8 streams, %d correspondences identical" % pairs)
    assert pairs > 10000


def test_product_window_bookkeeping_against_the_references_committed_outputs(product_window):
    g = np.load(GOLDEN)
    pairs = 0
    for s in range(len(g["td"])):
        msgs = [(g["ids"][s][m, :g["n"][s][m]], g["uvv"][s][m, :g["n"][s][m]]) for m in range(24)]
        cor_r = {(m, l): g["cor"][s][m, l, :g["ncor"][s][m, l]] for m in range(10, 24) for l in range(10)}
        assert g["ans"][s].all()
        cor_p, cnt_p = product_window(float(g["td"][s]), msgs)
        pairs += _compare(cor_p, cnt_p, cor_r, g["cnt"][s])
    print("window bookkeeping against tests/golden/ref_window.npz: %d streams, %d correspondences identical" % (len(g["td"]), pairs))
    assert pairs > 5000
