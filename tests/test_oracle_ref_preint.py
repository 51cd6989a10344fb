"""The IMU pre-integration of the moving-start initialiser against the REFERENCE ITSELF: /root/reference/include/Initializer/
ImuPreintegration.h (IntegrationBase: mid-point rule :62-142, propagate :144-167, repropagate :48-61) compiled in place
(oracle/Makefile target `ref` -> oracle/_ref/liblvref_preint.so; Eigen is not installed, so Matrix / MatrixXd / Quaterniond are the
stand-ins of oracle/ref_shim/lvref_eigen.hpp: plain loops, no claim about Eigen's rounding).  Held to the reference's own text here,
to 1e-12: BOTH restatements of row N4 - the product's lvk_init::PreInt (larvio_amd/csrc/be_init.h, through the host-only harness
tests/host/preint_dump.hip) and the independent numpy one (oracle/dyn_init.py PreInt) - on delta_p, delta_q, delta_v, sum_dt and the
bias Jacobian d(delta_q)/d(b_g) the gyroscope-bias solve reads (jacobian.block<3,3>(O_R, O_BG)), after pushing the samples and after a
re-propagation about another gyro bias.  A sign, an operand order (delta_q * small rotation vs the reverse), the product with the not yet
normalised quaternion (:77-78) or a swapped Jacobian block would each be orders of magnitude above the bound.
The first test runs the compiled reference live on fresh streams; the second holds both restatements to the committed outputs of the
reference (tests/golden/ref_preint.npz, written by tests/golden/make_ref_preint.py), which needs nothing but the file."""
import importlib.util
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "ref_preint.npz")


@pytest.fixture(scope="module")
def product_preint(tmp_path_factory):
    """-> f(list of case dicts) -> (n, 20) array: the product's PreInt on each stream (dp, dq, dv, sum_dt, J_R_bg row-major)"""
    exe = str(tmp_path_factory.mktemp("preint") / "preint_dump")
    cxx = shutil.which("g++") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = ["-O2", "-std=c++17", "-ffp-contract=off", "-w", "-x", "c++"] if cxx.endswith("g++") else ["-O2", "-std=c++17", "-ffp-contract=off", "-w", "-x", "hip", "--offload-arch=gfx950"]
    subprocess.check_call([cxx] + flags + [os.path.join(ROOT, "tests", "host", "preint_dump.hip"), "-o", exe])

    def run(cs):
        path = exe + ".in"
        with open(path, "w") as f:
            f.write("%d\n" % len(cs))
            for c in cs:
                f.write(" ".join("%.17g" % x for x in np.concatenate([c["acc0"], c["gyr0"], c["ba"], c["bg"]])) + "\n%d\n" % len(c["dt"]))
                for i in range(len(c["dt"])):
                    f.write("%.17g %s %s\n" % (c["dt"][i], " ".join("%.17g" % x for x in c["acc"][i]), " ".join("%.17g" % x for x in c["gyr"][i])))
                rb = c["rebias"]
                f.write(" ".join("%.17g" % x for x in ([1.0] + list(rb[0]) + list(rb[1]) if rb is not None else [0.0] * 7)) + "\n")
        out = subprocess.run([exe, path], capture_output=True, text=True, check=True, timeout=60).stdout
        return np.array([[float(x) for x in l.split()] for l in out.strip().splitlines()])
    return run


def _oracle(c):
    from oracle import dyn_init as D
    p = D.PreInt(c["acc0"], c["gyr0"], c["bg"])
    for i in range(len(c["dt"])):
        p.push_back(float(c["dt"][i]), c["acc"][i], c["gyr"][i])
    if c["rebias"] is not None:
        p.repropagate(c["rebias"][1])
    return np.concatenate([p.dp, p.dq, p.dv, [p.sum_dt], np.asarray(p.J).ravel()])


def _ref_vec(r):
    return np.concatenate([r["dp"], r["dq"], r["dv"], [r["sum_dt"]], r["dq_dbg"].ravel()])


def _worst(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1.0))


def test_both_preintegrations_against_the_compiled_reference(product_preint):
    from oracle import lvref
    if not lvref.preint_available():
        pytest.skip("oracle/_ref/liblvref_preint.so not built and /root/reference absent")
    spec = importlib.util.spec_from_file_location("make_ref_preint", os.path.join(ROOT, "tests", "golden", "make_ref_preint.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    cs = list(gen.cases(11, 40))                                                     # other streams than the committed fixture's
    P = product_preint(cs)
    w_prod = w_ora = 0.0
    for k, c in enumerate(cs):
        r = _ref_vec(lvref.preintegrate(c["acc0"], c["gyr0"], c["ba"], c["bg"], c["dt"], c["acc"], c["gyr"], c["rebias"]))
        w_prod = max(w_prod, _worst(P[k], r)); w_ora = max(w_ora, _worst(_oracle(c), r))
    print("pre-integration against the compiled reference, 40 streams: product %.1e, independent restatement %.1e" % (w_prod, w_ora))
    assert w_prod < 1e-12 and w_ora < 1e-12


def test_both_preintegrations_against_the_references_committed_outputs(product_preint):
    g = np.load(GOLDEN)
    cs = []
    for k in range(len(g["n"])):
        n = int(g["n"][k]); h = g["head"][k]; s = g["samples"][k, :n]; rb = g["rebias"][k]
        cs.append(dict(acc0=h[0:3], gyr0=h[3:6], ba=h[6:9], bg=h[9:12], dt=s[:, 0], acc=s[:, 1:4], gyr=s[:, 4:7], rebias=(rb[1:4], rb[4:7]) if rb[0] else None))
    P = product_preint(cs)
    w_prod = w_ora = 0.0
    for k, c in enumerate(cs):
        r = g["out"][k][:20]
        w_prod = max(w_prod, _worst(P[k], r)); w_ora = max(w_ora, _worst(_oracle(c), r))
    print("pre-integration against tests/golden/ref_preint.npz, %d streams: product %.1e, independent restatement %.1e" % (len(cs), w_prod, w_ora))
    assert w_prod < 1e-12 and w_ora < 1e-12
