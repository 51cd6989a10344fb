"""Host arithmetic of the product that runs on the CPU by design (larvio.cpp:520-578: the IMU samples of a frame are integrated on
the host and composed into one (Phi, Q) pair that a single kernel applies).  The product's composition skips structurally-zero
blocks and nests its loops for vectorisation; tests/host/imu_compose_check.hip recompiles backend.hip's host code (no GPU is touched)
and requires it to equal the dense recurrences bit for bit, with and without IMU-intrinsics calibration (L = 22 / 46)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_host_check(tmp_path, name):
    """compile tests/host/<name>.hip (it #includes backend.hip: host code only is exercised) against the built objects and run it"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    csrc = os.path.join(ROOT, "larvio_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "-j8", "-s"])                      # the objects the check links against (no-op when built)
    obj = str(tmp_path / (name + ".o")); exe = str(tmp_path / name)
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-Xarch_host", "-mavx2", "-w", "-I", csrc]
    subprocess.check_call([hipcc] + flags + ["-c", os.path.join(ROOT, "tests", "host", name + ".hip"), "-o", obj])
    objs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".o") and f != "backend.o"]
    subprocess.check_call([hipcc, "--offload-arch=gfx950", obj] + objs + ["-pthread", "-o", exe])
    return subprocess.run([exe], capture_output=True, text=True, timeout=120)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc (host compile only)")
def test_imu_composition_equals_dense_recurrence_bit_for_bit(tmp_path):
    r = _run_host_check(tmp_path, "imu_compose_check")
    assert r.returncode == 0 and r.stdout.count("ok L=") == 2, r.stdout + r.stderr


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc (host compile only)")
def test_observation_lists_and_map_cursor_against_std_map_models(tmp_path):
    """Feature::find / set / erase (sorted vector, newest-first + bisection) and add_observations' cursor walk of the ordered feature map
    (incl. out-of-order messages and features erased in between) against plain std::map models over random operation sequences."""
    r = _run_host_check(tmp_path, "feature_obs_check")
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc (host compile only)")
def test_early_erase_count_equals_the_sequential_count_whenever_it_is_taken(tmp_path):
    """lvk_vio_pipe_submit's early IMU erase count (same count for td - 0.5 ms and td + 0.5 ms => taken at once from the last published
    td) against the sequential rule (the count with the td the update starts from, larvio.cpp:464-517) on jittered IMU streams with a
    random phase against the image grid and a random-walk td: equal whenever it is taken, the state time it leaves behind too, waiting
    frames in the share the geometry predicts, and never on the benchmark's grid-aligned input."""
    r = _run_host_check(tmp_path, "erase_count_check")
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr
    print(r.stdout.strip())


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc (host compile only)")
def test_moving_start_initialiser_blocks_and_whole_on_closed_form_cases(tmp_path):
    """be_init.h (DynamicInitializer.cpp and what it takes from OpenCV / Ceres, restated on the host): 3x3 SVD, 8-point + recoverPose,
    PnP, the window's bundle adjustment and the pre-integration's bias Jacobian on closed-form cases; then the whole initialiser on a
    simulated moving start with exact measurements (gravity direction < 2e-3, body velocity < 2 cm/s, gyro bias < 2e-4 rad/s, state
    time and IMU erase count)."""
    r = _run_host_check(tmp_path, "init_check")
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr
    print(r.stdout.strip())


def test_product_host_math_against_scipy(tmp_path):
    """be_host_math.h (quaternion / 3x3 helpers of the product's host side; Hamilton [x y z w] as math_utils.hpp:54-102, Eigen's
    Quaternion <-> matrix formulas) against scipy / numpy on 200 random inputs incl. the trace <= 0 branches of the matrix -> quaternion
    conversion - and, where oracle/_ref is built, against the reference's own math_utils.hpp compiled in place (skewSymmetric, both small-angle
    quaternions, quaternionToRotation, rotationToQuaternion, quaternionMultiplication).  These helpers share their text with
    oracle/be_math.h: this is their pin that does not pass through the oracle."""
    import json
    import numpy as np
    from scipy.spatial.transform import Rotation
    cxx = shutil.which("g++") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "host_math_dump")
    flags = ["-O2", "-std=c++17", "-ffp-contract=off", "-w", "-x", "c++"] if cxx.endswith("g++") else ["-O2", "-std=c++17", "-ffp-contract=off", "-w", "-x", "hip", "--offload-arch=gfx950"]
    subprocess.check_call([cxx] + flags + [os.path.join(ROOT, "tests", "host", "host_math_dump.hip"), "-o", exe])
    rows = [json.loads(l) for l in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines()]
    assert len(rows) == 200
    from oracle import lvref
    ref = lvref if lvref.feature_available() else None
    saw_neg_trace = 0
    for r in rows:
        q, p, w = np.array(r["q"]), np.array(r["p"]), np.array(r["w"]); A, B = np.reshape(r["A"], (3, 3)), np.reshape(r["B"], (3, 3)); v = np.array(r["v"])
        R = Rotation.from_quat(q).as_matrix()                                      # scipy: scalar-last, active rotation - Eigen's convention
        assert np.abs(np.reshape(r["R"], (3, 3)) - R).max() < 1e-14
        saw_neg_trace += np.trace(R) <= 0
        q2 = np.array(r["q2"]); assert min(np.abs(q2 - q).max(), np.abs(q2 + q).max()) < 1e-13 and abs(np.linalg.norm(q2) - 1) < 1e-14
        qp = (Rotation.from_quat(q) * Rotation.from_quat(p / np.linalg.norm(p))).as_quat() * np.linalg.norm(p)      # Hamilton product, q then scaled p
        got = np.array(r["qp"]); assert min(np.abs(got - qp).max(), np.abs(got + qp).max()) < 1e-13
        assert np.abs(np.reshape(r["AB"], (3, 3)) - A @ B).max() < 1e-14 and np.array_equal(np.reshape(r["At"], (3, 3)), A.T)
        assert np.abs(np.array(r["Av"]) - A @ v).max() < 1e-14 and np.abs(np.array(r["Atv"]) - A.T @ v).max() < 1e-14
        assert np.abs(np.reshape(r["S"], (3, 3)) @ v - np.cross(w, v)).max() < 1e-14 and abs(r["n"][0] - np.linalg.norm(v)) < 1e-15
        d = w / 2; n2 = d @ d                                                       # smallAngleQuaternion (math_utils.hpp:85-102)
        dq = np.concatenate([d, [np.sqrt(1 - n2)]]) if n2 <= 1 else np.concatenate([d, [1.0]]) / np.sqrt(1 + n2)
        assert np.abs(np.array(r["dq"]) - dq).max() < 1e-15
        if ref is not None:                                                         # ... and against the reference's own math_utils.hpp compiled in place (oracle/_ref)
            S_r, qa, qb = ref.math_small_angle(w)
            assert np.array_equal(np.reshape(r["S"], (3, 3)), S_r) and np.abs(np.array(r["dq"]) - qa).max() < 1e-15 and np.abs(np.array(r["dq"]) - qb).max() < 1e-15
            R_r, q2_r, qp_r = ref.math_quat(q, p)                                   # quaternionToRotation / rotationToQuaternion / quaternionMultiplication
            assert np.abs(np.reshape(r["R"], (3, 3)) - R_r).max() < 1e-15
            assert min(np.abs(q2 - q2_r).max(), np.abs(q2 + q2_r).max()) < 1e-13     # two branch rules (Eigen's, the reference's own), one rotation
            gn = got / np.linalg.norm(got); assert min(np.abs(gn - qp_r).max(), np.abs(gn + qp_r).max()) < 1e-14
    assert saw_neg_trace >= 20


def test_descriptor_rotation_cosf_sinf_restatement_is_libms_on_every_angle(tmp_path):
    """larvio_amd/csrc/lvk_sincosf.h - what the HIP descriptor kernel computes for ORBDescriptor.cpp:343's `(float)cos(angle),
    (float)sin(angle)` (std::cos(float) / std::sin(float) = libm's cosf / sinf under `using namespace std`) - compiled as host code and
    compared with this host's libm on every float in [0, 6.2832]: 1,086,918,650 angles, bit for bit."""
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("needs g++")
    exe = str(tmp_path / "sincosf_check")
    subprocess.check_call([gxx, "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-I", os.path.join(ROOT, "larvio_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", "sincosf_check.cpp"), "-o", exe, "-lm"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok 1086918650"), r.stdout + r.stderr
