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
