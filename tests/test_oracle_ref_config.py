"""The configuration file, read by the product's loaders and by the REFERENCE'S OWN: LarVio::loadParameters (larvio.cpp:58-311) and
ImageProcessor::loadParameters (image_processor.cpp:44-113) of the reference compiled in place (oracle/_ref/), on the configuration
files the reference ships (config/euroc.yaml, config/mynteye.yaml - read where they lie, so this test runs only where /root/reference
exists) - against include/lvk_config.hpp (the C++ loader behind the adapter's `ImageProcessor(config_file)` / `LarVio(config_file)`,
through examples/host_tools) and larvio_amd/config.py (its Python twin).  What is compared is what each side makes of every key the
hot path reads: names, types, the squares the reference takes of its noise parameters, the transposes and inversions of T_cam_imu.
cv::FileStorage itself is a stand-in (oracle/ref_shim2/, ref_shim3/): OpenCV's parser is not what is pinned, the reference's use of
the file is."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CONFIG = "/root/reference/config"
TOOL = os.path.join(ROOT, "examples", "host_tools")


def _shipped(name, tmp_path):
    """the shipped file with its output_dir pointed at a directory that exists (the reference opens two log files there)"""
    src = open(os.path.join(REF_CONFIG, name)).read()
    out = str(tmp_path / "out") + "/"; os.makedirs(out, exist_ok=True)
    txt, n = re.subn(r'output_dir:\s*"[^"]*"', 'output_dir: "%s"' % out, src)
    assert n == 1
    p = str(tmp_path / name); open(p, "w").write(txt)
    return p


def _host_tools_fields(path):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "examples"), "host_tools"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([TOOL, "config", path], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    got = {}
    for line in r.stdout.splitlines():
        k, _, v = line.partition(" ")
        got[k.strip()] = v.split()
    return got


@pytest.mark.parametrize("name", ["euroc.yaml", "mynteye.yaml"])
def test_shipped_configuration_files_product_loaders_against_the_references_own(name, tmp_path):
    from oracle import lvref, lvo_be
    if not (os.path.isdir(REF_CONFIG) and lvref.larvio_available() and lvref.imgproc_available()):
        pytest.skip("needs /root/reference and oracle/_ref (make -C oracle ref)")
    from larvio_amd.config import load_config
    path = _shipped(name, tmp_path)
    ref_be = lvref.RefLarVio.from_yaml(path).params()
    ref_fe = lvref.RefImageProcessor.from_yaml(path).params()
    cpp = _host_tools_fields(path)
    fcfg, bcfg = load_config(path)[:2]
    # ---- the filter's side
    alias = dict(fx=("intrinsics", 0), fy=("intrinsics", 1), cx=("intrinsics", 2), cy=("intrinsics", 3))
    checked = 0
    for k in lvref.RefLarVio.PARAMS:
        want = ref_be[k]
        if k in alias:
            field, i = alias[k]; got_cpp = float(cpp["ekf." + field][i])
            v = bcfg[field]; got_py = float(v[i] if not isinstance(v, dict) else v[k])
        else:
            got_cpp = float(cpp["ekf." + k][0]); got_py = float(bcfg[k])
        assert abs(got_cpp - want) <= 2e-16 * abs(want) and abs(got_py - want) <= 2e-16 * abs(want), (k, want, got_cpp, got_py)     # (the reference keeps its noises squared: one rounding)
        checked += 1
    assert checked == 43
    T = np.array([float(x) for x in cpp["ekf.T_cam_imu"]]).reshape(4, 4)
    assert np.array_equal(T, np.asarray(bcfg["T_cam_imu"], float).reshape(4, 4))
    # what the reference derives from that matrix (larvio.cpp:232-247) against what the oracle's filter starts with (the HIP filter is held to it on the GPU)
    st = lvo_be.Ekf(dict(bcfg)).state()
    R_o = np.asarray(st["R_b2c"] if "R_b2c" in st else st["R_imu_cam"]).reshape(3, 3); t_o = np.asarray(st["t_c_b"] if "t_c_b" in st else st["t_cam_imu"])
    assert np.abs(R_o - ref_be["R_imu_cam0"]).max() < 1e-15 and np.abs(t_o - ref_be["t_cam0_imu"]).max() < 1e-15
    # ---- the front-end's side
    for k in lvref.RefImageProcessor.FE_PARAMS:
        want = ref_fe[k]
        got_py = fcfg[k]
        if k == "distortion_model":
            got_py = {"radtan": 0, "equidistant": 1}.get(got_py, got_py)
        assert float(cpp["fe." + k][0]) == want and float(got_py) == want, (k, want, cpp["fe." + k], fcfg[k])
    for k in ("intrinsics", "distortion"):
        assert np.array_equal(np.array([float(x) for x in cpp["fe." + k]]), ref_fe[k]), k
    assert np.array_equal(np.array([float(x) for x in cpp["fe.R_cam_imu"]]).reshape(3, 3), ref_fe["R_cam_imu"])          # the transpose (image_processor.cpp:90-93)
