"""Reader for LARVIO configuration files (the OpenCV-FileStorage YAML 1.0 subset of config/euroc.yaml) — the Python twin of
include/lvk_config.hpp, so that ``ImageProcessor("euroc.yaml")`` / ``LarVio("euroc.yaml")`` work like the reference's constructors
(image_processor.cpp:28-33,44-113; larvio.cpp:40-44,58-311).  Host-only; tests/test_host_tools.py holds it to the C++ reader."""
import re

import numpy as np

_NUM = re.compile(r"^[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?)$")


def _strip_comment(line):
    q = None
    for i, ch in enumerate(line):
        if q:
            if ch == q:
                q = None
        elif ch in "\"'":
            q = ch
        elif ch == "#" and (i == 0 or line[i - 1] in " \t"):
            return line[:i]
    return line


def parse(text):
    """-> dict: 'key' / 'parent.key' -> str, float or list of floats (flow sequences)"""
    out, parent, seq_key, seq_txt = {}, "", None, ""

    def store_seq(key, txt):
        a, b = txt.find("["), txt.rfind("]")
        out[key] = [float(x) for x in txt[a + 1:b].replace(",", " ").split()]
    for raw in text.splitlines():
        line = _strip_comment(raw).rstrip()
        if seq_key is not None:
            seq_txt += " " + line
            if "]" in line:
                store_seq(seq_key, seq_txt); seq_key = None
            continue
        s = line.strip()
        if not s or s[0] == "%" or s.startswith("---") or s.startswith("..."):
            continue
        indent = len(line) - len(line.lstrip(" \t"))
        if s[0] == "[" and parent:
            seq_key, seq_txt = parent + ".data", s
            if "]" in s:
                store_seq(seq_key, seq_txt); seq_key = None
            continue
        m = re.match(r"^([^:\[]+):(\s+(.*))?$", s)
        if not m:
            raise ValueError("expected 'key: value': " + raw)
        key, val = m.group(1).strip(), (m.group(3) or "").strip()
        if indent == 0:
            parent = ""
        elif parent:
            key = parent + "." + key
        if not val or val.startswith("!!"):
            if indent == 0:
                parent = key
            out[key] = val
        elif val[0] == "[":
            if "]" in val:
                store_seq(key, val)
            else:
                seq_key, seq_txt = key, val
        else:
            if len(val) >= 2 and val[0] in "\"'" and val[-1] == val[0]:
                out[key] = val[1:-1]
            else:
                out[key] = float(val) if _NUM.match(val) else val
    if seq_key is not None:
        raise ValueError("unterminated '[' of " + seq_key)
    return out


def _num(d, k):
    v = d.get(k, 0.0)
    return float(v) if not isinstance(v, str) else 0.0


def _int(d, k):
    return int(round(_num(d, k)))


def load_config(path):
    """-> (frontend_config dict, backend_config dict, output_dir) with the keys larvio_amd.synthetic.frontend_config /
    backend_config use; raises ValueError for settings the library does not implement (as lvk::load_*_config refuses them)"""
    with open(path) as f:
        d = parse(f.read())
    rows, cols, data = _int(d, "T_cam_imu.rows"), _int(d, "T_cam_imu.cols"), d.get("T_cam_imu.data")
    if rows != 4 or cols != 4 or not isinstance(data, list) or len(data) != 16:
        raise ValueError("T_cam_imu is not a 4x4 matrix")
    T = np.array(data, np.float64).reshape(4, 4)
    model = d.get("distortion_model", "")
    if model not in ("radtan", "equidistant"):
        raise ValueError(f"distortion_model '{model}' (radtan and equidistant are supported)")
    intr = tuple(_num(d, "intrinsics." + k) for k in ("fx", "fy", "cx", "cy"))
    fe = dict(width=_int(d, "resolution_width"), height=_int(d, "resolution_height"), pyramid_levels=_int(d, "pyramid_levels"),
              patch_size=_int(d, "patch_size"), max_iteration=_int(d, "max_iteration"), track_precision=_num(d, "track_precision"),
              max_features_num=_int(d, "max_features_num"), min_distance=_int(d, "min_distance"), flag_equalize=1 if _int(d, "flag_equalize") else 0,
              pub_frequency=_int(d, "pub_frequency"), distortion_model=0 if model == "radtan" else 1, intrinsics=intr,
              distortion=tuple(_num(d, "distortion_coeffs." + k) for k in ("k1", "k2", "p1", "p2")), R_cam_imu=T[:3, :3].T.copy())
    be = dict(if_fej=1 if _int(d, "if_FEJ") else 0, estimate_extrin=1 if _int(d, "estimate_extrin") else 0, estimate_td=1 if _int(d, "estimate_td") else 0,
              if_zupt_valid=1 if _int(d, "if_ZUPT_valid") else 0, sw_size=_int(d, "sw_size"), max_track_len=_int(d, "max_track_len"),
              least_observation_number=_int(d, "least_observation_number"), max_features_in_one_grid=_int(d, "max_features_in_one_grid"),
              aug_grid_rows=_int(d, "aug_grid_rows"), aug_grid_cols=_int(d, "aug_grid_cols"), pub_frequency=_num(d, "pub_frequency"),
              imu_rate=_num(d, "imu_rate"), width=fe["width"], height=fe["height"], intrinsics=intr, T_cam_imu=T, td=_num(d, "td"),
              feature_idp_dim=_int(d, "feature_idp_dim"), use_schmidt=1 if _int(d, "use_schmidt") else 0,
              calib_imu_instrinsic=1 if _int(d, "calib_imu_instrinsic") else 0, max_features=fe["max_features_num"])
    for k in ("noise_gyro", "noise_acc", "noise_gyro_bias", "noise_acc_bias", "noise_feature", "initial_covariance_orientation",
              "initial_covariance_velocity", "initial_covariance_position", "initial_covariance_gyro_bias", "initial_covariance_acc_bias",
              "initial_covariance_extrin_rot", "initial_covariance_extrin_trans", "rotation_threshold", "translation_threshold",
              "tracking_rate_threshold", "feature_translation_threshold", "zupt_max_feature_dis", "zupt_noise_v", "zupt_noise_p", "zupt_noise_q",
              "static_duration"):
        be[k] = _num(d, k)
    if be["feature_idp_dim"] != 1:
        raise ValueError("feature_idp_dim must be 1 (the 3-D inverse-depth parametrisation is not implemented)")
    if be["use_schmidt"]:
        raise ValueError("use_schmidt must be 0 (the Schmidt variant is not implemented)")
    out_dir = d.get("output_dir", "")
    return fe, be, out_dir if isinstance(out_dir, str) else ""
