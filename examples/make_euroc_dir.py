#!/usr/bin/env python3
"""Write the synthetic EuRoC-shaped sequence (larvio_amd/synthetic.py) in the ASL directory layout the dataset driver
examples/larvio_euroc reads — mav0/imu0/data.csv, mav0/cam0/data.csv, mav0/cam0/data/<ns>.png, the ground truth in
mav0/state_groundtruth_estimate0/data.csv — plus a configuration file with the keys of the reference's config/euroc.yaml.

usage: examples/make_euroc_dir.py out_dir [first_frame] [n_frames]
then:  examples/larvio_euroc out_dir/mav0/imu0/data.csv out_dir/mav0/cam0/data.csv out_dir/mav0/cam0/data out_dir/config.yaml --tum traj.txt
       tools/traj_rmse.py traj.txt out_dir/mav0/state_groundtruth_estimate0/data.csv
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def stamp_ns(t):
    return int(round(float(t) * 1e9))


def write_config_yaml(path, fe, be, output_dir="", img_rate=20, distortion_model=None):
    """fe / be: the dicts of larvio_amd.synthetic.frontend_config / backend_config.  Keys and layout as OpenCV's FileStorage
    reads them (image_processor.cpp:44-113, larvio.cpp:58-311)."""
    T = np.asarray(be["T_cam_imu"], np.float64).reshape(4, 4)
    model = distortion_model or ("equidistant" if fe["distortion_model"] in (1, "equidistant") else "radtan")
    rows = ",\n     ".join(", ".join(repr(float(x)) for x in T[r]) for r in range(4))
    fx, fy, cx, cy = [float(x) for x in fe["intrinsics"]]
    k1, k2, p1, p2 = [float(x) for x in fe["distortion"]]
    g = lambda d, k: repr(float(d[k]))
    text = f"""%YAML:1.0

output_dir: "{output_dir}"

if_FEJ: {int(be['if_fej'])}
estimate_extrin: {int(be['estimate_extrin'])}
estimate_td: {int(be['estimate_td'])}
calib_imu_instrinsic: {int(be.get('calib_imu_instrinsic', 0))}

camera_model: "pinhole"
distortion_model: "{model}"
resolution_width: {int(fe['width'])}
resolution_height: {int(fe['height'])}
intrinsics:
   fx: {fx!r}
   fy: {fy!r}
   cx: {cx!r}
   cy: {cy!r}
distortion_coeffs:
   k1: {k1!r}
   k2: {k2!r}
   p1: {p1!r}
   p2: {p2!r}

T_cam_imu: !!opencv-matrix
   rows: 4
   cols: 4
   dt: d
   data:
    [{rows}]
td: {g(be, 'td')}

pyramid_levels: {int(fe['pyramid_levels'])}
patch_size: {int(fe['patch_size'])}
fast_threshold: 30
max_iteration: {int(fe['max_iteration'])}
track_precision: {g(fe, 'track_precision')}
ransac_threshold: 1
max_features_num: {int(fe['max_features_num'])}
min_distance: {int(fe['min_distance'])}
flag_equalize: {int(fe['flag_equalize'])}    # 0(false) or 1(true)
pub_frequency: {int(fe['pub_frequency'])}

sw_size: {int(be['sw_size'])}

position_std_threshold: 8.0
rotation_threshold: {g(be, 'rotation_threshold')}
translation_threshold: {g(be, 'translation_threshold')}
tracking_rate_threshold: {g(be, 'tracking_rate_threshold')}

least_observation_number: {int(be['least_observation_number'])}
max_track_len: {int(be['max_track_len'])}
feature_translation_threshold: {g(be, 'feature_translation_threshold')}

noise_gyro: {g(be, 'noise_gyro')}
noise_acc: {g(be, 'noise_acc')}
noise_gyro_bias: {g(be, 'noise_gyro_bias')}
noise_acc_bias: {g(be, 'noise_acc_bias')}
noise_feature: {g(be, 'noise_feature')}

initial_covariance_orientation: {g(be, 'initial_covariance_orientation')}
initial_covariance_velocity: {g(be, 'initial_covariance_velocity')}
initial_covariance_position: {g(be, 'initial_covariance_position')}
initial_covariance_gyro_bias: {g(be, 'initial_covariance_gyro_bias')}
initial_covariance_acc_bias: {g(be, 'initial_covariance_acc_bias')}
initial_covariance_extrin_rot: {g(be, 'initial_covariance_extrin_rot')}
initial_covariance_extrin_trans: {g(be, 'initial_covariance_extrin_trans')}

reset_fej_threshold: 10.11

if_ZUPT_valid: {int(be['if_zupt_valid'])}
zupt_max_feature_dis: {g(be, 'zupt_max_feature_dis')}
zupt_noise_v: {g(be, 'zupt_noise_v')}    # std
zupt_noise_p: {g(be, 'zupt_noise_p')}
zupt_noise_q: {g(be, 'zupt_noise_q')}

static_duration: {g(be, 'static_duration')}

imu_rate: {int(be['imu_rate'])}
img_rate: {int(img_rate)}

max_features_in_one_grid: {int(be['max_features_in_one_grid'])}
aug_grid_rows: {int(be['aug_grid_rows'])}
aug_grid_cols: {int(be['aug_grid_cols'])}
feature_idp_dim: 1

use_schmidt: 0
"""
    with open(path, "w") as f:
        f.write(text)


def write_euroc_dir(root, frames, imu, fe, be, ground_truth=None, output_dir=""):
    """frames: [(t, HxW uint8)], imu: structured array (t, gyro[3], acc[3]).  Returns the stamps (seconds, as the driver will read
    them back: 1e-9 * integer nanoseconds) of the frames and of the IMU samples."""
    from PIL import Image
    mav = os.path.join(root, "mav0")
    os.makedirs(os.path.join(mav, "imu0"), exist_ok=True)
    os.makedirs(os.path.join(mav, "cam0", "data"), exist_ok=True)
    ts_img, ts_imu = [], []
    with open(os.path.join(mav, "cam0", "data.csv"), "w", newline="") as f:
        f.write("#timestamp [ns],filename\r\n")                     # the EuRoC files have CRLF line ends
        for t, img in frames:
            ns = stamp_ns(t); ts_img.append(1e-9 * ns)
            f.write(f"{ns},{ns}.png\r\n")
            Image.fromarray(np.ascontiguousarray(img, np.uint8)).save(os.path.join(mav, "cam0", "data", f"{ns}.png"), compress_level=1)
    with open(os.path.join(mav, "imu0", "data.csv"), "w", newline="") as f:
        f.write("#timestamp [ns],w_RS_S_x [rad s^-1],w_RS_S_y [rad s^-1],w_RS_S_z [rad s^-1],a_RS_S_x [m s^-2],a_RS_S_y [m s^-2],a_RS_S_z [m s^-2]\r\n")
        for s in imu:
            ns = stamp_ns(s["t"]); ts_imu.append(1e-9 * ns)
            f.write(",".join([str(ns)] + [repr(float(x)) for x in s["gyro"]] + [repr(float(x)) for x in s["acc"]]) + "\r\n")
    if ground_truth is not None:
        os.makedirs(os.path.join(mav, "state_groundtruth_estimate0"), exist_ok=True)
        with open(os.path.join(mav, "state_groundtruth_estimate0", "data.csv"), "w", newline="") as f:
            f.write("#timestamp, p_RS_R_x [m], p_RS_R_y [m], p_RS_R_z [m], q_RS_w [], q_RS_x [], q_RS_y [], q_RS_z []\r\n")
            for t, p, q_wxyz in ground_truth:
                f.write(",".join([str(stamp_ns(t))] + [repr(float(x)) for x in p] + [repr(float(x)) for x in q_wxyz]) + "\r\n")
    write_config_yaml(os.path.join(root, "config.yaml"), fe, be, output_dir=output_dir)
    return np.array(ts_img), np.array(ts_imu)


def rot_to_quat_wxyz(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        return np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    i = int(np.argmax(np.diag(R))); j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
    q = np.zeros(4); q[1 + i] = 0.25 * s; q[0] = (R[k, j] - R[j, k]) / s; q[1 + j] = (R[j, i] + R[i, j]) / s; q[1 + k] = (R[k, i] + R[i, k]) / s
    return q


def synthetic_euroc_dir(root, first=0, count=120, max_features=150, sw_size=20, output_dir=""):
    from larvio_amd import synthetic as S
    seq = S.Sequence()
    frames = [seq.frame(first + i) for i in range(count)]
    ts = [f[0] for f in frames]
    imu = seq.imu_array(max(int(ts[0] * 200) - 4, 0), int(ts[-1] * 200) + 40)
    fe = S.frontend_config(max_features_num=max_features)
    be = S.backend_config(sw_size=sw_size)
    gt_t = np.arange(ts[0], ts[-1] + 0.0051, 0.005)
    gt = [(t, seq.traj.p_wb(t), rot_to_quat_wxyz(seq.traj.R_wb(t))) for t in gt_t]
    t_img, t_imu = write_euroc_dir(root, frames, imu, fe, be, ground_truth=gt, output_dir=output_dir)
    return dict(frames=frames, imu=imu, fe=fe, be=be, t_img=t_img, t_imu=t_imu)


if __name__ == "__main__":
    out = sys.argv[1]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    synthetic_euroc_dir(out, first, count)
    print("wrote", out)
