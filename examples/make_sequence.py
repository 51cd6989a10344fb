#!/usr/bin/env python3
"""Write the binary sequence file examples/larvio_main reads: configuration structs, IMU samples and frames of the synthetic
EuRoC-shaped sequence (larvio_amd/synthetic.py).  usage: examples/make_sequence.py out.bin [first_frame] [n_frames]"""
import ctypes as C
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def write_sequence(path, first=40, count=50, max_features=150, sw_size=15, frames=None):
    from larvio_amd import synthetic as S
    from larvio_amd.image_processor import make_fe_config
    from larvio_amd.larvio import make_ekf_config
    seq = S.imu_only_sequence()
    if frames is None:
        full = S.Sequence()
        frames = [full.frame(first + i) for i in range(count)]
    ts = [f[0] for f in frames]
    h, w = frames[0][1].shape
    imu = seq.imu_array(max(int(ts[0] * 200) - 4, 0), int(ts[-1] * 200) + 40)
    fcfg = make_fe_config(S.frontend_config(max_features_num=max_features))
    bcfg = make_ekf_config(S.backend_config(sw_size=sw_size, if_zupt_valid=0))
    init_frame = 1
    k = int(np.searchsorted(imu["t"], ts[init_frame], side="right")) - 1
    t0 = imu["t"][k]; tr = seq.traj
    R = tr.R_wb(t0); t_ = np.trace(R); s_ = np.sqrt(t_ + 1) * 2
    q = np.array([(R[2, 1] - R[1, 2]) / s_, (R[0, 2] - R[2, 0]) / s_, (R[1, 0] - R[0, 1]) / s_, 0.25 * s_])
    init = np.concatenate([[t0], q, tr.p_wb(t0), tr.vel(t0), np.zeros(3), np.zeros(3), imu["gyro"][k], imu["acc"][k]]).astype(np.float64)
    assert init.shape == (23,)
    with open(path, "wb") as f:
        f.write(b"LVKSEQ1\0")
        f.write(struct.pack("<5i", len(frames), w, h, len(imu), init_frame))
        f.write(bytes(fcfg)); f.write(bytes(bcfg)); f.write(init.tobytes())
        f.write(np.ascontiguousarray(imu).tobytes())
        for t, img in frames:
            f.write(struct.pack("<d", t)); f.write(np.ascontiguousarray(img, np.uint8).tobytes())
    return dict(ts=ts, imu=imu, init=init, init_frame=init_frame, fcfg=fcfg, bcfg=bcfg)


if __name__ == "__main__":
    out = sys.argv[1]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    write_sequence(out, first, count)
    print("wrote", out, os.path.getsize(out), "bytes")
