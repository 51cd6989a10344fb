"""Generates tests/golden/ref_dyninit_rest.npz from the REFERENCE'S OWN moving-start initialiser (as make_ref_dyninit.py: the sources of
/root/reference compiled in place into oracle/_ref/liblvref_dyninit.so, stand-in minimisers).  The outputs stored here are NOT the oracle's.
Needs /root/reference; run from the repo root:
    python tests/golden/make_ref_dyninit_rest.py
The start is case 20 of tools/gpu/fuzz_whole_program.py: an equidistant 512 x 512 camera publishing at 10 Hz FROM REST, whose static
initialiser does not fire - the moving-start initialiser sees windows that have barely moved (far points without depth curvature: the
bundle adjustment's valley), refuses the first ones and gets through at message 19.  The messages are the tracker's (the oracle's
ImageProcessor = the reference's, byte for byte, tests/test_oracle_ref_imgproc.py) on the rendered synthetic sequence; stored: messages
0..21, the IMU samples up to them, and what DynamicInitializer::assignInitialState handed over."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools", "gpu")); sys.path.insert(0, os.path.join(ROOT, "examples"))
from oracle import lvref, lvo  # noqa: E402
from larvio_amd import synthetic as S  # noqa: E402
from tests.conftest import synth_frames  # noqa: E402
import fuzz_whole_program as F  # noqa: E402

cam, n, fo, bo, first = F.draw(20)
assert first == 0 and fo["pub_frequency"] == 10 and cam["distortion_model"] == 1
frames = synth_frames(first, 46, cam=cam)
seq = S.imu_only_sequence(cam=cam); ts = [f[0] for f in frames]
imu = seq.imu_array(max(int(ts[0] * 200) - 4, 0), int(ts[-1] * 200) + 40)
fe = lvo.Frontend(S.frontend_config(cam=cam, **fo)); msgs = []
for t, img in frames:
    have, m = fe.process(img, t, imu[:int(np.count_nonzero(imu["t"] - t < 0.05))][-60:])
    if have:
        msgs.append((t, m))
msgs = msgs[:22]
T = np.asarray(cam["T_cam_imu"], float); R_b2c = T[:3, :3]; t_c_b = -R_b2c.T @ T[:3, 3]
r = lvref.dynamic_init(msgs, imu, R_b2c, t_c_b)
assert r is not None
hi = int(np.searchsorted(imu["t"], msgs[-1][0] + 0.06))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_dyninit_rest.npz"), R_b2c=R_b2c, t_c_b=t_c_b, imu=imu[:hi],
                    msg_ts=np.array([m[0] for m in msgs]), msg_len=np.array([len(m[1]) for m in msgs]), msg_obs=np.concatenate([m[1] for m in msgs]),
                    message=r["message"], state_time=r["state_time"], erase=r["erase"], q=r["q"], v=r["v"], bg=r["bg"], g=r["g"])
print("ref_dyninit_rest.npz written: success at message", r["message"], "state time", r["state_time"], "erase", r["erase"], "v", r["v"])
