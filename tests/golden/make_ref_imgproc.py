"""Generates tests/golden/ref_imgproc.npz from the REFERENCE'S OWN FRONT-END CLASS - /root/reference/src/image_processor.cpp
(ImageProcessor::processImage) with src/ORBDescriptor.cpp compiled in place into oracle/_ref/liblvref_imgproc.so (oracle/Makefile target
`ref`; behind OpenCV's image-algorithm names stand the oracle's restatements, see oracle/ref_shim3/lvref_cv3.hpp - what the reference
contributes is everything around them).  The outputs stored here are NOT the oracle's front-end object's.
Needs /root/reference; run from the repo root:
    python tests/golden/make_ref_imgproc.py
Stream: 26 frames of 240 x 180, crops of one stored texture walking 2 x 1 pixels per frame, three featureless frames at the start and two
in the middle (the bootstrap waits; later every track is lost and ids continue with new corners), 60 tracks, min_distance 12.
Per frame: processImage's answer, image_state, the tracks the next frame will find (ids, lifetimes, points, descriptors), the new corners,
the feature message's bytes."""
import os
import sys
import tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lvref  # noqa: E402
from tests.test_gpu_frontend_edge import _texture, _walk  # noqa: E402
from tests.test_oracle_ref_imgproc import fixture_stream  # noqa: E402

W, H = 240, 180
tex = _texture(21, H + 60, W + 90)
walk = _walk(21, start=(10, 8))
offsets = [(-1, -1)] * 3 + walk[:9] + [(-1, -1)] * 2 + walk[9:]
z = dict(texture=tex, w=W, h=H, offsets=np.array(offsets, np.int32), max_features_num=60, min_distance=12)
frames, ts_all, imu_all, cfg = fixture_stream(z)
ref = lvref.RefImageProcessor(cfg, tempfile.mkdtemp())
have, state, trk_off, new_off, msg_off = [], [], [0], [0], [0]
ids, life, pts, desc, newp, msgs = [], [], [], [], [], []
for ts, img in zip(ts_all, frames):
    imu = imu_all[(imu_all["t"] < ts + 0.05)][-60:]
    h, m = ref.process(img, ts, imu)
    t = ref.tracks(); n = ref.new_pts()
    have.append(int(h)); state.append(ref.state)
    ids.append(t["ids"]); life.append(t["lifetime"]); pts.append(t["pts"]); desc.append(t["desc"]); newp.append(n)
    b = np.frombuffer(m.tobytes(), np.uint8) if h else np.zeros(0, np.uint8); msgs.append(b)
    trk_off.append(trk_off[-1] + len(t["ids"])); new_off.append(new_off[-1] + len(n)); msg_off.append(msg_off[-1] + len(b))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_imgproc.npz"), have=np.array(have), state=np.array(state),
                    trk_off=np.array(trk_off), new_off=np.array(new_off), msg_off=np.array(msg_off), trk_ids=np.concatenate(ids), trk_life=np.concatenate(life),
                    trk_pts=np.concatenate(pts).astype(np.float32).reshape(-1, 2), trk_desc=np.concatenate(desc).reshape(-1, 32), new_pts=np.concatenate(newp).astype(np.float32).reshape(-1, 2),
                    msg=np.concatenate(msgs), **z)
print("ref_imgproc.npz written:", len(frames), "frames, states", state, "messages", sum(have), "tracks per frame", np.diff(trk_off).tolist())
