"""Generates tests/golden/ref_orb.npz from the REFERENCE's own ORB descriptor code - /root/reference/src/ORBDescriptor.cpp compiled
in place into oracle/_ref/liblvref_orb.so (oracle/Makefile target `ref`; OpenCV calls served by the stand-ins of oracle/ref_shim/).
Unlike the other fixtures of this directory, the outputs stored here are NOT the oracle's: they are what the reference's text computes
(sampling pattern, umax table, mosaic layout, IC angle, rotated BRIEF, Hamming distance).  Needs /root/reference; run from the repo root:
    python tests/golden/make_ref_orb.py
The input frame is synthetic (seeded), equalised by the oracle's CLAHE only to give it the front-end's contrast; points include the
cvRound half-way cases (x.5) and positions next to the image border, where the descriptor samples the mosaic's frame."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lvo, lvref  # noqa: E402
from scipy import ndimage  # noqa: E402

rng = np.random.default_rng(20260925)
W, H, PAD = 376, 240, 21
base = ndimage.zoom(rng.uniform(0, 255, (32, 49)), 8, order=3)[:H, :W]
img = lvo.clahe(np.clip(base + rng.normal(0, 5, base.shape), 0, 255).astype(np.uint8))
pyr = lvo.LkPyramid(img)
padded = pyr.image(0, True).copy()                    # level 0 + its reflect-101 frame: the buffer the reference's image view sits in
ref = lvref.RefOrb(padded, PAD, 2)
n = 600
pts = np.stack([rng.uniform(0, W - 1, n), rng.uniform(0, H - 1, n)], 1).astype(np.float32)
pts[:120] = np.floor(pts[:120]) + 0.5                                   # half-way cases of cvRound (ties to even)
pts[120:160, 0] = rng.uniform(0, 3, 40); pts[160:200, 1] = rng.uniform(H - 4, H - 1, 40)      # patches that reach into the mosaic's border
desc, ang = ref.describe(pts)
ext, blur = ref.planes()
a = rng.integers(0, 256, (64, 32), dtype=np.uint8); b = rng.integers(0, 256, (64, 32), dtype=np.uint8)
ham = np.array([lvref.hamming(a[i], b[i]) for i in range(64)], np.int32)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_orb.npz"), img=img, pad=PAD, pts=pts, desc=desc, angle=ang,
                    ext_rows=ext[[0, 17, 31, 32, 150, H + 32, H + 63]], blur_rows=blur[[32, 33, 150, H + 31]],
                    ext_sum=np.int64(ext.astype(np.int64).sum()), blur_sum=np.int64(blur.astype(np.int64).sum()),
                    umax=ref.umax(), pattern=ref.pattern().astype(np.int8), ham_a=a, ham_b=b, ham=ham)
print("ref_orb.npz written:", n, "points,", int(np.unpackbits(desc).sum()), "descriptor bits set")
