"""Generates tests/golden/*.npz from the CPU oracle (the reference itself cannot run in this environment and has no
fixtures of its own: SURVEY.md §8c "parity unpinned").  The fixtures freeze the oracle's arithmetic so that any later
edit that changes a byte is caught by tests/test_oracle_frontend.py::test_golden_fixtures and by the GPU parity tests.
Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lvo, lvo_be  # noqa: E402
from scipy import ndimage  # noqa: E402

rng = np.random.default_rng(20260924)
base = ndimage.zoom(rng.uniform(0, 255, (40, 52)), 4, order=3)[:150, :200]
img0 = np.clip(base + rng.normal(0, 4, base.shape), 0, 255).astype(np.uint8)
img1 = np.clip(ndimage.shift(base, (0.8, -1.7), order=3, mode="reflect") + rng.normal(0, 4, base.shape), 0, 255).astype(np.uint8)
p0 = lvo.LkPyramid(lvo.clahe(img0)); p1 = lvo.LkPyramid(lvo.clahe(img1))
pts = p0.good_features(40, 0.01, 10.0)
out, st, it = lvo.lk_track(p0, p1, pts, pts)
e, b = p0.orb_prepare()
d, a = lvo.orb_describe(e, b, pts)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_frontend_stages import _two_view  # noqa: E402
x1, x2 = _two_view(120, 0.25, 77)
ok, mask, iters = lvo.ransac_fundamental(x1, x2)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frontend_small.npz"), img0=img0, img1=img1, clahe0=lvo.clahe(img0),
                    lvl2=p0.image(2), der1=p0.deriv(1), corners=pts, lk_status=st, lk_pts=out, desc=d, x1=x1, x2=x2,
                    ransac_mask=mask, ransac_iters=iters)
print("frontend_small.npz written:", len(pts), "corners,", int(st.sum()), "tracked,", int(mask.sum()), "ransac inliers")
