"""The oracle's ORB block against the REFERENCE ITSELF: /root/reference/src/ORBDescriptor.cpp + include/ORB/ORBDescriptor.h compiled in
place (oracle/Makefile target `ref` -> oracle/_ref/liblvref_orb.so; OpenCV is not installed, so cv::Mat / copyMakeBorder / GaussianBlur /
fastAtan2 / cvRound are the stand-ins of oracle/ref_shim/lvref_cv.hpp).  What these tests pin to the reference's own text, bit for bit:
the sampling pattern (ORBDescriptor.cpp:27-285), the umax table (:313-328), the mosaic layout and border (:418-484), IC_Angle (:486-514),
the rotated-BRIEF descriptor (:334-383, incl. whichever cos/sin overload the reference's `cos(angle)` resolves to), computeDescriptors
(:385-416) and the Hamming distance (ORBDescriptor.h:43-59).  The first half runs the compiled reference live (here, where /root/reference
exists, or wherever the prebuilt library travelled); the second half holds the oracle to the committed outputs of the reference
(tests/golden/ref_orb.npz, written by tests/golden/make_ref_orb.py), which needs nothing but the file."""
import os

import numpy as np
import pytest

from oracle import lvo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_orb.npz")


def _ref():
    from oracle import lvref
    if not lvref.available():
        pytest.skip("oracle/_ref/liblvref_orb.so not built and /root/reference absent")
    return lvref


def _frame(seed, w, h):
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    base = ndimage.zoom(rng.uniform(0, 255, (h // 8 + 2, w // 8 + 2)), 8, order=3)[:h, :w]
    return lvo.clahe(np.clip(base + rng.normal(0, 5, base.shape), 0, 255).astype(np.uint8)), rng


def test_pattern_and_umax_are_the_references():
    lvref = _ref()
    img, _ = _frame(1, 96, 80)
    pyr = lvo.LkPyramid(img)
    r = lvref.RefOrb(pyr.image(0, True).copy(), 21, 2)
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import re
    for inc in ("oracle/orb_pattern.inc", "larvio_amd/csrc/orb_pattern_dev.inc"):          # the oracle's table AND the product's
        txt = open(os.path.join(here, inc)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S); txt = re.sub(r"//[^\n]*", "", txt)
        body = txt[txt.index("{") + 1:txt.rindex("}")]
        vals = np.array([int(v) for v in re.findall(r"-?\d+", body)], np.int32)
        assert np.array_equal(vals, r.pattern()), inc
    assert list(r.umax()) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]


@pytest.mark.parametrize("w,h,pad", [(200, 150, 21), (376, 240, 21), (752, 480, 21), (131, 97, 15), (120, 90, 40)])
def test_mosaic_and_blur_equal_the_compiled_reference(w, h, pad):
    """level 0 of mImagePyramid / mBluredImagePyramid: non-isolated copyMakeBorder of a view into the LK buffer (the first `pad` border
    pixels are that buffer's), 7x7 sigma-2 blur of the image area; pad 40 > 32: the whole border comes from the parent"""
    lvref = _ref()
    img, _ = _frame(w * 7 + h, w, h)
    pyr = lvo.LkPyramid(img, win=pad) if pad != 21 else lvo.LkPyramid(img)
    r = lvref.RefOrb(pyr.image(0, True).copy(), pad, 2)
    e, b = pyr.orb_prepare(); er, br = r.planes()
    assert np.array_equal(e, er) and np.array_equal(b, br)


@pytest.mark.parametrize("w,h,n", [(200, 150, 3000), (752, 480, 20000)])
def test_descriptors_and_angles_equal_the_compiled_reference(w, h, n):
    lvref = _ref()
    img, rng = _frame(w + h, w, h)
    pyr = lvo.LkPyramid(img)
    r = lvref.RefOrb(pyr.image(0, True).copy(), 21, 2)
    e, b = pyr.orb_prepare()
    pts = np.stack([rng.uniform(0, w - 1, n), rng.uniform(0, h - 1, n)], 1).astype(np.float32)
    pts[: n // 5] = np.floor(pts[: n // 5]) + 0.5                         # cvRound ties
    pts[n // 5: n // 4, 0] = rng.uniform(0, 2, n // 4 - n // 5)           # patches reaching the frame
    d, a = lvo.orb_describe(e, b, pts); dr, ar = r.describe(pts)
    assert np.array_equal(a.view(np.uint32), ar.view(np.uint32))
    assert np.array_equal(d, dr)
    # and the distance the front-end gates on (TH 58): every pair of consecutive descriptors
    for i in range(0, 400, 2):
        assert lvo.hamming(d[i], d[i + 1]) == lvref.hamming(dr[i], dr[i + 1])


def test_oracle_equals_the_committed_reference_outputs():
    """no library needed: the reference's outputs as stored by tests/golden/make_ref_orb.py"""
    z = np.load(GOLDEN)
    img, pad = z["img"], int(z["pad"])
    h, w = img.shape
    pyr = lvo.LkPyramid(img)
    e, b = pyr.orb_prepare()
    assert int(e.astype(np.int64).sum()) == int(z["ext_sum"]) and int(b.astype(np.int64).sum()) == int(z["blur_sum"])
    assert np.array_equal(e[[0, 17, 31, 32, 150, h + 32, h + 63]], z["ext_rows"]) and np.array_equal(b[[32, 33, 150, h + 31]], z["blur_rows"])
    d, a = lvo.orb_describe(e, b, z["pts"])
    assert np.array_equal(d, z["desc"]) and np.array_equal(a.view(np.uint32), z["angle"].view(np.uint32))
    assert [lvo.hamming(x, y) for x, y in zip(z["ham_a"], z["ham_b"])] == list(z["ham"])
    assert pad == 21
