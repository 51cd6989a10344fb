"""The oracle's triangulation block against the REFERENCE ITSELF: /root/reference/include/larvio/feature.hpp compiled in place
(oracle/Makefile target `ref` -> oracle/_ref/liblvref_feature.so; Eigen is not installed, so Matrix / Isometry3d / Quaterniond / ldlt()
are the stand-ins of oracle/ref_shim/lvref_eigen.hpp: plain loops, no claim about Eigen's rounding).  What these tests pin to the
reference's own text: checkMotion (:334-381), the view selection and frame bookkeeping of initializePosition (:383-552),
initializePosition_AssignAnchor (:554-721) and initializeInvParamPosition (:723-890) - which observations enter, relative poses towards
the LAST view, the two-view initial guess or the stored position, the Levenberg-Marquardt schedule with its Huber weights, the
positive-depth and reprojection tests, the anchor, inverse depth and corrected anchor observation handed back, the quaternion
convention of orientation_cam.  The oracle's lvo_triangulate / lvo_check_motion take poses as rotation matrices: the tests build them
from the same quaternions with scipy, so a Hamilton/JPL or transpose mix-up on either side shows up as a gross error.
Agreement is asked to 1e-6 relative (measured: 1e-8 - two solvers of the damped 3x3 system, where each stops), validity flags and
motion answers exactly.  The first half runs the compiled reference live; the second half holds the oracle to the committed outputs
of the reference (tests/golden/ref_feature.npz, written by tests/golden/make_ref_feature.py), which needs nothing but the file; the
GPU suite holds the product's k_triangulate to the same file (tests/test_gpu_zz_golden.py)."""
import ctypes as C
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from oracle import lvo_be

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_feature.npz")


def selected_views(ids, mode, curr):
    """the views initializePosition* keeps: every observation whose state exists, minus the current frame in modes 0 and 2"""
    return [i for i in range(len(ids)) if not (mode != 1 and ids[i] == curr)]


def oracle_side(ids, q, pc, uv, mode, curr, is_init, pos_in):
    sel = selected_views(ids, mode, curr)
    poses = np.zeros(len(sel), lvo_be.POSE)
    for k, i in enumerate(sel):
        poses[k]["R"] = Rotation.from_quat(q[i]).as_matrix().ravel(); poses[k]["t"] = pc[i]
    # initializeInvParamPosition always starts from the two-view guess (:780-783); the other two from the stored position when there is one
    ok, pos, sol, idp, oa = lvo_be.triangulate(poses, uv[sel], use_position=bool(is_init) and mode != 2, position_in=pos_in)
    return ok, pos, sol, idp, oa, int(ids[sel[-1]])


def oracle_motion(ids, q, pc, uv, tracked, thr):
    last = len(ids) - 2 if tracked else len(ids) - 1
    P0 = np.zeros(1, lvo_be.POSE); P1 = np.zeros(1, lvo_be.POSE)
    P0[0]["R"] = Rotation.from_quat(q[0]).as_matrix().ravel(); P0[0]["t"] = pc[0]
    P1[0]["R"] = Rotation.from_quat(q[last]).as_matrix().ravel(); P1[0]["t"] = pc[last]
    z = np.ascontiguousarray(uv[0], np.float64)
    return bool(lvo_be._lib().lvo_check_motion(P0.ctypes.data, P1.ctypes.data, z.ctypes.data, C.c_double(thr)))


def compare(ok_ref, m_ref, ok_o, pos, sol, idp, oa, anchor):
    assert ok_ref == ok_o
    if not ok_o:
        return 0.0
    assert m_ref["id_anchor"] == anchor and m_ref["is_initialized"]
    return max(np.abs(pos - m_ref["position"]).max() / max(np.abs(pos).max(), 1.0), abs(idp - m_ref["inv_depth"]) / abs(idp),
               np.abs(oa - m_ref["obs_anchor"]).max(), np.abs(sol - m_ref["inv_param"]).max() / max(np.abs(sol).max(), 1.0))


def test_oracle_triangulation_and_motion_check_against_the_compiled_reference():
    from oracle import lvref
    if not lvref.feature_available():
        pytest.skip("oracle/_ref/liblvref_feature.so not built and /root/reference absent")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_ref_feature", os.path.join(os.path.dirname(GOLDEN), "make_ref_feature.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    worst = 0.0; n_ok = 0; n_motion = 0
    for c in gen.cases(7, 400):                                                       # other cases than the committed fixture's
        ok_r, m = lvref.feature_initialize(c["mode"], c["ids"], c["q"], c["pc"], c["ids"], c["uv"], c["curr"], c["is_init"], c["pos_in"])
        ok_o, pos, sol, idp, oa, anchor = oracle_side(c["ids"], c["q"], c["pc"], c["uv"], c["mode"], c["curr"], c["is_init"], c["pos_in"])
        worst = max(worst, compare(ok_r, m, ok_o, pos, sol, idp, oa, anchor)); n_ok += ok_o
        for t in (0, 1):
            mr = lvref.feature_check_motion(c["ids"], c["q"], c["pc"], c["ids"], c["uv"], t, c["thr"])
            assert mr == oracle_motion(c["ids"], c["q"], c["pc"], c["uv"], t, c["thr"]); n_motion += mr
    print("oracle against the compiled reference: 400 features, %d valid, worst relative difference %.1e; checkMotion true in %d of 800" % (n_ok, worst, n_motion))
    assert n_ok > 300 and worst < 1e-6


def test_oracle_against_the_references_committed_outputs():
    g = np.load(GOLDEN)
    worst = 0.0; n_ok = 0
    for k in range(len(g["n_views"])):
        n = int(g["n_views"][k]); ids = g["ids"][k, :n]; q = g["q_cam"][k, :n]; pc = g["p_cam"][k, :n]; uv = g["uv"][k, :n]
        o = g["out"][k]
        m = dict(position=o[0:3], inv_depth=o[6], obs_anchor=o[7:10], id_anchor=int(o[10]), inv_param=o[11:14], is_initialized=bool(o[14]))
        ok_o, pos, sol, idp, oa, anchor = oracle_side(ids, q, pc, uv, int(g["mode"][k]), int(g["curr_id"][k]), int(g["is_initialized"][k]), g["position_in"][k])
        worst = max(worst, compare(bool(g["ok"][k]), m, ok_o, pos, sol, idp, oa, anchor)); n_ok += ok_o
        for t in (0, 1):
            assert bool(g["motion"][k, t]) == oracle_motion(ids, q, pc, uv, t, float(g["threshold"][k]))
    print("oracle against tests/golden/ref_feature.npz: %d of %d valid, worst relative difference %.1e" % (n_ok, len(g["n_views"]), worst))
    assert n_ok == int(g["ok"].sum()) and worst < 1e-6
