"""HIP path against the COMMITTED golden fixtures (tests/golden/, generated from the oracle by make_golden.py): the same bytes
the oracle is held to in tests/test_oracle_frontend.py::test_golden_fixtures and tests/test_oracle_backend.py::
test_backend_golden_fixture, without the oracle in the loop for the front-end stages."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_frontend_stages_against_the_committed_golden_fixture(gpu_ctx):
    from larvio_amd import ops
    z = np.load(os.path.join(GOLDEN, "frontend_small.npz"))
    img0, img1 = z["img0"], z["img1"]
    h, w = img0.shape
    assert np.array_equal(ops.clahe(gpu_ctx, img0), z["clahe0"])
    p0 = ops.Pyramid(gpu_ctx, w, h, 21, 2).build(img0, clahe=True)
    p1 = ops.Pyramid(gpu_ctx, w, h, 21, 2).build(img1, clahe=True)
    assert np.array_equal(p0.image(2), z["lvl2"]) and np.array_equal(p0.deriv(1), z["der1"])
    pts = p0.good_features(40, 0.01, 10.0)
    assert np.array_equal(pts, z["corners"])
    out, st, it = ops.lk_track(gpu_ctx, p0, p1, pts, pts)
    assert np.array_equal(st, z["lk_status"])
    assert np.array_equal(np.ascontiguousarray(out, np.float32).view(np.uint32), np.ascontiguousarray(z["lk_pts"], np.float32).view(np.uint32))
    p0.orb_prepare()
    d, a = ops.orb_describe(gpu_ctx, p0, pts)
    assert np.array_equal(d, z["desc"])
    mask, iters = ops.ransac_fundamental(gpu_ctx, z["x1"], z["x2"])
    assert np.array_equal(mask, z["ransac_mask"]) and iters == int(z["ransac_iters"])


def test_filter_against_the_committed_golden_fixture(gpu_ctx):
    """stored inputs of a short simulated run -> lvk_ekf_process; every update is compared with the live oracle (1e-5 relative,
    measured ~1e-9) by the pair runner, and the oracle's end state with the stored one"""
    from tests.test_oracle_backend import _load_backend_golden
    from tests.test_gpu_backend import _run_pair
    z, cfg, init, msgs = _load_backend_golden()

    class _Seq:
        traj = None
    n_upd, wx, wP, c, ora = _run_pair(gpu_ctx, msgs, z["imu"], _Seq, cfg, init_args=init)
    assert n_upd == len(z["trace"])
    s = ora.state()
    assert np.allclose(np.concatenate([[s["t"]], s["q"], s["p"], s["v"]]), z["trace"][-1][:-1], rtol=1e-9, atol=1e-12)
    assert [c[k] for k in ("hybrid", "msckf", "zupt", "gated_in", "gated_out", "map")] == list(z["counters"])


def test_orb_block_against_the_references_own_outputs(gpu_ctx):
    """HIP ORB path against tests/golden/ref_orb.npz - outputs of the REFERENCE's ORBDescriptor.cpp compiled in place
    (tests/golden/make_ref_orb.py; oracle/_ref), not of the oracle: mosaic + blur planes, IC angles, descriptors (cvRound ties and
    border-reaching patches included) and the Hamming distance, bit for bit, with neither the oracle nor the library in the loop."""
    from larvio_amd import ops
    z = np.load(os.path.join(GOLDEN, "ref_orb.npz"))
    img = z["img"]; h, w = img.shape
    p = ops.Pyramid(gpu_ctx, w, h, int(z["pad"]), 2).build(img, clahe=False)
    e, b = p.orb_prepare()
    assert int(e.astype(np.int64).sum()) == int(z["ext_sum"]) and int(b.astype(np.int64).sum()) == int(z["blur_sum"])
    assert np.array_equal(e[[0, 17, 31, 32, 150, h + 32, h + 63]], z["ext_rows"]) and np.array_equal(b[[32, 33, 150, h + 31]], z["blur_rows"])
    d, a = ops.orb_describe(gpu_ctx, p, z["pts"])
    assert np.array_equal(d, z["desc"])
    assert np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32), z["angle"].view(np.uint32))
    assert list(ops.hamming_rows(gpu_ctx, z["ham_a"], z["ham_b"])) == list(z["ham"])


def test_triangulation_kernel_against_the_references_own_outputs(gpu_ctx):
    """k_triangulate (lvk_triangulate) against tests/golden/ref_feature.npz - outputs of the REFERENCE's Feature::initializePosition family
    (include/larvio/feature.hpp:383-890) compiled in place (tests/golden/make_ref_feature.py; oracle/_ref), not of the oracle: validity flags
    exactly, world position / inverse depth / corrected anchor observation / (alpha, beta, rho) to 1e-6 relative (measured 1e-7: where two
    solvers of the damped 3x3 system stop), with neither the oracle nor the reference library in the loop."""
    from scipy.spatial.transform import Rotation
    from larvio_amd import larvio as lv
    from tests.test_oracle_ref_feature import selected_views
    g = np.load(os.path.join(GOLDEN, "ref_feature.npz"))
    worst = 0.0; n_ok = 0
    for k in range(len(g["n_views"])):
        n = int(g["n_views"][k]); ids = g["ids"][k, :n]; q = g["q_cam"][k, :n]; pc = g["p_cam"][k, :n]; uv = g["uv"][k, :n]
        mode = int(g["mode"][k]); sel = selected_views(ids, mode, int(g["curr_id"][k]))
        poses = np.zeros(len(sel), lv.POSE)
        for j, i in enumerate(sel):
            poses[j]["R"] = Rotation.from_quat(q[i]).as_matrix().ravel(); poses[j]["t"] = pc[i]
        ok, pos, sol, idp, oa = lv.triangulate(gpu_ctx, poses, uv[sel], use_position=bool(g["is_initialized"][k]) and mode != 2, position_in=g["position_in"][k])
        assert ok == bool(g["ok"][k]), k
        if not ok:
            continue
        o = g["out"][k]; n_ok += 1
        worst = max(worst, np.abs(pos - o[0:3]).max() / max(np.abs(pos).max(), 1.0), abs(idp - o[6]) / abs(idp), np.abs(oa - o[7:10]).max(),
                    np.abs(sol - o[11:14]).max() / max(np.abs(sol).max(), 1.0))
    print("k_triangulate against the reference's committed outputs: %d of %d valid, worst relative difference %.1e" % (n_ok, len(g["n_views"]), worst))
    assert n_ok == int(g["ok"].sum()) and worst < 1e-6


def test_filter_against_the_references_own_outputs(gpu_ctx):
    """the HIP filter against what the REFERENCE'S OWN LarVio::processFeatures made of the stored back-end inputs - /root/reference/src/
    larvio.cpp compiled in place (oracle/Makefile target `ref`), outputs written by tests/golden/make_ref_larvio.py into ref_larvio.npz
    (stream A: the inputs of backend_sim.npz - start from a state, td + extrinsics estimated, 8-clone window, 19 updates with hybrid,
    MSCKF and pruning steps); nothing of the reference is needed here.  After EVERY update: state dimension, the 30 state numbers,
    P's leading 15 x 15 block, trace(P), |P|_F and the number of in-state features against the reference's record (1e-6 relative; the
    oracle, which the pair runner also holds the HIP filter to, agrees with the reference to 1e-10 on this stream).  The file's second
    stream (a start at rest: static initialiser + zero-velocity updates) is held on the CPU side, oracle against reference
    (tests/test_oracle_ref_larvio.py)."""
    from tests.test_oracle_backend import _load_backend_golden
    from tests.test_gpu_backend import _run_pair
    from tests.test_oracle_ref_larvio import check_against_row
    z = np.load(os.path.join(GOLDEN, "ref_larvio.npz"))

    class _Seq:
        traj = None
    za, cfg, init, msgs = _load_backend_golden()
    k = [0]

    def on_update(sg, Pg, ids):
        row = z["a_state"][k[0]]
        check_against_row(sg, Pg, row, z["a_p15"][k[0]], z["a_pnorm"][k[0]], tol=1e-6, exact_time=False)
        assert len(ids) == int(row[31])
        k[0] += 1
    n_upd, wx, wP, c, ora = _run_pair(gpu_ctx, msgs, za["imu"], _Seq, cfg, init_args=init, on_update=on_update)
    assert n_upd == k[0] == len(z["a_state"])
    assert np.array_equal(ora.features()[0], z["a_feat_ids"])
    print("HIP filter vs the reference's record: updates", n_upd, "(HIP vs oracle: state", wx, "P", wP, ")")


def test_frontend_against_the_references_own_outputs(gpu_ctx):
    """the HIP front-end against what the REFERENCE'S OWN ImageProcessor::processImage made of 26 stored frames - /root/reference/src/
    image_processor.cpp compiled in place (oracle/Makefile target `ref`; behind OpenCV's image-algorithm names the oracle's restatements,
    around them the reference's own text), outputs written by tests/golden/make_ref_imgproc.py into ref_imgproc.npz; nothing of the
    reference is needed here.  Featureless frames at the start (the bootstrap waits) and in the middle (every track lost, ids continue).
    After EVERY frame, byte for byte: processImage's answer, image_state, track ids / lifetimes / points / descriptors, the new corners
    and the feature message (the frame runner holds the HIP front-end to the live oracle as well)."""
    from tests.test_gpu_frontend_edge import _run
    from tests.test_oracle_ref_imgproc import fixture_stream, check_frame_against_record
    z = np.load(os.path.join(GOLDEN, "ref_imgproc.npz"))
    frames, ts_all, imu_all, cfg = fixture_stream(z)
    seen = [0]

    def on_frame(i, have, msg_bytes, state, tracks, new_pts):
        check_frame_against_record(z, i, have, msg_bytes, state, tracks, new_pts)
        seen[0] += 1
    states, n_tracks, n_msgs = _run(gpu_ctx, frames, cfg, on_frame=on_frame)
    assert seen[0] == len(frames) and n_msgs == int(z["have"].sum())
