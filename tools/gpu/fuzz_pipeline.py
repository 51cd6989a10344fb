#!/usr/bin/env python3
"""Fuzz of the PIPELINED driver (lvk_vio_pipe_*: what bench.py times - front-end of frame k+1 on one stream while the filter's update of
frame k runs on another, erase counts taken early) against the sequential driver step, on the random configurations of
tools/gpu/fuzz_whole_program.py: same library, same kernels, only the schedule differs - so everything must be equal BIT FOR BIT (state,
covariance, clone ids, in-state feature ids, counters, track set), from the filter's own start (static or moving).
usage: tools/gpu/fuzz_pipeline.py <first> <count> [wide] [sizes] [params]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def one(k, ctx, ctx2):
    import fuzz_whole_program as F
    import larvio_amd
    from larvio_amd import synthetic as S
    from larvio_amd.vio import VioDriver, VioPipeline
    from tests.conftest import synth_frames
    cam, n, fo, bo, first = F.draw(k)
    bo = dict(bo, max_features=fo["max_features_num"])       # the filter's capacity follows the tracker's budget (as the adapter sets it from the YAML file)
    n = min(n, 200)
    frames = synth_frames(first, n, cam=cam)
    seq = S.imu_only_sequence(cam=cam); ts = [f[0] for f in frames]
    imu_all = seq.imu_array(max(int(ts[0] * 200) - 4, 0), int(ts[-1] * 200) + 40)
    fcfg = S.frontend_config(cam=cam, **fo); bcfg = S.backend_config(cam=cam, **bo)
    out = []; early = wrong = 0
    for mode in ("seq", "pipe"):
        fe = larvio_amd.ImageProcessor(fcfg, ctx); assert fe.initialize()
        be = larvio_amd.LarVio(bcfg, ctx2 if mode == "pipe" else ctx); assert be.initialize()
        drv = (VioPipeline if mode == "pipe" else VioDriver)(fe, be, imu_all)
        n_msg = 0; err = None
        try:
            for t, img in frames:
                r = drv.step(t, drv.visible_end(t), img=img)
                n_msg += int(r if mode == "pipe" else r[0])
            if mode == "pipe":
                n_upd, n_m = drv.drain(); assert n_m == n_msg
                early, wrong = drv.early_counts()
        except Exception as e:          # a filter that reports an error does so in both schedules, at the same message
            err = str(e)[:120]
        if mode == "pipe":
            try:
                drv.close()
            except Exception:
                pass
        if err is None:
            st = be.state()
            out.append((n_msg, be.dim, {kk: np.array(v, copy=True) for kk, v in st.items()}, be.cov(), be.clones()["id"].copy(), be.features()[0].copy(), be.counters(), fe.tracks(), None))
        else:
            out.append((n_msg, 0, {}, None, None, None, None, None, err))
        be.close(); fe.close()
    a, b = out
    tag = "case %3d %s %dx%d budget %3d sw %2d pub %2d %s start" % (k, "fisheye" if cam["distortion_model"] == 1 else "radtan ", cam["width"], cam["height"], fo["max_features_num"], bo["sw_size"], fo["pub_frequency"], "moving" if first else "static")
    if a[8] or b[8]:
        return tag + "  errors: sequential %r pipelined %r%s" % (a[8], b[8], "" if a[8] == b[8] else "  <-- DIFFERS"), a[8] == b[8]
    same = a[0] == b[0] and a[1] == b[1] and all(np.array_equal(a[2][kk], b[2][kk]) for kk in a[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) \
        and np.array_equal(a[5], b[5]) and a[6] == b[6] and all(np.array_equal(a[7][kk], b[7][kk]) for kk in ("ids", "pts", "lifetime"))
    return tag + "  %3d messages, dim %3d, %d updates, %d erase counts taken early (%d wrong): %s" % (a[0], a[1], a[6]["hybrid"] + a[6]["msckf"], early, wrong, "identical" if same and not wrong else "DIFFERS"), same and not wrong


def main():
    import fuzz_whole_program as F
    import larvio_amd
    first, count = int(sys.argv[1]), int(sys.argv[2]); F.WIDE = "wide" in sys.argv[3:]; F.SIZES = "sizes" in sys.argv[3:]; F.PARAMS = "params" in sys.argv[3:]
    ctx = larvio_amd.Context(0); ctx2 = larvio_amd.Context(0)
    bad = 0
    for k in range(first, first + count):
        line, ok = one(k, ctx, ctx2)
        print(line, flush=True); bad += 0 if ok else 1
    print("%d configurations, pipelined against sequential: %d differ" % (count, bad))


if __name__ == "__main__":
    main()
