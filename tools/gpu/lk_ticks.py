#!/usr/bin/env python3
"""Where k_fe_lk_both spends its time, from the shader-clock account a -DLVK_LK_TIMING build keeps per block (fe_track_dev.h):
usage: LVK_LIB=variants/lkt.so lk_ticks.py <config> <frames>.  Front-end only (no filter): the track count stays at the budget.
Prints, over the old-track launches of the frames after the bootstrap: the mean block and the slowest block (the one the launch waits for)."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    cfg = sys.argv[1]; n = int(sys.argv[2])
    from larvio_amd import synthetic as S
    wl = S.workload(cfg)
    first = int(2.0 * wl["img_rate"])
    ts, frames = S.render_frames(first, n, cam=wl["cam"], seed=S.MASTER_SEED, img_rate=wl["img_rate"], procs=min(32, os.cpu_count() or 1))
    seq = S.imu_only_sequence(S.MASTER_SEED, cam=wl["cam"])
    imu_all = seq.imu_array(max(int(ts[0] * 200) - 4, 0), int(ts[-1] * 200) + 40)
    import larvio_amd
    from larvio_amd._lib import lib
    ctx = larvio_amd.Context(0)
    fe = larvio_amd.ImageProcessor(wl["fcfg"], ctx)
    assert fe.initialize()
    L = lib()
    L.lvk_debug_lk_ticks.argtypes = [C.c_void_p, C.c_void_p]; L.lvk_debug_lk_ticks.restype = None
    tk = np.zeros((4096, 12), np.uint64); sp = np.zeros((4096, 4), np.uint64)
    names = ["level set-up (template blends, wait for fetch)", "A sums", "eigen test + inverse", "iteration: origin, weights, load, blends", "iteration: b sums", "iteration: solve + stop tests"]
    mean_rows, slow_rows = [], []
    lo = 0
    for i in range(n):
        hi = int(np.searchsorted(imu_all["t"], ts[i] + 0.0049, side="left"))
        fe.processImage(frames[i], imu_all[lo:hi], ts=float(ts[i]))
        lo = max(hi - 4, 0)
        if i < 6:
            continue
        L.lvk_debug_lk_ticks(tk.ctypes.data_as(C.c_void_p), sp.ctypes.data_as(C.c_void_p))
        nt = len(fe.tracks()["ids"])
        t = tk[64:nt].astype(np.float64); s = sp[64:nt].astype(np.float64)      # blocks >= 64: the new points' launch (other stream, at most a few dozen points) writes the same table
        ok = s[:, 2] > 0
        if ok.sum() < 10:
            continue
        t = t[ok]; s = s[ok]
        cyc_per_us = (s[:, 2] / (s[:, 3] * 0.01)).mean()                 # shader cycles per microsecond (100 MHz wall ticks)
        k = int(np.argmax(s[:, 1]))
        row = lambda a, b: dict({names[q]: round(float(a[q] / cyc_per_us), 2) for q in range(6)}, iterations=float(a[6]), levels=float(a[7]), loads=float(a[8]),
                                fwd_us=round(float(b[0] / cyc_per_us), 2), fwd_rev_us=round(float(b[1] / cyc_per_us), 2), block_us=round(float(b[2] / cyc_per_us), 2))
        mean_rows.append(row(t.mean(0), s.mean(0))); slow_rows.append(row(t[k], s[k]))
        mean_rows[-1]["cycles_per_us"] = round(float(cyc_per_us), 1); mean_rows[-1]["tracks"] = int(ok.sum())
    fe.close(); ctx.close()
    avg = lambda rows: {k: round(float(np.mean([r[k] for r in rows])), 2) for k in rows[0]}
    print(json.dumps({"config": cfg, "launches": len(mean_rows), "mean_block": avg(mean_rows), "slowest_block": avg(slow_rows)}, indent=1))


if __name__ == "__main__":
    main()
