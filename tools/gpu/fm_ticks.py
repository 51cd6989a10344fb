#!/usr/bin/env python3
"""Where k_fe_ransac_commit spends its time (mode 0 = old tracks), from the wall-clock ticks a -DLVK_FM_TIMING build records:
usage: LVK_LIB=variants/fmt.so fm_ticks.py <config> <frames>.  Front-end only (no filter): the track count stays at the budget."""
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
    L.lvk_debug_fm_ticks.argtypes = [C.c_void_p]; L.lvk_debug_fm_ticks.restype = None
    tk = np.zeros(32, np.uint64)
    rows = []
    lo = 0
    for i in range(n):
        hi = int(np.searchsorted(imu_all["t"], ts[i] + 0.0049, side="left"))
        fe.processImage(frames[i], imu_all[lo:hi], ts=float(ts[i]))
        lo = max(hi - 4, 0)
        L.lvk_debug_fm_ticks(tk.ctypes.data_as(C.c_void_p))
        t = tk.astype(np.int64)
        if i < 4 or t[16 + 2] != 0:
            continue
        d = lambda a, b: round(float(t[a] - t[b]) * 0.01, 2)       # 100 MHz counter -> us
        rows.append(dict(m=int(t[16]), iters=int(t[17]), compact=d(1, 0), undistort=d(2, 1), ransac=d(9, 2), commit=d(10, 9), total=d(10, 0),
                         last_round=dict(draw=d(5, 4), qr=d(11, 5), null=d(12, 11), finish=d(6, 12), score=d(7, 6), replay=d(8, 7))))
    fe.close(); ctx.close()
    keys = ("m", "iters", "compact", "undistort", "ransac", "commit", "total")
    out = {"config": cfg, "frames": len(rows), "mean": {k: round(float(np.mean([r[k] for r in rows])), 2) for k in keys},
           "last_round_mean": {k: round(float(np.mean([r["last_round"][k] for r in rows])), 2) for k in rows[0]["last_round"]},
           "iters_hist": {str(k): int(v) for k, v in zip(*np.unique([r["iters"] for r in rows], return_counts=True))}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
