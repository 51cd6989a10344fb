#!/usr/bin/env python3
"""Where k_gftt_select spends its time, from the wall-clock ticks a -DLVK_GF_TIMING build records (first bucket + total):
usage: LVK_LIB=variants/liblvk_gft.so gf_ticks.py <config> <frames>.  Front-end only (no filter): the track count stays at the budget."""
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
    L.lvk_debug_gf_ticks.argtypes = [C.c_void_p]; L.lvk_debug_gf_ticks.restype = None
    tk = np.zeros(32, np.uint64); last = None
    rows = []
    lo = 0
    for i in range(n):
        hi = int(np.searchsorted(imu_all["t"], ts[i] + 0.0049, side="left"))
        has, _ = fe.processImage(frames[i], imu_all[lo:hi], ts=float(ts[i]))
        lo = max(hi - 4, 0)
        L.lvk_debug_gf_ticks(tk.ctypes.data_as(C.c_void_p))
        t = tk.astype(np.int64)
        if i < 6 or not has or (last is not None and t[7] == last):
            continue
        last = t[7]
        d = lambda a, b: round(float(t[a] - t[b]) * 0.01, 2)       # 100 MHz counter -> us
        rows.append(dict(histogram=d(1, 0), fold=d(2, 1), bucket_choice=d(3, 2), collect=d(4, 3), sort=d(5, 4), greedy=d(6, 5), later_buckets=d(7, 6), total=d(7, 0)))
    fe.close(); ctx.close()
    out = {"config": cfg, "detections": len(rows), "mean_us": {k: round(float(np.mean([r[k] for r in rows])), 2) for k in rows[0]}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
