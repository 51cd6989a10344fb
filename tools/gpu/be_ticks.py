#!/usr/bin/env python3
"""Phase times inside the filter's kernels, from the 100 MHz wall-clock ticks a -DLVK_BE_TIMING build records (lvk_internal.h: BE_TICK):
usage: LVK_LIB=variants/liblvk_ticks.so tools/gpu/be_ticks.py [config] [frames].  Runs the blocking driver over the benchmark's
workload and, after every update, reads the ticks of the LAST launch of k_feature_rows (workgroup 0), k_cov_propagate_augment
(copy row 100, strip 0, the IMU-block workgroup) and k_chol_fused (factor workgroup); prints the mean phase lengths in us."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "A"; n = int(sys.argv[2]) if len(sys.argv) > 2 else 160
    from larvio_amd import synthetic as S
    wl = S.workload(cfg)
    first = int(2.0 * wl["img_rate"])
    ts, frames = S.render_frames(first, n, cam=wl["cam"], seed=S.MASTER_SEED, img_rate=wl["img_rate"], procs=min(32, os.cpu_count() or 1))
    seq = S.imu_only_sequence(S.MASTER_SEED, cam=wl["cam"])
    imu_all = seq.imu_array(max(int(ts[0] * 200) - 4, 0), int(ts[-1] * 200) + 40)
    import larvio_amd
    from larvio_amd._lib import lib
    from larvio_amd.vio import VioDriver
    ctx = larvio_amd.Context(0)
    fe = larvio_amd.ImageProcessor(wl["fcfg"], ctx); assert fe.initialize()
    be = larvio_amd.LarVio(wl["bcfg"], ctx); assert be.initialize()
    drv = VioDriver(fe, be, imu_all)
    L = lib()
    for fn in ("lvk_debug_ticks_feature_rows", "lvk_debug_ticks_linalg"):
        getattr(L, fn).argtypes = [C.c_void_p]; getattr(L, fn).restype = None
    fr = np.zeros(64, np.uint64); la = np.zeros(64, np.uint64)
    acc = {}

    def add(name, v):
        acc.setdefault(name, []).append(v)
    R2q = lambda R: (lambda s: np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]))(np.sqrt(np.trace(R) + 1) * 2)
    for i in range(n):
        t = float(ts[i])
        if i == 1:
            k = int(np.searchsorted(imu_all["t"], t, side="right")) - 1
            t0 = imu_all["t"][k]; tr = seq.traj
            be.set_state(t0, R2q(tr.R_wb(t0)), tr.p_wb(t0), tr.vel(t0), np.zeros(3), np.zeros(3), imu_all["gyro"][k], imu_all["acc"][k])
        has, upd = drv.step(t, drv.visible_end(t), img=frames[i])
        if not upd or i < 70:
            continue
        L.lvk_debug_ticks_feature_rows(fr.ctypes.data_as(C.c_void_p)); L.lvk_debug_ticks_linalg(la.ctypes.data_as(C.c_void_p))
        f = fr.astype(np.int64); a = la.astype(np.int64)
        d = lambda x, hi, lo: float(x[hi] - x[lo]) * 0.01
        for nm, hi, lo in (("staged data arrived", 1, 0), ("zero G + column map", 2, 1), ("P_cc loads issued + Jacobians", 3, 2), ("P_cc parked", 4, 3), ("null-space projection", 5, 4),
                           ("T = G'P_cc, S", 6, 5), ("gate solve", 7, 6), ("staging write", 8, 7), ("result + direct output", 9, 8), ("TOTAL", 9, 0)):
            add("k_feature_rows: " + nm, d(f, hi, lo))
        for nm, hi, lo in (("copy row", 2, 0), ("strip: Phi + R -> LDS", 9, 8), ("strip: W = Phi R", 10, 9), ("strip: scatter", 11, 10),
                           ("imu block: loads", 17, 16), ("imu block: compute + scatter", 18, 17)):
            add("k_cov_propagate_augment: " + nm, d(a, hi, lo))
        m = int(a[42]); nblk = (m + 31) // 32
        add("k_chol_fused: rows", m)
        add("k_chol_fused: prologue (S -> LDS)", d(a, 23, 22))
        for p in range(nblk):
            add("k_chol_fused: panel %d factor+inverse (C1, wave 0)" % p, d(a, 24 + 3 * p, 23 if p == 0 else 26 + 3 * (p - 1) + 0))
            add("k_chol_fused: panel %d C1 barrier wait" % p, d(a, 25 + 3 * p, 24 + 3 * p))
            if p + 1 < nblk:
                add("k_chol_fused: panel %d L21 (C2)" % p, d(a, 26 + 3 * p, 25 + 3 * p))
        add("k_chol_fused: TOTAL factor workgroup", d(a, 40, 22))
    be.close(); fe.close(); ctx.close()
    print(json.dumps({"config": cfg, "updates": len(acc.get("k_feature_rows: TOTAL", [])), "mean_us": {k: round(float(np.mean(v)), 2) for k, v in acc.items()}}, indent=1))


if __name__ == "__main__":
    main()
