#!/usr/bin/env python3
"""CPU twin of tools/gpu/fuzz_whole_program.py (not part of the suite: a minute or two per case): the ORACLE's loop (lvo.Frontend +
lvo_be.Ekf, after a moving start with the reference's own initialiser compiled in place) against the reference's whole program
(oracle/_ref/larvio_ref_full: app/larvioMain.cpp + every src/*.cpp) on the fuzz's configurations - what holds the test infrastructure
itself to the reference on the axes the fuzz varies.  Needs /root/reference at build time (make -C oracle ref).
usage: tools/fuzz_oracle_whole_program.py <case> [<case> ...] [wide] [sizes] [params]"""
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples")); sys.path.insert(0, os.path.join(ROOT, "tools", "gpu"))


def main():
    import fuzz_whole_program as F
    from larvio_amd import synthetic as S
    from tests.conftest import synth_frames
    from tests.test_oracle_ref_main import oracle_loop_from_motion, oracle_loop, run_binary, FULL
    from make_euroc_dir import write_euroc_dir
    from oracle import lvo, lvref
    F.WIDE = "wide" in sys.argv; F.SIZES = "sizes" in sys.argv; F.PARAMS = "params" in sys.argv
    lvo.set_threads(min(8, os.cpu_count() or 1))
    for k in [int(a) for a in sys.argv[1:] if a.isdigit()]:
        cam, n, fo, bo, first = F.draw(k)
        fcfg = S.frontend_config(cam=cam, **fo); bcfg = S.backend_config(cam=cam, **bo)
        frames = synth_frames(first, n, cam=cam)
        seq = S.imu_only_sequence(cam=cam); ts = [f[0] for f in frames]
        imu_all = seq.imu_array(max(int(ts[0] * 200) - 4, 0), int(ts[-1] * 200) + 40)
        d = tempfile.mkdtemp(prefix="lv", dir="/tmp")
        try:
            os.makedirs(d + "/logs")
            write_euroc_dir(d, frames, imu_all, fcfg, bcfg, output_dir=d + "/logs/")
            args = [d + "/mav0/imu0/data.csv", d + "/mav0/cam0/data.csv", d + "/mav0/cam0/data", d + "/config.yaml"]
            M, out = run_binary(FULL, args, d, "pf.txt")
            dyn = "Dynamic initialization success" in out
            a = oracle_loop_from_motion(args, fcfg, bcfg, frames, lvref.dynamic_init)[0] if dyn else oracle_loop(args, fcfg, bcfg, frames)[:, :3]
            m = min(len(a), len(M)); dd = np.linalg.norm(M[:m, 12:15] - a[:m], axis=1)
            print("case %3d%s%s%s: %3d / %3d poses, %s start, %d of the reference's updates above 1 m / 0.5 m/s; oracle against the reference's whole program: %.2e m (first pose above 1e-6 m: %s)"
                  % (k, " wide" if F.WIDE else "", " sizes" if F.SIZES else "", " params" if F.PARAMS else "", len(a), len(M), "moving" if dyn else "static", out.count("Update change is too large"),
                     dd.max() if m else float("nan"), int(np.argmax(dd > 1e-6)) if (dd > 1e-6).any() else None), flush=True)
        finally:
            shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
