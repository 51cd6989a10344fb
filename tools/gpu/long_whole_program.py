#!/usr/bin/env python3
"""A LONG whole-program run against the reference (not part of the suite): bench.py's workload as an ASL directory, through the
reference's main() twice - with the reference's own classes (oracle/_ref/larvio_ref_full, CPU) and on the product
(oracle/_ref/larvio_ref_main: the same main() over adapter/ + liblvk_hip.so) - every pose the viewer gets compared.
usage: tools/gpu/long_whole_program.py <config A|4> <frames> [tracker budget]"""
import os
import shutil
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    cfg = sys.argv[1]; n = int(sys.argv[2]); mf = int(sys.argv[3]) if len(sys.argv) > 3 else None
    from tests import test_gpu_zzz_ref_main as T
    d = tempfile.mkdtemp(prefix="lv", dir="/tmp")
    try:
        args, wl, _ = T.write_workload_sequence(d, cfg, n, max_features=mf)
        npos, dp, dR, n_map, path = T._product_against_whole_program(args, d, n // 3, 1e-6)
        print("long whole program %s: %d frames, %d poses, %.2f m flown, %d stable map points: the reference's main() on the product against the reference's whole program: position %.2e m, rotation %.2e"
              % (cfg, n, npos, path, n_map, dp, dR))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
