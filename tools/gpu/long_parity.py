#!/usr/bin/env python3
"""A LONG whole-loop parity run (not part of the suite: minutes of CPU oracle time): bench.py's workload through the HIP path and the
oracle side by side, state and covariance compared after EVERY update.  usage: tools/gpu/long_parity.py <config A|3|4|5> <frames>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    cfg = sys.argv[1]; n = int(sys.argv[2])
    import larvio_amd
    from larvio_amd import synthetic as S
    from tests import test_gpu_vio_driver as T
    wl = S.workload(cfg)
    ctx = larvio_amd.Context(0)
    n_upd, worst, c, n_tracks, n_clones, dim = T._driver_pair(ctx, None, int(2.0 * wl["img_rate"]), n, {}, {}, init_from_gt=True, min_updates=n // 3,
                                                              workload=wl, oracle_threads=min(16, os.cpu_count() or 1))
    print("long parity %s: %d frames, %d updates, worst relative difference %.3e (%s), counters %s, tracks %d, clones %d, dim %d"
          % (cfg, n, n_upd, worst, dict(T._WORST_AT), c, n_tracks, n_clones, dim))
    ctx.close()


if __name__ == "__main__":
    main()
