#!/usr/bin/env python3
"""One case of tools/gpu/fuzz_whole_program.py through the HIP path and the oracle side by side (tests/test_gpu_vio_driver._driver_pair:
both started from the true state at the second frame, state and covariance compared after EVERY update, every discrete thing asserted
identical): where a whole-program difference comes from.  usage: tools/gpu/fuzz_case_parity.py <case> [wide] [frames]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import fuzz_whole_program as F
    k = int(sys.argv[1]); F.WIDE = "wide" in sys.argv[2:]; F.SIZES = "sizes" in sys.argv[2:]; F.PARAMS = "params" in sys.argv[2:]
    nums = [int(a) for a in sys.argv[2:] if a.isdigit()]
    cam, n, fo, bo, first = F.draw(k)
    bo = dict(bo, max_features=fo["max_features_num"])       # the filter's capacity follows the tracker's budget (as the adapter sets it from the YAML file)
    if nums:
        n = nums[0]
    import larvio_amd
    from tests import test_gpu_vio_driver as T
    ctx = larvio_amd.Context(0)
    try:
        r = T._driver_pair(ctx, cam, max(first, 40), n, fo, bo, init_from_gt=True, min_updates=1, oracle_threads=min(16, os.cpu_count() or 1))
        print("case %d%s: %d updates, worst relative difference %.3e (%s), counters %s" % (k, " wide" if F.WIDE else "", r[0], r[1], dict(T._WORST_AT), r[2]))
    except Exception:
        traceback.print_exc()
        print("case %d%s: stopped; worst so far %s" % (k, " wide" if F.WIDE else "", dict(T._WORST_AT)))


if __name__ == "__main__":
    main()
