#!/usr/bin/env python3
"""Fuzz of the oracle's front-end object against the REFERENCE's own ImageProcessor compiled in place (oracle/_ref/liblvref_imgproc.so;
needs /root/reference at build time): random streams of tests/test_oracle_ref_imgproc.py::_random_frames, every frame byte for byte
(ids, lifetimes, points, init points, descriptors, new corners, the feature message).  usage:
  tools/fuzz_frontend.py <first> <count> [--small] [--procs N]
--small: 96..160 x 80..128 images with 3 levels (ORB layers narrower than the 32 px mosaic border).  Prints one line per failing case."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(arg):
    k, small = arg
    from oracle import lvref
    from tests.test_oracle_ref_imgproc import _random_frames, run_both
    frames, ts_all, imu_all, cfg = _random_frames(k, small)
    d = tempfile.mkdtemp(prefix="fz", dir="/tmp")
    try:
        states, n_tracks, n_msgs = run_both(frames, ts_all, imu_all, cfg, d, lvref)
        return k, None, len(frames), n_msgs, max(n_tracks) if n_tracks else 0
    except AssertionError as e:
        return k, str(e)[:200], len(frames), 0, 0
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    small = "--small" in sys.argv
    procs = int(sys.argv[sys.argv.index("--procs") + 1]) if "--procs" in sys.argv else (os.cpu_count() or 1)
    import multiprocessing as mp
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(one, [(k, small) for k in range(first, first + count)], chunksize=4)
    bad = [r for r in res if r[1] is not None]
    for r in bad:
        print("case %d%s: %s" % (r[0], " (small)" if small else "", r[1]))
    print("%d streams%s, %d frames, %d feature messages, %d with tracks; %d differ" % (len(res), " (small)" if small else "", sum(r[2] for r in res), sum(r[3] for r in res), sum(1 for r in res if r[4] > 0), len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
