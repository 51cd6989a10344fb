#!/usr/bin/env python3
"""A measured number behind the declined "LK patch windows shard across GPUs" half of north_star: what a 2-way split of
the tracker at configs[4] could save (k_fe_lk_both at 2000 vs 1000 tracks, event-bracketed launch duration from bench.py's roofline
block) against what it would add per frame - an all-gather of 48 bytes per track (point, status, descriptor) - timed here on the
RCCL transport with a one-rank communicator (a LOWER bound: no xGMI hop).  usage: lk_split_probe.py bench_c5_2000.json bench_c5_1000.json"""
import ctypes as C
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import larvio_amd
from larvio_amd import sharding
from larvio_amd._lib import lib


def last_json(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return None


def main():
    out = {}
    for tag, path in zip(("tracks_2000", "tracks_1000"), sys.argv[1:3]):
        d = last_json(path)
        if d:
            out[tag] = {"lk_avg_launch_us": d["roofline"]["avg_launch_us"], "live_tracks": d["config"]["live_tracks"], "frames_per_s": d["value"]}
    ctx = larvio_amd.Context(0)
    sh = sharding.RcclShard(ctx, 0, 1, sharding.unique_id())
    L = lib()
    L.lvk_shard_allgather_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]; L.lvk_shard_allgather_rccl.restype = C.c_int
    res = {}
    for n_tracks in (1000, 2000):
        nbytes = 48 * n_tracks
        d_s = ctx.to_device(np.zeros(nbytes, np.uint8)); d_r = ctx.to_device(np.zeros(nbytes, np.uint8))
        st = C.c_void_p(ctx.stream)
        for _ in range(20):
            L.lvk_shard_allgather_rccl(sh._h, C.c_void_p(d_s.ptr), C.c_void_p(d_r.ptr), nbytes, st)
        ctx.sync()
        single = []
        for _ in range(50):                               # one collective, waited for: what a frame would see on its dependent chain
            t0 = time.perf_counter(); L.lvk_shard_allgather_rccl(sh._h, C.c_void_p(d_s.ptr), C.c_void_p(d_r.ptr), nbytes, st); ctx.sync(); single.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        for _ in range(200):                              # back to back: device-side cost per collective
            L.lvk_shard_allgather_rccl(sh._h, C.c_void_p(d_s.ptr), C.c_void_p(d_r.ptr), nbytes, st)
        ctx.sync()
        res["%d_tracks_%d_bytes" % (n_tracks, nbytes)] = {"launch_to_done_us_p50": round(float(np.median(single)) * 1e6, 2), "back_to_back_us": round((time.perf_counter() - t0) / 200 * 1e6, 2)}
    sh.close(); ctx.close()
    out["rccl_allgather_world1"] = res
    if "tracks_2000" in out and "tracks_1000" in out:
        save = out["tracks_2000"]["lk_avg_launch_us"] - out["tracks_1000"]["lk_avg_launch_us"]
        cost = res["1000_tracks_48000_bytes"]["launch_to_done_us_p50"]
        out["verdict"] = {"lk_saving_us_per_launch": round(save, 2), "exchange_cost_us_lower_bound": cost,
                          "note": "two LK launches per frame (old tracks, new points) would each save lk_saving_us; each needs its results gathered before the "
                                  "RANSAC commit: one exchange per launch on the frame's dependent chain (one-rank loop-back, no xGMI hop: a lower bound)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
