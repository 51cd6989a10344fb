#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X: VIO frames/sec + p50 ms/frame on EuRoC-shaped synthetic input
(BASELINE.json: 752x480, ~150 tracks, 30-clone window).

A "step" is one camera frame through the hot path, timed the way the reference times it (app/larvioMain.cpp:106-116: processImage,
then processFeatures when a message was produced).  The image is handed over as a HOST buffer (the reference's cv::Mat): the copy
into the front-end's pinned staging and the H2D transfer are inside the timed region.  A second pass over the same frames with the
images already resident in HBM is reported beside it (`device_resident`).

The timed region always starts from the STATED steady state, whatever --steps/--warmup say: an untimed pre-roll runs until the
sliding window has filled and cycled (clones >= sw_size - 2 and >= 2 pruning MSCKF updates, larvio.cpp:2316-2320), then >= 20 more
updates with the H P GEMM bracketed by HIP events (MFMA utilisation); only then come the W warm-up and the K timed steps.  If the
steady state is not reached no value is printed.

  python bench.py --gpus N --steps K --warmup W [--config A|3|4|5] [--sequential] [--sharded]
  (N>1: launched by torch.distributed.run; --sharded needs --config 5)

N>1 without --sharded runs N independent estimators, one camera stream per GPU (replicas: at 150 tracks the R-factor all-gather
costs more than the whole update); with --sharded every rank sees the same stream, builds the measurement rows of its
contiguous slice of the features and the compressed R factors are all-gathered over RCCL before the replicated update.
value = frames of all ranks / max-over-ranks time.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense FP64 matrix peak (datasheet; profiles/README.md holds the measured saturation figure)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s achievable)


def pmc_lk_traffic(path):
    """average corrected FETCH_SIZE bytes per launch of the LK kernel family in a tools/pmc_summary.py table (kernel,launches,avg_kb,avg_bytes;
    kernel names carry commas - k_fe_lk_both<21, 1> - so the rows are split from the right); None when the table has no LK row"""
    with open(path) as f:
        rows = [l.strip().rsplit(",", 3) for l in f.readlines()[1:] if l.strip()]
    lk = [(int(r[1]), float(r[3])) for r in rows if len(r) == 4 and r[0].startswith("k_fe_lk_")]
    if not lk:
        return None
    return round(sum(n * b for n, b in lk) / sum(n for n, _ in lk), 1)


def R2q(R):
    t = np.trace(R); s_ = np.sqrt(t + 1) * 2
    return np.array([(R[2, 1] - R[1, 2]) / s_, (R[0, 2] - R[2, 0]) / s_, (R[1, 0] - R[0, 1]) / s_, 0.25 * s_])


def _cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def runtime_env(local_rank=0):
    """The library's own runtime settings (include/lvk_c.h: lvk_runtime_env) instead of a copy of them here: one hardware queue per
    stream, kernel arguments in device memory, and - LVK_BENCH_BIND=0 turns it off - the whole process (Python, the HIP runtime's own
    threads, our two driver threads) on the physical cores of ONE L3 group of the socket it runs on (rank r takes the r-th group).  Called
    after `import torch` (which loads, but does not initialise, the HIP runtime the library then shares) and before the first HIP call.
    Returns the flags that took effect."""
    from larvio_amd._lib import lib
    flags = 3 | (0 if os.environ.get("LVK_BENCH_BIND", "l3") == "0" else 4)
    return int(lib().lvk_runtime_env(flags, int(local_rank)))


class Run:
    """One estimator (front-end + filter + driver) fed frame by frame; the same object drives the pre-roll, the warm-up and the
    timed steps."""

    def __init__(self, wl, args, local_rank, imu_all, seq, ts, sequential, torch_stream=None, shard=None, deferred=None):
        """sequential: one blocking lvk_vio_process per frame.  deferred ("immediate" | "late"): the adapter's schedule under a blocking
        driver - lvk_vio_process_deferred per frame (processImage waits for its message, processFeatures queues the update and returns),
        the pose (lvk_ekf_get_state = getTbw + getVel) read right after processFeatures as app/larvioMain.cpp:139 does, or one frame late."""
        import larvio_amd
        from larvio_amd.vio import VioDriver, VioPipeline, VioDeferred
        self.deferred = deferred; self.owed = False
        if deferred:
            sequential = True                            # same per-frame call shape as the blocking step: no pipeline object
        self.wl, self.seq, self.ts, self.imu_all, self.sequential = wl, seq, ts, imu_all, sequential
        self.ctx = larvio_amd.Context(local_rank, stream=torch_stream)
        self.fe = larvio_amd.ImageProcessor(wl["fcfg"], self.ctx)
        assert self.fe.initialize()
        # the filter gets its own context (= its own HIP stream): its update overlaps the next frames' front-end
        self.ctx_be = self.ctx if (sequential and not deferred) else larvio_amd.Context(local_rank)
        self.be = larvio_amd.LarVio(wl["bcfg"], self.ctx_be)
        assert self.be.initialize()
        self.shard = None
        if shard is not None:                            # (rank, world, dist): RCCL communicator for THIS filter's context
            from larvio_amd import sharding
            self.shard = sharding.make_shard(self.ctx_be, *shard)
            self.be.set_shard(*self.shard.args())
        self.drv = VioDeferred(self.fe, self.be, imu_all) if deferred else VioDriver(self.fe, self.be, imu_all) if sequential else VioPipeline(self.fe, self.be, imu_all)
        self.his = [self.drv.visible_end(float(t)) for t in ts]
        self.inited = False
        self.i = 0                      # next frame

    def step(self, host_img=None, dev_ptr=None, stride=None):
        i = self.i; ts = self.ts
        if not self.inited and i >= 1:
            # the initializers are a cold path: the filter starts from ground truth at the second frame,
            # before the front-end's first feature message
            imu_all = self.imu_all
            k = int(np.searchsorted(imu_all["t"], ts[i], side="right")) - 1
            t_i = imu_all["t"][k]; tr = self.seq.traj
            if not self.sequential:
                self.drv.drain()
            self.be.set_state(t_i, R2q(tr.R_wb(t_i)), tr.p_wb(t_i), tr.vel(t_i), np.zeros(3), np.zeros(3), imu_all["gyro"][k], imu_all["acc"][k])
            self.inited = True
        if self.deferred:
            has, upd = self.drv.step(float(ts[i]), self.his[i], img=host_img, device_ptr=dev_ptr, stride=stride)
            if self.deferred == "immediate":
                if upd:
                    self.be.state()                      # getTbw / getVel right after processFeatures: waits for the update
            elif upd:
                self.owed = True
            elif self.owed:
                self.be.state(); self.owed = False       # the pose of the previous frame's update, read after this frame's processImage
        elif self.sequential:
            has, _ = self.drv.step(float(ts[i]), self.his[i], img=host_img, device_ptr=dev_ptr, stride=stride)
            self.ctx.sync()
        else:
            # returns when this frame's front-end is done (tracks + message); the update it triggers runs behind it
            has = self.drv.step(float(ts[i]), self.his[i], img=host_img, device_ptr=dev_ptr, stride=stride)
        self.i += 1
        return has

    def drain(self):
        if self.deferred:
            self.be.wait()
        if not self.sequential:
            self.drv.drain()
        self.ctx.sync(); self.ctx_be.sync()

    def close(self):
        self.drain()
        if not self.sequential:
            self.drv.close()
        self.be.close(); self.fe.close()
        if self.shard is not None:
            self.shard.close()
        if self.ctx_be is not self.ctx:
            self.ctx_be.close()
        self.ctx.close()


def preroll(run, frames, d_frames, n_max, sw_size, extra_updates, warmup=0, period=2):
    """Untimed: run until the window has filled and cycled, then `extra_updates` more updates with the H P GEMM bracketed, then up to
    period-1 more frames so that the timed region (which starts `warmup` frames later) begins with a publish frame: it then holds whole
    publish cycles [frame with a feature message, frames without] - K/period messages either way, but a window that ENDS on a message
    frame measures one more update's worth of pipeline fill.
    Returns (frames consumed, H P profile) or raises SystemExit when the steady state is not reached."""
    fsz = frames.shape[1] * frames.shape[2]; stride = frames.shape[2]
    last_pub = [None]

    def feed():
        i = run.i
        if d_frames is not None:
            has = run.step(dev_ptr=d_frames.data_ptr() + i * fsz, stride=stride)
        else:
            has = run.step(host_img=frames[i])
        if has:
            last_pub[0] = i
    steady_at = None
    while run.i < n_max:
        feed()
        if run.i % 4 == 0 and run.i >= 2 * (sw_size - 4):
            run.drain()
            c = run.be.counters()
            if len(run.be.clones()) >= sw_size - 2 and c["msckf"] >= 2:
                steady_at = run.i
                break
    if steady_at is None:
        run.drain()
        raise SystemExit("bench.py: steady state not reached in %d pre-roll frames (clones %d of sw_size %d, msckf updates %d): no value printed"
                         % (n_max, len(run.be.clones()), sw_size, run.be.counters()["msckf"]))
    run.be.profile(True)
    u0 = run.be.counters(); u0 = u0["hybrid"] + u0["msckf"]
    hp = None
    while run.i < n_max:
        feed()
        if run.i % 4 == 0:
            run.drain()
            c = run.be.counters()
            if c["hybrid"] + c["msckf"] - u0 >= 2 * extra_updates:       # hybrid + pruning update per message in the steady state
                break
    while last_pub[0] is not None and (run.i + warmup - last_pub[0]) % period != 0 and run.i < n_max + period:
        feed()
    run.drain()
    hp = run.be.profile(False)
    if hp["launches"] < extra_updates:
        raise SystemExit("bench.py: only %d bracketed H P launches in the pre-roll (need %d): no value printed" % (hp["launches"], extra_updates))
    return run.i, hp


LK_EVENT_STRIDE = int(os.environ.get("LVK_BENCH_LK_EVENT_STRIDE", "5"))


def timed(run, frames, d_frames, W, K, dist, torch):
    """W untimed warm-up steps, then exactly K timed steps bracketed by barrier + synchronize; returns a dict of measurements."""
    fsz = frames.shape[1] * frames.shape[2]; stride = frames.shape[2]

    def feed():
        i = run.i
        if d_frames is not None:
            return run.step(dev_ptr=d_frames.data_ptr() + i * fsz, stride=stride)
        return run.step(host_img=frames[i])
    # The driver loop is this interpreter: a generation-2 collection inside the timed region is a 10-40 ms stall that no C++ driver
    # has.  Collect BEFORE the warm-up steps (a full collection walks the whole heap: it leaves the caches cold and the GPU idle for
    # tens of milliseconds - the warm-up is there to undo exactly that) and keep the collector off until the timed region has ended.
    gc.collect()
    gc.disable()
    try:
        return _timed_body(run, feed, W, K, dist, torch)
    finally:
        gc.enable()


def _timed_body(run, feed, W, K, dist, torch):
    for _ in range(W):
        feed()
    run.drain()
    if not run.sequential:
        run.drv.stats(reset=True); run.drv.latencies(reset=True)
    pl0, it0 = run.fe.lk_stats()
    mg0 = run.fe.msg_stats()
    c0 = run.be.counters()
    run.fe.profile_enable((1 << 2) | (LK_EVENT_STRIDE << 16))   # HIP events around the LK launches (dominant kernel family) of every 5th frame: an event record is a barrier packet on the frame's dependent chain (odd stride: publish and non-publish frames alike)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    lat = np.empty(K); msg_mask = np.zeros(K, bool)
    t_begin = time.perf_counter()
    for k in range(K):
        t0 = time.perf_counter()
        msg_mask[k] = feed()
        lat[k] = time.perf_counter() - t0
    run.drain()                                          # every queued filter update has completed
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t_begin
    if dist is not None:
        elapsed = max_over_ranks(dist, torch, elapsed)
    prof = run.fe.profile_read()
    pl1, it1 = run.fe.lk_stats()
    mg1 = run.fe.msg_stats()
    c1 = run.be.counters()
    out = dict(msgs=mg1[0] - mg0[0], msg_features=mg1[1] - mg0[1], elapsed=elapsed, lat=lat, msg_mask=msg_mask, lk_pl=pl1 - pl0, lk_it=it1 - it0, lk_prof=prof["lk_fwd_rev"],
               n_updates=(c1["hybrid"] + c1["msckf"]) - (c0["hybrid"] + c0["msckf"]), n_hybrid=c1["hybrid"] - c0["hybrid"], n_msckf=c1["msckf"] - c0["msckf"],
               pst=None, e2e=None)
    if not run.sequential:
        out["pst"] = run.drv.stats()
        out["early"] = run.drv.early_counts()
        out["n_msgs_total"] = run.drv.drain()[1]         # messages since the pipeline was created (the early counts are since then too)
        e2e = lat.copy() * 1e6                           # image-in -> state-out: front-end completion, or the end of the update it triggered
        pl = run.drv.latencies()
        idx = np.flatnonzero(msg_mask)
        n = min(len(pl), len(idx))
        e2e[idx[:n]] = pl[:n]
        out["e2e"] = e2e * 1e-6
    else:
        out["e2e"] = lat.copy()
    run.fe.profile_enable(0)
    return out


def cpu_baseline(wl, frames, ts, imu_all, his, seq, n_pre, n_sample, all_cpus):
    """The CPU oracle (a restatement: kind "port") on this box's host cores: pre-roll with all cores (untimed, same results for any
    thread count), then n_sample steady-state frames on ONE thread (LARVIO is single-threaded) and the next n_sample frames with
    the loops over rows / tiles / tracks / key points spread over all cores."""
    from oracle import lvo, lvo_be
    if hasattr(os, "sched_setaffinity") and all_cpus:
        try:
            os.sched_setaffinity(0, all_cpus)
        except OSError:
            pass
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ora = lvo.Frontend(wl["fcfg"]); orb = lvo_be.Ekf(wl["bcfg"])
    st = dict(lo=0, inited=False)

    def one(i):
        buf = imu_all[st["lo"]:his[i]]
        if not st["inited"] and i >= 1:
            k = int(np.searchsorted(imu_all["t"], ts[i], side="right")) - 1
            t_i = imu_all["t"][k]; tr = seq.traj
            orb.set_state(t_i, R2q(tr.R_wb(t_i)), tr.p_wb(t_i), tr.vel(t_i), np.zeros(3), np.zeros(3), imu_all["gyro"][k], imu_all["acc"][k])
            st["inited"] = True
        ta = time.perf_counter()
        have, m = ora.process(frames[i], float(ts[i]), buf)
        tb = time.perf_counter()
        if have:
            ok, used = orb.process(float(ts[i]), m, buf); st["lo"] += used
        return tb - ta, time.perf_counter() - tb
    lvo.set_threads(min(ncores, 64))
    for i in range(n_pre):
        one(i)
    legs = {}
    i = n_pre

    def leg(nt, n, budget_s=None):
        # budget_s: a leg of the thread sweep ends early once it has used that much wall time (an oversubscribed thread count can be
        # a thousand times slower than the best one: r5_a measured 0.08 frames/s with 256 threads - 460 s for one sweep leg)
        nonlocal i
        lvo.set_threads(nt)
        c_fe = c_be = 0.0; done = 0
        t0 = time.perf_counter()
        for _ in range(n):
            a, b = one(i); c_fe += a; c_be += b; i += 1; done += 1
            if budget_s is not None and time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
        return dict(value=round(done / dt, 2), cores=nt, seconds=round(dt, 2), frames=done, front_end_ms_per_frame=round(c_fe / done * 1e3, 3),
                    back_end_ms_per_frame=round(c_be / done * 1e3, 3))
    legs["one_thread"] = leg(1, n_sample)
    # The multi-threaded leg: at these sizes (0.36 MB images, ~150 tracks) fork/join of hundreds of threads costs more than the loops
    # it spreads, so the thread count is swept on short samples and the best one is timed over the full sample; the all-core figure
    # is reported beside it.
    n_short = max(16, n_sample // 8)
    sweep = {}
    for nt in sorted(set(x for x in (4, 8, 16, 32, 64, ncores) if x <= ncores)):
        sweep[nt] = leg(nt, n_short, budget_s=2.0)
        if nt >= 16 and sweep[nt]["value"] < 0.25 * max(v["value"] for v in sweep.values()):
            break                                        # past the knee: more threads only oversubscribe (the all-core figure is then the last one measured)
    best_nt = max(sweep, key=lambda k: sweep[k]["value"])
    legs["best"] = leg(best_nt, n_sample)
    legs["all_cores"] = sweep[max(sweep)]
    lvo.set_threads(1)
    dim_cpu = orb.dim
    one_t = legs["one_thread"]
    return {"value": one_t["value"], "unit": "frames/s", "cores": 1, "kind": "port",
            "front_end_ms_per_frame": one_t["front_end_ms_per_frame"], "back_end_ms_per_frame": one_t["back_end_ms_per_frame"],
            "all_cores": {"value": legs["best"]["value"], "cores": legs["best"]["cores"], "front_end_ms_per_frame": legs["best"]["front_end_ms_per_frame"],
                          "back_end_ms_per_frame": legs["best"]["back_end_ms_per_frame"], "host_cores": ncores,
                          "every_core": {"value": legs["all_cores"]["value"], "cores": legs["all_cores"]["cores"]},
                          "sweep_frames_per_s": {str(k): v["value"] for k, v in sweep.items()},
                          "note": "OpenMP over image rows, CLAHE tiles, tracks, key points and the dense update's rows/columns (bit-identical results); "
                                  "thread count swept over 4..all host cores on short samples, the best one timed over the full sample"},
            "state_dim": dim_cpu, "host": cpu_model(),
            "sample": "%d steady-state frames of the same synthetic sequence per leg (window full after a %d-frame pre-roll; %.1f s + %.1f s of CPU work), "
                      "CPU oracle front-end + back-end: 1 thread (LARVIO is single-threaded), then OpenMP with the best thread count of a sweep up to all %d "
                      "host cores; a restatement, not the Eigen/OpenCV build"
                      % (n_sample, n_pre, one_t["seconds"], legs["best"]["seconds"], ncores)}


def backend_only(args, rank, world, local_rank, dist, torch, K=None, W=None, sharded=None):
    """The filter alone at configs[4] depth, fed by the feature-level simulator (larvio_amd.synthetic.simulate_features: 2000
    features per message, 60-clone window): one step = one feature message through processFeatures.  With --sharded every rank
    runs the same filter on the same messages, does the per-feature device work of its slice and the compressed blocks are
    all-gathered over RCCL (strong scaling of the update); otherwise rank r runs its own stream (replicas).
    At N = 1, --sharded runs the sharded branch through RCCL with a one-rank communicator (loop-back: pack -> ncclAllGather -> unpack
    -> second stage on one GPU); without it the line carries that run as `rccl_loopback` next to the unsharded value.
    The last 40 messages of the pre-roll run with HIP events around every level of the structure-aware TSQR compression (k_qr_sparse_reg)
    and around the H P GEMM: `roofline` (TSQR) and `roofline_mfma` of this workload."""
    import larvio_amd
    from larvio_amd import synthetic as S
    K = args.steps if K is None else K; W = args.warmup if W is None else W
    sharded = args.sharded if sharded is None else sharded
    max_features = (args.max_features if args.backend_only else None) or 2000
    sw_size = (args.sw_size if args.backend_only else None) or 60
    want_cpu = rank == 0 and world == 1 and not getattr(args, "no_cpu_baseline", False)
    n_pre = 2 * 60 + 12
    sim = S.simulate_features(21 + (0 if sharded else rank), t0=2.0, t1=2.0 + 0.1 * (n_pre + W + K + 2), max_feat=max_features,
                              n_per_batch=500, sw_size=sw_size, max_features_in_one_grid=2, estimate_td=1, estimate_extrin=1,
                              max_features=max_features)
    imu = sim["imu"]; msgs = sim["msgs"]

    def run_filter(use_shard, profile):
        ctx = larvio_amd.Context(local_rank)
        be = larvio_amd.LarVio(sim["cfg"], ctx); assert be.initialize()
        shard = None
        if use_shard:
            from larvio_amd import sharding
            shard = sharding.make_shard(ctx, rank, world, dist) if world > 1 else sharding.RcclShard(ctx, 0, 1, sharding.unique_id())
            be.set_shard(*shard.args())
        be.set_state(*sim["init"])
        st = dict(lo=0)

        def one(i):
            ts, m = msgs[i]
            hi = int(np.searchsorted(imu["t"], ts + 0.05, side="left"))
            upd, rest = be.processFeatures((ts, m), imu[st["lo"]:hi]); st["lo"] = hi - len(rest)
        i = 0; qr = hp = None
        while i < n_pre + W:
            if profile and i == n_pre - 40:
                be.profile(True); be.profile_qr()
            if profile and i == n_pre:
                qr = be.profile_qr(); hp = be.profile(False)
            one(i); i += 1
        if len(be.clones()) < sim["cfg"]["sw_size"] - 2:
            raise SystemExit("bench.py --backend-only: the window did not fill in the pre-roll: no value printed")
        c0 = be.counters(); s0 = be.shard_stats()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        lat = np.empty(K)
        gc.collect(); gc.disable()
        try:
            t_begin = time.perf_counter()
            for k in range(K):
                t0 = time.perf_counter(); one(i); i += 1; lat[k] = time.perf_counter() - t0
            ctx.sync(); torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            elapsed = time.perf_counter() - t_begin
        finally:
            gc.enable()
        if dist is not None:
            elapsed = max_over_ranks(dist, torch, elapsed)
        c1 = be.counters(); s1 = be.shard_stats()
        r = dict(elapsed=elapsed, lat=lat, dim=be.dim, clones=len(be.clones()), n_feat=len(msgs[i - 1][1]), qr=qr, hp=hp,
                 timed={k: c1[k] - c0[k] for k in ("hybrid", "msckf", "gated_in", "gated_out", "triangulations")}, shard={k: s1[k] - s0[k] for k in s1})
        be.close()
        if shard is not None:
            shard.close()
        ctx.close()
        return r

    m = run_filter(sharded, True)
    loop = None
    if world == 1 and not sharded and getattr(args, "loopback_probe", False):      # the probe of the default line: also execute the RCCL transport (one-rank communicator)
        try:
            lb = run_filter(True, False)
            loop = {"value": round(K / lb["elapsed"], 2), "unit": "messages/s", "ms_per_step": round(lb["elapsed"] / K * 1e3, 4), "shard": lb["shard"],
                    "p50_ms_per_message": round(float(np.median(lb["lat"])) * 1e3, 4), "p95_ms_per_message": round(float(np.percentile(lb["lat"], 95)) * 1e3, 4),
                    "slowest_ms": [round(float(x) * 1e3, 3) for x in np.sort(lb["lat"])[-6:]],
                    "note": "the same filter through the SHARDED branch with RCCL as the transport and a one-rank communicator: per-rank rows -> "
                            "first compression stage -> k_shard_pack -> ncclAllGather on the filter's stream -> k_shard_unpack -> replicated second stage"}
        except Exception as exc:                                   # the transport probe must not take the line down
            loop = {"error": str(exc)[:300]}
    cpu = None
    if want_cpu:
        cpu = cpu_baseline_backend(sim, n_pre, 12)
    streams = 1 if sharded else world
    if rank != 0:
        return None
    qr, hp = m["qr"], m["hp"]
    roof_qr = None
    if qr and qr["launches"] > 0 and qr["ms"] > 0:
        ach = qr["flops"] / qr["ms"] / 1e9
        roof_qr = {"kernel": "k_qr_sparse_reg (one level of the structure-aware TSQR compression of the stacked measurement rows; FP64 Householder nodes, columns in registers, reflector through LDS)",
                   "bound": "fp64-valu", "achieved": round(ach, 4), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / FP64_MFMA_PEAK_TFLOPS, 6),
                   "traffic": None, "flops_per_launch": round(qr["flops"] / qr["launches"], 1), "avg_launch_us": round(qr["ms"] / qr["launches"] * 1e3, 3),
                   "launches": qr["launches"], "rows_in_per_launch": round(qr["rows"] / qr["launches"], 1),
                   "flops_definition": "Householder flops on the structure actually factored: sum over nodes (r rows restricted to their c columns + residual) "
                                       "and reflectors j < min(r, c) of 4 (r - j)(c + 1 - j) ~ 2 r c^2 - 2/3 c^3 per node (SURVEY 8d)",
                   "peak_note": "MI355X FP64 vector peak = FP64 matrix peak = 78.6 TFLOP/s (datasheet)",
                   "measured_over": "the last 40 messages of the pre-roll (window full), HIP events on the filter's stream around every level launch"}
    roof_hp = None
    if hp and hp["launches"] > 0 and hp["ms"] > 0:
        ach = hp["flops"] / hp["ms"] / 1e9
        roof_hp = {"kernel": "k_dgemm_sk<false,false> (H P: four wavefronts split K for each 16x16 tile, ordered LDS reduction)", "bound": "mfma", "achieved": round(ach, 4), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": round(ach / FP64_MFMA_PEAK_TFLOPS, 6), "flops_per_launch": round(hp["flops"] / hp["launches"], 1),
                   "avg_launch_us": round(hp["ms"] / hp["launches"] * 1e3, 3), "launches": hp["launches"]}
    lat = m["lat"]
    out = {"metric": "EKF feature messages/sec (filter only, %d features per message, %d-clone window, state dim %d)" % (m["n_feat"], m["clones"], m["dim"]),
           "value": round(streams * K / m["elapsed"], 2), "unit": "messages/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(m["elapsed"] / K * 1e3, 4),
           "p50_ms_per_message": round(float(np.median(lat)) * 1e3, 4), "p95_ms_per_message": round(float(np.percentile(lat, 95)) * 1e3, 4),
           "slowest_ms": [round(float(x) * 1e3, 3) for x in np.sort(lat)[-6:]],
           "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "configs[4] depth, back-end only: simulated feature messages (no images), max_features %d, sw_size %d, 1d-hybrid"
                                  % (sim["cfg"]["max_features"], sim["cfg"]["sw_size"]),
                      "rehearsal_one_gpu_gloo": bool(os.environ.get("LVK_BENCH_ONE_GPU")),
                      "parallelism": ("sharded x%d: contiguous feature ranges per rank, one RCCL all-gather of the compressed blocks + gate results per update" % world)
                                     if sharded else "replicas x%d" % world,
                      "state_dim": m["dim"], "clones": m["clones"], "timed_region": m["timed"], "shard": m["shard"]},
           "roofline": roof_qr, "roofline_mfma": roof_hp, "cpu_baseline": cpu, "rccl_loopback": loop}
    if cpu:
        out["x_cpu_one_thread"] = round(out["value"] / cpu["value"], 2)
    return out


def cpu_baseline_backend(sim, n_pre, n_sample):
    """The CPU oracle's filter (kind "port") on the same simulated messages: pre-roll with OpenMP (same results for any thread count),
    then n_sample steady-state messages on ONE thread."""
    from oracle import lvo, lvo_be
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    be = lvo_be.Ekf(sim["cfg"]); be.set_state(*sim["init"])
    imu = sim["imu"]; msgs = sim["msgs"]; lo = 0
    lvo.set_threads(min(ncores, 16))
    t1 = 0.0
    for i in range(min(n_pre + n_sample, len(msgs))):
        if i == n_pre:
            lvo.set_threads(1); t1 = time.perf_counter()
        ts, m = msgs[i]
        hi = int(np.searchsorted(imu["t"], ts + 0.05, side="left"))
        upd, used = be.process(ts, m, imu[lo:hi]); lo += used
    dt = time.perf_counter() - t1
    n = min(n_pre + n_sample, len(msgs)) - n_pre
    return {"value": round(n / dt, 3), "unit": "messages/s", "cores": 1, "kind": "port", "state_dim": be.dim, "host": cpu_model(),
            "sample": "%d steady-state messages of the same simulated stream (%.1f s of CPU work on one thread, after a %d-message OpenMP pre-roll): "
                      "the CPU oracle's filter (dense Householder compression); a restatement, not the Eigen/SuiteSparse build" % (n, dt, n_pre)}


def adapter_cpp(wl, frames0, ts0, seq, n_timed, period):
    """frames/s of the C++ binary adapter/adapter_main - the loop of app/larvioMain.cpp:84-117 (processImage, processFeatures, every
    getter of :117-170 except the viewer's picture) written against the adapter classes with the reference's signatures - on the same
    synthetic camera written as an ASL directory from t = 0 (static initialiser, window fill), its LAST n_timed frames timed by the binary
    itself with the images decoded beforehand (--bench).  A separate process: it initialises its own HIP runtime, through the
    library's lvk_runtime_env defaults, and inherits this process's core binding."""
    import subprocess, tempfile, shutil, re
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    from make_euroc_dir import write_euroc_dir
    exe = os.path.join(ROOT, "adapter", "adapter_main")
    if not os.path.exists(exe):
        return {"error": "adapter/adapter_main not built"}
    d = tempfile.mkdtemp(prefix="lvk_asl_")
    try:
        imu = seq.imu_array(0, int(ts0[-1] * 200) + 40)
        write_euroc_dir(d, list(zip(ts0, frames0)), imu, wl["fcfg"], wl["bcfg"], output_dir=d + "/")
        cmd = [exe, d + "/mav0/imu0/data.csv", d + "/mav0/cam0/data.csv", d + "/mav0/cam0/data", d + "/config.yaml", "--bench", str(n_timed), "--no-vis"]
        best = None; runs = []
        for _ in range(2):
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            mt = re.search(r"bench frames (\d+) seconds ([0-9.]+) frames_per_s ([0-9.]+)", r.stdout)
            if r.returncode != 0 or not mt:
                return {"error": "adapter_main failed (rc %d): %s" % (r.returncode, (r.stdout + r.stderr)[-300:])}
            runs.append(float(mt.group(3)))
            tail = r.stdout.strip().splitlines()[-1]
        best = max(runs)
        return {"value": round(best, 2), "unit": "frames/s", "runs": runs, "timed_frames": n_timed, "frames_total": len(ts0), "binary": "adapter/adapter_main --bench %d --no-vis" % n_timed,
                "summary": tail,
                "note": "the reference's driver loop as a C++ process (adapter classes over the C ABI; deferred processFeatures, pose + covariances + window poses + "
                        "map points read right after it as app/larvioMain.cpp:117-170 does): static start at t = 0, window filled before the timed frames; "
                        "PNG decode outside the timed part; runtime settings from the library (lvk_runtime_env via lvk_context_create), core binding inherited"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def shard_probe(rank, world, local_rank):
    """Run `bench.py --backend-only [--sharded]` as a CHILD process per rank (own rendezvous on MASTER_PORT + 1, own RCCL
    communicator) with a 240 s time limit: a failure or a hang of the sharded path cannot take the headline measurement down with it."""
    import subprocess
    env = dict(os.environ)
    env["RANK"], env["WORLD_SIZE"], env["LOCAL_RANK"] = str(rank), str(world), str(local_rank)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
    env["LVK_BENCH_BIND"] = "0"                           # the parent already bound this process tree to one socket
    for k in [k for k in env if k.startswith("TORCHELASTIC_") or k in ("GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE")]:
        del env[k]                                       # the children rendezvous among themselves: with torchrun's TORCHELASTIC_USE_AGENT_STORE
                                                         # rank 0 would wait for an agent-hosted store on the new port instead of creating it
    cmd = [sys.executable, os.path.abspath(__file__), "--backend-only", "--gpus", str(world), "--steps", "40", "--warmup", "4"] + (["--sharded"] if world > 1 else ["--loopback-probe"])
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        return {"error": "the sharded-update probe did not finish within 240 s"} if rank == 0 else None
    if rank != 0:
        return None
    for line in reversed(r.stdout.strip().splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                break
    return {"error": "probe exited with code %d" % r.returncode, "stderr_tail": r.stderr[-400:]}


def dist_setup(torch, world, local_rank):
    """One process per GPU over RCCL (backend "nccl").  LVK_BENCH_BACKEND=gloo + LVK_BENCH_ONE_GPU=1 is a rehearsal of the multi-rank
    control flow on a single-GPU box (every rank on device 0, collectives through the host): not a measurement mode."""
    dev = 0 if os.environ.get("LVK_BENCH_ONE_GPU") else local_rank
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        if os.environ.get("LVK_BENCH_BACKEND", "nccl") == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    return dist, dev


def max_over_ranks(dist, torch, elapsed):
    t = torch.tensor([elapsed], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20, help="untimed steps right before the timed ones (the window is already full: see the pre-roll)")
    ap.add_argument("--config", default="A", choices=["A", "3", "4", "5"], help="BASELINE.json configs[1..4]; the metric is quoted on A")
    ap.add_argument("--sw-size", type=int, default=None)
    ap.add_argument("--max-features", type=int, default=None, help="tracker budget")
    ap.add_argument("--cpu-baseline-frames", type=int, default=None, help="steady-state frames per CPU leg (default 300 at A/3/4, 24 at 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-device-pass", action="store_true", help="skip the second (device-resident) pass")
    ap.add_argument("--dump-latencies", action="store_true", help="per-step caller time of the timed region (us) and which steps published a message, on stderr")
    ap.add_argument("--no-adapter-cpp", action="store_true", help="skip the C++ adapter binary's own frames/s (adapter/adapter_main on an ASL directory written from the synthetic camera)")
    ap.add_argument("--no-adapter-pass", action="store_true", help="skip the passes through the adapter's schedule (deferred processFeatures under a blocking driver)")
    ap.add_argument("--unaligned", action="store_true", help="camera stamps off the IMU grid (phase + jitter, jittered IMU stamps: larvio_amd.synthetic.unaligned_stamps): "
                                                               "the pipelined driver's early erase count is then not always available")
    ap.add_argument("--no-unaligned-pass", action="store_true", help="skip the extra pipelined pass on off-grid stamps appended to the default line")
    ap.add_argument("--sequential", action="store_true", help="one blocking lvk_vio_process per frame instead of the two-stream pipeline")
    ap.add_argument("--sharded", action="store_true", help="config 5 across ranks: per-rank feature rows, RCCL all-gather of the compressed R")
    ap.add_argument("--no-shard-probe", action="store_true", help="skip the configs[4]-depth (sharded) filter probe appended to the default line")
    ap.add_argument("--loopback-probe", action="store_true", help="with --backend-only at N=1: run the filter a second time through the sharded branch with a one-rank RCCL communicator")
    ap.add_argument("--backend-only", action="store_true", help="the filter alone at configs[4] depth on simulated feature messages (no images): "
                                                                  "the cheap way to time the (sharded) update at 2000 features")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.sharded and args.config != "5" and not args.backend_only:
        raise SystemExit("--sharded is the configs[4] path (2000 tracks): use --config 5 (or --backend-only)")
    if args.backend_only:
        import torch
        runtime_env(local_rank)                          # lvk_runtime_env: hardware queues, kernel arguments, L3-group binding - before the first HIP call
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: liblvk_hip.so has no CPU fallback")
        dist, local_rank = dist_setup(torch, world, local_rank)
        out = backend_only(args, rank, world, local_rank, dist, torch)
        if out is not None:
            print(json.dumps(out))
        if dist is not None:
            dist.destroy_process_group()
        return

    from larvio_amd import synthetic as S
    wl = S.workload(args.config, args.max_features, args.sw_size)
    K, W = args.steps, args.warmup
    sw = wl["sw_size"]
    n_pre_max = 2 * sw + 40 + 48                         # fill (one clone per message, every other frame) + cycle + 20 bracketed updates
    n_cpu = 0 if (args.no_cpu_baseline or world > 1) else (args.cpu_baseline_frames or (24 if args.config == "5" else 300))
    period = max(1, int(round(wl["img_rate"] / wl["fcfg"]["pub_frequency"])))      # frames per feature message
    n_frames = n_pre_max + period + max(W + K, 3 * n_cpu + 8 * 16) + 2
    first = int(2.0 * wl["img_rate"])                    # t = 2.0 s: the trajectory is moving
    all_cpus = set(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else set()
    seed_off = 0 if args.sharded else rank               # sharded: every rank sees the same camera
    procs = max(1, min(32, (len(all_cpus) or os.cpu_count() or 1) // max(world, 1)))
    ts, frames = S.render_frames(first, n_frames, cam=wl["cam"], seed=S.MASTER_SEED + seed_off, img_rate=wl["img_rate"], procs=procs)
    cpp_leg = rank == 0 and world == 1 and args.config == "A" and not args.sequential and not args.no_adapter_cpp and not args.unaligned
    n_cpp_timed = 200
    if cpp_leg:                                           # the C++ adapter binary starts at rest: frames from t = 0 (static start 1.2 s, window fill, 200 timed frames)
        ts0, frames0 = S.render_frames(0, 30 + n_pre_max + n_cpp_timed, cam=wl["cam"], seed=S.MASTER_SEED, img_rate=wl["img_rate"], procs=procs)
    seq = S.imu_only_sequence(S.MASTER_SEED + seed_off, cam=wl["cam"])
    k_lo = max(int(ts[0] * 200) - 4, 0)
    imu_all = seq.imu_array(k_lo, int(ts[-1] * 200) + 40)
    ts_grid, imu_grid = ts, imu_all
    if args.unaligned:
        ts, imu_all = S.unaligned_stamps(ts, imu_all, seed=S.MASTER_SEED + seed_off)

    import torch
    # three streams of ours + torch's: every stream on its own hardware queue (HIP's default is 4 queues, shared beyond that), kernel
    # arguments in device memory, one L3 group: the library's lvk_runtime_env, before the process's first HIP call
    rt_flags = runtime_env(local_rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: liblvk_hip.so has no CPU fallback")
    dist, local_rank = dist_setup(torch, world, local_rank)
    shard = (rank, world, dist) if (args.sharded and world > 1) else None

    stream = torch.cuda.current_stream()
    # Device spin-up (untimed, before anything of the estimator exists): ~1.5 s of plain GPU work, as any GPU benchmark warms its device
    # (sclk idles at 104 MHz; the estimator's own pre-roll is ~15 ms of sparse work).  Measured on the driver's flags as the FIRST process
    # of an idle box: filter 224 us per message without it (8,318 frames/s; 205 us / 9,011-9,075 for the second and third process),
    # 196 us with it (8,558) - but the 20-frame window's run-to-run spread (8,420-9,070 on one box, the caller's thread 80-87 us per
    # frame) is larger than that effect: it removes a known cold-start cost, it does not make the line repeatable (profiles/r6_s_*, r6_u_*).
    # LVK_BENCH_SPINUP_S=0 turns it off.
    spin_s = float(os.environ.get("LVK_BENCH_SPINUP_S", "1.5"))
    if spin_s > 0:
        a_ = torch.randn(2048, 2048, device="cuda", dtype=torch.float32)
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < spin_s:
            for _ in range(10):
                a_ = (a_ @ a_).clamp_(-1.0, 1.0)
            torch.cuda.synchronize()
        del a_
        torch.cuda.synchronize()
    # ---- pass 1 (headline): host images, H2D inside the timed region
    run = Run(wl, args, local_rank, imu_all, seq, ts, args.sequential, torch_stream=stream.cuda_stream, shard=shard)
    n_pre, hp = preroll(run, frames, None, n_pre_max, sw, 20, warmup=W, period=period)
    m = timed(run, frames, None, W, K, dist, torch)
    state_dim = run.be.dim; n_clones = len(run.be.clones()); counters = run.be.counters()
    if args.dump_latencies and rank == 0:
        sys.stderr.write("caller us per step: %s\nmessage steps: %s\nelapsed %.1f us, sum of steps %.1f us\n" % (
            " ".join("%.0f" % (x * 1e6) for x in m["lat"]), " ".join(str(int(x)) for x in m["msg_mask"]), m["elapsed"] * 1e6, m["lat"].sum() * 1e6))
    # tracks the tracker holds = mean size of the feature messages published inside the timed region (device counter)
    live = int(round(m["msg_features"] / m["msgs"])) if m["msgs"] else int(len(run.fe.tracks()["ids"]))
    run.close()
    if args.config == "A" and args.max_features is None and live < 145:
        raise SystemExit("bench.py: the tracker held only %d tracks per message in the timed region (the metric is quoted at ~150): no value printed" % live)
    # ---- pass 2: the same frames already resident in HBM (camera DMA case)
    md = None
    if not args.no_device_pass:
        d_frames = torch.from_numpy(frames[:n_pre_max + period + W + K + 2]).cuda()
        run2 = Run(wl, args, local_rank, imu_all, seq, ts, args.sequential, torch_stream=stream.cuda_stream, shard=shard)
        while run2.i < n_pre:                            # same pre-roll length as pass 1: the timed frames are the same frames
            i = run2.i
            run2.step(dev_ptr=d_frames.data_ptr() + i * frames.shape[1] * frames.shape[2], stride=frames.shape[2])
        md = timed(run2, frames, d_frames, W, K, dist, torch)
        run2.close()
        del d_frames

    # ---- the schedule the reference's own blocking drivers reach through the adapter classes (adapter/): processImage waits for its
    #      message, processFeatures is deferred (lvk_ekf_process_async), the pose is read (a) right after processFeatures, as
    #      app/larvioMain.cpp:139 does, (b) one frame late.  Same frames, same pre-roll, same K timed steps.
    adapter = None
    if not args.sequential and not args.no_adapter_pass and shard is None:
        adapter = {}
        for getters in ("immediate", "late"):
            run3 = Run(wl, args, local_rank, imu_all, seq, ts, True, torch_stream=stream.cuda_stream, deferred=getters)
            while run3.i < n_pre:
                run3.step(host_img=frames[run3.i])
            ma = timed(run3, frames, None, W, K, dist, torch)
            run3.close()
            adapter[getters] = ma
    # ---- the same pipelined schedule on a camera that is NOT on the IMU grid (every real one): the early erase count is only taken when
    #      no IMU sample lies within 0.5 ms of its bound, otherwise the frame waits for the running update (larvio.cpp:464-512)
    unal = None
    if not args.unaligned and not args.no_unaligned_pass and not args.sequential and shard is None and world == 1:
        ts_u, imu_u = S.unaligned_stamps(ts_grid, imu_grid, seed=S.MASTER_SEED + seed_off)
        run4 = Run(wl, args, local_rank, imu_u, seq, ts_u, False, torch_stream=stream.cuda_stream)
        while run4.i < n_pre:
            run4.step(host_img=frames[run4.i])
        mu = timed(run4, frames, None, W, K, dist, torch)
        run4.close()
        unal = {"value": round(K / mu["elapsed"], 2), "unit": "frames/s", "ms_per_step": round(mu["elapsed"] / K * 1e3, 4), "messages": int(mu["msg_mask"].sum()),
                "erase_counts_taken_early_since_start": mu["early"][0], "found_wrong": mu["early"][1], "messages_since_start": mu["n_msgs_total"],
                "back_end_ms_per_message": round(mu["pst"]["filter_us"] / max(int(mu["msg_mask"].sum()), 1) * 1e-3, 4),
                "caller_wait_ms_per_frame": round(mu["pst"]["caller_wait_us"] / K * 1e-3, 4),
                "note": "same frames and IMU values, stamps off the grid: image stamps + 1.7 ms phase +- 0.3 ms jitter, IMU stamps +- 50 us "
                        "(larvio_amd.synthetic.unaligned_stamps); frames whose erase count has an IMU sample within the margin wait for the running update"}
    # The sharded update at configs[4] depth, measured in the same invocation (filter only, simulated feature messages - seconds, no
    # rendering): at N > 1 the per-feature work is split over the N ranks with one RCCL all-gather per update, at N = 1 it is the
    # unsharded baseline of the same workload.  This is the strong-scaling curve north_star asks for "when the tracked-feature count
    # justifies it"; the headline value above stays the metric's own configuration.
    cpp = adapter_cpp(wl, frames0, ts0, seq, n_cpp_timed, period) if cpp_leg else None
    probe = None
    if not args.no_shard_probe and args.config == "A" and not args.sequential:
        probe = shard_probe(rank, world, local_rank)
    if rank == 0:
        win = wl["fcfg"]["patch_size"]
        # ---- roofline of the dominant kernel family (pyramidal LK): algorithmic bytes per SURVEY §8d
        lk_bytes = m["lk_pl"] * (win + 3) ** 2 + m["lk_it"] * (win + 1) ** 2
        lk_ms, lk_timed = m["lk_prof"]                   # event-bracketed launches: those of every LK_EVENT_STRIDE-th frame
        lk_var_env = os.environ.get("LVK_LK_VARIANT")
        lk_merged = win == 21 and os.environ.get("LVK_LK_MERGED") == "1" and (lk_var_env == "2" or (lk_var_env not in ("0", "1") and wl["max_features"] <= 600))
        lk_launches = (1 if lk_merged else 2) * K        # k_fe_lk_pipe carries both track sets (old tracks, new points) in ONE launch per frame; k_fe_lk_both: two (frontend.hip: lk_merged_ok)
        achieved = (lk_bytes / max(lk_launches, 1)) / (lk_ms / max(lk_timed, 1) * 1e-3) / 1e9 if lk_ms > 0 else 0.0
        # HBM traffic per launch: not measurable inside this process (PMC needs rocprofv3) - taken from the committed counter pass
        # of this same command (profiles/*_pmc_fetch_size.csv; FETCH_SIZE corrected as profiles/README.md's calibration states)
        traffic, traffic_src = None, None
        import glob
        tag = "c5" if args.config == "5" else "a"
        pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc_fetch_size.csv" % tag))) or \
             (sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_fetch_size.csv"))) if args.config == "A" else [])
        if pm:
            traffic = pmc_lk_traffic(pm[-1])
            if traffic is not None:
                traffic_src = os.path.basename(pm[-1])
        lk_var = os.environ.get("LVK_LK_VARIANT")
        lk_pipe = win == 21 and (lk_var == "2" or (lk_var not in ("0", "1") and wl["max_features"] <= 600))     # frontend.hip: launch_track_chain, LVK_LK_PIPE_MAX_TRACKS
        lk_name = ("k_fe_lk_pipe<%d> (forward + reverse LK of every track, five wavefronts per track: one iterates, three build the levels' templates of a pass, "
                   "one computes the ORB descriptors of the gate and the undistorted pair - the descriptor bytes are NOT counted in `achieved`)" % win) if lk_pipe else \
                  ("k_fe_lk_both<%d> (forward + reverse LK of every track; a second wavefront per track computes the ORB descriptors of the gate in their "
                   "shadow - its bytes are NOT counted in `achieved`)" % win)
        roofline = {"kernel": lk_name, "bound": "hbm", "achieved": round(achieved, 3),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                    "bytes_per_launch": round(lk_bytes / max(lk_launches, 1), 1), "avg_launch_us": round(lk_ms / max(lk_timed, 1) * 1e3, 3),
                    "launches": lk_launches, "launches_timed": lk_timed,
                    "timing": "HIP events on the launch's own stream around the LK launches of every %d-th frame of the timed region (an event record is a barrier "
                              "packet on the frame's dependent chain: bracketing every launch slowed the run it measured by ~10 %%)" % LK_EVENT_STRIDE}
        cpu = None
        if n_cpu:
            his = [int(np.searchsorted(imu_all["t"], float(t) + 0.05, side="left")) for t in ts]
            cpu = cpu_baseline(wl, frames, ts, imu_all, his, seq, n_pre, n_cpu, all_cpus)
        streams = 1 if args.sharded else world            # sharded: every rank works on the SAME camera stream
        value = streams * K / m["elapsed"]
        lat, e2e = m["lat"], m["e2e"]; mm = m["msg_mask"]; pst = m["pst"]

        def pct(a, q):
            return round(float(np.percentile(a, q)) * 1e3, 4) if len(a) else None
        metric = "VIO frames/sec (%dx%d, %d live tracks, %d-clone window)" % (wl["cam"]["width"], wl["cam"]["height"], live, n_clones)
        out = {"metric": metric, "value": round(value, 2), "unit": "frames/s",
               "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(m["elapsed"] / K * 1e3, 4),
               "timed_region_ms": round(m["elapsed"] * 1e3, 3),
               "timed_region_note": "K frames fed + every queued filter update drained, between two barrier + synchronize pairs; with few steps the region is mostly pipeline "
                                    "fill and drain (one update is ~0.2 ms: at --steps 20 the region is ~2 ms and the value ~10 %% below the 500-step one)",
               "p50_ms_per_frame": pct(e2e, 50), "p95_ms_per_frame": pct(e2e, 95),
               "p50_ms_frame_without_message": pct(e2e[~mm], 50), "p50_ms_frame_with_message": pct(e2e[mm], 50), "p95_ms_frame_with_message": pct(e2e[mm], 95),
               "latency_definition": "image-in -> state-out per frame: front-end completion for frames without a feature message, the end of the "
                                     "filter update the frame triggered otherwise (host wall time, lvk_vio_pipe_latency)",
               "p50_ms_front_end_completion": pct(lat, 50),
               "front_end_ms_per_frame": None if pst is None else round(pst["front_end_us"] / K * 1e-3, 4),
               "back_end_ms_per_message": None if pst is None else round(pst["filter_us"] / max(int(mm.sum()), 1) * 1e-3, 4),
               "caller_wait_ms_per_frame": None if pst is None else round(pst["caller_wait_us"] / K * 1e-3, 4),
               "worker_idle_ms_per_message": None if pst is None else round(pst["worker_idle_us"] / max(int(mm.sum()), 1) * 1e-3, 4),
               "erase_counts_taken_early": None if pst is None else {"since_start": m["early"][0], "found_wrong": m["early"][1], "messages_since_start": m["n_msgs_total"],
                                                                     "note": "lvk_vio_pipe_early_counts: the caller's thread took the IMU erase count of a queued update from the last published td "
                                                                             "(no IMU sample within 0.5 ms of the bound) instead of waiting for the running update; checked by the filter's thread"},
               "device_resident": None if md is None else {"value": round(streams * K / md["elapsed"], 2), "unit": "frames/s", "ms_per_step": round(md["elapsed"] / K * 1e3, 4),
                                                           "p50_ms_per_frame": pct(md["e2e"], 50),
                                                           "note": "same frames, already in HBM when the timed region starts (no staging copy, no H2D)"},
               "adapter_path": None if not adapter else {
                   "value": round(streams * K / adapter["immediate"]["elapsed"], 2), "unit": "frames/s", "ms_per_step": round(adapter["immediate"]["elapsed"] / K * 1e3, 4),
                   "p50_ms_per_frame": pct(adapter["immediate"]["lat"], 50),
                   "pose_read_one_frame_late": {"value": round(streams * K / adapter["late"]["elapsed"], 2), "ms_per_step": round(adapter["late"]["elapsed"] / K * 1e3, 4),
                                                "p50_ms_per_frame": pct(adapter["late"]["lat"], 50)},
                   "note": "what an UNCHANGED blocking driver (app/larvioMain.cpp:104-116 + getters at :139) gets through the adapter classes: "
                           "lvk_frontend_process (waits for its message) + lvk_ekf_process_async per frame, pose read right after processFeatures "
                           "(value) or after the next frame's processImage (pose_read_one_frame_late); identical results "
                           "(tests/test_gpu_vio_driver.py::test_deferred_update_is_identical_to_blocking)"},
               "adapter_cpp": cpp,
               "unaligned_stamps": unal,
               "runtime_env": {"applied_flags": rt_flags, "source": "lvk_runtime_env (liblvk_hip.so): 1 GPU_MAX_HW_QUEUES=8, 2 HIP_FORCE_DEV_KERNARG=1, 4 one L3 group"},
               "higher_is_better": True, "scaling": ("strong" if args.sharded else "weak"), "vs_baseline": None, "dtype": "u8/f32 front-end, f64 back-end",
               "data": "synthetic",
               "config": {"workload": wl["label"] + ", pyramid 3 levels, win %d, pub %g Hz" % (win, wl["fcfg"]["pub_frequency"]),
                          "input": "host images (pageable numpy, as a cv::Mat): staging copy + H2D inside the timed region",
                          "schedule": ("sequential: one blocking lvk_vio_process per frame" if args.sequential else
                                       "pipelined: filter update of frame k (own stream + worker thread) overlaps the front-end of frames k+1..; "
                                       "identical results (tests/test_gpu_vio_driver.py); all updates drained inside the timed region"),
                          "stages": "processImage every frame + processFeatures on every feature message, as app/larvioMain.cpp:106-116",
                          "pre_roll_frames": n_pre, "timed_region_alignment": "starts on a publish frame: %d whole publish cycles of %d frames" % (K // period, period) if bool(m["msg_mask"][0]) else "starts %d frame(s) before a publish frame" % int(np.argmax(m["msg_mask"])), "sw_size": sw, "clones": n_clones, "state_dim": state_dim, "backend": counters,
                          "timed_region": {"messages": int(mm.sum()), "hybrid_updates": m["n_hybrid"], "msckf_pruning_updates": m["n_msckf"]},
                          "live_tracks": live, "tracker_budget_max_features_num": wl["fcfg"]["max_features_num"], "nominal_tracks": wl["nominal_tracks"],
                          "stamps": "off the IMU grid (--unaligned)" if args.unaligned else "on the IMU grid (synthetic)",
                          "rehearsal_one_gpu_gloo": bool(os.environ.get("LVK_BENCH_ONE_GPU")), "parallelism": ("sharded x%d: contiguous feature ranges per rank, RCCL all-gather of the packed R factors" % world) if args.sharded
                                         else "replicas x%d" % world},
               "roofline": roofline,
               # the one GEMM-shaped contraction of the path (P H^T as H P, FP64 MFMA 16x16x4): utilisation against the dense FP64 matrix peak
               "roofline_mfma": {"kernel": "k_dgemm_sk<false,false> (H P: four wavefronts split K for each 16x16 tile, ordered LDS reduction)", "bound": "mfma", "achieved": round(hp["flops"] / max(hp["ms"], 1e-9) / 1e9, 4),
                                 "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(hp["flops"] / max(hp["ms"], 1e-9) / 1e9 / FP64_MFMA_PEAK_TFLOPS, 6),
                                 "flops_per_launch": round(hp["flops"] / max(hp["launches"], 1), 1),
                                 "avg_launch_us": round(hp["ms"] / max(hp["launches"], 1) * 1e3, 3), "launches": hp["launches"],
                                 "measured_over": "the last >= 20 updates of the pre-roll (window full and cycling; kept out of the timed region: "
                                                  "its event records would add ~10% to the filter chain)"},
               "cpu_baseline": cpu, "sharded_update_probe": probe,
               # the TSQR compression north_star names, measured where it runs (configs[4] depth; at configs[1] the stack is < 480 rows and is not compressed)
               "roofline_qr": (probe or {}).get("roofline") if isinstance(probe, dict) else None}
        if cpu:
            out["x_cpu_one_thread"] = round(value / cpu["value"], 2)
            out["x_cpu_all_cores"] = round(value / cpu["all_cores"]["value"], 2)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
