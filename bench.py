#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X: VIO frames/sec + p50 ms/frame on EuRoC-shaped
synthetic input (752x480 @20 Hz, ~150 tracks, 30-clone window; BASELINE.json / SURVEY.md §8d).

A "step" is one camera frame through the hot path, timed the way the reference times it
(app/larvioMain.cpp:106-116: processImage, then processFeatures when a message was produced), with the
frames already resident in HBM.  Each step ends with a stream synchronise (a VIO consumes frame k's
result before frame k+1 exists), so value = K / sum(frame latencies).

  python bench.py --gpus N --steps K --warmup W         (N>1: launched by torch.distributed.run)

N>1 at this configuration runs N independent estimators, one camera stream per GPU ("replicas only",
DESIGN.md §multi-GPU: at 150 tracks the R-factor all-gather costs more than the whole update);
value = frames of all ranks / max-over-ranks time, scaling "weak".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense FP64 matrix peak (datasheet)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s achievable)


def render_frames(first, count, seed_offset=0):
    from larvio_amd import synthetic as S
    path = "/tmp/lvk_bench_frames_%d_%d_%d.npz" % (first, count, seed_offset)
    if os.path.exists(path):
        z = np.load(path)
        return z["ts"], z["img"]
    seq = S.Sequence(seed=S.MASTER_SEED + seed_offset)
    ts = np.empty(count); img = np.empty((count, S.EUROC["height"], S.EUROC["width"]), np.uint8)
    for i in range(count):
        ts[i], img[i] = seq.frame(first + i)
    try:
        np.savez(path, ts=ts, img=img)
    except OSError:
        pass
    return ts, img


def imu_stream(seed_offset=0):
    from larvio_amd import synthetic as S
    seq = S.imu_only_sequence(S.MASTER_SEED + seed_offset)
    return seq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=80, help="untimed frames; >= 70 so the 30-clone window is full when timing starts")
    ap.add_argument("--sw-size", type=int, default=30)
    ap.add_argument("--max-features", type=int, default=150, help="tracker budget (holds ~150 live tracks)")
    ap.add_argument("--cpu-baseline-frames", type=int, default=2000, help="frames of the same sequence the 1-thread CPU oracle is timed on (capped at steps+warmup)")
    ap.add_argument("--sequential", action="store_true", help="one blocking lvk_vio_process per frame instead of the two-stream pipeline")
    args = ap.parse_args()

    # Dual-socket hosts: keep the whole process - Python, the HIP runtime's own threads and queues, our two driver threads - on ONE
    # socket, before anything initialises HIP.  Which socket made no difference in A/B runs; a process whose threads straddle both
    # did (0.30 vs 0.35 ms per filter update).  LVK_BENCH_BIND=0 disables.
    if os.environ.get("LVK_BENCH_BIND", "1") != "0" and hasattr(os, "sched_setaffinity"):
        try:
            cpu = os.sched_getcpu() if hasattr(os, "sched_getcpu") else min(os.sched_getaffinity(0))
            import glob as _glob
            for node in _glob.glob("/sys/devices/system/node/node*/cpulist"):
                cpus = set()
                for part in open(node).read().strip().split(","):
                    a, _, b = part.partition("-")
                    cpus.update(range(int(a), int(b or a) + 1))
                if cpu in cpus:
                    keep = cpus & os.sched_getaffinity(0)
                    if keep:
                        os.sched_setaffinity(0, keep)
                    break
        except (OSError, ValueError):
            pass
    # three streams of ours + torch's: keep every stream on its own hardware queue (HIP's default is 4 queues, shared beyond that)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: liblvk_hip.so has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import larvio_amd
    from larvio_amd import synthetic as S
    K, W = args.steps, args.warmup
    first = 40                                           # t = 2.0 s: the trajectory is moving
    ts, frames = render_frames(first, K + W, seed_offset=rank)
    seq = imu_stream(seed_offset=rank)
    d_frames = torch.from_numpy(frames).cuda()           # inputs resident in HBM before the timed region
    stream = torch.cuda.current_stream()
    ctx = larvio_amd.Context(local_rank, stream=stream.cuda_stream)
    cfg = S.frontend_config(max_features_num=args.max_features)
    fe = larvio_amd.ImageProcessor(cfg, ctx)
    assert fe.initialize()
    bcfg = S.backend_config(sw_size=args.sw_size, max_features=args.max_features)
    # the filter gets its own context (= its own HIP stream): its update overlaps the next frames' front-end
    ctx_be = ctx if args.sequential else larvio_amd.Context(local_rank)
    be = larvio_amd.LarVio(bcfg, ctx_be)
    assert be.initialize()
    stride = frames.shape[2]
    fsz = frames.shape[1] * frames.shape[2]
    # the driver's IMU buffer (app/larvioMain.cpp:98-102): samples with t < t_img + 0.05 are appended, processFeatures erases
    k_lo = max(int(ts[0] * 200) - 4, 0)
    imu_all = seq.imu_array(k_lo, int(ts[-1] * 200) + 40)
    state = {"inited": False, "n_be": 0}

    def R2q(R):
        t = np.trace(R); s_ = np.sqrt(t + 1) * 2
        return np.array([(R[2, 1] - R[1, 2]) / s_, (R[0, 2] - R[2, 0]) / s_, (R[1, 0] - R[0, 1]) / s_, 0.25 * s_])

    from larvio_amd.vio import VioDriver, VioPipeline
    drv = VioDriver(fe, be, imu_all) if args.sequential else VioPipeline(fe, be, imu_all)
    his = [drv.visible_end(float(t)) for t in ts]

    def step(i):
        # one C-ABI call = the reference driver's loop body (processImage; processFeatures when a message came out)
        if not state["inited"] and i >= 1:
            # the initializers are a cold path outside the scope: the filter starts from ground truth at the
            # second frame, before the front-end's first feature message
            k = int(np.searchsorted(imu_all["t"], ts[i], side="right")) - 1
            t_i = imu_all["t"][k]; tr = seq.traj
            if not args.sequential:
                drv.drain()
            be.set_state(t_i, R2q(tr.R_wb(t_i)), tr.p_wb(t_i), tr.vel(t_i), np.zeros(3), np.zeros(3), imu_all["gyro"][k], imu_all["acc"][k])
            state["inited"] = True
        if args.sequential:
            has, upd = drv.step(float(ts[i]), his[i], device_ptr=d_frames.data_ptr() + i * fsz, stride=stride)
            ctx.sync()
        else:
            # returns when this frame's front-end is done (tracks + message); the update it triggers runs behind it
            has = upd = drv.step(float(ts[i]), his[i], device_ptr=d_frames.data_ptr() + i * fsz, stride=stride)
        if upd:
            state["n_be"] += 1
        return has, upd
    for i in range(W):
        if i == W // 2:
            be.profile(True)                             # H P GEMM bracket (MFMA utilisation): second half of the warm-up only,
        step(i)                                          # its event records would add ~10% to the filter chain of the timed region
    if not args.sequential:
        drv.drain(); drv.stats(reset=True)
    hp = be.profile(False)
    pl0, it0 = fe.lk_stats()
    fe.profile_enable((1 << 2) | (1 << 3))               # HIP events around the LK launches only (dominant kernel)
    state["n_be"] = 0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    lat = np.empty(K); n_msgs = 0; n_tracks = []; upd_mask = np.zeros(K, bool)
    t_begin = time.perf_counter()
    for k in range(K):
        t0 = time.perf_counter()
        have, upd = step(W + k)
        lat[k] = time.perf_counter() - t0
        n_msgs += int(have); upd_mask[k] = upd
    if not args.sequential:
        drv.drain()                                      # every queued filter update has completed
    ctx.sync(); ctx_be.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t_begin
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = fe.profile_read()
    pl1, it1 = fe.lk_stats()
    pst = None if args.sequential else drv.stats()

    if rank == 0:
        # ---- roofline of the dominant kernel family (pyramidal LK): algorithmic bytes per SURVEY §8d
        win = cfg["patch_size"]
        lk_bytes = (pl1 - pl0) * (win + 3) ** 2 + (it1 - it0) * (win + 1) ** 2
        lk_ms = prof["lk_fwd_rev"][0]
        lk_launches = prof["lk_fwd_rev"][1]
        achieved = (lk_bytes / max(lk_launches, 1)) / (lk_ms / max(lk_launches, 1) * 1e-3) / 1e9 if lk_ms > 0 else 0.0
        # HBM traffic per launch: not measurable inside this process (PMC needs rocprofv3) - taken from the committed counter pass
        # of this same command (profiles/*_pmc_fetch_size.csv, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)
        traffic, traffic_src = None, None
        import glob
        pm = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_fetch_size.csv")))
        if pm:
            with open(pm[-1]) as f:
                rows = [l.strip().split(",") for l in f.readlines()[1:]]
            lk = [(int(r[1]), float(r[3])) for r in rows if r[0].startswith("k_fe_lk_")]
            if lk:
                traffic = round(sum(n * b for n, b in lk) / sum(n for n, _ in lk), 1); traffic_src = os.path.basename(pm[-1])
        roofline = {"kernel": "k_fe_lk_both<21> (forward + reverse LK of every track)", "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                    "bytes_per_launch": round(lk_bytes / max(lk_launches, 1), 1), "avg_launch_us": round(lk_ms / max(lk_launches, 1) * 1e3, 3),
                    "launches": lk_launches}
        # ---- CPU baseline: the oracle (a restatement, "port") on this box's host cores, 1 thread, bounded sample
        from oracle import lvo, lvo_be
        nb = min(args.cpu_baseline_frames, K + W) if world == 1 else 0       # rank 0 at N = 1 only
        ora = lvo.Frontend(cfg); orb = lvo_be.Ekf(bcfg)
        lo, inited, c_fe, c_be = 0, False, 0.0, 0.0
        t0 = time.perf_counter()
        for i in range(nb):
            buf = imu_all[lo:his[i]]
            if not inited and i >= 1:
                k = int(np.searchsorted(imu_all["t"], ts[i], side="right")) - 1
                t_i = imu_all["t"][k]; tr = seq.traj
                orb.set_state(t_i, R2q(tr.R_wb(t_i)), tr.p_wb(t_i), tr.vel(t_i), np.zeros(3), np.zeros(3), imu_all["gyro"][k], imu_all["acc"][k])
                inited = True
            ta = time.perf_counter()
            have, m = ora.process(frames[i], float(ts[i]), buf)
            tb = time.perf_counter(); c_fe += tb - ta
            if have:
                ok, used = orb.process(float(ts[i]), m, buf); lo += used
                c_be += time.perf_counter() - tb
        cpu_s = time.perf_counter() - t0
        cpu_baseline = None if nb == 0 else {"value": round(nb / cpu_s, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                        "front_end_ms_per_frame": round(c_fe / nb * 1e3, 3), "back_end_ms_per_frame": round(c_be / nb * 1e3, 3),
                        "sample": f"the first {nb} frames of the same synthetic sequence ({cpu_s:.1f} s of CPU work), CPU oracle front-end + back-end, "
                                  f"1 thread of {os.cpu_count()} (LARVIO is single-threaded); a restatement, not the Eigen/OpenCV build"}
        value = world * K / elapsed
        out = {"metric": "VIO frames/sec (752x480, ~150 tracks, 30-clone window)", "value": round(value, 2), "unit": "frames/s",
               "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 4),
               "p50_ms_per_frame": round(float(np.median(lat)) * 1e3, 4), "p95_ms_per_frame": round(float(np.percentile(lat, 95)) * 1e3, 4),
               "p50_ms_frame_without_update": round(float(np.median(lat[~upd_mask])) * 1e3, 4) if (~upd_mask).any() else None,
               "p50_ms_frame_with_update": round(float(np.median(lat[upd_mask])) * 1e3, 4) if upd_mask.any() else None,
               "front_end_ms_per_frame": None if pst is None else round(pst["front_end_us"] / K * 1e-3, 4),
               "back_end_ms_per_update": None if pst is None else round(pst["filter_us"] / max(state["n_be"], 1) * 1e-3, 4),
               "caller_wait_ms_per_frame": None if pst is None else round(pst["caller_wait_us"] / K * 1e-3, 4),
               "worker_idle_ms_per_update": None if pst is None else round(pst["worker_idle_us"] / max(state["n_be"], 1) * 1e-3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/f32 front-end, f64 back-end",
               "data": "synthetic",
               "config": {"workload": "configs[1] shape: EuRoC-shaped synthetic 752x480 @20Hz, max_features %d, pyramid 3 levels, win 21, pub 10 Hz" % args.max_features,
                          "schedule": ("sequential: one blocking lvk_vio_process per frame" if args.sequential else
                                       "pipelined: filter update of frame k (own stream + worker thread) overlaps the front-end of frames k+1..; "
                                       "identical results (tests/test_gpu_vio_driver.py); per-frame times are front-end completion times, "
                                       "all updates drained inside the timed region"),
                          "stages": "processImage every frame + processFeatures on every feature message (10 Hz), as app/larvioMain.cpp:106-116",
                          "sw_size": args.sw_size, "state_dim": be.dim, "backend": be.counters(),
                          "live_tracks": int(len(fe.tracks()["ids"])), "messages": n_msgs,
                          "parallelism": "replicas x%d" % world},
               "roofline": roofline,
               # the one GEMM-shaped contraction of the path (P H^T as H P, FP64 MFMA 16x16x4): utilisation against the dense FP64 matrix peak
               "roofline_mfma": {"kernel": "k_dgemm<false,false> (H P)", "bound": "mfma", "achieved": round(hp["flops"] / max(hp["ms"], 1e-9) / 1e9, 4),
                                 "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(hp["flops"] / max(hp["ms"], 1e-9) / 1e9 / FP64_MFMA_PEAK_TFLOPS, 6),
                                 "flops_per_launch": round(hp["flops"] / max(hp["launches"], 1), 1),
                                 "avg_launch_us": round(hp["ms"] / max(hp["launches"], 1) * 1e3, 3), "launches": hp["launches"],
                                 "measured_over": "the second half of the warm-up frames (same workload; kept out of the timed region)"},
               "cpu_baseline": cpu_baseline}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
