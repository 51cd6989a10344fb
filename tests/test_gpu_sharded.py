"""The sharded measurement update as a product path (lvk_ekf_set_shard; SURVEY 8e, BASELINE.json configs[4]) on ONE GPU: two
processes = two ranks share device 0, each runs the whole filter on the same simulated feature messages and does the per-feature
device work of its own contiguous slice; the all-gather of the compressed blocks goes through the host over gloo (RCCL refuses two
ranks on one device - on a multi-GPU node bench.py --sharded uses lvk_shard_allgather_rccl instead, same pack / unpack kernels).
Required: both ranks end with IDENTICAL bits (state, covariance, clone and feature ids, gate counters), equal to the unsharded
filter within the parity tolerance, and the exchange really happened."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sim():
    from tests import feature_sim as F
    return F.simulate(11, t0=2.0, t1=6.0, max_feat=900, n_per_batch=260, sw_size=24, max_features_in_one_grid=1, estimate_td=1, estimate_extrin=1,
                      max_features=900)


def _run(be, sim):
    from tests import feature_sim as F
    out = []
    n = F.drive(_Wrap(be), sim, on_update=lambda ts: None)
    s = be.state()
    return n, s, be.cov(), be.clones()["id"].copy(), be.features()[0].copy(), be.counters()


class _Wrap:
    """the oracle's (process, set_state) spelling on top of larvio_amd.LarVio"""

    def __init__(self, be):
        self.be = be

    def set_state(self, *a):
        self.be.set_state(*a)

    def process(self, ts, m, imu):
        upd, rest = self.be.processFeatures((ts, m), imu)
        return upd, len(imu) - len(rest)


def _worker(rank, world, port, q):
    try:
        import torch.distributed as dist
        import larvio_amd
        from larvio_amd import sharding
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        sim = _sim()
        ctx = larvio_amd.Context(0)
        be = larvio_amd.LarVio(sim["cfg"], ctx); assert be.initialize()
        ex = sharding.HostExchange(ctx, dist, rank, world)
        be.set_shard(*ex.args())
        n, s, P, cid, fid, cnt = _run(be, sim)
        st = be.shard_stats()
        q.put((rank, n, {k: np.array(v) for k, v in s.items()}, P, cid, fid, cnt, st, ex.calls))
        dist.barrier()
        be.close(); ctx.close()
        dist.destroy_process_group()
    except Exception as exc:                                  # surface the failure instead of a queue timeout
        import traceback
        q.put((rank, "error", traceback.format_exc()))
        raise


def test_two_ranks_on_one_gpu_are_bit_identical_and_match_the_unsharded_filter(gpu_ctx):
    import torch.multiprocessing as mp
    import larvio_amd
    sim = _sim()
    ref = larvio_amd.LarVio(sim["cfg"], gpu_ctx); assert ref.initialize()
    n_ref, s_ref, P_ref, cid_ref, fid_ref, cnt_ref = _run(ref, sim)
    ref.close()
    assert n_ref >= 38 and cnt_ref["hybrid"] >= 30
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    for _ in range(2):
        item = q.get(timeout=600)
        assert item[1] != "error", item[2]
        res.append(item)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    a, b = res
    assert a[1] == b[1] == n_ref
    for k in a[2]:
        assert np.array_equal(a[2][k], b[2][k]), k                                # replicas: identical bits
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5]) and a[6] == b[6]
    rel = lambda x, y: float(np.abs(np.asarray(x) - np.asarray(y)).max() / max(np.abs(np.asarray(y)).max(), 1e-300))
    worst = max(rel(a[2][k], s_ref[k]) for k in ("q", "v", "p", "bg", "ba", "R_b2c", "t_c_b"))
    assert worst < 1e-6 and rel(a[3], P_ref) < 1e-6, (worst, rel(a[3], P_ref))   # vs the unsharded filter: another reduction tree, same information
    assert np.array_equal(a[4], cid_ref) and np.array_equal(a[5], fid_ref)
    for k in ("hybrid", "msckf", "gated_in", "gated_out", "map"):
        assert a[6][k] == cnt_ref[k], (k, a[6], cnt_ref)
    for r in (a, b):
        st = r[7]
        # ONE all-gather per sharded update, also when new features enter the state (their gate is computed on every rank)
        assert st["sharded_updates"] >= 30 and st["exchanges"] == st["sharded_updates"] and st["exchanges"] == r[8] and st["bytes_sent"] > 0
    assert 0.25 < a[7]["rows_stacked"] / max(a[7]["rows_stacked"] + b[7]["rows_stacked"], 1) < 0.75   # the rows really were split
    print("sharded x2 on one GPU: updates", n_ref, "worst rel vs unsharded", worst, rel(a[3], P_ref), a[7], b[7])


def test_rccl_transport_executes_at_world_one(gpu_ctx):
    """The built-in transport on hardware: lvk_shard_unique_id -> lvk_shard_comm_create(rank 0, world 1) (ncclCommInitRank through the
    late-bound librccl) -> lvk_shard_allgather_rccl on device buffers, enqueued on the context's own stream; the bytes must arrive."""
    import ctypes as C
    from larvio_amd import sharding
    from larvio_amd._lib import lib
    uid = sharding.unique_id()
    assert len(uid) == 128 and any(uid)
    sh = sharding.RcclShard(gpu_ctx, 0, 1, uid)
    L = lib()
    L.lvk_shard_allgather_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]; L.lvk_shard_allgather_rccl.restype = C.c_int
    L.lvk_shard_comm_error.argtypes = [C.c_void_p]; L.lvk_shard_comm_error.restype = C.c_char_p
    rng = np.random.default_rng(3)
    for n in (256, 600_000):
        src = rng.integers(0, 256, n).astype(np.uint8)
        d_s = gpu_ctx.to_device(src); d_r = gpu_ctx.to_device(np.zeros(n, np.uint8))
        rc = L.lvk_shard_allgather_rccl(sh._h, C.c_void_p(d_s.ptr), C.c_void_p(d_r.ptr), n, C.c_void_p(gpu_ctx.stream))
        assert rc == 0, L.lvk_shard_comm_error(sh._h)
        gpu_ctx.sync()
        assert np.array_equal(gpu_ctx.to_host(d_r, np.uint8, (n,)), src)
    sh.close()


def test_sharded_filter_through_rccl_at_world_one_equals_the_unsharded_filter(gpu_ctx):
    """lvk_ekf_set_shard(rank 0, world 1, lvk_shard_allgather_rccl): the filter takes the sharded branch - per-rank rows, first
    compression stage, k_shard_pack -> ncclAllGather -> k_shard_unpack, replicated second stage - with RCCL as the transport on one
    GPU, through 40 updates including the ones that admit new in-state features, and must agree with the unsharded filter
    (another reduction tree, same information) with identical discrete decisions; one exchange per sharded update."""
    import larvio_amd
    from larvio_amd import sharding
    sim = _sim()
    ref = larvio_amd.LarVio(sim["cfg"], gpu_ctx); assert ref.initialize()
    n_ref, s_ref, P_ref, cid_ref, fid_ref, cnt_ref = _run(ref, sim)
    ref.close()
    be = larvio_amd.LarVio(sim["cfg"], gpu_ctx); assert be.initialize()
    sh = sharding.RcclShard(gpu_ctx, 0, 1, sharding.unique_id())
    be.set_shard(*sh.args())
    n, s, P, cid, fid, cnt = _run(be, sim)
    st = be.shard_stats()
    be.close(); sh.close()
    rel = lambda x, y: float(np.abs(np.asarray(x) - np.asarray(y)).max() / max(np.abs(np.asarray(y)).max(), 1e-300))
    worst = max(rel(s[k], s_ref[k]) for k in ("q", "v", "p", "bg", "ba", "R_b2c", "t_c_b"))
    assert n == n_ref and worst < 1e-7 and rel(P, P_ref) < 1e-7, (n, n_ref, worst, rel(P, P_ref))
    assert np.array_equal(cid, cid_ref) and np.array_equal(fid, fid_ref)
    for k in ("hybrid", "msckf", "gated_in", "gated_out", "map"):
        assert cnt[k] == cnt_ref[k], (k, cnt, cnt_ref)
    assert st["sharded_updates"] >= 30 and st["exchanges"] == st["sharded_updates"] and st["bytes_sent"] > 0 and st["rows_stacked"] > 0, st
    print("sharded through RCCL at world 1: updates", n, "worst rel vs unsharded", worst, rel(P, P_ref), st)
