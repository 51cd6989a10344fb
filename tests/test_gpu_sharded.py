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
        assert st["sharded_updates"] >= 30 and st["exchanges"] >= st["sharded_updates"] and st["exchanges"] == r[8] and st["bytes_sent"] > 0
    assert 0.25 < a[7]["rows_stacked"] / max(a[7]["rows_stacked"] + b[7]["rows_stacked"], 1) < 0.75   # the rows really were split
    print("sharded x2 on one GPU: updates", n_ref, "worst rel vs unsharded", worst, rel(a[3], P_ref), a[7], b[7])
