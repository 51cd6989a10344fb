"""The N>1 path on CPU: two gloo ranks shard a tall measurement block by contiguous feature-id ranges, reduce locally,
all-gather the packed triangles and reduce the rank-ordered stack; every rank must hold the same R, and it must carry the
same information (R^T R, R^T rhs) as the unsharded block.  The per-rank QR here is numpy (the checker); on the GPU it is
lvk_ekf_compress_qr (tests/test_gpu_backend.py covers that kernel)."""
import os
import numpy as np
import pytest


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from larvio_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7)                      # same data on every rank
    n = 40
    counts = rng.integers(1, 12, 60)                    # rows per feature (2M-3), ascending id order
    H = rng.normal(0, 1, (int(counts.sum()), n)); H[:, :15] = 0.0
    r = rng.normal(0, 1, int(counts.sum()))
    lo, hi = sharding.shard_ranges(counts, world)[rank]
    a, b = int(counts[:lo].sum()), int(counts[:hi].sum())
    Q, R = np.linalg.qr(H[a:b], mode="reduced")          # local reduction (rows may be fewer than n)
    stack, rhs = sharding.allgather_triangles(np.triu(R), Q.T @ r[a:b], n)
    assert stack.shape == (world * n, n)
    Q2, R2 = np.linalg.qr(stack, mode="reduced")
    rhs2 = Q2.T @ rhs
    q.put((rank, lo, hi, R2.T @ R2, R2.T @ rhs2, H.T @ H, H.T @ r))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_are_contiguous_and_balanced():
    from larvio_amd import sharding
    counts = np.array([9] * 100 + [1] * 50)
    for world in (1, 2, 4, 8):
        rg = sharding.shard_ranges(counts, world)
        assert rg[0][0] == 0 and rg[-1][1] == len(counts)
        assert all(rg[i][1] == rg[i + 1][0] for i in range(world - 1))
        loads = [counts[a:b].sum() for a, b in rg]
        assert max(loads) - min(loads) <= 2 * counts.max()
    assert sharding.shard_ranges([], 4) == [(0, 0)] * 4


def test_pack_roundtrip():
    from larvio_amd import sharding
    rng = np.random.default_rng(1)
    R = np.triu(rng.normal(0, 1, (7, 7))); rhs = rng.normal(0, 1, 7)
    buf = sharding.pack_upper(R, rhs)
    assert len(buf) == sharding.packed_len(7) == 35
    R2, r2 = sharding.unpack_upper(buf, 7)
    assert np.array_equal(R2, R) and np.array_equal(r2, rhs)
    R3, _ = sharding.unpack_upper(sharding.pack_upper(R[:3], rhs[:3]), 7)     # a rank with fewer rows than columns
    assert np.array_equal(R3[:3], R[:3]) and not R3[3:].any()


def test_two_rank_allgather_of_triangles_preserves_information():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, lo0, hi0, G0, g0, Gt, gt), (r1, lo1, hi1, G1, g1, _, _) = res
    assert hi0 == lo1 and lo0 == 0
    assert np.array_equal(G0, G1) and np.array_equal(g0, g1)                   # replicas agree bit-for-bit (rank-ordered stack)
    assert np.abs(G0 - Gt).max() < 1e-10 * np.abs(Gt).max()
    assert np.abs(g0 - gt).max() < 1e-10 * np.abs(gt).max()
