"""Multi-GPU sharding of the MSCKF measurement block — host logic.

Per-feature Jacobian blocks are independent given the prior covariance, so for very large feature counts
(config 5: 2000 tracks, 60 clones, up to 18,000 raw rows) each rank reduces the rows of a CONTIGUOUS range of
feature ids (the reference stacks rows in ascending id order, larvio.cpp:2185-2201) to its n x n triangle R_g and
n-vector Q^T r, the ranks exchange them with ONE all-gather of the packed upper triangle (n(n+1)/2 + n doubles:
166 KB at n = 202, 588 KB at n = 382 — xGMI is point-to-point, so every rank puts its block on each of its 7 links
once; latency-bound), and every rank QR-reduces the rank-ordered stack [R_0; ...; R_{G-1}] and performs the
identical update on its replica of P.  Rank order fixes the reduction order, so replicas stay bit-identical.
At the north-star size (150 tracks) this does not pay: bench.py runs replicas instead.

This module is pure host logic over torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" on CPU in tests);
the per-rank numerical work is liblvk_hip.so's lvk_ekf_compress_qr / lvk_ekf_update.
"""
import numpy as np


def shard_ranges(row_counts, world_size):
    """Contiguous [lo, hi) feature ranges per rank, balanced by row count.  row_counts: rows per feature in ascending id order."""
    row_counts = np.asarray(row_counts, np.int64)
    total = int(row_counts.sum())
    bounds = [0]
    csum = np.cumsum(row_counts)
    for g in range(1, world_size):
        target = total * g / world_size
        bounds.append(int(np.searchsorted(csum, target, side="left")) if total else 0)
    bounds.append(len(row_counts))
    for g in range(1, len(bounds)):
        bounds[g] = max(bounds[g], bounds[g - 1])
    return [(bounds[g], bounds[g + 1]) for g in range(world_size)]


def packed_len(n):
    return n * (n + 1) // 2 + n


def pack_upper(R, rhs):
    """n x n upper-triangular R (rows beyond the available ones are zero) + rhs (n) -> 1-D float64 of packed_len(n)."""
    R = np.asarray(R, np.float64); rhs = np.asarray(rhs, np.float64)
    n = R.shape[1]
    full = np.zeros((n, n)); full[:R.shape[0]] = R[:n]
    r = np.zeros(n); r[:len(rhs)] = rhs[:n]
    iu = np.triu_indices(n)
    return np.concatenate([full[iu], r])


def unpack_upper(buf, n):
    buf = np.asarray(buf, np.float64)
    R = np.zeros((n, n)); R[np.triu_indices(n)] = buf[:n * (n + 1) // 2]
    return R, buf[n * (n + 1) // 2:].copy()


def allgather_triangles(R, rhs, n, group=None, device=None):
    """All-gather of the packed triangles; returns (stack (G*n x n), rhs (G*n)) in RANK order on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    mine = torch.from_numpy(pack_upper(R, rhs))
    if device is not None:
        mine = mine.to(device)
    out = torch.empty(world * mine.numel(), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    out = out.cpu().numpy().reshape(world, -1)
    Rs, rs = zip(*(unpack_upper(out[g], n) for g in range(world)))
    return np.concatenate(Rs, 0), np.concatenate(rs, 0)
