"""Multi-GPU sharding of the MSCKF measurement block — host logic.

Per-feature Jacobian blocks are independent given the prior covariance, so for very large feature counts
(config 5: 2000 tracks, 60 clones, up to 18,000 raw rows) each rank reduces the rows of a CONTIGUOUS range of
feature ids (the reference stacks rows in ascending id order, larvio.cpp:2185-2201) to its n x n triangle R_g and
n-vector Q^T r, the ranks exchange them with ONE all-gather of the packed upper triangle (n(n+1)/2 + n doubles:
166 KB at n = 202, 588 KB at n = 382 — xGMI is point-to-point, so every rank puts its block on each of its 7 links
once; latency-bound), and every rank QR-reduces the rank-ordered stack [R_0; ...; R_{G-1}] and performs the
identical update on its replica of P.  Rank order fixes the reduction order, so replicas stay bit-identical.
At the north-star size (150 tracks) this does not pay: bench.py runs replicas instead.

The PRODUCT path is inside liblvk_hip.so (lvk_ekf_set_shard, backend.hip shard_stage1, be_shard.hip): per-rank feature rows ->
structure-aware TSQR -> pack kernel -> ncclAllGather on the filter's stream, device buffers in place -> unpack kernel -> replicated
second stage and update.  What crosses the wire is the rank's compressed block as rows (k_g x (N + 1) doubles, k_g ~ 50..200 at
configs[4]) plus the gate results of its features - smaller than the n x n triangle above, which the early-design helpers at the
bottom of this module still describe (kept for the CPU gloo test).  This module makes the transports:

  make_shard(rank, world, dist, device)   RCCL: the 128-byte ncclUniqueId is broadcast over torch.distributed, every rank creates
                                          its communicator in the library (lvk_shard_comm_create) -> arguments for LarVio.set_shard
  HostExchange(ctx, dist)                 a transport for tests: device -> host, torch.distributed.all_gather (gloo), host -> device
"""
import ctypes as C

import numpy as np

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class RcclShard:
    """One rank's RCCL communicator inside liblvk_hip.so; pass .args() to LarVio.set_shard."""

    def __init__(self, ctx, rank, world, uid):
        from ._lib import lib
        L = lib()
        L.lvk_shard_comm_create.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]; L.lvk_shard_comm_create.restype = C.c_int
        L.lvk_shard_comm_destroy.argtypes = [C.c_void_p]; L.lvk_shard_comm_destroy.restype = None
        h = C.c_void_p()
        ctx.check(L.lvk_shard_comm_create(ctx.h, bytes(uid), rank, world, C.byref(h)))
        self._h, self.rank, self.world, self._L = h, rank, world, L

    def args(self):
        fn = C.cast(self._L.lvk_shard_allgather_rccl, C.c_void_p)
        return self.rank, self.world, fn, self._h, self

    def close(self):
        if self._h:
            self._L.lvk_shard_comm_destroy(self._h); self._h = None


def unique_id():
    from ._lib import lib, LvkError
    L = lib()
    L.lvk_shard_unique_id.argtypes = [C.c_char_p]; L.lvk_shard_unique_id.restype = C.c_int
    buf = C.create_string_buffer(128)
    st = L.lvk_shard_unique_id(buf)
    if st != 0:
        raise LvkError("lvk_shard_unique_id failed with status %d (librccl not available?)" % st)
    return buf.raw


def make_shard(ctx, rank, world, dist):
    """RCCL transport for the filter on `ctx`: rank 0 draws the unique id, torch.distributed carries it to the others."""
    import torch
    if dist.get_backend() != "nccl":                      # rehearsal on one GPU (bench.py LVK_BENCH_BACKEND=gloo): RCCL refuses two ranks per device
        hx = HostExchange(ctx, dist, rank, world)
        hx.close = lambda: None
        return hx
    if rank == 0:
        uid = torch.tensor(list(unique_id()), dtype=torch.uint8)
    else:
        uid = torch.zeros(128, dtype=torch.uint8)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    uid = uid.to(dev)
    dist.broadcast(uid, src=0)
    return RcclShard(ctx, rank, world, bytes(uid.cpu().tolist()))


class HostExchange:
    """All-gather through host memory and torch.distributed (any backend) - the transport of the two-process single-GPU test
    (RCCL refuses two ranks on one device).  Same contract as lvk_shard_allgather_rccl, just slower."""

    def __init__(self, ctx, dist, rank, world):
        self.ctx, self.dist, self.rank, self.world = ctx, dist, rank, world
        self.calls = 0; self.bytes = 0
        self._cb = EXCHANGE_FN(self._call)

    def _call(self, user, d_send, d_recv, nbytes, stream):
        import torch
        from ._lib import lib
        try:
            L = lib()
            if nbytes == 0:
                # abort (lvk_exchange_fn contract: this rank failed before it knew the exchange's size).  torch.distributed has no
                # per-collective abort on gloo: announce it in the size handshake below, the peers' exchange then returns an error
                self.dist.all_gather([torch.empty(1, dtype=torch.int64) for _ in range(self.world)], torch.tensor([-1], dtype=torch.int64))
                return 2
            sizes = [torch.empty(1, dtype=torch.int64) for _ in range(self.world)]
            self.dist.all_gather(sizes, torch.tensor([nbytes], dtype=torch.int64))
            if any(int(x) != nbytes for x in sizes):
                print("HostExchange: a peer aborted or disagrees about the block size:", [int(x) for x in sizes])
                return 2
            self.ctx.sync()                                                   # everything queued before the exchange has run
            mine = np.empty(nbytes, np.uint8)
            self.ctx.check(L.lvk_memcpy_d2h(self.ctx.h, mine.ctypes.data_as(C.c_void_p), C.c_void_p(d_send), nbytes))
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
            self.dist.all_gather(parts, torch.from_numpy(mine))
            allb = np.concatenate([p.numpy() for p in parts])
            self.ctx.check(L.lvk_memcpy_h2d(self.ctx.h, C.c_void_p(d_recv), allb.ctypes.data_as(C.c_void_p), allb.nbytes))
            self.ctx.sync()
            self.calls += 1; self.bytes += nbytes
            return 0
        except Exception as exc:                                             # never let an exception cross the C frame
            print("HostExchange failed:", exc)
            return 2

    def args(self):
        return self.rank, self.world, C.cast(self._cb, C.c_void_p), None, self


# ------------------------------------------------------------------ early-design helpers (packed-triangle wire format; CPU gloo test)


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
