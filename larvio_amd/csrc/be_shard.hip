// be_shard.hip — the exchange step of the sharded measurement update (BASELINE.json configs[4]): every rank builds
// and gates the Jacobian rows of its contiguous slice of the features, reduces them to a short block of R rows (structure-aware
// TSQR, be_qr.hip), and ONE all-gather hands every rank all blocks (plus the gate results of every feature) in rank order; the
// rest of the update is replicated and bit-identical across ranks.  This file holds the pack / unpack kernels around the
// collective and the built-in RCCL transport (ncclAllGather on the filter's own stream, straight on device buffers).
// The reference has no counterpart: it is single-process (the rows it stacks at larvio.cpp:2185-2201 are the ones sharded here,
// in the same order).
#include "lvk_internal.h"
#include "be_dev.h"
#include <dlfcn.h>
#include <limits.h>
#include <stdlib.h>
#include <string>

struct ShardMeta { int job_lo, job_n, k, row_off; };

// Every rank's block starts with a header the receivers check: a rank that failed locally after the (symmetric) capacity checks
// still enters the collective - with a poisoned header - so that its peers return an error instead of waiting for it forever.
#define LVK_SHARD_HDR 256
#define LVK_SHARD_MAGIC 0x4c564b58u      // "LVKX"
struct ShardHeader { unsigned magic; int rank; int k; int n_res; };

// send layout: [ header (256 B) | FeatResult x res_cap | k rows of (ncols + 1) doubles: H row, then the residual ]
__global__ void __launch_bounds__(256) k_shard_pack(const FeatResult* __restrict__ res, int n_res, const double* __restrict__ X, int ld,
                                                   const double* __restrict__ rX, int k, int ncols, char* __restrict__ send, size_t res_bytes, int rank)
{
    const int b = blockIdx.x, t = threadIdx.x;
    if (b == 0 && t == 0) { ShardHeader h; h.magic = LVK_SHARD_MAGIC; h.rank = rank; h.k = k; h.n_res = n_res; *(ShardHeader*)send = h; }
    send += LVK_SHARD_HDR;
    if (b < k) {
        double* dst = (double*)(send + res_bytes) + (size_t)b * (ncols + 1);
        const double* src = X + (size_t)b * ld;
        for (int j = t; j < ncols; j += 256) dst[j] = src[j];
        if (t == 0) dst[ncols] = rX[b];
    } else {
        double* dst = (double*)send; const double* src = (const double*)res;
        const size_t n = (size_t)n_res * (sizeof(FeatResult) / sizeof(double));
        for (size_t j = (size_t)(b - k) * 256 + t; j < n; j += (size_t)(gridDim.x - k) * 256) dst[j] = src[j];
    }
}

// grid (k_max + 1, world): block (i, g) copies row i of rank g's block to its place in the stacked matrix; block (k_max, g) checks
// rank g's header (a bad one raises *peer_fail, a device-mapped host word the filter reads at its next sync) and copies rank g's
// gate results to the job array and to its device-mapped host mirror
__global__ void __launch_bounds__(256) k_shard_unpack(const char* __restrict__ recv, size_t bytes_per_rank, size_t res_bytes, const ShardMeta* __restrict__ meta,
                                                     int ncols, int k_max, FeatResult* __restrict__ fout, FeatResult* __restrict__ fout_host,
                                                     double* __restrict__ H, int ld, double* __restrict__ r, int* __restrict__ peer_fail)
{
    const int i = blockIdx.x, g = blockIdx.y, t = threadIdx.x;
    const ShardMeta m = meta[g];
    const char* base = recv + (size_t)g * bytes_per_rank;
    const ShardHeader hd = *(const ShardHeader*)base;
    const bool good = hd.magic == LVK_SHARD_MAGIC && hd.rank == g && hd.k == m.k && hd.n_res == m.job_n;
    base += LVK_SHARD_HDR;
    if (i < k_max) {
        if (i >= m.k) return;
        double* dst = H + (size_t)(m.row_off + i) * ld;
        if (!good) { for (int j = t; j < ncols; j += 256) dst[j] = 0.0; if (t == 0) r[m.row_off + i] = 0.0; return; }
        const double* src = (const double*)(base + res_bytes) + (size_t)i * (ncols + 1);
        for (int j = t; j < ncols; j += 256) dst[j] = src[j];
        if (t == 0) r[m.row_off + i] = src[ncols];
    } else {
        if (!good) { if (t == 0 && peer_fail) atomicOr(peer_fail, 1 << (g < 31 ? g : 31)); return; }
        const FeatResult* src = (const FeatResult*)base;
        for (int j = t; j < m.job_n; j += 256) { const FeatResult v = src[j]; fout[m.job_lo + j] = v; if (fout_host) fout_host[m.job_lo + j] = v; }
    }
}

lvk_status lvk_shard_pack(lvk_context* ctx, const FeatResult* d_res, int n_res, const double* d_X, int ld, const double* d_rX, int k, int ncols,
                          char* d_send, size_t res_bytes, int rank)
{
    const int extra = n_res > 0 ? (n_res * 4 + 255) / 256 : 0;
    const int grid = k + (extra > 0 ? (extra > 64 ? 64 : extra) : 0);
    hipLaunchKernelGGL(k_shard_pack, dim3(grid > 0 ? grid : 1), dim3(256), 0, ctx->stream, d_res, n_res, d_X, ld, d_rX, k, ncols, d_send, res_bytes, rank);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}
lvk_status lvk_shard_unpack(lvk_context* ctx, const char* d_recv, size_t bytes_per_rank, size_t res_bytes, const ShardMeta* d_meta, int world, int ncols, int k_max,
                            FeatResult* d_fout, FeatResult* d_fout_host, double* d_H, int ld, double* d_r, int* d_peer_fail)
{
    hipLaunchKernelGGL(k_shard_unpack, dim3(k_max + 1, world), dim3(256), 0, ctx->stream, d_recv, bytes_per_rank, res_bytes, d_meta, ncols, k_max, d_fout, d_fout_host, d_H, ld, d_r, d_peer_fail);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

// ------------------------------------------------------------------------- built-in transport: RCCL (backend "nccl" on ROCm)
// librccl is bound at run time (dlopen), so liblvk_hip.so loads on boxes without it; a copy a host process such as PyTorch has
// already loaded is shared (same SONAME).  Types and signatures come from RCCL's own header - only the symbols are late-bound.
#include <rccl/rccl.h>
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclCommAbort) comm_abort = nullptr;
    decltype(&ncclGetErrorString) err_string = nullptr;
    char why[160] = {0};
    char path[512] = {0};                               // the librccl that was bound
};
static const RcclApi* rccl_api()
{   // opened once per process and kept (a communicator may outlive any single call)
    static const RcclApi api = [] {
        RcclApi a;
        // The RCCL that belongs to the HIP runtime THIS library runs on: a process may hold two ROCm stacks (PyTorch wheels bundle
        // their own libamdhip64 / librccl next to /opt/rocm's; which libamdhip64 serves liblvk_hip.so depends on the load order), and
        // a librccl bound to the other runtime fails in ncclCommInitRank ("unhandled cuda error").  So: first the librccl in the
        // directory our hipMalloc comes from, then the loader's default search.
        Dl_info di;
        if (dladdr((void*)&hipGetDeviceCount, &di) && di.dli_fname) {
            char real[4096];
            if (realpath(di.dli_fname, real)) {
                std::string dir(real); const size_t sl = dir.rfind('/'); dir = sl == std::string::npos ? std::string(".") : dir.substr(0, sl);
                for (const char* n : {"/librccl.so.1", "/librccl.so"}) if ((a.lib = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_LOCAL)) != nullptr) { snprintf(a.path, sizeof a.path, "%s%s", dir.c_str(), n); break; }
            }
        }
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (!a.lib) for (const char* n : names) if ((a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) { snprintf(a.path, sizeof a.path, "%s", n); break; }
        if (!a.lib) { const char* er = dlerror(); snprintf(a.why, sizeof a.why, "librccl not found: %s", er ? er : "?"); return a; }
        a.get_unique_id = (decltype(a.get_unique_id))dlsym(a.lib, "ncclGetUniqueId");
        a.comm_init_rank = (decltype(a.comm_init_rank))dlsym(a.lib, "ncclCommInitRank");
        a.all_gather = (decltype(a.all_gather))dlsym(a.lib, "ncclAllGather");
        a.comm_destroy = (decltype(a.comm_destroy))dlsym(a.lib, "ncclCommDestroy");
        a.comm_abort = (decltype(a.comm_abort))dlsym(a.lib, "ncclCommAbort");
        a.err_string = (decltype(a.err_string))dlsym(a.lib, "ncclGetErrorString");
        if (!a.get_unique_id || !a.comm_init_rank || !a.all_gather || !a.comm_destroy) snprintf(a.why, sizeof a.why, "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy");
        return a;
    }();
    return &api;
}
struct lvk_shard_comm { ncclComm_t comm; int rank, world; char err[160]; };

extern "C" {

lvk_status lvk_shard_unique_id(char* h_out128)
{
    if (!h_out128) return LVK_ERR_ARG;
    const RcclApi* a = rccl_api();
    if (a->why[0]) return LVK_ERR_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == 128, "lvk_shard_unique_id hands out 128 bytes");
    ncclUniqueId id;
    if (a->get_unique_id(&id) != ncclSuccess) return LVK_ERR_DEVICE;
    memcpy(h_out128, id.internal, 128);
    return LVK_OK;
}

lvk_status lvk_shard_comm_create(lvk_context* ctx, const char* h_uid128, int rank, int world, lvk_shard_comm** out)
{
    if (!ctx) return LVK_ERR_ARG;
    if (!h_uid128 || !out || world < 1 || rank < 0 || rank >= world) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_shard_comm_create: bad argument");
    const RcclApi* a = rccl_api();
    if (a->why[0]) return lvk_set_error(ctx, LVK_ERR_UNSUPPORTED, "%s", a->why);
    lvk_shard_comm* c = new lvk_shard_comm();
    c->comm = nullptr; c->rank = rank; c->world = world; c->err[0] = 0;
    ncclUniqueId id; memcpy(id.internal, h_uid128, 128);
    if (hipSetDevice(ctx->device) != hipSuccess) { delete c; return lvk_set_error(ctx, LVK_ERR_DEVICE, "lvk_shard_comm_create: hipSetDevice(%d) failed", ctx->device); }
    const ncclResult_t rc = a->comm_init_rank(&c->comm, world, id, rank);
    if (rc != ncclSuccess) { const char* es = a->err_string ? a->err_string(rc) : "?"; delete c; return lvk_set_error(ctx, LVK_ERR_DEVICE, "ncclCommInitRank (%s): %s", a->path, es); }
    *out = c;
    return LVK_OK;
}

void lvk_shard_comm_destroy(lvk_shard_comm* c)
{
    if (!c) return;
    if (c->comm) rccl_api()->comm_destroy(c->comm);
    delete c;
}

const char* lvk_shard_comm_error(const lvk_shard_comm* c) { return c ? c->err : "null communicator"; }
const char* lvk_shard_rccl_path(void) { const RcclApi* a = rccl_api(); return a->why[0] ? a->why : a->path; }

// lvk_exchange_fn over RCCL: user = lvk_shard_comm*.  The collective is enqueued on the caller's stream; nothing blocks the host.
lvk_status lvk_shard_allgather_rccl(void* user, const void* d_send, void* d_recv, size_t bytes_per_rank, void* hip_stream)
{
    lvk_shard_comm* c = (lvk_shard_comm*)user;
    if (!c) return LVK_ERR_ARG;
    const RcclApi* a = rccl_api();
    if (bytes_per_rank == 0) {
        // abort (lvk_exchange_fn contract): this rank cannot take part in the update's collective - break the communicator so that
        // the peers' all-gather ends with an error instead of waiting for it
        if (c->comm && a->comm_abort) { a->comm_abort(c->comm); c->comm = nullptr; }
        snprintf(c->err, sizeof c->err, "communicator aborted: this rank failed before the exchange");
        return LVK_ERR_DEVICE;
    }
    if (!d_send || !d_recv) return LVK_ERR_ARG;
    if (!c->comm) { snprintf(c->err, sizeof c->err, "communicator was aborted"); return LVK_ERR_DEVICE; }
    const ncclResult_t rc = a->all_gather(d_send, d_recv, bytes_per_rank, ncclInt8, c->comm, (hipStream_t)hip_stream);
    if (rc == ncclSuccess) return LVK_OK;
    snprintf(c->err, sizeof c->err, "ncclAllGather: %s", a->err_string ? a->err_string(rc) : "?");
    return LVK_ERR_DEVICE;
}

}  // extern "C"
