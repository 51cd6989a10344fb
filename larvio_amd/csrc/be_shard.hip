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

struct ShardMeta { int job_lo, job_n, k, row_off; };

// send layout: [ FeatResult x res_cap | k_max rows of (ncols + 1) doubles: H row, then the residual ]
__global__ void __launch_bounds__(256) k_shard_pack(const FeatResult* __restrict__ res, int n_res, const double* __restrict__ X, int ld,
                                                   const double* __restrict__ rX, int k, int ncols, char* __restrict__ send, size_t res_bytes)
{
    const int b = blockIdx.x, t = threadIdx.x;
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

// grid (k_max + 1, world): block (i, g) copies row i of rank g's block to its place in the stacked matrix; block (k_max, g) copies
// rank g's gate results to the job array and to its device-mapped host mirror
__global__ void __launch_bounds__(256) k_shard_unpack(const char* __restrict__ recv, size_t bytes_per_rank, size_t res_bytes, const ShardMeta* __restrict__ meta,
                                                     int ncols, int k_max, FeatResult* __restrict__ fout, FeatResult* __restrict__ fout_host,
                                                     double* __restrict__ H, int ld, double* __restrict__ r)
{
    const int i = blockIdx.x, g = blockIdx.y, t = threadIdx.x;
    const ShardMeta m = meta[g];
    const char* base = recv + (size_t)g * bytes_per_rank;
    if (i < k_max) {
        if (i >= m.k) return;
        const double* src = (const double*)(base + res_bytes) + (size_t)i * (ncols + 1);
        double* dst = H + (size_t)(m.row_off + i) * ld;
        for (int j = t; j < ncols; j += 256) dst[j] = src[j];
        if (t == 0) r[m.row_off + i] = src[ncols];
    } else {
        const FeatResult* src = (const FeatResult*)base;
        for (int j = t; j < m.job_n; j += 256) { const FeatResult v = src[j]; fout[m.job_lo + j] = v; if (fout_host) fout_host[m.job_lo + j] = v; }
    }
}

lvk_status lvk_shard_pack(lvk_context* ctx, const FeatResult* d_res, int n_res, const double* d_X, int ld, const double* d_rX, int k, int ncols,
                          char* d_send, size_t res_bytes)
{
    const int extra = n_res > 0 ? (n_res * 4 + 255) / 256 : 0;
    if (k + extra <= 0) return LVK_OK;
    hipLaunchKernelGGL(k_shard_pack, dim3(k + (extra > 0 ? (extra > 64 ? 64 : extra) : 0)), dim3(256), 0, ctx->stream, d_res, n_res, d_X, ld, d_rX, k, ncols, d_send, res_bytes);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}
lvk_status lvk_shard_unpack(lvk_context* ctx, const char* d_recv, size_t bytes_per_rank, size_t res_bytes, const ShardMeta* d_meta, int world, int ncols, int k_max,
                            FeatResult* d_fout, FeatResult* d_fout_host, double* d_H, int ld, double* d_r)
{
    hipLaunchKernelGGL(k_shard_unpack, dim3(k_max + 1, world), dim3(256), 0, ctx->stream, d_recv, bytes_per_rank, res_bytes, d_meta, ncols, k_max, d_fout, d_fout_host, d_H, ld, d_r);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

// ------------------------------------------------------------------------- built-in transport: RCCL (backend "nccl" on ROCm)
// librccl is bound at run time (dlopen), so liblvk_hip.so loads on boxes without it and shares the copy a host process such as
// PyTorch may already have loaded.
typedef struct { char internal[128]; } rccl_unique_id;
typedef void* rccl_comm;
struct lvk_shard_comm {
    void* lib; rccl_comm comm; int rank, world;
    int (*all_gather)(const void*, void*, size_t, int, rccl_comm, hipStream_t);
    int (*comm_destroy)(rccl_comm);
    const char* (*err_string)(int);
};
static void* rccl_open()
{
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) if (void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) return h;
    return nullptr;
}

extern "C" {

lvk_status lvk_shard_unique_id(char* h_out128)
{
    if (!h_out128) return LVK_ERR_ARG;
    void* lib = rccl_open();
    if (!lib) return LVK_ERR_UNSUPPORTED;
    auto get = (int (*)(rccl_unique_id*))dlsym(lib, "ncclGetUniqueId");
    rccl_unique_id id;
    if (!get || get(&id) != 0) return LVK_ERR_DEVICE;
    memcpy(h_out128, id.internal, 128);
    return LVK_OK;
}

lvk_status lvk_shard_comm_create(lvk_context* ctx, const char* h_uid128, int rank, int world, lvk_shard_comm** out)
{
    if (!ctx || !h_uid128 || !out || world < 1 || rank < 0 || rank >= world) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_shard_comm_create: bad argument");
    void* lib = rccl_open();
    if (!lib) return lvk_set_error(ctx, LVK_ERR_UNSUPPORTED, "librccl not found: %s", dlerror());
    auto init = (int (*)(rccl_comm*, int, rccl_unique_id, int))dlsym(lib, "ncclCommInitRank");
    lvk_shard_comm* c = new lvk_shard_comm();
    c->lib = lib; c->rank = rank; c->world = world;
    c->all_gather = (int (*)(const void*, void*, size_t, int, rccl_comm, hipStream_t))dlsym(lib, "ncclAllGather");
    c->comm_destroy = (int (*)(rccl_comm))dlsym(lib, "ncclCommDestroy");
    c->err_string = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
    if (!init || !c->all_gather || !c->comm_destroy) { delete c; return lvk_set_error(ctx, LVK_ERR_UNSUPPORTED, "librccl lacks ncclCommInitRank / ncclAllGather"); }
    rccl_unique_id id; memcpy(id.internal, h_uid128, 128);
    LVK_HIP(ctx, hipSetDevice(ctx->device));
    const int rc = init(&c->comm, world, id, rank);
    if (rc != 0) { const char* es = c->err_string ? c->err_string(rc) : "?"; delete c; return lvk_set_error(ctx, LVK_ERR_DEVICE, "ncclCommInitRank: %s", es); }
    *out = c;
    return LVK_OK;
}

void lvk_shard_comm_destroy(lvk_shard_comm* c)
{
    if (!c) return;
    if (c->comm_destroy && c->comm) c->comm_destroy(c->comm);
    delete c;
}

// lvk_exchange_fn over RCCL: user = lvk_shard_comm*.  The collective is enqueued on the caller's stream; nothing blocks the host.
lvk_status lvk_shard_allgather_rccl(void* user, const void* d_send, void* d_recv, size_t bytes_per_rank, void* hip_stream)
{
    lvk_shard_comm* c = (lvk_shard_comm*)user;
    if (!c || !d_send || !d_recv) return LVK_ERR_ARG;
    return c->all_gather(d_send, d_recv, bytes_per_rank, /*ncclInt8*/ 0, c->comm, (hipStream_t)hip_stream) == 0 ? LVK_OK : LVK_ERR_DEVICE;
}

}  // extern "C"
