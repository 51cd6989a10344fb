// frontend.hip — context management and the frame-level front-end object of liblvk_hip.so.
// lvk_frontend_process replaces ImageProcessor::processImage (/root/reference/src/image_processor.cpp:130-219).
//
// Device-resident design (MI355X): the track table (ids, points, first-seen descriptors, lifetimes),
// both pyramids and both ORB mosaics stay in HBM across frames; every data-dependent count
// (tracks alive, new corners, survivors per stage) lives in a device struct, kernels are launched
// over the full capacity and dead wavefronts exit at once.  The host therefore never waits on the
// GPU inside a steady-state frame except to fetch the feature message on publish frames.
// Stages keep points IN PLACE with a stage code instead of compacting six vectors four times
// (image_processor.h:215-230); the one order-preserving compaction happens inside the RANSAC
// workgroup, which needs the compacted order anyway (OpenCV's RNG draws indices into it).
#include "lvk_internal.h"
#include <atomic>
#include <sched.h>
#include "fe_track_dev.h"
#include <math.h>
#include <float.h>
#include <new>
#include <vector>

lvk_status lvk_gftt_run(lvk_context* ctx, const float* d_eig, const uint8_t* d_mask, int w, int h, int max_corners,
                        double quality, double min_distance, unsigned* d_scratch, unsigned long long* d_cands, int cand_cap,
                        lvk_pt2f* d_out, int cap, int* d_n_out, const int* d_sub, bool prepared, bool max_done);
lvk_status lvk_mask_and_max(lvk_context* ctx, const lvk_pt2f* d_pts, const int* d_n, int w, int h, int md, const float* d_eig, uint8_t* d_mask, unsigned* d_scratch);

// =========================================================================== runtime environment, BAR self-test
// plain loads, as the product kernels do them
__global__ void k_bar_probe(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int n, int stride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[(size_t)i * stride];
}
static bool bar_selftest(int device)
{
    int large_bar = 0;
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) != hipSuccess || !large_bar) { (void)hipGetLastError(); return false; }
    int prev = -1; (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return false; }
    const int n = 256, stride = 64;                          // 256 words, one per 256-byte line, over 64 KB
    unsigned* d = nullptr; unsigned* h = nullptr; unsigned* dh = nullptr; hipStream_t st = nullptr;
    bool ok = hipExtMallocWithFlags((void**)&d, sizeof(unsigned) * n * stride, hipDeviceMallocFinegrained) == hipSuccess && d;
    ok = ok && hipHostMalloc((void**)&h, sizeof(unsigned) * n) == hipSuccess && hipHostGetDevicePointer((void**)&dh, h, 0) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    if (ok) {
        bool writable = false;
        if (FILE* f = fopen("/proc/self/maps", "r")) {
            char line[512]; unsigned long lo = 0, hi = 0; char perm[8] = {0};
            while (fgets(line, sizeof line, f))
                if (sscanf(line, "%lx-%lx %7s", &lo, &hi, perm) == 3 && (unsigned long)d >= lo && (unsigned long)d < hi) { writable = perm[0] == 'r' && perm[1] == 'w'; break; }
            fclose(f);
        }
        ok = writable;
    }
    for (int round = 0; round < 6 && ok; ++round) {           // the same addresses rewritten before every launch, as the filter's arena is every frame
        volatile unsigned* w = (volatile unsigned*)d;
        for (int i = 0; i < n; ++i) w[(size_t)i * stride] = 0x9e3779b9u * (unsigned)(round + 1) + (unsigned)i;
        LVK_STORE_FENCE();
        memset(h, 0, sizeof(unsigned) * n);
        hipLaunchKernelGGL(k_bar_probe, dim3(n / 64), dim3(64), 0, st, (const unsigned*)d, dh, n, stride);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
        for (int i = 0; i < n && ok; ++i) ok = h[i] == 0x9e3779b9u * (unsigned)(round + 1) + (unsigned)i;
    }
    if (st) hipStreamDestroy(st);
    if (h) hipHostFree(h);
    if (d) hipFree(d);
    (void)hipGetLastError();
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    return ok;
}
bool lvk_bar_usable(int device)
{
    static std::atomic<int> verdict[64];                     // 0 unknown, 1 usable, 2 not
    if (device < 0 || device >= 64) return false;
    int v = verdict[device].load(std::memory_order_acquire);
    if (v == 0) {
        const char* sw = getenv("LVK_BAR_PUSH");
        bool ok = !(sw && !strcmp(sw, "0"));
        if (ok) ok = bar_selftest(device);
        else if (getenv("LVK_VERBOSE")) fprintf(stderr, "[lvk] LVK_BAR_PUSH=0: staging through pinned host memory\n");
        if (!ok && !(sw && !strcmp(sw, "0")) && getenv("LVK_VERBOSE")) fprintf(stderr, "[lvk] device %d: BAR push not usable (no large BAR, no writable mapping, or stale reads in the self-test): pinned host staging\n", device);
        v = ok ? 1 : 2;
        verdict[device].store(v, std::memory_order_release);
    }
    return v == 1;
}

// lvk_runtime_env (include/lvk_c.h): what a deployment should set before the HIP runtime initialises, done by the library.
//  * GPU_MAX_HW_QUEUES=8: the front-end's three streams + the filter's (+ the application's own) each get a hardware queue; with HIP's
//    default of 4, streams share queues and serialise (measured: 5800 -> 4800 frames/s).
//  * HIP_FORCE_DEV_KERNARG=1: kernel arguments in device memory (~1 % on these chains of small kernels).
//  * LVK_RT_BIND_L3 (opt-in: a library should not move its caller's threads unasked): the calling thread - and every thread it
//    starts afterwards: the HIP runtime's, the filter's worker - onto the physical cores of ONE L3 group of the socket it runs on
//    (rank r takes the r-th group): the caller's and the filter's thread hand each other messages every ~100 us.
// Variables the user has set are left alone.  Effective only if called before the process's first HIP call; lvk_context_create
// calls it with LVK_RT_DEFAULT itself (LVK_RUNTIME_ENV=0 turns that off), which is early enough when the library makes that first call.
static bool cpulist_parse(const char* path, cpu_set_t* out)
{
    CPU_ZERO(out);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    char list[4096] = {0};
    const bool got = fgets(list, sizeof list, f) != nullptr;
    fclose(f);
    if (!got) return false;
    for (char* p = list; *p && *p != '\n';) {
        char* end = nullptr;
        long a = strtol(p, &end, 10); if (end == p) break;
        long b = a; p = end;
        if (*p == '-') { b = strtol(p + 1, &end, 10); p = end; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET((int)c, out);
        if (*p == ',') ++p;
    }
    return true;
}
static bool bind_l3_group(int local_rank)
{
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    const int cpu = sched_getcpu();
    if (cpu < 0) return false;
    char path[160]; cpu_set_t node; bool have_node = false;
    for (int nd = 0; nd < 64 && !have_node; ++nd) {
        snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", nd);
        if (!cpulist_parse(path, &node)) break;
        have_node = CPU_ISSET(cpu, &node);
    }
    if (!have_node) return false;
    // one logical CPU per physical core, grouped by L3; groups in ascending order of their first CPU
    std::vector<std::pair<int, cpu_set_t>> groups;
    for (int c = 0; c < CPU_SETSIZE; ++c) {
        if (!CPU_ISSET(c, &node) || !CPU_ISSET(c, &allowed)) continue;
        cpu_set_t l3, sib;
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", c);
        if (!cpulist_parse(path, &l3)) return false;
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
        if (!cpulist_parse(path, &sib)) return false;
        int first_sib = -1, first_l3 = -1;
        for (int k = 0; k < CPU_SETSIZE && (first_sib < 0 || first_l3 < 0); ++k) { if (first_sib < 0 && CPU_ISSET(k, &sib)) first_sib = k; if (first_l3 < 0 && CPU_ISSET(k, &l3)) first_l3 = k; }
        if (c != first_sib) continue;
        size_t g = 0; while (g < groups.size() && groups[g].first != first_l3) ++g;
        if (g == groups.size()) { cpu_set_t z; CPU_ZERO(&z); groups.push_back(std::make_pair(first_l3, z)); }
        CPU_SET(c, &groups[g].second);
    }
    std::vector<cpu_set_t> usable;
    for (auto& g : groups) if (CPU_COUNT(&g.second) >= 4) usable.push_back(g.second);
    if (usable.empty()) return false;
    const cpu_set_t& pick = usable[(size_t)(local_rank < 0 ? 0 : local_rank) % usable.size()];
    return sched_setaffinity(0, sizeof pick, &pick) == 0;
}

// =========================================================================== context
extern "C" {

static std::atomic<bool> g_hip_touched{false};          // this library has made a HIP call in this process: the runtime has read its environment
unsigned lvk_runtime_env(unsigned flags, int local_rank)
{
    unsigned done = 0;
    // a flag is reported only if it can still take effect: the variable was not set by the user and the runtime has not been
    // initialised through this library (a host application that initialised HIP itself is beyond what can be seen from here:
    // lvk_c.h says "call it first in main()")
    const bool early = !g_hip_touched.load();
    if ((flags & LVK_RT_HW_QUEUES) && early && !getenv("GPU_MAX_HW_QUEUES")) { if (setenv("GPU_MAX_HW_QUEUES", "8", 0) == 0) done |= LVK_RT_HW_QUEUES; }
    if ((flags & LVK_RT_DEV_KERNARG) && early && !getenv("HIP_FORCE_DEV_KERNARG")) { if (setenv("HIP_FORCE_DEV_KERNARG", "1", 0) == 0) done |= LVK_RT_DEV_KERNARG; }
    if (flags & LVK_RT_BIND_L3) { if (bind_l3_group(local_rank)) done |= LVK_RT_BIND_L3; }
    return done;
}

const char* lvk_version(void) { return "lvk-hip 0.1 (gfx950)"; }

// Optional placement help for dual-socket hosts (LVK_NUMA_BIND=1): restrict the thread that creates a context (and the threads it
// starts later: the pipelined filter's worker) to the CPUs of the device's NUMA node.  Off by default: a library should not move
// its caller's threads, and on the 2 x EPYC 9575F boxes measured here the node itself made no difference — what did was having
// the WHOLE process (HIP runtime initialisation included) on one socket, which is the launcher's business (bench.py does it,
// INTEGRATION.md says how: taskset / numactl).
static void bind_thread_to_device_node(int device)
{
    const char* on = getenv("LVK_NUMA_BIND");
    if (!(on && !strcmp(on, "1"))) return;
    int node = -1;
    if (hipDeviceGetAttribute(&node, hipDeviceAttributeHostNumaId, device) != hipSuccess) { (void)hipGetLastError(); node = -1; }
    char path[160];
    if (node < 0) {                                         // older runtimes: ask sysfs through the PCI address
        char bdf[32] = {0};
        if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) { (void)hipGetLastError(); return; }
        for (char* c = bdf; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
        snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
        if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
        if (node < 0) return;
    }
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return;
    char list[1024] = {0};
    const bool got = fgets(list, sizeof list, f) != nullptr;
    fclose(f);
    if (!got) return;
    cpu_set_t allowed, want; CPU_ZERO(&allowed); CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
    int n_want = 0;
    for (char* p = list; *p && *p != '\n';) {               // "0-63,128-191"
        char* end = nullptr;
        long a = strtol(p, &end, 10); if (end == p) break;
        long b = a; p = end;
        if (*p == '-') { b = strtol(p + 1, &end, 10); p = end; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) if (CPU_ISSET((int)c, &allowed)) { CPU_SET((int)c, &want); ++n_want; }
        if (*p == ',') ++p;
    }
    if (n_want > 0) sched_setaffinity(0, sizeof want, &want);
}

lvk_status lvk_context_create(int device, lvk_context** out)
{
    if (!out) return LVK_ERR_ARG;
    {   // the library's runtime settings, in case this is the process's first HIP call (see lvk_runtime_env)
        static std::atomic<bool> once{false};
        if (!once.exchange(true)) { const char* sw = getenv("LVK_RUNTIME_ENV"); if (!(sw && !strcmp(sw, "0"))) {
            const char* bd = getenv("LVK_RUNTIME_BIND");               // "l3" or "l3:<local rank>": the opt-in core binding for a driver that cannot call lvk_runtime_env (the reference's own main())
            const bool l3 = bd && !strncmp(bd, "l3", 2);
            lvk_runtime_env(LVK_RT_DEFAULT | (l3 ? LVK_RT_BIND_L3 : 0u), l3 && bd[2] == ':' ? atoi(bd + 3) : 0);
        } }
    }
    int count = 0;
    g_hip_touched.store(true);
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return LVK_ERR_DEVICE;   // no CPU fallback
    if (hipSetDevice(device) != hipSuccess) return LVK_ERR_DEVICE;
    // LVK_WAIT_POLICY=spin: keep waiting host threads spinning on the completion signal (the runtime's default spins for 100 us,
    // then sleeps until the interrupt).  Process-wide, so opt-in; two A/B pairs of bench runs showed no consistent difference.
    {
        const char* wp = getenv("LVK_WAIT_POLICY");
        if (wp && !strcmp(wp, "spin")) { if (hipSetDeviceFlags(hipDeviceScheduleSpin) != hipSuccess) (void)hipGetLastError(); }
    }
    bind_thread_to_device_node(device);
    lvk_context* c = new (std::nothrow) lvk_context();
    if (!c) return LVK_ERR_DEVICE;
    memset(c, 0, sizeof *c);
    c->device = device; c->own_stream = true;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return LVK_ERR_DEVICE; }
    *out = c;
    return LVK_OK;
}

void lvk_context_destroy(lvk_context* ctx)
{
    if (!ctx) return;
    hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < LVK_SCRATCH_SLOTS; ++i) if (ctx->scratch[i]) hipFree(ctx->scratch[i]);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

lvk_status lvk_context_set_stream(lvk_context* ctx, void* hip_stream)
{
    if (!ctx) return LVK_ERR_ARG;
    // whatever the objects of this context still have queued on the OLD stream (lvk_ekf_process returns with its tail launches
    // outstanding) must not race what the next call queues on the new one: drain it, also when the caller owns it
    if (ctx->stream) LVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) { hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
    if (hip_stream) ctx->stream = (hipStream_t)hip_stream;
    else { LVK_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->own_stream = true; }
    return LVK_OK;
}

void* lvk_context_get_stream(const lvk_context* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
lvk_status lvk_sync(lvk_context* ctx) { if (!ctx) return LVK_ERR_ARG; LVK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return LVK_OK; }
const char* lvk_last_error(const lvk_context* ctx) { return ctx ? ctx->err : "null context"; }

lvk_status lvk_malloc(lvk_context* ctx, size_t bytes, void** d_out)
{
    if (!ctx || !d_out) return LVK_ERR_ARG;
    LVK_HIP(ctx, hipMalloc(d_out, bytes ? bytes : 1));
    return LVK_OK;
}
lvk_status lvk_free(lvk_context* ctx, void* d_ptr) { if (!ctx) return LVK_ERR_ARG; LVK_HIP(ctx, hipStreamSynchronize(ctx->stream)); LVK_HIP(ctx, hipFree(d_ptr)); return LVK_OK; }
lvk_status lvk_memcpy_h2d(lvk_context* ctx, void* d_dst, const void* h_src, size_t bytes)
{
    if (!ctx) return LVK_ERR_ARG;
    LVK_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    LVK_HIP(ctx, hipStreamSynchronize(ctx->stream));     // pageable source: finish before the caller reuses it
    return LVK_OK;
}
lvk_status lvk_memcpy_d2h(lvk_context* ctx, void* h_dst, const void* d_src, size_t bytes)
{
    if (!ctx) return LVK_ERR_ARG;
    LVK_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    LVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LVK_OK;
}
lvk_status lvk_memset(lvk_context* ctx, void* d_dst, int value, size_t bytes)
{
    if (!ctx) return LVK_ERR_ARG;
    LVK_HIP(ctx, hipMemsetAsync(d_dst, value, bytes, ctx->stream));
    return LVK_OK;
}

// ---- gyro prediction: host math, float32 as the reference (image_processor.cpp:222-293) -----------
lvk_status lvk_predict_homography(const lvk_imu* imu, int n_imu, double t_prev, double t_curr, const double R_cam_imu[9],
                                  const double intr[4], float H[9])
{
    if ((n_imu > 0 && !imu) || !R_cam_imu || !intr || !H) return LVK_ERR_ARG;
    int b = 0;
    while (b < n_imu && imu[b].t - t_prev < -0.0049) ++b;
    int e = b;
    while (e < n_imu && imu[e].t - t_curr < 0.0049) ++e;
    float mw[3] = {0.f, 0.f, 0.f};
    for (int i = b; i < e; ++i) { mw[0] += (float)imu[i].gyro[0]; mw[1] += (float)imu[i].gyro[1]; mw[2] += (float)imu[i].gyro[2]; }
    if (e - b > 0) { float s = 1.0f / (e - b); mw[0] *= s; mw[1] *= s; mw[2] *= s; }
    float cw[3];
    for (int i = 0; i < 3; ++i) {
        double s = 0.;
        for (int k = 0; k < 3; ++k) s += R_cam_imu[k * 3 + i] * (double)mw[k];
        cw[i] = (float)s;
    }
    const double dtime = t_curr - t_prev;
    const float rv[3] = {(float)(cw[0] * dtime), (float)(cw[1] * dtime), (float)(cw[2] * dtime)};
    double rx = rv[0], ry = rv[1], rz = rv[2];
    const double theta = sqrt(rx * rx + ry * ry + rz * rz);
    double R[9];
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1. : 0.;
    } else {
        const double c = cos(theta), s = sin(theta), c1 = 1. - c, it = theta ? 1. / theta : 0.;
        rx *= it; ry *= it; rz *= it;
        const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
        const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
        for (int i = 0; i < 9; ++i) R[i] = c * ((i % 4 == 0) ? 1. : 0.) + c1 * rrt[i] + s * r_x[i];
    }
    float Rt[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[i * 3 + j] = (float)R[j * 3 + i];
    const float K[9] = {(float)intr[0], 0.f, (float)intr[2], 0.f, (float)intr[1], (float)intr[3], 0.f, 0.f, 1.f};
    float Ki[9];
    {
        const float* a = K;
        float d = a[0] * (a[4] * a[8] - a[7] * a[5]) - a[1] * (a[3] * a[8] - a[6] * a[5]) + a[2] * (a[3] * a[7] - a[6] * a[4]);
        if (d == 0) { for (int i = 0; i < 9; ++i) Ki[i] = 0.f; }
        else {
            d = 1 / d;
            Ki[0] = (a[4] * a[8] - a[5] * a[7]) * d; Ki[1] = (a[2] * a[7] - a[1] * a[8]) * d; Ki[2] = (a[1] * a[5] - a[2] * a[4]) * d;
            Ki[3] = (a[5] * a[6] - a[3] * a[8]) * d; Ki[4] = (a[0] * a[8] - a[2] * a[6]) * d; Ki[5] = (a[2] * a[3] - a[0] * a[5]) * d;
            Ki[6] = (a[3] * a[7] - a[4] * a[6]) * d; Ki[7] = (a[1] * a[6] - a[0] * a[7]) * d; Ki[8] = (a[0] * a[4] - a[1] * a[3]) * d;
        }
    }
    float T[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { float s = 0; for (int k = 0; k < 3; ++k) s += K[i * 3 + k] * Rt[k * 3 + j]; T[i * 3 + j] = s; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { float s = 0; for (int k = 0; k < 3; ++k) s += T[i * 3 + k] * Ki[k * 3 + j]; H[i * 3 + j] = s; }
    return LVK_OK;
}

}  // extern "C"

// =========================================================================== frame-level device state
struct FeDev {
    unsigned long long next_id;
    unsigned long long lk_point_levels, lk_iterations;
    int n_tracks[2];     // live tracks in set 0 / set 1
    int n_new;           // new_pts_ size
    int n_msg;           // features in the last message
    int boot_ok;         // result of the last bootstrap attempt
    int pad_;
    unsigned long long msg_features, msg_count;   // features in all published messages / messages published (mean tracks per message)
};

struct TrackSet {       // structure of arrays, capacity cap
    unsigned long long* id;
    lvk_pt2f* pts;       // curr_pts_ of the frame that wrote the set == prev_pts_ of the next frame
    lvk_pt2f* ppts;      // prev_pts_ of the frame that wrote the set (needed by getFeatureMsg)
    lvk_pt2f* init;
    int* life;
    unsigned long long* desc;   // 4 x u64 per track: first-seen descriptor
};

struct HMat { float h[9]; };

__device__ __forceinline__ lvk_pt2f apply_h(const HMat& H, lvk_pt2f p)
{   // predictFeatureTracking (:285-290): Matx33f * Vec3f, s = 0; s += H(i,k)*p(k)
    float q[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) { float s = 0.f; s += H.h[r * 3] * p.x; s += H.h[r * 3 + 1] * p.y; s += H.h[r * 3 + 2] * 1.0f; q[r] = s; }
    lvk_pt2f o; o.x = q[0] / q[2]; o.y = q[1] / q[2];
    return o;
}

// stage codes in the per-point status byte
#define ST_ALIVE 0
#define ST_FWD   1
#define ST_REV   2
#define ST_ORB   3

// Forward LK (prev -> curr, seeded with the gyro-predicted point, :558-590 / :830-860), reverse LK (curr -> prev, seeded with the
// original point; in-image and <= 1 px tests, :616-642) and the ORB descriptor gate (:677-699 old tracks: descriptor at the current
// point vs the stored first-seen one; :909-930 new points: descriptor in the previous image vs in the current one, the previous one is
// kept) of a point in ONE launch, TWO wavefronts per point: wavefront 0 runs the two LK passes back to back; wavefront 1 computes the
// descriptors in their shadow - the previous-image one (new points) during the forward pass, the current-image one as soon as the
// forward pass has produced the point, i.e. while the reverse pass runs.  The gate's launch (13 us + its barrier) leaves the frame's
// dependent chain altogether; a descriptor computed for a point that then fails the reverse check is simply not used.  The same
// wavefront also undistorts the point pair (K -> K, what findFundamentalMat is fed, :701-712 / :932-943) into w_und[2p], [2p+1]:
// two sequential double-precision fixed-point loops per point that the one-workgroup commit kernel would otherwise run for
// every point of its set (24 us of its 120 at 2000 tracks).
#ifdef LVK_LK_TIMING
static __device__ unsigned long long g_lk_span[4096][4];     // per block: cycles entry -> forward done, -> reverse done, -> end; 100 MHz ticks entry -> end
#endif
template <int WIN, int VAR>
__global__ void __launch_bounds__(128) k_fe_lk_both(PyrView prev, PyrView next, const lvk_pt2f* __restrict__ src_pts, const int* __restrict__ n_ptr,
                                                   HMat H, int width, int height, int max_count, double epsilon,
                                                   lvk_pt2f* __restrict__ w_curr, uint8_t* __restrict__ w_status, FeDev* __restrict__ dev,
                                                   const uint8_t* __restrict__ cur_ext, const uint8_t* __restrict__ cur_blur,
                                                   const uint8_t* __restrict__ prv_ext, const uint8_t* __restrict__ prv_blur,
                                                   const unsigned long long* __restrict__ stored_desc /*old*/, unsigned long long* __restrict__ w_desc /*new: out*/, int is_new,
                                                   CamParams cam, lvk_pt2f* __restrict__ w_und)
{
    __shared__ lvk_pt2f s_np;
    __shared__ int s_st, s_dist;
    __shared__ __attribute__((aligned(16))) unsigned long long s_acc[4];
    const int p = blockIdx.x;
#ifdef LVK_LK_TIMING
    const unsigned long long lkt_c0 = clock64(), lkt_w0 = wall_clock64();
    unsigned long long lkt_c1 = 0, lkt_c2 = 0;
#endif
    if (p >= *n_ptr) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_levels = prev.n_levels < next.n_levels ? prev.n_levels : next.n_levels;
    const int step = width + 2 * LVK_ORB_BORDER;
    const lvk_pt2f pp = src_pts[p];
    lvk_pt2f np = pp;
    int st = 1, its = 0;
    unsigned long long dp[4] = {0, 0, 0, 0};
    LkLdsAcc acc; acc.off = 0; acc.prev[0] = acc.prev[1] = acc.prev[2] = 0;
    if (wave == 0) {
        acc = lk_acc_init(s_acc);
        np = apply_h(H, pp);
        its = lk_point<WIN, VAR>(prev, next, n_levels, pp, np, st, max_count, epsilon, nullptr, acc);
        if (st && (np.y < 0 || np.y > height - 1 || np.x < 0 || np.x > width - 1)) st = 0;
        if (lane == 0) { s_np = np; s_st = st; }
#ifdef LVK_LK_TIMING
        lkt_c1 = clock64();
#endif
    } else {
        if (is_new) {
            orb_point(prv_ext, prv_blur, step, pp, dp);
            if (lane == 0) { unsigned long long* o = w_desc + (size_t)p * 4; o[0] = dp[0]; o[1] = dp[1]; o[2] = dp[2]; o[3] = dp[3]; }
        }
        if (lane == 0) w_und[2 * (size_t)p] = undistort_point(pp, cam, cam.intr);
    }
    __syncthreads();
    int code = ST_FWD, passes = 1;
    if (wave == 0) {
        code = st ? ST_ALIVE : ST_FWD;
        if (st) {
            lvk_pt2f back = pp;
            int sr = 1;
            its += lk_point<WIN, VAR, 1>(next, prev, n_levels, np, back, sr, max_count, epsilon, nullptr, acc);
            passes = 2;
            if (sr) {
                if (back.y < 0 || back.y > height - 1 || back.x < 0 || back.x > width - 1) sr = 0;
                else {
                    float dx = back.x - pp.x, dy = back.y - pp.y;
                    float dis = (float)sqrt((double)dx * dx + (double)dy * dy);      // cv::norm(Point2f) is double
                    if (dis > 1) sr = 0;
                }
            }
            if (!sr) code = ST_REV;
        }
#ifdef LVK_LK_TIMING
        lkt_c2 = clock64();
#endif
    } else if (s_st) {
        unsigned long long dc[4];
        orb_point(cur_ext, cur_blur, step, s_np, dc);
        const int dist = is_new ? hamming256_u64(dc, dp) : hamming256_u64(dc, stored_desc + (size_t)p * 4);
        if (lane == 0) { s_dist = dist; w_und[2 * (size_t)p + 1] = undistort_point(s_np, cam, cam.intr); }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (code == ST_ALIVE && s_dist > 58) code = ST_ORB;
        w_curr[p] = np; w_status[p] = (uint8_t)code;
        atomicAdd(&dev->lk_point_levels, (unsigned long long)(passes * n_levels));
        atomicAdd(&dev->lk_iterations, (unsigned long long)its);
#ifdef LVK_LK_TIMING
        if (p < 4096) { g_lk_span[p][0] = lkt_c1 - lkt_c0; g_lk_span[p][1] = lkt_c2 - lkt_c0; g_lk_span[p][2] = clock64() - lkt_c0; g_lk_span[p][3] = wall_clock64() - lkt_w0; }
#endif
    }
}

// The same stages with the track's work spread over FIVE wavefronts (LK variant 2, fe_track_dev.h): wavefront 0 owns the track and
// only iterates; wavefronts 1-3 build the levels' templates of a pass at once (level w - 1, w + 2, ...) before it starts - forward: from
// the point in the previous image, known at launch; reverse: from the forward result - and wavefront 4 is the descriptor / undistortion
// wavefront of k_fe_lk_both.  Four block barriers, every wavefront passes each of them.
// One track set of a launch: the old tracks (trackFeatures) or the new points (trackNewFeatures).  The kernel can carry BOTH in one launch
// (blocks [0, grid0) the first set, the rest the second; LVK_LK_MERGED=1): the two chains are normally two launches on two streams joined
// by an event between the commits - a barrier packet of ~7 us on every frame's dependent chain (profiles/r6_final_a_queue_gaps.txt:
// k_fe_ransac_commit -> k_fe_ransac_commit 6.6-7.3 us).  Measured: the merged launch loses more than that packet costs (lk_merged_ok).
struct LkSet {
    const lvk_pt2f* src_pts; const int* n_ptr;
    lvk_pt2f* w_curr; uint8_t* w_status;
    const unsigned long long* stored_desc;   // old tracks: the first-seen descriptors the gate compares with
    unsigned long long* w_desc;              // new points: the previous-image descriptor (out)
    lvk_pt2f* w_und;
    int is_new, pad_;
};
template <int WIN>
__global__ void __launch_bounds__(320) k_fe_lk_pipe(PyrView prev, PyrView next, LkSet set0, LkSet set1, int grid0,
                                                   HMat H, int width, int height, int max_count, double epsilon, FeDev* __restrict__ dev,
                                                   const uint8_t* __restrict__ cur_ext, const uint8_t* __restrict__ cur_blur,
                                                   const uint8_t* __restrict__ prv_ext, const uint8_t* __restrict__ prv_blur, CamParams cam)
{
    static_assert(WIN == 21, "row-segment layout");
    __shared__ lvk_pt2f s_np;
    __shared__ int s_st, s_dist;
    __shared__ __attribute__((aligned(16))) unsigned long long s_acc[4][4];
    __shared__ __attribute__((aligned(16))) LkTplLevel s_tpl[LK_PIPE_MAX_LEVELS];
    const bool second = (int)blockIdx.x >= grid0;
    const LkSet& S = second ? set1 : set0;
    const int p = (int)blockIdx.x - (second ? grid0 : 0);
    const lvk_pt2f* __restrict__ src_pts = S.src_pts; const int* __restrict__ n_ptr = S.n_ptr;
    lvk_pt2f* __restrict__ w_curr = S.w_curr; uint8_t* __restrict__ w_status = S.w_status;
    const unsigned long long* __restrict__ stored_desc = S.stored_desc; unsigned long long* __restrict__ w_desc = S.w_desc;
    lvk_pt2f* __restrict__ w_und = S.w_und; const int is_new = S.is_new;
#ifdef LVK_LK_TIMING
    const unsigned long long lkt_c0 = clock64(), lkt_w0 = wall_clock64();
    unsigned long long lkt_c1 = 0, lkt_c2 = 0;
#endif
    if (p >= *n_ptr) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_levels = prev.n_levels < next.n_levels ? prev.n_levels : next.n_levels;
    const int step = width + 2 * LVK_ORB_BORDER;
    const lvk_pt2f pp = src_pts[p];
    lvk_pt2f np = pp;
    int st = 1, its = 0;
    unsigned long long dp[4] = {0, 0, 0, 0};
    LkLdsAcc acc; acc.off = 0; acc.prev[0] = acc.prev[1] = acc.prev[2] = 0;
    if (wave < 4) acc = lk_acc_init(s_acc[wave]);
#ifdef LVK_LK_BOUNDS
    if (!(pp.x >= 0 && pp.x <= width - 1 && pp.y >= 0 && pp.y <= height - 1)) {
        if (threadIdx.x == 0 && atomicAdd(&g_lk_oob[0], 1) == 0) { g_lk_oob[1] = 5; g_lk_oob[2] = *n_ptr; g_lk_oob[3] = p; g_lk_oob[4] = is_new; g_lk_oob[7] = __builtin_bit_cast(int, pp.x); g_lk_oob[8] = __builtin_bit_cast(int, pp.y); g_lk_oob[9] = blockIdx.x; }
        return;
    }
#endif
    // ---- before the forward pass: its templates, the gyro-predicted start, the previous-image descriptor of a new point
    if (wave == 0) np = apply_h(H, pp);
    else if (wave < 4) { for (int level = wave - 1; level < n_levels; level += 3) lk_tpl_build21(prev, level, pp, s_tpl[level], acc); }
    else {
        if (is_new) {
            orb_point(prv_ext, prv_blur, step, pp, dp);
            if (lane == 0) { unsigned long long* o = w_desc + (size_t)p * 4; o[0] = dp[0]; o[1] = dp[1]; o[2] = dp[2]; o[3] = dp[3]; }
        }
        if (lane == 0) w_und[2 * (size_t)p] = undistort_point(pp, cam, cam.intr);
    }
    __syncthreads();
    if (wave == 0) {
        its = lk_pass_iterate21(next, n_levels, s_tpl, np, st, max_count, epsilon, acc);
        if (st && (np.y < 0 || np.y > height - 1 || np.x < 0 || np.x > width - 1)) st = 0;
        if (lane == 0) { s_np = np; s_st = st; }
#ifdef LVK_LK_TIMING
        lkt_c1 = clock64();
#endif
    }
    __syncthreads();
    const int fwd_ok = s_st;
    const lvk_pt2f fnp = s_np;
    // ---- before the reverse pass: its templates, from the point the forward pass found in the current image
    if (fwd_ok && wave >= 1 && wave < 4) { for (int level = wave - 1; level < n_levels; level += 3) lk_tpl_build21(next, level, fnp, s_tpl[level], acc); }
    __syncthreads();
    int code = ST_FWD, passes = 1;
    if (wave == 0) {
        code = st ? ST_ALIVE : ST_FWD;
        if (st) {
            lvk_pt2f back = pp;
            int sr = 1;
            its += lk_pass_iterate21(prev, n_levels, s_tpl, back, sr, max_count, epsilon, acc);
            passes = 2;
            if (sr) {
                if (back.y < 0 || back.y > height - 1 || back.x < 0 || back.x > width - 1) sr = 0;
                else {
                    float dx = back.x - pp.x, dy = back.y - pp.y;
                    float dis = (float)sqrt((double)dx * dx + (double)dy * dy);      // cv::norm(Point2f) is double
                    if (dis > 1) sr = 0;
                }
            }
            if (!sr) code = ST_REV;
        }
#ifdef LVK_LK_TIMING
        lkt_c2 = clock64();
#endif
    } else if (wave == 4 && fwd_ok) {
        unsigned long long dc[4];
#ifdef LVK_LK_BOUNDS
        if (!(fnp.x >= 0 && fnp.x <= width - 1 && fnp.y >= 0 && fnp.y <= height - 1)) {
            if (lane == 0 && atomicAdd(&g_lk_oob[0], 1) == 0) { g_lk_oob[1] = 6; g_lk_oob[2] = *n_ptr; g_lk_oob[3] = p; g_lk_oob[4] = is_new; g_lk_oob[7] = __builtin_bit_cast(int, fnp.x); g_lk_oob[8] = __builtin_bit_cast(int, fnp.y); g_lk_oob[9] = blockIdx.x; }
        } else
#endif
        orb_point(cur_ext, cur_blur, step, fnp, dc);
        const int dist = is_new ? hamming256_u64(dc, dp) : hamming256_u64(dc, stored_desc + (size_t)p * 4);
        if (lane == 0) { s_dist = dist; w_und[2 * (size_t)p + 1] = undistort_point(fnp, cam, cam.intr); }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (code == ST_ALIVE && s_dist > 58) code = ST_ORB;
        w_curr[p] = np; w_status[p] = (uint8_t)code;
        atomicAdd(&dev->lk_point_levels, (unsigned long long)(passes * n_levels));
        atomicAdd(&dev->lk_iterations, (unsigned long long)its);
#ifdef LVK_LK_TIMING
        if (!second && p < 4096) { g_lk_span[p][0] = lkt_c1 - lkt_c0; g_lk_span[p][1] = lkt_c2 - lkt_c0; g_lk_span[p][2] = clock64() - lkt_c0; g_lk_span[p][3] = wall_clock64() - lkt_w0; }
#endif
    }
}

// One workgroup of NT threads: count survivors per stage, order-preserving compaction of the alive points (wavefront ballots), the
// undistorted pairs the LK kernel left in w_und, cv::findFundamentalMat mask, then write the destination
// track set.  mode 0: old tracks (trackFeatures :701-808), 1: new points appended (trackNewFeatures
// :932-1001), 2: bootstrap (initializeFirstFeatures :464-536).
template <int NT>
__device__ __forceinline__ void fe_commit_block(int mode, int cap,
                                                const lvk_pt2f* src_pts, const int* n_ptr, const lvk_pt2f* w_curr, const uint8_t* w_status, const lvk_pt2f* w_und,
                                                const unsigned long long* src_id, const lvk_pt2f* src_init, const int* src_life, const unsigned long long* src_desc,
                                                const TrackSet& dst, int* dst_n, FeDev* dev,
                                                lvk_pt2f* s1, lvk_pt2f* s2, uint8_t* smask, unsigned short* sidx)
{   // s1, s2, smask, sidx: LDS scratch for FM_MAX_N points, owned by the calling kernel
    __shared__ int cnt[4];
    __shared__ int wtot[NT / 64];
    const int t = threadIdx.x;
#ifdef LVK_FM_TIMING
    if (t == 0) g_fm_on = mode == 0;
#endif
    FM_TICK(0);
    const int n = min(*n_ptr, cap);
    if (t < 4) cnt[t] = 0;
    __syncthreads();
    // survivors after each stage + ordered compaction of the alive ones (block scan over chunks)
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += NT) {
        const int i = c0 + t;
        const int st = i < n ? w_status[i] : 255;
        const bool alive = st == ST_ALIVE;
        const unsigned long long b_fwd = __ballot(i < n && st != ST_FWD);                    // after forward LK
        const unsigned long long b_rev = __ballot(i < n && st != ST_FWD && st != ST_REV);    // after reverse LK
        if ((t & 63) == 0) { atomicAdd(&cnt[0], (int)__popcll(b_fwd)); atomicAdd(&cnt[1], (int)__popcll(b_rev)); }
        int tot;
        const int rk = block_rank<NT>(alive, wtot, tot);
        if (alive) sidx[base + rk] = (unsigned short)i;
        base += tot;
    }
    __syncthreads();
    FM_TICK(1);
    const int m = base;                                                    // after the ORB gate
    bool fail = false;
    if (mode == 2) fail = cnt[0] < 20 || cnt[1] < 20 || m < 20;
    else if (mode == 1) fail = m < 20;
    int wrote = 0, iters = 0;
    if (!fail && m > 0) {
        for (int k = t; k < m; k += NT) {
            const int i = sidx[k];
            s1[k] = w_und[2 * (size_t)i];
            s2[k] = w_und[2 * (size_t)i + 1];
        }
        __syncthreads();
        FM_TICK(2);
        wrote = fm_mask_block<NT>(s1, s2, m, 1.0, 0.99, 1000, 0, smask, &iters);
        __syncthreads();
        FM_TICK(9);
    }
    // survivors of the mask (size mismatch => everything is kept, image_processor.h:219-223)
    int kept = 0;
    if (!fail) {
        int basek = 0;
        const int dst_base = (mode == 1) ? *dst_n : 0;
        for (int c0 = 0; c0 < m; c0 += NT) {
            const int k = c0 + t;
            const bool keep = k < m ? (wrote ? smask[k] != 0 : true) : false;
            int tot;
            const int rk = block_rank<NT>(keep, wtot, tot);
            if (keep) {
                const int i = sidx[k];
                const int d = dst_base + basek + rk;
                if (d < cap) {
                    dst.pts[d] = w_curr[i];
                    dst.ppts[d] = src_pts[i];
                    const unsigned long long* ds = src_desc + (size_t)i * 4;
                    unsigned long long* dd = dst.desc + (size_t)d * 4;
                    dd[0] = ds[0]; dd[1] = ds[1]; dd[2] = ds[2]; dd[3] = ds[3];
                    if (mode == 0) { dst.id[d] = src_id[i]; dst.life[d] = src_life[i] + 1; dst.init[d] = src_init[i]; }
                    else {
                        dst.id[d] = dev->next_id + (unsigned long long)(basek + rk);
                        dst.life[d] = 2;
                        if (mode == 1) dst.init[d] = src_pts[i];
                        else { dst.init[d].x = -1.f; dst.init[d].y = -1.f; }
                    }
                }
            }
            basek += tot;
        }
        kept = basek;
    }
    if (mode == 2 && kept < 20) fail = true;
    if (mode == 1 && kept <= 0) fail = true;
    __syncthreads();
    FM_TICK(10);
#ifdef LVK_FM_TIMING
    if (t == 0 && g_fm_on) { g_fm_tick[16 + 0] = m; g_fm_tick[16 + 1] = iters; g_fm_tick[16 + 2] = mode; }
#endif
    if (t == 0) {
        if (mode == 0) *dst_n = kept;
        else if (mode == 1) { if (!fail) { *dst_n = min(*dst_n + kept, cap); dev->next_id += (unsigned long long)kept; dev->n_new = 0; } }
        else {
            dev->boot_ok = fail ? 0 : 1;
            if (!fail) { *dst_n = kept; dev->next_id += (unsigned long long)kept; dev->n_new = 0; }
            else *dst_n = 0;
        }
    }
}

template <int NT>
__global__ void __launch_bounds__(NT) k_fe_ransac_commit(int mode, int cap,
                                                       const lvk_pt2f* __restrict__ src_pts, const int* __restrict__ n_ptr,
                                                       const lvk_pt2f* __restrict__ w_curr, const uint8_t* __restrict__ w_status, const lvk_pt2f* __restrict__ w_und,
                                                       const unsigned long long* __restrict__ src_id, const lvk_pt2f* __restrict__ src_init,
                                                       const int* __restrict__ src_life, const unsigned long long* __restrict__ src_desc,
                                                       TrackSet dst, int* __restrict__ dst_n, FeDev* __restrict__ dev)
{
    __shared__ lvk_pt2f s1[FM_MAX_N], s2[FM_MAX_N];
    __shared__ uint8_t smask[FM_MAX_N];
    __shared__ unsigned short sidx[FM_MAX_N];
    fe_commit_block<NT>(mode, cap, src_pts, n_ptr, w_curr, w_status, w_und, src_id, src_init, src_life, src_desc, dst, dst_n, dev, s1, s2, smask, sidx);
}

// getFeatureMsg (:1076-1128): undistort to normalised coordinates, finite-difference velocities
__device__ __forceinline__ void fe_msg_one(const TrackSet& ts, int i, const CamParams& cam, double dt_1, double dt_2, int prev_is_last, lvk_feature_obs* out)
{
    const double unit[4] = {1, 1, 0, 0};
    const lvk_pt2f uc = undistort_point(ts.pts[i], cam, unit);
    const lvk_pt2f ini = ts.init[i];
    const lvk_pt2f ui = undistort_point(ini, cam, unit);
    const lvk_pt2f up = undistort_point(ts.ppts[i], cam, unit);
    lvk_feature_obs f;
    f.id = ts.id[i];
    f.u = uc.x; f.v = uc.y;
    f.u_vel = (uc.x - up.x) / dt_1;
    f.v_vel = (uc.y - up.y) / dt_1;
    f.u_init_vel = 0.0; f.v_init_vel = 0.0;
    if (ini.x == -1 && ini.y == -1) { f.u_init = -1; f.v_init = -1; }
    else {
        f.u_init = ui.x; f.v_init = ui.y;
        ts.init[i].x = -1.f; ts.init[i].y = -1.f;
        if (prev_is_last) { f.u_init_vel = (uc.x - ui.x) / dt_2; f.v_init_vel = (uc.y - ui.y) / dt_2; }
        else { f.u_init_vel = (up.x - ui.x) / dt_2; f.v_init_vel = (up.y - ui.y) / dt_2; }
    }
    out[i] = f;
}
__global__ void k_fe_msg(TrackSet ts, const int* __restrict__ n_ptr, CamParams cam, double dt_1, double dt_2, int prev_is_last,
                         lvk_feature_obs* __restrict__ out, FeDev* __restrict__ dev, int* __restrict__ n_host)
{   // `out` and `n_host` are device-mapped pinned host memory: the message needs no copy command, only the stream sync
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = *n_ptr;
    if (i == 0) { dev->n_msg = n; *n_host = n; dev->msg_features += (unsigned long long)n; dev->msg_count += 1ull; }
    if (i >= n) return;
    fe_msg_one(ts, i, cam, dt_1, dt_2, prev_is_last, out);
}

struct FeMsgArgs { int publish, prev_is_last; double dt_1, dt_2; lvk_feature_obs* out; int* n_host; };

// =========================================================================== host object
#define LVK_MSG_SLOTS 4
struct lvk_frontend {
    lvk_context* ctx;
    lvk_fe_config cfg;
    int cap;
    int image_state;           // 1 FIRST_IMAGE, 2 SECOND_IMAGE, 3 OTHER_IMAGES
    bool b_first_img;
    long pub_counter;
    double last_pub_time, curr_img_time, prev_img_time;
    int cur;                   // track set holding prev_pts_ (written by the previous frame)
    lvk_pyramid* pyr[3];       // [0] = prev, [1] = curr, [2] = spare (rotated every frame: the image stage of frame k+1 fills its
    uint8_t *ext[3], *blur[3]; //  buffers on its own stream while frame k's tracking still reads [0] and [1])
    uint8_t* d_img;            // device copy of a host image
    // Host images (the reference's cv::Mat, pageable) are copied by the calling thread into a ring of pinned, device-mapped staging
    // slots; the GPU side is either one asynchronous H2D copy into d_img (default) or, with LVK_FE_ZEROCOPY=1, the first two image
    // kernels reading the slot in place over PCIe.  The caller's buffer is free as soon as the call returns (as in the reference).
    uint8_t* h_stage[3] = {nullptr, nullptr, nullptr};
    // Blocking calls (lvk_frontend_process without lvk_frontend_begin: the adapter's and the sequential driver's schedule) on a host with
    // a large BAR (every MI355X host; hipDeviceAttributeIsLargeBar): the calling thread copies the image straight into one of three
    // DEVICE buffers instead - the same ~10 us of CPU copy as into a pinned slot (tools/gpu/bar_probe.hip) and no copy command, whose
    // engine start-up + transfer (~20 us) stood in front of the frame's first kernel: sequential driver +5.6 %, adapter's schedule
    // +2.6 % (profiles/r4_al_image_bar_push_ab.txt).  The pipelined driver queues its image stage a frame ahead, where that latency
    // is hidden, and keeps the copy engine (the BAR copy measured -0.8 % there).  A buffer is reused three frames later, behind an
    // event recorded after the two kernels that read it (ev_img).
    uint8_t* d_ring[3] = {nullptr, nullptr, nullptr}; hipEvent_t ev_img[3] = {nullptr, nullptr, nullptr}; bool ev_img_set[3] = {false, false, false}; bool bar_push = false;

    int stage_next = 0;
    TrackSet set[2];
    lvk_pt2f *w_curr, *wn_curr, *new_pts;
    lvk_pt2f *w_und, *wn_und;  // undistorted (prev, curr) pair per point, written by the LK kernel next to w_curr / wn_curr
    uint8_t *w_status, *wn_status;
    unsigned long long* wn_desc;
    float* eig; uint8_t* mask; unsigned* gf_scratch; unsigned long long* gf_cands; int gf_cand_cap;
    // The message buffer: pinned host memory (h_msg) and its device-mapped view (d_msg), and its length, same arrangement - a ring
    // of LVK_MSG_SLOTS messages so that a pipelined driver can let the GPU run ahead: processImage returns as soon as the frame is
    // queued and the filter's thread picks the message up when ev_msg[slot] has fired (lvk_frontend_fetch_msg).
    lvk_feature_obs* d_msg; lvk_feature_obs* h_msg;
    int* h_nmsg; int* d_nmsg;
    hipEvent_t ev_msg[LVK_MSG_SLOTS] = {};
    std::atomic<int> msg_pending[LVK_MSG_SLOTS];         // 1 from the publish until the message has been fetched
    int msg_next = 0;
    FeDev* dev; FeDev* h_dev;                            // device + pinned host mirror
    CamParams cam;
    // Side stream side[0] ("new points"): goodFeaturesToTrack after a publish, then the next frame's LK / descriptor gate of those
    // points.  The main stream keeps the old tracks and both commits.  With the filter's stream that makes four: HIP maps streams
    // onto 4 hardware queues by default, and two streams sharing a queue serialise (measured), so there is no fifth.
    // side[1] ("image"): upload, CLAHE, pyramid + Scharr planes and the ORB planes of a frame.  They depend on nothing but the image,
    // so with three buffer sets they run ahead of the tracking chain of the previous frame (~55 us per frame off the main chain).
    lvk_context* side[2];
    hipEvent_t ev_main[2] = {nullptr, nullptr}, ev_side[2] = {nullptr, nullptr};   // end of frame f's work on the main / side stream, by parity of f
    hipEvent_t ev_end[2] = {nullptr, nullptr};           // what stands for "frame f's main-stream work is done": ev_main[par], or the message's own event when the frame published (nothing follows it on that stream: one record, not two)
    bool frame_early = false;                            // the current frame's image stage was queued ahead of its tracking (lvk_frontend_begin: the pipelined driver)
    int last_msg_slot = -1;                              // ring entry of the message published by the current frame (-1: none)
    long n_img = 0;                                                                // image stages queued so far
    bool image_done; double image_done_ts;       // lvk_frontend_begin already queued this frame's image stage
    bool pyr_event = true;                       // this frame's image stage recorded ev_pyr (blocking API); false: ev_orb stands for the whole stage
    // fork/join events.  Each record or wait is a barrier packet on its stream (~5 us on the frame's chain), so there are as few as
    // the data flow allows: ev_pyr / ev_orb (image stream -> main and side: pyramid / ORB planes of this frame ready), ev_new (side
    // stream -> main), ev_commit (main -> side), ev_tail (bootstrap only), ev_main / ev_side (end of a frame on either stream)
    hipEvent_t ev_pyr, ev_orb, ev_new, ev_commit, ev_tail;
    bool ev_trim = true;                                 // LVK_FE_EVENT_TRIM=0: queue every event record / stream wait as rounds 1-4 did (A/B switch)
    // sticky: a frame that failed AFTER its image stage was queued (device error, ring overrun) leaves the buffer-set rotation and the
    // end-of-frame events out of step with the frame count; the handle then refuses further frames instead of racing on its buffers
    lvk_status failed = LVK_OK; char failed_msg[200] = {0};
    // HIP-event profiling of stages
    unsigned prof_mask; unsigned prof_stride; bool prof_take;     // stride: bracket every n-th frame only (event records are barrier packets on the frame's critical chain)
    struct Pending { int stage; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> ev_free;
    double prof_ms[LVK_FE_STAGES];
    uint64_t prof_n[LVK_FE_STAGES];
};

static hipEvent_t prof_event(lvk_frontend* fe)
{
    if (!fe->ev_free.empty()) { hipEvent_t e = fe->ev_free.back(); fe->ev_free.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
}
struct ProfScope {
    lvk_frontend* fe; int stage; hipEvent_t a; bool on;
    hipStream_t st;
    ProfScope(lvk_frontend* f, int s, hipStream_t stream = nullptr) : fe(f), stage(s), on(((f->prof_mask >> s) & 1u) && f->prof_take), st(stream ? stream : f->ctx->stream)
    { if (on) { a = prof_event(fe); hipEventRecord(a, st); } }
    ~ProfScope() { if (on) { hipEvent_t b = prof_event(fe); hipEventRecord(b, st); fe->pending.push_back({stage, a, b}); } }
};
static void prof_collect(lvk_frontend* fe)
{
    if (fe->pending.empty()) return;
    hipStreamSynchronize(fe->ctx->stream);
    for (int i = 0; i < 2; ++i) if (fe->side[i]) hipStreamSynchronize(fe->side[i]->stream);
    for (auto& p : fe->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { fe->prof_ms[p.stage] += ms; fe->prof_n[p.stage] += 1; }
        fe->ev_free.push_back(p.a); fe->ev_free.push_back(p.b);
    }
    fe->pending.clear();
}

template <typename T> static bool dalloc(T** p, size_t n) { return hipMalloc((void**)p, sizeof(T) * (n ? n : 1)) == hipSuccess; }

// host-side phase tracer of the calling thread (LVK_FE_TRACE=1; printed by lvk_frontend_destroy): where the caller's time per frame goes
#include <chrono>
enum { FT_SLOT_WAIT, FT_STAGE_COPY, FT_UPLOAD, FT_IMAGE_LAUNCH, FT_OUTSIDE, FT_FRAME_WAITS, FT_PREDICT, FT_TRACK_LAUNCH, FT_COMMIT_LAUNCH, FT_PUBLISH, FT_DETECT, FT_END, FT_N };
static const char* const FT_NAMES[FT_N] = {"staging slot free (end-of-frame event of f-2)", "image -> pinned staging slot (memcpy)", "H2D copy command", "image stage: event queries + 5 launches",
    "OUTSIDE the front-end: between lvk_frontend_begin and lvk_frontend_process of the frame (the pipelined caller's own submit logic and its wait for the erase count)",
    "frame start: query / wait for the pyramid event (ev_pyr or ev_orb) + the side stream's wait for the previous frame's end event", "frame checks + predict_homography (host math)",
    "track chains: 2 launches + events", "commits: 2 launches + events", "message: slot + launch + event", "detection: 4 launches", "end-of-frame events + rotation"};
struct FeTrace {
    bool on = false; double acc[FT_N] = {0}; long n = 0; std::chrono::steady_clock::time_point last;
    void start() { if (on) last = std::chrono::steady_clock::now(); }
    void mark(int k) { if (!on) return; auto t = std::chrono::steady_clock::now(); acc[k] += std::chrono::duration<double, std::micro>(t - last).count(); last = t; }
};
static FeTrace g_ft;
#define FT(k) g_ft.mark(k)

static void set_free(TrackSet& s)
{
    if (s.id) hipFree(s.id); if (s.pts) hipFree(s.pts); if (s.ppts) hipFree(s.ppts); if (s.init) hipFree(s.init);
    if (s.life) hipFree(s.life); if (s.desc) hipFree(s.desc);
    memset(&s, 0, sizeof s);
}

template <int WIN>
static void launch_track_chain(lvk_frontend* fe, hipStream_t s, const PyrView& pv, const PyrView& cv, const lvk_pt2f* src_pts, const int* n_ptr, int grid,
                               const HMat& H, lvk_pt2f* w_curr, uint8_t* w_status, const unsigned long long* stored_desc,
                               unsigned long long* w_desc, int is_new, int max_count, double epsilon)
{
    const int W = fe->cfg.width, Hh = fe->cfg.height;
    if (fe->pyr_event) hipStreamWaitEvent(s, fe->ev_orb, 0);      // the frame start waited for the pyramid only: the ORB planes (read by the kernel's second wavefront) follow it on the image stream
    ProfScope ps(fe, 2, s);
#define LVK_LK_LAUNCH(V) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fe_lk_both<WIN, V>), dim3(grid), dim3(128), 0, s, pv, cv, src_pts, n_ptr, H, W, Hh, max_count, epsilon, w_curr, w_status, fe->dev, \
                           (const uint8_t*)fe->ext[1], (const uint8_t*)fe->blur[1], (const uint8_t*)fe->ext[0], (const uint8_t*)fe->blur[0], stored_desc, w_desc, is_new, \
                           fe->cam, w_curr == fe->w_curr ? fe->w_und : fe->wn_und)
    int var = WIN == 21 ? lvk_lk_variant() : 1;
    if (var < 0) var = grid <= LVK_LK_PIPE_MAX_TRACKS ? 2 : 1;
    if constexpr (WIN == 21) {
        if (var == 2 && pv.n_levels <= LK_PIPE_MAX_LEVELS && cv.n_levels <= LK_PIPE_MAX_LEVELS) {
            LkSet a; a.src_pts = src_pts; a.n_ptr = n_ptr; a.w_curr = w_curr; a.w_status = w_status; a.stored_desc = stored_desc; a.w_desc = w_desc;
            a.w_und = w_curr == fe->w_curr ? fe->w_und : fe->wn_und; a.is_new = is_new; a.pad_ = 0;
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fe_lk_pipe<21>), dim3(grid), dim3(320), 0, s, pv, cv, a, a, grid, H, W, Hh, max_count, epsilon, fe->dev,
                               (const uint8_t*)fe->ext[1], (const uint8_t*)fe->blur[1], (const uint8_t*)fe->ext[0], (const uint8_t*)fe->blur[0], fe->cam);
            return;
        }
    }
    if (var == 0) LVK_LK_LAUNCH(0);
    else LVK_LK_LAUNCH(1);
#undef LVK_LK_LAUNCH
}

static lvk_status track_chain(lvk_frontend* fe, hipStream_t stream, const lvk_pt2f* src_pts, const int* n_ptr, const HMat& H, lvk_pt2f* w_curr, uint8_t* w_status,
                              const unsigned long long* stored_desc, unsigned long long* w_desc, int is_new)
{
    int max_count = fe->cfg.max_iteration < 0 ? 0 : fe->cfg.max_iteration > 100 ? 100 : fe->cfg.max_iteration;
    double epsilon = fe->cfg.track_precision < 0. ? 0. : fe->cfg.track_precision > 10. ? 10. : fe->cfg.track_precision;
    epsilon *= epsilon;
    PyrView pv = make_view(fe->pyr[0]), cv = make_view(fe->pyr[1]);
    switch (fe->cfg.patch_size) {
        case 21: launch_track_chain<21>(fe, stream, pv, cv, src_pts, n_ptr, fe->cap, H, w_curr, w_status, stored_desc, w_desc, is_new, max_count, epsilon); break;
        case 15: launch_track_chain<15>(fe, stream, pv, cv, src_pts, n_ptr, fe->cap, H, w_curr, w_status, stored_desc, w_desc, is_new, max_count, epsilon); break;
        case 31: launch_track_chain<31>(fe, stream, pv, cv, src_pts, n_ptr, fe->cap, H, w_curr, w_status, stored_desc, w_desc, is_new, max_count, epsilon); break;
        default: return lvk_set_error(fe->ctx, LVK_ERR_UNSUPPORTED, "patch_size %d not instantiated (15, 21, 31)", fe->cfg.patch_size);
    }
    LVK_LAUNCH_CHECK(fe->ctx);
    return LVK_OK;
}

// can the frame's two track chains ride in ONE launch?  (the five-wavefront kernel, patch 21, the library's own choice of variant or 2)
static bool lk_merged_ok(const lvk_frontend* fe)
{
    const int var = lvk_lk_variant();
    if (fe->cfg.patch_size != 21 || !(var == 2 || (var < 0 && fe->cap <= LVK_LK_PIPE_MAX_TRACKS))) return false;
    // opt-in (LVK_LK_MERGED=1): built and measured in round 6 - bit-exact, and SLOWER in the pipelined driver (10,260 against 10,640 frames/s,
    // LK launch 28.2 against 25.9 us: the launch then waits for the previous frame's detection on the side stream, and carries twice the
    // blocks), +0.5 % through the adapter's schedule; profiles/r6_o_lk_merged_launch_ab.json
    static const bool on = [] { const char* e = getenv("LVK_LK_MERGED"); return e && !strcmp(e, "1"); }();
    return on && fe->pyr[0]->n_levels <= LK_PIPE_MAX_LEVELS && fe->pyr[1]->n_levels <= LK_PIPE_MAX_LEVELS;
}
// trackFeatures' and trackNewFeatures' LK + descriptor gate in one launch on `stream`: blocks [0, cap) the old tracks of set `src`, [cap, 2 cap) the new points
static lvk_status track_chain_merged(lvk_frontend* fe, hipStream_t stream, int src, const HMat& H)
{
    int max_count = fe->cfg.max_iteration < 0 ? 0 : fe->cfg.max_iteration > 100 ? 100 : fe->cfg.max_iteration;
    double epsilon = fe->cfg.track_precision < 0. ? 0. : fe->cfg.track_precision > 10. ? 10. : fe->cfg.track_precision;
    epsilon *= epsilon;
    PyrView pv = make_view(fe->pyr[0]), cv = make_view(fe->pyr[1]);
    LkSet a, b;
    a.src_pts = fe->set[src].pts; a.n_ptr = &fe->dev->n_tracks[src]; a.w_curr = fe->w_curr; a.w_status = fe->w_status; a.stored_desc = fe->set[src].desc; a.w_desc = nullptr;
    a.w_und = fe->w_und; a.is_new = 0; a.pad_ = 0;
    b.src_pts = fe->new_pts; b.n_ptr = &fe->dev->n_new; b.w_curr = fe->wn_curr; b.w_status = fe->wn_status; b.stored_desc = nullptr; b.w_desc = fe->wn_desc;
    b.w_und = fe->wn_und; b.is_new = 1; b.pad_ = 0;
    if (fe->pyr_event) hipStreamWaitEvent(stream, fe->ev_orb, 0);
    {
        ProfScope ps(fe, 2, stream);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fe_lk_pipe<21>), dim3(2 * fe->cap), dim3(320), 0, stream, pv, cv, a, b, fe->cap, H, fe->cfg.width, fe->cfg.height, max_count, epsilon, fe->dev,
                           (const uint8_t*)fe->ext[1], (const uint8_t*)fe->blur[1], (const uint8_t*)fe->ext[0], (const uint8_t*)fe->blur[0], fe->cam);
    }
    LVK_LAUNCH_CHECK(fe->ctx);
    return LVK_OK;
}

static lvk_status commit(lvk_frontend* fe, int mode, const lvk_pt2f* src_pts, const int* n_ptr, const lvk_pt2f* w_curr, const uint8_t* w_status,
                         const TrackSet* src, const unsigned long long* desc_src, int dst_set)
{
    ProfScope ps(fe, 5);
    const lvk_pt2f* w_und = w_curr == fe->w_curr ? fe->w_und : fe->wn_und;
    // one workgroup either way: 1024 threads when the per-point loops (compaction, scoring, mask) are what the call costs
    if (fe->cap > FM_WIDE_FROM)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fe_ransac_commit<FM_THREADS_WIDE>), dim3(1), dim3(FM_THREADS_WIDE), 0, fe->ctx->stream, mode, fe->cap, src_pts, n_ptr, w_curr, w_status, w_und,
                           (const unsigned long long*)(src ? src->id : nullptr), (const lvk_pt2f*)(src ? src->init : nullptr),
                           (const int*)(src ? src->life : nullptr), desc_src, fe->set[dst_set], &fe->dev->n_tracks[dst_set], fe->dev);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fe_ransac_commit<FM_THREADS>), dim3(1), dim3(FM_THREADS), 0, fe->ctx->stream, mode, fe->cap, src_pts, n_ptr, w_curr, w_status, w_und,
                           (const unsigned long long*)(src ? src->id : nullptr), (const lvk_pt2f*)(src ? src->init : nullptr),
                           (const int*)(src ? src->life : nullptr), desc_src, fe->set[dst_set], &fe->dev->n_tracks[dst_set], fe->dev);
    LVK_LAUNCH_CHECK(fe->ctx);
    return LVK_OK;
}

lvk_context* lvk_frontend_context(lvk_frontend* fe) { return fe ? fe->ctx : nullptr; }

#ifdef LVK_LK_TIMING
extern "C" void lvk_debug_lk_ticks(unsigned long long* out, unsigned long long* span) { hipDeviceSynchronize(); hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lk_tick), sizeof(unsigned long long) * 4096 * 12); hipMemcpyFromSymbol(span, HIP_SYMBOL(g_lk_span), sizeof(unsigned long long) * 4096 * 4); }
#endif
#ifdef LVK_FM_TIMING
extern "C" void lvk_debug_fm_ticks(unsigned long long* out) { hipDeviceSynchronize(); hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fm_tick), sizeof(unsigned long long) * 32); }
#endif

extern "C" {

void lvk_frontend_destroy(lvk_frontend* fe)
{
    if (!fe) return;
    if (g_ft.on && g_ft.n > 0) {
        double tot = 0; for (int i = 0; i < FT_N; ++i) tot += g_ft.acc[i];
        fprintf(stderr, "[lvk_frontend trace] %ld frames, %.1f us/frame of caller time inside the front-end\n", g_ft.n, tot / g_ft.n);
        for (int i = 0; i < FT_N; ++i) fprintf(stderr, "  %-44s %8.1f us\n", FT_NAMES[i], g_ft.acc[i] / g_ft.n);
        g_ft = FeTrace();
    }
    prof_collect(fe);
    hipStreamSynchronize(fe->ctx->stream);
#ifdef LVK_LK_BOUNDS
    {
        int o[16] = {0}; hipDeviceSynchronize(); hipMemcpyFromSymbol(o, HIP_SYMBOL(g_lk_oob), sizeof o);
        float fx, fy; memcpy(&fx, &o[7], 4); memcpy(&fy, &o[8], 4);
        fprintf(stderr, "[lvk LK bounds] %d window loads outside their plane; first: site %d level %d of %d, row %d col %d of %d x %d, position (%g, %g), block %d thread %d\n",
                o[0], o[1], o[2], o[11], o[3], o[4], o[5], o[6], fx, fy, o[9], o[10]);
    }
#endif
    for (int i = 0; i < 2; ++i) if (fe->side[i]) { hipStreamSynchronize(fe->side[i]->stream); lvk_context_destroy(fe->side[i]); }
    hipEvent_t evs[] = {fe->ev_pyr, fe->ev_orb, fe->ev_new, fe->ev_commit, fe->ev_tail, fe->ev_main[0], fe->ev_main[1], fe->ev_side[0], fe->ev_side[1]};
    for (hipEvent_t e : evs) if (e) hipEventDestroy(e);
    for (hipEvent_t e : fe->ev_free) hipEventDestroy(e);
    for (int i = 0; i < 3; ++i) {
        if (fe->pyr[i]) lvk_pyramid_destroy(fe->pyr[i]);
        if (fe->ext[i]) hipFree(fe->ext[i]); if (fe->blur[i]) hipFree(fe->blur[i]);
        if (i < 2) set_free(fe->set[i]);
    }
    void* ptrs[] = {fe->d_img, fe->w_curr, fe->wn_curr, fe->w_und, fe->wn_und, fe->new_pts, fe->w_status, fe->wn_status, fe->wn_desc, fe->eig, fe->mask,
                    fe->gf_scratch, fe->gf_cands, fe->dev};
    for (void* p : ptrs) if (p) hipFree(p);
    for (int i = 0; i < 3; ++i) { if (fe->h_stage[i]) hipHostFree(fe->h_stage[i]); if (fe->d_ring[i]) hipFree(fe->d_ring[i]); if (fe->ev_img[i]) hipEventDestroy(fe->ev_img[i]); }
    for (int i = 0; i < LVK_MSG_SLOTS; ++i) if (fe->ev_msg[i]) hipEventDestroy(fe->ev_msg[i]);
    if (fe->h_msg) hipHostFree(fe->h_msg);
    if (fe->h_nmsg) hipHostFree(fe->h_nmsg);
    if (fe->h_dev) hipHostFree(fe->h_dev);
    delete fe;
}

lvk_status lvk_frontend_create(lvk_context* ctx, const lvk_fe_config* cfg, lvk_frontend** out)
{
    if (!ctx || !cfg || !out) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_frontend_create: bad argument");
    g_ft.on = getenv("LVK_FE_TRACE") != nullptr;
    if (cfg->width < 64 || cfg->height < 64) return lvk_set_error(ctx, LVK_ERR_ARG, "image too small");
    if (cfg->max_features_num <= 0 || cfg->max_features_num > FM_MAX_N) return lvk_set_error(ctx, LVK_ERR_UNSUPPORTED, "max_features_num must be in 1..%d", FM_MAX_N);
    if (cfg->distortion_model != 0 && cfg->distortion_model != 1) return lvk_set_error(ctx, LVK_ERR_ARG, "distortion_model must be 0 (radtan) or 1 (equidistant)");
    if (cfg->min_distance < 1 || cfg->pub_frequency <= 0) return lvk_set_error(ctx, LVK_ERR_ARG, "min_distance >= 1 and pub_frequency > 0 required");
    if (cfg->patch_size != 15 && cfg->patch_size != 21 && cfg->patch_size != 31)
        return lvk_set_error(ctx, LVK_ERR_UNSUPPORTED, "patch_size %d not instantiated (15, 21, 31)", cfg->patch_size);
    if (cfg->max_iteration < 1 || cfg->max_iteration > 100)      // cv::calcOpticalFlowPyrLK clamps the count to 0..100; 0 iterations is not a tracker
        return lvk_set_error(ctx, LVK_ERR_UNSUPPORTED, "max_iteration must be in 1..100");
    if (cfg->pyramid_levels < 0) return lvk_set_error(ctx, LVK_ERR_ARG, "pyramid_levels must be >= 0");
    {   // more levels than buffers exist only matter if the image is large enough for OpenCV's stop rule to build them
        const int small = cfg->width < cfg->height ? cfg->width : cfg->height;
        if (cfg->pyramid_levels >= LVK_MAX_LEVELS && (small >> (LVK_MAX_LEVELS - 1)) > cfg->patch_size)
            return lvk_set_error(ctx, LVK_ERR_UNSUPPORTED, "pyramid_levels %d: at most %d levels are buffered", cfg->pyramid_levels, LVK_MAX_LEVELS - 1);
    }
    lvk_frontend* fe = new (std::nothrow) lvk_frontend();     // value-initialised: all PODs zero
    if (!fe) return LVK_ERR_DEVICE;
    fe->ctx = ctx; fe->cfg = *cfg; fe->cap = cfg->max_features_num; fe->image_state = 1;
    { const char* v = getenv("LVK_FE_EVENT_TRIM"); fe->ev_trim = !(v && !strcmp(v, "0")); }
    const int w = cfg->width, h = cfg->height, cap = fe->cap;
    const size_t esz = (size_t)(w + 64) * (h + 64);
    bool ok = true;
    for (int i = 0; i < 3 && ok; ++i) {
        ok = ok && lvk_pyramid_create(ctx, w, h, cfg->patch_size, cfg->pyramid_levels, &fe->pyr[i]) == LVK_OK;
        ok = ok && dalloc(&fe->ext[i], esz) && dalloc(&fe->blur[i], esz);
        if (i == 2) break;
        TrackSet& s = fe->set[i];
        ok = ok && dalloc(&s.id, cap) && dalloc(&s.pts, cap) && dalloc(&s.ppts, cap) && dalloc(&s.init, cap) && dalloc(&s.life, cap) && dalloc(&s.desc, (size_t)cap * 4);
    }
    fe->gf_cand_cap = w * h;
    size_t cand_alloc = 1; while (cand_alloc < (size_t)fe->gf_cand_cap) cand_alloc <<= 1;
    ok = ok && dalloc(&fe->d_img, (size_t)w * h) && dalloc(&fe->w_curr, cap) && dalloc(&fe->wn_curr, cap) && dalloc(&fe->w_und, 2 * (size_t)cap) && dalloc(&fe->wn_und, 2 * (size_t)cap) && dalloc(&fe->new_pts, cap) &&
         dalloc(&fe->w_status, cap) && dalloc(&fe->wn_status, cap) && dalloc(&fe->wn_desc, (size_t)cap * 4) && dalloc(&fe->eig, (size_t)w * h) &&
         dalloc(&fe->mask, (size_t)w * h) && dalloc(&fe->gf_scratch, 4 + 8192) && dalloc(&fe->gf_cands, cand_alloc) && dalloc(&fe->dev, 1);
    ok = ok && hipHostMalloc((void**)&fe->h_msg, sizeof(lvk_feature_obs) * (size_t)cap * LVK_MSG_SLOTS) == hipSuccess &&
         hipHostMalloc((void**)&fe->h_nmsg, 64 * LVK_MSG_SLOTS) == hipSuccess && hipHostMalloc((void**)&fe->h_dev, sizeof(FeDev)) == hipSuccess;
    for (int i = 0; i < LVK_MSG_SLOTS && ok; ++i) { fe->msg_pending[i].store(0); ok = hipEventCreateWithFlags(&fe->ev_msg[i], hipEventDisableTiming) == hipSuccess; }
    if (ok) {
        void *dm = nullptr, *dn = nullptr;
        ok = hipHostGetDevicePointer(&dm, fe->h_msg, 0) == hipSuccess && hipHostGetDevicePointer(&dn, fe->h_nmsg, 0) == hipSuccess && dm && dn;
        fe->d_msg = (lvk_feature_obs*)dm; fe->d_nmsg = (int*)dn;
    }
    fe->bar_push = true;
    for (int i = 0; i < 3 && fe->bar_push; ++i) {
        fe->d_ring[i] = (uint8_t*)lvk_bar_alloc(ctx->device, (size_t)w * h);
        if (!fe->d_ring[i] || hipEventCreateWithFlags(&fe->ev_img[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); fe->bar_push = false; }
    }
    for (int i = 0; i < 3 && ok; ++i) ok = hipHostMalloc((void**)&fe->h_stage[i], (size_t)w * h) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i) ok = lvk_context_create(ctx->device, &fe->side[i]) == LVK_OK;
    hipEvent_t* evs[] = {&fe->ev_pyr, &fe->ev_orb, &fe->ev_new, &fe->ev_commit, &fe->ev_tail, &fe->ev_main[0], &fe->ev_main[1], &fe->ev_side[0], &fe->ev_side[1]};
    for (hipEvent_t* e : evs) ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    if (!ok) { lvk_frontend_destroy(fe); return lvk_set_error(ctx, LVK_ERR_DEVICE, "lvk_frontend_create: allocation failed"); }
    hipMemsetAsync(fe->dev, 0, sizeof(FeDev), ctx->stream);
    hipMemsetAsync(fe->gf_scratch, 0, sizeof(unsigned) * (4 + 8192), ctx->stream);      // the selection kernel keeps its words zeroed from here on (fe_detect_new queues no fill)
    memset(&fe->cam, 0, sizeof fe->cam);
    for (int i = 0; i < 4; ++i) { fe->cam.intr[i] = cfg->intrinsics[i]; fe->cam.dist[i] = cfg->distortion[i]; }
    fe->cam.model = cfg->distortion_model; fe->cam.width = w; fe->cam.height = h;
    hipStreamSynchronize(ctx->stream);                  // the clears above are done before any of the three streams is used
    *out = fe;
    return LVK_OK;
}

static lvk_status fe_quiesce(lvk_frontend* fe)
{   // the getters look at buffers the side streams may still be filling
    for (int i = 0; i < 2; ++i) LVK_HIP(fe->ctx, hipStreamSynchronize(fe->side[i]->stream));
    return LVK_OK;
}
static lvk_status fe_read_dev(lvk_frontend* fe)
{
    LVK_HIP(fe->ctx, hipMemcpyAsync(fe->h_dev, fe->dev, sizeof(FeDev), hipMemcpyDeviceToHost, fe->ctx->stream));
    LVK_HIP(fe->ctx, hipStreamSynchronize(fe->ctx->stream));
    return LVK_OK;
}

// findNewFeaturesToBeTracked (:1005-1037).  Nothing in this frame's message depends on it, so it runs on side[0] behind the
// commit and overlaps the message read-back, the filter update and the next frame's pyramid; the next frame's LK of the new
// points is queued on the same stream, right behind it.
extern "C" lvk_status lvk_frontend_fetch_msg(lvk_frontend* fe, int slot, lvk_feature_obs* h_out, int cap, int* n_out);
static lvk_status fe_detect_new(lvk_frontend* fe, int dst)
{
    lvk_context* cx = fe->side[0];
    const lvk_fe_config& c = fe->cfg;
    lvk_status st;
    // no fill launches: the mask is written whole by k_mask_max, the previous selection left the scratch words zeroed
    { ProfScope ps(fe, 6, cx->stream); st = lvk_min_eigen_map(cx, fe->pyr[1], fe->eig); }
    if (st != LVK_OK) return lvk_set_error(fe->ctx, st, "%s", cx->err);
    // the frame's tracks are final behind its second commit; with the event trims the detection waits for the message's own event instead
    // (recorded right behind the message kernel, one launch later): one record less between the commit and the message the caller waits for
    // (blocking schedules only: there the caller waits for the message; in the pipelined driver the detection is on the frame rate's recurrence
    //  and starts behind the commit, without the message kernel in front of it)
    hipStreamWaitEvent(cx->stream, (fe->ev_trim && !fe->frame_early && fe->last_msg_slot >= 0) ? fe->ev_msg[fe->last_msg_slot] : fe->ev_commit, 0);
    ProfScope ps(fe, 7, cx->stream);
    st = lvk_mask_and_max(cx, fe->set[dst].pts, &fe->dev->n_tracks[dst], c.width, c.height, c.min_distance, fe->eig, fe->mask, fe->gf_scratch);
    if (st == LVK_OK) st = lvk_gftt_run(cx, fe->eig, fe->mask, c.width, c.height, c.max_features_num, 0.01, (double)c.min_distance, fe->gf_scratch,
                                        fe->gf_cands, fe->gf_cand_cap, fe->new_pts, fe->cap, &fe->dev->n_new, &fe->dev->n_tracks[dst], true, true);
    if (st != LVK_OK) return lvk_set_error(fe->ctx, st, "%s", cx->err);
    return LVK_OK;
}
// getFeatureMsg (:1076-1128) + publish bookkeeping (:1170-1172).  async_slot == nullptr: wait for the message and copy it out (the
// reference's synchronous processImage); otherwise return the ring slot at once - lvk_frontend_fetch_msg collects it later.
// ring entry for this frame's message (waits only if the consumer is four publishes behind)
static lvk_status fe_publish_slot(lvk_frontend* fe, int* slot_out)
{
    lvk_context* ctx = fe->ctx;
    const int slot = fe->msg_next; fe->msg_next = (slot + 1) % LVK_MSG_SLOTS;
    if (fe->msg_pending[slot].load(std::memory_order_acquire)) {       // four publishes ago and still not fetched: the consumer is far behind
        LVK_HIP(ctx, hipEventSynchronize(fe->ev_msg[slot]));
        for (int spin = 0; fe->msg_pending[slot].load(std::memory_order_acquire); ++spin) { if (spin > 2000000) return lvk_set_error(ctx, LVK_ERR_CAPACITY, "feature-message ring overrun"); LVK_CPU_RELAX(); }
    }
    *slot_out = slot;
    return LVK_OK;
}
static FeMsgArgs fe_msg_args(lvk_frontend* fe, int slot)
{
    FeMsgArgs m;
    m.publish = 1;
    m.dt_1 = fe->curr_img_time - fe->prev_img_time;
    m.prev_is_last = fe->prev_img_time == fe->last_pub_time;
    m.dt_2 = m.prev_is_last ? m.dt_1 : fe->prev_img_time - fe->last_pub_time;
    m.out = fe->d_msg + (size_t)slot * fe->cap; m.n_host = fe->d_nmsg + 16 * slot;
    return m;
}
// after the kernel that writes the message has been queued on the main stream: event, detection of new corners, bookkeeping
// (:1170-1172); async_slot == nullptr: wait for the message and copy it out (the reference's synchronous processImage)
static lvk_status fe_publish_finish(lvk_frontend* fe, int dst, double ts, int slot, lvk_feature_obs* h_out, int cap, int* n_out, int* async_slot)
{
    hipEventRecord(fe->ev_msg[slot], fe->ctx->stream);
    fe->last_msg_slot = slot;
    fe->msg_pending[slot].store(1, std::memory_order_release);
    FT(FT_PUBLISH);
    lvk_status st = fe_detect_new(fe, dst);              // queued on side[0] before anybody blocks on the message
    if (st != LVK_OK) return st;
    FT(FT_DETECT);
    fe->last_pub_time = ts; fe->pub_counter++;
    if (async_slot) { *async_slot = slot; *n_out = 0; return LVK_OK; }
    return lvk_frontend_fetch_msg(fe, slot, h_out, cap, n_out);
}
// getFeatureMsg (:1076-1128) + publish bookkeeping (:1170-1172)
static lvk_status fe_publish(lvk_frontend* fe, int dst, double ts, lvk_feature_obs* h_out, int cap, int* n_out, int* async_slot)
{
    lvk_context* ctx = fe->ctx;
    int slot = 0;
    lvk_status st = fe_publish_slot(fe, &slot);
    if (st != LVK_OK) return st;
    { ProfScope ps8(fe, 8);
    const FeMsgArgs m = fe_msg_args(fe, slot);
    hipLaunchKernelGGL(k_fe_msg, dim3((fe->cap + 63) / 64), dim3(64), 0, ctx->stream, fe->set[dst], (const int*)&fe->dev->n_tracks[dst], fe->cam, m.dt_1, m.dt_2,
                       m.prev_is_last, m.out, fe->dev, m.n_host);
    LVK_LAUNCH_CHECK(ctx); }
    return fe_publish_finish(fe, dst, ts, slot, h_out, cap, n_out, async_slot);
}

// The IMU-independent part of a frame: image upload, createImagePyramids (:318-334) on the main stream, ORBdescriptor ctor (:150)
// on the side stream as soon as level 0 exists (steady state) or in line (bootstrap frames).  Split out so that a pipelined
// driver can queue it before it knows which IMU samples the previous update erased (lvk_frontend_begin).
// A stream wait is a barrier packet on the waiting queue, ~6 us of queue time between two kernels even when the event fired long ago
// (profiles/r5_d_a_queue_gaps.txt: scharr -> blur 8.3 us with one event record between them against 1.5 us without; the last kernel of
// a frame -> the next frame's LK 26 us with three).  An event that HAS fired needs no packet: everything recorded before it is done.
static inline void fe_wait_unless_done(const lvk_frontend* fe, hipStream_t s, hipEvent_t ev)
{
    if (fe->ev_trim && hipEventQuery(ev) == hipSuccess) return;
    (void)hipGetLastError();                             // hipErrorNotReady is not an error
    hipStreamWaitEvent(s, ev, 0);
}
static lvk_status fe_check_image(lvk_frontend* fe, const lvk_image* img)
{   // a cv::Mat carries its own size; a caller that hands over anything but the configured resolution gets an error, not a read
    // past its buffer (e.g. TUM-VI 512x512 images under a 752x480 configuration)
    const lvk_fe_config& c = fe->cfg;
    if (!img || !img->data) return lvk_set_error(fe->ctx, LVK_ERR_ARG, "image: null pointer");
    if (img->width != c.width || img->height != c.height)
        return lvk_set_error(fe->ctx, LVK_ERR_ARG, "image is %dx%d, the front-end is configured for %dx%d", img->width, img->height, c.width, c.height);
    if (img->stride < c.width) return lvk_set_error(fe->ctx, LVK_ERR_ARG, "image stride %d is smaller than the width %d", img->stride, c.width);
    return LVK_OK;
}

static lvk_status fe_image_stage(lvk_frontend* fe, const lvk_image* image, bool early)
{
    lvk_context* ctx = fe->ctx;
    const lvk_fe_config& c = fe->cfg;
    lvk_status stc = fe_check_image(fe, image);
    if (stc != LVK_OK) return stc;
    const uint8_t* d_img = image->data; int d_stride = image->stride;
    int slot = -1;
    g_ft.start();
    if (!image->is_device && fe->bar_push && !early) {
        slot = fe->stage_next; fe->stage_next = (slot + 1) % 3;
        if (fe->ev_img_set[slot]) LVK_HIP(ctx, hipEventSynchronize(fe->ev_img[slot]));      // the kernels that read this buffer three frames ago
        if (fe->n_img >= 3) LVK_HIP(ctx, hipEventSynchronize(fe->ev_end[fe->n_img & 1]));   // (the same bound on frames in flight as the pinned-slot path)
        FT(FT_SLOT_WAIT);
        uint8_t* dd = fe->d_ring[slot];
        if (image->stride == c.width) memcpy(dd, image->data, (size_t)c.width * c.height);
        else for (int y = 0; y < c.height; ++y) memcpy(dd + (size_t)y * c.width, image->data + (size_t)y * image->stride, (size_t)c.width);
        LVK_STORE_FENCE();
        FT(FT_STAGE_COPY);
        d_img = dd; d_stride = c.width;
        FT(FT_UPLOAD);
    } else if (!image->is_device) {
        slot = fe->stage_next; fe->stage_next = (slot + 1) % 3;
        // The slot was last used three frames back.  Its upload precedes frame f-2's image stage on the image stream, which the
        // main stream waited for before it recorded the end of frame f-2: that event covers it (and has long fired) - no per-slot event.
        if (fe->n_img >= 3) LVK_HIP(ctx, hipEventSynchronize(fe->ev_end[fe->n_img & 1]));
        FT(FT_SLOT_WAIT);
        uint8_t* hs = fe->h_stage[slot];
        if (image->stride == c.width) memcpy(hs, image->data, (size_t)c.width * c.height);
        else for (int y = 0; y < c.height; ++y) memcpy(hs + (size_t)y * c.width, image->data + (size_t)y * image->stride, (size_t)c.width);
        FT(FT_STAGE_COPY);
        LVK_HIP(ctx, hipMemcpyAsync(fe->d_img, hs, (size_t)c.width * c.height, hipMemcpyHostToDevice, fe->side[1]->stream)); d_img = fe->d_img;
        d_stride = c.width;
        FT(FT_UPLOAD);
    }
    lvk_status st;
    lvk_context* icx = fe->side[1];
    hipStream_t S0 = icx->stream;
    // The buffer set about to be written (frame f) was the spare set of frame f-1 and the "prev" set of frame f-2: its last readers are
    // frame f-2's tracking chains and the detection queued behind frame f-3 (same side stream, earlier in order).  Waiting for the
    // ends of frame f-2's two chains is therefore enough, and frame f-1's tracking runs concurrently with this image stage.
    if (fe->n_img >= 2) {
        const int par = (int)(fe->n_img & 1);            // parity of f-2
        fe_wait_unless_done(fe, S0, fe->ev_end[par]); fe_wait_unless_done(fe, S0, fe->ev_side[par]);
    }
    int mosaic_done = 0;
    {
        ProfScope ps(fe, 0, S0);
        st = lvk_pyramid_build_with_orb(icx, fe->pyr[1], d_img, d_stride, c.flag_equalize, 3.0, 8, 8, fe->ext[1], &mosaic_done);
    }
    if (st != LVK_OK) return lvk_set_error(ctx, st, "%s", icx->err);
    // (the slot's readers - the two kernels above - are covered by the end-of-frame event of frame f-2 the caller waits for anyway, as in
    //  the pinned-slot path: no event of their own)
    if (slot >= 0 && fe->bar_push && !fe->ev_trim) { hipEventRecord(fe->ev_img[slot], S0); fe->ev_img_set[slot] = true; }
    // queued ahead of the frame's tracking (pipelined driver): the ORB planes are done long before anybody asks, one event (ev_orb)
    // stands for the whole stage; queued together with the tracking (blocking API): LK may start as soon as the pyramid exists
    // (rounds 1-4 recorded ev_pyr here in the blocking schedules so that the frame could start on the pyramid alone; but the first thing a
    //  steady-state frame queues is LK, whose second wavefront reads the ORB planes and waits for ev_orb anyway: the split only put a
    //  record between the last Scharr launch and the blur, and a second wait on both tracking streams)
    fe->frame_early = early;
    const bool split = !early && !fe->ev_trim;
    if (split) hipEventRecord(fe->ev_pyr, S0);
    fe->pyr_event = split;
    { ProfScope ps(fe, 1, S0); st = mosaic_done ? lvk_orb_blur_only(icx, fe->pyr[1], fe->ext[1], fe->blur[1]) : lvk_orb_prepare(icx, fe->pyr[1], fe->ext[1], fe->blur[1]); }
    if (st != LVK_OK) return lvk_set_error(ctx, st, "%s", icx->err);
    hipEventRecord(fe->ev_orb, S0);
    fe->n_img += 1;
    FT(FT_IMAGE_LAUNCH);
    return LVK_OK;
}

static lvk_status fe_check_failed(lvk_frontend* fe)
{
    if (fe->failed == LVK_OK) return LVK_OK;
    return lvk_set_error(fe->ctx, fe->failed, "the front-end is in a failed state after an earlier error (%s); destroy and re-create it", fe->failed_msg);
}
static lvk_status fe_note_failure(lvk_frontend* fe, lvk_status st)
{   // argument errors are raised before anything is queued and leave the handle usable; everything else is sticky
    if (st != LVK_OK && st != LVK_ERR_ARG && fe->failed == LVK_OK) {
        fe->failed = st; snprintf(fe->failed_msg, sizeof fe->failed_msg, "%s", fe->ctx->err);
        hipStreamSynchronize(fe->ctx->stream);
        for (int i = 0; i < 2; ++i) if (fe->side[i]) hipStreamSynchronize(fe->side[i]->stream);
    }
    return st;
}

lvk_status lvk_frontend_begin(lvk_frontend* fe, const lvk_image* img, double ts)
{
    if (!fe || !img) return LVK_ERR_ARG;
    { lvk_status fs = fe_check_failed(fe); if (fs != LVK_OK) return fs; }
    if (!fe->b_first_img) return fe_check_image(fe, img);   // the first-image gate needs the IMU buffer (:134-142): nothing to do early
    // one image stage per frame: the buffer-set rotation and the end-of-frame events are indexed by the stages queued so far
    if (fe->image_done) return lvk_set_error(fe->ctx, LVK_ERR_ARG, "lvk_frontend_begin: the image stage of t = %.6f is still waiting for its lvk_frontend_process", fe->image_done_ts);
    lvk_status st = fe_image_stage(fe, img, true);
    if (st != LVK_OK) return fe_note_failure(fe, st);
    fe->image_done = true; fe->image_done_ts = ts;
    return LVK_OK;
}

// processImage is synchronous in the reference: when it returns the caller may free or overwrite the image.  Same here: a host
// image has been copied into the pinned staging ring by the time the call returns (the frame's kernels keep running).
static lvk_status frontend_process(lvk_frontend* fe, const lvk_image* img, double ts, const lvk_imu* h_imu, int n_imu,
                                   lvk_feature_obs* h_out, int cap, int* n_out, int* has_msg, int* async_slot);
lvk_status lvk_frontend_process(lvk_frontend* fe, const lvk_image* img, double ts, const lvk_imu* h_imu, int n_imu,
                                lvk_feature_obs* h_out, int cap, int* n_out, int* has_msg)
{
    if (fe) { lvk_status fs = fe_check_failed(fe); if (fs != LVK_OK) return fs; }
    const lvk_status st = frontend_process(fe, img, ts, h_imu, n_imu, h_out, cap, n_out, has_msg, nullptr);
    return fe ? fe_note_failure(fe, st) : st;
}
// the pipelined driver's variant: nothing waits for the GPU; *slot names the ring entry the message will appear in
lvk_status lvk_frontend_process_async(lvk_frontend* fe, const lvk_image* img, double ts, const lvk_imu* h_imu, int n_imu, int* has_msg, int* slot)
{
    int n = 0;
    if (!slot) return LVK_ERR_ARG;
    *slot = -1;
    if (fe) { lvk_status fs = fe_check_failed(fe); if (fs != LVK_OK) return fs; }
    const lvk_status st = frontend_process(fe, img, ts, h_imu, n_imu, nullptr, 0, &n, has_msg, slot);
    return fe ? fe_note_failure(fe, st) : st;
}
// wait for the message of ring entry `slot` and copy it out (any thread)
lvk_status lvk_frontend_fetch_msg(lvk_frontend* fe, int slot, lvk_feature_obs* h_out, int cap, int* n_out)
{
    if (!fe || slot < 0 || slot >= LVK_MSG_SLOTS || !n_out) return LVK_ERR_ARG;
    const hipError_t er = hipEventSynchronize(fe->ev_msg[slot]);
    if (er != hipSuccess) return lvk_set_error(fe->ctx, LVK_ERR_DEVICE, "feature message: %s", hipGetErrorString(er));
    int n = fe->h_nmsg[16 * slot];
    if (n > cap) n = cap;
    if (h_out && n > 0) memcpy(h_out, fe->h_msg + (size_t)slot * fe->cap, sizeof(lvk_feature_obs) * (size_t)n);
    *n_out = n;
    fe->msg_pending[slot].store(0, std::memory_order_release);
    return LVK_OK;
}
static lvk_status frontend_process(lvk_frontend* fe, const lvk_image* img, double ts, const lvk_imu* h_imu, int n_imu,
                                   lvk_feature_obs* h_out, int cap, int* n_out, int* has_msg, int* async_slot)
{
    if (!fe || !img || !n_out || !has_msg || (n_imu > 0 && !h_imu)) return lvk_set_error(fe ? fe->ctx : nullptr, LVK_ERR_ARG, "lvk_frontend_process: bad argument");
    lvk_context* ctx = fe->ctx;
    const lvk_fe_config& c = fe->cfg;
    *n_out = 0; *has_msg = 0;
    if (!fe->b_first_img) {                                          // :134-142
        if (n_imu > 0 && h_imu[0].t - ts <= 0.0) fe->b_first_img = true;
        else return fe_check_image(fe, img);
    }
    lvk_status st = LVK_OK;
    if (fe->image_done && fe->image_done_ts != ts)
        return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_frontend_process(t = %.6f) after lvk_frontend_begin(t = %.6f): the queued image stage belongs to another frame", ts, fe->image_done_ts);
    if (!fe->image_done) st = fe_image_stage(fe, img, false);
    else FT(FT_OUTSIDE);                                 // (the stage was queued by lvk_frontend_begin: what lies between is the caller's)
    fe->image_done = false;
    if (st != LVK_OK) return st;
    hipStream_t S1 = ctx->stream, S2 = fe->side[0]->stream;
    {   // this frame's pyramid (and ORB planes) come from the image stream
        hipEvent_t ready = fe->pyr_event ? fe->ev_pyr : fe->ev_orb;
        if (!(fe->ev_trim && hipEventQuery(ready) == hipSuccess)) { (void)hipGetLastError(); hipStreamWaitEvent(S1, ready, 0); hipStreamWaitEvent(S2, ready, 0); }
    }
    // the side stream reads (new points, their count) and overwrites (wn_*) what the previous frame's commits on the main stream used
    if (fe->n_img >= 2) fe_wait_unless_done(fe, S2, fe->ev_end[fe->n_img & 1]);
    FT(FT_FRAME_WAITS);
    fe->prof_take = fe->prof_stride <= 1 || fe->n_img % fe->prof_stride == 0;
    fe->last_msg_slot = -1;
    fe->curr_img_time = ts;
    const double pub_gate = 0.9 * (1.0 / c.pub_frequency);
    const int src = fe->cur, dst = fe->cur ^ 1;
    bool curr_valid = false;            // curr_pts_ (set[dst]) holds this frame's tracks

    if (fe->image_state != 3) {
        // bootstrap frames run on the main stream alone, behind whatever side[0] still has in flight
        hipEventRecord(fe->ev_tail, S2);
        hipStreamWaitEvent(S1, fe->ev_tail, 0);
    }
    if (fe->image_state == 1) {
        // initializeFirstFrame (:337-352): goodFeaturesToTrack(max_features_num, 0.01, min_distance), no mask
        { ProfScope ps(fe, 6); st = lvk_min_eigen_map(ctx, fe->pyr[1], fe->eig); }
        if (st == LVK_OK) st = lvk_gftt_run(ctx, fe->eig, nullptr, c.width, c.height, c.max_features_num, 0.01, (double)c.min_distance, fe->gf_scratch,
                                            fe->gf_cands, fe->gf_cand_cap, fe->new_pts, fe->cap, &fe->dev->n_new, nullptr, false, false);
        if (st == LVK_OK) st = fe_read_dev(fe);
        if (st != LVK_OK) return st;
        fe->last_pub_time = ts;
        if (fe->h_dev->n_new > 20) fe->image_state = 2;
    } else {
        HMat H;
        st = lvk_predict_homography(h_imu, n_imu, fe->prev_img_time, ts, c.R_cam_imu, c.intrinsics, H.h);
        if (st != LVK_OK) return lvk_set_error(ctx, st, "predict_homography failed");
        FT(FT_PREDICT);
        if (fe->image_state == 2) {
            // initializeFirstFeatures (:355-537)
            st = track_chain(fe, S1, fe->new_pts, &fe->dev->n_new, H, fe->wn_curr, fe->wn_status, nullptr, fe->wn_desc, 1);
            if (st == LVK_OK) st = commit(fe, 2, fe->new_pts, &fe->dev->n_new, fe->wn_curr, fe->wn_status, nullptr, fe->wn_desc, dst);
            if (st == LVK_OK) st = fe_read_dev(fe);
            if (st != LVK_OK) return st;
            if (!fe->h_dev->boot_ok) fe->image_state = 1;
            else {
                curr_valid = true;
                if (!fe->ev_trim || (fe->frame_early && ts - fe->last_pub_time >= pub_gate)) hipEventRecord(fe->ev_commit, S1);      // only the detection of a publish frame waits for it - and in the blocking schedules for the message's event instead (fe_detect_new)
                if (ts - fe->last_pub_time >= pub_gate) {
                    st = fe_publish(fe, dst, ts, h_out, cap, n_out, async_slot);
                    if (st != LVK_OK) return st;
                    *has_msg = 1;
                }
                fe->image_state = 3;
            }
        } else {
            // trackFeatures (:540-811) on the main stream; trackNewFeatures' LK and descriptor gate (:813-931) on side[0], queued
            // behind the detection that produced the points; its RANSAC + append (:932-1001) joins the main stream.
            // (the side stream already waited for ev_pyr before the ORB planes: this frame's pyramid and everything the previous
            //  frame left on the main stream are done)
            const TrackSet& so = fe->set[src];
            if (lk_merged_ok(fe)) {
                // ONE launch for both chains (LkSet), on the main stream.  The new points were written by the detection the previous publish
                // frame queued on side[0]: its end-of-frame event (fired long ago unless the caller runs a frame ahead of the GPU)
                if (fe->n_img >= 2) fe_wait_unless_done(fe, S1, fe->ev_side[(int)(fe->n_img & 1)]);
                st = track_chain_merged(fe, S1, src, H);
                if (st != LVK_OK) return st;
                FT(FT_TRACK_LAUNCH);
                st = commit(fe, 0, so.pts, &fe->dev->n_tracks[src], fe->w_curr, fe->w_status, &so, so.desc, dst);
                if (st == LVK_OK) st = commit(fe, 1, fe->new_pts, &fe->dev->n_new, fe->wn_curr, fe->wn_status, nullptr, fe->wn_desc, dst);
                if (st != LVK_OK) return st;
            } else {
            st = track_chain(fe, S2, fe->new_pts, &fe->dev->n_new, H, fe->wn_curr, fe->wn_status, nullptr, fe->wn_desc, 1);
            hipEventRecord(fe->ev_new, S2);
            if (st == LVK_OK) st = track_chain(fe, S1, fe->set[src].pts, &fe->dev->n_tracks[src], H, fe->w_curr, fe->w_status, fe->set[src].desc, nullptr, 0);
            if (st != LVK_OK) return st;
            FT(FT_TRACK_LAUNCH);
            // the old tracks' RANSAC + commit overlaps the wait for the new points' chain; their append and the message follow.
            // (Both commits - and all three stages - as ONE launch were measured in same-box A/B runs and were slower: the old tracks'
            //  commit then waits for the new points' chain; profiles/r3_jk_frontend_chain_ab.json)
            st = commit(fe, 0, so.pts, &fe->dev->n_tracks[src], fe->w_curr, fe->w_status, &so, so.desc, dst);
            hipStreamWaitEvent(S1, fe->ev_new, 0);
            if (st == LVK_OK) st = commit(fe, 1, fe->new_pts, &fe->dev->n_new, fe->wn_curr, fe->wn_status, nullptr, fe->wn_desc, dst);
            if (st != LVK_OK) return st;
            }
            curr_valid = true;
            if (!fe->ev_trim || (fe->frame_early && ts - fe->last_pub_time >= pub_gate)) hipEventRecord(fe->ev_commit, S1);          // only the detection of a publish frame waits for it - and in the blocking schedules for the message's event instead (fe_detect_new)
            FT(FT_COMMIT_LAUNCH);
            if (ts - fe->last_pub_time >= pub_gate) {
                st = fe_publish(fe, dst, ts, h_out, cap, n_out, async_slot);
                if (st != LVK_OK) return st;
                *has_msg = 1;
            }
        }
    }
    if (!curr_valid) LVK_HIP(ctx, hipMemsetAsync(&fe->dev->n_tracks[dst], 0, sizeof(int), ctx->stream));
    {   const int par = (int)((fe->n_img - 1) & 1);
        if (fe->ev_trim && fe->last_msg_slot >= 0 && curr_valid) fe->ev_end[par] = fe->ev_msg[fe->last_msg_slot];     // recorded right behind the message kernel, the frame's last launch on this stream
        else { hipEventRecord(fe->ev_main[par], S1); fe->ev_end[par] = fe->ev_main[par]; }
        hipEventRecord(fe->ev_side[par], S2); }
    // rotation (:207-216), three-way: curr becomes prev, the spare set becomes the next frame's curr
    { lvk_pyramid* p = fe->pyr[0]; fe->pyr[0] = fe->pyr[1]; fe->pyr[1] = fe->pyr[2]; fe->pyr[2] = p; }
    { uint8_t* p = fe->ext[0]; fe->ext[0] = fe->ext[1]; fe->ext[1] = fe->ext[2]; fe->ext[2] = p;
      p = fe->blur[0]; fe->blur[0] = fe->blur[1]; fe->blur[1] = fe->blur[2]; fe->blur[2] = p; }
    fe->cur = dst;
    fe->prev_img_time = ts;
    if (fe->pending.size() > 4096) prof_collect(fe);
    FT(FT_END); g_ft.n++;
    return LVK_OK;
}

lvk_status lvk_frontend_tracks(lvk_frontend* fe, uint64_t* h_ids, lvk_pt2f* h_pts, int* h_lifetime, lvk_pt2f* h_init, uint8_t* h_desc, int cap, int* n_out)
{
    if (!fe || !n_out) return LVK_ERR_ARG;
    { lvk_status qs = fe_quiesce(fe); if (qs != LVK_OK) return qs; }
    lvk_status st = fe_read_dev(fe);
    if (st != LVK_OK) return st;
    const TrackSet& s = fe->set[fe->cur];
    int n = fe->h_dev->n_tracks[fe->cur];
    if (n > cap) n = cap;
    lvk_context* ctx = fe->ctx;
    if (n > 0) {
        if (h_ids) LVK_HIP(ctx, hipMemcpy(h_ids, s.id, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost));
        if (h_pts) LVK_HIP(ctx, hipMemcpy(h_pts, s.pts, sizeof(lvk_pt2f) * (size_t)n, hipMemcpyDeviceToHost));
        if (h_lifetime) LVK_HIP(ctx, hipMemcpy(h_lifetime, s.life, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
        if (h_init) LVK_HIP(ctx, hipMemcpy(h_init, s.init, sizeof(lvk_pt2f) * (size_t)n, hipMemcpyDeviceToHost));
        if (h_desc) LVK_HIP(ctx, hipMemcpy(h_desc, s.desc, (size_t)32 * n, hipMemcpyDeviceToHost));
    }
    *n_out = n;
    return LVK_OK;
}

lvk_status lvk_frontend_new_pts(lvk_frontend* fe, lvk_pt2f* h_pts, int cap, int* n_out)
{
    if (!fe || !n_out) return LVK_ERR_ARG;
    { lvk_status qs = fe_quiesce(fe); if (qs != LVK_OK) return qs; }
    lvk_status st = fe_read_dev(fe);
    if (st != LVK_OK) return st;
    int n = fe->h_dev->n_new; if (n > cap) n = cap;
    if (n > 0 && h_pts) LVK_HIP(fe->ctx, hipMemcpy(h_pts, fe->new_pts, sizeof(lvk_pt2f) * (size_t)n, hipMemcpyDeviceToHost));
    *n_out = n;
    return LVK_OK;
}

int lvk_frontend_state(const lvk_frontend* fe) { return fe ? fe->image_state : 0; }

lvk_status lvk_frontend_profile_enable(lvk_frontend* fe, unsigned stage_mask)
{
    if (!fe) return LVK_ERR_ARG;
    prof_collect(fe);
    fe->prof_mask = stage_mask & 0xFFFFu; fe->prof_stride = (stage_mask >> 16) & 0xFFu; fe->prof_take = true;
    return LVK_OK;
}
lvk_status lvk_frontend_profile_read(lvk_frontend* fe, double ms_sum[LVK_FE_STAGES], uint64_t launches[LVK_FE_STAGES], int reset)
{
    if (!fe) return LVK_ERR_ARG;
    prof_collect(fe);
    for (int i = 0; i < LVK_FE_STAGES; ++i) {
        if (ms_sum) ms_sum[i] = fe->prof_ms[i];
        if (launches) launches[i] = fe->prof_n[i];
        if (reset) { fe->prof_ms[i] = 0.; fe->prof_n[i] = 0; }
    }
    return LVK_OK;
}
const char* lvk_frontend_stage_name(int stage)
{
    static const char* names[LVK_FE_STAGES] = {"pyramid_clahe", "orb_prepare", "lk_fwd_rev", "(unused)", "orb_gate", "ransac_commit", "min_eigen", "gftt_select", "feature_msg"};
    return stage >= 0 && stage < LVK_FE_STAGES ? names[stage] : "?";
}

lvk_status lvk_frontend_lk_stats(lvk_frontend* fe, uint64_t* point_levels, uint64_t* iterations)
{
    if (!fe) return LVK_ERR_ARG;
    { lvk_status qs = fe_quiesce(fe); if (qs != LVK_OK) return qs; }
    lvk_status st = fe_read_dev(fe);
    if (st != LVK_OK) return st;
    if (point_levels) *point_levels = fe->h_dev->lk_point_levels;
    if (iterations) *iterations = fe->h_dev->lk_iterations;
    return LVK_OK;
}

lvk_status lvk_frontend_msg_stats(lvk_frontend* fe, uint64_t* messages, uint64_t* features)
{
    if (!fe) return LVK_ERR_ARG;
    { lvk_status qs = fe_quiesce(fe); if (qs != LVK_OK) return qs; }
    lvk_status st = fe_read_dev(fe);
    if (st != LVK_OK) return st;
    if (messages) *messages = fe->h_dev->msg_count;
    if (features) *features = fe->h_dev->msg_features;
    return LVK_OK;
}

}  // extern "C"
