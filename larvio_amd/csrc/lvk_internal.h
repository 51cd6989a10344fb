// lvk_internal.h — shared internals of liblvk_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/lvk_c.h"

// spin-wait hint of a host thread (x86 pause; nothing elsewhere)
#if defined(__x86_64__) || defined(__i386__)
#define LVK_CPU_RELAX() __builtin_ia32_pause()
#else
#define LVK_CPU_RELAX() do { } while (0)
#endif

// store fence after host writes that a kernel launched next must see (write-combined BAR mappings: sfence on x86)
#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#define LVK_STORE_FENCE() _mm_sfence()
#else
#define LVK_STORE_FENCE() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#endif

// May this process push host writes into device memory through the PCIe BAR on `device`?  frontend.hip: LVK_BAR_PUSH=0 says no; a
// device without a large BAR says no; otherwise a self-test decides, once per device and process - the host writes a pattern into a
// fine-grained device buffer, a kernel reads it, the host REWRITES the same addresses, a second kernel reads again: a stale second
// read (L2 lines of the first read surviving the kernel boundary under this driver's MTYPE / partition settings) says no.
bool lvk_bar_usable(int device);

// Device memory that the HOST writes through the PCIe BAR (the filter's upload arena, the blocking front-end's image buffers): a
// fine-grained device allocation when the device reports a large BAR AND the process really has a writable mapping of it (checked in
// /proc/self/maps: a container may hide what the attribute promises); nullptr otherwise - callers then keep their pinned-host path.
static inline void* lvk_bar_alloc(int device, size_t bytes)
{
    void* p = nullptr;
    if (!lvk_bar_usable(device)) return nullptr;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess || !p) { (void)hipGetLastError(); return nullptr; }
    bool writable = false;
    if (FILE* f = fopen("/proc/self/maps", "r")) {
        char line[512]; unsigned long lo = 0, hi = 0; char perm[8] = {0};
        while (fgets(line, sizeof line, f))
            if (sscanf(line, "%lx-%lx %7s", &lo, &hi, perm) == 3 && (unsigned long)p >= lo && (unsigned long)p < hi) { writable = perm[0] == 'r' && perm[1] == 'w'; break; }
        fclose(f);
    }
    if (!writable) { (void)hipFree(p); return nullptr; }
    return p;
}

#define LVK_MAX_LEVELS 8
#define LVK_ORB_BORDER 32

#define LVK_SCRATCH_SLOTS 13
struct lvk_context {
    int device;
    hipStream_t stream;
    bool own_stream;
    char err[512];
    // grow-only scratch buffers for the stage-level entry points (no stream-ordered pool: a fresh
    // pool block handed out mid-stream was observed to corrupt an in-flight launch sequence)
    void* scratch[LVK_SCRATCH_SLOTS];
    size_t scratch_bytes[LVK_SCRATCH_SLOTS];
    // dynamic-LDS opt-ins (hipFuncAttributeMaxDynamicSharedMemorySize) already made through THIS context: function attributes are
    // per device, so the cache lives here and not in a process-wide static (one slot per kernel instantiation, numbered at the call sites)
    size_t lds_optin[24];
    // fused Cholesky + solve (be_linalg.hip, k_chol_fused): the factor workgroup hands panels to the solver workgroups through a flag
    // that only ever grows; chol_epoch numbers the launches, chol_ws (scratch slot 12) holds the diagonal-block inverses and the flag
    int chol_epoch;
};
// opt in to `bytes` of dynamic LDS for `fn` if this context has not already asked for at least that much
#define LVK_LDS_OPTIN(ctx, slot, fn, bytes)                                                                             \
    do {                                                                                                                \
        if ((ctx)->lds_optin[slot] < (size_t)(bytes)) {                                                                 \
            LVK_HIP(ctx, hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
            (ctx)->lds_optin[slot] = (size_t)(bytes);                                                                   \
        }                                                                                                               \
    } while (0)
void* lvk_ctx_scratch(lvk_context* ctx, int slot, size_t bytes);
struct lvk_pyramid;
lvk_status lvk_pyramid_build_with_orb(lvk_context* ctx, lvk_pyramid* p, const uint8_t* d_img, int stride, int clahe, double clip_limit,
                                      int tiles_x, int tiles_y, uint8_t* d_ext, int* mosaic_done);
lvk_status lvk_orb_blur_only(lvk_context* ctx, const lvk_pyramid* p, const uint8_t* d_ext, uint8_t* d_blur);
struct lvk_frontend;
lvk_context* lvk_frontend_context(lvk_frontend* fe);      // frontend.hip
extern "C" lvk_status lvk_frontend_begin(lvk_frontend* fe, const lvk_image* img, double ts);   // image stage only (internal)
// the pipelined driver's non-blocking processImage: *slot = ring entry of the feature message (when *has_msg), collected later
extern "C" lvk_status lvk_frontend_process_async(lvk_frontend* fe, const lvk_image* img, double ts, const lvk_imu* h_imu, int n_imu, int* has_msg, int* slot);
extern "C" lvk_status lvk_frontend_fetch_msg(lvk_frontend* fe, int slot, lvk_feature_obs* h_out, int cap, int* n_out);

struct lvk_pyramid {
    lvk_context* ctx;
    int n_levels, pad, max_level;
    int w[LVK_MAX_LEVELS], h[LVK_MAX_LEVELS];
    int istride[LVK_MAX_LEVELS];   // bytes per padded image row
    int dstride[LVK_MAX_LEVELS];   // int16 per padded derivative row
    uint8_t* img[LVK_MAX_LEVELS];  // padded buffers
    int16_t* der[LVK_MAX_LEVELS];
    uint8_t* clahe_lut;            // tiles*256 scratch for the fused CLAHE path
    int clahe_lut_cap;
    hipEvent_t ev_level0;          // optional: recorded on the build stream once level 0 is complete (side streams fork here)
};

// plain-data view of a pyramid passed to kernels by value
struct PyrView {
    int n_levels, pad;
    int w[LVK_MAX_LEVELS], h[LVK_MAX_LEVELS], istride[LVK_MAX_LEVELS], dstride[LVK_MAX_LEVELS];
    const uint8_t* img[LVK_MAX_LEVELS];   // pointer to pixel (0,0) of the level (inside the padded buffer)
    const int16_t* der[LVK_MAX_LEVELS];   // pointer to (Ix,Iy) of pixel (0,0)
};

static inline PyrView make_view(const lvk_pyramid* p)
{
    PyrView v;
    memset(&v, 0, sizeof v);
    v.n_levels = p->n_levels; v.pad = p->pad;
    for (int l = 0; l < p->n_levels; ++l) {
        v.w[l] = p->w[l]; v.h[l] = p->h[l]; v.istride[l] = p->istride[l]; v.dstride[l] = p->dstride[l];
        v.img[l] = p->img[l] + (size_t)p->pad * p->istride[l] + p->pad;
        v.der[l] = p->der[l] + (size_t)p->pad * p->dstride[l] + 2 * p->pad;
    }
    return v;
}

lvk_status lvk_set_error(lvk_context* ctx, lvk_status code, const char* fmt, ...);

#define LVK_HIP(ctx, call)                                                                    \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return lvk_set_error((ctx), LVK_ERR_DEVICE, "%s:%d %s: %s", __FILE__, __LINE__, #call, \
                                 hipGetErrorString(e_));                                      \
    } while (0)

#define LVK_LAUNCH_CHECK(ctx) LVK_HIP(ctx, hipGetLastError())

// ---- compile-time instrumentation (make CXXFLAGS+=-DLVK_BE_TIMING; tools/gpu/be_ticks.py): thread 0 of ONE chosen workgroup of a kernel
// stores the 100 MHz wall clock at its phase boundaries into a per-file device array; off, the macros vanish
#ifdef LVK_BE_TIMING
#define BE_TICK_DECL(name) static __device__ unsigned long long name[64]
#define BE_TICK(arr, cond, k) do { if (threadIdx.x == 0 && (cond)) arr[k] = wall_clock64(); } while (0)
#define BE_TICK_GETTER(fn, arr) extern "C" void fn(unsigned long long* out) { hipDeviceSynchronize(); hipMemcpyFromSymbol(out, HIP_SYMBOL(arr), sizeof(unsigned long long) * 64); }
#else
#define BE_TICK_DECL(name)
#define BE_TICK(arr, cond, k) do { } while (0)
#define BE_TICK_GETTER(fn, arr)
#endif

// LK kernel variant (fe_track_dev.h): LVK_LK_VARIANT in the environment (0, 1, 2), read once per process; unset = -1 = the library chooses
// (frame path: the five-wavefront kernel up to LVK_LK_PIPE_MAX_TRACKS track slots, the two-wavefront one above - at 2000 tracks five
// wavefronts and 12 KB of LDS per track no longer fit the chip in one round: 100 us against 61, profiles/r6_h_lk_pipe_ab.json)
#define LVK_LK_PIPE_MAX_TRACKS 600
inline int lvk_lk_variant()
{
    static const int v = [] { const char* e = getenv("LVK_LK_VARIANT"); const int x = e ? atoi(e) : -1; return x < -1 || x > 2 ? -1 : x; }();
    return v;
}

// ---- device helpers shared by the kernels --------------------------------------------------
__device__ __forceinline__ int d_reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { p = p < 0 ? -p : 2 * len - 2 - p; }
    return p;
}
__device__ __forceinline__ int d_cv_round(float v) { return __float2int_rn(v); }   // half-to-even
__device__ __forceinline__ int d_cv_floor(float v) { return (int)floorf(v); }
__device__ __forceinline__ uint8_t d_sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

// parameter block for the per-point kernels of the frame-level path (device memory, refreshed per frame)
struct CamParams {
    double intr[4];
    double dist[4];
    int model;
    int width, height;
};
