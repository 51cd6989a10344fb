// fe_track.hip — per-point stages of the front-end for gfx950 (wave64):
//   pyramidal LK (one wavefront per track, all levels and iterations inside one launch, template
//   window and gradients in registers, 2x2 system via exact integer wave reductions),
//   ORB angle + rotated-BRIEF descriptor (one wavefront per keypoint, descriptor bits via ballot),
//   Hamming rows, undistortion, fundamental-matrix RANSAC / LMedS (one workgroup, hypotheses
//   evaluated in parallel, OpenCV's sequential adaptive-termination rule replayed exactly).
// Replaces the OpenCV / ORBdescriptor calls at /root/reference/src/image_processor.cpp:368-377,
// 405-414, 444-454, 479-500, 558-567, 618-627, 679-699, 736-757 and src/ORBDescriptor.cpp:335-416,486-514.
// Built with -ffp-contract=off: float32/float64 sequences must match the CPU oracle bit-for-bit.
#include "lvk_internal.h"
#include "fe_track_dev.h"

// =========================================================================== stage-level kernels
template <int WIN, int VAR>
__global__ void __launch_bounds__(64) k_lk_track(PyrView prev, PyrView next, const lvk_pt2f* __restrict__ prev_pts,
                                                lvk_pt2f* __restrict__ next_pts, uint8_t* __restrict__ status, int n,
                                                int max_count, double epsilon, int* __restrict__ iters)
{
    const int p = blockIdx.x;
    if (p >= n) return;
    const int n_levels = prev.n_levels < next.n_levels ? prev.n_levels : next.n_levels;
    lvk_pt2f np = next_pts[p];
    int st = 1;
    __shared__ __attribute__((aligned(16))) unsigned long long s_acc[4];
    LkLdsAcc acc = lk_acc_init(s_acc);
    lk_point<WIN, VAR>(prev, next, n_levels, prev_pts[p], np, st, max_count, epsilon, iters ? iters + (size_t)p * n_levels : nullptr, acc);
    if ((threadIdx.x & 63) == 0) { next_pts[p] = np; status[p] = (uint8_t)st; }
}

__global__ void __launch_bounds__(64) k_orb_describe(const uint8_t* __restrict__ ext, const uint8_t* __restrict__ blur, int w,
                                                    const lvk_pt2f* __restrict__ pts, int n, uint8_t* __restrict__ desc,
                                                    float* __restrict__ angle_out)
{
    const int p = blockIdx.x;
    if (p >= n) return;
    unsigned long long d[4];
    float ang = orb_point(ext, blur, w + 2 * LVK_ORB_BORDER, pts[p], d);
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(desc + (size_t)p * 32);
        o[0] = d[0]; o[1] = d[1]; o[2] = d[2]; o[3] = d[3];
        if (angle_out) angle_out[p] = ang;
    }
}

__global__ void k_hamming_rows(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int n, int* __restrict__ dist)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dist[i] = hamming256(reinterpret_cast<const uint32_t*>(a + (size_t)i * 32), reinterpret_cast<const uint32_t*>(b + (size_t)i * 32));
}

__global__ void k_undistort(const lvk_pt2f* __restrict__ in, int n, CamParams cam, double n0, double n1, double n2, double n3,
                            lvk_pt2f* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ni[4] = {n0, n1, n2, n3};
    out[i] = undistort_point(in[i], cam, ni);
}

template <int NT>
__global__ void __launch_bounds__(NT) k_fundamental_mask(const lvk_pt2f* __restrict__ p1, const lvk_pt2f* __restrict__ p2, int n,
                                                       double thresh, double conf, int max_iters, int force_ransac,
                                                       uint8_t* __restrict__ mask, int* __restrict__ info, double* __restrict__ model)
{
    __shared__ lvk_pt2f s1[FM_MAX_N], s2[FM_MAX_N];
    __shared__ uint8_t smask[FM_MAX_N];
    for (int i = threadIdx.x; i < n; i += NT) { s1[i] = p1[i]; s2[i] = p2[i]; }
    __syncthreads();
    int iters = 0;
    int wrote = fm_mask_block<NT>(s1, s2, n, thresh, conf, max_iters, force_ransac, smask, &iters, model);
    __syncthreads();
    if (wrote) for (int i = threadIdx.x; i < n; i += NT) mask[i] = smask[i];
    if (threadIdx.x == 0 && info) { info[0] = wrote; info[1] = iters; }
}

// =========================================================================== host side of the ABI
template <int WIN>
static void launch_lk(lvk_context* ctx, const PyrView& a, const PyrView& b, const lvk_pt2f* pp, lvk_pt2f* np, uint8_t* st, int n,
                      int max_count, double epsilon, int* iters)
{
    if (WIN == 21 && lvk_lk_variant() == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lk_track<WIN, 0>), dim3(n), dim3(64), 0, ctx->stream, a, b, pp, np, st, n, max_count, epsilon, iters);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lk_track<WIN, 1>), dim3(n), dim3(64), 0, ctx->stream, a, b, pp, np, st, n, max_count, epsilon, iters);
}

extern "C" {

lvk_status lvk_lk_track(lvk_context* ctx, const lvk_pyramid* prev, const lvk_pyramid* next, const lvk_pt2f* d_prev_pts,
                        lvk_pt2f* d_next_pts, uint8_t* d_status, int n, int max_iter, double eps, int* d_iters)
{
    if (!ctx || !prev || !next || (n > 0 && (!d_prev_pts || !d_next_pts || !d_status)) || n < 0)
        return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_lk_track: bad argument");
    if (prev->pad != next->pad) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_lk_track: pyramids built with different windows");
    if (n == 0) return LVK_OK;
    int max_count = max_iter < 0 ? 0 : max_iter > 100 ? 100 : max_iter;
    double epsilon = eps < 0. ? 0. : eps > 10. ? 10. : eps;
    epsilon *= epsilon;
    PyrView a = make_view(prev), b = make_view(next);
    switch (prev->pad) {
        case 21: launch_lk<21>(ctx, a, b, d_prev_pts, d_next_pts, d_status, n, max_count, epsilon, d_iters); break;
        case 15: launch_lk<15>(ctx, a, b, d_prev_pts, d_next_pts, d_status, n, max_count, epsilon, d_iters); break;
        case 31: launch_lk<31>(ctx, a, b, d_prev_pts, d_next_pts, d_status, n, max_count, epsilon, d_iters); break;
        default: return lvk_set_error(ctx, LVK_ERR_UNSUPPORTED, "LK window %d not instantiated (15, 21, 31)", prev->pad);
    }
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

lvk_status lvk_orb_describe(lvk_context* ctx, const uint8_t* d_ext, const uint8_t* d_blur, int w, int h, const lvk_pt2f* d_pts, int n,
                            uint8_t* d_desc, float* d_angle)
{
    (void)h;
    if (!ctx || !d_ext || !d_blur || n < 0 || (n > 0 && (!d_pts || !d_desc))) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_orb_describe: bad argument");
    if (n == 0) return LVK_OK;
    hipLaunchKernelGGL(k_orb_describe, dim3(n), dim3(64), 0, ctx->stream, d_ext, d_blur, w, d_pts, n, d_desc, d_angle);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

lvk_status lvk_hamming256_rows(lvk_context* ctx, const uint8_t* d_a, const uint8_t* d_b, int n, int* d_dist)
{
    if (!ctx || n < 0 || (n > 0 && (!d_a || !d_b || !d_dist))) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_hamming256_rows: bad argument");
    if (n == 0) return LVK_OK;
    hipLaunchKernelGGL(k_hamming_rows, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_a, d_b, n, d_dist);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

lvk_status lvk_undistort_points(lvk_context* ctx, const lvk_pt2f* d_in, int n, const double intr[4], int model, const double dist[4],
                                const double new_intr[4], lvk_pt2f* d_out)
{
    if (!ctx || n < 0 || (n > 0 && (!d_in || !d_out)) || !intr || !dist || !new_intr || (model != 0 && model != 1))
        return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_undistort_points: bad argument");
    if (n == 0) return LVK_OK;
    CamParams cam; memset(&cam, 0, sizeof cam);
    for (int i = 0; i < 4; ++i) { cam.intr[i] = intr[i]; cam.dist[i] = dist[i]; }
    cam.model = model;
    hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_in, n, cam, new_intr[0], new_intr[1], new_intr[2], new_intr[3], d_out);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

static lvk_status fm_launch(lvk_context* ctx, const lvk_pt2f* p1, const lvk_pt2f* p2, int n, double thresh, double conf, int max_iters,
                            int force_ransac, uint8_t* mask, int* info, double* model = nullptr)
{
    if (!ctx || n < 0 || (n > 0 && (!p1 || !p2 || !mask))) return lvk_set_error(ctx, LVK_ERR_ARG, "fundamental: bad argument");
    if (n > FM_MAX_N) return lvk_set_error(ctx, LVK_ERR_CAPACITY, "fundamental: n=%d exceeds %d", n, FM_MAX_N);
    // same split as the frame path (frontend.hip: commit): the wide workgroup for point sets the per-point loops dominate
    if (n > FM_WIDE_FROM) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fundamental_mask<FM_THREADS_WIDE>), dim3(1), dim3(FM_THREADS_WIDE), 0, ctx->stream, p1, p2, n, thresh, conf, max_iters, force_ransac, mask, info, model);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fundamental_mask<FM_THREADS>), dim3(1), dim3(FM_THREADS), 0, ctx->stream, p1, p2, n, thresh, conf, max_iters, force_ransac, mask, info, model);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

lvk_status lvk_find_fundamental_mask(lvk_context* ctx, const lvk_pt2f* d_p1, const lvk_pt2f* d_p2, int n, double thresh, double conf,
                                     uint8_t* d_mask, int* d_info)
{
    return fm_launch(ctx, d_p1, d_p2, n, thresh, conf, 1000, 0, d_mask, d_info);
}

lvk_status lvk_find_fundamental(lvk_context* ctx, const lvk_pt2f* d_p1, const lvk_pt2f* d_p2, int n, double thresh, double conf,
                                uint8_t* d_mask, int* d_info, double* d_F)
{
    if (!d_F) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_find_fundamental: d_F is null");
    if (n < 7) { if (ctx) hipMemsetAsync(d_F, 0, 9 * sizeof(double), ctx->stream); }      // (the kernel returns before it touches anything)
    return fm_launch(ctx, d_p1, d_p2, n, thresh, conf, 1000, 0, d_mask, d_info, d_F);
}

lvk_status lvk_ransac_fundamental(lvk_context* ctx, const lvk_pt2f* d_p1, const lvk_pt2f* d_p2, int n, double thresh, double conf,
                                  int max_iters, uint8_t* d_mask, int* d_info)
{
    if (n < 8) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_ransac_fundamental: n >= 8 required");
    return fm_launch(ctx, d_p1, d_p2, n, thresh, conf, max_iters, 1, d_mask, d_info);
}

}  // extern "C"
