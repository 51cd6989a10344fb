// fe_image.hip — full-image passes of the front-end for gfx950 (wave64):
// CLAHE, LK pyramid (+reflect-101 padding, Scharr planes), ORB mosaic + 7x7 blur, GFTT.
// Replaces the OpenCV calls at /root/reference/src/image_processor.cpp:318-334, 343, 1005-1037 and
// /root/reference/src/ORBDescriptor.cpp:418-484.  All of these are HBM/L2-bound byte passes; none
// is GEMM-shaped, so no MFMA here: the rules that matter are coalesced row access, LDS tiles for
// the stencils and as few dependent launches as possible (each costs ~1.5-2 us on MI355X).
#include "lvk_internal.h"
#include <vector>
#include <stdarg.h>
#include <algorithm>

lvk_status lvk_set_error(lvk_context* ctx, lvk_status code, const char* fmt, ...)
{
    if (ctx) {
        va_list ap; va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

void* lvk_ctx_scratch(lvk_context* ctx, int slot, size_t bytes)
{
    if (slot < 0 || slot >= LVK_SCRATCH_SLOTS) return nullptr;
    if (ctx->scratch_bytes[slot] >= bytes && ctx->scratch[slot]) return ctx->scratch[slot];
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return nullptr;
    if (ctx->scratch[slot]) hipFree(ctx->scratch[slot]);
    ctx->scratch[slot] = nullptr; ctx->scratch_bytes[slot] = 0;
    if (hipMalloc(&ctx->scratch[slot], bytes) != hipSuccess) return nullptr;
    ctx->scratch_bytes[slot] = bytes;
    return ctx->scratch[slot];
}

// =========================================================================== CLAHE
// Histogram of one tile in LDS (one sub-histogram per wave), clip + redistribute, inclusive scan -> LUT.  [cv::CLAHE CLAHE_CalcLut_Body]
// One workgroup per tile; when the tiles divide the image (every supported camera) the pixels come in as 32-bit words, four per load:
// 33 -> 13 us at 1920x1080, where a tile has 32,400 pixels.  The kernel can also cover a tile with S workgroups (each counts a band of
// rows, adds its 256 counts to the tile's histogram in global memory, the last arrival - ticket counter - finishes the tile and leaves
// histogram and ticket zeroed; integer counts, so the order of arrival does not matter), but that was measured SLOWER at every S
// (S = 2: 25 us, 4: 34 us, 8: 57 us at 1080p): the device-scope atomics and fences of the merge cost more than the idle CUs.  S stays 1;
// LVK_CLAHE_S overrides it for experiments.
__global__ void __launch_bounds__(256) k_clahe_lut(const uint8_t* __restrict__ src, int w, int h, int sstride,
                                                  int tw, int th, int tiles_x, int clip, float lut_scale,
                                                  uint8_t* __restrict__ lut, int vec4)
{
    __shared__ int sh[4][256];
    __shared__ int hist[256];
    __shared__ int red[4];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    for (int k = 0; k < 4; ++k) sh[k][t] = 0;
    __syncthreads();
    const int r0 = 0, r1 = th;
    // the frame was just written by another agent (camera DMA / another XCD): a dependent load is a trip to the memory side, so
    // the pixels of this thread are fetched in batches with all loads in flight before any of them is counted
    if (vec4) {                                                      // tiles divide the image, rows are word-aligned: four pixels per load
        const int wpr = tw >> 2, total = (r1 - r0) * wpr;
        const uint8_t* base = src + (size_t)(ty * th + r0) * sstride + (size_t)tx * tw;
        for (int i0 = t; i0 < total; i0 += 256 * 8) {
            unsigned v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 256 * u;
                if (i < total) { const int y = i / wpr, x4 = i - y * wpr; v[u] = *(const unsigned*)(base + (size_t)y * sstride + 4 * x4); }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + 256 * u < total) {
                    atomicAdd(&sh[wave][v[u] & 255u], 1); atomicAdd(&sh[wave][(v[u] >> 8) & 255u], 1);
                    atomicAdd(&sh[wave][(v[u] >> 16) & 255u], 1); atomicAdd(&sh[wave][v[u] >> 24], 1);
                }
        }
    } else {
        const int total = (r1 - r0) * tw;
        for (int i0 = t; i0 < total; i0 += 256 * 16) {
            int v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = i0 + 256 * u;
                if (i < total) {
                    int y = i / tw, x = i - y * tw;
                    int px = d_reflect101(tx * tw + x, w), py = d_reflect101(ty * th + r0 + y, h);   // ext tiles reflect (non-divisible sizes)
                    v[u] = src[(size_t)py * sstride + px];
                } else v[u] = -1;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) if (v[u] >= 0) atomicAdd(&sh[wave][v[u]], 1);
        }
    }
    __syncthreads();
    int v = sh[0][t] + sh[1][t] + sh[2][t] + sh[3][t];
    if (clip > 0) {
        int over = v > clip ? v - clip : 0;
        if (v > clip) v = clip;
        // block sum of `over`
        int s = over;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        int clipped = red[0] + red[1] + red[2] + red[3];
        int batch = clipped / 256, residual = clipped - batch * 256;
        v += batch;
        if (residual != 0) {
            int step = 256 / residual; if (step < 1) step = 1;
            if (t % step == 0 && t / step < residual) v += 1;
        }
    }
    hist[t] = v;
    __syncthreads();
    // inclusive scan over 256 ints (Hillis-Steele in LDS)
    for (int o = 1; o < 256; o <<= 1) {
        int add = t >= o ? hist[t - o] : 0;
        __syncthreads();
        hist[t] += add;
        __syncthreads();
    }
    lut[(size_t)tile * 256 + t] = d_sat_u8(d_cv_round((float)hist[t] * lut_scale));
}
// how a tile is split and whether the word-load path applies
static void clahe_launch_shape(const uint8_t* src, int w, int h, int sstride, int tw, int th, int tiles_x, int tiles_y, int* vec4)
{
    *vec4 = (tiles_x * tw == w && tiles_y * th == h && (tw & 3) == 0 && (sstride & 3) == 0 && ((size_t)src & 3) == 0) ? 1 : 0;
}

// CLAHE bilinear LUT blend at image coordinate (x,y)  [CLAHE_Interpolation_Body]
__device__ __forceinline__ uint8_t clahe_pixel(const uint8_t* __restrict__ lut, int tiles_x, int tiles_y,
                                               float inv_tw, float inv_th, int x, int y, int v)
{
    float tyf = y * inv_th - 0.5f;
    int ty1 = d_cv_floor(tyf), ty2 = ty1 + 1;
    float ya = tyf - ty1, ya1 = 1.0f - ya;
    ty1 = max(ty1, 0); ty2 = min(ty2, tiles_y - 1);
    float txf = x * inv_tw - 0.5f;
    int tx1 = d_cv_floor(txf), tx2 = tx1 + 1;
    float xa = txf - tx1, xa1 = 1.0f - xa;
    tx1 = max(tx1, 0); tx2 = min(tx2, tiles_x - 1);
    const uint8_t* p1 = lut + (size_t)ty1 * tiles_x * 256;
    const uint8_t* p2 = lut + (size_t)ty2 * tiles_x * 256;
    int i1 = tx1 * 256 + v, i2 = tx2 * 256 + v;
    float res = (p1[i1] * xa1 + p1[i2] * xa) * ya1 + (p2[i1] * xa1 + p2[i2] * xa) * ya;
    return d_sat_u8(d_cv_round(res));
}

__global__ void k_clahe_apply(const uint8_t* __restrict__ src, int w, int h, int sstride,
                              const uint8_t* __restrict__ lut, int tiles_x, int tiles_y, float inv_tw, float inv_th,
                              uint8_t* __restrict__ dst, int dstride)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    dst[(size_t)y * dstride + x] = clahe_pixel(lut, tiles_x, tiles_y, inv_tw, inv_th, x, y, src[(size_t)y * sstride + x]);
}

// level 0 of the LK pyramid incl. its reflect-101 frame, optionally through the CLAHE LUT blend:
// one thread per PADDED pixel; the value of a frame pixel is the value at its reflected coordinate.
template <bool EQ>
__global__ void k_level0_pad(const uint8_t* __restrict__ src, int w, int h, int sstride,
                             const uint8_t* __restrict__ lut, int tiles_x, int tiles_y, float inv_tw, float inv_th,
                             uint8_t* __restrict__ dst /*padded base*/, int pad, int dstride)
{
    int ex = blockIdx.x * blockDim.x + threadIdx.x, ey = blockIdx.y;
    if (ex >= w + 2 * pad) return;
    int x = d_reflect101(ex - pad, w), y = d_reflect101(ey - pad, h);
    int v = src[(size_t)y * sstride + x];
    uint8_t o = EQ ? clahe_pixel(lut, tiles_x, tiles_y, inv_tw, inv_th, x, y, v) : (uint8_t)v;
    dst[(size_t)ey * dstride + ex] = o;
}

// Level 0 of the LK pyramid AND the ORB mosaic (k_orb_ext) in one pass over the larger of the two frames: both are the same equalised
// image under two border rules, and the mosaic's rule goes through the pyramid's own frame (non-isolated copyMakeBorder: reflect about the
// frame-grown image, then the frame pixel is the image pixel at its reflect-101 coordinate).  Requires pad <= LVK_ORB_BORDER (always:
// patch sizes above the ORB border are refused), so the pyramid's padded domain lies inside the mosaic's.  One launch and one read of the
// camera image instead of two on the frame's image stage.
template <bool EQ>
__global__ void k_level0_pad_ext(const uint8_t* __restrict__ src, int w, int h, int sstride,
                                 const uint8_t* __restrict__ lut, int tiles_x, int tiles_y, float inv_tw, float inv_th,
                                 uint8_t* __restrict__ dst /*padded base*/, int pad, int dstride, uint8_t* __restrict__ ext, int estride)
{
    const int B = LVK_ORB_BORDER;
    const int ex = blockIdx.x * blockDim.x + threadIdx.x, ey = blockIdx.y;
    if (ex >= w + 2 * B) return;
    const int gx = d_reflect101(ex - B + pad, w + 2 * pad) - pad, gy = d_reflect101(ey - B + pad, h + 2 * pad) - pad;     // grow == pad
    const int x = d_reflect101(gx, w), y = d_reflect101(gy, h);
    const int v = src[(size_t)y * sstride + x];
    const uint8_t o = EQ ? clahe_pixel(lut, tiles_x, tiles_y, inv_tw, inv_th, x, y, v) : (uint8_t)v;
    ext[(size_t)ey * estride + ex] = o;
    const int px = ex - B + pad, py = ey - B + pad;            // inside the pyramid's frame gx == ex - B: the same value
    if (px >= 0 && px < w + 2 * pad && py >= 0 && py < h + 2 * pad) dst[(size_t)py * dstride + px] = o;
}

// =========================================================================== pyrDown (+ frame) and Scharr planes
// [cv::pyrDown u8]  dst(x,y) = (sum_{5x5} w_i w_j src(2x-2+i, 2y-2+j) + 128) >> 8, w = [1 4 6 4 1].
// The source's own reflect-101 frame (pad >= 2) supplies the border taps, so no index math.
// [calcSharrDeriv]  Ix = [3 10 3]^T (x) [-1 0 1],  Iy = [-1 0 1]^T (x) [3 10 3], int16 interleaved.
// Reads the padded image (reflect-101 frame == calcSharrDeriv's own border rule).
// Scharr plane of level l and the (padded) image of level l+1 in ONE launch: both read only level l, and a launch on the
// frame's dependent chain costs ~7 us (kernel + inter-kernel barrier) whatever it computes.  Rows [0, h) of the grid are Scharr
// rows, rows [h, h + dh + 2 pad) are rows of the next level (none when dh == 0).
// pyrDown of one pixel of the NEXT level straight from this level (the 5 x 5 binomial, (sum + 128) >> 8; the caller has reflected x, y)
__device__ __forceinline__ int d_pyr_down_px(const uint8_t* __restrict__ s0, int sstride, int x, int y)
{
    int acc = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const uint8_t* r = s0 + (ptrdiff_t)(2 * y - 2 + j) * sstride + (2 * x - 2);
        int row = r[0] + r[4] + 4 * (r[1] + r[3]) + 6 * r[2];
        const int wj = (j == 0 || j == 4) ? 1 : (j == 2 ? 6 : 4);
        acc += wj * row;
    }
    return (acc + 128) >> 8;
}
// Rows of workgroups: [0, h) the Scharr plane of this level; [h, h + dh + 2 pad) the next level's padded image; and - when d1 is given
// (the next level is the LAST one: round 6, one launch less per frame) - [.., + dh) the NEXT level's Scharr plane too, each of its nine
// taps recomputed from this level (the next level's own pixels are being written by other workgroups of this very launch; the value of
// a tap, reflect-101 about the next level's frame included, is the same expression: the same bits).
__global__ void k_scharr_and_down(const uint8_t* __restrict__ s0, int w, int h, int sstride, int16_t* __restrict__ d0, int dstride,
                                  int dw, int dh, uint8_t* __restrict__ dst, int pad, int nstride, int16_t* __restrict__ d1, int dstride1)
{
    const int gx = blockIdx.x * blockDim.x + threadIdx.x;
    if ((int)blockIdx.y < h) {
        const int x = gx, y = blockIdx.y;
        if (x >= w) return;
        const uint8_t* r0 = s0 + (ptrdiff_t)(y - 1) * sstride + x;
        const uint8_t* r1 = r0 + sstride;
        const uint8_t* r2 = r1 + sstride;
        int a_m = (r0[-1] + r2[-1]) * 3 + r1[-1] * 10, a_p = (r0[1] + r2[1]) * 3 + r1[1] * 10;
        int b_m = r2[-1] - r0[-1], b_c = r2[0] - r0[0], b_p = r2[1] - r0[1];
        short2 o;
        o.x = (short)(a_p - a_m);
        o.y = (short)((b_p + b_m) * 3 + b_c * 10);
        *reinterpret_cast<short2*>(d0 + (size_t)y * dstride + 2 * x) = o;
    } else if ((int)blockIdx.y < h + dh + 2 * pad) {
        const int ex = gx, ey = blockIdx.y - h;
        if (ex >= dw + 2 * pad) return;
        int x = d_reflect101(ex - pad, dw), y = d_reflect101(ey - pad, dh);
        dst[(size_t)ey * nstride + ex] = (uint8_t)d_pyr_down_px(s0, sstride, x, y);
    } else {
        const int x = gx, y = blockIdx.y - (h + dh + 2 * pad);
        if (x >= dw) return;
        int v[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) v[j][i] = d_pyr_down_px(s0, sstride, d_reflect101(x - 1 + i, dw), d_reflect101(y - 1 + j, dh));
        int a_m = (v[0][0] + v[2][0]) * 3 + v[1][0] * 10, a_p = (v[0][2] + v[2][2]) * 3 + v[1][2] * 10;
        int b_m = v[2][0] - v[0][0], b_c = v[2][1] - v[0][1], b_p = v[2][2] - v[0][2];
        short2 o;
        o.x = (short)(a_p - a_m);
        o.y = (short)((b_p + b_m) * 3 + b_c * 10);
        *reinterpret_cast<short2*>(d1 + (size_t)y * dstride1 + 2 * x) = o;
    }
}

// =========================================================================== ORB mosaic + blur
// ext(e) = level0[ reflect about the frame-grown image ] (non-isolated copyMakeBorder, see oracle)
__global__ void k_orb_ext(const uint8_t* __restrict__ s0, int w, int h, int sstride, int grow,
                          uint8_t* __restrict__ ext, int estride)
{
    int ex = blockIdx.x * blockDim.x + threadIdx.x, ey = blockIdx.y;
    if (ex >= w + 2 * LVK_ORB_BORDER) return;
    int gx = d_reflect101(ex - LVK_ORB_BORDER + grow, w + 2 * grow) - grow;
    int gy = d_reflect101(ey - LVK_ORB_BORDER + grow, h + 2 * grow) - grow;
    ext[(size_t)ey * estride + ex] = s0[(ptrdiff_t)gy * sstride + gx];
}

// 7x7 fixed-point Gaussian {18,34,49,55,49,34,18}, (sum + 2^15) >> 16, interior only; frame copied.
// LDS-tiled separable: 64x16 output tile per 256-thread workgroup.
#define BL_TX 64
#define BL_TY 16
__global__ void __launch_bounds__(256) k_orb_blur(const uint8_t* __restrict__ ext, int w, int h, int estride,
                                                 uint8_t* __restrict__ blur)
{
    __shared__ uint8_t raw[BL_TY + 6][BL_TX + 8];
    __shared__ int hs[BL_TY + 6][BL_TX];
    const int B = LVK_ORB_BORDER;
    const int ew = w + 2 * B, eh = h + 2 * B;
    const int x0 = blockIdx.x * BL_TX, y0 = blockIdx.y * BL_TY;      // ext coordinates of the tile
    const int t = threadIdx.x;
    for (int i = t; i < (BL_TY + 6) * (BL_TX + 6); i += 256) {
        int ry = i / (BL_TX + 6), rx = i - ry * (BL_TX + 6);
        int gx = min(max(x0 + rx - 3, 0), ew - 1), gy = min(max(y0 + ry - 3, 0), eh - 1);
        raw[ry][rx] = ext[(size_t)gy * estride + gx];
    }
    __syncthreads();
    for (int i = t; i < (BL_TY + 6) * BL_TX; i += 256) {
        int ry = i / BL_TX, rx = i - ry * BL_TX;
        const uint8_t* r = &raw[ry][rx];
        hs[ry][rx] = 18 * (r[0] + r[6]) + 34 * (r[1] + r[5]) + 49 * (r[2] + r[4]) + 55 * r[3];
    }
    __syncthreads();
    for (int i = t; i < BL_TY * BL_TX; i += 256) {
        int ry = i / BL_TX, rx = i - ry * BL_TX;
        int gx = x0 + rx, gy = y0 + ry;
        if (gx >= ew || gy >= eh) continue;
        uint8_t o;
        if (gx >= B && gx < B + w && gy >= B && gy < B + h) {
            int acc = 18 * (hs[ry][rx] + hs[ry + 6][rx]) + 34 * (hs[ry + 1][rx] + hs[ry + 5][rx]) +
                      49 * (hs[ry + 2][rx] + hs[ry + 4][rx]) + 55 * hs[ry + 3][rx];
            o = d_sat_u8((acc + (1 << 15)) >> 16);
        } else {
            o = raw[ry + 3][rx + 3];
        }
        blur[(size_t)gy * estride + gx] = o;
    }
}

// The same pass for word-aligned mosaics (stride a multiple of 4, i.e. every supported camera width): 128x32 outputs per workgroup,
// the raw tile staged with 32-bit loads that are all in flight at once, four horizontally adjacent outputs per thread in both
// passes and one 32-bit store each.  Halo over-read 1.26x instead of 1.5x, a quarter of the load/store instructions.
#define BW_TX 128
#define BW_TY 32
__global__ void __launch_bounds__(256) k_orb_blur_w(const uint8_t* __restrict__ ext, int w, int h, int estride, uint8_t* __restrict__ blur)
{
    __shared__ unsigned raw[BW_TY + 6][BW_TX / 4 + 2];            // bytes x0-4 .. x0+131 of rows y0-3 .. y0+34
    __shared__ int hs[BW_TY + 6][BW_TX];
    const int B = LVK_ORB_BORDER;
    const int ew = w + 2 * B, eh = h + 2 * B;
    const int x0 = blockIdx.x * BW_TX, y0 = blockIdx.y * BW_TY;
    const int t = threadIdx.x;
    constexpr int WPR = BW_TX / 4 + 2, NW = (BW_TY + 6) * WPR, PER = (NW + 255) / 256;
    unsigned v[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int i = t + 256 * u;
        if (i < NW) {
            const int ry = i / WPR, rw = i - ry * WPR;
            const int gy = min(max(y0 + ry - 3, 0), eh - 1), gx = x0 - 4 + 4 * rw;
            // whole words inside the row are loaded as such; words that straddle an end of the row (only in the frame, whose outputs are
            // copies) are clamped to the nearest full word
            const int cx = min(max(gx, 0), ((ew - 4) & ~3));
            v[u] = *(const unsigned*)(ext + (size_t)gy * estride + cx);
        }
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) { const int i = t + 256 * u; if (i < NW) raw[i / WPR][i - (i / WPR) * WPR] = v[u]; }
    __syncthreads();
    // horizontal pass: outputs x0 + 4q .. +3 of a row need bytes x0 + 4q - 3 .. x0 + 4q + 6 = words q .. q+2 of the staged row (offset 1 byte)
    for (int i = t; i < (BW_TY + 6) * (BW_TX / 4); i += 256) {
        const int ry = i / (BW_TX / 4), q = i - ry * (BW_TX / 4);
        const unsigned a = raw[ry][q], b = raw[ry][q + 1], c = raw[ry][q + 2];
        int p[10];
        p[0] = (a >> 8) & 255; p[1] = (a >> 16) & 255; p[2] = a >> 24;
        p[3] = b & 255; p[4] = (b >> 8) & 255; p[5] = (b >> 16) & 255; p[6] = b >> 24;
        p[7] = c & 255; p[8] = (c >> 8) & 255; p[9] = (c >> 16) & 255;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            hs[ry][4 * q + k] = 18 * (p[k] + p[k + 6]) + 34 * (p[k + 1] + p[k + 5]) + 49 * (p[k + 2] + p[k + 4]) + 55 * p[k + 3];
    }
    __syncthreads();
    for (int i = t; i < BW_TY * (BW_TX / 4); i += 256) {
        const int ry = i / (BW_TX / 4), q = i - ry * (BW_TX / 4);
        const int gx = x0 + 4 * q, gy = y0 + ry;
        if (gx >= ew || gy >= eh) continue;
        const unsigned centre = raw[ry + 3][q + 1];
        unsigned out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = gx + k, rx = 4 * q + k;
            unsigned o;
            if (x >= B && x < B + w && gy >= B && gy < B + h) {
                const int acc = 18 * (hs[ry][rx] + hs[ry + 6][rx]) + 34 * (hs[ry + 1][rx] + hs[ry + 5][rx]) +
                                49 * (hs[ry + 2][rx] + hs[ry + 4][rx]) + 55 * hs[ry + 3][rx];
                o = d_sat_u8((acc + (1 << 15)) >> 16);
            } else o = (centre >> (8 * k)) & 255u;
            out |= o << (8 * k);
        }
        *(unsigned*)(blur + (size_t)gy * estride + gx) = out;         // ew is a multiple of 4 here: the word never leaves the row
    }
}

// =========================================================================== GFTT
// min-eigenvalue response [cornerMinEigenVal, Sobel 3, block 3] with the oracle's fixed float order.
#define EG_TX 64
#define EG_TY 8
__device__ __forceinline__ void cov_at(const uint8_t* __restrict__ s0, int sstride, int x, int y, float ke, float kc,
                                       float& c0, float& c1, float& c2)
{
    const uint8_t* r0 = s0 + (ptrdiff_t)(y - 1) * sstride + x;
    const uint8_t* r1 = r0 + sstride;
    const uint8_t* r2 = r1 + sstride;
    float a00 = r0[-1], a01 = r0[0], a02 = r0[1], a10 = r1[-1], a12 = r1[1], a20 = r2[-1], a21 = r2[0], a22 = r2[1];
    float q0 = a02 - a00, q1 = a12 - a10, q2 = a22 - a20;
    float dx = kc * q1 + ke * (q0 + q2);
    float t0 = kc * a01 + ke * (a00 + a02);
    float t2 = kc * a21 + ke * (a20 + a22);
    float dy = t2 - t0;
    c0 = dx * dx; c1 = dx * dy; c2 = dy * dy;
}

__global__ void __launch_bounds__(256) k_min_eigen(const uint8_t* __restrict__ s0, int w, int h, int sstride,
                                                  float* __restrict__ eig)
{
    __shared__ float cv[3][EG_TY + 2][EG_TX + 2];
    __shared__ float hs[3][EG_TY + 2][EG_TX];
    const float ke = (float)(1.0 / 3060.0), kc = 2.0f * ke;
    const int x0 = blockIdx.x * EG_TX, y0 = blockIdx.y * EG_TY, t = threadIdx.x;
    for (int i = t; i < (EG_TY + 2) * (EG_TX + 2); i += 256) {
        int ry = i / (EG_TX + 2), rx = i - ry * (EG_TX + 2);
        int gx = x0 + rx - 1, gy = y0 + ry - 1;
        // box filter border = reflect-101 of the covariance map; clamp tile overhang to a valid pixel
        gx = d_reflect101(min(gx, w), w); gy = d_reflect101(min(gy, h), h);
        float c0, c1, c2;
        cov_at(s0, sstride, gx, gy, ke, kc, c0, c1, c2);
        cv[0][ry][rx] = c0; cv[1][ry][rx] = c1; cv[2][ry][rx] = c2;
    }
    __syncthreads();
    for (int i = t; i < (EG_TY + 2) * EG_TX; i += 256) {
        int ry = i / EG_TX, rx = i - ry * EG_TX;
#pragma unroll
        for (int c = 0; c < 3; ++c) hs[c][ry][rx] = (cv[c][ry][rx] + cv[c][ry][rx + 1]) + cv[c][ry][rx + 2];
    }
    __syncthreads();
    for (int i = t; i < EG_TY * EG_TX; i += 256) {
        int ry = i / EG_TX, rx = i - ry * EG_TX;
        int gx = x0 + rx, gy = y0 + ry;
        if (gx >= w || gy >= h) continue;
        float s[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) s[c] = (hs[c][ry][rx] + hs[c][ry + 1][rx]) + hs[c][ry + 2][rx];
        float a = s[0] * 0.5f, b = s[1], c2 = s[2] * 0.5f;
        eig[(size_t)gy * w + gx] = (a + c2) - sqrtf((a - c2) * (a - c2) + b * b);
    }
}

__device__ __forceinline__ unsigned f2ord(float f) { unsigned b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned k) { return k == 0 ? 0.f : __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// gftt scratch layout (uints): [0] max key, [1] n candidates, [2] overflow flag, [3] spare, [GF_HIST_OFF ...) strength histogram
#define GF_HIST_BITS 13
#define GF_HIST_OFF 4
#define GF_SCRATCH_UINTS 4
__global__ void __launch_bounds__(256) k_masked_max(const float* __restrict__ eig, const uint8_t* __restrict__ mask, int n,
                                                   unsigned* __restrict__ scratch)
{
    unsigned best = 0;
    // four pixels per lane-load (16-byte response values, 4-byte mask words): the pass is a pure stream (12 MB at 1080p)
    // a caller's mask (lvk_good_features) need not be word-aligned: then everything takes the scalar tail below
    const bool vec = (((size_t)eig & 15) | ((size_t)mask & 3)) == 0;
    const int n4 = vec ? n >> 2 : 0;
    const float4* e4 = (const float4*)eig; const unsigned* m4 = (const unsigned*)mask;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const float4 v = e4[i]; const unsigned mk = mask ? m4[i] : 0xFFFFFFFFu;
        if (mk & 0x000000FFu) best = max(best, f2ord(v.x));
        if (mk & 0x0000FF00u) best = max(best, f2ord(v.y));
        if (mk & 0x00FF0000u) best = max(best, f2ord(v.z));
        if (mk & 0xFF000000u) best = max(best, f2ord(v.w));
    }
    for (int i = (n4 << 2) + blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        if (!mask || mask[i]) best = max(best, f2ord(eig[i]));
    for (int o = 32; o > 0; o >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, o));
    __shared__ unsigned wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) { best = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])); if (best) atomicMax(&scratch[0], best); }
}

// The frame path's detection mask and the masked maximum of the response map in ONE pass (findNewFeaturesToBeTracked's mask, image_processor.cpp:1005-1030,
// and goodFeaturesToTrack's minMaxLoc under it): a workgroup owns MM_ROWS image rows, builds their mask in LDS - all 255, then the row
// segments of every (2 md + 1)^2 box around round(pt) that reaches into the strip, zeroed -, writes it out for k_gftt_candidates and takes the
// maximum of the response values it leaves visible (one atomicMax per workgroup).  Same bytes as a 255-filled image with the boxes cut out (rounds 1-4: a fill and a kernel of one workgroup per box), same
// key as k_masked_max; one launch and one kernel boundary instead of a fill and two launches on the chain commit -> detection -> the next
// frame's new-point tracking.  The strip's response values are requested before the mask is built (they do not depend on it).
#define MM_ROWS 4
#define MM_PRE 8
__global__ void __launch_bounds__(256) k_mask_max(const lvk_pt2f* __restrict__ pts, const int* __restrict__ n_pts, int w, int h, int md,
                                                 const float* __restrict__ eig, uint8_t* __restrict__ mask, unsigned* __restrict__ scratch)
{
    extern __shared__ unsigned mm_sh[];                     // MM_ROWS x wp bytes, wp = w rounded up to 4
    uint8_t* mb = (uint8_t*)mm_sh;
    const int t = threadIdx.x, y0 = blockIdx.x * MM_ROWS, rows = min(MM_ROWS, h - y0);
    const int wq = (w + 3) >> 2, wp = wq * 4;
    const bool vec = (w & 3) == 0 && ((((size_t)eig) & 15) | (((size_t)mask) & 3)) == 0;
    const int nq = rows * wq;                               // 4-pixel groups of the strip
    float4 pre[MM_PRE];
    const bool use_pre = vec && nq <= MM_PRE * 256;
    if (use_pre) {
#pragma unroll
        for (int u = 0; u < MM_PRE; ++u) { const int i = t + 256 * u; if (i < nq) { const int r = i / wq, q = i - r * wq; pre[u] = ((const float4*)(eig + (size_t)(y0 + r) * w))[q]; } }
    }
    for (int i = t; i < MM_ROWS * wq; i += 256) mm_sh[i] = 0xFFFFFFFFu;
    __syncthreads();
    const int n = *n_pts;
    for (int i = t; i < n; i += 256) {
        // round(): half away from zero on the float coordinate
        const int ry = (int)roundf(pts[i].y), rx = (int)roundf(pts[i].x);
        const int r0 = max(max(ry - md, 0), y0), r1 = min(min(ry + md, h - 1), y0 + rows - 1), c0 = max(rx - md, 0), c1 = min(rx + md, w - 1);
        for (int y = r0; y <= r1; ++y) for (int x = c0; x <= c1; ++x) mb[(y - y0) * wp + x] = 0;
    }
    __syncthreads();
    unsigned best = 0;
    if (vec) {
#pragma unroll
        for (int u = 0; u < MM_PRE; ++u) {
            const int i = t + 256 * u;
            if (!use_pre || i >= nq) break;
            const int r = i / wq, q = i - r * wq;
            const unsigned mk = mm_sh[r * wq + q];
            ((unsigned*)(mask + (size_t)(y0 + r) * w))[q] = mk;
            const float4 v = pre[u];
            if (mk & 0x000000FFu) best = max(best, f2ord(v.x));
            if (mk & 0x0000FF00u) best = max(best, f2ord(v.y));
            if (mk & 0x00FF0000u) best = max(best, f2ord(v.z));
            if (mk & 0xFF000000u) best = max(best, f2ord(v.w));
        }
        if (!use_pre)
            for (int i = t; i < nq; i += 256) {
                const int r = i / wq, q = i - r * wq;
                const unsigned mk = mm_sh[r * wq + q];
                ((unsigned*)(mask + (size_t)(y0 + r) * w))[q] = mk;
                const float4 v = ((const float4*)(eig + (size_t)(y0 + r) * w))[q];
                if (mk & 0x000000FFu) best = max(best, f2ord(v.x));
                if (mk & 0x0000FF00u) best = max(best, f2ord(v.y));
                if (mk & 0x00FF0000u) best = max(best, f2ord(v.z));
                if (mk & 0xFF000000u) best = max(best, f2ord(v.w));
            }
    } else {
        for (int i = t; i < rows * w; i += 256) {
            const int r = i / w, x = i - r * w;
            const uint8_t b = mb[r * wp + x];
            mask[(size_t)(y0 + r) * w + x] = b;
            if (b) best = max(best, f2ord(eig[(size_t)(y0 + r) * w + x]));
        }
    }
    for (int o = 32; o > 0; o >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, o));
    __shared__ unsigned wmax[4];
    if ((t & 63) == 0) wmax[t >> 6] = best;
    __syncthreads();
    if (t == 0) { best = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])); if (best) atomicMax(&scratch[0], best); }
}

// 3x3 non-maximum suppression above quality*max, inside the mask.  One workgroup scans 256 columns x GC_ROWS rows, collects its
// candidates in LDS and reserves space in the global list with ONE atomic (a per-wavefront atomic on a single counter serialises
// ~2,000 L2 atomics per frame: 35 us).
#define GC_ROWS 8
__global__ void __launch_bounds__(256) k_gftt_candidates(const float* __restrict__ eig, const uint8_t* __restrict__ mask, int w, int h,
                                                        float quality, unsigned* __restrict__ scratch,
                                                        unsigned long long* __restrict__ cands, int cap)
{
    __shared__ unsigned long long loc[256 * GC_ROWS / 2];
    __shared__ unsigned cnt, base;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const int x = blockIdx.x * 256 + threadIdx.x + 1;
    const int y0 = blockIdx.y * GC_ROWS + 1;
    // every value a thread will look at - its column and the two next to it over GC_ROWS + 2 rows, its mask bytes - is asked for
    // before the first one is used: the map was written by the previous kernel on other XCDs, and a row loop that loads, tests and
    // only then loads the neighbours was three dependent round trips per row (11.5 us for a 361 KB image)
    float e[GC_ROWS + 2][3]; uint8_t mk[GC_ROWS];
    const bool xin = x < w - 1;
#pragma unroll
    for (int r = 0; r < GC_ROWS + 2; ++r) {
        const int yy = y0 - 1 + r;
        const bool ok = xin && yy < h;
        const float* row = eig + (size_t)yy * w + x;
        e[r][0] = ok ? row[-1] : 0.f; e[r][1] = ok ? row[0] : 0.f; e[r][2] = ok ? row[1] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < GC_ROWS; ++r) { const int y = y0 + r; mk[r] = (mask && xin && y < h - 1) ? mask[(size_t)y * w + x] : (uint8_t)1; }
    const float max_val = ord2f(scratch[0]);
    const float thresh = (float)((double)max_val * (double)quality);
#pragma unroll
    for (int ry = 0; ry < GC_ROWS; ++ry) {
        const int y = y0 + ry;
        if (x >= w - 1 || y >= h - 1) continue;
        float v = e[ry + 1][1];
        v = v > thresh ? v : 0.f;
        if (v == 0.f) continue;
        if (mask && !mk[ry]) continue;
        float m = v;
#pragma unroll
        for (int dy = 0; dy <= 2; ++dy)
#pragma unroll
            for (int dx = 0; dx <= 2; ++dx) {
                float u = e[ry + dy][dx];
                u = u > thresh ? u : 0.f;
                m = fmaxf(m, u);
            }
        if (v != m) continue;
        const unsigned slot = atomicAdd(&cnt, 1u);
        if (slot < 256 * GC_ROWS / 2) loc[slot] = ((unsigned long long)f2ord(v) << 32) | (unsigned)(y * w + x);
    }
    __syncthreads();
    const unsigned n_loc = min(cnt, (unsigned)(256 * GC_ROWS / 2));
    if (threadIdx.x == 0) { base = n_loc ? atomicAdd(&scratch[1], n_loc) : 0u; if (cnt > n_loc) scratch[2] = 1u; }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < n_loc; i += 256) { const unsigned slot = base + i; if (slot < (unsigned)cap) cands[slot] = loc[i]; else scratch[2] = 1u; }
}

// single workgroup: the greedy min-distance pass of goodFeaturesToTrack ("visit candidates by descending strength, accept
// one iff no accepted corner is closer than minDistance") WITHOUT sorting all candidates.  3x3 NMS leaves ~1/9 of the
// unmasked pixels as candidates (15-40 K at 752x480) but almost all of them are rejected by the first few hundred accepted
// corners.  So candidates are visited in strength BUCKETS (boundaries from the histogram the candidate kernel filled; sizes
// 1 K, 2 K, 4 K, then <= 8 K): every thread tests the bucket's candidates against the accepted grid (cell = minDistance, so
// conflicts live in the 3x3 cells around a candidate, <= 4 corners per cell), only the SURVIVORS are sorted (bitonic, LDS)
// and wave 0 resolves them in order exactly like the sequential rule.  Rejections by the grid are final because the grid
// only grows; buckets are visited in descending strength, survivors in descending (strength, index): identical result.
#define GF_MAX_OUT 4096
#define GF_SURV 8192
__device__ __forceinline__ bool gf_grid_conflict(const unsigned short (*cells)[4], const short2* acc, int gw, int gh, int cell, float md2, int x, int y)
{
    const int xc = x / cell, yc = y / cell;
    const int x1 = max(xc - 1, 0), y1 = max(yc - 1, 0), x2 = min(xc + 1, gw - 1), y2 = min(yc + 1, gh - 1);
    for (int yy = y1; yy <= y2; ++yy)
        for (int xx = x1; xx <= x2; ++xx)
            for (int sl = 0; sl < 4; ++sl) {
                const unsigned short a = cells[yy * gw + xx][sl];
                if (a == 0xFFFF) break;
                const float dx = (float)x - (float)acc[a].x, dy = (float)y - (float)acc[a].y;
                if (dx * dx + dy * dy < md2) return true;
            }
    return false;
}

#ifdef LVK_GF_TIMING
static __device__ unsigned long long g_gf_tick[32];
#define GF_TICK(k) do { if (threadIdx.x == 0) g_gf_tick[k] = wall_clock64(); } while (0)
extern "C" void lvk_debug_gf_ticks(unsigned long long* out) { hipDeviceSynchronize(); hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gf_tick), sizeof(unsigned long long) * 32); }
#else
#define GF_TICK(k) do { } while (0)
#endif
__global__ void __launch_bounds__(1024) k_gftt_select(const unsigned long long* __restrict__ cands, int cap, int w, int h,
                                                     int max_corners, int cell, float md2, int surv_cap /* GF_SURV, or half of it when the grid is large */, int acc_cap,
                                                     unsigned* __restrict__ scratch /* read, then left zeroed for the next detection (no fill launch) */, lvk_pt2f* __restrict__ out, int out_cap,
                                                     int* __restrict__ n_out, const int* __restrict__ d_sub)
{
    extern __shared__ unsigned long long gf_sh[];           // surv [surv_cap] u64 | cells [gw*gh][4] u16 | acc [acc_cap] short2
    const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
    unsigned long long* surv = gf_sh;
    unsigned short (*cells)[4] = reinterpret_cast<unsigned short (*)[4]>(gf_sh + surv_cap);
    short2* acc = reinterpret_cast<short2*>(gf_sh + surv_cap + gw * gh);
    __shared__ unsigned coarse[1024];                        // histogram folded to 1024 groups (8 bins each at 13 bits)
    __shared__ unsigned hist[1 << GF_HIST_BITS];
    __shared__ int sh_ns, sh_na, sh_done, sh_lo, sh_hi;
    unsigned* scan = reinterpret_cast<unsigned*>(surv);      // the survivor buffer is idle while the next bucket is being chosen
    const int t = threadIdx.x, lane = t & 63;
    if (d_sub) {   // image_processor.cpp:1034-1036: maxCorners = max_features_num - curr_pts_.size(), skipped when 0
        max_corners -= *d_sub;
        if (max_corners <= 0) { if (t == 0) { *n_out = 0; for (int q = 0; q < GF_SCRATCH_UINTS; ++q) scratch[q] = 0u; } return; }
    }
    GF_TICK(0);
    const int n = (int)min(scratch[1], (unsigned)cap);
    constexpr int GROUP = (1 << GF_HIST_BITS) / 1024;
    // strength histogram of the candidates (LDS atomics; one pass over the candidate keys)
    for (int q = 0; q < GROUP; ++q) hist[t * GROUP + q] = 0u;
    for (int i = t; i < gw * gh; i += 1024) { cells[i][0] = cells[i][1] = cells[i][2] = cells[i][3] = 0xFFFF; }
    if (t == 0) { sh_na = 0; sh_done = 0; sh_hi = 1 << GF_HIST_BITS; }
    __syncthreads();
    // the candidate keys were written by the previous kernel on other XCDs: 8 loads in flight per thread, not one trip per key
    for (int i0 = t; i0 < n; i0 += 1024 * 8) {
        unsigned long long kv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + 1024 * u; kv[u] = i < n ? cands[i] : 0ull; }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (i0 + 1024 * u < n) atomicAdd(&hist[(unsigned)(kv[u] >> (64 - GF_HIST_BITS))], 1u);
    }
    __syncthreads();
    GF_TICK(1);
    {
        unsigned s = 0;
        for (int q = 0; q < GROUP; ++q) s += hist[t * GROUP + q];
        coarse[t] = s;
    }
    __syncthreads();
    GF_TICK(2);
    // first bucket: a few times the corners still wanted (the publish-frame case asks for 10-50), later buckets double
    int target = 1024;
    if (max_corners > 0) { target = 64; while (target < 4 * max_corners && target < 1024) target <<= 1; }   // (any bucket boundaries give the same corners)
    for (int bucket = 0; bucket < 4096; ++bucket) {
        // next bucket = whole histogram groups [lo, hi) below the current top holding >= target candidates (or all that is left):
        // suffix sums of the 1024 group counts by a block scan, then ONE thread-parallel boundary test - a serial walk over
        // thousands of mostly empty bins by one thread was 26 us.  (Any bucket boundaries give the same corners.)
        bool serial_walk = (sh_hi % GROUP) != 0;           // only after a bin-granular bucket (see below)
        if (!serial_walk) {
            const int hg = sh_hi / GROUP;                   // groups >= hg are done
            scan[t] = t < hg ? coarse[t] : 0u;
            __syncthreads();
            for (int o = 1; o < 1024; o <<= 1) {            // suffix sum: scan[g] = candidates in groups [g, hg)
                const unsigned add = t + o < 1024 ? scan[t + o] : 0u;
                __syncthreads();
                scan[t] += add;
                __syncthreads();
            }
            if (t == 0) { sh_lo = -1; sh_ns = 0; }
            __syncthreads();
            const unsigned mine = scan[t], above = t + 1 < 1024 ? scan[t + 1] : 0u;
            if (t < hg && mine >= (unsigned)target && above < (unsigned)target) sh_lo = t * GROUP;       // exactly one such group, if any
            __syncthreads();
            if (t == 0 && sh_lo < 0) sh_lo = scan[0] > 0 ? 0 : sh_hi;                                     // fewer than target left: all of it
            __syncthreads();
            serial_walk = sh_lo < sh_hi && scan[sh_lo / GROUP] > (unsigned)surv_cap;
            __syncthreads();
        }
        if (serial_walk && t == 0) {
            // the group-aligned bucket would overflow the survivor buffer (pathological strength distribution): bin-by-bin walk
            int hi = sh_hi, lo = hi; unsigned cnt = 0;
            sh_ns = 0;
            while (lo > 0) {
                if ((lo & (GROUP - 1)) == 0 && coarse[lo / GROUP - 1] == 0) { lo -= GROUP; continue; }     // empty group: skip
                const unsigned c = hist[lo - 1];
                if (cnt > 0 && cnt + c > (unsigned)target) break;
                cnt += c; --lo;
                if (cnt >= (unsigned)target) break;
            }
            sh_lo = lo;
            if (cnt > (unsigned)surv_cap) sh_done = 2;                 // one histogram bin alone exceeds the survivor buffer
        }
        __syncthreads();
        if (bucket == 0) GF_TICK(3);
        const int lo = sh_lo, hi = sh_hi;
        if (sh_done == 2) break;
        if (hi == 0 || lo == hi) break;
        const unsigned klo = (unsigned)lo << (32 - GF_HIST_BITS), khi_excl = hi >= (1 << GF_HIST_BITS) ? 0xFFFFFFFFu : ((unsigned)hi << (32 - GF_HIST_BITS));
        for (int i0 = t; i0 < n; i0 += 1024 * 8) {
            unsigned long long kv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + 1024 * u; kv[u] = i < n ? cands[i] : 0ull; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned long long k = kv[u];
                const unsigned sv = (unsigned)(k >> 32);
                const bool keep = i0 + 1024 * u < n && !(sv < klo || (hi < (1 << GF_HIST_BITS) && sv >= khi_excl));
                // one LDS atomic per wavefront (a single-address atomic per survivor serialised: ~1000 x 16 ns)
                const unsigned long long mk = __ballot(keep);
                if (mk) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&sh_ns, __popcll(mk));
                    base = __shfl(base, 0);
                    if (keep) { const int slot = base + __popcll(mk & ((1ull << lane) - 1ull)); if (slot < surv_cap) surv[slot] = k; }
                }
            }
        }
        __syncthreads();
        if (bucket == 0) GF_TICK(4);
        const int ns = min(sh_ns, surv_cap);
        // candidates of this bucket that an already accepted corner rules out are dropped BEFORE the sort (key 0 sorts last and ends
        // the greedy pass); one candidate per thread, so the grid walk is paid once, not once per diverged lane group.  The grid
        // is empty in the first bucket.
        if (sh_na > 0) {
            for (int i = t; i < ns; i += 1024) {
                const unsigned idx = (unsigned)(surv[i] & 0xFFFFFFFFull);
                const int y = idx / w, x = idx - y * w;
                if (gf_grid_conflict(cells, acc, gw, gh, cell, md2, x, y)) surv[i] = 0ull;
            }
            __syncthreads();
        }
        if (ns <= 512) {
            // rank sort, 1024 / P threads per key (P = ns rounded up to a power of two): rank = number of larger keys (keys are unique:
            // the pixel index is in the low word).  The usual first bucket of a publish frame holds 100-200 keys: 4-8 threads per key,
            // 16-32 comparisons each (two threads per key and 256 comparisons each were 13 of the kernel's 28 us)
            unsigned* rank = reinterpret_cast<unsigned*>(surv + 1024);
            unsigned long long* sorted = surv + 2048;
            // sorted[] is cleared too: keys the grid test above set to 0 all get the SAME rank (the number of keys left), so the slots behind
            // the first 0 are written by nobody - they held the previous bucket's keys (or whatever an earlier kernel left in LDS), and the
            // greedy pass below took them for survivors: a corner of an earlier bucket accepted twice, or a "pixel index" that is no pixel
            // (found by the whole-program fuzz, case 19: a new point at y = 1.9e6 and a memory fault in the LK kernel that read there)
            if (t < 512) { rank[t] = 0u; sorted[t] = 0ull; }
            __syncthreads();
            int P = 64; while (P < ns) P <<= 1;
            const int parts = 1024 / P, e = t & (P - 1), part = t / P, span = (ns + parts - 1) / parts;
            if (e < ns) {
                const unsigned long long mine = surv[e];
                const int j0 = part * span, j1 = min(ns, j0 + span);
                unsigned c = 0;
#pragma unroll 4
                for (int j = j0; j < j1; ++j) c += surv[j] > mine;
                if (c) atomicAdd(&rank[e], c);
            }
            __syncthreads();
            if (t < ns) sorted[rank[t]] = surv[t];
            __syncthreads();
            if (t < ns) surv[t] = sorted[t];
            __syncthreads();
        } else {
            int np2 = 1; while (np2 < ns) np2 <<= 1;
            for (int i = ns + t; i < np2; i += 1024) surv[i] = 0ull;
            __syncthreads();
            for (int k = 2; k <= np2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = t; i < np2; i += 1024) {
                        const int l = i ^ j;
                        if (l > i) {
                            const unsigned long long a = surv[i], b = surv[l];
                            const bool desc = (i & k) == 0;
                            if (desc ? a < b : a > b) { surv[i] = b; surv[l] = a; }
                        }
                    }
                    __syncthreads();
                }
        }
        if (bucket == 0) GF_TICK(5);
        if (t < 64) {
            // wavefront 0 resolves the sorted survivors in order, 64 at a time.  Inside a batch the loop only DECIDES (one iteration per
            // accepted corner: its coordinates are broadcast from registers, the lanes behind it that it rules out drop out); the
            // batch's accepted corners are then written - output list, accepted list, grid cell - by their own lanes in parallel
            // (a one-lane section with two integer divisions and a dependent LDS slot search per corner was 7.6 of the kernel's 28 us)
            int na = sh_na;
            bool done = false;
            for (int sb = 0; sb < ns && !done; sb += 64) {
                const int si = sb + lane;
                bool g = si < ns && surv[si] != 0ull;      // 0 = ruled out before the sort
                int sx = 0, sy = 0;
                if (g) {
                    const unsigned idx = (unsigned)(surv[si] & 0xFFFFFFFFull);
                    sy = idx / w; sx = idx - sy * w;
                    if (sb > 0 && gf_grid_conflict(cells, acc, gw, gh, cell, md2, sx, sy)) g = false;   // corners accepted earlier in this bucket
                }
                unsigned long long mm = __ballot(g), accm = 0ull;
                const int na0 = na;
                while (mm) {
                    const int j = __builtin_amdgcn_readfirstlane(__ffsll((long long)mm) - 1);
                    const int xj = __builtin_amdgcn_readlane(sx, j), yj = __builtin_amdgcn_readlane(sy, j);
                    accm |= 1ull << j;
                    ++na;
                    if ((max_corners > 0 && na == max_corners) || na >= acc_cap) { done = true; break; }
                    if (g && lane > j) {
                        const float dx = (float)sx - (float)xj, dy = (float)sy - (float)yj;
                        if (dx * dx + dy * dy < md2) g = false;
                    }
                    mm = __ballot(g && lane > j);
                }
                if ((accm >> lane) & 1ull) {
                    const int r = na0 + __popcll(accm & ((1ull << lane) - 1ull));
                    if (r < out_cap) { out[r].x = (float)sx; out[r].y = (float)sy; }
                    if (r < acc_cap) {
                        acc[r] = make_short2((short)sx, (short)sy);
                        // first free slot of the corner's cell (four 16-bit slots = one 64-bit word; slot order is irrelevant to the
                        // conflict test; a full cell drops the entry, as before)
                        unsigned long long* c64 = reinterpret_cast<unsigned long long*>(cells) + ((sy / cell) * gw + (sx / cell));
                        unsigned long long old = *c64;
                        for (;;) {
                            int sl = -1;
                            for (int q = 0; q < 4; ++q) if (((old >> (16 * q)) & 0xFFFFull) == 0xFFFFull) { sl = q; break; }
                            if (sl < 0) break;
                            const unsigned long long nw = (old & ~(0xFFFFull << (16 * sl))) | ((unsigned long long)(unsigned)r << (16 * sl));
                            const unsigned long long prev = atomicCAS(c64, old, nw);
                            if (prev == old) break;
                            old = prev;
                        }
                    }
                }
                __threadfence_block();                      // the next batch's grid test reads what this one inserted
            }
            if (lane == 0) { sh_na = na; if (done) sh_done = 1; sh_hi = lo; }
        }
        __syncthreads();
        if (bucket == 0) GF_TICK(6);
        if (sh_done) break;
        if (target < surv_cap) target <<= 1;
    }
    GF_TICK(7);
    if (t == 0) { *n_out = sh_na < out_cap ? sh_na : out_cap; for (int q = 0; q < GF_SCRATCH_UINTS; ++q) scratch[q] = 0u; }   // every thread read scratch[1] many barriers ago
}

// =========================================================================== host side of the ABI
extern "C" {

lvk_status lvk_clahe_u8(lvk_context* ctx, const uint8_t* d_src, int w, int h, int sstride,
                        uint8_t* d_dst, int dstride, double clip_limit, int tiles_x, int tiles_y)
{
    if (!ctx || !d_src || !d_dst || w <= 0 || h <= 0 || tiles_x <= 0 || tiles_y <= 0) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_clahe_u8: bad argument");
    int ew = w, eh = h;
    if (!(w % tiles_x == 0 && h % tiles_y == 0)) { ew = w + (tiles_x - (w % tiles_x)); eh = h + (tiles_y - (h % tiles_y)); }
    const int tw = ew / tiles_x, th = eh / tiles_y, total = tw * th;
    const float lut_scale = (float)(255) / total;
    int clip = 0;
    if (clip_limit > 0.0) { clip = (int)(clip_limit * total / 256); if (clip < 1) clip = 1; }
    const size_t nt = (size_t)tiles_x * tiles_y;
    uint8_t* lut = (uint8_t*)lvk_ctx_scratch(ctx, 0, nt * 256);
    if (!lut) return lvk_set_error(ctx, LVK_ERR_DEVICE, "scratch allocation failed");
    int vec4; clahe_launch_shape(d_src, w, h, sstride, tw, th, tiles_x, tiles_y, &vec4);
    hipLaunchKernelGGL(k_clahe_lut, dim3(tiles_x * tiles_y), dim3(256), 0, ctx->stream, d_src, w, h, sstride, tw, th, tiles_x, clip, lut_scale, lut, vec4);
    hipLaunchKernelGGL(k_clahe_apply, dim3((w + 255) / 256, h), dim3(256), 0, ctx->stream, d_src, w, h, sstride, lut, tiles_x, tiles_y,
                       1.0f / tw, 1.0f / th, d_dst, dstride);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

lvk_status lvk_pyramid_create(lvk_context* ctx, int w, int h, int win, int max_level, lvk_pyramid** out)
{
    if (!ctx || !out || w <= 0 || h <= 0 || win < 3 || max_level < 0) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_pyramid_create: bad argument");
    if (win > LVK_ORB_BORDER) return lvk_set_error(ctx, LVK_ERR_UNSUPPORTED, "patch_size %d > %d not supported", win, LVK_ORB_BORDER);
    lvk_pyramid* p = new lvk_pyramid();
    memset(p, 0, sizeof *p);
    p->ctx = ctx; p->pad = win; p->max_level = max_level;
    int lw = w, lh = h;
    for (int level = 0; level <= max_level && level < LVK_MAX_LEVELS; ++level) {
        p->w[level] = lw; p->h[level] = lh;
        p->istride[level] = (lw + 2 * win + 63) & ~63;
        p->dstride[level] = 2 * p->istride[level];
        size_t ib = (size_t)p->istride[level] * (lh + 2 * win), db = (size_t)p->dstride[level] * (lh + 2 * win) * sizeof(int16_t);
        if (hipMalloc((void**)&p->img[level], ib) != hipSuccess || hipMalloc((void**)&p->der[level], db) != hipSuccess) {
            lvk_pyramid_destroy(p);
            return lvk_set_error(ctx, LVK_ERR_DEVICE, "lvk_pyramid_create: hipMalloc failed");
        }
        hipMemsetAsync(p->img[level], 0, ib, ctx->stream);
        hipMemsetAsync(p->der[level], 0, db, ctx->stream);     // BORDER_CONSTANT frame of the derivative planes stays 0
        p->n_levels = level + 1;
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
        if (lw <= win || lh <= win) break;                     // buildOpticalFlowPyramid stop rule
    }
    p->clahe_lut_cap = 64 * 256;
    if (hipMalloc((void**)&p->clahe_lut, (size_t)p->clahe_lut_cap) != hipSuccess) { lvk_pyramid_destroy(p); return lvk_set_error(ctx, LVK_ERR_DEVICE, "hipMalloc lut"); }
    *out = p;
    return LVK_OK;
}

void lvk_pyramid_destroy(lvk_pyramid* p)
{
    if (!p) return;
    for (int i = 0; i < LVK_MAX_LEVELS; ++i) { if (p->img[i]) hipFree(p->img[i]); if (p->der[i]) hipFree(p->der[i]); }
    if (p->clahe_lut) hipFree(p->clahe_lut);
    delete p;
}

static lvk_status build_levels(lvk_context* ctx, lvk_pyramid* p)
{
    if (p->ev_level0) hipEventRecord(p->ev_level0, ctx->stream);
    // per level: its Scharr plane and the next level's image in one launch (both depend on level l only)
    for (int l = 0; l < p->n_levels; ++l) {
        const uint8_t* s0 = p->img[l] + (size_t)p->pad * p->istride[l] + p->pad;
        int16_t* d0 = p->der[l] + (size_t)p->pad * p->dstride[l] + 2 * p->pad;
        const bool next = l + 1 < p->n_levels;
        const bool last2 = next && l + 2 == p->n_levels;     // the next level is the last: its Scharr plane rides in this launch
        const int dw = next ? p->w[l + 1] : 0, dh = next ? p->h[l + 1] : 0;
        const int gx = std::max((p->w[l] + 255) / 256, next ? (dw + 2 * p->pad + 255) / 256 : 0);
        int16_t* d1 = last2 ? p->der[l + 1] + (size_t)p->pad * p->dstride[l + 1] + 2 * p->pad : nullptr;
        hipLaunchKernelGGL(k_scharr_and_down, dim3(gx, p->h[l] + (next ? dh + 2 * p->pad : 0) + (last2 ? dh : 0)), dim3(256), 0, ctx->stream,
                           s0, p->w[l], p->h[l], p->istride[l], d0, p->dstride[l], dw, dh, next ? p->img[l + 1] : nullptr, p->pad, next ? p->istride[l + 1] : 0,
                           d1, last2 ? p->dstride[l + 1] : 0);
        if (last2) break;
    }
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

lvk_status lvk_pyramid_build(lvk_context* ctx, lvk_pyramid* p, const uint8_t* d_img, int stride)
{
    if (!ctx || !p || !d_img) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_pyramid_build: bad argument");
    hipLaunchKernelGGL(k_level0_pad<false>, dim3((p->w[0] + 2 * p->pad + 255) / 256, p->h[0] + 2 * p->pad), dim3(256), 0, ctx->stream,
                       d_img, p->w[0], p->h[0], stride, (const uint8_t*)nullptr, 1, 1, 1.f, 1.f, p->img[0], p->pad, p->istride[0]);
    return build_levels(ctx, p);
}

lvk_status lvk_pyramid_build_clahe(lvk_context* ctx, lvk_pyramid* p, const uint8_t* d_img, int stride,
                                   double clip_limit, int tiles_x, int tiles_y)
{
    if (!ctx || !p || !d_img || tiles_x <= 0 || tiles_y <= 0) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_pyramid_build_clahe: bad argument");
    if (tiles_x * tiles_y * 256 > p->clahe_lut_cap) return lvk_set_error(ctx, LVK_ERR_CAPACITY, "too many CLAHE tiles");
    const int w = p->w[0], h = p->h[0];
    int ew = w, eh = h;
    if (!(w % tiles_x == 0 && h % tiles_y == 0)) { ew = w + (tiles_x - (w % tiles_x)); eh = h + (tiles_y - (h % tiles_y)); }
    const int tw = ew / tiles_x, th = eh / tiles_y, total = tw * th;
    const float lut_scale = (float)(255) / total;
    int clip = 0;
    if (clip_limit > 0.0) { clip = (int)(clip_limit * total / 256); if (clip < 1) clip = 1; }
    int vec4; clahe_launch_shape(d_img, w, h, stride, tw, th, tiles_x, tiles_y, &vec4);
    hipLaunchKernelGGL(k_clahe_lut, dim3(tiles_x * tiles_y), dim3(256), 0, ctx->stream, d_img, w, h, stride, tw, th, tiles_x, clip, lut_scale, p->clahe_lut, vec4);
    hipLaunchKernelGGL(k_level0_pad<true>, dim3((w + 2 * p->pad + 255) / 256, h + 2 * p->pad), dim3(256), 0, ctx->stream,
                       d_img, w, h, stride, (const uint8_t*)p->clahe_lut, tiles_x, tiles_y, 1.0f / tw, 1.0f / th, p->img[0], p->pad, p->istride[0]);
    return build_levels(ctx, p);
}

int lvk_pyramid_levels(const lvk_pyramid* p) { return p ? p->n_levels : 0; }

}  // extern "C"

// internal (frontend.hip): lvk_pyramid_build[_clahe] + the mosaic half of lvk_orb_prepare for one frame, level 0 and the ORB mosaic
// written by ONE kernel; same bytes as the public entry points produce.  The blur follows with lvk_orb_blur_only.
lvk_status lvk_pyramid_build_with_orb(lvk_context* ctx, lvk_pyramid* p, const uint8_t* d_img, int stride, int clahe, double clip_limit,
                                      int tiles_x, int tiles_y, uint8_t* d_ext, int* mosaic_done)
{
    if (!ctx || !p || !d_img || !d_ext || !mosaic_done) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_pyramid_build_with_orb: bad argument");
    const int w = p->w[0], h = p->h[0], B = LVK_ORB_BORDER, es = w + 2 * B, eh = h + 2 * B;
    *mosaic_done = 0;
    if (p->pad > B)                                     // the caller follows up with lvk_orb_prepare
        return clahe ? lvk_pyramid_build_clahe(ctx, p, d_img, stride, clip_limit, tiles_x, tiles_y) : lvk_pyramid_build(ctx, p, d_img, stride);
    *mosaic_done = 1;
    float inv_tw = 1.f, inv_th = 1.f;
    if (clahe) {
        if (tiles_x <= 0 || tiles_y <= 0) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_pyramid_build_with_orb: bad tile grid");
        if (tiles_x * tiles_y * 256 > p->clahe_lut_cap) return lvk_set_error(ctx, LVK_ERR_CAPACITY, "too many CLAHE tiles");
        int ew = w, eh2 = h;
        if (!(w % tiles_x == 0 && h % tiles_y == 0)) { ew = w + (tiles_x - (w % tiles_x)); eh2 = h + (tiles_y - (h % tiles_y)); }
        const int tw = ew / tiles_x, th = eh2 / tiles_y, total = tw * th;
        const float lut_scale = (float)(255) / total;
        int clip = 0;
        if (clip_limit > 0.0) { clip = (int)(clip_limit * total / 256); if (clip < 1) clip = 1; }
        int vec4; clahe_launch_shape(d_img, w, h, stride, tw, th, tiles_x, tiles_y, &vec4);
        hipLaunchKernelGGL(k_clahe_lut, dim3(tiles_x * tiles_y), dim3(256), 0, ctx->stream, d_img, w, h, stride, tw, th, tiles_x, clip, lut_scale, p->clahe_lut, vec4);
        inv_tw = 1.0f / tw; inv_th = 1.0f / th;
        hipLaunchKernelGGL(k_level0_pad_ext<true>, dim3((es + 255) / 256, eh), dim3(256), 0, ctx->stream, d_img, w, h, stride, (const uint8_t*)p->clahe_lut,
                           tiles_x, tiles_y, inv_tw, inv_th, p->img[0], p->pad, p->istride[0], d_ext, es);
    } else {
        hipLaunchKernelGGL(k_level0_pad_ext<false>, dim3((es + 255) / 256, eh), dim3(256), 0, ctx->stream, d_img, w, h, stride, (const uint8_t*)nullptr,
                           1, 1, 1.f, 1.f, p->img[0], p->pad, p->istride[0], d_ext, es);
    }
    return build_levels(ctx, p);
}
// the blur half of lvk_orb_prepare, for a mosaic that lvk_pyramid_build_with_orb already wrote (*mosaic_done == 1)
lvk_status lvk_orb_blur_only(lvk_context* ctx, const lvk_pyramid* p, const uint8_t* d_ext, uint8_t* d_blur)
{
    const int w = p->w[0], h = p->h[0], B = LVK_ORB_BORDER, es = w + 2 * B, eh = h + 2 * B;
    if ((es & 3) == 0 && (((size_t)d_ext | (size_t)d_blur) & 3) == 0 && es >= 8)
        hipLaunchKernelGGL(k_orb_blur_w, dim3((es + BW_TX - 1) / BW_TX, (eh + BW_TY - 1) / BW_TY), dim3(256), 0, ctx->stream, d_ext, w, h, es, d_blur);
    else
        hipLaunchKernelGGL(k_orb_blur, dim3((es + BL_TX - 1) / BL_TX, (eh + BL_TY - 1) / BL_TY), dim3(256), 0, ctx->stream, d_ext, w, h, es, d_blur);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

extern "C" {

lvk_status lvk_pyramid_level(const lvk_pyramid* p, int level, int* w, int* h, int* pad, int* istride, int* dstride,
                             const uint8_t** d_img, const int16_t** d_der)
{
    if (!p || level < 0 || level >= p->n_levels) return LVK_ERR_ARG;
    if (w) *w = p->w[level]; if (h) *h = p->h[level]; if (pad) *pad = p->pad;
    if (istride) *istride = p->istride[level]; if (dstride) *dstride = p->dstride[level];
    if (d_img) *d_img = p->img[level]; if (d_der) *d_der = p->der[level];
    return LVK_OK;
}

lvk_status lvk_orb_prepare(lvk_context* ctx, const lvk_pyramid* p, uint8_t* d_ext, uint8_t* d_blur)
{
    if (!ctx || !p || !d_ext || !d_blur) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_orb_prepare: bad argument");
    const int w = p->w[0], h = p->h[0], B = LVK_ORB_BORDER, es = w + 2 * B, eh = h + 2 * B;
    const uint8_t* s0 = p->img[0] + (size_t)p->pad * p->istride[0] + p->pad;
    const int grow = p->pad < B ? p->pad : B;
    hipLaunchKernelGGL(k_orb_ext, dim3((es + 255) / 256, eh), dim3(256), 0, ctx->stream, s0, w, h, p->istride[0], grow, d_ext, es);
    if ((es & 3) == 0 && (((size_t)d_ext | (size_t)d_blur) & 3) == 0 && es >= 8)
        hipLaunchKernelGGL(k_orb_blur_w, dim3((es + BW_TX - 1) / BW_TX, (eh + BW_TY - 1) / BW_TY), dim3(256), 0, ctx->stream, (const uint8_t*)d_ext, w, h, es, d_blur);
    else
        hipLaunchKernelGGL(k_orb_blur, dim3((es + BL_TX - 1) / BL_TX, (eh + BL_TY - 1) / BL_TY), dim3(256), 0, ctx->stream, (const uint8_t*)d_ext, w, h, es, d_blur);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

lvk_status lvk_min_eigen_map(lvk_context* ctx, const lvk_pyramid* p, float* d_eig)
{
    if (!ctx || !p || !d_eig) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_min_eigen_map: bad argument");
    const uint8_t* s0 = p->img[0] + (size_t)p->pad * p->istride[0] + p->pad;
    hipLaunchKernelGGL(k_min_eigen, dim3((p->w[0] + EG_TX - 1) / EG_TX, (p->h[0] + EG_TY - 1) / EG_TY), dim3(256), 0, ctx->stream,
                       s0, p->w[0], p->h[0], p->istride[0], d_eig);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

}  // extern "C"

// internal: GFTT on a precomputed eig map with caller-provided scratch (used by the frame-level path too)
lvk_status lvk_gftt_run(lvk_context* ctx, const float* d_eig, const uint8_t* d_mask, int w, int h, int max_corners,
                        double quality, double min_distance, unsigned* d_scratch, unsigned long long* d_cands, int cand_cap,
                        lvk_pt2f* d_out, int cap, int* d_n_out, const int* d_sub, bool prepared, bool max_done)
{   // max_done: scratch[0] already holds the masked maximum (lvk_mask_and_max)
   // prepared: the scratch words are zero (k_gftt_select leaves them so; a scratch of unknown content is cleared here) and the mask is final
    if (min_distance < 1.0) return lvk_set_error(ctx, LVK_ERR_UNSUPPORTED, "goodFeaturesToTrack with minDistance < 1 is not supported");
    const int cell = (int)rint(min_distance);
    const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
    if (max_corners > GF_MAX_OUT || max_corners <= 0) return lvk_set_error(ctx, LVK_ERR_UNSUPPORTED, "maxCorners must be in 1..%d", GF_MAX_OUT);
    if (!prepared) LVK_HIP(ctx, hipMemsetAsync(d_scratch, 0, GF_SCRATCH_UINTS * sizeof(unsigned), ctx->stream));
    if (!max_done)
    { int mb = (w * h / 4 + 2047) / 2048; mb = mb < 128 ? 128 : mb > 1024 ? 1024 : mb;      // ~8 four-pixel loads per lane
      hipLaunchKernelGGL(k_masked_max, dim3(mb), dim3(256), 0, ctx->stream, d_eig, d_mask, w * h, d_scratch); }
    hipLaunchKernelGGL(k_gftt_candidates, dim3((w - 2 + 255) / 256, (h - 2 + GC_ROWS - 1) / GC_ROWS), dim3(256), 0, ctx->stream, d_eig, d_mask, w, h, (float)quality, d_scratch, d_cands, cand_cap);
    // LDS of the selection kernel: survivors | one 8-byte cell per minDistance x minDistance square | the accepted corners (as many as can be
    // asked for), next to its static 36 KB (histogram).  A fine grid (minDistance 6 on 752 x 480: 10,080 cells; found by the whole-program fuzz
    // at 512 x 512 / 6, which asked for 177 KB and left the front-end in a failed state) takes the survivor buffer at half size - any
    // bucket boundaries give the same corners, the buckets only get shorter.
    static const size_t lds_static = [] { hipFuncAttributes fa; return hipFuncGetAttributes(&fa, (const void*)k_gftt_select) == hipSuccess ? (size_t)fa.sharedSizeBytes : (size_t)40 * 1024; }();
    const size_t lds_avail = (size_t)160 * 1024 - lds_static;
    const int acc_cap = max_corners;
    int surv_cap = GF_SURV;
    size_t shm = (size_t)surv_cap * 8 + (size_t)gw * gh * 8 + (size_t)acc_cap * 4;
    if (shm > lds_avail) { surv_cap = GF_SURV / 2; shm = (size_t)surv_cap * 8 + (size_t)gw * gh * 8 + (size_t)acc_cap * 4; }
    if (shm > lds_avail) return lvk_set_error(ctx, LVK_ERR_CAPACITY, "GFTT: a %dx%d grid of minDistance cells and %d corners need %zu bytes of LDS (%zu available)", gw, gh, max_corners, shm, lds_avail);
    LVK_LDS_OPTIN(ctx, 2, k_gftt_select, shm);   // the opt-in must leave room for the kernel's static LDS: ask for what is launched
    hipLaunchKernelGGL(k_gftt_select, dim3(1), dim3(1024), shm, ctx->stream, (const unsigned long long*)d_cands, cand_cap, w, h, max_corners, cell,
                       (float)(min_distance * min_distance), surv_cap, acc_cap, d_scratch, d_out, cap, d_n_out, d_sub);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

// frame path: mask + masked maximum in one launch (k_mask_max); the scratch words must be zero (k_gftt_select leaves them so)
lvk_status lvk_mask_and_max(lvk_context* ctx, const lvk_pt2f* d_pts, const int* d_n, int w, int h, int md, const float* d_eig, uint8_t* d_mask, unsigned* d_scratch)
{
    const size_t shm = (size_t)MM_ROWS * (((size_t)w + 3) & ~(size_t)3);
    if (shm > 60 * 1024) return lvk_set_error(ctx, LVK_ERR_CAPACITY, "image width %d too large for the mask kernel", w);
    hipLaunchKernelGGL(k_mask_max, dim3((h + MM_ROWS - 1) / MM_ROWS), dim3(256), shm, ctx->stream, d_pts, d_n, w, h, md, d_eig, d_mask, d_scratch);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

extern "C" lvk_status lvk_good_features(lvk_context* ctx, const lvk_pyramid* p, const uint8_t* d_mask, int max_corners,
                                        double quality, double min_distance, lvk_pt2f* d_out, int cap, int* d_n_out)
{
    if (!ctx || !p || !d_out || !d_n_out) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_good_features: bad argument");
    const int w = p->w[0], h = p->h[0];
    const int cand_cap = w * h;
    size_t cand_alloc = 1; while (cand_alloc < (size_t)cand_cap) cand_alloc <<= 1;     // bitonic sort pads to a power of two in place
    float* eig = (float*)lvk_ctx_scratch(ctx, 1, sizeof(float) * (size_t)w * h);
    unsigned* scratch = (unsigned*)lvk_ctx_scratch(ctx, 2, GF_SCRATCH_UINTS * sizeof(unsigned));
    unsigned long long* cands = (unsigned long long*)lvk_ctx_scratch(ctx, 3, sizeof(unsigned long long) * cand_alloc);
    if (!eig || !scratch || !cands) return lvk_set_error(ctx, LVK_ERR_DEVICE, "scratch allocation failed");
    lvk_status st = lvk_min_eigen_map(ctx, p, eig);
    if (st == LVK_OK) st = lvk_gftt_run(ctx, eig, d_mask, w, h, max_corners, quality, min_distance, scratch, cands, cand_cap, d_out, cap, d_n_out, nullptr, false, false);
    return st;
}
