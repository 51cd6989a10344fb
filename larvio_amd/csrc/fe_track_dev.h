// fe_track_dev.h — device functions shared by the stage-level kernels (fe_track.hip) and the
// frame-level kernels (frontend.hip).  wave64 only.
#pragma once
#include "lvk_internal.h"
#include <float.h>
#include <limits.h>
#include "lvk_sincosf.h"

// ------------------------------------------------------------------------- wave reductions
// integer all-reduce over the wavefront: four DPP row rotations (every lane gets its 16-lane row sum), then the four row
// sums are combined through v_readlane (scalar).  Integer adds are exact, so any order gives the same bits.
__device__ __forceinline__ int wave_sum_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false);   // row_ror:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, false);   // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xF, 0xF, false);   // row_ror:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xF, 0xF, false);   // row_ror:1
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
// exact sum over the wave of per-lane int32 partials, as int64 (all lanes get the result)
__device__ __forceinline__ long long wave_sum_i64(int v)
{
    int lo = v & 0xFFFF, hi = v >> 16;
    lo = wave_sum_i32(lo); hi = wave_sum_i32(hi);
    return ((long long)hi << 16) + (long long)lo;
}

// ------------------------------------------------------------------------- pyramidal LK
// [cv::calcOpticalFlowPyrLK / LKTrackerInvoker], OPTFLOW_USE_INITIAL_FLOW, minEigThreshold 1e-4.
// One wavefront per point, all levels and iterations in one launch.  The WIN x WIN template (I, Ix, Iy after the 14-bit bilinear
// blend) lives in registers, PL = ceil(WIN^2/64) pixels per lane; A11/A12/A22 and b1/b2 are EXACT integer sums (int32 per lane,
// int64 across the wave) converted to float once, so the result does not depend on the reduction order (see oracle/fe_track.c).
// All lanes hold identical copies of the scalar state.
// Measured cost model (MI355X, 21x21): ~4 us launch + ~2.9 us per level + ~0.7 us per iteration of the slowest track - the
// dependent instruction stream of ONE wavefront (~5 cycles per instruction), not bytes.  Three restructurings were measured and
// were not faster: LDS-staged search window; row-segment lanes with two unaligned 8-byte loads per iteration instead of 28 byte
// loads; requesting all levels' template/gradient runs up front.
#define LK_W_BITS 14
// The bilinear blends multiply bytes (or int16 derivatives) by 14-bit weights: both operands fit 24 bits, so the full-rate 24-bit
// multiplier gives the exact product (v_mul_lo_u32 is a quarter-rate instruction: 28 of them per iteration were ~40% of its issue time).
__device__ __forceinline__ int lk_blend_u8(int s00, int s01, int s10, int s11, int iw00, int iw01, int iw10, int iw11)
{
    return __mul24(s00, iw00) + __mul24(s01, iw01) + __mul24(s10, iw10) + __mul24(s11, iw11);     // SIGNED: iw11 = 2^14 - (the other three) can be -1 after rounding
}
__device__ __forceinline__ int lk_blend_i16(int x00, int x01, int x10, int x11, int iw00, int iw01, int iw10, int iw11)
{
    return __mul24(x00, iw00) + __mul24(x01, iw01) + __mul24(x10, iw10) + __mul24(x11, iw11);
}
template <int WIN>
__device__ __forceinline__ int lk_point_generic(const PyrView& prev, const PyrView& next, int n_levels, lvk_pt2f prev_pt, lvk_pt2f& next_pt,
                                        int& status, int max_count, double epsilon, int* __restrict__ iters_out)
{
    int total_it = 0;
    constexpr int NPIX = WIN * WIN;
    constexpr int PL = (NPIX + 63) / 64;
    const int lane = threadIdx.x & 63;
    const float half = (WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const int max_level = n_levels - 1;
    // per-lane pixel coordinates inside the window
    int wy[PL], wx[PL];
#pragma unroll
    for (int k = 0; k < PL; ++k) { int p = lane + 64 * k; wy[k] = p / WIN; wx[k] = p - wy[k] * WIN; }

    for (int level = max_level; level >= 0; --level) {
        const int cols = prev.w[level], rows = prev.h[level];
        const int stepI = prev.istride[level], stepJ = next.istride[level], dstep = prev.dstride[level];
        const uint8_t* __restrict__ Ibase = prev.img[level];
        const uint8_t* __restrict__ Jbase = next.img[level];
        const int16_t* __restrict__ Dbase = prev.der[level];
        const float lscale = (float)(1. / (1 << level));
        float prx = prev_pt.x * lscale, pry = prev_pt.y * lscale;
        float nx, ny;
        if (level == max_level) { nx = next_pt.x * lscale; ny = next_pt.y * lscale; }
        else { nx = next_pt.x * 2.f; ny = next_pt.y * 2.f; }
        next_pt.x = nx; next_pt.y = ny;
        int n_it = 0;

        prx -= half; pry -= half;
        const int ipx = d_cv_floor(prx), ipy = d_cv_floor(pry);
        if (ipx < -WIN || ipx >= cols || ipy < -WIN || ipy >= rows) {
            if (level == 0) status = 0;
            if (iters_out && lane == 0) iters_out[level] = 0;
            continue;
        }
        float a = prx - ipx, b = pry - ipy;
        int iw00 = d_cv_round((1.f - a) * (1.f - b) * (1 << LK_W_BITS));
        int iw01 = d_cv_round(a * (1.f - b) * (1 << LK_W_BITS));
        int iw10 = d_cv_round((1.f - a) * b * (1 << LK_W_BITS));
        int iw11 = (1 << LK_W_BITS) - iw00 - iw01 - iw10;

        short Iv[PL], Ixv[PL], Iyv[PL];
        int pA11 = 0, pA12 = 0, pA22 = 0;
#pragma unroll
        for (int k = 0; k < PL; ++k) {
            Iv[k] = 0; Ixv[k] = 0; Iyv[k] = 0;
            if (lane + 64 * k < NPIX) {
                const uint8_t* src = Ibase + (ptrdiff_t)(wy[k] + ipy) * stepI + (wx[k] + ipx);
                const int16_t* ds = Dbase + (ptrdiff_t)(wy[k] + ipy) * dstep + 2 * (wx[k] + ipx);
                int ival = (lk_blend_u8(src[0], src[1], src[stepI], src[stepI + 1], iw00, iw01, iw10, iw11) + (1 << (LK_W_BITS - 5 - 1))) >> (LK_W_BITS - 5);
                int ixval = (lk_blend_i16(ds[0], ds[2], ds[dstep], ds[dstep + 2], iw00, iw01, iw10, iw11) + (1 << (LK_W_BITS - 1))) >> LK_W_BITS;
                int iyval = (lk_blend_i16(ds[1], ds[3], ds[dstep + 1], ds[dstep + 3], iw00, iw01, iw10, iw11) + (1 << (LK_W_BITS - 1))) >> LK_W_BITS;
                Iv[k] = (short)ival; Ixv[k] = (short)ixval; Iyv[k] = (short)iyval;
                pA11 += ixval * ixval; pA12 += ixval * iyval; pA22 += iyval * iyval;
            }
        }
        const long long sA11 = wave_sum_i64(pA11), sA12 = wave_sum_i64(pA12), sA22 = wave_sum_i64(pA22);
        const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float min_eig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
        if (min_eig < (float)1e-4 || D < FLT_EPSILON) {      // float against float: LKTrackerInvoker keeps minEigThreshold as a float member
            if (level == 0) status = 0;
            if (iters_out && lane == 0) iters_out[level] = 0;
            continue;
        }
        D = 1.f / D;
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < max_count; ++j) {
            const int inx = d_cv_floor(nx), iny = d_cv_floor(ny);
            if (inx < -WIN || inx >= cols || iny < -WIN || iny >= rows) {
                if (level == 0) status = 0;
                break;
            }
            ++n_it;
            a = nx - inx; b = ny - iny;
            iw00 = d_cv_round((1.f - a) * (1.f - b) * (1 << LK_W_BITS));
            iw01 = d_cv_round(a * (1.f - b) * (1 << LK_W_BITS));
            iw10 = d_cv_round((1.f - a) * b * (1 << LK_W_BITS));
            iw11 = (1 << LK_W_BITS) - iw00 - iw01 - iw10;
            int pb1 = 0, pb2 = 0;
#pragma unroll
            for (int k = 0; k < PL; ++k) {
                if (lane + 64 * k < NPIX) {
                    const uint8_t* Jp = Jbase + (ptrdiff_t)(wy[k] + iny) * stepJ + (wx[k] + inx);
                    int diff = ((lk_blend_u8(Jp[0], Jp[1], Jp[stepJ], Jp[stepJ + 1], iw00, iw01, iw10, iw11) + (1 << (LK_W_BITS - 5 - 1))) >> (LK_W_BITS - 5)) - Iv[k];
                    pb1 += diff * Ixv[k]; pb2 += diff * Iyv[k];
                }
            }
            const long long sb1 = wave_sum_i64(pb1), sb2 = wave_sum_i64(pb2);
            const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;
            nx += dx; ny += dy;
            next_pt.x = nx + half; next_pt.y = ny + half;
            if ((double)dx * dx + (double)dy * dy <= epsilon) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                next_pt.x -= dx * 0.5f; next_pt.y -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (iters_out && lane == 0) iters_out[level] = n_it;
        total_it += n_it;
    }
    return total_it;
}

// ---- row-segment variant for WIN = 21 (every configuration of the reference uses patch_size 21).
// Lane l owns SEVEN CONSECUTIVE pixels of window row l / 3 (segment l % 3): 63 lanes cover the 21 x 21 window.  The bilinear taps
// of 7 neighbouring pixels are 8 consecutive bytes of two image rows, so a lane's share of the template is 2 eight-byte loads (I)
// + 4 sixteen-byte loads ((Ix, Iy) pairs) instead of 84 one- and two-byte loads, and an iteration is 2 eight-byte loads instead
// of 28 byte loads - all of a phase's loads in flight at once.  (gfx950 runs with unaligned access enabled: a dwordx2 load at any
// byte address is one instruction.)  The per-pixel integer arithmetic is exactly that of the generic path and the 2x2 system is
// made of exact integer sums, so which lane holds which pixel changes no bit of the result.
#define LK_RS_SEG 7
__device__ __forceinline__ int lk_point_rs21(const PyrView& prev, const PyrView& next, int n_levels, lvk_pt2f prev_pt, lvk_pt2f& next_pt,
                                             int& status, int max_count, double epsilon, int* __restrict__ iters_out)
{
    constexpr int WIN = 21;
    int total_it = 0;
    const int lane = threadIdx.x & 63;
    const float half = (WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const int max_level = n_levels - 1;
    const int wrow = lane / 3, x0 = (lane - 3 * wrow) * LK_RS_SEG;
    const bool act = lane < 3 * WIN;
    // The template of a level depends on prev_pt alone, not on what the iterations of the coarser level find: its raw bytes are
    // requested ONE LEVEL AHEAD (before the coarser level's iterations start) and sit in registers when that level begins - the
    // first-touch round trip of levels max-1..0 disappears from the dependent chain.
    struct Raw { unsigned long long i0, i1; uint4 d00, d01, d10, d11; int ipx, ipy; bool ok; };
    auto fetch = [&](int level) {
        Raw r; r.i0 = 0; r.i1 = 0; r.d00 = r.d01 = r.d10 = r.d11 = uint4{0, 0, 0, 0};
        const float lscale = (float)(1. / (1 << level));
        const float prx = prev_pt.x * lscale - half, pry = prev_pt.y * lscale - half;
        r.ipx = d_cv_floor(prx); r.ipy = d_cv_floor(pry);
        const int cols = prev.w[level], rows = prev.h[level];
        r.ok = !(r.ipx < -WIN || r.ipx >= cols || r.ipy < -WIN || r.ipy >= rows);
        if (r.ok && act) {
            const int stepI = prev.istride[level], dstep = prev.dstride[level];
            const uint8_t* src = prev.img[level] + (ptrdiff_t)(wrow + r.ipy) * stepI + (x0 + r.ipx);
            const int16_t* ds = prev.der[level] + (ptrdiff_t)(wrow + r.ipy) * dstep + 2 * (x0 + r.ipx);
            __builtin_memcpy(&r.i0, src, 8); __builtin_memcpy(&r.i1, src + stepI, 8);
            __builtin_memcpy(&r.d00, ds, 16); __builtin_memcpy(&r.d01, ds + 8, 16);
            __builtin_memcpy(&r.d10, ds + dstep, 16); __builtin_memcpy(&r.d11, ds + dstep + 8, 16);
        }
        return r;
    };
    Raw cur = fetch(max_level);

    for (int level = max_level; level >= 0; --level) {
        const Raw raw = cur;
        if (level > 0) cur = fetch(level - 1);
        const int cols = prev.w[level], rows = prev.h[level];
        const int stepJ = next.istride[level];
        const uint8_t* __restrict__ Jbase = next.img[level];
        const float lscale = (float)(1. / (1 << level));
        float prx = prev_pt.x * lscale, pry = prev_pt.y * lscale;
        float nx, ny;
        if (level == max_level) { nx = next_pt.x * lscale; ny = next_pt.y * lscale; }
        else { nx = next_pt.x * 2.f; ny = next_pt.y * 2.f; }
        next_pt.x = nx; next_pt.y = ny;
        int n_it = 0;

        prx -= half; pry -= half;
        const int ipx = raw.ipx, ipy = raw.ipy;               // = cvFloor(prx), cvFloor(pry): the same expressions as in fetch()
        if (!raw.ok) {
            if (level == 0) status = 0;
            if (iters_out && lane == 0) iters_out[level] = 0;
            continue;
        }
        float a = prx - ipx, b = pry - ipy;
        int iw00 = d_cv_round((1.f - a) * (1.f - b) * (1 << LK_W_BITS));
        int iw01 = d_cv_round(a * (1.f - b) * (1 << LK_W_BITS));
        int iw10 = d_cv_round((1.f - a) * b * (1 << LK_W_BITS));
        int iw11 = (1 << LK_W_BITS) - iw00 - iw01 - iw10;

        short Iv[LK_RS_SEG], Ixv[LK_RS_SEG], Iyv[LK_RS_SEG];
        int pA11 = 0, pA12 = 0, pA22 = 0;
        {
            const unsigned long long i0 = raw.i0, i1 = raw.i1;
            const uint4 d00 = raw.d00, d01 = raw.d01, d10 = raw.d10, d11 = raw.d11;
            const unsigned dr0[8] = {d00.x, d00.y, d00.z, d00.w, d01.x, d01.y, d01.z, d01.w};
            const unsigned dr1[8] = {d10.x, d10.y, d10.z, d10.w, d11.x, d11.y, d11.z, d11.w};
#pragma unroll
            for (int j = 0; j < LK_RS_SEG; ++j) {
                const int s00 = (int)((i0 >> (8 * j)) & 0xFF), s01 = (int)((i0 >> (8 * j + 8)) & 0xFF);
                const int s10 = (int)((i1 >> (8 * j)) & 0xFF), s11 = (int)((i1 >> (8 * j + 8)) & 0xFF);
                const int x00 = (short)(dr0[j] & 0xFFFF), y00 = (short)(dr0[j] >> 16), x01 = (short)(dr0[j + 1] & 0xFFFF), y01 = (short)(dr0[j + 1] >> 16);
                const int x10 = (short)(dr1[j] & 0xFFFF), y10 = (short)(dr1[j] >> 16), x11 = (short)(dr1[j + 1] & 0xFFFF), y11 = (short)(dr1[j + 1] >> 16);
                int ival = (lk_blend_u8(s00, s01, s10, s11, iw00, iw01, iw10, iw11) + (1 << (LK_W_BITS - 5 - 1))) >> (LK_W_BITS - 5);
                int ixval = (lk_blend_i16(x00, x01, x10, x11, iw00, iw01, iw10, iw11) + (1 << (LK_W_BITS - 1))) >> LK_W_BITS;
                int iyval = (lk_blend_i16(y00, y01, y10, y11, iw00, iw01, iw10, iw11) + (1 << (LK_W_BITS - 1))) >> LK_W_BITS;
                if (!act) { ival = 0; ixval = 0; iyval = 0; }
                Iv[j] = (short)ival; Ixv[j] = (short)ixval; Iyv[j] = (short)iyval;
                pA11 += ixval * ixval; pA12 += ixval * iyval; pA22 += iyval * iyval;
            }
        }
        const long long sA11 = wave_sum_i64(pA11), sA12 = wave_sum_i64(pA12), sA22 = wave_sum_i64(pA22);
        const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float min_eig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
        if (min_eig < (float)1e-4 || D < FLT_EPSILON) {      // float against float: LKTrackerInvoker keeps minEigThreshold as a float member
            if (level == 0) status = 0;
            if (iters_out && lane == 0) iters_out[level] = 0;
            continue;
        }
        D = 1.f / D;
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < max_count; ++j) {
            const int inx = d_cv_floor(nx), iny = d_cv_floor(ny);
            if (inx < -WIN || inx >= cols || iny < -WIN || iny >= rows) {
                if (level == 0) status = 0;
                break;
            }
            ++n_it;
            a = nx - inx; b = ny - iny;
            iw00 = d_cv_round((1.f - a) * (1.f - b) * (1 << LK_W_BITS));
            iw01 = d_cv_round(a * (1.f - b) * (1 << LK_W_BITS));
            iw10 = d_cv_round((1.f - a) * b * (1 << LK_W_BITS));
            iw11 = (1 << LK_W_BITS) - iw00 - iw01 - iw10;
            int pb1 = 0, pb2 = 0;
            unsigned long long j0 = 0, j1 = 0;
            if (act) {
                const uint8_t* Jp = Jbase + (ptrdiff_t)(wrow + iny) * stepJ + (x0 + inx);
                __builtin_memcpy(&j0, Jp, 8); __builtin_memcpy(&j1, Jp + stepJ, 8);
            }
#pragma unroll
            for (int k = 0; k < LK_RS_SEG; ++k) {
                const int s00 = (int)((j0 >> (8 * k)) & 0xFF), s01 = (int)((j0 >> (8 * k + 8)) & 0xFF);
                const int s10 = (int)((j1 >> (8 * k)) & 0xFF), s11 = (int)((j1 >> (8 * k + 8)) & 0xFF);
                int diff = ((lk_blend_u8(s00, s01, s10, s11, iw00, iw01, iw10, iw11) + (1 << (LK_W_BITS - 5 - 1))) >> (LK_W_BITS - 5)) - Iv[k];
                if (!act) diff = 0;
                pb1 += diff * Ixv[k]; pb2 += diff * Iyv[k];
            }
            const long long sb1 = wave_sum_i64(pb1), sb2 = wave_sum_i64(pb2);
            const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;
            nx += dx; ny += dy;
            next_pt.x = nx + half; next_pt.y = ny + half;
            if ((double)dx * dx + (double)dy * dy <= epsilon) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                next_pt.x -= dx * 0.5f; next_pt.y -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (iters_out && lane == 0) iters_out[level] = n_it;
        total_it += n_it;
    }
    return total_it;
}

// ---- the same row-segment layout with the reductions through LDS (round 6).
// What an iteration of lk_point_rs21 costs is instruction issue of ONE wavefront (~280 instructions, ~5.8 cycles each), and a third
// of them are the two exact 64-bit wave sums: 16 DPP adds + 16 v_readlane + the scalar carry chain + two int64 -> float conversions
// with find-first-bit.  Here: three DPP rotations leave the sum of each aligned group of 8 lanes in the group's last lane (7 pixels x
// 8 lanes x |diff * Ix| <= 8 * 7 * 8160 * 4080 = 1.86e9 < 2^31: still an exact int32), the eight group leaders add theirs,
// sign-extended, to a 64-bit LDS accumulator (ds_add_u64) that is NEVER reset - a running total modulo 2^64 whose difference to the
// previous reading is this reduction's sum, exactly - and every lane reads the accumulators back.  LDS instructions of one wavefront
// execute in issue order, so the read follows the adds without a barrier.  The int64 -> float conversion goes through double
// (hi * 2^32 + lo is exact below 2^53, and (float)(double) rounds to nearest-even like (float)(int64)): three instructions.
// Integer sums in a different order: the same bits.  Second change: the 16 bytes of the next image a lane holds are reloaded only
// when the window's integer origin moved - after the first step of a level it usually has not (the update is sub-pixel), and the
// L2 round trip leaves the iteration's dependent chain.
struct LkLdsAcc {
    unsigned off;                   // LDS byte address of 4 x u64 owned by this wavefront (3 running totals + pad, 16-byte aligned; zeroed once at kernel start)
    unsigned long long prev[3];
};
__device__ __forceinline__ int lk_group8_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, false);   // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xF, 0xF, false);   // row_ror:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xF, 0xF, false);   // row_ror:1  -> lane l: sum over lanes l-7..l of its row (cyclic)
    return v;
}
// LDS instructions as written.  ds_add_u64: through atomicAdd the compiler's atomic optimiser turns a same-address LDS atomic into a
// per-active-lane v_readlane loop (8 trips x 21 instructions here) - the very chain this variant removes.  The reads: through a
// generic pointer they become flat loads with system-scope cache bits.
__device__ __forceinline__ void lds_add_u64(unsigned off, unsigned long long v)
{
    asm volatile("ds_add_u64 %0, %1" :: "v"(off), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_read_2xu64(unsigned off, unsigned long long& a, unsigned long long& b)
{
    uint4 r;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(off) : "memory");
    a = ((unsigned long long)r.y << 32) | r.x; b = ((unsigned long long)r.w << 32) | r.z;
}
__device__ __forceinline__ float lk_i64_to_f32_scaled(long long s, float scale)
{   // exact for |s| < 2^53: hi * 2^32 + lo in double, then one rounding to float - the same value as (float)s
    const double d = __builtin_fma((double)(int)(s >> 32), 4294967296.0, (double)(unsigned)(s & 0xFFFFFFFFll));
    return (float)d * scale;
}
template <int NV>
__device__ __forceinline__ void lk_lds_sums(LkLdsAcc& A, const int (&part)[NV], float (&out)[NV], float scale)
{
    static_assert(NV == 2 || NV == 3, "two or three sums");
    const int lane = threadIdx.x & 63;
    int g[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) g[k] = lk_group8_sum(part[k]);
    if ((lane & 7) == 7) {
#pragma unroll
        for (int k = 0; k < NV; ++k) lds_add_u64(A.off + 8 * k, (unsigned long long)(long long)g[k]);
    }
    unsigned long long now[4];
    lds_read_2xu64(A.off, now[0], now[1]);
    if (NV == 3) lds_read_2xu64(A.off + 16, now[2], now[3]);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const long long s = (long long)(now[k] - A.prev[k]);
        A.prev[k] = now[k];
        // every lane read the same totals: say so, the loop control around this stays scalar
        out[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lk_i64_to_f32_scaled(s, scale))));
    }
}
#ifdef LVK_LK_TIMING
static __device__ unsigned long long g_lk_tick[4096][12];    // -DLVK_LK_TIMING=1: the kernel's spans only (g_lk_span, frontend.hip); =2: this phase account too
#endif
#if defined(LVK_LK_TIMING) && LVK_LK_TIMING >= 2
// per-block phase account of the LK wavefront (variants/lkt.so, tools/gpu/lk_ticks.py): shader-clock cycles by category
// 0 level set-up (template blends incl. the wait for the fetched bytes)  1 A sums  2 eigen test / inverse  3 iteration: origin, weights,
// load (when the origin moved), blends  4 iteration: b sums  5 iteration: solve + stop tests  6 iterations  7 levels entered  8 loads issued
#define LKT_DECL unsigned long long lkt_prev = clock64(); unsigned long long lkt_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}
#define LKT(k) do { const unsigned long long lkt_now = clock64(); lkt_acc[k] += lkt_now - lkt_prev; lkt_prev = lkt_now; } while (0)
#define LKT_COUNT(k) do { lkt_acc[k] += 1; } while (0)
#define LKT_FLUSH(pass) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) { for (int q = 0; q < 9; ++q) { if (pass == 0) g_lk_tick[blockIdx.x][q] = lkt_acc[q]; else g_lk_tick[blockIdx.x][q] += lkt_acc[q]; } } } while (0)
#else
#define LKT_DECL do { } while (0)
#define LKT(k) do { } while (0)
#define LKT_COUNT(k) do { } while (0)
#define LKT_FLUSH(pass) do { } while (0)
#endif
// -DLVK_LK_BOUNDS (debug builds only): every window load of variants 1 / 2 is checked against the level's padded plane; the first
// violation is recorded (frontend.hip prints it when the front-end is destroyed) and the load skipped
#ifdef LVK_LK_BOUNDS
static __device__ int g_lk_oob[16];
__device__ __forceinline__ bool lk_in_bounds(const PyrView& V, int level, int row, int col, int tag, float fx, float fy)
{
    const bool ok = level >= 0 && level < V.n_levels && row >= -V.pad && row + 1 < V.h[level] + V.pad && col >= -V.pad && col + 8 <= V.w[level] + V.pad;
    if (!ok && atomicAdd(&g_lk_oob[0], 1) == 0) {
        g_lk_oob[1] = tag; g_lk_oob[2] = level; g_lk_oob[3] = row; g_lk_oob[4] = col; g_lk_oob[5] = V.w[level & 7]; g_lk_oob[6] = V.h[level & 7];
        g_lk_oob[7] = __builtin_bit_cast(int, fx); g_lk_oob[8] = __builtin_bit_cast(int, fy); g_lk_oob[9] = blockIdx.x; g_lk_oob[10] = threadIdx.x; g_lk_oob[11] = V.n_levels;
    }
    return ok;
}
#define LK_INB(V, level, row, col, tag, fx, fy) lk_in_bounds(V, level, row, col, tag, fx, fy)
#else
#define LK_INB(V, level, row, col, tag, fx, fy) true
#endif
template <int LKT_PASS = 0>
__device__ __forceinline__ int lk_point_rs21_lds(const PyrView& prev, const PyrView& next, int n_levels, lvk_pt2f prev_pt, lvk_pt2f& next_pt,
                                                 int& status, int max_count, double epsilon, int* __restrict__ iters_out, LkLdsAcc& acc)
{
    // every lane holds the same points: say so (v_readfirstlane), the level / iteration control flow below then compiles to scalar branches
    auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    prev_pt.x = uni(prev_pt.x); prev_pt.y = uni(prev_pt.y); next_pt.x = uni(next_pt.x); next_pt.y = uni(next_pt.y);
    constexpr int WIN = 21;
    int total_it = 0;
    const int lane = threadIdx.x & 63;
    const float half = (WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const int max_level = n_levels - 1;
    const int wrow = lane / 3, x0 = (lane - 3 * wrow) * LK_RS_SEG;
    const bool act = lane < 3 * WIN;
    struct Raw { unsigned long long i0, i1; uint4 d00, d01, d10, d11; int ipx, ipy; bool ok; };
    auto fetch = [&](int level) {
        Raw r; r.i0 = 0; r.i1 = 0; r.d00 = r.d01 = r.d10 = r.d11 = uint4{0, 0, 0, 0};
        const float lscale = (float)(1. / (1 << level));
        const float prx = prev_pt.x * lscale - half, pry = prev_pt.y * lscale - half;
        r.ipx = d_cv_floor(prx); r.ipy = d_cv_floor(pry);
        const int cols = prev.w[level], rows = prev.h[level];
        r.ok = !(r.ipx < -WIN || r.ipx >= cols || r.ipy < -WIN || r.ipy >= rows);
        if (r.ok && act && LK_INB(prev, level, wrow + r.ipy, x0 + r.ipx, 1, prev_pt.x, prev_pt.y)) {
            const int stepI = prev.istride[level], dstep = prev.dstride[level];
            const uint8_t* src = prev.img[level] + (ptrdiff_t)(wrow + r.ipy) * stepI + (x0 + r.ipx);
            const int16_t* ds = prev.der[level] + (ptrdiff_t)(wrow + r.ipy) * dstep + 2 * (x0 + r.ipx);
            __builtin_memcpy(&r.i0, src, 8); __builtin_memcpy(&r.i1, src + stepI, 8);
            __builtin_memcpy(&r.d00, ds, 16); __builtin_memcpy(&r.d01, ds + 8, 16);
            __builtin_memcpy(&r.d10, ds + dstep, 16); __builtin_memcpy(&r.d11, ds + dstep + 8, 16);
        }
        return r;
    };
    Raw cur = fetch(max_level);
    LKT_DECL;

    for (int level = max_level; level >= 0; --level) {
        const Raw raw = cur;
        if (level > 0) cur = fetch(level - 1);
        const int cols = prev.w[level], rows = prev.h[level];
        const int stepJ = next.istride[level];
        const uint8_t* __restrict__ Jbase = next.img[level];
        const float lscale = (float)(1. / (1 << level));
        float prx = prev_pt.x * lscale, pry = prev_pt.y * lscale;
        float nx, ny;
        if (level == max_level) { nx = next_pt.x * lscale; ny = next_pt.y * lscale; }
        else { nx = next_pt.x * 2.f; ny = next_pt.y * 2.f; }
        next_pt.x = nx; next_pt.y = ny;
        int n_it = 0;

        prx -= half; pry -= half;
        const int ipx = raw.ipx, ipy = raw.ipy;
        if (!raw.ok) {
            if (level == 0) status = 0;
            if (iters_out && lane == 0) iters_out[level] = 0;
            continue;
        }
        float a = prx - ipx, b = pry - ipy;
        int iw00 = d_cv_round((1.f - a) * (1.f - b) * (1 << LK_W_BITS));
        int iw01 = d_cv_round(a * (1.f - b) * (1 << LK_W_BITS));
        int iw10 = d_cv_round((1.f - a) * b * (1 << LK_W_BITS));
        int iw11 = (1 << LK_W_BITS) - iw00 - iw01 - iw10;

        short Iv[LK_RS_SEG], Ixv[LK_RS_SEG], Iyv[LK_RS_SEG];
        int pA[3] = {0, 0, 0};
        {
            const unsigned long long i0 = raw.i0, i1 = raw.i1;
            const uint4 d00 = raw.d00, d01 = raw.d01, d10 = raw.d10, d11 = raw.d11;
            const unsigned dr0[8] = {d00.x, d00.y, d00.z, d00.w, d01.x, d01.y, d01.z, d01.w};
            const unsigned dr1[8] = {d10.x, d10.y, d10.z, d10.w, d11.x, d11.y, d11.z, d11.w};
#pragma unroll
            for (int j = 0; j < LK_RS_SEG; ++j) {
                const int s00 = (int)((i0 >> (8 * j)) & 0xFF), s01 = (int)((i0 >> (8 * j + 8)) & 0xFF);
                const int s10 = (int)((i1 >> (8 * j)) & 0xFF), s11 = (int)((i1 >> (8 * j + 8)) & 0xFF);
                const int x00 = (short)(dr0[j] & 0xFFFF), y00 = (short)(dr0[j] >> 16), x01 = (short)(dr0[j + 1] & 0xFFFF), y01 = (short)(dr0[j + 1] >> 16);
                const int x10 = (short)(dr1[j] & 0xFFFF), y10 = (short)(dr1[j] >> 16), x11 = (short)(dr1[j + 1] & 0xFFFF), y11 = (short)(dr1[j + 1] >> 16);
                int ival = (lk_blend_u8(s00, s01, s10, s11, iw00, iw01, iw10, iw11) + (1 << (LK_W_BITS - 5 - 1))) >> (LK_W_BITS - 5);
                int ixval = (lk_blend_i16(x00, x01, x10, x11, iw00, iw01, iw10, iw11) + (1 << (LK_W_BITS - 1))) >> LK_W_BITS;
                int iyval = (lk_blend_i16(y00, y01, y10, y11, iw00, iw01, iw10, iw11) + (1 << (LK_W_BITS - 1))) >> LK_W_BITS;
                if (!act) { ival = 0; ixval = 0; iyval = 0; }
                Iv[j] = (short)ival; Ixv[j] = (short)ixval; Iyv[j] = (short)iyval;
                pA[0] += ixval * ixval; pA[1] += ixval * iyval; pA[2] += iyval * iyval;
            }
        }
        LKT(0); LKT_COUNT(7);
        float fA[3];
        lk_lds_sums<3>(acc, pA, fA, FLT_SCALE);
        LKT(1);
        const float A11 = fA[0], A12 = fA[1], A22 = fA[2];
        float D = A11 * A22 - A12 * A12;
        const float min_eig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
        if (min_eig < (float)1e-4 || D < FLT_EPSILON) {      // float against float: LKTrackerInvoker keeps minEigThreshold as a float member
            if (level == 0) status = 0;
            if (iters_out && lane == 0) iters_out[level] = 0;
            continue;
        }
        D = 1.f / D;
        nx -= half; ny -= half;
        LKT(2);
        float pdx = 0.f, pdy = 0.f;
        unsigned long long j0 = 0, j1 = 0;
        int held_x = INT_MIN, held_y = INT_MIN;            // window origin the bytes in j0 / j1 were loaded for
        for (int j = 0; j < max_count; ++j) {
            const int inx = d_cv_floor(nx), iny = d_cv_floor(ny);
            if (inx < -WIN || inx >= cols || iny < -WIN || iny >= rows) {
                if (level == 0) status = 0;
                break;
            }
            ++n_it;
            if (inx != held_x || iny != held_y) {
                held_x = inx; held_y = iny; LKT_COUNT(8);
                if (act && LK_INB(next, level, wrow + iny, x0 + inx, 2, nx, ny)) {
                    const uint8_t* Jp = Jbase + (ptrdiff_t)(wrow + iny) * stepJ + (x0 + inx);
                    __builtin_memcpy(&j0, Jp, 8); __builtin_memcpy(&j1, Jp + stepJ, 8);
                }
            }
            a = nx - inx; b = ny - iny;
            iw00 = d_cv_round((1.f - a) * (1.f - b) * (1 << LK_W_BITS));
            iw01 = d_cv_round(a * (1.f - b) * (1 << LK_W_BITS));
            iw10 = d_cv_round((1.f - a) * b * (1 << LK_W_BITS));
            iw11 = (1 << LK_W_BITS) - iw00 - iw01 - iw10;
            int pb[2] = {0, 0};
#pragma unroll
            for (int k = 0; k < LK_RS_SEG; ++k) {
                const int s00 = (int)((j0 >> (8 * k)) & 0xFF), s01 = (int)((j0 >> (8 * k + 8)) & 0xFF);
                const int s10 = (int)((j1 >> (8 * k)) & 0xFF), s11 = (int)((j1 >> (8 * k + 8)) & 0xFF);
                const int diff = ((lk_blend_u8(s00, s01, s10, s11, iw00, iw01, iw10, iw11) + (1 << (LK_W_BITS - 5 - 1))) >> (LK_W_BITS - 5)) - Iv[k];
                pb[0] += diff * Ixv[k]; pb[1] += diff * Iyv[k];          // lane 63 (no pixels): Ixv = Iyv = 0, j0 = j1 = 0
            }
            LKT(3); LKT_COUNT(6);
            float fb[2];
            lk_lds_sums<2>(acc, pb, fb, FLT_SCALE);
            LKT(4);
            const float b1 = fb[0], b2 = fb[1];
            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;
            nx += dx; ny += dy;
            next_pt.x = nx + half; next_pt.y = ny + half;
            if ((double)dx * dx + (double)dy * dy <= epsilon) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                next_pt.x -= dx * 0.5f; next_pt.y -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
            LKT(5);
        }
        LKT(5);
        if (iters_out && lane == 0) iters_out[level] = n_it;
        total_it += n_it;
    }
    LKT_FLUSH(LKT_PASS);
    return total_it;
}

// ---- variant 2: the same arithmetic, the levels' templates built by OTHER wavefronts of the block (k_fe_lk_pipe, frontend.hip).
// What a track costs is one wavefront's dependent instruction stream (profiles/r6_e_*: fewer instructions in one phase did not make it
// shorter), and a third of that stream - per level: six loads, 21 bilinear blends, three exact sums, a square root and two divisions,
// ~1.7 us - does not depend on what the iterations find: the template of a level is a function of the point in the FIRST image alone.
// So the wavefront that iterates no longer builds it: three builder wavefronts (levels l, l + 3, ...) do, at once, before the pass
// starts, and leave per lane the 21 shorts (I, Ix, Iy of its seven pixels) and per level A11, A12, A22, 1 / D and the verdict of the
// eigenvalue test in LDS.  Same loads, same integer blends, same exact sums, same float expressions: the same bits.
#define LK_PIPE_MAX_LEVELS 4
struct LkTplLevel {
    unsigned long long px[64][6];       // per lane: Iv[7], Ixv[7], Iyv[7] (21 shorts, 48 bytes)
    float A11, A12, A22, Dinv;
    int state;                          // 0 the window's origin lies outside the image, 1 the eigenvalue / determinant test failed (either: no iterations, status 0 at level 0), 2 iterate
    int pad_[3];
};
__device__ __forceinline__ void lk_tpl_build21(const PyrView& I, int level, lvk_pt2f prev_pt, LkTplLevel& T, LkLdsAcc& acc)
{
    constexpr int WIN = 21;
    auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    prev_pt.x = uni(prev_pt.x); prev_pt.y = uni(prev_pt.y);
    const int lane = threadIdx.x & 63;
    const float half = (WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const int wrow = lane / 3, x0 = (lane - 3 * wrow) * LK_RS_SEG;
    const bool act = lane < 3 * WIN;
    const float lscale = (float)(1. / (1 << level));
    const float prx = prev_pt.x * lscale - half, pry = prev_pt.y * lscale - half;
    const int ipx = d_cv_floor(prx), ipy = d_cv_floor(pry);
    const int cols = I.w[level], rows = I.h[level];
    if (ipx < -WIN || ipx >= cols || ipy < -WIN || ipy >= rows) {
        if (lane == 0) T.state = 0;
        return;
    }
    unsigned long long i0 = 0, i1 = 0; uint4 d00 = {0, 0, 0, 0}, d01 = d00, d10 = d00, d11 = d00;
    if (act && LK_INB(I, level, wrow + ipy, x0 + ipx, 3, prev_pt.x, prev_pt.y)) {
        const int stepI = I.istride[level], dstep = I.dstride[level];
        const uint8_t* src = I.img[level] + (ptrdiff_t)(wrow + ipy) * stepI + (x0 + ipx);
        const int16_t* ds = I.der[level] + (ptrdiff_t)(wrow + ipy) * dstep + 2 * (x0 + ipx);
        __builtin_memcpy(&i0, src, 8); __builtin_memcpy(&i1, src + stepI, 8);
        __builtin_memcpy(&d00, ds, 16); __builtin_memcpy(&d01, ds + 8, 16);
        __builtin_memcpy(&d10, ds + dstep, 16); __builtin_memcpy(&d11, ds + dstep + 8, 16);
    }
    const float a = prx - ipx, b = pry - ipy;
    const int iw00 = d_cv_round((1.f - a) * (1.f - b) * (1 << LK_W_BITS));
    const int iw01 = d_cv_round(a * (1.f - b) * (1 << LK_W_BITS));
    const int iw10 = d_cv_round((1.f - a) * b * (1 << LK_W_BITS));
    const int iw11 = (1 << LK_W_BITS) - iw00 - iw01 - iw10;
    unsigned short q[24];
#pragma unroll
    for (int j = 21; j < 24; ++j) q[j] = 0;
    int pA[3] = {0, 0, 0};
    {
        const unsigned dr0[8] = {d00.x, d00.y, d00.z, d00.w, d01.x, d01.y, d01.z, d01.w};
        const unsigned dr1[8] = {d10.x, d10.y, d10.z, d10.w, d11.x, d11.y, d11.z, d11.w};
#pragma unroll
        for (int j = 0; j < LK_RS_SEG; ++j) {
            const int s00 = (int)((i0 >> (8 * j)) & 0xFF), s01 = (int)((i0 >> (8 * j + 8)) & 0xFF);
            const int s10 = (int)((i1 >> (8 * j)) & 0xFF), s11 = (int)((i1 >> (8 * j + 8)) & 0xFF);
            const int x00 = (short)(dr0[j] & 0xFFFF), y00 = (short)(dr0[j] >> 16), x01 = (short)(dr0[j + 1] & 0xFFFF), y01 = (short)(dr0[j + 1] >> 16);
            const int x10 = (short)(dr1[j] & 0xFFFF), y10 = (short)(dr1[j] >> 16), x11 = (short)(dr1[j + 1] & 0xFFFF), y11 = (short)(dr1[j + 1] >> 16);
            int ival = (lk_blend_u8(s00, s01, s10, s11, iw00, iw01, iw10, iw11) + (1 << (LK_W_BITS - 5 - 1))) >> (LK_W_BITS - 5);
            int ixval = (lk_blend_i16(x00, x01, x10, x11, iw00, iw01, iw10, iw11) + (1 << (LK_W_BITS - 1))) >> LK_W_BITS;
            int iyval = (lk_blend_i16(y00, y01, y10, y11, iw00, iw01, iw10, iw11) + (1 << (LK_W_BITS - 1))) >> LK_W_BITS;
            if (!act) { ival = 0; ixval = 0; iyval = 0; }
            q[j] = (unsigned short)(short)ival; q[7 + j] = (unsigned short)(short)ixval; q[14 + j] = (unsigned short)(short)iyval;
            pA[0] += ixval * ixval; pA[1] += ixval * iyval; pA[2] += iyval * iyval;
        }
    }
    float fA[3];
    lk_lds_sums<3>(acc, pA, fA, FLT_SCALE);
    const float A11 = fA[0], A12 = fA[1], A22 = fA[2];
    const float D = A11 * A22 - A12 * A12;
    const float min_eig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
    const bool bad = min_eig < (float)1e-4 || D < FLT_EPSILON;
#pragma unroll
    for (int j = 0; j < 6; ++j)
        T.px[lane][j] = (unsigned long long)q[4 * j] | ((unsigned long long)q[4 * j + 1] << 16) | ((unsigned long long)q[4 * j + 2] << 32) | ((unsigned long long)q[4 * j + 3] << 48);
    if (lane == 0) { T.A11 = A11; T.A12 = A12; T.A22 = A22; T.Dinv = bad ? 0.f : 1.f / D; T.state = bad ? 1 : 2; }
}
// the iterations of one pass over all levels, by the wavefront that owns the track; T[level] was filled by lk_tpl_build21 (and a barrier)
__device__ __forceinline__ int lk_pass_iterate21(const PyrView& next, int n_levels, const LkTplLevel* T, lvk_pt2f& next_pt, int& status,
                                                 int max_count, double epsilon, LkLdsAcc& acc)
{
    constexpr int WIN = 21;
    auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    next_pt.x = uni(next_pt.x); next_pt.y = uni(next_pt.y);
    int total_it = 0;
    const int lane = threadIdx.x & 63;
    const float half = (WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const int max_level = n_levels - 1;
    const int wrow = lane / 3, x0 = (lane - 3 * wrow) * LK_RS_SEG;
    const bool act = lane < 3 * WIN;
    for (int level = max_level; level >= 0; --level) {
        const int cols = next.w[level], rows = next.h[level];
        const int stepJ = next.istride[level];
        const uint8_t* __restrict__ Jbase = next.img[level];
        const float lscale = (float)(1. / (1 << level));
        float nx, ny;
        if (level == max_level) { nx = next_pt.x * lscale; ny = next_pt.y * lscale; }
        else { nx = next_pt.x * 2.f; ny = next_pt.y * 2.f; }
        next_pt.x = nx; next_pt.y = ny;
        int n_it = 0;
        const LkTplLevel& L = T[level];
        const int state = __builtin_amdgcn_readfirstlane(L.state);
        if (state != 2) {
            if (level == 0) status = 0;
            continue;
        }
        short Iv[LK_RS_SEG], Ixv[LK_RS_SEG], Iyv[LK_RS_SEG];
        {
            unsigned long long w[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) w[j] = L.px[lane][j];
#pragma unroll
            for (int j = 0; j < LK_RS_SEG; ++j) {
                Iv[j] = (short)(w[j >> 2] >> (16 * (j & 3)));
                Ixv[j] = (short)(w[(7 + j) >> 2] >> (16 * ((7 + j) & 3)));
                Iyv[j] = (short)(w[(14 + j) >> 2] >> (16 * ((14 + j) & 3)));
            }
        }
        const float A11 = uni(L.A11), A12 = uni(L.A12), A22 = uni(L.A22), D = uni(L.Dinv);
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        unsigned long long j0 = 0, j1 = 0;
        int held_x = INT_MIN, held_y = INT_MIN;
        for (int j = 0; j < max_count; ++j) {
            const int inx = d_cv_floor(nx), iny = d_cv_floor(ny);
            if (inx < -WIN || inx >= cols || iny < -WIN || iny >= rows) {
                if (level == 0) status = 0;
                break;
            }
            ++n_it;
            if (inx != held_x || iny != held_y) {
                held_x = inx; held_y = iny;
                if (act && LK_INB(next, level, wrow + iny, x0 + inx, 4, nx, ny)) {
                    const uint8_t* Jp = Jbase + (ptrdiff_t)(wrow + iny) * stepJ + (x0 + inx);
                    __builtin_memcpy(&j0, Jp, 8); __builtin_memcpy(&j1, Jp + stepJ, 8);
                }
            }
            const float a = nx - inx, b = ny - iny;
            const int iw00 = d_cv_round((1.f - a) * (1.f - b) * (1 << LK_W_BITS));
            const int iw01 = d_cv_round(a * (1.f - b) * (1 << LK_W_BITS));
            const int iw10 = d_cv_round((1.f - a) * b * (1 << LK_W_BITS));
            const int iw11 = (1 << LK_W_BITS) - iw00 - iw01 - iw10;
            int pb[2] = {0, 0};
#pragma unroll
            for (int k = 0; k < LK_RS_SEG; ++k) {
                const int s00 = (int)((j0 >> (8 * k)) & 0xFF), s01 = (int)((j0 >> (8 * k + 8)) & 0xFF);
                const int s10 = (int)((j1 >> (8 * k)) & 0xFF), s11 = (int)((j1 >> (8 * k + 8)) & 0xFF);
                const int diff = ((lk_blend_u8(s00, s01, s10, s11, iw00, iw01, iw10, iw11) + (1 << (LK_W_BITS - 5 - 1))) >> (LK_W_BITS - 5)) - Iv[k];
                pb[0] += diff * Ixv[k]; pb[1] += diff * Iyv[k];
            }
            float fb[2];
            lk_lds_sums<2>(acc, pb, fb, FLT_SCALE);
            const float b1 = fb[0], b2 = fb[1];
            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;
            nx += dx; ny += dy;
            next_pt.x = nx + half; next_pt.y = ny + half;
            if ((double)dx * dx + (double)dy * dy <= epsilon) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                next_pt.x -= dx * 0.5f; next_pt.y -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        total_it += n_it;
    }
    return total_it;
}

// LK variants (run-time: LVK_LK_VARIANT, read once per process; A/B records in profiles/):
//   0  lk_point_rs21      wave-wide sums by DPP + v_readlane + scalar carry chain (rounds 3-5)
//   1  lk_point_rs21_lds  sums through LDS accumulators, next-image bytes kept while the window origin stands
//   2  k_fe_lk_pipe (frontend.hip): variant 1's arithmetic with the levels' templates built by three more wavefronts of the block
// (tried and dropped, profiles/r6_e_*: sums by DPP + one FP64 MFMA; touching the next image's lines of all levels at the start of a pass)
// LVK_LK_GENERIC (compile-time): the one-pixel-per-lane-slot path for every window size
#define LVK_LK_VARIANTS 3
template <int WIN, int VAR, int PASS = 0>
__device__ __forceinline__ int lk_point(const PyrView& prev, const PyrView& next, int n_levels, lvk_pt2f prev_pt, lvk_pt2f& next_pt,
                                        int& status, int max_count, double epsilon, int* __restrict__ iters_out, LkLdsAcc& acc)
{
#ifndef LVK_LK_GENERIC
    if constexpr (WIN == 21 && VAR >= 1) return lk_point_rs21_lds<PASS>(prev, next, n_levels, prev_pt, next_pt, status, max_count, epsilon, iters_out, acc);
    else if constexpr (WIN == 21) return lk_point_rs21(prev, next, n_levels, prev_pt, next_pt, status, max_count, epsilon, iters_out);
    else
#endif
    return lk_point_generic<WIN>(prev, next, n_levels, prev_pt, next_pt, status, max_count, epsilon, iters_out);
}
// zero the wavefront's LDS accumulators (4 x u64, 16-byte aligned; call once, by the wavefront that will use them)
__device__ __forceinline__ LkLdsAcc lk_acc_init(unsigned long long* lds4)
{
    LkLdsAcc a; a.prev[0] = a.prev[1] = a.prev[2] = 0;
    a.off = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned long long*)lds4;
    if ((threadIdx.x & 63) < 4) lds4[threadIdx.x & 63] = 0;
    __builtin_amdgcn_wave_barrier();
    return a;
}

// ------------------------------------------------------------------------- ORB
#include "orb_pattern_dev.inc"     // __constant__ int8_t k_orb_pattern[1024]
// umax of a radius-15 disc (ORBDescriptor.cpp:313-328 evaluated; checked against the formula in tests)
static __constant__ int8_t k_orb_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

__device__ __forceinline__ float d_fast_atan2(float y, float x)
{   // cv::fastAtan2 scalar polynomial, degrees
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// one wavefront: IC angle on ext, 256 rotated tests on blur -> 4 x u64 descriptor (all lanes get it).
__device__ __forceinline__ float orb_point(const uint8_t* __restrict__ ext, const uint8_t* __restrict__ blur, int step, lvk_pt2f pt,
                                           unsigned long long d[4])
{
    const int lane = threadIdx.x & 63, B = LVK_ORB_BORDER;
    const int cx = d_cv_round(pt.x * 1.0f), cy = d_cv_round(pt.y * 1.0f);
    const uint8_t* center = ext + (ptrdiff_t)(cy + B) * step + cx + B;
    int m10 = 0, m01 = 0;
    // The 31 x 31 square around the point, 16 pixels per lane, ALL loads issued before the first is used: as a loop with the disc test
    // (a table read) in front of each pixel read it was two dependent memory round trips per trip, fifteen trips - 6-9 us, the longest
    // chain of the whole LK launch (the descriptor wavefront, not the reverse pass, was what the block waited for; round 6 spans).
    // The half-widths of the disc's rows (k_orb_umax) come from a packed constant; pixels outside the disc are read and not counted.
    const unsigned long long umax_nibbles = 0x3689ABCDDEEEFFFFull;            // nibble |v| = umax[|v|] = {15,15,15,15,14,14,14,13,13,12,11,10,9,8,6,3}
    int val[16], uu[16], vv[16]; bool in[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int pidx = lane + 64 * k;
        const int pc = pidx < 31 * 31 ? pidx : 31 * 31 - 1;
        const int vrow = pc / 31;
        vv[k] = vrow - 15; uu[k] = pc - vrow * 31 - 15;
        const int av = vv[k] < 0 ? -vv[k] : vv[k], au = uu[k] < 0 ? -uu[k] : uu[k];
        in[k] = pidx < 31 * 31 && au <= (int)((umax_nibbles >> (4 * av)) & 0xF);
        val[k] = center[vv[k] * step + uu[k]];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) { m10 += in[k] ? uu[k] * val[k] : 0; m01 += in[k] ? vv[k] * val[k] : 0; }
    m10 = wave_sum_i32(m10); m01 = wave_sum_i32(m01);
    const float angle = d_fast_atan2((float)m01, (float)m10);
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float ang = angle * factorPI;
    float a, b;
    lvk_sincosf(ang, &a, &b);                             // cosf / sinf as the reference's libm computes them (lvk_sincosf.h; ORBDescriptor.cpp:343)
    const uint8_t* bc = blur + (ptrdiff_t)(cy + B) * step + cx + B;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int T = s * 64 + lane;
        const int x0 = k_orb_pattern[4 * T], y0 = k_orb_pattern[4 * T + 1], x1 = k_orb_pattern[4 * T + 2], y1 = k_orb_pattern[4 * T + 3];
        float fx0 = x0 * a - y0 * b, fy0 = x0 * b + y0 * a;
        float fx1 = x1 * a - y1 * b, fy1 = x1 * b + y1 * a;
        int t0 = bc[d_cv_round(fy0) * step + d_cv_round(fx0)];
        int t1 = bc[d_cv_round(fy1) * step + d_cv_round(fx1)];
        d[s] = __ballot(t0 < t1);
    }
    return angle;
}

__device__ __forceinline__ int hamming256(const uint32_t* a, const uint32_t* b)
{
    int dist = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) dist += __popc(a[i] ^ b[i]);
    return dist;
}
__device__ __forceinline__ int hamming256_u64(const unsigned long long a[4], const unsigned long long* b)
{
    return __popcll(a[0] ^ b[0]) + __popcll(a[1] ^ b[1]) + __popcll(a[2] ^ b[2]) + __popcll(a[3] ^ b[3]);
}

// ------------------------------------------------------------------------- undistortion
// [cv::undistortPoints 5 iterations | cv::fisheye::undistortPoints Newton] double inside, Point2f out.
__device__ __forceinline__ lvk_pt2f undistort_point(lvk_pt2f in, const CamParams& cam, const double ni[4])
{
    const double fx = cam.intr[0], fy = cam.intr[1], cx = cam.intr[2], cy = cam.intr[3];
    const double RR00 = ni[0], RR01 = 0.0, RR02 = ni[2], RR10 = 0.0, RR11 = ni[1], RR12 = ni[3], RR20 = 0.0, RR21 = 0.0, RR22 = 1.0;
    lvk_pt2f o;
    if (cam.model == 0) {
        const double ifx = 1. / fx, ify = 1. / fy;
        const double k0 = cam.dist[0], k1 = cam.dist[1], p1 = cam.dist[2], p2 = cam.dist[3];
        double x = in.x, y = in.y;
        const double u = x, v = y;
        x = (x - cx) * ifx; y = (y - cy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; ++j) {
            double r2 = x * x + y * y;
            double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((0. * r2 + k1) * r2 + k0) * r2);
            if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
            double dX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x) + 0. * r2 + 0. * r2 * r2;
            double dY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y + 0. * r2 + 0. * r2 * r2;
            x = (x0 - dX) * icdist;
            y = (y0 - dY) * icdist;
        }
        double xx = RR00 * x + RR01 * y + RR02;
        double yy = RR10 * x + RR11 * y + RR12;
        double ww = 1. / (RR20 * x + RR21 * y + RR22);
        o.x = (float)(xx * ww); o.y = (float)(yy * ww);
    } else {
        const double PI_2 = 3.1415926535897932384626433832795 / 2.;
        double pwx = ((double)in.x - cx) / fx, pwy = ((double)in.y - cy) / fy;
        double scale = 1.0;
        double theta_d = sqrt(pwx * pwx + pwy * pwy);
        theta_d = fmin(fmax(-PI_2, theta_d), PI_2);
        if (theta_d > 1e-8) {
            double theta = theta_d;
            for (int j = 0; j < 10; ++j) {
                double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
                double k0t2 = cam.dist[0] * t2, k1t4 = cam.dist[1] * t4, k2t6 = cam.dist[2] * t6, k3t8 = cam.dist[3] * t8;
                double fix = (theta * (1 + k0t2 + k1t4 + k2t6 + k3t8) - theta_d) / (1 + 3 * k0t2 + 5 * k1t4 + 7 * k2t6 + 9 * k3t8);
                theta = theta - fix;
                if (fabs(fix) < 1e-8) break;
            }
            scale = tan(theta) / theta_d;
        }
        double pux = pwx * scale, puy = pwy * scale;
        double prx = RR00 * pux + RR01 * puy + RR02 * 1.0;
        double pry = RR10 * pux + RR11 * puy + RR12 * 1.0;
        double prz = RR20 * pux + RR21 * puy + RR22 * 1.0;
        o.x = (float)(prx / prz); o.y = (float)(pry / prz);
    }
    return o;
}

// ------------------------------------------------------------------------- fundamental matrix
#ifdef LVK_FM_TIMING
static __device__ unsigned long long g_fm_tick[32];
static __device__ int g_fm_on = 1;
#define FM_TICK(k) do { if (threadIdx.x == 0 && g_fm_on) g_fm_tick[k] = wall_clock64(); } while (0)
#else
#define FM_TICK(k) do { } while (0)
#endif
#define FM_THREADS 256
#define FM_THREADS_WIDE 512    // workgroup for large point sets (configs[4]: 2000 tracks): the per-point loops (compaction, scoring, mask) are
#define FM_WIDE_FROM 512       // what the call costs there; the hypothesis machinery stays on the first FM_THREADS threads
#define FM_MAX_N 4096
#define FM_ROUND 16            // hypotheses solved and scored per round
#define FM_FIRST 2             // ... in the first round and
#define FM_SECOND 6            // ... in the second (LARVIO's sets: 1 iteration in ~60 % of the calls, 2 in ~37 %, 3+ in ~3 %)

__device__ inline void d_nullspace_7x9(const double* A, double* f1, double* f2)
{   // Householder QR of A^T; last two columns of Q (same operation order as the oracle)
    double M[9][7], V[7][9], beta[7];
    for (int r = 0; r < 9; ++r) for (int c = 0; c < 7; ++c) M[r][c] = A[c * 9 + r];
    for (int k = 0; k < 7; ++k) {
        double nrm2 = 0.;
        for (int r = k; r < 9; ++r) nrm2 += M[r][k] * M[r][k];
        double nrm = sqrt(nrm2);
        for (int r = 0; r < 9; ++r) V[k][r] = 0.;
        if (nrm == 0.) { beta[k] = 0.; continue; }
        double alpha = M[k][k] >= 0. ? -nrm : nrm;
        double v0 = M[k][k] - alpha;
        V[k][k] = v0;
        for (int r = k + 1; r < 9; ++r) V[k][r] = M[r][k];
        double vnorm2 = v0 * v0;
        for (int r = k + 1; r < 9; ++r) vnorm2 += M[r][k] * M[r][k];
        beta[k] = vnorm2 == 0. ? 0. : 2. / vnorm2;
        for (int c = k; c < 7; ++c) {
            double s = 0.;
            for (int r = k; r < 9; ++r) s += V[k][r] * M[r][c];
            s *= beta[k];
            for (int r = k; r < 9; ++r) M[r][c] -= s * V[k][r];
        }
    }
    for (int j = 7; j < 9; ++j) {
        double q[9];
        for (int r = 0; r < 9; ++r) q[r] = (r == j) ? 1. : 0.;
        for (int k = 6; k >= 0; --k) {
            double s = 0.;
            for (int r = k; r < 9; ++r) s += V[k][r] * q[r];
            s *= beta[k];
            for (int r = k; r < 9; ++r) q[r] -= s * V[k][r];
        }
        double* f = (j == 7) ? f1 : f2;
        for (int r = 0; r < 9; ++r) f[r] = q[r];
    }
}

__device__ inline int d_solve_cubic(const double* coef, double* roots)
{   // cv::solveCubic, 1x4 form
    double a0 = coef[0], a1 = coef[1], a2 = coef[2], a3 = coef[3];
    double x0 = 0., x1 = 0., x2 = 0.;
    int n = 0;
    const double PI = 3.1415926535897932384626433832795;
    if (a0 == 0) {
        if (a1 == 0) {
            if (a2 == 0) n = a3 == 0 ? -1 : 0;
            else { x0 = -a3 / a2; n = 1; }
        } else {
            double d = a2 * a2 - 4 * a1 * a3;
            if (d >= 0) {
                d = sqrt(d);
                double q1 = (-a2 + d) * 0.5;
                double q2 = (a2 + d) * -0.5;
                if (fabs(q1) > fabs(q2)) { x0 = q1 / a1; x1 = a3 / q1; }
                else { x0 = q2 / a1; x1 = a3 / q2; }
                n = d > 0 ? 2 : 1;
            }
        }
    } else {
        a0 = 1. / a0; a1 *= a0; a2 *= a0; a3 *= a0;
        double Q = (a1 * a1 - 3 * a2) * (1. / 9);
        double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
        double Qcubed = Q * Q * Q;
        double d = Qcubed - R * R;
        if (d > 0) {
            double theta = acos(R / sqrt(Qcubed));
            double sqrtQ = sqrt(Q);
            double t0 = -2 * sqrtQ, t1 = theta * (1. / 3), t2 = a1 * (1. / 3);
            x0 = t0 * cos(t1) - t2;
            x1 = t0 * cos(t1 + (2. * PI / 3)) - t2;
            x2 = t0 * cos(t1 + (4. * PI / 3)) - t2;
            n = 3;
        } else if (d == 0) {
            if (R >= 0) { x0 = -2 * pow(R, 1. / 3) - a1 / 3; x1 = pow(R, 1. / 3) - a1 / 3; }
            else { x0 = 2 * pow(-R, 1. / 3) - a1 / 3; x1 = -pow(-R, 1. / 3) - a1 / 3; }
            x2 = 0;
            n = x0 == x1 ? 1 : 2;
            x1 = x0 == x1 ? 0 : x1;
        } else {
            double e;
            d = sqrt(-d);
            e = pow(d + fabs(R), 1. / 3);
            if (R > 0) e = -e;
            x0 = (e + Q / e) - a1 * (1. / 3);
            n = 1;
        }
    }
    roots[0] = x0; roots[1] = x1; roots[2] = x2;
    return n;
}

__device__ inline int d_7pt_finish(double* f1, double* f2, double* fmatrix);
__device__ inline int d_fundamental_7pt(const lvk_pt2f* m1, const lvk_pt2f* m2, double* fmatrix)
{   // run7Point
    double a[7 * 9], f1[9], f2[9];
    for (int i = 0; i < 7; ++i) {
        double x0 = m1[i].x, y0 = m1[i].y, x1 = m2[i].x, y1 = m2[i].y;
        a[i * 9 + 0] = x1 * x0; a[i * 9 + 1] = x1 * y0; a[i * 9 + 2] = x1;
        a[i * 9 + 3] = y1 * x0; a[i * 9 + 4] = y1 * y0; a[i * 9 + 5] = y1;
        a[i * 9 + 6] = x0; a[i * 9 + 7] = y0; a[i * 9 + 8] = 1;
    }
    d_nullspace_7x9(a, f1, f2);
    return d_7pt_finish(f1, f2, fmatrix);
}

// second half of run7Point: det(lambda*f1 + (1-lambda)*f2) = 0, up to three models
__device__ inline int d_7pt_finish(double* f1, double* f2, double* fmatrix)
{
    double c[4], r[3] = {0, 0, 0};
    for (int i = 0; i < 9; ++i) f1[i] -= f2[i];
    double t0 = f2[4] * f2[8] - f2[5] * f2[7];
    double t1 = f2[3] * f2[8] - f2[5] * f2[6];
    double t2 = f2[3] * f2[7] - f2[4] * f2[6];
    c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
    c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 -
           f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) +
           f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) -
           f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) +
           f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) -
           f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
           f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    t0 = f1[4] * f1[8] - f1[5] * f1[7];
    t1 = f1[3] * f1[8] - f1[5] * f1[6];
    t2 = f1[3] * f1[7] - f1[4] * f1[6];
    c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
    c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 -
           f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) +
           f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) -
           f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) +
           f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) -
           f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
           f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
    int n = d_solve_cubic(c, r);
    if (n < 1 || n > 3) return 0;
    for (int k = 0; k < n; ++k, fmatrix += 9) {
        double lambda = r[k], mu = 1.;
        double s = f1[8] * r[k] + f2[8];
        if (fabs(s) > DBL_EPSILON) { mu = 1. / s; lambda *= mu; fmatrix[8] = 1.; }
        else fmatrix[8] = 0.;
        for (int i = 0; i < 8; ++i) fmatrix[i] = f1[i] * lambda + f2[i] * mu;
    }
    return n;
}

__device__ __forceinline__ float d_fm_error(lvk_pt2f p1, lvk_pt2f p2, const double* F)
{   // FMEstimatorCallback::computeError, one point
    double a, b, c, d1, d2, s1, s2;
    a = F[0] * p1.x + F[1] * p1.y + F[2];
    b = F[3] * p1.x + F[4] * p1.y + F[5];
    c = F[6] * p1.x + F[7] * p1.y + F[8];
    s2 = 1. / (a * a + b * b);
    d2 = p2.x * a + p2.y * b + c;
    a = F[0] * p2.x + F[3] * p2.y + F[6];
    b = F[1] * p2.x + F[4] * p2.y + F[7];
    c = F[2] * p2.x + F[5] * p2.y + F[8];
    s1 = 1. / (a * a + b * b);
    d1 = p1.x * a + p1.y * b + c;
    double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (float)(e1 > e2 ? e1 : e2);
}

struct d_rng { unsigned long long state; };
__device__ __forceinline__ unsigned d_rng_next(d_rng& r)
{
    r.state = (unsigned long long)(unsigned)r.state * 4164903690ULL + (unsigned)(r.state >> 32);
    return (unsigned)r.state;
}
__device__ __forceinline__ int d_rng_uniform(d_rng& r, int a, int b) { return a == b ? a : (int)(d_rng_next(r) % (unsigned)(b - a) + a); }

// cv::findFundamentalMat seeds a fresh cv::RNG(-1) on every call, so the raw 32-bit draws of one call are always the same
// sequence: FM_RNG.s[k] is the generator state after k+1 draws (draw k = its low word).  Built at compile time.
#define FM_RNG_DRAWS 4096
struct FmRngTable {
    unsigned long long s[FM_RNG_DRAWS];
    constexpr FmRngTable() : s{} {
        unsigned long long st = ~0ULL;
        for (int k = 0; k < FM_RNG_DRAWS; ++k) { st = (unsigned long long)(unsigned)st * 4164903690ULL + (unsigned)(st >> 32); s[k] = st; }
    }
};
static __device__ const FmRngTable FM_RNG = FmRngTable();
#define FM_WIN FM_THREADS      // draws looked at per round (16 subsets use ~115 of them)
#define FM_NOFIT 0xffffu

__device__ inline bool d_have_collinear(const lvk_pt2f* ptr, int count)
{
    int i = count - 1;
    for (int j = 0; j < i; ++j) {
        double dx1 = ptr[j].x - ptr[i].x, dy1 = ptr[j].y - ptr[i].y;
        for (int k = 0; k < j; ++k) {
            double dx2 = ptr[k].x - ptr[i].x, dy2 = ptr[k].y - ptr[i].y;
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return true;
        }
    }
    return false;
}

__device__ inline bool d_get_subset(const lvk_pt2f* m1, const lvk_pt2f* m2, int count, lvk_pt2f* ms1, lvk_pt2f* ms2, d_rng& rng, int max_attempts)
{
    int idx[7], iters = 0;
    for (; iters < max_attempts; ++iters) {
        for (int i = 0; i < 7; ++i) {
            int idx_i;
            for (;;) {
                idx_i = d_rng_uniform(rng, 0, count);
                bool dup = false;
                for (int q = 0; q < i; ++q) if (idx[q] == idx_i) { dup = true; break; }
                if (!dup) break;
            }
            idx[i] = idx_i;
            ms1[i] = m1[idx_i]; ms2[i] = m2[idx_i];
        }
        if (!d_have_collinear(ms1, 7) && !d_have_collinear(ms2, 7)) break;
    }
    return iters < max_attempts;
}

// RANSACUpdateNumIters in two halves: everything that does not depend on the current iteration bound (pow, two logs, the quotient)
// and the comparison against the bound.  The first half runs for all models of a round in parallel, the sequential replay of the
// "better model -> fewer iterations" rule only applies the second.
struct FmIterRule { double num, denom; int q, zero; };
__device__ inline FmIterRule d_ransac_iter_rule(double p, double ep, int model_points)
{
    FmIterRule o;
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, (double)model_points);
    o.zero = denom < DBL_MIN; o.q = 0; o.num = 0.; o.denom = 0.;
    if (o.zero) return o;
    o.num = log(num); o.denom = log(denom);
    if (o.denom < 0) o.q = (int)rint(o.num / o.denom);
    return o;
}
__device__ __forceinline__ int d_ransac_iter_apply(const FmIterRule& u, int max_iters)
{
    if (u.zero) return 0;
    return u.denom >= 0 || -u.num >= max_iters * (-u.denom) ? max_iters : u.q;
}
__device__ inline int d_ransac_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    return d_ransac_iter_apply(d_ransac_iter_rule(p, ep, model_points), max_iters);
}

// Order-preserving compaction inside an NT-thread workgroup: exclusive rank of this thread's flag (wavefront ballots + one LDS
// word per wavefront) and the total.  Two barriers; wtot has NT/64 words and is free again on return.
template <int NT>
__device__ __forceinline__ int block_rank(bool flag, int* wtot, int& total)
{
    const unsigned long long b = __ballot(flag);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int within = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wtot[w] = __popcll(b);
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) { const int v = wtot[k]; tot += v; before += k < w ? v : 0; }
    __syncthreads();
    total = tot;
    return before + within;
}

// d_fundamental_7pt for the hypotheses of one RANSAC round, eight lanes per hypothesis: lane c owns column c of A^T (the nine
// monomials of correspondence c) in registers; reflector k is built by lane k, published through LDS and applied by lanes
// c >= k to their own column; lanes 0/1 back-accumulate the two null vectors; lane 0 finishes.  Every sum runs in the order
// of d_nullspace_7x9, so the models are the same bits.  Called by the whole workgroup (barriers inside).
struct FmSolveLds { double V[FM_ROUND][7][9]; double beta[FM_ROUND][7]; double f12[FM_ROUND][2][9]; };
__device__ inline void fm_solve_round(int nr, const lvk_pt2f (*sub1)[7], const lvk_pt2f (*sub2)[7], const int* found,
                                      double (*models)[27], int* nmodels, FmSolveLds& L)
{
    const int t = threadIdx.x, g = t >> 3, c = t & 7;
    const bool act = g < nr && c < 7 && found[g];
    double M[9];
    if (act) {
        const double x0 = sub1[g][c].x, y0 = sub1[g][c].y, x1 = sub2[g][c].x, y1 = sub2[g][c].y;
        M[0] = x1 * x0; M[1] = x1 * y0; M[2] = x1; M[3] = y1 * x0; M[4] = y1 * y0; M[5] = y1; M[6] = x0; M[7] = y0; M[8] = 1;
    } else {
#pragma unroll
        for (int r = 0; r < 9; ++r) M[r] = 0.;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        if (act && c == k) {
            double nrm2 = 0.;
#pragma unroll
            for (int r = k; r < 9; ++r) nrm2 += M[r] * M[r];
            const double nrm = sqrt(nrm2);
            double b = 0.;
#pragma unroll
            for (int r = 0; r < 9; ++r) L.V[g][k][r] = 0.;
            if (nrm != 0.) {
                const double alpha = M[k] >= 0. ? -nrm : nrm;
                const double v0 = M[k] - alpha;
                L.V[g][k][k] = v0;
                double vnorm2 = v0 * v0;
#pragma unroll
                for (int r = k + 1; r < 9; ++r) { L.V[g][k][r] = M[r]; vnorm2 += M[r] * M[r]; }
                b = vnorm2 == 0. ? 0. : 2. / vnorm2;
                // a negative beta marks "norm was zero: skip" for the appliers (2/vnorm2 is never negative)
            } else b = -1.;
            L.beta[g][k] = b;
        }
        __syncthreads();
        if (act && c >= k) {
            const double b = L.beta[g][k];
            if (b >= 0.) {
                double sacc = 0.;
#pragma unroll
                for (int r = k; r < 9; ++r) sacc += L.V[g][k][r] * M[r];
                sacc *= b;
#pragma unroll
                for (int r = k; r < 9; ++r) M[r] -= sacc * L.V[g][k][r];
            }
        }
    }
    FM_TICK(11);
    if (g < nr && c < 2 && found[g]) {
        double q[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) q[r] = (r == 7 + c) ? 1. : 0.;
#pragma unroll
        for (int k = 6; k >= 0; --k) {
            double b = L.beta[g][k];
            b = b < 0. ? 0. : b;
            double sacc = 0.;
#pragma unroll
            for (int r = k; r < 9; ++r) sacc += L.V[g][k][r] * q[r];
            sacc *= b;
#pragma unroll
            for (int r = k; r < 9; ++r) q[r] -= sacc * L.V[g][k][r];
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) L.f12[g][c][r] = q[r];
    }
    __syncthreads();
    FM_TICK(12);
    if (g < nr && c == 0) {
        if (found[g]) {
            double f1[9], f2[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) { f1[r] = L.f12[g][0][r]; f2[r] = L.f12[g][1][r]; }
            nmodels[g] = d_7pt_finish(f1, f2, models[g]);
        } else nmodels[g] = 0;
    }
}

// LMedS (8 <= n < 15): niters from outlier ratio 0.45, one hypothesis per thread per pass (the first FM_THREADS threads).  Kept out
// of line: its per-thread arrays (27-double model block, the sorted errors, the 7x9 null-space solve) are the register hogs of the
// whole call, and inlined they would set the allocation of the 1024-thread RANSAC path (LARVIO's steady state never gets here:
// fewer than 15 alive points).  LDS scratch comes from the caller.
template <int NT>
__device__ __noinline__ int fm_lmeds_block(const lvk_pt2f* s1, const lvk_pt2f* s2, int n, double conf, uint8_t* smask, int* iters_out,
                                           lvk_pt2f (*lsub1)[7], lvk_pt2f (*lsub2)[7], int* lfound, float* lm_med, int* lm_seq,
                                           double* best_model, int* sh_ctl, unsigned long long* sh_rng_p)
{
    const int t = threadIdx.x;
    const int niters = d_ransac_update_num_iters(conf, 0.45, 7, 1000);
    float my_med = FLT_MAX; int my_seq = 0x7fffffff; double my_model[9];
    if (t == 0) *sh_rng_p = ~0ULL;
    __syncthreads();
    int stop_at = niters;      // iteration at which getSubset failed (uniform via LDS)
    for (int base = 0; base < niters; base += FM_THREADS) {
        if (t == 0) {
            d_rng rng; rng.state = *sh_rng_p;
            int failed = 0;
            for (int r = 0; r < FM_THREADS && base + r < niters; ++r) {
                lfound[r] = failed ? 0 : (d_get_subset(s1, s2, n, lsub1[r], lsub2[r], rng, 1000) ? 1 : 0);
                if (!lfound[r] && !failed) { failed = 1; sh_ctl[2] = base + r; }
            }
            *sh_rng_p = rng.state;
            sh_ctl[0] = failed;
        }
        __syncthreads();
        if (t < FM_THREADS && base + t < niters && lfound[t]) {
            double mdl[27];
            int nm = d_fundamental_7pt(lsub1[t], lsub2[t], mdl);
            for (int m = 0; m < nm; ++m) {
                float e[16];
                for (int i = 0; i < n; ++i) e[i] = d_fm_error(s1[i], s2[i], mdl + 9 * m);
                // median = element count/2 of the ascending order (errors are >= 0: int view == float order)
                for (int i = 1; i < n; ++i) { float v = e[i]; int j = i - 1; while (j >= 0 && e[j] > v) { e[j + 1] = e[j]; --j; } e[j + 1] = v; }
                float med = e[n / 2];
                int seq = (base + t) * 3 + m;
                if (med < my_med) { my_med = med; my_seq = seq; for (int q = 0; q < 9; ++q) my_model[q] = mdl[9 * m + q]; }
            }
        }
        int failed = sh_ctl[0];
        if (failed) { stop_at = sh_ctl[2]; __syncthreads(); break; }
        __syncthreads();
    }
    // hypotheses at or after a getSubset failure are not run by OpenCV: none were solved (lfound = 0)
    (void)stop_at;
    if (t < FM_THREADS) { lm_med[t] = my_med; lm_seq[t] = my_seq; }
    __syncthreads();
    if (t == 0) {
        int bi = -1; float bm = FLT_MAX; int bs = 0x7fffffff;
        for (int i = 0; i < FM_THREADS; ++i)
            if (lm_seq[i] != 0x7fffffff && (lm_med[i] < bm || (lm_med[i] == bm && lm_seq[i] < bs))) { bm = lm_med[i]; bs = lm_seq[i]; bi = i; }
        sh_ctl[1] = bi;
    }
    __syncthreads();
    const int bi = sh_ctl[1];
    if (bi < 0) { for (int i = t; i < n; i += NT) smask[i] = 0; __syncthreads(); *iters_out = niters; return 1; }
    if (t == bi) { for (int q = 0; q < 9; ++q) best_model[q] = my_model[q]; }
    __syncthreads();
    {
        double min_median = (double)lm_med[bi];
        double sigma = 2.5 * 1.4826 * (1 + 5. / (n - 7)) * sqrt(min_median);
        sigma = sigma > 0.001 ? sigma : 0.001;
        const float tt = (float)(sigma * sigma);
        for (int i = t; i < n; i += NT) smask[i] = d_fm_error(s1[i], s2[i], best_model) <= tt;
    }
    *iters_out = niters;
    __syncthreads();
    return 1;
}

// The whole of cv::findFundamentalMat(..., FM_RANSAC, thresh, conf, mask) for one point set held in
// LDS, executed by one workgroup of NT threads (NT >= FM_THREADS, a multiple of 64: the per-point loops use all of them, the
// hypothesis machinery the first FM_THREADS).  Returns (uniformly) 1 if smask[0..n) was written.
//   n < 7: nothing;  n == 7: ones;  8..14: LMedS (300 hypotheses, all in parallel);
//   n >= 15 (or force_ransac): RANSAC — rounds of FM_FIRST, FM_SECOND, then FM_ROUND hypotheses: thread 0 draws the subsets in
//   OpenCV's RNG order, eight lanes per hypothesis solve the 7-point systems, all threads score, thread 0 replays
//   the sequential "better model -> shrink niters" rule over the round in order.
template <int NT>
__device__ inline int fm_mask_block(const lvk_pt2f* s1, const lvk_pt2f* s2, int n, double thresh, double conf, int max_iters,
                                    int force_ransac, uint8_t* smask, int* iters_out, double* model_out = nullptr)
{   // model_out (optional, global memory, 9 doubles): the matrix cv::findFundamentalMat RETURNS - the registrator's best minimal-sample
    // model, zeros when it has none (the empty Mat); for n == 7 the first of the solver's models.  nullptr in the frame path.
    __shared__ lvk_pt2f sub1[FM_ROUND][7], sub2[FM_ROUND][7];
    __shared__ double models[FM_ROUND][27];
    __shared__ int nmodels[FM_ROUND], found[FM_ROUND], good[FM_ROUND][3];
    __shared__ FmIterRule rule[FM_ROUND][3];
    __shared__ double best_model[9];
    __shared__ int sh_ctl[4];                 // [0] stop, [1] have best, [2] iterations, [3] niters
    __shared__ unsigned long long sh_rng;
    __shared__ unsigned short idxw[FM_WIN], fw[FM_WIN], selw[FM_WIN][7];
    __shared__ int off[FM_ROUND + 1];
    __shared__ FmSolveLds solve_lds;
    __shared__ int sh_pos, sh_serial, sh_anybad;
    __shared__ float lm_med[FM_THREADS]; __shared__ int lm_seq[FM_THREADS];
    const int t = threadIdx.x;
    *iters_out = 0;
    if (n < 7) return 0;
    if (model_out && t < 9) model_out[t] = 0.;
    if (n == 7) {
        for (int i = t; i < n; i += NT) smask[i] = 1;
        if (model_out) {                                   // (stage-level call only)
            __shared__ double m7[27]; __shared__ int nm7;
            if (t == 0) { lvk_pt2f a[7], b[7]; for (int i = 0; i < 7; ++i) { a[i] = s1[i]; b[i] = s2[i]; } nm7 = d_fundamental_7pt(a, b, m7); }
            __syncthreads();
            if (nm7 > 0 && t < 9) model_out[t] = m7[t];
        }
        __syncthreads(); return 1;
    }
    if (thresh <= 0) thresh = 3;
    if (conf < DBL_EPSILON || conf > 1 - DBL_EPSILON) conf = 0.99;

    if (n >= 15 || force_ransac) {
        const float tthr = (float)(thresh * thresh);
        if (t == 0) { sh_ctl[0] = 0; sh_ctl[1] = 0; sh_ctl[2] = 0; sh_ctl[3] = max_iters > 1 ? max_iters : 1; sh_rng = ~0ULL; sh_pos = 0; sh_serial = 0; sh_anybad = 0; }
        __syncthreads();
        int max_good = 0;       // thread 0 only
        int base = 0;
        for (int round = 0;; ++round) {
            // LARVIO's point sets are mostly inliers after the LK and descriptor gates: the first hypotheses usually end the loop
            // (niters collapses to a handful), so the first round is short
            const int nr = round == 0 ? FM_FIRST : round == 1 ? FM_SECOND : FM_ROUND;
            FM_TICK(4);
            // ---- the round's FM_ROUND subsets, in OpenCV's RNG order.  Fast path: every thread maps one table draw to a point
            // index, then works out the subset that WOULD start at its draw (7 distinct indices, duplicates redrawn one at a
            // time as getSubset does); thread 0 only chases the 16 start offsets.  A collinear subset (redraw of all 7) or a
            // window overrun drops to the sequential path for the rest of the call.
            const int pos = sh_pos;
            bool serial = sh_serial != 0;
            if (!serial && pos + FM_WIN > FM_RNG_DRAWS) {
                serial = true;
                if (t == 0) { sh_rng = pos ? FM_RNG.s[pos - 1] : ~0ULL; sh_serial = 1; }
            }
            if (!serial) {
                if (t < FM_WIN) idxw[t] = (unsigned short)((unsigned)FM_RNG.s[pos + t] % (unsigned)n);
                if (t == 0) sh_anybad = 0;
                __syncthreads();
                if (t < FM_WIN) {
                    int sel[7] = {-1, -1, -1, -1, -1, -1, -1};
                    int c = 0, q = t;
                    while (c < 7 && q < FM_WIN) {
                        const int v = idxw[q++];
                        bool dup = false;
#pragma unroll
                        for (int k = 0; k < 7; ++k) dup |= sel[k] == v;
                        if (!dup) {
#pragma unroll
                            for (int k = 0; k < 7; ++k) if (k == c) sel[k] = v;
                            ++c;
                        }
                    }
                    fw[t] = c == 7 ? (unsigned short)(q - t) : (unsigned short)FM_NOFIT;
#pragma unroll
                    for (int k = 0; k < 7; ++k) selw[t][k] = (unsigned short)sel[k];
                }
                __syncthreads();
                if (t == 0) {
                    int o = 0;
                    for (int r = 0; r < nr; ++r) {
                        if (o >= FM_WIN || fw[o] == FM_NOFIT) { sh_anybad = 1; break; }
                        off[r] = o; o += fw[o];
                    }
                    off[nr] = o;
                }
                __syncthreads();
                if (!sh_anybad) {
                    if (t < nr * 7) {
                        const int r = t / 7, i = t - 7 * r, k = selw[off[r]][i];
                        sub1[r][i] = s1[k]; sub2[r][i] = s2[k];
                    }
                    if (t < nr) found[t] = 1;
                    __syncthreads();
                    if (t < nr * 15) {            // haveCollinearPoints: the last point against every pair of the others
                        const int r = t / 15, pr = t - 15 * r;
                        int j = 1, k = pr;
                        while (k >= j) { k -= j; ++j; }  // pr -> (j, k), 0 <= k < j <= 5
                        bool bad = false;
                        for (int im = 0; im < 2; ++im) {
                            const lvk_pt2f* ptr = im ? sub2[r] : sub1[r];
                            double dx1 = ptr[j].x - ptr[6].x, dy1 = ptr[j].y - ptr[6].y;
                            double dx2 = ptr[k].x - ptr[6].x, dy2 = ptr[k].y - ptr[6].y;
                            bad |= fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2));
                        }
                        if (bad) sh_anybad = 1;
                    }
                    __syncthreads();
                }
                serial = sh_anybad != 0;
                if (serial) { if (t == 0) { sh_rng = pos ? FM_RNG.s[pos - 1] : ~0ULL; sh_serial = 1; } }
                else if (t == 0) sh_pos = pos + off[nr];
                __syncthreads();
            }
            if (serial && t == 0) {
                d_rng rng; rng.state = sh_rng;
                for (int r = 0; r < nr; ++r) found[r] = d_get_subset(s1, s2, n, sub1[r], sub2[r], rng, 10000) ? 1 : 0;
                sh_rng = rng.state;
            }
            if (t < nr * 3) good[t / 3][t % 3] = 0;
            __syncthreads();
            FM_TICK(5);
            fm_solve_round(nr, sub1, sub2, found, models, nmodels, solve_lds);
            __syncthreads();
            FM_TICK(6);
            {   // inlier counts: (model slot, point) pairs dealt round-robin to the threads; integer counts, order-free
                int slot = 0, i = t;
                const int nslots = nr * 3;
                while (i >= n) { i -= n; ++slot; }
                while (slot < nslots) {
                    const int r = slot / 3, m = slot - 3 * r;
                    if (m < nmodels[r] && d_fm_error(s1[i], s2[i], &models[r][9 * m]) <= tthr) atomicAdd(&good[r][m], 1);
                    i += NT;
                    while (i >= n) { i -= n; ++slot; }
                }
            }
            __syncthreads();
            if (t < nr * 3) {                     // the iteration rule of every model that could become the best one
                const int r = t / 3, m = t - 3 * r;
                if (m < nmodels[r] && good[r][m] > 6) rule[r][m] = d_ransac_iter_rule(conf, (double)(n - good[r][m]) / n, 7);
            }
            __syncthreads();
            FM_TICK(7);
            if (t == 0) {
                // replay of RANSACPointSetRegistrator::run's loop over this round, in order
                int niters = sh_ctl[3], iter = base;
                bool stop = false;
                for (int r = 0; r < nr; ++r, ++iter) {
                    if (iter >= niters) { stop = true; break; }
                    if (!found[r]) { stop = true; break; }            // getSubset failed: iter==0 -> no model, else stop
                    for (int m = 0; m < nmodels[r]; ++m) {
                        int g = good[r][m];
                        if (g > (max_good > 6 ? max_good : 6)) {
                            for (int q = 0; q < 9; ++q) best_model[q] = models[r][9 * m + q];
                            max_good = g; sh_ctl[1] = 1;
                            niters = d_ransac_iter_apply(rule[r][m], niters);
                        }
                    }
                }
                if (!stop && iter >= niters) stop = true;
                sh_ctl[0] = stop ? 1 : 0; sh_ctl[2] = iter; sh_ctl[3] = niters;
            }
            __syncthreads();
            FM_TICK(8);
            if (sh_ctl[0]) break;
            base += nr;
        }
        // iterations drawn = value of `iter` when OpenCV's loop exits
        if (sh_ctl[1]) { for (int i = t; i < n; i += NT) smask[i] = d_fm_error(s1[i], s2[i], best_model) <= tthr; }
        else { for (int i = t; i < n; i += NT) smask[i] = 0; }
        *iters_out = sh_ctl[2];
        if (model_out && sh_ctl[1] && t < 9) model_out[t] = best_model[t];
        __syncthreads();
        return 1;
    }

    // ---- LMedS (8 <= n < 15)
    __shared__ lvk_pt2f lsub1[FM_THREADS][7], lsub2[FM_THREADS][7];
    __shared__ int lfound[FM_THREADS];
    const int wrote = fm_lmeds_block<NT>(s1, s2, n, conf, smask, iters_out, lsub1, lsub2, lfound, lm_med, lm_seq, best_model, sh_ctl, &sh_rng);
    if (model_out) { __syncthreads(); if (wrote && sh_ctl[1] >= 0 && t < 9) model_out[t] = best_model[t]; }      // (sh_ctl[1] = the winning thread, -1: none)
    return wrote;
}
