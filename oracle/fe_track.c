/*
 * fe_track.c — ORACLE (test infrastructure, see lvo.h): per-point stages of the front-end:
 * pyramidal LK, ORB angle/descriptor/Hamming, undistortion, fundamental-matrix RANSAC/LMedS,
 * gyro-predicted homography.  The ORB block is PINNED to src/ORBDescriptor.cpp compiled in place
 * (tests/test_oracle_ref_orb.py); LK / findFundamentalMat restate OpenCV's published algorithms and stay
 * PARITY UNPINNED against reference outputs (OpenCV is not vendored; see lvo.h, "PINNING").
 */
#include "lvo.h"
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

static inline int cv_round_f(float v) { return (int)rintf(v); }
static inline int cv_round_d(double v) { return (int)rint(v); }
static inline int cv_floor_f(float v) { return (int)floorf(v); }

/* ======================================================================== pyramidal LK
 * [upstream lkpyramid.cpp: calcOpticalFlowPyrLK + LKTrackerInvoker::operator()], scalar path.
 * Deviation fixed by the oracle: OpenCV accumulates A11/A12/A22 and b1/b2 in float32 (or in
 * SIMD int32 pairs, build-dependent), so its low bits depend on the SIMD width.  The oracle
 * accumulates the same integer products EXACTLY in int64 and converts once, which is the
 * correctly-rounded value of the sum every OpenCV build approximates, and is independent of
 * summation order (so a wave-parallel reduction reproduces it bit-for-bit). */
#define W_BITS 14
/* Sensitivity probe (tests/test_oracle_frontend.py): 1 = accumulate A11/A12/A22 and b1/b2 the way OpenCV's scalar (non-SIMD)
 * LKTrackerInvoker does - float32 running sums, pixel by pixel in row-major order - instead of exactly.  Not used by any parity
 * test: it measures how far a float-accumulating OpenCV build can be from the exact sums the product reproduces. */
int lvo_lk_float_accum_ = 0;
void lvo_set_lk_float_accum(int on) { lvo_lk_float_accum_ = on ? 1 : 0; }
void lvo_lk_track(const lvo_pyramid* prev, const lvo_pyramid* next,
                  const lvo_pt2f* prev_pts, lvo_pt2f* next_pts, uint8_t* status, int n,
                  int max_iter, double eps, int* iters_out)
{
    const int win = prev->pad;
    const float half = (win - 1) * 0.5f;
    int n_levels = prev->n_levels < next->n_levels ? prev->n_levels : next->n_levels;
    const int max_level = n_levels - 1;
    int max_count = max_iter < 0 ? 0 : max_iter > 100 ? 100 : max_iter;
    double epsilon = eps < 0. ? 0. : eps > 10. ? 10. : eps;
    epsilon *= epsilon;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float min_eig_threshold = (float)1e-4;     /* calcOpticalFlowPyrLK takes a double, LKTrackerInvoker stores and compares it as a float */
    extern int lvo_threads_;
    short* Iwin_all = (short*)malloc(sizeof(short) * (size_t)win * win * 3 * (size_t)lvo_threads_);
    for (int i = 0; i < n; ++i) status[i] = 1;
    if (iters_out) memset(iters_out, 0, sizeof(int) * (size_t)n * n_levels);

    for (int level = max_level; level >= 0; --level) {
        const int cols = prev->w[level], rows = prev->h[level];
        const int stepI = prev->istride[level], stepJ = next->istride[level], dstep = prev->dstride[level];
        const uint8_t* Ibase = prev->img[level] + (size_t)prev->pad * stepI + prev->pad;
        const uint8_t* Jbase = next->img[level] + (size_t)next->pad * stepJ + next->pad;
        const int16_t* Dbase = prev->der[level] + (size_t)prev->pad * dstep + 2 * prev->pad;
        const float lscale = (float)(1. / (1 << level));
        /* tracks are independent (each has its own template window): the all-core baseline spreads them over threads */
        _Pragma("omp parallel for schedule(dynamic, 4) num_threads(lvo_threads_) if(lvo_threads_ > 1)")
        for (int p = 0; p < n; ++p) {
            short* Iwin = Iwin_all + (size_t)win * win * 3 * (size_t)omp_get_thread_num();
            short* dIwin = Iwin + (size_t)win * win;
            float prx = prev_pts[p].x * lscale, pry = prev_pts[p].y * lscale;
            float nx, ny;
            if (level == max_level) { nx = next_pts[p].x * lscale; ny = next_pts[p].y * lscale; }
            else { nx = next_pts[p].x * 2.f; ny = next_pts[p].y * 2.f; }
            next_pts[p].x = nx; next_pts[p].y = ny;

            prx -= half; pry -= half;
            int ipx = cv_floor_f(prx), ipy = cv_floor_f(pry);
            if (ipx < -win || ipx >= cols || ipy < -win || ipy >= rows) {
                if (level == 0) status[p] = 0;
                continue;
            }
            float a = prx - ipx, b = pry - ipy;
            int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
            int iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int64_t sA11 = 0, sA12 = 0, sA22 = 0;
            float fA11 = 0.f, fA12 = 0.f, fA22 = 0.f;
            for (int y = 0; y < win; ++y) {
                const uint8_t* src = Ibase + (ptrdiff_t)(y + ipy) * stepI + ipx;
                const int16_t* dsrc = Dbase + (ptrdiff_t)(y + ipy) * dstep + 2 * ipx;
                for (int x = 0; x < win; ++x, dsrc += 2) {
                    int ival = (src[x] * iw00 + src[x + 1] * iw01 + src[x + stepI] * iw10 + src[x + stepI + 1] * iw11
                                + (1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
                    int ixval = (dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[dstep] * iw10 + dsrc[dstep + 2] * iw11
                                 + (1 << (W_BITS - 1))) >> W_BITS;
                    int iyval = (dsrc[1] * iw00 + dsrc[3] * iw01 + dsrc[dstep + 1] * iw10 + dsrc[dstep + 3] * iw11
                                 + (1 << (W_BITS - 1))) >> W_BITS;
                    Iwin[y * win + x] = (short)ival;
                    dIwin[2 * (y * win + x)] = (short)ixval;
                    dIwin[2 * (y * win + x) + 1] = (short)iyval;
                    sA11 += (int64_t)(ixval * ixval);
                    sA12 += (int64_t)(ixval * iyval);
                    sA22 += (int64_t)(iyval * iyval);
                    fA11 += (float)(ixval * ixval); fA12 += (float)(ixval * iyval); fA22 += (float)(iyval * iyval);
                }
            }
            float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
            if (lvo_lk_float_accum_) { A11 = fA11 * FLT_SCALE; A12 = fA12 * FLT_SCALE; A22 = fA22 * FLT_SCALE; }
            float D = A11 * A22 - A12 * A12;
            float min_eig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
            if (min_eig < min_eig_threshold || D < FLT_EPSILON) {
                if (level == 0) status[p] = 0;
                continue;
            }
            D = 1.f / D;
            nx -= half; ny -= half;
            float pdx = 0.f, pdy = 0.f;
            int j;
            for (j = 0; j < max_count; ++j) {
                int inx = cv_floor_f(nx), iny = cv_floor_f(ny);
                if (inx < -win || inx >= cols || iny < -win || iny >= rows) {
                    if (level == 0) status[p] = 0;
                    break;
                }
                if (iters_out) iters_out[(size_t)p * n_levels + level]++;
                a = nx - inx; b = ny - iny;
                iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
                iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                int64_t sb1 = 0, sb2 = 0;
                float fb1 = 0.f, fb2 = 0.f;
                for (int y = 0; y < win; ++y) {
                    const uint8_t* Jp = Jbase + (ptrdiff_t)(y + iny) * stepJ + inx;
                    for (int x = 0; x < win; ++x) {
                        int diff = ((Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + stepJ] * iw10 + Jp[x + stepJ + 1] * iw11
                                     + (1 << (W_BITS - 5 - 1))) >> (W_BITS - 5)) - Iwin[y * win + x];
                        sb1 += (int64_t)(diff * dIwin[2 * (y * win + x)]);
                        sb2 += (int64_t)(diff * dIwin[2 * (y * win + x) + 1]);
                        fb1 += (float)(diff * dIwin[2 * (y * win + x)]); fb2 += (float)(diff * dIwin[2 * (y * win + x) + 1]);
                    }
                }
                float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
                if (lvo_lk_float_accum_) { b1 = fb1 * FLT_SCALE; b2 = fb2 * FLT_SCALE; }
                float dx = (A12 * b2 - A22 * b1) * D;
                float dy = (A12 * b1 - A11 * b2) * D;
                nx += dx; ny += dy;
                next_pts[p].x = nx + half; next_pts[p].y = ny + half;
                if ((double)dx * dx + (double)dy * dy <= epsilon) break;
                if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                    next_pts[p].x -= dx * 0.5f; next_pts[p].y -= dy * 0.5f;
                    break;
                }
                pdx = dx; pdy = dy;
            }
        }
    }
    free(Iwin_all);
}

/* ======================================================================== ORB
 * ORBDescriptor.cpp:27-285 (pattern), 304-328 (umax), 335-383, 386-416, 486-514 */
#include "orb_pattern.inc"   /* static const int8_t lvo_orb_pattern[1024] */

float lvo_fast_atan2(float y, float x)
{   /* cv::fastAtan2 scalar path [upstream mathfuncs_core.simd.hpp atan_f32] */
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

/* cosf / sinf as the reference calls them: ORBDescriptor.cpp:343 is `(float)cos(angle), (float)sin(angle)` on a float under
 * `using namespace std;` (:16) - overload resolution picks std::cos(float) / std::sin(float), i.e. libm's cosf / sinf, NOT the double
 * functions (rounds 1-5 restated cos((double)angle) rounded to float here: another float for 0.04 % / 0.09 % of the floats in
 * [0, 2 pi], and - fuzz case 379 of the front-end streams - once in a while a sampling point of the rotated pattern one pixel off).
 * Restated: glibc >= 2.28's algorithm for both (ARM Optimized Routines' sincosf; sysdeps/ieee754/flt-32/s_sincosf.h, s_sinf.c,
 * s_cosf.c, s_sincosf_data.c), plain C with a fixed operation sequence so that the HIP kernel computes the same bits
 * (larvio_amd/csrc/lvk_sincosf.h is the twin).  PINNED against this host's libm over EVERY float in [0, 6.2832]
 * (tests/test_oracle_frontend.py::test_sincosf_restatement_is_libms_on_every_angle: 1,086,918,650 values, zero mismatches). */
static uint32_t sc_abstop12(float x) { uint32_t u; memcpy(&u, &x, 4); return (u >> 20) & 0x7ff; }
static float sc_poly(double x, double x2, int n, int neg)
{
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    if (neg) { C0 = -C0; C1 = -C1; C2 = -C2; C3 = -C3; C4 = -C4; }
    if ((n & 1) == 0) {
        const double x3 = x * x2, s1 = S2 + x2 * S3, x7 = x3 * x2, s = x + x3 * S1;
        return (float)(s + x7 * s1);
    }
    const double x4 = x2 * x2, c2 = C3 + x2 * C4, c1 = C0 + x2 * C1, x6 = x4 * x2, c = c1 + x4 * C2;
    return (float)(c + x6 * c2);
}
static void glibc_sincosf(float y, float* c_out, float* s_out)
{
    const double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;    /* 2/pi * 2^24, pi/2 */
    const double x = y;
    if (sc_abstop12(y) < sc_abstop12(0x1.921FB6p-1f)) {
        const double x2 = x * x;
        const int tiny = sc_abstop12(y) < sc_abstop12(0x1p-12f);
        *s_out = tiny ? y : sc_poly(x, x2, 0, 0);
        *c_out = tiny ? 1.0f : sc_poly(x, x2, 1, 0);
        return;
    }
    const double r = x * HPI_INV;
    const int n = ((int32_t)r + 0x800000) >> 24;
    const double xr = x - n * HPI;
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    *s_out = sc_poly(xr * sgn, xr * xr, n, (n & 2) != 0);
    *c_out = sc_poly(xr * sgn, xr * xr, n ^ 1, ((n + 1) & 2) != 0);
}
/* test hook: the restatement against THIS host's libm on the floats with bit patterns [lo_bits, hi_bits]; returns the number of
 * inputs where either result differs (and the first such bit pattern) */
long lvo_sincosf_sweep(uint32_t lo_bits, uint32_t hi_bits, uint32_t* first_bad)
{
    long bad = 0; uint32_t first = 0xFFFFFFFFu;
#pragma omp parallel for reduction(+ : bad) reduction(min : first) schedule(static)
    for (int64_t b = lo_bits; b <= (int64_t)hi_bits; ++b) {
        const uint32_t u = (uint32_t)b; float x, c, s; memcpy(&x, &u, 4);
        glibc_sincosf(x, &c, &s);
        const float cl = cosf(x), sl = sinf(x);
        if (memcmp(&c, &cl, 4) || memcmp(&s, &sl, 4)) { ++bad; if (u < first) first = u; }
    }
    if (first_bad) *first_bad = first;
    return bad;
}

static void orb_umax(int* umax /*[16]*/)
{   /* ORBDescriptor.cpp:313-328, halfPatchSize 15 */
    const int hp = 15;
    int v, v0, vmax = (int)floor(hp * sqrt(2.f) / 2 + 1);
    int vmin = (int)ceil(hp * sqrt(2.f) / 2);
    const double hp2 = hp * hp;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = hp, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

void lvo_orb_describe(const uint8_t* ext, const uint8_t* blur, int w, int h,
                      const lvo_pt2f* pts, int n, uint8_t* desc, float* angle_out)
{
    (void)h;
    const int B = LVO_ORB_BORDER, step = w + 2 * B, hp = 15;
    int umax[16];
    orb_umax(umax);
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    extern int lvo_threads_;
    _Pragma("omp parallel for schedule(static) num_threads(lvo_threads_) if(lvo_threads_ > 1)")
    for (int i = 0; i < n; ++i) {
        /* IC_Angle (ORBDescriptor.cpp:486-514); pt * mvInvScaleFactor[0] (=1.0f) */
        float px = pts[i].x * 1.0f, py = pts[i].y * 1.0f;
        const uint8_t* center = ext + (ptrdiff_t)(cv_round_f(py) + B) * step + cv_round_f(px) + B;
        int m_01 = 0, m_10 = 0;
        for (int u = -hp; u <= hp; ++u) m_10 += u * center[u];
        for (int v = 1; v <= hp; ++v) {
            int v_sum = 0, d = umax[v];
            for (int u = -d; u <= d; ++u) {
                int vp = center[u + v * step], vm = center[u - v * step];
                v_sum += (vp - vm);
                m_10 += u * (vp + vm);
            }
            m_01 += v * v_sum;
        }
        float angle = lvo_fast_atan2((float)m_01, (float)m_10);
        if (angle_out) angle_out[i] = angle;
        /* computeOrbDescriptor (ORBDescriptor.cpp:335-383); scale = 1/mvLayerScale[0] = 1 */
        float ang = angle * factorPI;
        float a, b;
        glibc_sincosf(ang, &a, &b);
        const uint8_t* bc = blur + (ptrdiff_t)(cv_round_f(pts[i].y * 1.f) + B) * step + cv_round_f(pts[i].x * 1.f) + B;
        const int8_t* pat = lvo_orb_pattern;
        for (int k = 0; k < 32; ++k, pat += 32) {
            int val = 0;
            for (int t = 0; t < 8; ++t) {
                int x0 = pat[4 * t], y0 = pat[4 * t + 1], x1 = pat[4 * t + 2], y1 = pat[4 * t + 3];
                float fx0 = x0 * a - y0 * b, fy0 = x0 * b + y0 * a;
                float fx1 = x1 * a - y1 * b, fy1 = x1 * b + y1 * a;
                int t0 = bc[cv_round_f(fy0) * step + cv_round_f(fx0)];
                int t1 = bc[cv_round_f(fy1) * step + cv_round_f(fx1)];
                val |= (t0 < t1) << t;
            }
            desc[(size_t)i * 32 + k] = (uint8_t)val;
        }
    }
}

int lvo_hamming256(const uint8_t* a, const uint8_t* b)
{   /* ORBDescriptor.h:43-59 */
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4); memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

/* ======================================================================== undistortion
 * image_processor.cpp:1040-1072 -> cv::undistortPoints (5 iterations, [upstream undistort.cpp
 * cvUndistortPointsInternal]) or cv::fisheye::undistortPoints (Newton, <=10 iterations,
 * [upstream fisheye.cpp, 3.4.6+]).  Point2f in/out, double inside. */
void lvo_undistort_points(const lvo_pt2f* in, int n, const double intr[4], int model,
                          const double dist[4], const double new_intr[4], lvo_pt2f* out)
{
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    /* RR = K_new * I */
    const double RR[3][3] = {{new_intr[0], 0.0, new_intr[2]}, {0.0, new_intr[1], new_intr[3]}, {0.0, 0.0, 1.0}};
    if (model == 0) {
        const double ifx = 1. / fx, ify = 1. / fy;
        const double k0 = dist[0], k1 = dist[1], p1 = dist[2], p2 = dist[3];
        for (int i = 0; i < n; ++i) {
            double x = in[i].x, y = in[i].y;
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
            double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
            double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
            double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
            out[i].x = (float)(xx * ww); out[i].y = (float)(yy * ww);
        }
    } else {
        const double PI_2 = 3.1415926535897932384626433832795 / 2.;
        for (int i = 0; i < n; ++i) {
            double pwx = ((double)in[i].x - cx) / fx, pwy = ((double)in[i].y - cy) / fy;
            double scale = 1.0;
            double theta_d = sqrt(pwx * pwx + pwy * pwy);
            theta_d = fmin(fmax(-PI_2, theta_d), PI_2);
            if (theta_d > 1e-8) {
                double theta = theta_d;
                for (int j = 0; j < 10; ++j) {
                    double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
                    double k0t2 = dist[0] * t2, k1t4 = dist[1] * t4, k2t6 = dist[2] * t6, k3t8 = dist[3] * t8;
                    double fix = (theta * (1 + k0t2 + k1t4 + k2t6 + k3t8) - theta_d) /
                                 (1 + 3 * k0t2 + 5 * k1t4 + 7 * k2t6 + 9 * k3t8);
                    theta = theta - fix;
                    if (fabs(fix) < 1e-8) break;
                }
                scale = tan(theta) / theta_d;
            }
            double pux = pwx * scale, puy = pwy * scale;
            double prx = RR[0][0] * pux + RR[0][1] * puy + RR[0][2] * 1.0;
            double pry = RR[1][0] * pux + RR[1][1] * puy + RR[1][2] * 1.0;
            double prz = RR[2][0] * pux + RR[2][1] * puy + RR[2][2] * 1.0;
            out[i].x = (float)(prx / prz); out[i].y = (float)(pry / prz);
        }
    }
}

/* ======================================================================== fundamental matrix
 * [upstream fundam.cpp FMEstimatorCallback, ptsetreg.cpp RANSAC/LMedS registrators, 3.4.6/4.1.2]
 * The 2-D null space of the 7x9 system is taken from a Householder QR of A^T (fixed operation
 * order) instead of OpenCV's Jacobi SVD: any orthonormal basis spans the same pencil
 * lambda*f1+(1-lambda)*f2, so the F matrices are the same up to rounding. */
static void nullspace_7x9(const double* A /*7x9 row-major*/, double* f1, double* f2)
{
    /* QR of M = A^T (9x7) by Householder; null(A) = last two columns of Q */
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
    /* q = H0 H1 ... H6 e_j for j = 7, 8 */
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

static int solve_cubic(const double* coef, double* roots)
{   /* cv::solveCubic [upstream mathfuncs.cpp], 1x4 coefficient form */
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

int lvo_fundamental_7pt(const lvo_pt2f* m1, const lvo_pt2f* m2, double* fmatrix)
{   /* run7Point [upstream fundam.cpp] */
    double a[7 * 9], f1[9], f2[9], c[4], r[3] = {0, 0, 0};
    for (int i = 0; i < 7; ++i) {
        double x0 = m1[i].x, y0 = m1[i].y, x1 = m2[i].x, y1 = m2[i].y;
        a[i * 9 + 0] = x1 * x0; a[i * 9 + 1] = x1 * y0; a[i * 9 + 2] = x1;
        a[i * 9 + 3] = y1 * x0; a[i * 9 + 4] = y1 * y0; a[i * 9 + 5] = y1;
        a[i * 9 + 6] = x0; a[i * 9 + 7] = y0; a[i * 9 + 8] = 1;
    }
    nullspace_7x9(a, f1, f2);
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
    int n = solve_cubic(c, r);
    if (n < 1 || n > 3) return n < 0 ? 0 : n > 3 ? 0 : n;
    for (int k = 0; k < n; ++k, fmatrix += 9) {
        double lambda = r[k], mu = 1.;
        double s = f1[8] * r[k] + f2[8];
        if (fabs(s) > DBL_EPSILON) { mu = 1. / s; lambda *= mu; fmatrix[8] = 1.; }
        else fmatrix[8] = 0.;
        for (int i = 0; i < 8; ++i) fmatrix[i] = f1[i] * lambda + f2[i] * mu;
    }
    return n;
}

static void fm_compute_error(const lvo_pt2f* m1, const lvo_pt2f* m2, int n, const double* F, float* err)
{   /* FMEstimatorCallback::computeError */
    for (int i = 0; i < n; ++i) {
        double a, b, c, d1, d2, s1, s2;
        a = F[0] * m1[i].x + F[1] * m1[i].y + F[2];
        b = F[3] * m1[i].x + F[4] * m1[i].y + F[5];
        c = F[6] * m1[i].x + F[7] * m1[i].y + F[8];
        s2 = 1. / (a * a + b * b);
        d2 = m2[i].x * a + m2[i].y * b + c;
        a = F[0] * m2[i].x + F[3] * m2[i].y + F[6];
        b = F[1] * m2[i].x + F[4] * m2[i].y + F[7];
        c = F[2] * m2[i].x + F[5] * m2[i].y + F[8];
        s1 = 1. / (a * a + b * b);
        d1 = m1[i].x * a + m1[i].y * b + c;
        double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
        err[i] = (float)(e1 > e2 ? e1 : e2);
    }
}

typedef struct { uint64_t state; } cv_rng;
static inline unsigned rng_next(cv_rng* r)
{   /* cv::RNG::next (MWC) */
    r->state = (uint64_t)(unsigned)r->state * 4164903690U + (unsigned)(r->state >> 32);
    return (unsigned)r->state;
}
static inline int rng_uniform(cv_rng* r, int a, int b) { return a == b ? a : (int)(rng_next(r) % (unsigned)(b - a) + a); }

static int have_collinear(const lvo_pt2f* ptr, int count)
{   /* haveCollinearPoints: last point vs all previous pairs */
    int i = count - 1;
    for (int j = 0; j < i; ++j) {
        double dx1 = ptr[j].x - ptr[i].x, dy1 = ptr[j].y - ptr[i].y;
        for (int k = 0; k < j; ++k) {
            double dx2 = ptr[k].x - ptr[i].x, dy2 = ptr[k].y - ptr[i].y;
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return 1;
        }
    }
    return 0;
}

static int get_subset(const lvo_pt2f* m1, const lvo_pt2f* m2, int count, lvo_pt2f* ms1, lvo_pt2f* ms2,
                      cv_rng* rng, int max_attempts)
{   /* PointSetRegistrator getSubset, modelPoints = 7 */
    int idx[7], iters = 0;
    for (; iters < max_attempts; ++iters) {
        int i;
        for (i = 0; i < 7; ++i) {
            int idx_i;
            for (;;) {
                idx_i = rng_uniform(rng, 0, count);
                int dup = 0;
                for (int q = 0; q < i; ++q) if (idx[q] == idx_i) { dup = 1; break; }
                if (!dup) break;
            }
            idx[i] = idx_i;
            ms1[i] = m1[idx_i]; ms2[i] = m2[idx_i];
        }
        if (!have_collinear(ms1, 7) && !have_collinear(ms2, 7)) break;
    }
    return iters < max_attempts;
}

static int ransac_update_num_iters(double p, double ep, int model_points, int max_iters)
{   /* cv::RANSACUpdateNumIters */
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num); denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : cv_round_d(num / denom);
}

static int find_inliers(const lvo_pt2f* m1, const lvo_pt2f* m2, int n, const double* F, float* err,
                        uint8_t* mask, double thresh)
{
    fm_compute_error(m1, m2, n, F, err);
    float t = (float)(thresh * thresh);
    int nz = 0;
    for (int i = 0; i < n; ++i) { int f = err[i] <= t; mask[i] = (uint8_t)f; nz += f; }
    return nz;
}

static int ransac_fundamental_model(const lvo_pt2f* m1, const lvo_pt2f* m2, int count,
                                    double thresh, double conf, int max_iters, uint8_t* mask_out, int* iters_out, double* best_out)
{   /* RANSACPointSetRegistrator::run, modelPoints 7; best_out (optional): the model the registrator copies out (bestModel) */
    cv_rng rng; rng.state = (uint64_t)-1;
    int niters = max_iters > 1 ? max_iters : 1, max_good = 0, iter;
    float* err = (float*)malloc(sizeof(float) * (size_t)count);
    uint8_t* mask = (uint8_t*)malloc((size_t)count);
    lvo_pt2f ms1[7], ms2[7];
    double model[27];
    for (iter = 0; iter < niters; ++iter) {
        if (!get_subset(m1, m2, count, ms1, ms2, &rng, 10000)) { if (iter == 0) { free(err); free(mask); return 0; } break; }
        int nmodels = lvo_fundamental_7pt(ms1, ms2, model);
        if (nmodels <= 0) continue;
        for (int i = 0; i < nmodels; ++i) {
            int good = find_inliers(m1, m2, count, model + 9 * i, err, mask, thresh);
            if (good > (max_good > 6 ? max_good : 6)) {
                memcpy(mask_out, mask, (size_t)count);
                if (best_out) memcpy(best_out, model + 9 * i, 9 * sizeof(double));
                max_good = good;
                niters = ransac_update_num_iters(conf, (double)(count - good) / count, 7, niters);
            }
        }
    }
    if (iters_out) *iters_out = iter;
    free(err); free(mask);
    return max_good > 0;
}

int lvo_ransac_fundamental(const lvo_pt2f* m1, const lvo_pt2f* m2, int count,
                           double thresh, double conf, int max_iters, uint8_t* mask_out, int* iters_out)
{
    return ransac_fundamental_model(m1, m2, count, thresh, conf, max_iters, mask_out, iters_out, NULL);
}

static int cmp_int(const void* a, const void* b) { int x = *(const int*)a, y = *(const int*)b; return x < y ? -1 : x > y; }

static int lmeds_fundamental(const lvo_pt2f* m1, const lvo_pt2f* m2, int count, double conf, uint8_t* mask_out, double* best_out)
{   /* LMeDSPointSetRegistrator::run, modelPoints 7, maxIters 1000 */
    cv_rng rng; rng.state = (uint64_t)-1;
    int niters = ransac_update_num_iters(conf, 0.45, 7, 1000);
    float* err = (float*)malloc(sizeof(float) * (size_t)count);
    int* srt = (int*)malloc(sizeof(int) * (size_t)count);
    lvo_pt2f ms1[7], ms2[7];
    double model[27], best[9], min_median = DBL_MAX;
    for (int iter = 0; iter < niters; ++iter) {
        if (!get_subset(m1, m2, count, ms1, ms2, &rng, 1000)) { if (iter == 0) { free(err); free(srt); return 0; } break; }
        int nmodels = lvo_fundamental_7pt(ms1, ms2, model);
        if (nmodels <= 0) continue;
        for (int i = 0; i < nmodels; ++i) {
            fm_compute_error(m1, m2, count, model + 9 * i, err);
            memcpy(srt, err, sizeof(float) * (size_t)count);       /* nth_element on the int view */
            qsort(srt, (size_t)count, sizeof(int), cmp_int);
            float medf; memcpy(&medf, &srt[count / 2], 4);
            double median = medf;
            if (median < min_median) { min_median = median; memcpy(best, model + 9 * i, sizeof best); }
        }
    }
    int ok = 0;
    if (min_median < DBL_MAX) {
        double sigma = 2.5 * 1.4826 * (1 + 5. / (count - 7)) * sqrt(min_median);
        sigma = sigma > 0.001 ? sigma : 0.001;
        find_inliers(m1, m2, count, best, err, mask_out, sigma);
        if (best_out) memcpy(best_out, best, sizeof best);
        ok = 1;
    }
    free(err); free(srt);
    return ok;
}

int lvo_find_fundamental(const lvo_pt2f* p1, const lvo_pt2f* p2, int n, double thresh, double conf, uint8_t* mask, double* F)
{   /* cv::findFundamentalMat(.., FM_RANSAC, ..) dispatch [upstream fundam.cpp]: the matrix it RETURNS is the registrator's best
     * minimal-sample model - there is no refit on the inliers.  F (optional, 9 doubles) = zeros when the registrator fails (OpenCV returns an
     * empty Mat); for n == 7 the first of the up to three models (OpenCV returns them stacked). */
    if (F) memset(F, 0, 9 * sizeof(double));
    if (n < 7) return 0;                       /* returns before touching the mask */
    if (n == 7) { memset(mask, 1, 7); if (F) { double m[27]; if (lvo_fundamental_7pt(p1, p2, m) > 0) memcpy(F, m, 9 * sizeof(double)); } return 1; }
    if (thresh <= 0) thresh = 3;
    if (conf < DBL_EPSILON || conf > 1 - DBL_EPSILON) conf = 0.99;
    if (n >= 15) {
        if (!ransac_fundamental_model(p1, p2, n, thresh, conf, 1000, mask, NULL, F)) memset(mask, 0, (size_t)n);
    } else {
        if (!lmeds_fundamental(p1, p2, n, conf, mask, F)) memset(mask, 0, (size_t)n);
    }
    return 1;
}
int lvo_find_fundamental_mask(const lvo_pt2f* p1, const lvo_pt2f* p2, int n, double thresh, double conf, uint8_t* mask)
{
    return lvo_find_fundamental(p1, p2, n, thresh, conf, mask, NULL);
}

/* ======================================================================== gyro prediction
 * image_processor.cpp:222-263 (integrateImuData), 266-293 (predictFeatureTracking);
 * cv::Rodrigues [upstream calibration.cpp] in double, result stored as float. */
void lvo_predict_homography(const lvo_imu* imu, int n_imu, double t_prev, double t_curr,
                            const double R_cam_imu[9], const double intr[4], float H[9])
{
    int b = 0;
    while (b < n_imu && imu[b].t - t_prev < -0.0049) ++b;
    int e = b;
    while (e < n_imu && imu[e].t - t_curr < 0.0049) ++e;
    float mw[3] = {0.f, 0.f, 0.f};
    for (int i = b; i < e; ++i) {
        mw[0] += (float)imu[i].gyro[0]; mw[1] += (float)imu[i].gyro[1]; mw[2] += (float)imu[i].gyro[2];
    }
    if (e - b > 0) { float s = 1.0f / (e - b); mw[0] *= s; mw[1] *= s; mw[2] *= s; }
    /* cam_mean = R_cam_imu^T * mean (double product, stored float) */
    float cw[3];
    for (int i = 0; i < 3; ++i) {
        double s = 0.;
        for (int k = 0; k < 3; ++k) s += R_cam_imu[k * 3 + i] * (double)mw[k];
        cw[i] = (float)s;
    }
    double dtime = t_curr - t_prev;
    float rv[3] = {(float)(cw[0] * dtime), (float)(cw[1] * dtime), (float)(cw[2] * dtime)};
    /* Rodrigues */
    double rx = rv[0], ry = rv[1], rz = rv[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    double R[9];
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1. : 0.;
    } else {
        double c = cos(theta), s = sin(theta), c1 = 1. - c, it = theta ? 1. / theta : 0.;
        rx *= it; ry *= it; rz *= it;
        double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
        double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
        for (int i = 0; i < 9; ++i) R[i] = c * ((i % 4 == 0) ? 1. : 0.) + c1 * rrt[i] + s * r_x[i];
    }
    float Rt[9];      /* cam_R_p2c = R^T, float */
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[i * 3 + j] = (float)R[j * 3 + i];
    /* K, K^-1 (Matx33f fast inverse), H = K * R * K^-1, all float */
    float K[9] = {(float)intr[0], 0.f, (float)intr[2], 0.f, (float)intr[1], (float)intr[3], 0.f, 0.f, 1.f};
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
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        float s = 0; for (int k = 0; k < 3; ++k) s += K[i * 3 + k] * Rt[k * 3 + j];
        T[i * 3 + j] = s;
    }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        float s = 0; for (int k = 0; k < 3; ++k) s += T[i * 3 + k] * Ki[k * 3 + j];
        H[i * 3 + j] = s;
    }
}

void lvo_apply_homography(const float H[9], const lvo_pt2f* in, int n, lvo_pt2f* out)
{
    for (int i = 0; i < n; ++i) {
        float p[3] = {in[i].x, in[i].y, 1.0f}, q[3];
        for (int r = 0; r < 3; ++r) { float s = 0; for (int k = 0; k < 3; ++k) s += H[r * 3 + k] * p[k]; q[r] = s; }
        out[i].x = q[0] / q[2]; out[i].y = q[1] / q[2];
    }
}
