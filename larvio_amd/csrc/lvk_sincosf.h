// lvk_sincosf.h — the cosine and sine the reference's descriptor rotation calls, restated.
// ORBDescriptor.cpp:343 writes `(float)cos(angle), (float)sin(angle)` with a float `angle` under `using namespace std;` (:16), so
// overload resolution picks std::cos(float) / std::sin(float) = libm's cosf / sinf - not the double functions (rounds 1-5 restated
// cos((double)angle) rounded to float: 0.04 % / 0.09 % of all floats in [0, 2 pi] give another float that way, and once in ~400
// fuzzed streams the rotated sampling pattern of a descriptor moved by one pixel: PARITY.md section 2, fuzz case 379).
// This is the algorithm glibc >= 2.28 ships for both (ARM Optimized Routines' sincosf: sysdeps/ieee754/flt-32/s_sincosf.h,
// s_sinf.c, s_cosf.c, s_sincosf_data.c): the argument widened to double, one multiply by 2/pi * 2^24 and an integer shift for the
// quadrant, x - n * (pi/2) in double, a degree-7 / degree-8 polynomial in double, one rounding to float.
// PINNED: tests/test_host_math_compose.py sweeps EVERY float in [0, 6.2832] (1,086,918,650 values: the descriptor's angle is
// fastAtan2's [0, 360) degrees times a float pi/180) against the host libm's cosf / sinf - zero mismatches on glibc 2.35, with and
// without fused multiply-adds (the double-precision intermediate leaves the float result the same either way on this domain).
// Valid for |x| < 120 (glibc's fast-reduction range); the descriptor never leaves [0, 2 pi].
#pragma once
#include <stdint.h>
#include <string.h>
#if defined(__HIPCC__)
#define LVK_SC_HD __host__ __device__ __forceinline__
#else
#define LVK_SC_HD static inline
#endif

LVK_SC_HD uint32_t lvk_sc_abstop12(float x) { uint32_t u; memcpy(&u, &x, 4); return (u >> 20) & 0x7ff; }
// polynomial on [-pi/4, pi/4]: sine for even n, cosine for odd n; neg = the negated cosine coefficients (quadrants 2, 3)
LVK_SC_HD float lvk_sc_poly(double x, double x2, int n, int neg)
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
LVK_SC_HD void lvk_sincosf(float y, float* cos_out, float* sin_out)
{
    const double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;    // 2/pi * 2^24, pi/2
    const double x = y;
    if (lvk_sc_abstop12(y) < lvk_sc_abstop12(0x1.921FB6p-1f)) {                 // |y| < pi/4 (by the top 12 bits)
        const double x2 = x * x;
        const bool tiny = lvk_sc_abstop12(y) < lvk_sc_abstop12(0x1p-12f);
        *sin_out = tiny ? y : lvk_sc_poly(x, x2, 0, 0);
        *cos_out = tiny ? 1.0f : lvk_sc_poly(x, x2, 1, 0);
        return;
    }
    const double r = x * HPI_INV;
    const int n = ((int32_t)r + 0x800000) >> 24;                                 // quadrant: round(x * 2/pi)
    const double xr = x - n * HPI;
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;              // sign[n & 3] = {1, -1, -1, 1}
    *sin_out = lvk_sc_poly(xr * sgn, xr * xr, n, (n & 2) != 0);
    *cos_out = lvk_sc_poly(xr * sgn, xr * xr, n ^ 1, ((n + 1) & 2) != 0);
}
