// lvk_wave.h - wavefront-level helpers shared by the back-end kernels (device code only)
#ifndef LVK_WAVE_H
#define LVK_WAVE_H
#include <hip/hip_runtime.h>

// FP64 all-reduce over the wavefront without LDS traffic: four DPP row rotations (every lane ends up with its 16-lane row sum; a
// 64-bit value moves as two 32-bit DPP movs), then the four row sums are read with v_readlane and added.  (__shfl_xor on a double is
// two ds_bpermute per stage, ~12 LDS round trips per reduction: it dominated the first version of the LDS-resident QR nodes.)
__device__ __forceinline__ double dpp_ror_f64(double v, const int ctrl_sel)
{   // (bound_ctrl set: a row rotation has no invalid source lane, and with it the compiler need not initialise the destination - two
    // v_mov per 64-bit move less on every reduction chain)
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (ctrl_sel) {
        case 8: lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xF, 0xF, true); break;
        case 4: lo = __builtin_amdgcn_update_dpp(0, lo, 0x124, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x124, 0xF, 0xF, true); break;
        case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x122, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x122, 0xF, 0xF, true); break;
        default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x121, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x121, 0xF, 0xF, true); break;
    }
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
    v += dpp_ror_f64(v, 8); v += dpp_ror_f64(v, 4); v += dpp_ror_f64(v, 2); v += dpp_ror_f64(v, 1);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double s = 0.;
#pragma unroll
    for (int r = 0; r < 4; ++r) s += __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * r), __builtin_amdgcn_readlane(lo, 16 * r));
    return s;
}

// 1/sqrt(x) to full double precision from the v_rsq_f64 seed by two coupled (Goldschmidt) steps, all FMAs: six dependent operations
// behind the seed (a Newton form compiled without contraction is nine, 1.0 / sqrt(x) is ~90 instructions of IEEE expansion)
__device__ __forceinline__ double rsqrt_goldschmidt(double x)
{
    const double y0 = __builtin_amdgcn_rsq(x);
    double g = x * y0, h = 0.5 * y0;
    double r = __builtin_fma(-g, h, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    r = __builtin_fma(-g, h, 0.5);
    h = __builtin_fma(h, r, h);
    return h + h;
}

#endif
