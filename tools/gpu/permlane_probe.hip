// permlane_probe.hip — what v_permlane16_swap_b32 (gfx950) does to a per-lane value when both operands are the same register copy:
// prints, for every lane, the lane index that ends up in result 0 and result 1.  Used once in round 6 to fix the data movement of
// chol32_inv_mfma2 (be_linalg.hip): row 2k of the wavefront (16 lanes) copied into row 2k + 1.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out)
{
    unsigned x = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    out[threadIdx.x] = (int)r[0]; out[64 + threadIdx.x] = (int)r[1];
}
int main()
{
    int* d; int h[128];
    if (hipMalloc(&d, sizeof h) != hipSuccess) { std::printf("no device\n"); return 1; }
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int q = 0; q < 2; ++q) { std::printf("result %d:", q); for (int i = 0; i < 64; ++i) std::printf(" %d", h[64 * q + i]); std::printf("\n"); }
    return 0;
}
