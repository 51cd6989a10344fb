// lvk_microbench.hip — instrument calibration for the roofline numbers bench.py quotes.
//
//   mfma_f64      saturation rate of v_mfma_f64_16x16x4_f64 on this GPU: every SIMD of every CU runs waves with several independent
//                 accumulator chains; reported TFLOP/s replaces / confirms the 78.6 TFLOP/s datasheet peak used for roofline_mfma.
//   stream16      wide coalesced streaming read (16 B per lane) of a buffer far larger than the 256 MiB Infinity Cache
//   gather24      the LK access shape: one wavefront per 24 x 24-byte window at a pseudo-random position of a large byte image
//                 (24 rows of 24 consecutive bytes, row pitch = image stride), 1 byte per lane-load, every window touched once
//   rows24        24-byte segments, one per 128-byte line, each line of the buffer touched exactly once (sector-granularity probe)
//
// Each access kernel prints the bytes it asked for (algorithmic), and the distinct 64-B sectors / 128-B lines it touched (host
// count).  Run it once plain (timings) and once under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (counter per kernel): the ratio
// FETCH_SIZE x 1024 / touched-line bytes is the correction factor for that access shape.  profiles/README.md holds the results.
//
// build: hipcc --offload-arch=gfx950 -O3 -o lvk_microbench lvk_microbench.hip       (make -C tools)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <unordered_set>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void __launch_bounds__(256) mb_mfma_f64(double* out, int iters, double seed)
{
    d4 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = d4{seed * c, 0., 0., 0.};
    const double a = seed + threadIdx.x * 1e-9, b = 1.0 - seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0.;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678) out[blockIdx.x * 256 + threadIdx.x] = s;          // never true: keeps the chains alive
}

// `magic` is a launch argument the compiler cannot see: the final compare keeps every load alive (a constant compare against a
// value the byte sums can never reach lets the optimiser delete the loads - the first version of gather24 measured nothing)
__global__ void __launch_bounds__(256) mb_stream16(const uint4* __restrict__ src, size_t n16, unsigned* sink, unsigned magic)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == magic) sink[0] = acc;
}

// one wavefront per window: 24 rows x 24 bytes; lane l < 24 reads byte l of each row (24 dependent-free byte loads per lane)
__global__ void __launch_bounds__(64) mb_gather24(const uint8_t* __restrict__ img, size_t stride, const uint32_t* __restrict__ wx, const uint32_t* __restrict__ wy, int n, unsigned* sink, unsigned magic)
{
    const int w = blockIdx.x; if (w >= n) return;
    const int lane = threadIdx.x;
    unsigned acc = 0;
    if (lane < 24) {
        const uint8_t* p = img + (size_t)wy[w] * stride + wx[w] + lane;
#pragma unroll
        for (int r = 0; r < 24; ++r) acc += p[(size_t)r * stride];
    }
    if (acc == magic) sink[0] = acc;
}

// one 24-byte segment per 128-byte line, every line once: wave w handles lines [64 w, 64 w + 64), lane-group of 24 lanes per line in turn
__global__ void __launch_bounds__(64) mb_rows24(const uint8_t* __restrict__ buf, size_t n_lines, unsigned off, unsigned* sink, unsigned magic)
{
    const int lane = threadIdx.x;
    unsigned acc = 0;
    const size_t l0 = (size_t)blockIdx.x * 64;
    for (int k = 0; k < 64; ++k) {
        const size_t line = l0 + k;
        if (line < n_lines && lane < 24) acc += buf[line * 128 + off + lane];
    }
    if (acc == magic) sink[0] = acc;
}

static double time_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char** argv)
{
    const char* what = argc > 1 ? argv[1] : "all";
    const bool all = !strcmp(what, "all");
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned* sink; CK(hipMalloc(&sink, 64));
    const unsigned magic = argc > 2 ? (unsigned)strtoul(argv[2], nullptr, 0) : 0xFFFFFFF1u;
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", prop.name, prop.gcnArchName, cus, prop.clockRate / 1000);

    if (all || !strcmp(what, "mfma_f64")) {
        double* out; CK(hipMalloc(&out, sizeof(double) * 256 * 4096));
        const int iters = 20000;
        double best = 0; int best_w = 0, best_c = 0;
        for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
            for (int chains = 2; chains <= 8; chains *= 2) {
                const int blocks = cus * waves_per_simd;                  // 256 threads = 4 waves = one per SIMD
                auto launch = [&]() {
                    if (chains == 2) hipLaunchKernelGGL(mb_mfma_f64<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.25);
                    else if (chains == 4) hipLaunchKernelGGL(mb_mfma_f64<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.25);
                    else hipLaunchKernelGGL(mb_mfma_f64<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.25);
                };
                launch(); CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                const double ms = time_ms(e0, e1);
                const double flops = (double)blocks * 4 * chains * iters * (2.0 * 16 * 16 * 4);
                const double tf = flops / (ms * 1e-3) / 1e12;
                printf("{\"bench\": \"mfma_f64\", \"waves_per_simd\": %d, \"chains\": %d, \"ms\": %.3f, \"tflops\": %.2f}\n", waves_per_simd, chains, ms, tf);
                if (tf > best) { best = tf; best_w = waves_per_simd; best_c = chains; }
            }
        }
        printf("{\"bench\": \"mfma_f64_peak\", \"tflops\": %.2f, \"waves_per_simd\": %d, \"chains\": %d, \"datasheet_tflops\": 78.6}\n", best, best_w, best_c);
        CK(hipFree(out));
    }

    const size_t S = (size_t)1 << 30;                                      // 1 GiB: 4x the Infinity Cache
    uint8_t* buf = nullptr;
    if (all || !strcmp(what, "stream16") || !strcmp(what, "gather24") || !strcmp(what, "rows24")) {
        CK(hipMalloc(&buf, S)); CK(hipMemset(buf, 1, S)); CK(hipDeviceSynchronize());
    }
    if (all || !strcmp(what, "stream16")) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(mb_stream16, dim3(cus * 16), dim3(256), 0, 0, (const uint4*)buf, S / 16, sink, magic);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        }
        const double ms = time_ms(e0, e1);
        printf("{\"bench\": \"stream16\", \"kernel\": \"mb_stream16\", \"algorithmic_bytes\": %zu, \"line128_bytes\": %zu, \"ms\": %.3f, \"gbs\": %.1f}\n", S, S, ms, S / (ms * 1e-3) / 1e9);
    }
    if (all || !strcmp(what, "rows24")) {
        const size_t n_lines = S / 128;
        for (unsigned off : {0u, 52u}) {                                   // 52: the 24 bytes straddle the two 64-byte halves of the line
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(mb_rows24, dim3((unsigned)((n_lines + 63) / 64)), dim3(64), 0, 0, (const uint8_t*)buf, n_lines, off, sink, magic);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            const double ms = time_ms(e0, e1);
            const size_t sectors = off == 0 ? n_lines : 2 * n_lines;
            printf("{\"bench\": \"rows24\", \"kernel\": \"mb_rows24\", \"offset\": %u, \"algorithmic_bytes\": %zu, \"sector64_bytes\": %zu, \"line128_bytes\": %zu, \"ms\": %.3f}\n",
                   off, n_lines * 24, sectors * 64, n_lines * 128, ms);
        }
    }
    if (all || !strcmp(what, "gather24")) {
        // a 32768 x 32768 byte image; windows on a jittered grid so that no two windows share a line, as many as LK handles in ~1000 frames
        const size_t stride = 32768; const int n = 400000;
        std::vector<uint32_t> wx(n), wy(n);
        uint64_t s = 88172645463325252ull;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
        std::unordered_set<uint64_t> lines, sectors;
        for (int i = 0; i < n; ++i) {
            wx[i] = (uint32_t)(rnd() % (stride - 32)); wy[i] = (uint32_t)(rnd() % (32768 - 32));
            for (int r = 0; r < 24; ++r) {
                const uint64_t a0 = (uint64_t)(wy[i] + r) * stride + wx[i], a1 = a0 + 23;
                lines.insert(a0 >> 7); lines.insert(a1 >> 7); sectors.insert(a0 >> 6); sectors.insert(a1 >> 6);
            }
        }
        uint32_t *dx, *dy; CK(hipMalloc(&dx, 4 * n)); CK(hipMalloc(&dy, 4 * n));
        CK(hipMemcpy(dx, wx.data(), 4 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, wy.data(), 4 * n, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(mb_gather24, dim3(n), dim3(64), 0, 0, (const uint8_t*)buf, stride, dx, dy, n, sink, magic);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        const double ms = time_ms(e0, e1);
        printf("{\"bench\": \"gather24\", \"kernel\": \"mb_gather24\", \"windows\": %d, \"algorithmic_bytes\": %zu, \"sector64_bytes\": %zu, \"line128_bytes\": %zu, \"ms\": %.3f}\n",
               n, (size_t)n * 576, sectors.size() * 64, lines.size() * 128, ms);
        CK(hipFree(dx)); CK(hipFree(dy));
    }
    if (buf) CK(hipFree(buf));
    CK(hipFree(sink));
    return 0;
}
