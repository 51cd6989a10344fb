// Host round trip on a dependent chain: kernel A ends -> the host learns of it -> the host launches kernel B -> B starts.
// Measured with the device's 100 MHz wall clock (end of A, start of B) for two ways of learning:
//   sync : hipStreamSynchronize after A
//   flag : A's last action is a store of a sequence number into device-mapped pinned host memory; the host spins on it
// build: hipcc -O2 --offload-arch=gfx950 tools/gpu/sync_probe.hip -o /tmp/sync_probe ; run: /tmp/sync_probe [work_us]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__global__ void k_a(unsigned long long* ticks, int slot, volatile int* host_flag, int seq, int spin_ticks)
{
    const unsigned long long t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < spin_ticks) { }
    if (threadIdx.x == 0) {
        ticks[2 * slot] = wall_clock64();
        if (host_flag) { __threadfence_system(); *host_flag = seq; }
    }
}
__global__ void k_b(unsigned long long* ticks, int slot)
{
    if (threadIdx.x == 0) ticks[2 * slot + 1] = wall_clock64();
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv)
{
    const int work_us = argc > 1 ? atoi(argv[1]) : 10;
    const int N = 400;
    hipStream_t s; hipStreamCreate(&s);
    unsigned long long* d_ticks; hipMalloc(&d_ticks, sizeof(unsigned long long) * 2 * N);
    int* h_flag; hipHostMalloc(&h_flag, 64, hipHostMallocMapped); *h_flag = 0;
    int* d_flag; hipHostGetDevicePointer((void**)&d_flag, h_flag, 0);
    std::vector<unsigned long long> h(2 * N);
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<double> gap, host_rt;
        for (int i = 0; i < N; ++i) {
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, s, d_ticks, i, mode ? (volatile int*)d_flag : nullptr, i + 1 + mode * 100000, work_us * 100);
            if (mode == 0) hipStreamSynchronize(s);
            else { volatile int* f = h_flag; while (*f != i + 1 + mode * 100000) __builtin_ia32_pause(); }
            auto t1 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, s, d_ticks, i);
            hipStreamSynchronize(s);
            host_rt.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        hipMemcpy(h.data(), d_ticks, sizeof(unsigned long long) * 2 * N, hipMemcpyDeviceToHost);
        for (int i = 20; i < N; ++i) gap.push_back((double)(long long)(h[2 * i + 1] - h[2 * i]) * 0.01);
        std::vector<double> rt(host_rt.begin() + 20, host_rt.end());
        printf("{\"mode\": \"%s\", \"kernel_work_us\": %d, \"gap_end_of_A_to_start_of_B_us_p50\": %.2f, \"host_launch_to_known_done_us_p50\": %.2f}\n",
               mode ? "flag" : "sync", work_us, med(gap), med(rt));
    }
    return 0;
}
