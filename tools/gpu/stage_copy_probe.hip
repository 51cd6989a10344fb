// Probe for the front-end's staging copy (frontend.hip: fe_image_stage): a driver's image (pageable, not in cache: the frames of a
// run are far larger than L3) into a pinned slot, single thread.  Variants: glibc memcpy, AVX2 loads + regular stores, AVX2 loads +
// non-temporal stores, both with and without software prefetch, `rep movsb`; destination = hipHostMalloc default / write-combined /
// plain malloc.  Prints microseconds per 752x480 frame.  Build: hipcc -O3 -Xarch_host -mavx2 tools/gpu/stage_copy_probe.hip -o /tmp/scp
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <string.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
template <bool NT, int PF> static inline void avx_copy(uint8_t* dst, const uint8_t* src, size_t n)
{
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        if (PF) { _mm_prefetch((const char*)src + i + PF, _MM_HINT_NTA); _mm_prefetch((const char*)src + i + PF + 64, _MM_HINT_NTA); }
        const __m256i a = _mm256_loadu_si256((const __m256i*)(src + i)), b = _mm256_loadu_si256((const __m256i*)(src + i + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i*)(src + i + 64)), d = _mm256_loadu_si256((const __m256i*)(src + i + 96));
        if (NT) { _mm256_stream_si256((__m256i*)(dst + i), a); _mm256_stream_si256((__m256i*)(dst + i + 32), b); _mm256_stream_si256((__m256i*)(dst + i + 64), c); _mm256_stream_si256((__m256i*)(dst + i + 96), d); }
        else { _mm256_storeu_si256((__m256i*)(dst + i), a); _mm256_storeu_si256((__m256i*)(dst + i + 32), b); _mm256_storeu_si256((__m256i*)(dst + i + 64), c); _mm256_storeu_si256((__m256i*)(dst + i + 96), d); }
    }
    if (i < n) memcpy(dst + i, src + i, n - i);
    if (NT) _mm_sfence();
}
static inline void movsb(uint8_t* dst, const uint8_t* src, size_t n) { asm volatile("rep movsb" : "+D"(dst), "+S"(src), "+c"(n) : : "memory"); }
int main()
{
    const size_t N = 752 * 480, F = 600;
    uint8_t* src = (uint8_t*)aligned_alloc(64, N * F); for (size_t i = 0; i < N * F; ++i) src[i] = (uint8_t)(i * 7);
    for (int kind = 0; kind < 3; ++kind) {
        uint8_t* dst[3];
        for (auto& d : dst) {
            if (kind == 0) { if (hipHostMalloc((void**)&d, N) != hipSuccess) return 1; }
            else if (kind == 1) { if (hipHostMalloc((void**)&d, N, hipHostMallocWriteCombined) != hipSuccess) return 1; }
            else d = (uint8_t*)aligned_alloc(4096, N);
            memset(d, 0, N);
        }
        const char* kn[3] = {"hipHostMalloc", "hipHostMalloc write-combined", "malloc"};
        for (int mode = 0; mode < 7; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                auto t0 = std::chrono::steady_clock::now();
                for (size_t f = 0; f < F; ++f) {
                    uint8_t* d = dst[f % 3]; const uint8_t* s = src + f * N;
                    switch (mode) {
                    case 0: memcpy(d, s, N); break;
                    case 1: avx_copy<false, 0>(d, s, N); break;
                    case 2: avx_copy<false, 2048>(d, s, N); break;
                    case 3: avx_copy<true, 0>(d, s, N); break;
                    case 4: avx_copy<true, 2048>(d, s, N); break;
                    case 5: avx_copy<true, 4096>(d, s, N); break;
                    case 6: movsb(d, s, N); break;
                    }
                }
                auto t1 = std::chrono::steady_clock::now();
                const char* mn[7] = {"memcpy", "avx2", "avx2 + prefetch 2K", "avx2 nt", "avx2 nt + prefetch 2K", "avx2 nt + prefetch 4K", "rep movsb"};
                if (rep) printf("%-30s %-24s %6.1f us/frame%s\n", kn[kind], mn[mode], std::chrono::duration<double, std::micro>(t1 - t0).count() / F, memcmp(dst[(F - 1) % 3], src + (F - 1) * N, N) ? "  MISMATCH" : "");
            }
        }
    }
    return 0;
}
