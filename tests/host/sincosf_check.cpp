// sincosf_check.cpp — larvio_amd/csrc/lvk_sincosf.h (the product's restatement of the cosf / sinf the reference's descriptor rotation
// calls, ORBDescriptor.cpp:343; compiled here as host code, the same text the HIP kernel compiles) against this host's libm on EVERY
// float in [0, 6.2832].  Prints "ok <count>" or the first mismatches.  Built by tests/test_host_math_compose.py with -ffp-contract=off.
#include "lvk_sincosf.h"
#include <cmath>
#include <cstdio>
#include <cstring>
int main()
{
    const float top = 6.2832f; uint32_t hi; memcpy(&hi, &top, 4);
    long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long b = 0; b <= (long)hi; ++b) {
        const uint32_t u = (uint32_t)b; float x, c, s; memcpy(&x, &u, 4);
        lvk_sincosf(x, &c, &s);
        const float cl = cosf(x), sl = sinf(x);
        if (memcmp(&c, &cl, 4) || memcmp(&s, &sl, 4)) { if (++bad < 5) std::printf("mismatch at %08x: cos %a / %a sin %a / %a\n", u, c, cl, s, sl); }
    }
    if (bad) { std::printf("%ld mismatches\n", bad); return 1; }
    std::printf("ok %u\n", hi + 1);
    return 0;
}
