// ORACLE / TEST INFRASTRUCTURE ONLY: boost::math::chi_squared + quantile for the reference's larvio.cpp compiled in place (oracle/Makefile,
// target `ref`; boost is not installed).  The quantile is the root of the regularised lower incomplete gamma function P(k/2, x/2) = p,
// found by bisection to the last bit: the series / continued fraction of Numerical-Recipes fame, written here from the definitions (the
// product's table, chi2_table.inc, was checked against scipy; this is a third, independent evaluation).
#pragma once
#include <cmath>
namespace boost { namespace math {
struct chi_squared { double k; explicit chi_squared(double dof) : k(dof) {} };
inline double lvref_gamma_p(double a, double x)
{
    if (x <= 0) return 0.0;
    const double lg = std::lgamma(a);
    if (x < a + 1.0) {                                   // series
        double ap = a, sum = 1.0 / a, del = sum;
        for (int n = 0; n < 100000; ++n) { ap += 1.0; del *= x / ap; sum += del; if (std::fabs(del) < std::fabs(sum) * 1e-17) break; }
        return sum * std::exp(-x + a * std::log(x) - lg);
    }
    double b = x + 1.0 - a, c = 1.0 / 1e-300, d = 1.0 / b, h = d;      // continued fraction for Q
    for (int i = 1; i < 100000; ++i) {
        const double an = -i * (i - a); b += 2.0;
        d = an * d + b; if (std::fabs(d) < 1e-300) d = 1e-300;
        c = b + an / c; if (std::fabs(c) < 1e-300) c = 1e-300;
        d = 1.0 / d; const double del = d * c; h *= del;
        if (std::fabs(del - 1.0) < 1e-17) break;
    }
    return 1.0 - std::exp(-x + a * std::log(x) - lg) * h;
}
inline double quantile(const chi_squared& dist, double p)
{
    double lo = 0.0, hi = dist.k + 10.0 * std::sqrt(2.0 * dist.k) + 10.0;
    for (int it = 0; it < 200; ++it) { const double mid = 0.5 * (lo + hi); if (lvref_gamma_p(0.5 * dist.k, 0.5 * mid) < p) lo = mid; else hi = mid; if (hi - lo <= 0) break; }
    return 0.5 * (lo + hi);
}
} }
