// Dumps larvio_amd/csrc/be_host_math.h (the product's host-side 3x3 / quaternion helpers) on seeded random inputs as JSON lines, for an
// independent check against numpy / scipy (tests/test_host_math_compose.py::test_product_host_math_against_scipy): these helpers share
// their text with oracle/be_math.h, so the GPU-vs-oracle parity tests cannot say anything about them.
#include "../../larvio_amd/csrc/be_host_math.h"
#include <stdio.h>
#include <random>
static void arr(const char* k, const double* v, int n, bool last = false) { printf("\"%s\": [", k); for (int i = 0; i < n; ++i) printf("%s%.17g", i ? ", " : "", v[i]); printf("]%s", last ? "" : ", "); }
int main()
{
    std::mt19937_64 rng(7); std::normal_distribution<double> N(0., 1.);
    for (int t = 0; t < 200; ++t) {
        double q[4], p[4], w[3], A[9], B[9], v[3];
        for (double& x : q) x = N(rng); for (double& x : p) x = N(rng); for (double& x : w) x = N(rng) * (t % 3 == 0 ? 2.0 : 0.05);
        for (double& x : A) x = N(rng); for (double& x : B) x = N(rng); for (double& x : v) x = N(rng);
        double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]); for (double& x : q) x /= nq;
        if (t % 5 == 0) { q[3] = -fabs(q[3]) * 0.01; nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]); for (double& x : q) x /= nq; }   // trace <= 0 branches
        double R[9], q2[4], qp[4], AB[9], At[9], Av[3], Atv[3], S[9], dq[4];
        quat_to_rot(q, R); rot_to_quat(R, q2); quat_mul(q, p, qp); m3_mul(A, B, AB); m3_t(A, At); m3_v(A, v, Av); m3t_v(A, v, Atv); skew3(w, S); small_angle_quat(w, dq);
        printf("{"); arr("q", q, 4); arr("p", p, 4); arr("w", w, 3); arr("A", A, 9); arr("B", B, 9); arr("v", v, 3);
        arr("R", R, 9); arr("q2", q2, 4); arr("qp", qp, 4); arr("AB", AB, 9); arr("At", At, 9); arr("Av", Av, 3); arr("Atv", Atv, 3); arr("S", S, 9); arr("dq", dq, 4);
        const double n = v3_norm(v); arr("n", &n, 1, true); printf("}\n");
    }
    return 0;
}
