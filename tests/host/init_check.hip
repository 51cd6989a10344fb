// Host-only check of the moving-start initialiser's building blocks (larvio_amd/csrc/be_init.h) on closed-form cases, and of the whole
// initialiser on a simulated moving start with exact measurements.  No GPU: the RANSAC stage (the only device stage it uses) is replaced
// by a callback that keeps every correspondence, which is what RANSAC returns on outlier-free input.
//   1  svd3 / jacobi_eig: A = U S V^T, orthonormal factors, descending singular values (random and rank-deficient matrices; through
//      A^T A, so a vanishing singular value is only known to sqrt(eps) |A|: 2e-7 here)
//   2  eight_point + recover_pose: essential matrix of a known relative pose from exact correspondences; (R, t/|t|) recovered
//   3  solve_pnp: from a pose 0.2 rad / 0.3 units off, converges to the exact pose
//   4  bundle_adjust: perturbed poses and points come back to zero reprojection cost with the gauge held
//   5  PreInt: J_R_bg against a finite difference of repropagate() (the reference's recursion is first order in w dt - its noise input
//      is -I dt, not -Jr(w dt) dt - so they agree to |w| dt / 2 of |J|: 5e-4 here); delta_q of a constant rate against exp(w T)
//   6  DynInit on a simulated start (sinusoidal motion, 200 Hz IMU with a gyro bias, 10 Hz feature messages): gravity direction,
//      body-frame velocity, gyro bias and state time against the truth
// Prints "ok <numbers>" or "FAIL <what>".
#include "../../larvio_amd/csrc/be_init.h"
#include <random>
#include <stdio.h>

using namespace lvk_init;
static std::mt19937_64 rng(4);
static double U(double a, double b) { return a + (b - a) * std::uniform_real_distribution<double>(0., 1.)(rng); }
#define FAIL(...) do { printf("FAIL " __VA_ARGS__); printf("\n"); return 1; } while (0)

static void rot_xyz(double rx, double ry, double rz, double* R)
{
    const double w1[3] = {rx, 0, 0}, w2[3] = {0, ry, 0}, w3[3] = {0, 0, rz}; double A[9], B[9], C[9], T[9];
    rodrigues(w1, A); rodrigues(w2, B); rodrigues(w3, C); m3_mul(C, B, T); m3_mul(T, A, R);
}
static double rot_angle_between(const double* A, const double* B)
{
    double At[9], D[9]; m3_t(A, At); m3_mul(At, B, D);
    return acos(std::min(1., std::max(-1., (D[0] + D[4] + D[8] - 1.) / 2.)));
}

// ---- the simulated platform of check 6
struct Traj {
    double p(int k, double t) const { const double A[3] = {1.2, 0.9, 0.4}, w[3] = {1.9, 1.3, 2.3}, ph[3] = {0.3, 1.1, 0.7}; return A[k] * sin(w[k] * t + ph[k]); }
    double v(int k, double t) const { const double A[3] = {1.2, 0.9, 0.4}, w[3] = {1.9, 1.3, 2.3}, ph[3] = {0.3, 1.1, 0.7}; return A[k] * w[k] * cos(w[k] * t + ph[k]); }
    double a(int k, double t) const { const double A[3] = {1.2, 0.9, 0.4}, w[3] = {1.9, 1.3, 2.3}, ph[3] = {0.3, 1.1, 0.7}; return -A[k] * w[k] * w[k] * sin(w[k] * t + ph[k]); }
    void R(double t, double* Rwb) const { rot_xyz(0.25 + 0.15 * sin(1.1 * t), -0.1 + 0.2 * sin(0.8 * t + 0.5), 0.4 + 0.3 * sin(0.6 * t + 1.0), Rwb); }
    void omega(double t, double* w) const
    {   // body rate = vee(R^T dR/dt), central difference
        const double h = 1e-6; double Ra[9], Rb[9], R0[9], R0t[9], dR[9], M[9];
        R(t - h, Ra); R(t + h, Rb); R(t, R0); for (int i = 0; i < 9; ++i) dR[i] = (Rb[i] - Ra[i]) / (2 * h);
        m3_t(R0, R0t); m3_mul(R0t, dR, M); w[0] = 0.5 * (M[7] - M[5]); w[1] = 0.5 * (M[2] - M[6]); w[2] = 0.5 * (M[3] - M[1]);
    }
};
// stand-in for the RANSAC stage (host-only test): every correspondence is an inlier, the matrix is the normalised 8-point fit of all of them
static bool keep_all(void*, const std::vector<Pt2>& ll, const std::vector<Pt2>& rr, double, double, std::vector<unsigned char>& mask, double* F) { mask.assign(ll.size(), 1); return eight_point(ll, rr, F); }

int main()
{
    // ------------------------------------------------------------------ 1
    double worst_svd = 0;
    for (int trial = 0; trial < 200; ++trial) {
        double A[9]; for (double& x : A) x = U(-2, 2);
        if (trial % 4 == 1) { for (int c = 0; c < 3; ++c) A[6 + c] = 0.3 * A[c] - 1.1 * A[3 + c]; }                 // rank 2
        if (trial % 4 == 2) { const double a[3] = {U(-1, 1), U(-1, 1), U(-1, 1)}, b[3] = {U(-1, 1), U(-1, 1), U(-1, 1)}; for (int i = 0; i < 9; ++i) A[i] = a[i / 3] * b[i % 3]; }   // rank 1
        double Um[9], S[3], V[9]; svd3(A, Um, S, V);
        double US[9], Vt[9], Rm[9], UtU[9], Ut[9], VtV[9];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) US[r * 3 + c] = Um[r * 3 + c] * S[c];
        m3_t(V, Vt); m3_mul(US, Vt, Rm); m3_t(Um, Ut); m3_mul(Ut, Um, UtU); m3_mul(Vt, V, VtV);
        for (int i = 0; i < 9; ++i) { worst_svd = std::max(worst_svd, fabs(Rm[i] - A[i])); worst_svd = std::max(worst_svd, fabs(UtU[i] - (i % 4 == 0))); worst_svd = std::max(worst_svd, fabs(VtV[i] - (i % 4 == 0))); }
        if (!(S[0] >= S[1] && S[1] >= S[2] && S[2] >= 0)) FAIL("svd3 order");
    }
    if (worst_svd > 2e-7) FAIL("svd3 %.3e", worst_svd);

    // ------------------------------------------------------------------ 2, 3
    double worst_R = 0, worst_t = 0, worst_pnp = 0;
    for (int trial = 0; trial < 50; ++trial) {
        double R[9], t[3] = {U(-1, 1), U(-1, 1), U(-0.3, 0.3)};                          // x2 = R x1 + t
        rot_xyz(U(-0.3, 0.3), U(-0.3, 0.3), U(-0.3, 0.3), R);
        std::vector<Pt2> p1, p2; std::vector<std::array<double, 3>> X;
        for (int i = 0; i < 60; ++i) {
            const double P[3] = {U(-2, 2), U(-2, 2), U(4, 9)}; double Q[3]; m3_v(R, P, Q); for (int k = 0; k < 3; ++k) Q[k] += t[k];
            p1.push_back({P[0] / P[2], P[1] / P[2]}); p2.push_back({Q[0] / Q[2], Q[1] / Q[2]}); X.push_back({P[0], P[1], P[2]});
        }
        double E[9]; if (!eight_point(p1, p2, E)) FAIL("eight_point");
        for (size_t i = 0; i < p1.size(); ++i) {
            const double l[3] = {E[0] * p1[i].x + E[1] * p1[i].y + E[2], E[3] * p1[i].x + E[4] * p1[i].y + E[5], E[6] * p1[i].x + E[7] * p1[i].y + E[8]};
            const double d = (p2[i].x * l[0] + p2[i].y * l[1] + l[2]) / sqrt(l[0] * l[0] + l[1] * l[1]);
            if (fabs(d) > 1e-8) FAIL("epipolar distance %.3e", d);
        }
        std::vector<unsigned char> mask(p1.size(), 1); double Re[9], te[3];
        const int inl = recover_pose(E, p1, p2, mask, Re, te);
        if (inl != (int)p1.size()) FAIL("recover_pose inliers %d", inl);
        const double nt = v3_norm(t); double dt = 0; for (int k = 0; k < 3; ++k) dt = std::max(dt, fabs(te[k] - t[k] / nt));
        worst_R = std::max(worst_R, rot_angle_between(R, Re)); worst_t = std::max(worst_t, dt);
        // PnP from a perturbed pose
        double Rg[9], dR[9], tg[3] = {t[0] + 0.3, t[1] - 0.2, t[2] + 0.25}; rot_xyz(0.12, -0.1, 0.1, dR); m3_mul(dR, R, Rg);
        if (!solve_pnp(X, p2, Rg, tg)) FAIL("solve_pnp");
        double e = rot_angle_between(R, Rg); for (int k = 0; k < 3; ++k) e = std::max(e, fabs(tg[k] - t[k]));
        worst_pnp = std::max(worst_pnp, e);
    }
    if (worst_R > 1e-7 || worst_t > 1e-7) FAIL("recover_pose R %.3e t %.3e", worst_R, worst_t);
    if (worst_pnp > 1e-7) FAIL("solve_pnp %.3e", worst_pnp);

    // ------------------------------------------------------------------ 4
    double ba_cost = 0;
    {
        const int nf = 6, l = 1;
        std::vector<std::array<double, 9>> Rc((size_t)nf), Rt((size_t)nf); std::vector<std::array<double, 3>> tc((size_t)nf), tt((size_t)nf);
        for (int f = 0; f < nf; ++f) {
            rot_xyz(0.05 * (f - l), -0.04 * (f - l), 0.03 * (f - l), Rt[(size_t)f].data()); tt[(size_t)f] = {0.3 * (f - l), -0.1 * (f - l), 0.05 * (f - l) * (f - l)};
            Rc[(size_t)f] = Rt[(size_t)f]; tc[(size_t)f] = tt[(size_t)f];
        }
        std::vector<SfmFeature> feats;
        for (int i = 0; i < 80; ++i) {
            SfmFeature sf; sf.state = true; sf.id = i; const double P[3] = {U(-2, 2), U(-2, 2), U(4, 9)};
            for (int f = 0; f < nf; ++f) { double Q[3]; m3_v(Rt[(size_t)f].data(), P, Q); for (int k = 0; k < 3; ++k) Q[k] += tt[(size_t)f][(size_t)k]; sf.obs.push_back({f, Pt2{Q[0] / Q[2], Q[1] / Q[2]}}); }
            for (int k = 0; k < 3; ++k) sf.position[k] = P[k] + U(-0.2, 0.2);
            feats.push_back(sf);
        }
        for (int f = 0; f < nf; ++f) {
            if (f != l) { double dR[9], Rn[9]; rot_xyz(U(-0.03, 0.03), U(-0.03, 0.03), U(-0.03, 0.03), dR); m3_mul(dR, Rc[(size_t)f].data(), Rn); memcpy(Rc[(size_t)f].data(), Rn, 72); }
            if (f != l && f != nf - 1) for (int k = 0; k < 3; ++k) tc[(size_t)f][(size_t)k] += U(-0.1, 0.1);
        }
        if (!bundle_adjust(nf, l, Rc, tc, feats)) FAIL("bundle_adjust did not converge");
        for (auto& sf : feats) for (auto& ob : sf.obs) {
            double Q[3]; m3_v(Rc[(size_t)ob.first].data(), sf.position, Q); for (int k = 0; k < 3; ++k) Q[k] += tc[(size_t)ob.first][(size_t)k];
            ba_cost += (Q[0] / Q[2] - ob.second.x) * (Q[0] / Q[2] - ob.second.x) + (Q[1] / Q[2] - ob.second.y) * (Q[1] / Q[2] - ob.second.y);
        }
        double worst = 0; for (int f = 0; f < nf; ++f) { worst = std::max(worst, rot_angle_between(Rc[(size_t)f].data(), Rt[(size_t)f].data())); for (int k = 0; k < 3; ++k) worst = std::max(worst, fabs(tc[(size_t)f][(size_t)k] - tt[(size_t)f][(size_t)k])); }
        if (ba_cost > 1e-10 || worst > 1e-4) FAIL("bundle_adjust cost %.3e pose error %.3e", ba_cost, worst);
    }

    // ------------------------------------------------------------------ 5
    double worst_J = 0, worst_q = 0;
    {
        const double z3[3] = {0, 0, 0}, w[3] = {0.4, -0.3, 0.2}, a[3] = {0.1, 0.2, 9.7};
        PreInt P; P.start(a, w, z3, z3);
        for (int i = 0; i < 40; ++i) { const double wi[3] = {w[0] + 0.05 * sin(0.3 * i), w[1], w[2] - 0.02 * i / 40.}; P.push_back(0.005, a, wi); }
        PreInt P0 = P; double J[9]; memcpy(J, P.J_R_bg, 72);
        for (int k = 0; k < 3; ++k) {
            double bg[3] = {0, 0, 0}; bg[k] = 1e-6; PreInt Q = P0; Q.repropagate(z3, bg);
            double qi[4] = {-P0.dq[0], -P0.dq[1], -P0.dq[2], P0.dq[3]}, d[4]; quat_mul(qi, Q.dq, d);
            for (int r = 0; r < 3; ++r) worst_J = std::max(worst_J, fabs(2 * d[r] / 1e-6 - J[r * 3 + k]));
        }
        PreInt C; C.start(a, w, z3, z3); for (int i = 0; i < 200; ++i) C.push_back(0.005, a, w);
        const double wt[3] = {w[0], w[1], w[2]}; double Rw[9], Rq[9]; rodrigues(wt, Rw); quat_to_rot(C.dq, Rq);
        worst_q = rot_angle_between(Rw, Rq);
    }
    if (worst_J > 5e-4 || worst_q > 1e-5) FAIL("PreInt J %.3e dq %.3e", worst_J, worst_q);

    // ------------------------------------------------------------------ 6
    double e_grav = 0, e_vel = 0, e_bg = 0; int frames_used = 0;
    {
        Traj tr; const double bg_true[3] = {0.012, -0.008, 0.005};
        const double Rbc_deg[3] = {-1.5, 0.02, -1.55};                                   // a forward-looking camera, slightly off-axis
        double R_b2c[9]; rot_xyz(Rbc_deg[0], Rbc_deg[1], Rbc_deg[2], R_b2c);
        const double t_c_b[3] = {0.06, -0.02, 0.01};
        DynInit d; d.reset(); d.td = 0; d.imu_img_time_th = 1. / 400.;
        m3_t(R_b2c, d.RIC); memcpy(d.TIC, t_c_b, 24);
        for (int i = 0; i < 9; ++i) { d.Ma[i] = d.Tg[i] = (i % 4 == 0); d.As[i] = 0; }
        d.ransac = keep_all;
        std::vector<lvk_imu> imu;
        for (int k = 0; k < 1200; ++k) {
            lvk_imu s; s.t = 0.0025 + 0.005 * k; double Rwb[9], aw[3] = {tr.a(0, s.t), tr.a(1, s.t), tr.a(2, s.t) + 9.81}, w[3];
            tr.R(s.t, Rwb); m3t_v(Rwb, aw, s.acc); tr.omega(s.t, w); for (int i = 0; i < 3; ++i) s.gyro[i] = w[i] + bg_true[i];
            imu.push_back(s);
        }
        // landmarks in a shell around the start, seen when in front of the camera and inside a 90-degree field of view
        std::vector<std::array<double, 3>> L; for (int i = 0; i < 1500; ++i) { const double az = U(0, 6.2832), el = U(-0.6, 0.6), r = U(4, 10); L.push_back({r * cos(el) * cos(az), r * cos(el) * sin(az), r * sin(el)}); }
        auto observe = [&](double t, size_t i, Pt2* z) {
            double Rwb[9], pc[3], pb[3]; tr.R(t, Rwb);
            const double dw[3] = {L[i][0] - tr.p(0, t), L[i][1] - tr.p(1, t), L[i][2] - tr.p(2, t)};
            m3t_v(Rwb, dw, pb); for (int k = 0; k < 3; ++k) pb[k] -= t_c_b[k]; m3_v(R_b2c, pb, pc);
            if (pc[2] < 0.5) return false;
            z->x = pc[0] / pc[2]; z->y = pc[1] / pc[2];
            return fabs(z->x) < 0.9 && fabs(z->y) < 0.6;
        };
        bool done = false; int erase = 0; double ts = 0; std::vector<int> gen(L.size(), 0), seen(L.size(), 0);
        for (int m = 0; m < 40 && !done; ++m) {
            ts = 0.5 + 0.1 * m;
            std::vector<lvk_feature_obs> f;
            for (size_t i = 0; i < L.size(); ++i) {
                Pt2 z, z0; if (!observe(ts, i, &z) || !observe(ts - 0.05, i, &z0)) { if (seen[i]) { seen[i] = 0; ++gen[i]; } continue; }   // a track that is lost never comes back under its old id
                seen[i] = 1;
                lvk_feature_obs o; memset(&o, 0, sizeof o); o.id = i + L.size() * (size_t)gen[i]; o.u = z.x; o.v = z.y; o.u_vel = (z.x - z0.x) / 0.05; o.v_vel = (z.y - z0.y) / 0.05; f.push_back(o);
            }
            int n_imu = 0; while (n_imu < (int)imu.size() && imu[(size_t)n_imu].t < ts + 0.05) ++n_imu;
            done = d.try_init(ts, f.data(), (int)f.size(), imu.data(), n_imu, &erase);
            ++frames_used;
        }
        if (!done) FAIL("DynInit did not initialise in %d messages", frames_used);
        const double t = d.out.state_time;
        if (fabs(t - ts) > 0.0026) FAIL("state time %.6f for a message at %.6f", t, ts);
        double Rwb[9], Re[9]; tr.R(t, Rwb); quat_to_rot(d.out.q, Re);
        // gravity direction in the body frame: third ROW of R_wb; body-frame velocity
        for (int k = 0; k < 3; ++k) e_grav = std::max(e_grav, fabs(Re[6 + k] - Rwb[6 + k]));
        const double vw[3] = {tr.v(0, t), tr.v(1, t), tr.v(2, t)}; double vb[3], vbe[3]; m3t_v(Rwb, vw, vb); m3t_v(Re, d.out.v, vbe);
        for (int k = 0; k < 3; ++k) { e_vel = std::max(e_vel, fabs(vb[k] - vbe[k])); e_bg = std::max(e_bg, fabs(d.out.bg[k] - bg_true[k])); }
        if (erase <= 0 || imu[(size_t)erase - 1].t > t || (erase < (int)imu.size() && imu[(size_t)erase].t <= t)) FAIL("erase count %d", erase);
        if (e_grav > 2e-3 || e_vel > 0.02 || e_bg > 2e-4) FAIL("DynInit gravity %.3e velocity %.3e bg %.3e after %d messages", e_grav, e_vel, e_bg, frames_used);
    }
    printf("ok svd %.1e pose %.1e/%.1e pnp %.1e ba %.1e J %.1e dq %.1e | init after %d messages: gravity %.2e velocity %.2e m/s bg %.2e rad/s\n",
           worst_svd, worst_R, worst_t, worst_pnp, ba_cost, worst_J, worst_q, frames_used, e_grav, e_vel, e_bg);
    return 0;
}
