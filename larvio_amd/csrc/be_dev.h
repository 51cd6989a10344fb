// be_dev.h — plain-data job/result records shared by the back-end's host code and kernels, and the FP64 device
// math of the per-observation Jacobians and of the triangulation (same operation order as the CPU oracle; the
// library is built with -ffp-contract=off, so +,-,*,/ and sqrt agree bit-for-bit).
#pragma once
#include "lvk_internal.h"

struct CamPose { double R[9]; double t[3]; };                       // camera-to-world rotation (row-major) + position
struct CloneDev { double q[4], p[3], p_fej[3], R_b2c[9], t_c_b[3]; };  // IMUState_Aug fields the Jacobians read (imu_state.h:72-117)
struct TriJob { int n, use_position, obs_off, out_slot1; double position_in[3]; };    // out_slot1 > 0: the result goes to slot out_slot1 - 1 (the row job that consumes it on the device), 0: to the job's own index
struct TriResult { int ok, pad; double position[3], solution[3], inv_depth, obs_anchor[3]; };
enum { JOB_MSCKF = 0, JOB_EKF_NEW = 1, JOB_EKF_TRACKED = 2 };
enum { FJ_GATE = 1, FJ_TRI_PENDING = 2 };
struct FeatJob {
    int type, n_obs, obs_off, anchor_rank, fcol, want_gate;      // want_gate bit 0: gate the job; bit 1 (FJ_TRI_PENDING): its landmark is TriResult[job index] of the triangulation queued ahead of it
    int ccol_off, dst_row1;          // dst_row1 > 0: the job's output rows also go, expanded, to rows dst_row1-1.. of the dense measurement matrix (k_feature_rows' direct output)
    long long stage_off;
    double p_w[3], p_fej[3], inv_depth, obs_anchor[3];
    double gate_thr;                 // chi-square 5 % lower-tail threshold for this job's dof (gatingTest, larvio.cpp:1865-1880)
};
struct FeatResult { double gamma, h2; int rows, first_row, c, accept; };   // accept = !want_gate || gamma < gate_thr
// job >= 0: the row is stacked only if that job's gate accepted it, otherwise the destination row is zeroed (a zero row with a
// zero residual leaves the update unchanged) - the host then never has to read the gate back before launching the update
struct StackRow { long long g_off, r_off; int src_row, c, ccol_off, dst_row, job, pad; };
struct FilterFlags { int leg_dim, if_fej, estimate_td, pad; double sigma2; };

#ifdef __HIPCC__
__device__ __forceinline__ void d_m3_mul(const double* A, const double* B, double* C)
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0.; for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j]; C[i * 3 + j] = s; }
}
__device__ __forceinline__ void d_m3_t(const double* A, double* T) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[j * 3 + i]; }
__device__ __forceinline__ void d_m3_v(const double* A, const double* v, double* o)
{   double t[3]; for (int i = 0; i < 3; ++i) t[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2]; o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; }
__device__ __forceinline__ void d_skew3(const double* w, double* S)
{   S[0] = 0; S[1] = -w[2]; S[2] = w[1]; S[3] = w[2]; S[4] = 0; S[5] = -w[0]; S[6] = -w[1]; S[7] = w[0]; S[8] = 0; }
__device__ __forceinline__ void d_quat_to_rot(const double* q, double* R)
{   // Eigen Quaterniond(w,x,y,z).toRotationMatrix(), q = [x y z w]
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

// ---- triangulation pieces (feature.hpp:252-310)
__device__ __forceinline__ double d_tri_cost(const double* R, const double* t, const double* x, const double* z)
{
    double a = x[0], b = x[1], rho = x[2];
    double h1 = R[0] * a + R[1] * b + R[2] * 1.0 + rho * t[0];
    double h2 = R[3] * a + R[4] * b + R[5] * 1.0 + rho * t[1];
    double h3 = R[6] * a + R[7] * b + R[8] * 1.0 + rho * t[2];
    double d0 = h1 / h3 - z[0], d1 = h2 / h3 - z[1];
    return d0 * d0 + d1 * d1;
}
__device__ __forceinline__ void d_tri_jacobian(const double* R, const double* t, const double* x, const double* z, double* J, double* r, double& w)
{
    double a = x[0], b = x[1], rho = x[2];
    double h1 = R[0] * a + R[1] * b + R[2] * 1.0 + rho * t[0];
    double h2 = R[3] * a + R[4] * b + R[5] * 1.0 + rho * t[1];
    double h3 = R[6] * a + R[7] * b + R[8] * 1.0 + rho * t[2];
    double W[9] = {R[0], R[1], t[0], R[3], R[4], t[1], R[6], R[7], t[2]};
    for (int c = 0; c < 3; ++c) {
        J[c] = 1 / h3 * W[c] - h1 / (h3 * h3) * W[6 + c];
        J[3 + c] = 1 / h3 * W[3 + c] - h2 / (h3 * h3) * W[6 + c];
    }
    r[0] = h1 / h3 - z[0]; r[1] = h2 / h3 - z[1];
    double e = sqrt(r[0] * r[0] + r[1] * r[1]);
    w = e <= 0.01 ? 1.0 : sqrt(2.0 * 0.01 / e);
}
__device__ __forceinline__ void d_solve3_spd(const double* A, const double* b, double* x)
{
    double d0 = A[0];
    double l10 = A[3] / d0, l20 = A[6] / d0;
    double d1 = A[4] - l10 * l10 * d0;
    double l21 = (A[7] - l20 * l10 * d0) / d1;
    double d2 = A[8] - l20 * l20 * d0 - l21 * l21 * d1;
    double y0 = b[0], y1 = b[1] - l10 * y0, y2 = b[2] - l20 * y0 - l21 * y1;
    double z0 = y0 / d0, z1 = y1 / d1, z2 = y2 / d2;
    x[2] = z2; x[1] = z1 - l21 * x[2]; x[0] = z0 - l10 * x[1] - l20 * x[2];
}

// ---- measurementJacobian_msckf (larvio.cpp:859-921)
__device__ inline void d_msckf_obs_jacobian(const CloneDev& c, const double* p_w, const double* z, int if_fej,
                                            double* Hx, double* He, double* Hf, double* r)
{
    double R_b2w[9], R_w2b[9], R_w2c[9];
    d_quat_to_rot(c.q, R_b2w); d_m3_t(R_b2w, R_w2b); d_m3_mul(c.R_b2c, R_w2b, R_w2c);
    double tcb_w[3]; d_m3_v(R_b2w, c.t_c_b, tcb_w);
    double t_c_w[3] = {c.p[0] + tcb_w[0], c.p[1] + tcb_w[1], c.p[2] + tcb_w[2]};
    double pcf[3] = {p_w[0] - t_c_w[0], p_w[1] - t_c_w[1], p_w[2] - t_c_w[2]}, p_c[3];
    d_m3_v(R_w2c, pcf, p_c);
    double pbf[3];
    for (int i = 0; i < 3; ++i) pbf[i] = if_fej ? p_w[i] - c.p_fej[i] : p_w[i] - c.p[i];
    double dz[6] = {1 / p_c[2], 0, -p_c[0] / (p_c[2] * p_c[2]), 0, 1 / p_c[2], -p_c[1] / (p_c[2] * p_c[2])};
    double S[9], A[9], B[9], C[9], St[9];
    d_skew3(pbf, S);
    d_m3_mul(R_w2c, S, A);
    d_m3_mul(A, R_b2w, B);
    d_skew3(c.t_c_b, St);
    d_m3_mul(c.R_b2c, St, C);
    double dxb[18], dxe[18];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        dxb[i * 6 + j] = A[i * 3 + j]; dxb[i * 6 + 3 + j] = -R_w2c[i * 3 + j];
        dxe[i * 6 + j] = B[i * 3 + j] - C[i * 3 + j]; dxe[i * 6 + 3 + j] = -c.R_b2c[i * 3 + j];
    }
    for (int i = 0; i < 2; ++i) {
        for (int j = 0; j < 6; ++j) {
            double s1 = 0, s2 = 0;
            for (int k = 0; k < 3; ++k) { s1 += dz[i * 3 + k] * dxb[k * 6 + j]; s2 += dz[i * 3 + k] * dxe[k * 6 + j]; }
            Hx[i * 6 + j] = s1; He[i * 6 + j] = s2;
        }
        for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += dz[i * 3 + k] * R_w2c[k * 3 + j]; Hf[i * 3 + j] = s; }
    }
    r[0] = z[0] - p_c[0] / p_c[2]; r[1] = z[1] - p_c[1] / p_c[2];
}

// ---- measurementJacobian_ekf_1didp (larvio.cpp:1117-1244), observing clone k != anchor a
__device__ inline void d_ekf_obs_jacobian(const CloneDev& k, const CloneDev& a, const FeatJob& f, const double* z, int if_fej,
                                          double* Hf, double* Ha, double* Hx, double* He, double* r)
{
    const double* R_b2c = k.R_b2c; const double* t_c_b = k.t_c_b; const double* f_an = f.obs_anchor;
    double R_bk2w[9], R_w2bk[9], R_w2ck[9], R_ba2w[9], R_w2ba[9], R_w2ca[9], tmp[3];
    d_quat_to_rot(k.q, R_bk2w); d_m3_t(R_bk2w, R_w2bk); d_m3_mul(R_b2c, R_w2bk, R_w2ck);
    d_m3_v(R_bk2w, t_c_b, tmp);
    double t_ck_w[3] = {k.p[0] + tmp[0], k.p[1] + tmp[1], k.p[2] + tmp[2]};
    d_quat_to_rot(a.q, R_ba2w); d_m3_t(R_ba2w, R_w2ba); d_m3_mul(R_b2c, R_w2ba, R_w2ca);
    double p_ca[3];
    if (if_fej) {
        double d[3] = {f.p_fej[0] - a.p_fej[0], f.p_fej[1] - a.p_fej[1], f.p_fej[2] - a.p_fej[2]}, q[3];
        d_m3_v(R_w2ba, d, q); q[0] -= t_c_b[0]; q[1] -= t_c_b[1]; q[2] -= t_c_b[2];
        d_m3_v(R_b2c, q, p_ca);
    } else { p_ca[0] = f_an[0] / f.inv_depth; p_ca[1] = f_an[1] / f.inv_depth; p_ca[2] = 1 / f.inv_depth; }
    const double* p_w = f.p_w;
    double d[3] = {p_w[0] - t_ck_w[0], p_w[1] - t_ck_w[1], p_w[2] - t_ck_w[2]}, p_ck[3];
    d_m3_v(R_w2ck, d, p_ck);
    r[0] = z[0] - p_ck[0] / p_ck[2]; r[1] = z[1] - p_ck[1] / p_ck[2];
    double Jk[6] = {1 / p_ck[2], 0, -p_ck[0] / (p_ck[2] * p_ck[2]), 0, 1 / p_ck[2], -p_ck[1] / (p_ck[2] * p_ck[2])};
    double R_ca2w[9], M1[9], Jd[3];
    d_m3_t(R_w2ca, R_ca2w); d_m3_mul(R_w2ck, R_ca2w, M1); d_m3_v(M1, f_an, Jd);
    double p_baf[3], p_bkf[3];
    for (int i = 0; i < 3; ++i) {
        p_baf[i] = if_fej ? f.p_fej[i] - a.p_fej[i] : p_w[i] - a.p[i];
        p_bkf[i] = if_fej ? f.p_fej[i] - k.p_fej[i] : p_w[i] - k.p[i];
    }
    double Sa[9], Sk[9], A1[9], K1[9];
    d_skew3(p_baf, Sa); d_skew3(p_bkf, Sk); d_m3_mul(R_w2ck, Sa, A1); d_m3_mul(R_w2ck, Sk, K1);
    double Jxa[18], Jxk[18], Je[18];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        Jxa[i * 6 + j] = -A1[i * 3 + j]; Jxa[i * 6 + 3 + j] = R_w2ck[i * 3 + j];
        Jxk[i * 6 + j] = K1[i * 3 + j]; Jxk[i * 6 + 3 + j] = -R_w2ck[i * 3 + j];
    }
    double v1[3], SkewMx[9], RR[9], R_c2b[9], v2[3], S2[9], Mx[9], D[9], E[9], JeL[9], JeR[9];
    d_m3_v(R_w2bk, p_bkf, v1); v1[0] -= t_c_b[0]; v1[1] -= t_c_b[1]; v1[2] -= t_c_b[2];
    d_skew3(v1, SkewMx);
    d_m3_mul(R_w2bk, R_ba2w, RR);
    d_m3_t(R_b2c, R_c2b); d_m3_v(R_c2b, p_ca, v2); d_skew3(v2, S2); d_m3_mul(RR, S2, Mx);
    for (int i = 0; i < 9; ++i) D[i] = SkewMx[i] - Mx[i];
    d_m3_mul(R_b2c, D, JeL);
    for (int i = 0; i < 9; ++i) E[i] = RR[i] - ((i % 4 == 0) ? 1.0 : 0.0);
    d_m3_mul(R_b2c, E, JeR);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Je[i * 6 + j] = JeL[i * 3 + j]; Je[i * 6 + 3 + j] = JeR[i * 3 + j]; }
    const double J_rho = -1 / (f.inv_depth * f.inv_depth);
    for (int i = 0; i < 2; ++i) {
        double s = 0; for (int c = 0; c < 3; ++c) s += Jk[i * 3 + c] * Jd[c];
        Hf[i] = s * J_rho;
        for (int j = 0; j < 6; ++j) {
            double s1 = 0, s2 = 0, s3 = 0;
            for (int c = 0; c < 3; ++c) { s1 += Jk[i * 3 + c] * Jxa[c * 6 + j]; s2 += Jk[i * 3 + c] * Jxk[c * 6 + j]; s3 += Jk[i * 3 + c] * Je[c * 6 + j]; }
            Ha[i * 6 + j] = s1; Hx[i * 6 + j] = s2; He[i * 6 + j] = s3;
        }
    }
}
#endif
