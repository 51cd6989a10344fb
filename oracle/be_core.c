/*
 * be_core.c — ORACLE (test infrastructure, see lvo.h): the numerical stages of the EKF update,
 * a plain-C restatement of /root/reference/src/larvio.cpp:859-981 (MSCKF Jacobians + nullspace),
 * :1117-1244 (1-D inverse-depth Jacobian), :1430-1460,1578-1594 (compression + update),
 * :1865-1880 (gate) and include/larvio/feature.hpp:252-552 (LM triangulation).
 * PINNED to the reference compiled in place (lvo.h, "PINNING"): src/larvio.cpp's filter after every update
 * (tests/test_oracle_ref_larvio.py) and include/larvio/feature.hpp's triangulation (tests/test_oracle_ref_feature.py).
 */
#include "lvo.h"
#include "be_math.h"
#include "chi2_table.inc"
#include <stdlib.h>
#include <float.h>

double lvo_chi2_table(int dof) { return (dof >= 1 && dof <= 99) ? k_chi2_005[dof] : 0.0; }

/* ------------------------------------------------------------------------ triangulation */
static void tri_cost(const double* R, const double* t, const double* x, const double* z, double* e)
{   /* Feature::cost (feature.hpp:252-270) */
    double a = x[0], b = x[1], rho = x[2];
    double h1 = R[0] * a + R[1] * b + R[2] * 1.0 + rho * t[0];
    double h2 = R[3] * a + R[4] * b + R[5] * 1.0 + rho * t[1];
    double h3 = R[6] * a + R[7] * b + R[8] * 1.0 + rho * t[2];
    double d0 = h1 / h3 - z[0], d1 = h2 / h3 - z[1];
    *e = d0 * d0 + d1 * d1;
}
static void tri_jacobian(const double* R, const double* t, const double* x, const double* z, double* J /*2x3*/, double* r, double* w)
{   /* Feature::jacobian (feature.hpp:272-310), huber_epsilon 0.01 */
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
    *w = e <= 0.01 ? 1.0 : sqrt(2.0 * 0.01 / e);
}
static void solve3_spd(const double* A, const double* b, double* x)
{   /* (A+damper).ldlt().solve(b): 3x3 LDL^T without pivoting (fixed order) */
    double d0 = A[0];
    double l10 = A[3] / d0, l20 = A[6] / d0;
    double d1 = A[4] - l10 * l10 * d0;
    double l21 = (A[7] - l20 * l10 * d0) / d1;
    double d2 = A[8] - l20 * l20 * d0 - l21 * l21 * d1;
    double y0 = b[0], y1 = b[1] - l10 * y0, y2 = b[2] - l20 * y0 - l21 * y1;
    double z0 = y0 / d0, z1 = y1 / d1, z2 = y2 / d2;
    x[2] = z2; x[1] = z1 - l21 * x[2]; x[0] = z0 - l10 * x[1] - l20 * x[2];
}

int lvo_check_motion(const lvo_pose* first, const lvo_pose* last, const double* first_obs, double translation_threshold)
{   /* Feature::checkMotion (feature.hpp:334-381) */
    double d[3] = {first_obs[0], first_obs[1], 1.0};
    double n = v3_norm(d);
    d[0] /= n; d[1] /= n; d[2] /= n;
    double dw[3]; m3_v(first->R, d, dw);
    double tr[3] = {last->t[0] - first->t[0], last->t[1] - first->t[1], last->t[2] - first->t[2]};
    double par = tr[0] * dw[0] + tr[1] * dw[1] + tr[2] * dw[2];
    double o[3] = {tr[0] - par * dw[0], tr[1] - par * dw[1], tr[2] - par * dw[2]};
    return v3_norm(o) > translation_threshold;
}

int lvo_triangulate(const lvo_pose* poses, const double* obs, int n, int use_position, const double* position_in,
                    double* position_out, double* solution_out, double* inv_depth_out, double* obs_anchor_out)
{   /* feature.hpp:383-552; poses are camera-to-world.  rel[i] = pose_i^-1 * pose_last */
    double (*Rr)[9] = malloc(sizeof(double[9]) * (size_t)n);
    double (*tr)[3] = malloc(sizeof(double[3]) * (size_t)n);
    const lvo_pose* L = &poses[n - 1];
    for (int i = 0; i < n; ++i) {
        double Rt[9]; m3_t(poses[i].R, Rt);
        m3_mul(Rt, L->R, Rr[i]);
        double dt[3] = {L->t[0] - poses[i].t[0], L->t[1] - poses[i].t[1], L->t[2] - poses[i].t[2]};
        /* Isometry inverse: translation = -R^T t ; product: R^T * t_last + (-R^T t_i) */
        double a[3], b[3];
        m3_v(Rt, L->t, a); m3_v(Rt, poses[i].t, b);
        tr[i][0] = a[0] + (-b[0]); tr[i][1] = a[1] + (-b[1]); tr[i][2] = a[2] + (-b[2]);
        (void)dt;
    }
    double ip[3];
    if (!use_position) {
        /* generateInitialGuess(cam_poses[0], z_last, z_first) (feature.hpp:312-331) */
        const double* z1 = obs + 2 * (n - 1); const double* z2 = obs;
        double v[3] = {z1[0], z1[1], 1.0}, m[3];
        m3_v(Rr[0], v, m);
        double A0 = m[0] - z2[0] * m[2], A1 = m[1] - z2[1] * m[2];
        double b0 = z2[0] * tr[0][2] - tr[0][0], b1 = z2[1] * tr[0][2] - tr[0][1];
        double depth = (1.0 / (A0 * A0 + A1 * A1)) * A0 * b0 + (1.0 / (A0 * A0 + A1 * A1)) * A1 * b1;
        ip[0] = z1[0] * depth; ip[1] = z1[1] * depth; ip[2] = depth;
    } else {
        double Lt[9]; m3_t(L->R, Lt);
        double d[3] = {position_in[0], position_in[1], position_in[2]}, a[3], b[3];
        m3_v(Lt, d, a); m3_v(Lt, L->t, b);
        ip[0] = a[0] + (-b[0]); ip[1] = a[1] + (-b[1]); ip[2] = a[2] + (-b[2]);
    }
    double sol[3] = {ip[0] / ip[2], ip[1] / ip[2], 1.0 / ip[2]};
    double lambda = 1e-3;
    int inner = 0, outer = 0, reduced = 0;
    double delta_norm = 0, total_cost = 0.0;
    for (int i = 0; i < n; ++i) { double c; tri_cost(Rr[i], tr[i], sol, obs + 2 * i, &c); total_cost += c; }
    do {
        double A[9] = {0}, b[3] = {0};
        for (int i = 0; i < n; ++i) {
            double J[6], r[2], w;
            tri_jacobian(Rr[i], tr[i], sol, obs + 2 * i, J, r, &w);
            double w2 = (w == 1) ? 1.0 : w * w;
            for (int p = 0; p < 3; ++p) {
                for (int q = 0; q < 3; ++q) {
                    double jtj = J[p] * J[q] + J[3 + p] * J[3 + q];
                    A[p * 3 + q] += (w == 1) ? jtj : w2 * jtj;
                }
                double jtr = J[p] * r[0] + J[3 + p] * r[1];
                b[p] += (w == 1) ? jtr : w2 * jtr;
            }
        }
        do {
            double Ad[9]; memcpy(Ad, A, sizeof Ad);
            Ad[0] += lambda; Ad[4] += lambda; Ad[8] += lambda;
            double delta[3]; solve3_spd(Ad, b, delta);
            double ns[3] = {sol[0] - delta[0], sol[1] - delta[1], sol[2] - delta[2]};
            delta_norm = v3_norm(delta);
            double new_cost = 0.0;
            for (int i = 0; i < n; ++i) { double c; tri_cost(Rr[i], tr[i], ns, obs + 2 * i, &c); new_cost += c; }
            if (new_cost < total_cost) {
                reduced = 1; sol[0] = ns[0]; sol[1] = ns[1]; sol[2] = ns[2]; total_cost = new_cost;
                lambda = lambda / 10 > 1e-10 ? lambda / 10 : 1e-10;
            } else {
                reduced = 0;
                lambda = lambda * 10 < 1e12 ? lambda * 10 : 1e12;
            }
        } while (inner++ < 10 && !reduced);
        inner = 0;
    } while (outer++ < 10 && delta_norm > 5e-7);
    double fp[3] = {sol[0] / sol[2], sol[1] / sol[2], 1.0 / sol[2]};
    int valid = 1;
    for (int i = 0; i < n; ++i) {
        double pz = Rr[i][6] * fp[0] + Rr[i][7] * fp[1] + Rr[i][8] * fp[2] + tr[i][2];
        if (pz <= 0) { valid = 0; break; }
    }
    double normalized_cost = total_cost / (2 * n * n);
    if (normalized_cost > 4.7673e-04) valid = 0;
    if (valid) {
        double pw[3]; m3_v(L->R, fp, pw);
        position_out[0] = pw[0] + L->t[0]; position_out[1] = pw[1] + L->t[1]; position_out[2] = pw[2] + L->t[2];
        solution_out[0] = sol[0]; solution_out[1] = sol[1]; solution_out[2] = sol[2];
        double idp = 1 / fp[2];
        *inv_depth_out = idp;
        obs_anchor_out[0] = fp[0] * idp; obs_anchor_out[1] = fp[1] * idp; obs_anchor_out[2] = 1;
    }
    free(Rr); free(tr);
    return valid;
}

/* ------------------------------------------------------------------------ MSCKF Jacobian + nullspace */
static void msckf_obs_jacobian(const lvo_clone* c, const double* p_w, const double* z, int if_fej,
                               double* Hx /*2x6*/, double* He /*2x6*/, double* Hf /*2x3*/, double* r)
{   /* measurementJacobian_msckf (larvio.cpp:859-921) */
    double R_b2w[9], R_w2b[9], R_w2c[9];
    quat_to_rot(c->q, R_b2w); m3_t(R_b2w, R_w2b); m3_mul(c->R_b2c, R_w2b, R_w2c);
    double tcb_w[3]; m3_v(R_b2w, c->t_c_b, tcb_w);
    double t_c_w[3] = {c->p[0] + tcb_w[0], c->p[1] + tcb_w[1], c->p[2] + tcb_w[2]};
    double pcf[3] = {p_w[0] - t_c_w[0], p_w[1] - t_c_w[1], p_w[2] - t_c_w[2]}, p_c[3];
    m3_v(R_w2c, pcf, p_c);
    double pbf[3];
    for (int i = 0; i < 3; ++i) pbf[i] = if_fej ? p_w[i] - c->p_fej[i] : p_w[i] - c->p[i];
    double dz[6] = {1 / p_c[2], 0, -p_c[0] / (p_c[2] * p_c[2]), 0, 1 / p_c[2], -p_c[1] / (p_c[2] * p_c[2])};
    double S[9], A[9], B[9], C[9];
    skew3(pbf, S);
    m3_mul(R_w2c, S, A);                 /* dpc_dxb.left = R_w2c [p_bf]x */
    m3_mul(A, R_b2w, B);                 /* R_w2c [p_bf]x R_b2w */
    double St[9]; skew3(c->t_c_b, St);
    m3_mul(c->R_b2c, St, C);             /* R_b2c [t_c_b]x */
    double dxb[18], dxe[18];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        dxb[i * 6 + j] = A[i * 3 + j]; dxb[i * 6 + 3 + j] = -R_w2c[i * 3 + j];
        dxe[i * 6 + j] = B[i * 3 + j] - C[i * 3 + j]; dxe[i * 6 + 3 + j] = -c->R_b2c[i * 3 + j];
    }
    for (int i = 0; i < 2; ++i) {
        for (int j = 0; j < 6; ++j) {
            double s1 = 0, s2 = 0;
            for (int k = 0; k < 3; ++k) { s1 += dz[i * 3 + k] * dxb[k * 6 + j]; s2 += dz[i * 3 + k] * dxe[k * 6 + j]; }
            Hx[i * 6 + j] = s1; He[i * 6 + j] = s2;
        }
        for (int j = 0; j < 3; ++j) {
            double s = 0; for (int k = 0; k < 3; ++k) s += dz[i * 3 + k] * R_w2c[k * 3 + j];
            Hf[i * 3 + j] = s;
        }
    }
    r[0] = z[0] - p_c[0] / p_c[2]; r[1] = z[1] - p_c[1] / p_c[2];
}

/* Householder QR of F (rows x nf, row-major, ld nf) applied in place to [G | g] (rows x ng and rows):
 * after the call the LAST rows-nf rows of G,g are A^T G, A^T g with A an orthonormal basis of null(F^T)
 * (the reference takes A from a full-U JacobiSVD, larvio.cpp:973-978: same subspace), the FIRST nf rows are
 * the range part.  F is overwritten by R. */
static void householder_apply(double* F, int rows, int nf, double* G, int ng, int ldg, double* g)
{
    double* v = (double*)malloc(sizeof(double) * (size_t)rows);
    for (int k = 0; k < nf && k < rows; ++k) {
        double nrm2 = 0.;
        for (int i = k; i < rows; ++i) nrm2 += F[i * nf + k] * F[i * nf + k];
        double nrm = sqrt(nrm2);
        if (nrm == 0.) continue;
        double alpha = F[k * nf + k] >= 0. ? -nrm : nrm;
        double vn2 = 0.;
        for (int i = k; i < rows; ++i) { v[i] = F[i * nf + k]; }
        v[k] -= alpha;
        for (int i = k; i < rows; ++i) vn2 += v[i] * v[i];
        if (vn2 == 0.) continue;
        double beta = 2. / vn2;
        for (int c = k; c < nf; ++c) {
            double s = 0.; for (int i = k; i < rows; ++i) s += v[i] * F[i * nf + c];
            s *= beta;
            for (int i = k; i < rows; ++i) F[i * nf + c] -= s * v[i];
        }
        for (int c = 0; c < ng; ++c) {
            double s = 0.; for (int i = k; i < rows; ++i) s += v[i] * G[(size_t)i * ldg + c];
            if (s == 0.) continue;
            s *= beta;
            for (int i = k; i < rows; ++i) G[(size_t)i * ldg + c] -= s * v[i];
        }
        if (g) {
            double s = 0.; for (int i = k; i < rows; ++i) s += v[i] * g[i];
            s *= beta;
            for (int i = k; i < rows; ++i) g[i] -= s * v[i];
        }
    }
    free(v);
}

int lvo_msckf_feature_jacobian(const lvo_clone* clones, const int* clone_rank, const double* obs, const double* obs_vel, int M,
                               const double p_w[3], int N, int leg_dim, int if_fej, int estimate_td, double* H, double* r)
{   /* featureJacobian_msckf (larvio.cpp:924-981) */
    const int rows = 2 * M;
    double* Hx = (double*)calloc((size_t)rows * N, sizeof(double));
    double* Hf = (double*)calloc((size_t)rows * 3, sizeof(double));
    double* rj = (double*)calloc((size_t)rows, sizeof(double));
    for (int i = 0; i < M; ++i) {
        double hx[12], he[12], hf[6], ri[2];
        msckf_obs_jacobian(&clones[clone_rank[i]], p_w, obs + 2 * i, if_fej, hx, he, hf, ri);
        for (int a = 0; a < 2; ++a) {
            double* row = Hx + (size_t)(2 * i + a) * N;
            for (int j = 0; j < 6; ++j) row[leg_dim + 6 * clone_rank[i] + j] = hx[a * 6 + j];
            for (int j = 0; j < 6; ++j) row[15 + j] = he[a * 6 + j];
            if (estimate_td) row[21] = obs_vel[2 * i + a];
            for (int j = 0; j < 3; ++j) Hf[(2 * i + a) * 3 + j] = hf[a * 3 + j];
            rj[2 * i + a] = ri[a];
        }
    }
    householder_apply(Hf, rows, 3, Hx, N, N, rj);
    const int k = rows - 3;
    memcpy(H, Hx + (size_t)3 * N, sizeof(double) * (size_t)k * N);
    memcpy(r, rj + 3, sizeof(double) * (size_t)k);
    free(Hx); free(Hf); free(rj);
    return k;
}

/* ------------------------------------------------------------------------ dense helpers */
extern int lvo_threads_;     /* fe_image.c: host threads for loops with independent iterations (same bits for any count) */
#define LVO_PAR _Pragma("omp parallel for schedule(static) num_threads(lvo_threads_) if(lvo_threads_ > 1)")
/* Cholesky A = L L^T in place (lower), n x n, ld.  Returns 0 on success. */
static int chol_lower(double* A, int n, int ld)
{
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * ld + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * ld + k] * A[(size_t)j * ld + k];
        if (d <= 0.) return 1;
        d = sqrt(d);
        A[(size_t)j * ld + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * ld + j];
            for (int k = 0; k < j; ++k) s -= A[(size_t)i * ld + k] * A[(size_t)j * ld + k];
            A[(size_t)i * ld + j] = s / d;
        }
    }
    return 0;
}
/* solve L Y = B in place, B is n x nb (ld ldb) */
static void trsm_lower(const double* L, int n, int ld, double* B, int nb, int ldb)
{   /* columns of B are independent: chunks of 64 columns may run on different threads */
    const int nchunk = (nb + 63) / 64;
    LVO_PAR
    for (int ch = 0; ch < nchunk; ++ch) {
        const int c0 = ch * 64, c1 = c0 + 64 < nb ? c0 + 64 : nb;
        for (int i = 0; i < n; ++i) {
            double* bi = B + (size_t)i * ldb;
            for (int k = 0; k < i; ++k) {
                const double l = L[(size_t)i * ld + k];
                if (l == 0.) continue;
                const double* bk = B + (size_t)k * ldb;
                for (int c = c0; c < c1; ++c) bi[c] -= l * bk[c];
            }
            const double inv = 1.0 / L[(size_t)i * ld + i];
            for (int c = c0; c < c1; ++c) bi[c] *= inv;
        }
    }
}
/* C (m x n) = A (m x k) * B (k x n), row-major, contiguous inner loop */
static void gemm_nn(const double* A, int lda, const double* B, int ldb, double* C, int ldc, int m, int k, int n)
{
    LVO_PAR
    for (int i = 0; i < m; ++i) {
        double* ci = C + (size_t)i * ldc;
        for (int j = 0; j < n; ++j) ci[j] = 0.;
        for (int p = 0; p < k; ++p) {
            const double a = A[(size_t)i * lda + p];
            if (a == 0.) continue;
            const double* bp = B + (size_t)p * ldb;
            for (int j = 0; j < n; ++j) ci[j] += a * bp[j];
        }
    }
}

double lvo_gating_gamma(const double* H, const double* r, int k, int N, const double* P, int ldp, double sigma2)
{   /* gatingTest (larvio.cpp:1865-1880) */
    double* HP = (double*)malloc(sizeof(double) * (size_t)k * N);
    double* S = (double*)malloc(sizeof(double) * (size_t)k * k);
    double* y = (double*)malloc(sizeof(double) * (size_t)k);
    gemm_nn(H, N, P, ldp, HP, N, k, N, N);
    for (int i = 0; i < k; ++i) for (int j = 0; j < k; ++j) {
        double s = 0.; for (int p = 0; p < N; ++p) s += HP[(size_t)i * N + p] * H[(size_t)j * N + p];
        S[(size_t)i * k + j] = s + (i == j ? sigma2 : 0.);
    }
    double gamma = DBL_MAX;
    if (chol_lower(S, k, k) == 0) {
        memcpy(y, r, sizeof(double) * (size_t)k);
        trsm_lower(S, k, k, y, 1, 1);
        gamma = 0.; for (int i = 0; i < k; ++i) gamma += y[i] * y[i];
    }
    free(HP); free(S); free(y);
    return gamma;
}

void lvo_qr_compress(double* H, double* r, int rows, int cols)
{   /* H <- top `cols` rows of Q^T H, r likewise (larvio.cpp:1430-1445); Householder, column by column */
    double* v = (double*)malloc(sizeof(double) * (size_t)rows);
    for (int k = 0; k < cols && k < rows; ++k) {
        double nrm2 = 0.;
        for (int i = k; i < rows; ++i) nrm2 += H[(size_t)i * cols + k] * H[(size_t)i * cols + k];
        double nrm = sqrt(nrm2);
        if (nrm == 0.) continue;
        double alpha = H[(size_t)k * cols + k] >= 0. ? -nrm : nrm;
        for (int i = k; i < rows; ++i) v[i] = H[(size_t)i * cols + k];
        v[k] -= alpha;
        double vn2 = 0.; for (int i = k; i < rows; ++i) vn2 += v[i] * v[i];
        if (vn2 == 0.) continue;
        double beta = 2. / vn2;
        LVO_PAR
        for (int c = k; c < cols; ++c) {
            double s = 0.; for (int i = k; i < rows; ++i) s += v[i] * H[(size_t)i * cols + c];
            if (s == 0.) continue;
            s *= beta;
            for (int i = k; i < rows; ++i) H[(size_t)i * cols + c] -= s * v[i];
        }
        double s = 0.; for (int i = k; i < rows; ++i) s += v[i] * r[i];
        s *= beta;
        for (int i = k; i < rows; ++i) r[i] -= s * v[i];
    }
    free(v);
}

void lvo_ekf_update(double* P, int N, int ldp, const double* H, int m, const double* r, double sigma2, double* dx)
{   /* S = H P H^T + sigma2 I;  K^T = S^-1 (H P);  dx = K r;  P <- (I - K H) P;  P <- (P + P^T)/2 */
    double* HP = (double*)malloc(sizeof(double) * (size_t)m * N);
    double* S = (double*)malloc(sizeof(double) * (size_t)m * m);
    double* W = (double*)malloc(sizeof(double) * (size_t)m * (N + 1));
    gemm_nn(H, N, P, ldp, HP, N, m, N, N);
    LVO_PAR
    for (int i = 0; i < m; ++i) for (int j = 0; j <= i; ++j) {
        double s = 0.; for (int p = 0; p < N; ++p) s += HP[(size_t)i * N + p] * H[(size_t)j * N + p];
        S[(size_t)i * m + j] = S[(size_t)j * m + i] = s + (i == j ? sigma2 : 0.);
    }
    chol_lower(S, m, m);
    for (int i = 0; i < m; ++i) { memcpy(W + (size_t)i * (N + 1), HP + (size_t)i * N, sizeof(double) * (size_t)N); W[(size_t)i * (N + 1) + N] = r[i]; }
    trsm_lower(S, m, m, W, N + 1, N + 1);           /* W = L^-1 [H P | r] */
    /* dx = (HP)^T S^-1 r = W(:, :N)^T w_r ;   K H P = W^T W */
    for (int j = 0; j < N; ++j) dx[j] = 0.;
    for (int i = 0; i < m; ++i) {
        const double wr = W[(size_t)i * (N + 1) + N];
        const double* wi = W + (size_t)i * (N + 1);
        for (int j = 0; j < N; ++j) dx[j] += wi[j] * wr;
    }
    /* (I-KH)P = P - (HP)^T S^-1 (HP) computed as in the reference order K = (S^-1 HP)^T, then symmetrised */
    double* KHP = (double*)calloc((size_t)N * N, sizeof(double));
    LVO_PAR
    for (int a = 0; a < N; ++a) {                       /* row a of W^T W; every element sums over i in ascending order */
        double* row = KHP + (size_t)a * N;
        for (int i = 0; i < m; ++i) {
            const double* wi = W + (size_t)i * (N + 1);
            const double wa = wi[a];
            if (wa == 0.) continue;
            for (int b = 0; b < N; ++b) row[b] += wa * wi[b];
        }
    }
    for (int a = 0; a < N; ++a) for (int b = 0; b < N; ++b) P[(size_t)a * ldp + b] -= KHP[(size_t)a * N + b];
    for (int a = 0; a < N; ++a) for (int b = a + 1; b < N; ++b) {
        double s = (P[(size_t)a * ldp + b] + P[(size_t)b * ldp + a]) / 2.0;
        P[(size_t)a * ldp + b] = P[(size_t)b * ldp + a] = s;
    }
    free(HP); free(S); free(W); free(KHP);
}
