// be_feature.hip — per-feature kernels of the EKF measurement update for gfx950 (wave64, FP64):
//   k_triangulate  : Levenberg-Marquardt inverse-depth triangulation, one wavefront per feature
//                    (/root/reference/include/larvio/feature.hpp:252-552; sums over views are taken in view
//                    order so that the CPU oracle is reproduced bit-for-bit)
//   k_feature_rows : one workgroup per feature: per-observation reprojection Jacobians
//                    (larvio.cpp:859-921 MSCKF, :1117-1244 1-D inverse depth), Householder left-null-space
//                    projection of the stacked block (:924-981; any orthonormal basis of null(H_f^T) gives the
//                    same chi-square value and the same update), chi-square gate value
//                    gamma = r^T (H P H^T + sigma^2 I)^-1 r (:1865-1880) on the touched columns of P only
//   k_stack_rows   : scatter the accepted compact rows into the dense measurement matrix H_o
// H blocks are kept COMPACT (only the columns a feature touches: extrinsics+td 15..21, the observing clones'
// 6-column blocks, the anchor's block and the feature's own column) — the reference forms dense (2M-3) x N
// blocks that are ~85 % zeros and multiplies them against the full P for every gate.
#include "lvk_internal.h"
#include "be_dev.h"
#include "lvk_wave.h"
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#include <algorithm>

// ========================================================================= triangulation
__global__ void __launch_bounds__(64) k_triangulate(const TriJob* __restrict__ jobs, int n_jobs, const CamPose* __restrict__ cams,
                                                   const int* __restrict__ obs_rank, const double* __restrict__ obs_z, TriResult* __restrict__ out,
                                                   TriResult* __restrict__ out_dev /* optional: a second copy in device memory, read by k_feature_rows (FJ_TRI_PENDING) */)
{
    __shared__ double red[64][13];
    const int jb = blockIdx.x;
    if (jb >= n_jobs) return;
    const TriJob job = jobs[jb];
    const int lane = threadIdx.x & 63, n = job.n;
    // rel_i = pose_i^-1 * pose_last  (feature.hpp:404-422)
    double Rr[9], tr[3], z[2] = {0, 0};
    const CamPose L = cams[obs_rank[job.obs_off + n - 1]];
    if (lane < n) {
        const CamPose Pi = cams[obs_rank[job.obs_off + lane]];
        double Rt[9]; d_m3_t(Pi.R, Rt);
        d_m3_mul(Rt, L.R, Rr);
        double a[3], b[3]; d_m3_v(Rt, L.t, a); d_m3_v(Rt, Pi.t, b);
        tr[0] = a[0] + (-b[0]); tr[1] = a[1] + (-b[1]); tr[2] = a[2] + (-b[2]);
        z[0] = obs_z[2 * (job.obs_off + lane)]; z[1] = obs_z[2 * (job.obs_off + lane) + 1];
    }
    // share view 0's relative pose and the first/last observations (initial guess needs them)
    double R0[9], t0[3], z_first[2], z_last[2];
    for (int k = 0; k < 9; ++k) R0[k] = __shfl(Rr[k], 0);
    for (int k = 0; k < 3; ++k) t0[k] = __shfl(tr[k], 0);
    z_first[0] = __shfl(z[0], 0); z_first[1] = __shfl(z[1], 0);
    z_last[0] = __shfl(z[0], n - 1); z_last[1] = __shfl(z[1], n - 1);
    double ip[3];
    if (!job.use_position) {
        double v[3] = {z_last[0], z_last[1], 1.0}, m[3];
        d_m3_v(R0, v, m);
        double A0 = m[0] - z_first[0] * m[2], A1 = m[1] - z_first[1] * m[2];
        double b0 = z_first[0] * t0[2] - t0[0], b1 = z_first[1] * t0[2] - t0[1];
        double depth = (1.0 / (A0 * A0 + A1 * A1)) * A0 * b0 + (1.0 / (A0 * A0 + A1 * A1)) * A1 * b1;
        ip[0] = z_last[0] * depth; ip[1] = z_last[1] * depth; ip[2] = depth;
    } else {
        double Lt[9]; d_m3_t(L.R, Lt);
        double a[3], b[3]; d_m3_v(Lt, job.position_in, a); d_m3_v(Lt, L.t, b);
        ip[0] = a[0] + (-b[0]); ip[1] = a[1] + (-b[1]); ip[2] = a[2] + (-b[2]);
    }
    double sol[3] = {ip[0] / ip[2], ip[1] / ip[2], 1.0 / ip[2]};
    double lambda = 1e-3;
    int inner = 0, outer = 0, reduced = 0;
    double delta_norm = 0., total_cost = 0.;
    // ordered sum of the per-view costs
    auto cost_sum = [&](const double* x) -> double {
        double c = 0.;
        if (lane < n) c = d_tri_cost(Rr, tr, x, z);
        red[lane][0] = c;
        __syncthreads();
        double s = 0.;
        for (int i = 0; i < n; ++i) s += red[i][0];
        __syncthreads();
        return s;
    };
    total_cost = cost_sum(sol);
    do {
        double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
        {
            double J[6], r[2], w = 1.0;
            if (lane < n) {
                d_tri_jacobian(Rr, tr, sol, z, J, r, w);
                const double w2 = (w == 1) ? 1.0 : w * w;
                for (int p = 0; p < 3; ++p) {
                    for (int q = 0; q < 3; ++q) { double jtj = J[p] * J[q] + J[3 + p] * J[3 + q]; red[lane][p * 3 + q] = (w == 1) ? jtj : w2 * jtj; }
                    double jtr = J[p] * r[0] + J[3 + p] * r[1];
                    red[lane][9 + p] = (w == 1) ? jtr : w2 * jtr;
                }
            }
            __syncthreads();
            for (int i = 0; i < n; ++i) { for (int k = 0; k < 9; ++k) A[k] += red[i][k]; for (int k = 0; k < 3; ++k) b[k] += red[i][9 + k]; }
            __syncthreads();
        }
        do {
            double Ad[9]; for (int k = 0; k < 9; ++k) Ad[k] = A[k];
            Ad[0] += lambda; Ad[4] += lambda; Ad[8] += lambda;
            double delta[3]; d_solve3_spd(Ad, b, delta);
            double ns[3] = {sol[0] - delta[0], sol[1] - delta[1], sol[2] - delta[2]};
            delta_norm = sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
            double new_cost = cost_sum(ns);
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
    const double fp[3] = {sol[0] / sol[2], sol[1] / sol[2], 1.0 / sol[2]};
    int bad = 0;
    if (lane < n) { double pz = Rr[6] * fp[0] + Rr[7] * fp[1] + Rr[8] * fp[2] + tr[2]; bad = pz <= 0; }
    int valid = __ballot(bad) == 0ull;
    const double normalized_cost = total_cost / (2 * n * n);
    if (normalized_cost > 4.7673e-04) valid = 0;
    if (lane == 0) {
        TriResult o;
        o.ok = valid;
        double pw[3]; d_m3_v(L.R, fp, pw);
        o.position[0] = pw[0] + L.t[0]; o.position[1] = pw[1] + L.t[1]; o.position[2] = pw[2] + L.t[2];
        o.solution[0] = sol[0]; o.solution[1] = sol[1]; o.solution[2] = sol[2];
        const double idp = 1 / fp[2];
        o.inv_depth = idp;
        o.obs_anchor[0] = fp[0] * idp; o.obs_anchor[1] = fp[1] * idp; o.obs_anchor[2] = 1;
        const int slot = job.out_slot1 > 0 ? job.out_slot1 - 1 : jb;
        out[slot] = o;
        if (out_dev) out_dev[slot] = o;
    }
}

// ========================================================================= per-feature rows + gate
BE_TICK_DECL(g_fr_tick);
BE_TICK_GETTER(lvk_debug_ticks_feature_rows, g_fr_tick)
#define FR_TICK(k) BE_TICK(g_fr_tick, SMALL && blockIdx.x == 0, k)
// Staging slot of a job (doubles): G [rows_raw x c] | T [rows_raw x c] | r [rows_raw]; ccol ints kept separately.
#define FR_THREADS 128
// SMALL = every job of the batch has <= FRS_ROWS raw rows and <= FRS_COLS compact columns (true in steady state: max_track_len 6
// caps a track at 7 observations): the block [G | r], the projected T and the touched sub-block P_cc live in LDS, P_cc is fetched
// with all loads of a thread in flight (one trip to the memory side instead of one per inner-product term), and [G | r] is written
// to the staging buffer once at the end.  The arithmetic and its order are those of the general path.
#define FRS_ROWS 16
#define FRS_COLS 64
#define FRS_PLD (FRS_COLS + 1)
template <bool SMALL>
__global__ void __launch_bounds__(FR_THREADS) k_feature_rows(const FeatJob* __restrict__ jobs, int n_jobs, const CloneDev* __restrict__ clones,
                                                            const int* __restrict__ obs_rank, const double* __restrict__ obs_z, const double* __restrict__ obs_zv,
                                                            const double* __restrict__ P, int ldp, FilterFlags fl,
                                                            double* __restrict__ staging, int* __restrict__ ccols, FeatResult* __restrict__ out,
                                                            FeatResult* __restrict__ out_host /* optional mirror in device-mapped host memory */,
                                                            double* __restrict__ H_out, int ldh, int ncols_out, double* __restrict__ r_out /* direct output, see the end */,
                                                            int obs_stride /* > 0: job jb's observations sit at [jb * obs_stride, ..): their address does not wait for the job record */,
                                                            int n_clones,
                                                            const TriResult* __restrict__ tri /* optional: results of the triangulation queued ahead of this launch, indexed like the jobs */)
{
    extern __shared__ double sh[];
    const int jb = blockIdx.x;
    if (jb >= n_jobs) return;
    const int t = threadIdx.x;
    // Everything the host staged for this job lives in the upload arena - device memory the host pushes through the BAR since round 4,
    // pinned HOST memory before that (and still, without a large BAR): job record -> observation ranks -> clone poses were three
    // dependent PCIe round trips before the first flop; in device memory they are three dependent HBM reads, which the layout below
    // still folds into one.  SMALL batches lay the observations
    // out at a fixed stride, so ranks / observations (and the whole clone table, a few KB, into LDS) are requested together with
    // the job record: one trip.
    FR_TICK(0);
    int my_rank = 0; double my_z[2] = {0., 0.}, my_zv[2] = {0., 0.};
    const bool pre = SMALL && obs_stride > 0;
    if (pre && t < obs_stride) {
        const int oi = jb * obs_stride + t;
        my_rank = obs_rank[oi]; my_z[0] = obs_z[2 * oi]; my_z[1] = obs_z[2 * oi + 1]; my_zv[0] = obs_zv[2 * oi]; my_zv[1] = obs_zv[2 * oi + 1];
    }
    // A job whose landmark is still being triangulated when the host queues this launch (FJ_TRI_PENDING) takes it from the
    // triangulation's result slot - requested together with the job record, not after it - and counts as rejected when the
    // triangulation failed: the host then never waits for the triangulation before it lays the update out.
    int tri_ok = 1; double tri_p[3] = {0., 0., 0.};
    if (tri) { tri_ok = tri[jb].ok; tri_p[0] = tri[jb].position[0]; tri_p[1] = tri[jb].position[1]; tri_p[2] = tri[jb].position[2]; }
    const FeatJob job = jobs[jb];
    const bool tri_pending = tri && (job.want_gate & FJ_TRI_PENDING);
    if (!tri_pending) tri_ok = 1;
    const double p_w[3] = {tri_pending ? tri_p[0] : job.p_w[0], tri_pending ? tri_p[1] : job.p_w[1], tri_pending ? tri_p[2] : job.p_w[2]};
    const int M = job.n_obs;
    const int rows = 2 * M;
    const int nf = (job.type == JOB_MSCKF) ? 3 : 1;                  // columns of H_f
    const int c = (job.type == JOB_MSCKF) ? 7 + 6 * M : 7 + 6 + 6 * M + 1;
    double* Gg = staging + job.stage_off;                             // staging slot: G [rows x c] | T [rows x c] | r [rows]
    double* rrg = Gg + (size_t)2 * rows * c;
    int* cc = ccols + job.ccol_off;
    // LDS: Hf [rows x 3] | v [rows] | S [k x k] | y [k] | scal[4]  (+ SMALL: G | T | r | P_cc)
    double* Hf = sh;
    double* v = Hf + rows * 3;
    double* S = v + rows;
    double* Gl = S + (size_t)rows * rows + rows + 8;                  // SMALL only
    double* G = SMALL ? Gl : Gg;
    double* T = SMALL ? Gl + FRS_ROWS * FRS_COLS : Gg + (size_t)rows * c;
    double* rr = SMALL ? Gl + 2 * FRS_ROWS * FRS_COLS : rrg;
    double* Pcc = Gl + 2 * FRS_ROWS * FRS_COLS + FRS_ROWS;           // SMALL only: [c][FRS_PLD]
    double* cl_s = Pcc + FRS_COLS * FRS_PLD;                          // SMALL only: the clone table (n_clones x 22 doubles)
    int* srank = (int*)(cl_s + (size_t)n_clones * (sizeof(CloneDev) / sizeof(double)));   // SMALL only: clone rank of every observation
    int* scc = srank + 16;                                            // SMALL only: the compact column map (also written to ccols for k_stack_rows)
    if (SMALL) {
        for (int e = t; e < n_clones * (int)(sizeof(CloneDev) / sizeof(double)); e += FR_THREADS) cl_s[e] = ((const double*)clones)[e];
        if (!pre && t < M) {
            const int oi = job.obs_off + t;
            my_rank = obs_rank[oi]; my_z[0] = obs_z[2 * oi]; my_z[1] = obs_z[2 * oi + 1]; my_zv[0] = obs_zv[2 * oi]; my_zv[1] = obs_zv[2 * oi + 1];
        }
        if (t < M) srank[t] = my_rank;
        __syncthreads();
    }
    FR_TICK(1);                                                      // job, observations and clone table have arrived
    // ---- zero the block, write the compact column map
    for (int e = t; e < rows * c; e += FR_THREADS) G[e] = 0.;
    for (int e = t; e < c; e += FR_THREADS) {
        int col;
        if (e < 7) col = 15 + e;
        else if (job.type == JOB_MSCKF) col = fl.leg_dim + 6 * (SMALL ? srank[(e - 7) / 6] : obs_rank[job.obs_off + (e - 7) / 6]) + (e - 7) % 6;
        else if (e < 13) col = fl.leg_dim + 6 * job.anchor_rank + (e - 7);
        else if (e < 13 + 6 * M) col = fl.leg_dim + 6 * (SMALL ? srank[(e - 13) / 6] : obs_rank[job.obs_off + (e - 13) / 6]) + (e - 13) % 6;
        else col = job.fcol;
        cc[e] = col;
        if (SMALL) scc[e] = col;
    }
    __syncthreads();
    FR_TICK(2);                                                      // G zeroed, column map written
    // P_cc = P[cc, cc]: up to 32 entries per thread, every load issued before the first store - and the stores wait until the
    // Jacobians below are done (their threads would otherwise sit on the loads' return before starting)
    // (index space 64 x 64, thread t keeps column t & 63 and walks rows (t >> 6) + 2 u: no divisions - this kernel runs its code ONCE per
    // launch, so it is paced by instruction fetch, and the 64 unrolled integer divisions of the first version were 2,500 instructions)
    constexpr int PER = SMALL ? (FRS_COLS * FRS_COLS + FR_THREADS - 1) / FR_THREADS : 1;
    double pv[PER];
    const int pj = t & 63, pi0 = t >> 6;
    if (SMALL && (job.want_gate & FJ_GATE) && pj < c) {
        const double* Pj = P + scc[pj];
#pragma unroll
        for (int u = 0; u < PER; ++u) { const int i = pi0 + 2 * u; if (i < c) pv[u] = Pj[(size_t)scc[i] * ldp]; }
    }
    // ---- per-observation Jacobians (one thread per observation)
    if (t < M) {
        const int oi = job.obs_off + t;
        const CloneDev ck = SMALL ? *(const CloneDev*)(cl_s + (size_t)my_rank * (sizeof(CloneDev) / sizeof(double))) : clones[obs_rank[oi]];
        const double z[2] = {SMALL ? my_z[0] : obs_z[2 * oi], SMALL ? my_z[1] : obs_z[2 * oi + 1]};
        const double zvv[2] = {SMALL ? my_zv[0] : obs_zv[2 * oi], SMALL ? my_zv[1] : obs_zv[2 * oi + 1]};
        double Hx[12], He[12], r2[2];
        if (job.type == JOB_MSCKF) {
            double hf[6];
            d_msckf_obs_jacobian(ck, p_w, z, fl.if_fej, Hx, He, hf, r2);
            for (int a = 0; a < 2; ++a) {
                double* row = G + (size_t)(2 * t + a) * c;
                for (int j = 0; j < 6; ++j) row[j] = He[a * 6 + j];
                if (fl.estimate_td) row[6] = zvv[a];
                for (int j = 0; j < 6; ++j) row[7 + 6 * t + j] = Hx[a * 6 + j];
                for (int j = 0; j < 3; ++j) Hf[(2 * t + a) * 3 + j] = hf[a * 3 + j];
                rr[2 * t + a] = r2[a];
            }
        } else {
            const CloneDev ca = SMALL ? *(const CloneDev*)(cl_s + (size_t)job.anchor_rank * (sizeof(CloneDev) / sizeof(double))) : clones[job.anchor_rank];
            double hf[2], Ha[12];
            d_ekf_obs_jacobian(ck, ca, job, z, fl.if_fej, hf, Ha, Hx, He, r2);
            for (int a = 0; a < 2; ++a) {
                double* row = G + (size_t)(2 * t + a) * c;
                for (int j = 0; j < 6; ++j) row[j] = He[a * 6 + j];
                if (fl.estimate_td) row[6] = zvv[a];
                for (int j = 0; j < 6; ++j) row[7 + j] = Ha[a * 6 + j];
                for (int j = 0; j < 6; ++j) row[13 + 6 * t + j] = Hx[a * 6 + j];
                row[c - 1] = hf[a];
                Hf[(2 * t + a) * 3] = hf[a];
                rr[2 * t + a] = r2[a];
            }
        }
    }
    FR_TICK(3);                                                      // Jacobians done
    if (SMALL && (job.want_gate & FJ_GATE) && pj < c) {
#pragma unroll
        for (int u = 0; u < PER; ++u) { const int i = pi0 + 2 * u; if (i < c) Pcc[i * FRS_PLD + pj] = pv[u]; }
    }
    __syncthreads();
    FR_TICK(4);                                                      // P_cc parked
    int first_row = 0, k_rows = rows;
    double h2 = 0.;
    if (job.type != JOB_EKF_TRACKED) {
        // ---- Householder on H_f, applied to [G | r]  (oracle householder_apply / the rotation W of larvio.cpp:2095-2119)
        const int gcols = (job.type == JOB_MSCKF) ? c : c - 1;       // the feature column of an EKF block is H_f itself
        for (int k = 0; k < nf && k < rows; ++k) {
            double nrm2 = 0.;
            for (int i = k; i < rows; ++i) nrm2 += Hf[i * 3 + k] * Hf[i * 3 + k];     // every thread: same ordered sum
            const double nrm = sqrt(nrm2);
            if (nrm == 0.) continue;
            const double alpha = Hf[k * 3 + k] >= 0. ? -nrm : nrm;
            __syncthreads();
            for (int i = k + t; i < rows; i += FR_THREADS) v[i] = Hf[i * 3 + k] - (i == k ? alpha : 0.);
            __syncthreads();
            double vn2 = 0.;
            for (int i = k; i < rows; ++i) vn2 += v[i] * v[i];
            if (vn2 == 0.) continue;
            const double beta = 2. / vn2;
            if (k == 0) h2 = alpha;
            // H_f columns k.. (few): thread q handles column k+q
            if (t < nf - k) {
                const int col = k + t;
                double s = 0.; for (int i = k; i < rows; ++i) s += v[i] * Hf[i * 3 + col];
                s *= beta;
                for (int i = k; i < rows; ++i) Hf[i * 3 + col] -= s * v[i];
            }
            for (int col = t; col < gcols; col += FR_THREADS) {
                double s = 0.; for (int i = k; i < rows; ++i) s += v[i] * G[(size_t)i * c + col];
                if (s == 0.) continue;
                s *= beta;
                for (int i = k; i < rows; ++i) G[(size_t)i * c + col] -= s * v[i];
            }
            if (t == FR_THREADS - 1) {
                double s = 0.; for (int i = k; i < rows; ++i) s += v[i] * rr[i];
                s *= beta;
                for (int i = k; i < rows; ++i) rr[i] -= s * v[i];
            }
            __syncthreads();
        }
        first_row = nf; k_rows = rows - nf;
        if (job.type == JOB_EKF_NEW) {
            // the null rows' feature column is zero after the rotation; the range row keeps H_2 = alpha in it
            for (int i = 1 + t; i < rows; i += FR_THREADS) G[(size_t)i * c + c - 1] = 0.;
            if (t == 0) G[c - 1] = h2;
            __syncthreads();
        }
    }
    FR_TICK(5);                                                      // null-space projection done
    // ---- gate: S = G' P_cc G'^T + sigma2 I on rows first_row.. ; gamma = r'^T S^-1 r'
    double gamma = 0.;
    if ((job.want_gate & FJ_GATE) && k_rows > 0) {
        const int k = k_rows;
        const double* Gp = G + (size_t)first_row * c;
        if (SMALL) {
            // branch-free (a zero entry of G' adds an exact zero); thread t: column t & 63, rows (t >> 6), (t >> 6) + 2, ...
            if (pj < c)
                for (int a = pi0; a < k; a += 2) {
                    double s = 0.;
#pragma unroll 4
                    for (int i = 0; i < c; ++i) s += Gp[(size_t)a * c + i] * Pcc[i * FRS_PLD + pj];
                    T[(size_t)a * c + pj] = s;
                }
        } else {
            for (int e = t; e < k * c; e += FR_THREADS) {
                int a = e / c, j = e - a * c;
                double s = 0.;
                const int colj = cc[j];
                for (int i = 0; i < c; ++i) {
                    const double g = Gp[(size_t)a * c + i];
                    if (g != 0.) s += g * P[(size_t)cc[i] * ldp + colj];
                }
                T[(size_t)a * c + j] = s;
            }
        }
        __syncthreads();
        for (int e = t; e < (SMALL ? 256 : k * k); e += FR_THREADS) {
            int a, b;
            if (SMALL) { a = e >> 4; b = e & 15; if (a >= k || b >= k) continue; } else { a = e / k; b = e - a * k; }
            if (b > a) continue;
            double s = 0.;
#pragma unroll 4
            for (int i = 0; i < c; ++i) s += T[(size_t)a * c + i] * Gp[(size_t)b * c + i];
            S[a * k + b] = s + (a == b ? fl.sigma2 : 0.);
        }
        __syncthreads();
        FR_TICK(6);                                                  // T = G' P_cc and S done
        if (SMALL) {
            // gamma = r'^T S^-1 r' on ONE wavefront: the bordered matrix [S r'; r'^T 0] (k + 1 <= 14 rows) sits in a 16x16 FP64-MFMA
            // accumulator tile and is eliminated by k rank-1 updates (row j of the tile is already laid out as K-slot j & 3 of both
            // operands, see chol32_inv_mfma in be_linalg.hip); what is left in the corner is -r'^T S^-1 r'.  Replaces a column-by-column
            // Cholesky with two barriers per column and a one-thread forward substitution (~4.5 us of the kernel's chain).
            double* y = S + k * k;
            if (t < 64) {
                const int cth = t & 15, gth = t >> 4;
                d4 acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = gth + 4 * r, col = cth;
                    double v;
                    if (row < k && col < k) v = S[(row > col ? row : col) * k + (row > col ? col : row)];
                    else if (row == k && col < k) v = rr[first_row + col];
                    else if (col == k && row < k) v = rr[first_row + row];
                    else v = (row == col && row > k) ? 1.0 : 0.0;
                    acc[r] = v;
                }
                int bad = 0;
#pragma nounroll
                for (int j = 0; j < k; ++j) {                                    // a real loop: the kernel is paced by instruction fetch
                    const int q = j >> 2, gj = j & 3, ln = 16 * gj + j;
                    const double rowv = q == 0 ? acc[0] : q == 1 ? acc[1] : q == 2 ? acc[2] : acc[3];
                    double piv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rowv), ln), __builtin_amdgcn_readlane(__double2loint(rowv), ln));
                    if (!(piv > 0.)) { bad = 1; piv = 1.0; }
                    const double rinv = rsqrt_goldschmidt(piv);
                    const double v = (gth == gj && cth >= j) ? rowv * rinv : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-v, v, acc, 0, 0, 0);
                }
                const int lk = 16 * (k & 3) + k;
                const double corner = (k >> 2) == 0 ? acc[0] : (k >> 2) == 1 ? acc[1] : (k >> 2) == 2 ? acc[2] : acc[3];
                const double g = -__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(corner), lk), __builtin_amdgcn_readlane(__double2loint(corner), lk));
                if (t == 0) y[k] = bad ? 1e300 : g;
            }
            __syncthreads();
            gamma = y[k];
        } else {
        // Cholesky in LDS (column by column), then forward substitution on r'
        int fail = 0;
        for (int j = 0; j < k; ++j) {
            double d = S[j * k + j];
            for (int q = 0; q < j; ++q) d -= S[j * k + q] * S[j * k + q];
            if (!(d > 0.)) { fail = 1; break; }
            d = sqrt(d);
            __syncthreads();
            if (t == 0) S[j * k + j] = d;
            for (int i = j + 1 + t; i < k; i += FR_THREADS) {
                double s = S[i * k + j];
                for (int q = 0; q < j; ++q) s -= S[i * k + q] * S[j * k + q];
                S[i * k + j] = s / d;
            }
            __syncthreads();
        }
        if (fail) gamma = 1e300;
        else {
            double* y = S + k * k;
            if (t == 0) {
                for (int i = 0; i < k; ++i) {
                    double s = rr[first_row + i];
                    for (int q = 0; q < i; ++q) s -= S[i * k + q] * y[q];
                    y[i] = s / S[i * k + i];
                }
                double g = 0.; for (int i = 0; i < k; ++i) g += y[i] * y[i];
                y[k] = g;
            }
            __syncthreads();
            gamma = y[k];
        }
        }
    }
    FR_TICK(7);                                                      // gate value known
    if (SMALL) {
        __syncthreads();
        for (int e = t; e < rows * c; e += FR_THREADS) Gg[e] = G[e];
        for (int e = t; e < rows; e += FR_THREADS) rrg[e] = rr[e];
    }
    FR_TICK(8);                                                      // staging written
    const int accept = (tri_ok && (!(job.want_gate & FJ_GATE) || gamma < job.gate_thr)) ? 1 : 0;      // gamma is the same in every thread (a NaN from a failed triangulation's landmark compares false)
    if (t == 0) {
        FeatResult o; o.gamma = gamma; o.rows = k_rows; o.first_row = first_row; o.c = c; o.h2 = h2;
        o.accept = accept;
        out[jb] = o;
        if (out_host) out_host[jb] = o;
    }
    // ---- direct output: when the host knows the job's slot in the stacked system before the launch (no feature enters the state
    // in this update), the rows go straight into the dense measurement matrix - zero row, then the compact columns scattered, or a
    // zero row with a zero residual for a rejected job (which leaves the update unchanged) - and the stacking launch (6 us + its gap
    // on the update's dependent chain) is not needed.  Same values as k_stack_rows writes.
    if (job.dst_row1 > 0 && H_out) {
        const int d0 = job.dst_row1 - 1;
        const int src0 = (job.type == JOB_EKF_TRACKED) ? 0 : first_row;
        const int nout = (job.type == JOB_EKF_TRACKED) ? 2 : k_rows;
        for (int e = t; e < nout * ncols_out; e += FR_THREADS) { const int a = e / ncols_out, j = e - a * ncols_out; H_out[(size_t)(d0 + a) * ldh + j] = 0.; }
        if (t < nout) r_out[d0 + t] = accept ? rr[src0 + t] : 0.;
        __syncthreads();
        if (accept)
            for (int e = t; e < nout * c; e += FR_THREADS) {
                const int a = e / c, j = e - a * c, col = cc[j];
                if (col >= 0 && col < ncols_out) H_out[(size_t)(d0 + a) * ldh + col] = G[(size_t)(src0 + a) * c + j];
            }
    }
    FR_TICK(9);
}

// dense H row d <- compact row src (job staging) ; r[d] likewise.  One workgroup per destination row.
__global__ void __launch_bounds__(128) k_stack_rows(const StackRow* __restrict__ map, int n_rows, const double* __restrict__ staging,
                                                   const int* __restrict__ ccols, double* __restrict__ H, int ldh, int ncols, double* __restrict__ r,
                                                   const FeatResult* __restrict__ fout)
{
    const int d = blockIdx.x;
    if (d >= n_rows) return;
    const StackRow m = map[d];
    double* row = H + (size_t)m.dst_row * ldh;
    for (int j = threadIdx.x; j < ncols; j += 128) row[j] = 0.;
    if (m.job >= 0 && !fout[m.job].accept) { if (threadIdx.x == 0) r[m.dst_row] = 0.; return; }
    __syncthreads();
    const double* src = staging + m.g_off + (size_t)m.src_row * m.c;
    const int* cc = ccols + m.ccol_off;
    for (int j = threadIdx.x; j < m.c; j += 128) { const int col = cc[j]; if (col >= 0 && col < ncols) row[col] = src[j]; }
    if (threadIdx.x == 0) r[m.dst_row] = staging[m.r_off + m.src_row];
}

// ========================================================================= host launchers (internal)
lvk_status lvk_launch_triangulate(lvk_context* ctx, const TriJob* d_jobs, int n_jobs, const CamPose* d_cams, const int* d_rank, const double* d_z,
                                  TriResult* d_out, TriResult* d_out_dev)
{
    if (n_jobs <= 0) return LVK_OK;
    hipLaunchKernelGGL(k_triangulate, dim3(n_jobs), dim3(64), 0, ctx->stream, d_jobs, n_jobs, d_cams, d_rank, d_z, d_out, d_out_dev);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

lvk_status lvk_launch_feature_rows(lvk_context* ctx, const FeatJob* d_jobs, int n_jobs, int max_rows, const CloneDev* d_clones, const int* d_rank,
                                   const double* d_z, const double* d_zv, const double* d_P, int ldp, FilterFlags fl, double* d_staging,
                                   int* d_ccols, FeatResult* d_out, FeatResult* d_out_host, double* d_Hout, int ldh, int ncols_out, double* d_rout,
                                   int obs_stride, int n_clones, const TriResult* d_tri)
{
    if (n_jobs <= 0) return LVK_OK;
    const size_t base = sizeof(double) * ((size_t)max_rows * 4 + (size_t)max_rows * max_rows + max_rows + 8);
    // max_rows = 2 M_max; compact columns <= 7 + 6 + 6 M_max + 1
    const bool small = max_rows <= FRS_ROWS && 14 + 3 * max_rows <= FRS_COLS;
    const size_t shmem = base + (small ? sizeof(double) * ((size_t)2 * FRS_ROWS * FRS_COLS + FRS_ROWS + (size_t)FRS_COLS * FRS_PLD + (size_t)n_clones * (sizeof(CloneDev) / sizeof(double)) + 8 + FRS_COLS / 2 + 8) : 0);     // + srank (16 ints) + scc (64 ints)
    if (shmem > 150 * 1024) return lvk_set_error(ctx, LVK_ERR_CAPACITY, "feature block with %d rows exceeds the LDS budget", max_rows);
    if (small) {
        if (shmem > 64 * 1024) LVK_LDS_OPTIN(ctx, 10, k_feature_rows<true>, shmem);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_feature_rows<true>), dim3(n_jobs), dim3(FR_THREADS), shmem, ctx->stream, d_jobs, n_jobs, d_clones, d_rank, d_z, d_zv,
                           d_P, ldp, fl, d_staging, d_ccols, d_out, d_out_host, d_Hout, ldh, ncols_out, d_rout, obs_stride, n_clones, d_tri);
    } else {
        LVK_LDS_OPTIN(ctx, 1, k_feature_rows<false>, shmem);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_feature_rows<false>), dim3(n_jobs), dim3(FR_THREADS), shmem, ctx->stream, d_jobs, n_jobs, d_clones, d_rank, d_z, d_zv,
                           d_P, ldp, fl, d_staging, d_ccols, d_out, d_out_host, d_Hout, ldh, ncols_out, d_rout, 0, n_clones, d_tri);
    }
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

lvk_status lvk_launch_stack_rows(lvk_context* ctx, const FeatResult* d_fout, const StackRow* d_map, int n_rows, const double* d_staging, const int* d_ccols, double* d_H, int ldh,
                                 int ncols, double* d_r)
{
    if (n_rows <= 0) return LVK_OK;
    hipLaunchKernelGGL(k_stack_rows, dim3(n_rows), dim3(128), 0, ctx->stream, d_map, n_rows, d_staging, d_ccols, d_H, ldh, ncols, d_r, d_fout);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

// ========================================================================= stage-level C ABI (parity tests, single-stage callers)
double lvk_chi2_005(int dof);

extern "C" lvk_status lvk_triangulate(lvk_context* ctx, const lvk_cam_pose* h_poses, const double* h_obs, int n, int use_position,
                                      const double* h_position_in, int* ok_out, double* h_position, double* h_solution, double* h_inv_depth,
                                      double* h_obs_anchor)
{
    if (!ctx || !h_poses || !h_obs || n < 2 || n > 64 || !ok_out) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_triangulate: 2..64 views required");
    static_assert(sizeof(lvk_cam_pose) == sizeof(CamPose), "lvk_cam_pose is CamPose");
    const size_t o_cams = 0, o_rank = o_cams + sizeof(CamPose) * n, o_z = (o_rank + sizeof(int) * n + 63) & ~(size_t)63, o_job = o_z + sizeof(double) * 2 * n,
                 o_out = (o_job + sizeof(TriJob) + 63) & ~(size_t)63, total = o_out + sizeof(TriResult);
    char* d = (char*)lvk_ctx_scratch(ctx, 9, total);
    if (!d) return lvk_set_error(ctx, LVK_ERR_DEVICE, "scratch allocation failed");
    std::vector<char> h(total, 0);
    memcpy(h.data() + o_cams, h_poses, sizeof(CamPose) * n);
    for (int i = 0; i < n; ++i) ((int*)(h.data() + o_rank))[i] = i;
    memcpy(h.data() + o_z, h_obs, sizeof(double) * 2 * n);
    TriJob* j = (TriJob*)(h.data() + o_job);
    j->n = n; j->use_position = use_position ? 1 : 0; j->obs_off = 0; j->out_slot1 = 0;
    if (h_position_in) memcpy(j->position_in, h_position_in, 24);
    LVK_HIP(ctx, hipMemcpyAsync(d, h.data(), o_out, hipMemcpyHostToDevice, ctx->stream));
    lvk_status st = lvk_launch_triangulate(ctx, (const TriJob*)(d + o_job), 1, (const CamPose*)(d + o_cams), (const int*)(d + o_rank), (const double*)(d + o_z), (TriResult*)(d + o_out), nullptr);
    if (st != LVK_OK) return st;
    TriResult r;
    LVK_HIP(ctx, hipMemcpyAsync(&r, d + o_out, sizeof r, hipMemcpyDeviceToHost, ctx->stream));
    LVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *ok_out = r.ok;
    if (h_position) memcpy(h_position, r.position, 24);
    if (h_solution) memcpy(h_solution, r.solution, 24);
    if (h_inv_depth) *h_inv_depth = r.inv_depth;
    if (h_obs_anchor) memcpy(h_obs_anchor, r.obs_anchor, 24);
    return LVK_OK;
}

extern "C" lvk_status lvk_ekf_gate_and_stack(lvk_context* ctx, const lvk_clone* h_clones, int n_clones, const lvk_msckf_feature* h_feats, int n_feats,
                                             const int* h_clone_rank, const double* h_obs, const double* h_obs_vel, const double* h_P, int N,
                                             int if_fej, int estimate_td, double sigma2, double* h_H, double* h_r, int rows_cap, int* rows_out,
                                             double* h_gamma, int* h_accept)
{
    if (!ctx || !h_clones || n_clones <= 0 || !h_feats || n_feats <= 0 || !h_clone_rank || !h_obs || !h_obs_vel || !h_P || !h_H || !h_r || !rows_out ||
        N < 22 + 6 * n_clones) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_ekf_gate_and_stack: bad argument");
    size_t tot = 0, stage = 0, ccols = 0; int max_rows = 2, cand_rows = 0;
    for (int f = 0; f < n_feats; ++f) {
        const int M = h_feats[f].n_obs;
        if (M < 2 || M > 64 || h_feats[f].obs_off < 0) return lvk_set_error(ctx, LVK_ERR_ARG, "feature %d: 2..64 observations required", f);
        for (int k = 0; k < M; ++k) { const int cr = h_clone_rank[h_feats[f].obs_off + k]; if (cr < 0 || cr >= n_clones) return lvk_set_error(ctx, LVK_ERR_ARG, "clone rank out of range"); }
        tot = std::max(tot, (size_t)h_feats[f].obs_off + M);
        const int c = 7 + 6 * M;
        stage += (size_t)2 * M * c * 2 + 2 * M; ccols += c; max_rows = std::max(max_rows, 2 * M); cand_rows += 2 * M - 3;
    }
    const int ld = (N + 7) & ~7;
    // one input blob, one staging blob, one output blob
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = (o + bytes + 63) & ~(size_t)63; return at; };
    const size_t o_cl = take(sizeof(CloneDev) * n_clones), o_job = take(sizeof(FeatJob) * n_feats), o_rk = take(sizeof(int) * tot), o_z = take(16 * tot),
                 o_zv = take(16 * tot), o_P = take(sizeof(double) * (size_t)N * ld), o_map = take(sizeof(StackRow) * (size_t)std::max(cand_rows, 1)), in_bytes = o;
    char* d_in = (char*)lvk_ctx_scratch(ctx, 9, in_bytes);
    char* d_st = (char*)lvk_ctx_scratch(ctx, 10, sizeof(double) * stage + sizeof(int) * ccols + 64);
    const size_t oH = 0, o_r = (sizeof(double) * (size_t)std::max(cand_rows, 1) * ld + 63) & ~(size_t)63, o_fo = (o_r + sizeof(double) * std::max(cand_rows, 1) + 63) & ~(size_t)63,
                 out_bytes = o_fo + sizeof(FeatResult) * n_feats;
    char* d_out = (char*)lvk_ctx_scratch(ctx, 11, out_bytes);
    if (!d_in || !d_st || !d_out) return lvk_set_error(ctx, LVK_ERR_DEVICE, "scratch allocation failed");
    std::vector<char> h(in_bytes, 0);
    CloneDev* hc = (CloneDev*)(h.data() + o_cl);
    for (int i = 0; i < n_clones; ++i) {
        memcpy(hc[i].q, h_clones[i].q, 32); memcpy(hc[i].p, h_clones[i].p, 24); memcpy(hc[i].p_fej, h_clones[i].p_fej, 24);
        memcpy(hc[i].R_b2c, h_clones[i].R_b2c, 72); memcpy(hc[i].t_c_b, h_clones[i].t_c_b, 24);
    }
    FeatJob* hj = (FeatJob*)(h.data() + o_job);
    size_t s_off = 0, c_off = 0;
    for (int f = 0; f < n_feats; ++f) {
        const int M = h_feats[f].n_obs, c = 7 + 6 * M;
        FeatJob& j = hj[f];
        j.type = JOB_MSCKF; j.n_obs = M; j.obs_off = h_feats[f].obs_off; j.want_gate = 1; j.stage_off = (long long)s_off; j.ccol_off = (int)c_off;
        memcpy(j.p_w, h_feats[f].p_w, 24); memcpy(j.p_fej, h_feats[f].p_w, 24);
        j.gate_thr = lvk_chi2_005(2 * M - 3);
        s_off += (size_t)2 * M * c * 2 + 2 * M; c_off += c;
    }
    memcpy(h.data() + o_rk, h_clone_rank, sizeof(int) * tot);
    memcpy(h.data() + o_z, h_obs, 16 * tot); memcpy(h.data() + o_zv, h_obs_vel, 16 * tot);
    for (int i = 0; i < N; ++i) memcpy(h.data() + o_P + sizeof(double) * (size_t)i * ld, h_P + (size_t)i * N, sizeof(double) * N);
    LVK_HIP(ctx, hipMemcpyAsync(d_in, h.data(), o_map, hipMemcpyHostToDevice, ctx->stream));
    FilterFlags fl; fl.leg_dim = 22; fl.if_fej = if_fej ? 1 : 0; fl.estimate_td = estimate_td ? 1 : 0; fl.pad = 0; fl.sigma2 = sigma2;
    double* d_staging = (double*)d_st; int* d_ccols = (int*)(d_st + ((sizeof(double) * stage + 63) & ~(size_t)63));
    FeatResult* d_fout = (FeatResult*)(d_out + o_fo);
    lvk_status st = lvk_launch_feature_rows(ctx, (const FeatJob*)(d_in + o_job), n_feats, max_rows, (const CloneDev*)(d_in + o_cl), (const int*)(d_in + o_rk),
                                            (const double*)(d_in + o_z), (const double*)(d_in + o_zv), (const double*)(d_in + o_P), ld, fl, d_staging, d_ccols, d_fout, nullptr, nullptr, 0, 0, nullptr, 0, n_clones, nullptr);
    if (st != LVK_OK) return st;
    std::vector<FeatResult> res(n_feats);
    LVK_HIP(ctx, hipMemcpyAsync(res.data(), d_fout, sizeof(FeatResult) * n_feats, hipMemcpyDeviceToHost, ctx->stream));
    LVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // accepted features' rows, in feature order (larvio.cpp:2185-2201 appends only the rows that pass the gate)
    std::vector<StackRow> map; int rows = 0;
    for (int f = 0; f < n_feats; ++f) {
        if (h_gamma) h_gamma[f] = res[f].gamma;
        if (h_accept) h_accept[f] = res[f].accept;
        if (!res[f].accept) continue;
        const int M = h_feats[f].n_obs, c = res[f].c;
        for (int k = 0; k < res[f].rows; ++k) {
            StackRow s; s.g_off = hj[f].stage_off; s.r_off = hj[f].stage_off + (long long)2 * M * c * 2; s.src_row = res[f].first_row + k; s.c = c; s.ccol_off = hj[f].ccol_off;
            s.dst_row = rows + k; s.job = -1; s.pad = 0;
            map.push_back(s);
        }
        rows += res[f].rows;
    }
    *rows_out = rows;
    if (rows > rows_cap) return lvk_set_error(ctx, LVK_ERR_CAPACITY, "%d stacked rows exceed rows_cap %d", rows, rows_cap);
    if (rows == 0) return LVK_OK;
    LVK_HIP(ctx, hipMemcpyAsync(d_in + o_map, map.data(), sizeof(StackRow) * map.size(), hipMemcpyHostToDevice, ctx->stream));
    st = lvk_launch_stack_rows(ctx, d_fout, (const StackRow*)(d_in + o_map), rows, d_staging, d_ccols, (double*)(d_out + oH), ld, N, (double*)(d_out + o_r));
    if (st != LVK_OK) return st;
    LVK_HIP(ctx, hipMemcpy2DAsync(h_H, sizeof(double) * N, d_out + oH, sizeof(double) * ld, sizeof(double) * N, rows, hipMemcpyDeviceToHost, ctx->stream));
    LVK_HIP(ctx, hipMemcpyAsync(h_r, d_out + o_r, sizeof(double) * rows, hipMemcpyDeviceToHost, ctx->stream));
    LVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LVK_OK;
}
