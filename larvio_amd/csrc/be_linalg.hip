// be_linalg.hip — FP64 dense algebra of the EKF measurement update for gfx950 (wave64).
// Replaces the Eigen expressions of /root/reference/src/larvio.cpp:1453-1460,1578-1594 (and the identical
// blocks at :1651-1659,1803-1819 and :2826-2834,2938-2957):
//     S = H P H^T + sigma^2 I ;  K^T = S.ldlt().solve(H P) ;  dx = K r ;  P <- (I - K H) P ;  P <- (P + P^T)/2
// as   HP = H P (MFMA f64) ; S = HP H^T + sigma^2 I (MFMA f64) ; S = L L^T ; W = L^-1 [HP | r] ;
//      dx = W^T w_r ; P <- P - W^T W (MFMA f64, symmetric by construction: K H P = (HP)^T S^-1 (HP)).
// Covariance entries span ~1e-8..1 and parity is 1e-5 relative on P, so everything is FP64; the dense
// contractions use v_mfma_f64_16x16x4_f64 (one 16x16 tile per wavefront, operands straight from L2: the
// matrices are <= ~2 MB and the update is latency-, not bandwidth-bound at N ~ 232).
// All matrices are ROW-MAJOR with a leading dimension in doubles.
#include "lvk_internal.h"
#include "lvk_wave.h"

typedef double d4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------- MFMA f64 GEMM
// C (M x N) = alpha * op(A) (M x K) * op(B) (K x N) + beta * C ; optional diag_add on the diagonal.
// v_mfma_f64_16x16x4_f64: lane l holds A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
// D: col = l&15, row = (l>>4) + 4*reg  (f64 has its own C/D map, cdna_hip_programming.md §3).
// Two optional riders save launches on the update's dependent chain: (xin, xin_col) copies a vector into column xin_col of C
// ([HP | r] in one launch); (xout, xout_col) diverts output column xout_col to a vector, unscaled (W^T [W | w] -> P update and dx).
struct GemmRider { const double* xin; int xin_col; double* xout; int xout_col; double* xout_host = nullptr; double* c00_host = nullptr; };   // xout_host: mirror of xout in device-mapped host memory; c00_host: mirror (16 x 16, row-major) of C's leading tile, likewise
// Split-K: the four wavefronts of a workgroup share ONE 16x16 tile and take every fourth K-chunk (kc <= 64 consecutive k, a
// multiple of 16) each.  The operands come from other XCDs' L2s / the memory side (they were written by the previous kernel), so a
// dependent load costs microseconds and a tile's time is (number of load round trips) x latency, not bytes or flops: a tile of the
// update's products (K = N ~ 216 or K = m ~ 110..260) costs ONE round trip per wavefront (all its loads in flight before the first
// MFMA) instead of the four dependent ones of the tile-per-wavefront kernel of rounds 1-3; the partial tiles meet in LDS and
// wavefront 0 adds them in a fixed order (chunk 0, 1, 2, 3: reproducible).  Inside a chunk lane (i, kk) takes k = k0 + 16 u + 4 kk + q
// (4 consecutive k per lane: contiguous for the row-major operand) - a permutation of the summation index shared by A and B.
template <bool TA, bool TB>
__global__ void __launch_bounds__(256) k_dgemm_sk(int M, int N, int K, int kc, const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                                 double* __restrict__ C, int ldc, double alpha, double beta, double diag_add, GemmRider rd)
{
    __shared__ double red[3][4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = blockIdx.y * 16, col0 = blockIdx.x * 16;
    if (wave == 0 && rd.xin && col0 == 0 && lane < 16 && row0 + lane < M) C[(size_t)(row0 + lane) * ldc + rd.xin_col] = rd.xin[row0 + lane];
    const int i = lane & 15, kk = lane >> 4;
    const int ar = row0 + i, bc = col0 + i;
    const bool a_ok = ar < M, b_ok = bc < N;
    d4 acc = {0., 0., 0., 0.};
    // the current value of C (beta != 0: P -= W^T W, S22 -= L21 L21^T) is requested up front, in the shadow of the operand loads
    double cin[4] = {0., 0., 0., 0.};
    if (wave == 0 && beta != 0.) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = row0 + kk + 4 * r, col = col0 + i; if (row < M && col < N) cin[r] = C[(size_t)row * ldc + col]; }
    }
    for (int k0 = wave * kc; k0 < K; k0 += 4 * kc) {
        double a[16], b[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = k0 + 16 * (u >> 2) + 4 * kk + (u & 3);
            const bool k_ok = 16 * (u >> 2) < kc && k < K;
            a[u] = (a_ok && k_ok) ? (TA ? A[(size_t)k * lda + ar] : A[(size_t)ar * lda + k]) : 0.;
            b[u] = (b_ok && k_ok) ? (TB ? B[(size_t)bc * ldb + k] : B[(size_t)k * ldb + bc]) : 0.;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (16 * (u >> 2) < kc && k0 + 16 * (u >> 2) < K) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
    if (wave) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double sum = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
        const int row = row0 + kk + 4 * r, col = col0 + i;
        if (row < M && col < N) {
            if (rd.xout && col == rd.xout_col) { rd.xout[row] = sum; if (rd.xout_host) rd.xout_host[row] = sum; continue; }
            double v = alpha * sum;
            if (beta != 0.) v += beta * cin[r];
            if (row == col) v += diag_add;
            C[(size_t)row * ldc + col] = v;
            if (rd.c00_host && row0 == 0 && col0 == 0) rd.c00_host[row * 16 + col] = v;
        }
    }
}

template <bool TA, bool TB>
static void launch_dgemm(hipStream_t s, int M, int N, int K, const double* A, int lda, const double* B, int ldb, double* C, int ldc,
                         double alpha, double beta, double diag_add, GemmRider rd = GemmRider{nullptr, 0, nullptr, 0, nullptr})
{
    if (M <= 0 || N <= 0) return;
    const int kc = K >= 193 ? 64 : 16 * ((K + 63) / 64);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dgemm_sk<TA, TB>), dim3((N + 15) / 16, (M + 15) / 16), dim3(256), 0, s, M, N, K, kc, A, lda, B, ldb, C, ldc, alpha, beta, diag_add, rd);
}


// ------------------------------------------------------------------------- structural covariance operations
// Pout[a][b] = Pin[idx[a]][idx[b]]   (clone augmentation larvio.cpp:752-798, row/col deletion :2563-2638, :3311-3327)
__global__ void k_cov_gather(const double* __restrict__ Pin, int ldin, double* __restrict__ Pout, int ldout, const int* __restrict__ idx, int n)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x, a = blockIdx.y;
    if (b >= n || a >= n) return;
    Pout[(size_t)a * ldout + b] = Pin[(size_t)idx[a] * ldin + idx[b]];
}

// (only when the upload arena is pinned HOST memory, i.e. without a large BAR)
// a few KB from the pinned upload arena (host memory) to device memory: one workgroup per table, 8-byte loads all in flight
__global__ void __launch_bounds__(256) k_stage_copy(double* __restrict__ dst0, const double* __restrict__ src0, int n0,
                                                   double* __restrict__ dst1, const double* __restrict__ src1, int n1)
{
    double* dst = blockIdx.x ? dst1 : dst0; const double* src = blockIdx.x ? src1 : src0; const int n = blockIdx.x ? n1 : n0;
    for (int e0 = threadIdx.x; e0 < n; e0 += 256 * 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; if (e < n) v[u] = src[e]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; if (e < n) dst[e] = v[u]; }
    }
}
lvk_status lvk_stage_copy2(lvk_context* ctx, void* d_dst0, const void* d_src0, size_t bytes0, void* d_dst1, const void* d_src1, size_t bytes1)
{
    hipLaunchKernelGGL(k_stage_copy, dim3(2), dim3(256), 0, ctx->stream, (double*)d_dst0, (const double*)d_src0, (int)(bytes0 / sizeof(double)),
                       (double*)d_dst1, (const double*)d_src1, (int)(bytes1 / sizeof(double)));
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

// Propagation AND clone augmentation in one launch (processModel's covariance part, larvio.cpp:553-571, then stateAugmentation's
// J P J^T, :752-798, which is a pure gather):  Pout[a][b] = Pprop[src(a)][src(b)]  with Pprop = the propagated Pin, never stored, and
//     src(a) = a                      a < pose_rows            (IMU block, old clones)
//            = {0,1,2,6,7,8}[a - pose_rows]                    (the new clone's six rows: copies of theta and p)
//            = a - 6                  otherwise                (in-state features)
// 81 % of the outputs (clone x clone, feature blocks) are plain copies; what propagation changes is the L-wide strip of IMU rows and
// columns, so the grid has three roles and only 1 + CPG_STRIPS workgroups read Phi (and one of them Q) from the pinned upload arena:
//   blockIdx <  n_out               row a of Pout, copies where both sources are outside the IMU block
//   next CPG_STRIPS workgroups      a chunk of the outside columns: W = Phi Pin[0:L, chunk], written to every output row / column
//                                   whose source is an IMU row (the six duplicated ones included), both orientations
//   last workgroup                  the IMU block itself: sym(Phi P_II Phi^T + Q), scattered to all (a, b) with both sources inside
// No index arrays: when this was written the arena was HOST memory and every byte a kernel read from it crossed PCIe at ~23 GB/s - the first version of this
// kernel had every one of its 330 workgroups fetch a 1.8 KB index list and took 17.6 us, 8 of them waiting for those bytes
// (profiles/r4_c_be_ticks.json).  Sums run over k ascending.
#define CPG_STRIPS 8
BE_TICK_DECL(g_la_tick);
BE_TICK_GETTER(lvk_debug_ticks_linalg, g_la_tick)
__device__ __forceinline__ int cpg_src(int a, int pose_rows) { return a < pose_rows ? a : a < pose_rows + 6 ? (a - pose_rows < 3 ? a - pose_rows : a - pose_rows + 3) : a - 6; }
// The composed transition of a frame has structure (backend.hip, compose_transition): rows 9.. of Phi are identity rows and Q is zero
// outside its leading 15 x 15 block.  For the 22-dimensional IMU block that is 9 x 22 + 15 x 15 doubles = 3.4 KB: it travels BY VALUE in
// the kernel arguments (device memory with HIP_FORCE_DEV_KERNARG=1, which the runtime section of INTEGRATION.md prescribes) instead of
// being fetched from the pinned arena by nine workgroups (6.8 us of the kernel's 16, profiles/r4_d_be_ticks.json).  With IMU-intrinsics
// calibration (L = 46) the block is too big for the 4 KB argument segment and still comes from the arena.
struct PhiQ22 { double phi[9 * 22]; double q[15 * 15]; };
template <int L, bool BYVAL>
__global__ void __launch_bounds__(256) k_cov_propagate_augment(const double* __restrict__ Pin, int ldin, double* __restrict__ Pout, int ldout,
                                                              int n_out, int pose_rows, const double* __restrict__ phiq, PhiQ22 pq)
{
    extern __shared__ double sh[];
    const int t = threadIdx.x, n_i = L + 6, n_c = n_out - n_i;
    if ((int)blockIdx.x < n_out) {
        const int a = blockIdx.x, i = cpg_src(a, pose_rows);
        if (i < L) return;
        BE_TICK(g_la_tick, blockIdx.x == 100, 0);
        for (int b = t; b < n_out; b += 256) { const int j = cpg_src(b, pose_rows); if (j >= L) Pout[(size_t)a * ldout + b] = Pin[(size_t)i * ldin + j]; }
        BE_TICK(g_la_tick, blockIdx.x == 100, 2);
        return;
    }
    const int role = (int)blockIdx.x - n_out;
    double* Phi = sh;                                   // L x L
    // output index of the ai-th row whose source is an IMU row / of the ci-th one whose source is not
    auto il = [&](int ai) { return ai < L ? ai : pose_rows + (ai - L); };
    auto cl = [&](int ci) { return ci < pose_rows - L ? L + ci : ci + L + 6; };
    auto load_phi = [&]() {
        for (int e = t; e < L * L; e += 256) {
            if (BYVAL) { const int i = e / L, k = e - i * L; Phi[e] = i < 9 ? pq.phi[e] : (i == k ? 1.0 : 0.0); }
            else Phi[e] = phiq[e];
        }
    };
    if (role < CPG_STRIPS) {
        const int cc = (n_c + CPG_STRIPS - 1) / CPG_STRIPS, c0 = role * cc, cw = min(cc, n_c - c0);
        if (cw <= 0) return;
        double* R = sh + L * L;                         // L x cc : Pin[0:L, source columns of the chunk]
        double* W = R + L * cc;                         // L x cc : Phi R
        BE_TICK(g_la_tick, role == 0, 8);
        load_phi();
        for (int e = t; e < L * cw; e += 256) { const int k = e / cw, q = e - k * cw; R[k * cc + q] = Pin[(size_t)k * ldin + cpg_src(cl(c0 + q), pose_rows)]; }
        __syncthreads();
        BE_TICK(g_la_tick, role == 0, 9);
        for (int e = t; e < L * cw; e += 256) {
            const int i = e / cw, q = e - i * cw; double s = 0.;
#pragma unroll
            for (int k = 0; k < L; ++k) s += Phi[i * L + k] * R[k * cc + q];
            W[i * cc + q] = s;
        }
        __syncthreads();
        BE_TICK(g_la_tick, role == 0, 10);
        for (int e = t; e < n_i * cw; e += 256) {
            const int ai = e / cw, q = e - ai * cw, a = il(ai), b = cl(c0 + q);
            const double v = W[cpg_src(a, pose_rows) * cc + q];
            Pout[(size_t)a * ldout + b] = v; Pout[(size_t)b * ldout + a] = v;
        }
        BE_TICK(g_la_tick, role == 0, 11);
        return;
    }
    BE_TICK(g_la_tick, true, 16);
    double* Q = sh + L * L; double* PII = Q + L * L; double* T = PII + L * L; double* Pn = T + L * L;
    load_phi();
    for (int e = t; e < L * L; e += 256) {
        const int i = e / L, j = e - i * L;
        if (BYVAL) Q[e] = (i < 15 && j < 15) ? pq.q[i * 15 + j] : 0.0; else Q[e] = phiq[L * L + e];
        PII[e] = Pin[(size_t)i * ldin + j];
    }
    __syncthreads();
    BE_TICK(g_la_tick, true, 17);
    for (int e = t; e < L * L; e += 256) {
        const int i = e / L, j = e - i * L; double s = 0.;
#pragma unroll
        for (int k = 0; k < L; ++k) s += Phi[i * L + k] * PII[k * L + j];
        T[e] = s;
    }
    __syncthreads();
    for (int e = t; e < L * L; e += 256) {
        const int i = e / L, j = e - i * L;
        if (j > i) continue;
        double s1 = 0., s2 = 0.;
#pragma unroll
        for (int k = 0; k < L; ++k) { s1 += T[i * L + k] * Phi[j * L + k]; s2 += T[j * L + k] * Phi[i * L + k]; }
        const double v = ((s1 + Q[i * L + j]) + (s2 + Q[j * L + i])) / 2.0;
        Pn[i * L + j] = v; Pn[j * L + i] = v;
    }
    __syncthreads();
    for (int e = t; e < n_i * n_i; e += 256) {
        const int ai = e / n_i, bi = e - ai * n_i, a = il(ai), b = il(bi);
        Pout[(size_t)a * ldout + b] = Pn[cpg_src(a, pose_rows) * L + cpg_src(b, pose_rows)];
    }
    BE_TICK(g_la_tick, true, 18);
}

// re-anchoring of a 1-D inverse-depth feature (updateFeatureCov_1didp, larvio.cpp:3125-3293): row/col fc <- J P, J P J^T
__global__ void __launch_bounds__(256) k_cov_reanchor(double* __restrict__ P, int ld, int n, const double* __restrict__ J, int fc)
{
    extern __shared__ double pf[];     // n
    __shared__ int nz_idx[64]; __shared__ double nz_val[64]; __shared__ int nnz_s;
    const int t = threadIdx.x;
    // J has ~19 non-zeros (feature, two clone blocks, extrinsics): ordered compaction (ballot prefix per wave)
    __shared__ int wave_cnt[8];
    if (t == 0) nnz_s = 0;
    for (int base = 0; base < n; base += 256) {
        const int k = base + t;
        const double v = k < n ? J[k] : 0.;
        const unsigned long long m = __ballot(v != 0.);
        if ((t & 63) == 0) wave_cnt[t >> 6] = (int)__popcll(m);
        __syncthreads();
        int off = nnz_s;
        for (int q = 0; q < (t >> 6); ++q) off += wave_cnt[q];
        if (v != 0.) { const int slot = off + (int)__popcll(m & ((1ull << (t & 63)) - 1ull)); if (slot < 64) { nz_idx[slot] = k; nz_val[slot] = v; } }
        __syncthreads();
        if (t == 0) nnz_s = min(nnz_s + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3], 64);
        __syncthreads();
    }
    const int nnz = nnz_s;
    for (int b = t; b < n; b += 256) { double s = 0.; for (int q = 0; q < nnz; ++q) s += nz_val[q] * P[(size_t)nz_idx[q] * ld + b]; pf[b] = s; }
    __syncthreads();
    for (int b = t; b < n; b += 256) if (b != fc) { P[(size_t)fc * ld + b] = pf[b]; P[(size_t)b * ld + fc] = pf[b]; }
    if (t == 0) { double s = 0.; for (int q = 0; q < nnz; ++q) s += pf[nz_idx[q]] * nz_val[q]; P[(size_t)fc * ld + fc] = s; }
}

// delayed initialisation of new in-state features (larvio.cpp:1821-1854), 1-D: HH = diag(H2)^-1 H1 (nn x n);
// rows/cols n..n+nn-1 of P:  P_new,old = -HH P ; P_new,new = HH P HH^T + sigma2 / H2^2 (diag).  tmp: nn x n scratch.
// All three pieces are dot products of length n (~200) behind a fresh P: every wavefront walks k with its lanes' loads in flight
// and finishes with a DPP reduction, instead of one thread per output walking k alone (20-30 us each, three launches).
// step 1: tmp[j][b] = -(HH P)[j][b], written to the new rows and columns of P; the extra workgroup column (blockIdx.x == gridDim.x - 1)
// computes dx_new = -HH dx + H2^-1 r1.  grid (ceil(n/64) + 1, nn), 256 threads: wavefront w takes k = w, w+4, ...
__global__ void __launch_bounds__(256) k_cov_append_rows(double* __restrict__ P, int ld, int n, int nn, const double* __restrict__ H1, int ldh,
                                                        const double* __restrict__ H2, double* __restrict__ tmp,
                                                        const double* __restrict__ r1, const double* __restrict__ dx, double* __restrict__ dx_new)
{
    extern __shared__ double hh[];     // row j of HH (n), then 4 x 64 partial sums
    double* part = hh + n;
    const int j = blockIdx.y, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const double inv = 1.0 / H2[j];
    for (int k = t; k < n; k += 256) hh[k] = H1[(size_t)j * ldh + k] * inv;
    __syncthreads();
    if (blockIdx.x == gridDim.x - 1) {                       // dx_new[j]
        double s = 0.;
        for (int k = t; k < n; k += 256) s += hh[k] * dx[k];
        s = wave_sum_f64(s);
        if (lane == 0) part[wave] = s;
        __syncthreads();
        if (t == 0) dx_new[j] = -(part[0] + part[1] + part[2] + part[3]) + r1[j] * inv;
        return;
    }
    const int b = blockIdx.x * 64 + lane;
    double s = 0.;
    if (b < n) {
        int k = wave;
        for (; k + 12 < n; k += 16) {                        // four loads in flight per lane
            const double p0 = P[(size_t)k * ld + b], p1 = P[(size_t)(k + 4) * ld + b], p2 = P[(size_t)(k + 8) * ld + b], p3 = P[(size_t)(k + 12) * ld + b];
            s += hh[k] * p0; s += hh[k + 4] * p1; s += hh[k + 8] * p2; s += hh[k + 12] * p3;
        }
        for (; k < n; k += 4) s += hh[k] * P[(size_t)k * ld + b];
    }
    part[wave * 64 + lane] = s;
    __syncthreads();
    if (wave == 0 && b < n) {
        const double v = -(((part[lane] + part[64 + lane]) + part[128 + lane]) + part[192 + lane]);
        tmp[(size_t)j * n + b] = v;
        P[(size_t)(n + j) * ld + b] = v; P[(size_t)b * ld + n + j] = v;
    }
}
// step 2: P_new,new = HH P HH^T + sigma2 / H2^2 (symmetrised).  one workgroup; wavefront per (j, l <= j) pair, lanes over k
__global__ void __launch_bounds__(256) k_cov_append_corner(double* __restrict__ P, int ld, int n, int nn, const double* __restrict__ H1, int ldh,
                                                          const double* __restrict__ H2, double sigma2, const double* __restrict__ tmp)
{
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int npair = nn * (nn + 1) / 2;
    for (int e = wave; e < npair; e += 4) {
        int j = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while ((j + 1) * (j + 2) / 2 <= e) ++j;
        while (j * (j + 1) / 2 > e) --j;
        const int l = e - j * (j + 1) / 2;
        const double il = 1.0 / H2[l], ij = 1.0 / H2[j];
        double s1 = 0., s2 = 0.;
        for (int k = lane; k < n; k += 64) { s1 += tmp[(size_t)j * n + k] * (H1[(size_t)l * ldh + k] * il); s2 += tmp[(size_t)l * n + k] * (H1[(size_t)j * ldh + k] * ij); }
        s1 = wave_sum_f64(s1); s2 = wave_sum_f64(s2);
        if (lane == 0) {
            const double dg = (j == l) ? sigma2 * (1.0 / (H2[j] * H2[j])) : 0.0;
            const double v = ((-s1 + dg) + (-s2 + dg)) / 2.0;
            P[(size_t)(n + j) * ld + n + l] = v; P[(size_t)(n + l) * ld + n + j] = v;
        }
    }
}


// ------------------------------------------------------------------------- Cholesky + forward substitution, multi-workgroup
#define CP_NB 32
__device__ __forceinline__ double readlane_f64(double v, int lane)
{   // lane is a compile-time constant after unrolling: two v_readlane_b32
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rsqrt_refined(double x)
{   // v_rsq_f64 seed + two Newton steps (full double precision for SPD pivots); replaces sqrt + divide on the pivot chain
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}

// One wavefront: Y = L^-1 for the Cholesky factor L of the 32x32 block in D (full symmetric block; identity padding beyond nb).
// Nothing downstream reads L_pp itself (the solves are products with L_pp^-1), so only Y is produced.
// The block lives in three FP64-MFMA accumulator tiles (lane (g = lane >> 4, c = lane & 15), register r <-> T[g + 4 r][c]):
// a00 = D[0:16, 0:16], a01 = D[0:16, 16:32] (read from the lower triangle), a11 = D[16:32, 16:32].  Row j of a tile sits in the 16
// lanes of group j & 3, register j >> 2 - which is exactly where v_mfma_f64_16x16x4_f64 wants K-slot j & 3 of BOTH operands - and
// the block is symmetric, so row j IS column j: one pivot step is
//     piv = D[j][j] (one v_readlane pair) ; rinv = rsq(piv) refined ; v = row j * rinv masked to its group and to c >= j ;
//     a00 -= v (x) v ; a01 -= v (x) u          (rank-1 updates as MFMAs whose other three K-slots are zero)
// with no per-element broadcasts at all (the register version spent 2 v_readlane + 1 FMA per (j, k) pair: ~6 us per block, this
// is ~2.7).  The columns' lower halves u_j (rows 16..31) of four consecutive pivots fill the four K-slots of one operand register,
// so a11 gets a rank-4 update per 4 pivots and L10 Y00 needs no data movement either.  The inverse runs in the MFMA shadow of
// the factorisation: step j scales row j of Y by rinv and subtracts L[i][j] (i > j) times it from the rows below - the same
// operand registers again.  Y10 = -Y11 (L10 Y00) closes the block (one LDS round trip for the transposed operand).
// What paces the block is instruction ISSUE on the one wavefront that owns the chain (~60 wave64 instructions per pivot at >= 4 cycles
// each: 6.4 us per 32 pivots, profiles/r4_e_be_ticks.json), not the matrix core and not the dependent chain alone: two variants were
// built and measured against this one and dropped - the inverse on a second wavefront fed through LDS (the hand-over's writes cost
// the leader what the MFMAs it shed had cost: 6.7 + 1.3 us, r4_g_be_ticks.json) and an LDL^T chain with the square roots deferred to
// the end (5.7 + 1.1 us, r4_h_be_ticks.json).
__device__ __forceinline__ void chol32_inv_mfma(double (*D)[CP_NB + 1], double (*Y)[CP_NB + 1], int nb, int lane, int* __restrict__ info, int j0, bool report)
{
    const int c = lane & 15, g = lane >> 4;
    d4 a00, a01, a11, y00, y11, lop;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        a00[r] = D[g + 4 * r][c];
        a01[r] = D[16 + c][g + 4 * r];
        a11[r] = D[16 + g + 4 * r][16 + c];
        y00[r] = (g + 4 * r == c) ? 1.0 : 0.0; y11[r] = y00[r]; lop[r] = 0.;
    }
    int first_bad = -1;                                  // wave-uniform: the first non-positive pivot of this block (reported once, after the loop)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        double piv = readlane_f64(half ? a11[0] : a00[0], 0);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            // pivots in the identity padding (the last panel of a matrix whose size is not a multiple of 32: half a panel on average,
            // ~3 us of the ~28 an update's factorisation takes) change nothing - D and Y keep their identity rows, L10's padding rows are
            // zero - and are skipped (wave-uniform scalar branch)
            if (16 * half + j >= nb) continue;
            const int q = j >> 2, gj = j & 3;
            d4& aa = half ? a11 : a00; d4& yy = half ? y11 : y00;
            const bool okp = piv > 0.;
            first_bad = (!okp && first_bad < 0 && 16 * half + j < nb) ? 16 * half + j : first_bad;
            const double pv = okp ? piv : 1.0;
            const double rinv = rsqrt_goldschmidt(pv);
            const bool rs = g == gj;
            const double v = (rs && c >= j) ? aa[q] * rinv : 0.0;                 // L[c][j] (the diagonal is piv * rinv = sqrt(piv))
            if (j < 15) {
                // the NEXT pivot, D[j+1][j+1] - L[j+1][j]^2, taken from registers while the rank-1 update below is still in the matrix
                // core (one product, one rounding: the same value the update leaves in the tile): its rsqrt chain overlaps the MFMA
                const double l1 = readlane_f64(v, 16 * gj + j + 1);
                piv = __builtin_fma(-l1, l1, readlane_f64(aa[(j + 1) >> 2], 16 * ((j + 1) & 3) + j + 1));
            }
            aa = __builtin_amdgcn_mfma_f64_16x16x4f64(-v, v, aa, 0, 0, 0);
            if (!half) {
                const double u = rs ? a01[q] * rinv : 0.0;                        // L[16 + c][j]
                a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(-v, u, a01, 0, 0, 0);
                lop[q] = rs ? u : lop[q];
            }
            yy[q] = rs ? yy[q] * rinv : yy[q];
            const double vm = (c > j) ? v : 0.0, yb = rs ? yy[q] : 0.0;
            yy = __builtin_amdgcn_mfma_f64_16x16x4f64(-vm, yb, yy, 0, 0, 0);
            if (!half && gj == 3) a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(-lop[q], lop[q], a11, 0, 0, 0);
        }
    }
    if (first_bad >= 0 && report && lane == 0 && info[0] == 0) info[0] = j0 + first_bad + 1;
    // T = L10 Y00 (operands are lop and y00 as they stand), then Y10 = -Y11 T with Y11 read back transposed from LDS
    d4 tt = {0., 0., 0., 0.};
#pragma unroll
    for (int q = 0; q < 4; ++q) tt = __builtin_amdgcn_mfma_f64_16x16x4f64(lop[q], y00[q], tt, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) { Y[g + 4 * r][c] = y00[r]; Y[16 + g + 4 * r][16 + c] = y11[r]; Y[g + 4 * r][16 + c] = 0.; }
    __builtin_amdgcn_wave_barrier();
    d4 yl = {0., 0., 0., 0.};
#pragma unroll
    for (int q = 0; q < 4; ++q) yl = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[16 + c][16 + 4 * q + g], tt[q], yl, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Y[16 + g + 4 * r][c] = -yl[r];
    __builtin_amdgcn_wave_barrier();
}

// Variant 2 of the block above (round 6; built, measured, NOT the default - LVK_CHOL_VARIANT=2 selects it): TWO pivots per step.  What a pivot costs is
// a dependent chain - read the pivot, 1/sqrt of it (a seed and two coupled refinements), scale its row, rank-1 update in the matrix core -
// and the next pivot cannot start its 1/sqrt before that update's entry is known: 178 ns per pivot, 5.7 us per 32-pivot block
// (profiles/r4_h_be_ticks.json).  For the leading 2 x 2 block [a b; b c] of what is left, both reciprocal roots come from numbers that are
// known NOW: 1/L11 = rsqrt(a) and 1/L22 = rsqrt(a c - b^2) * (a * rsqrt(a)) - two chains side by side instead of one after the other.
// Rows j and j + 1 of a tile live in lane groups j & 3 and (j & 3) + 1 of the SAME register (j even), which is where the matrix core
// wants K-slots j & 3 and (j & 3) + 1: v = [L[:, j] in group gj | L[:, j+1] in group gj + 1] makes the two rank-1 updates ONE MFMA per
// tile.  Column j + 1 needs column j in the neighbouring lane group, L[c][j+1] = (A[c][j+1] - L[j+1][j] L[c][j]) / L22: one
// v_permlane16_swap_b32 per 32-bit half (gfx950: swaps the odd 16-lane rows of one operand with the even rows of the other; with the same
// value in both, row 2k is copied into row 2k + 1 - tools/gpu/permlane_probe.hip).  The inverse's rows follow the same pattern.
// Rounding differs from the single-pivot chain in the last bits (a c - b^2 instead of c - (b / sqrt a)^2, and the look-ahead entries
// are formed in registers); parity with the oracle's update is asked at 1e-5 and measured at 1e-9 either way (78 back-end tests + smoke).
// MEASURED (profiles/r6_k_chol_two_pivot_ab.json, r6_l_be_ticks_chol_variant*.json): no gain - 5.2 us per 32-pivot block either way,
// k_chol_fused 27.0 against 26.3 us.  The 40 matrix-core instructions it sheds per block are paid back by ~400 vector instructions (the
// second 1/sqrt chain, the look-ahead's seven 64-bit v_readlane pairs, the lane-group moves): the block is paced by instruction ISSUE
// on its one wavefront (~1450 instructions per block in both forms), not by the depth of the pivot chain.
__device__ __forceinline__ double row_even_to_odd_f64(double x)
{
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]);
}
__device__ __forceinline__ void chol32_inv_mfma2(double (*D)[CP_NB + 1], double (*Y)[CP_NB + 1], int nb, int lane, int* __restrict__ info, int j0, bool report)
{
    const int c = lane & 15, g = lane >> 4;
    d4 a00, a01, a11, y00, y11, lop;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        a00[r] = D[g + 4 * r][c];
        a01[r] = D[16 + c][g + 4 * r];
        a11[r] = D[16 + g + 4 * r][16 + c];
        y00[r] = (g + 4 * r == c) ? 1.0 : 0.0; y11[r] = y00[r]; lop[r] = 0.;
    }
    int first_bad = -1;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        d4& aa = half ? a11 : a00; d4& yy = half ? y11 : y00;
        // the leading 2 x 2 block of the half: [pa pb; pb pc]
        double pa = readlane_f64(aa[0], 0), pb = readlane_f64(aa[0], 16 + 0), pc = readlane_f64(aa[0], 16 + 1);
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            // nb is the number of real rows of the 32-row block; beyond it D is the identity and nothing changes (see the single-pivot variant)
            if (16 * half + j >= nb) continue;
            const int q = j >> 2, gj = j & 3;                             // gj is 0 or 2
            const bool ok1 = pa > 0.;
            first_bad = (!ok1 && first_bad < 0) ? 16 * half + j : first_bad;
            const double a_ = ok1 ? pa : 1.0;                             // a pivot that is not positive is replaced by 1 (as in the single-pivot chain)
            const double b_ = pb;
            const double det = __builtin_fma(a_, pc, -(b_ * b_));
            const bool real2 = 16 * half + j + 1 < nb;                     // an odd nb: the second pivot of the last pair is identity padding (b = 0, c = 1: left exactly alone)
            const bool ok2 = det > 0.;
            first_bad = (!ok2 && real2 && first_bad < 0) ? 16 * half + j + 1 : first_bad;
            const double det_ = ok2 ? det : a_;                           // L22 = 1 for a pivot that is not positive
            const double rinv1 = rsqrt_goldschmidt(a_);
            const double rdet = rsqrt_goldschmidt(det_);
            const double l21 = real2 ? b_ * rinv1 : 0.0;                  // L[j+1][j]
            const double inv22 = real2 ? rdet * (a_ * rinv1) : 1.0;       // 1 / L[j+1][j+1] = sqrt(a) / sqrt(a c - b^2)
            const bool r1 = g == gj, r2 = g == gj + 1;
            // column j in group gj, column j + 1 in group gj + 1 (both from register q: rows j and j + 1 of the tile, by symmetry its columns)
            const double v1 = (r1 && c >= j) ? aa[q] * rinv1 : 0.0;
            const double v1n = row_even_to_odd_f64(v1);                   // column j, seen from group gj + 1
            const double v2 = (r2 && real2 && c >= j + 1) ? __builtin_fma(-l21, v1n, aa[q]) * inv22 : 0.0;
            const double v = r1 ? v1 : v2;                                // zero outside the two groups
            if (j < 14) {
                // the next pair's 2 x 2 block from registers, while the rank-2 update is in the matrix core
                const int n0 = j + 2, n1 = j + 3;
                const double x0 = readlane_f64(v, 16 * gj + n0), x1 = readlane_f64(v, 16 * gj + n1);                // L[n0][j], L[n1][j]
                const double z0 = readlane_f64(v, 16 * (gj + 1) + n0), z1 = readlane_f64(v, 16 * (gj + 1) + n1);    // L[n0][j+1], L[n1][j+1]
                const double t00 = readlane_f64(aa[n0 >> 2], 16 * (n0 & 3) + n0), t10 = readlane_f64(aa[n0 >> 2], 16 * (n0 & 3) + n1),
                             t11 = readlane_f64(aa[n1 >> 2], 16 * (n1 & 3) + n1);
                pa = __builtin_fma(-z0, z0, __builtin_fma(-x0, x0, t00));
                pb = __builtin_fma(-z1, z0, __builtin_fma(-x1, x0, t10));
                pc = __builtin_fma(-z1, z1, __builtin_fma(-x1, x1, t11));
            }
            aa = __builtin_amdgcn_mfma_f64_16x16x4f64(-v, v, aa, 0, 0, 0);
            if (!half) {
                const double u1 = r1 ? a01[q] * rinv1 : 0.0;                                      // L[16 + c][j]
                const double u1n = row_even_to_odd_f64(u1);
                const double u2 = r2 ? __builtin_fma(-l21, u1n, a01[q]) * inv22 : 0.0;            // L[16 + c][j + 1]
                const double u = r1 ? u1 : u2;
                a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(-v, u, a01, 0, 0, 0);
                lop[q] = (r1 || r2) ? u : lop[q];
            }
            // the inverse: row j scaled, row j + 1 = (row j + 1 - L21 row j) / L22, rows below minus their two multiples
            const double y1 = r1 ? yy[q] * rinv1 : 0.0;
            const double y1n = row_even_to_odd_f64(y1);
            const double y2 = r2 ? __builtin_fma(-l21, y1n, yy[q]) * inv22 : 0.0;
            yy[q] = r1 ? y1 : (r2 ? y2 : yy[q]);
            const double vm = (c > j + 1) ? v : 0.0, yb = (r1 || r2) ? yy[q] : 0.0;
            yy = __builtin_amdgcn_mfma_f64_16x16x4f64(-vm, yb, yy, 0, 0, 0);
            if (!half && gj == 2) a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(-lop[q], lop[q], a11, 0, 0, 0);
        }
    }
    if (first_bad >= 0 && report && lane == 0 && info[0] == 0) info[0] = j0 + first_bad + 1;
    d4 tt = {0., 0., 0., 0.};
#pragma unroll
    for (int q = 0; q < 4; ++q) tt = __builtin_amdgcn_mfma_f64_16x16x4f64(lop[q], y00[q], tt, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) { Y[g + 4 * r][c] = y00[r]; Y[16 + g + 4 * r][16 + c] = y11[r]; Y[g + 4 * r][16 + c] = 0.; }
    __builtin_amdgcn_wave_barrier();
    d4 yl = {0., 0., 0., 0.};
#pragma unroll
    for (int q = 0; q < 4; ++q) yl = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[16 + c][16 + 4 * q + g], tt[q], yl, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Y[16 + g + 4 * r][c] = -yl[r];
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------- fused Cholesky + solve, m <= 160: ONE launch
// The per-panel launches above spend ~19 us per 32 columns, of which only a few us are the dependent pivot chain of the diagonal block;
// the rest is launch latency, first-touch trips to the memory side and the left-looking re-read of S.  Here workgroup 0 keeps the whole
// lower triangle of S in LDS (32x32 blocks) and factors it right-looking with the critical path first:
//   C1  wavefront 0 factors + inverts diagonal block p (chol32_inv_mfma) while wavefronts 1..3 finish panel p-1 off the critical
//       path: publish L[:, p-1] and L_pp^-1 to global memory and raise the flag (ONE wavefront pays the agent-scope release - the
//       L2 write-back used to sit on every panel of the chain), and the trailing update of everything right of block column p;
//   C2  all four: L[p+1.., p] = A Y^T ;   C3  all four: diagonal block p+1 -= L[p+1, p] L[p+1, p]^T ;  then C1 of panel p+1.
// Workgroups 1.. own 16 columns of B = [HP | r] per wavefront, keep their W rows in LDS, and trail the factorisation by one
// panel: B_p -= L[p, 0:p] W[0:p] while the diagonal block is still being factored, then W_p = L_pp^-1 B_p as soon as the flag
// arrives.  Workgroup 0 never waits for anybody, so the spinning solvers cannot starve it; their spin is bounded all the same
// (CF_SPIN_MAX polls: a factor workgroup that never shows up - a broken device - ends the launch with info[1] set instead of hanging it).
#define CF_MAXB 5                                   // m <= 160
#define CF_LD (CP_NB + 1)
#define CF_WLD 17
#define CF_SPIN_MAX (1 << 21)
typedef double cf_blk[CP_NB][CF_LD];
__device__ __forceinline__ int cf_idx(int bi, int bj) { return bi * (bi + 1) / 2 + bj; }
// C[tile tr,tc of block Cb] -= A[rows of tile tr] . Bm[rows of tile tc]^T over K = 32
__device__ __forceinline__ void cf_tile_sub(cf_blk& Cb, const cf_blk& A, const cf_blk& Bm, int tr, int tc, int i16, int kk)
{
    d4 acc = {0., 0., 0., 0.};
#pragma unroll
    for (int k0 = 0; k0 < CP_NB; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[16 * tr + i16][k0 + kk], Bm[16 * tc + i16][k0 + kk], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Cb[16 * tr + kk + 4 * r][16 * tc + i16] -= acc[r];
}
__device__ __forceinline__ bool cf_wait(const int* flag, int target, int lane, int* __restrict__ info)
{
    int ok = 1;
    if (lane == 0) {
        int polls = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++polls > CF_SPIN_MAX) { ok = 0; info[1] = target; break; }
        }
    }
    ok = __builtin_amdgcn_readfirstlane(ok);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok != 0;
}
// what wavefronts 1..3 of the factor workgroup do for panel p while wavefront 0 factors diagonal block p+1 (or after the last panel)
__device__ __forceinline__ void cf_deferred(cf_blk* Sb, cf_blk& Yp, int p, int nblk, int m, double* __restrict__ S, int lds_, double* __restrict__ Yg,
                                            int* __restrict__ flag, int base, int wave, int lane)
{
    const int i16 = lane & 15, kk = lane >> 4;
    if (wave == 3) {
        // publish Y_pp = L_pp^-1 and the blocks of L below the diagonal (the diagonal blocks of L are not stored: nothing reads them)
        double* yg = Yg + (size_t)p * CP_NB * CP_NB;
        for (int e = lane; e < CP_NB * CP_NB / 4; e += 64) {
            const int a = e >> 3, b4 = (e & 7) * 4;
            d4 v; v[0] = Yp[a][b4]; v[1] = Yp[a][b4 + 1]; v[2] = Yp[a][b4 + 2]; v[3] = Yp[a][b4 + 3];
            *(d4*)(yg + a * CP_NB + b4) = v;
        }
        for (int bi = p + 1; bi < nblk; ++bi) {
            const cf_blk& A = Sb[cf_idx(bi, p)];
            for (int e = lane; e < CP_NB * CP_NB / 4; e += 64) {
                const int a = e >> 3, b4 = (e & 7) * 4, gr = 32 * bi + a;
                d4 v; v[0] = A[a][b4]; v[1] = A[a][b4 + 1]; v[2] = A[a][b4 + 2]; v[3] = A[a][b4 + 3];
                if (gr < m) *(d4*)(S + (size_t)gr * lds_ + 32 * p + b4) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_store(flag, base + p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // trailing update with panel p of everything right of block column p+1 and below block row p+1 (block column p+1 itself included;
    // diagonal block p+1 was done on the critical path).  Of a diagonal block only the tiles chol32_inv_mfma reads: (0,0) (1,0) (1,1).
    // Wavefront 3 (busy publishing first) takes one tile in five.
    int u = 0;
    for (int bi = p + 2; bi < nblk; ++bi)
        for (int bj = p + 1; bj <= bi; ++bj)
            for (int tt = 0; tt < 4; ++tt) {
                if (bi == bj && tt == 1) continue;
                const int owner = (u % 5 == 4) ? 3 : 1 + ((u % 5) >> 1);
                ++u;
                if (owner == wave) cf_tile_sub(Sb[cf_idx(bi, bj)], Sb[cf_idx(bi, p)], Sb[cf_idx(bj, p)], tt >> 1, tt & 1, i16, kk);
            }
}

// A second right-hand side (B2: m x nbcols2, own leading dimension) gets its own solver workgroups behind those of B: the caller
// passes the block of S to the RIGHT of the diagonal block being factored (rows 0..m-1, the columns of the rows still to come), so
// that the launch also produces L21^T = L11^-1 A21^T for a matrix taller than 160 rows (launch_chol_solve).  row0 = the first row
// of this diagonal block in the whole matrix (error reporting).
template <int CHV>      // 1: one pivot per step (chol32_inv_mfma, the default), 2: two (chol32_inv_mfma2; LVK_CHOL_VARIANT=2)
__global__ void __launch_bounds__(256) k_chol_fused(double* __restrict__ S, int lds_, int m, double* __restrict__ B, int ldb, int nbcols,
                                                   double* __restrict__ B2, int ldb2, int nbcols2, int row0,
                                                   double* __restrict__ Yg, int* __restrict__ flag, int base, int* __restrict__ info)
{
    extern __shared__ __attribute__((aligned(32))) double cf_smem[];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, i16 = lane & 15, kk = lane >> 4;
    const int nblk = (m + CP_NB - 1) / CP_NB;
    const d4 zero4 = {0., 0., 0., 0.};
    if (blockIdx.x == 0) {
        BE_TICK(g_la_tick, true, 22);
        // ===================================================================== factor role
        cf_blk* Sb = (cf_blk*)cf_smem;                                   // lower blocks, cf_idx(bi, bj)
        cf_blk* Yb = (cf_blk*)(cf_smem + (size_t)(CF_MAXB * (CF_MAXB + 1) / 2) * CP_NB * CF_LD);    // two: panel p's inverse is published while p+1's is being built
        {   // whole lower triangle: one 32-byte load per thread and block, all in flight
            const int row = t >> 3, c4 = (t & 7) * 4;
            d4 v[CF_MAXB * (CF_MAXB + 1) / 2];
#pragma unroll
            for (int bi = 0; bi < CF_MAXB; ++bi)
#pragma unroll
                for (int bj = 0; bj <= bi; ++bj) {
                    const int gr = 32 * bi + row, gc = 32 * bj + c4;
                    v[cf_idx(bi, bj)] = (bi < nblk && gr < m && gc + 3 < lds_) ? *(const d4*)(S + (size_t)gr * lds_ + gc) : zero4;
                }
#pragma unroll
            for (int bi = 0; bi < CF_MAXB; ++bi)
#pragma unroll
                for (int bj = 0; bj <= bi; ++bj) {
                    if (bi >= nblk) continue;
                    const int gr = 32 * bi + row;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int gc = 32 * bj + c4 + q;
                        Sb[cf_idx(bi, bj)][row][c4 + q] = (gr < m && gc < m) ? v[cf_idx(bi, bj)][q] : (gr == gc ? 1.0 : 0.0);   // identity padding
                    }
                }
        }
        __syncthreads();
        BE_TICK(g_la_tick, true, 23);
        for (int p = 0; p < nblk; ++p) {
            cf_blk& Yi = Yb[p & 1];
            // ---- C1
            if (wave == 0) {
                if constexpr (CHV == 2) chol32_inv_mfma2(Sb[cf_idx(p, p)], Yi, min(CP_NB, m - 32 * p), lane, info, row0 + 32 * p, true);
                else chol32_inv_mfma(Sb[cf_idx(p, p)], Yi, min(CP_NB, m - 32 * p), lane, info, row0 + 32 * p, true);
                BE_TICK(g_la_tick, true, 24 + 3 * p);
            }
            else if (p > 0) cf_deferred(Sb, Yb[(p - 1) & 1], p - 1, nblk, m, S, lds_, Yg, flag, base, wave, lane);
            __syncthreads();
            BE_TICK(g_la_tick, true, 25 + 3 * p);
            // ---- C2: X = A Y^T for the blocks below the diagonal, 16 rows x 32 columns per unit (block p+1's two units first)
            for (int u = wave; u < 2 * (nblk - p - 1); u += 4) {
                const int bi = p + 1 + (u >> 1), tr = u & 1;
                cf_blk& A = Sb[cf_idx(bi, p)];
                d4 o0 = {0., 0., 0., 0.}, o1 = {0., 0., 0., 0.};
#pragma unroll
                for (int k0 = 0; k0 < CP_NB; k0 += 4) {
                    const double a = A[16 * tr + i16][k0 + kk];
                    o0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Yi[i16][k0 + kk], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Yi[16 + i16][k0 + kk], o1, 0, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int lr = 16 * tr + kk + 4 * r; A[lr][i16] = o0[r]; A[lr][16 + i16] = o1[r]; }
            }
            if (p + 1 < nblk) {
                __syncthreads();
                BE_TICK(g_la_tick, true, 26 + 3 * p);
                // ---- C3: the next diagonal block, one tile per wavefront ((0,1) is never read)
                if (wave != 1) cf_tile_sub(Sb[cf_idx(p + 1, p + 1)], Sb[cf_idx(p + 1, p)], Sb[cf_idx(p + 1, p)], wave >> 1, wave & 1, i16, kk);
                __syncthreads();
            }
        }
        if (wave != 0) cf_deferred(Sb, Yb[(nblk - 1) & 1], nblk - 1, nblk, m, S, lds_, Yg, flag, base, wave, lane);
        BE_TICK(g_la_tick, true, 40);
        if (t == 0) { BE_TICK(g_la_tick, true, 41); }
#ifdef LVK_BE_TIMING
        if (t == 0) g_la_tick[42] = (unsigned long long)m;
#endif
        return;
    }
    // ========================================================================= solver role: 16 columns of B per wavefront
    double (*Wl)[CF_WLD] = (double (*)[CF_WLD])(cf_smem + (size_t)wave * (CF_MAXB * CP_NB + CP_NB) * CF_WLD);    // W rows of the finished panels
    double (*Cs)[CF_WLD] = Wl + CF_MAXB * CP_NB;                                                              // staging of the current panel
    const int nwg1 = (nbcols + 63) / 64;
    int cb = (int)blockIdx.x - 1;
    if (cb >= nwg1) { cb -= nwg1; B = B2; ldb = ldb2; nbcols = nbcols2; }
    const int col = cb * 64 + wave * 16 + i16;
    const bool colok = col < nbcols;
    if (cb * 64 + wave * 16 >= nbcols) return;
    double own[CF_MAXB][2][4];
#pragma unroll
    for (int q = 0; q < CF_MAXB; ++q)
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 32 * q + 16 * tl + kk + 4 * r;
                own[q][tl][r] = (q < nblk && row < m && colok) ? B[(size_t)row * ldb + col] : 0.;
            }
#pragma unroll
    for (int p = 0; p < CF_MAXB; ++p) {
        if (p >= nblk) break;
        d4 c0 = {0., 0., 0., 0.}, c1 = {0., 0., 0., 0.};
        if (p > 0) {
            // B_p -= L[p, 0:32p] W[0:32p]; the flag of panel p-1 (waited for below) covers every block of this row of L
            const int r0 = 32 * p + i16, r1 = r0 + 16;
            d4 a0[2 * (CF_MAXB - 1)], a1[2 * (CF_MAXB - 1)];
#pragma unroll
            for (int u = 0; u < 2 * (CF_MAXB - 1); ++u) {
                a0[u] = (u < 2 * p && r0 < m) ? *(const d4*)(S + (size_t)r0 * lds_ + 16 * u + 4 * kk) : zero4;
                a1[u] = (u < 2 * p && r1 < m) ? *(const d4*)(S + (size_t)r1 * lds_ + 16 * u + 4 * kk) : zero4;
            }
#pragma unroll
            for (int u = 0; u < 2 * (CF_MAXB - 1); ++u) {
                if (u >= 2 * p) break;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double b = Wl[16 * u + 4 * kk + q][i16];
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u][q], b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u][q], b, c1, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { Cs[kk + 4 * r][i16] = own[p][0][r] - c0[r]; Cs[16 + kk + 4 * r][i16] = own[p][1][r] - c1[r]; }
        if (!cf_wait(flag, base + p + 1, lane, info)) return;
        const double* yg = Yg + (size_t)p * CP_NB * CP_NB;
        d4 y0[2], y1[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { y0[u] = *(const d4*)(yg + (size_t)i16 * CP_NB + 16 * u + 4 * kk); y1[u] = *(const d4*)(yg + (size_t)(16 + i16) * CP_NB + 16 * u + 4 * kk); }
        __builtin_amdgcn_wave_barrier();
        d4 o0 = {0., 0., 0., 0.}, o1 = {0., 0., 0., 0.};
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double b = Cs[16 * u + 4 * kk + q][i16];
                o0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0[u][q], b, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y1[u][q], b, o1, 0, 0, 0);
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int l0 = kk + 4 * r, l1 = 16 + l0, g0 = 32 * p + l0, g1 = 32 * p + l1;
            Wl[g0][i16] = o0[r]; Wl[g1][i16] = o1[r];
            if (colok && g0 < m) B[(size_t)g0 * ldb + col] = o0[r];
            if (colok && g1 < m) B[(size_t)g1 * ldb + col] = o1[r];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// S = L L^T (lower, in place) and B <- L^-1 B for B (m x nbcols); m arbitrary.
// m <= 160: one fused launch.  Taller: blocked recursion over 160-row super-panels, each ONE fused launch + two GEMM launches -
//   [S11 . ; S21 S22]:  the fused launch factors S11 = L11 L11^T, solves W1 = L11^-1 B1 and - through the second right-hand side, the
//   block S12 = S21^T that k_dgemm left to the right of S11 (S is computed in full) - L21^T = L11^-1 S21^T, in place;
//   then S22 -= L21 L21^T and B2 -= L21 W1 on the FP64 matrix cores, and the recursion continues on (S22, B2).
static lvk_status launch_chol_solve(lvk_context* ctx, double* S, int lds_, int m, double* B, int ldb, int nbcols, int* info)
{
    hipStream_t s = ctx->stream;
    const int MB = CF_MAXB * CP_NB;
    const size_t ybytes = sizeof(double) * CF_MAXB * CP_NB * CP_NB;
    const bool fresh = ctx->scratch_bytes[12] < ybytes + 64;
    char* ws = (char*)lvk_ctx_scratch(ctx, 12, ybytes + 64);
    if (!ws) return lvk_set_error(ctx, LVK_ERR_DEVICE, "scratch allocation failed");
    int* flag = (int*)(ws + ybytes);
    if (fresh || ctx->chol_epoch > (1 << 27)) { LVK_HIP(ctx, hipMemsetAsync(flag, 0, 64, s)); ctx->chol_epoch = 0; }
    const size_t lds_f = sizeof(double) * (size_t)(CF_MAXB * (CF_MAXB + 1) / 2 + 2) * CP_NB * CF_LD;
    const size_t lds_s = sizeof(double) * (size_t)4 * (CF_MAXB * CP_NB + CP_NB) * CF_WLD;
    const size_t shm = lds_f > lds_s ? lds_f : lds_s;
    // Residency: 1 + ceil(nbcols / 64) (+ ceil(rest / 64)) workgroups (<= 12 at the largest state the filter accepts), one per CU
    // because of their LDS (144 KB of the CU's 160: the launch has one size, the larger of the two roles'); the factor workgroup
    // never waits for anybody, so solvers that were dispatched ahead of it only spin until it gets a CU - no ordering assumption
    // is needed for progress, only for speed.
    static const int chv = [] { const char* e = getenv("LVK_CHOL_VARIANT"); return (e && atoi(e) == 2) ? 2 : 1; }();
    if (chv == 2) LVK_LDS_OPTIN(ctx, 8, k_chol_fused<2>, shm); else LVK_LDS_OPTIN(ctx, 13, k_chol_fused<1>, shm);
    for (int off = 0; off < m; off += MB) {
        const int mb = (m - off) < MB ? (m - off) : MB, rest = m - off - mb;
        double* S11 = S + (size_t)off * lds_ + off; double* B1 = B + (size_t)off * ldb;
        double* S12 = S11 + mb;                       // rows off..off+mb-1, columns off+mb..m-1: A21^T
        const int base = 8 * ctx->chol_epoch++;
        if (chv == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chol_fused<2>), dim3(1 + (nbcols + 63) / 64 + (rest + 63) / 64), dim3(256), shm, s, S11, lds_, mb, B1, ldb, nbcols,
                                         S12, lds_, rest, off, (double*)ws, flag, base, info);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chol_fused<1>), dim3(1 + (nbcols + 63) / 64 + (rest + 63) / 64), dim3(256), shm, s, S11, lds_, mb, B1, ldb, nbcols,
                                S12, lds_, rest, off, (double*)ws, flag, base, info);
        LVK_LAUNCH_CHECK(ctx);
        if (rest > 0) {
            double* S22 = S + (size_t)(off + mb) * lds_ + off + mb; double* B2 = B + (size_t)(off + mb) * ldb;
            launch_dgemm<true, false>(s, rest, rest, mb, S12, lds_, S12, lds_, S22, lds_, -1.0, 1.0, 0.0);        // S22 -= L21 L21^T
            launch_dgemm<true, false>(s, rest, nbcols, mb, S12, lds_, B1, ldb, B2, ldb, -1.0, 1.0, 0.0);          // B2 -= L21 W1
        }
    }
    return LVK_OK;
}

// ------------------------------------------------------------------------- host drivers (internal + C ABI)
struct UpdateWs { double* B; int ldb; double* S; int lds; int* info; hipEvent_t ev_a = nullptr, ev_b = nullptr; double* dx_host = nullptr; double* p00_host = nullptr; };   // ev_*: optional bracket around the H P GEMM; dx_host: host-mapped mirror of dx; p00_host: of the updated P's leading 16 x 16 block

// dx (device, n) and P updated in place.  B: m x (n+1) workspace, S: m x m workspace.
lvk_status lvk_update_core(lvk_context* ctx, double* P, int ldp, int n, const double* H, int ldh, int m, const double* r, double sigma2,
                           double* dx, UpdateWs ws)
{
    if (m <= 0) { LVK_HIP(ctx, hipMemsetAsync(dx, 0, sizeof(double) * (size_t)n, ctx->stream)); return LVK_OK; }
    hipStream_t s = ctx->stream;
    if (ws.ev_a) hipEventRecord(ws.ev_a, s);
    launch_dgemm<false, false>(s, m, n, n, H, ldh, P, ldp, ws.B, ws.ldb, 1.0, 0.0, 0.0, GemmRider{r, n, nullptr, 0, nullptr});   // [HP | r]
    if (ws.ev_b) hipEventRecord(ws.ev_b, s);
    launch_dgemm<false, true>(s, m, m, n, ws.B, ws.ldb, H, ldh, ws.S, ws.lds, 1.0, 0.0, sigma2);           // S = HP H^T + sigma2 I
    { lvk_status cs = launch_chol_solve(ctx, ws.S, ws.lds, m, ws.B, ws.ldb, n + 1, ws.info); if (cs != LVK_OK) return cs; }                                    // S = L L^T ; W = L^-1 [HP | r]
    // W^T [W | w]: columns 0..n-1 update P (P -= W^T W), column n is dx = W^T w
    launch_dgemm<true, false>(s, n, n + 1, m, ws.B, ws.ldb, ws.B, ws.ldb, P, ldp, -1.0, 1.0, 0.0, GemmRider{nullptr, 0, dx, n, ws.dx_host, ws.p00_host});
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

lvk_status lvk_cov_gather(lvk_context* ctx, const double* Pin, int ldin, double* Pout, int ldout, const int* d_idx, int n)
{
    hipLaunchKernelGGL(k_cov_gather, dim3((n + 127) / 128, n), dim3(128), 0, ctx->stream, Pin, ldin, Pout, ldout, d_idx, n);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}
// h_phi, h_q: the composed L x L transition and noise matrices on the HOST (row-major); d_phiq: their copy in the upload arena ([Phi | Q]),
// used only when L != 22
lvk_status lvk_cov_propagate_augment(lvk_context* ctx, const double* Pin, int ldin, double* Pout, int ldout, int n_out, int pose_rows, int L,
                                     const double* h_phi, const double* h_q, const double* d_phiq)
{
    const int n_c = n_out - L - 6, cc = (n_c + CPG_STRIPS - 1) / CPG_STRIPS;
    const size_t strip = sizeof(double) * ((size_t)L * L + (size_t)2 * L * (cc > 0 ? cc : 1)), core = sizeof(double) * (size_t)5 * L * L;
    const size_t shmem = strip > core ? strip : core;
    if (shmem > 160 * 1024) return lvk_set_error(ctx, LVK_ERR_CAPACITY, "covariance dimension %d too large for the propagate kernel", n_out);
    const dim3 grid(n_out + CPG_STRIPS + 1);
    if (L == 22) {
        PhiQ22 pq;
        memcpy(pq.phi, h_phi, sizeof pq.phi);
        for (int i = 0; i < 15; ++i) memcpy(pq.q + 15 * i, h_q + (size_t)22 * i, sizeof(double) * 15);
        if (shmem > 64 * 1024) LVK_LDS_OPTIN(ctx, 3, (k_cov_propagate_augment<22, true>), shmem);     // strips of > ~1400 clone / feature columns
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cov_propagate_augment<22, true>), grid, dim3(256), shmem, ctx->stream, Pin, ldin, Pout, ldout, n_out, pose_rows, (const double*)nullptr, pq);
    } else if (L == 46) {
        if (shmem > 64 * 1024) LVK_LDS_OPTIN(ctx, 9, (k_cov_propagate_augment<46, false>), shmem);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cov_propagate_augment<46, false>), grid, dim3(256), shmem, ctx->stream, Pin, ldin, Pout, ldout, n_out, pose_rows, d_phiq, PhiQ22());
    } else return lvk_set_error(ctx, LVK_ERR_UNSUPPORTED, "IMU block of %d states", L);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}
lvk_status lvk_cov_reanchor(lvk_context* ctx, double* P, int ld, int n, const double* d_J, int fc)
{
    hipLaunchKernelGGL(k_cov_reanchor, dim3(1), dim3(256), sizeof(double) * (size_t)n, ctx->stream, P, ld, n, d_J, fc);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}
lvk_status lvk_cov_append_features(lvk_context* ctx, double* P, int ld, int n, int nn, const double* H1, int ldh, const double* H2, const double* r1,
                                   const double* dx, double sigma2, double* tmp, double* dx_new)
{
    if (nn <= 0) return LVK_OK;
    hipLaunchKernelGGL(k_cov_append_rows, dim3((n + 63) / 64 + 1, nn), dim3(256), sizeof(double) * ((size_t)n + 256), ctx->stream, P, ld, n, nn, H1, ldh, H2, tmp, r1, dx, dx_new);
    hipLaunchKernelGGL(k_cov_append_corner, dim3(1), dim3(256), 0, ctx->stream, P, ld, n, nn, H1, ldh, H2, sigma2, (const double*)tmp);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

extern "C" lvk_status lvk_ekf_update(lvk_context* ctx, double* d_P, int ldp, int n, const double* d_H, int ldh, int m, const double* d_r,
                                     double sigma2, double* d_dx)
{
    if (!ctx || !d_P || !d_dx || n <= 0 || m < 0 || (m > 0 && (!d_H || !d_r)) || ldp < n || ldh < n)
        return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_ekf_update: bad argument");
    UpdateWs ws;
    ws.ldb = (n + 1 + 7) & ~7; ws.lds = (m + 7) & ~7;
    ws.B = (double*)lvk_ctx_scratch(ctx, 4, sizeof(double) * (size_t)(m > 0 ? m : 1) * ws.ldb);
    ws.S = (double*)lvk_ctx_scratch(ctx, 5, sizeof(double) * (size_t)(m > 0 ? m : 1) * (ws.lds > 0 ? ws.lds : 8));
    ws.info = (int*)lvk_ctx_scratch(ctx, 6, 64);
    if (!ws.B || !ws.S || !ws.info) return lvk_set_error(ctx, LVK_ERR_DEVICE, "scratch allocation failed");
    LVK_HIP(ctx, hipMemsetAsync(ws.info, 0, 64, ctx->stream));
    lvk_status st = lvk_update_core(ctx, d_P, ldp, n, d_H, ldh, m, d_r, sigma2, d_dx, ws);
    if (st != LVK_OK) return st;
    // the stage-level call reports the factorisation's health itself (the frame-level filter reads the same words from mapped memory
    // at the sync it needs anyway): this entry point waits for its update
    int info[2] = {0, 0};
    LVK_HIP(ctx, hipMemcpyAsync(info, ws.info, sizeof info, hipMemcpyDeviceToHost, ctx->stream));
    LVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (info[1]) return lvk_set_error(ctx, LVK_ERR_DEVICE, "lvk_ekf_update: a solver workgroup waited for panel %d of the factorisation in vain", info[1]);
    if (info[0]) return lvk_set_error(ctx, LVK_ERR_NUMERIC, "lvk_ekf_update: H P H^T + sigma2 I is not positive definite (pivot %d)", info[0] - 1);
    return LVK_OK;
}

extern "C" lvk_status lvk_dgemm(lvk_context* ctx, int transa, int transb, int M, int N, int K, double alpha, const double* d_A, int lda,
                                const double* d_B, int ldb, double beta, double* d_C, int ldc)
{
    if (!ctx || !d_A || !d_B || !d_C) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_dgemm: bad argument");
    hipStream_t s = ctx->stream;
    if (!transa && !transb) launch_dgemm<false, false>(s, M, N, K, d_A, lda, d_B, ldb, d_C, ldc, alpha, beta, 0.0);
    else if (!transa && transb) launch_dgemm<false, true>(s, M, N, K, d_A, lda, d_B, ldb, d_C, ldc, alpha, beta, 0.0);
    else if (transa && !transb) launch_dgemm<true, false>(s, M, N, K, d_A, lda, d_B, ldb, d_C, ldc, alpha, beta, 0.0);
    else launch_dgemm<true, true>(s, M, N, K, d_A, lda, d_B, ldb, d_C, ldc, alpha, beta, 0.0);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}
