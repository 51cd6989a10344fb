// be_qr_dense.hip — dense compression of a tall measurement block: [H | r] (rows x cols) -> top min(rows, cols) rows of Q^T [H | r],
// in place.  The fallback behind the structure-aware path of be_qr.hip (blocks whose column unions are too wide for one LDS node -
// tracks longer than ~20 clones - and the stage-level entry point lvk_ekf_compress_qr, which knows nothing about structure).
// Replaces the SuiteSparse SPQR call of /root/reference/src/larvio.cpp:1430-1445,2209-2229 by a communication-avoiding blocked
// Householder QR (CAQR):
//   for every panel of NB columns
//     k_caqr_factor  level 1: each chunk of CH = 16384 / NB rows is factored on its own workgroup, panel in LDS (column-major, one
//                             wavefront per column, DPP wave reductions, one barrier per column); reflectors V and the compact-WY
//                             factor T go to a workspace, R_c stays in the chunk's top NB rows
//                    level 2: the stacked R_c (top rows of every chunk) are factored the same way by one workgroup; its R is the
//                             panel's final R, already in place (rows j0 .. j0 + NB of the matrix)
//     k_caqr_apply   level 1 / level 2: C <- (I - V T^T V^T) C on the columns right of the panel and the residual, on the FP64
//                             matrix cores (v_mfma_f64_16x16x4_f64): W = V^T C with the chunk's rows as the K dimension (V staged in
//                             LDS once per workgroup, C tiles straight from global, coalesced 128-byte row segments), W <- T^T W,
//                             C -= V W
// i.e. every matrix element is touched O(cols / NB) times instead of O(cols) times (the level-2 BLAS kernel this replaces read the
// whole trailing matrix once per column with a stride of one row per lane), the panel work runs on all chunks in parallel, and the
// O(rows cols^2) flops are MFMA GEMMs.  NB = 32 for rows <= 8192, NB = 16 up to 65536 rows.
#include "lvk_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double cq_dpp_ror(double v, const int sel)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (sel) {
        case 8: lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xF, 0xF, false); break;
        case 4: lo = __builtin_amdgcn_update_dpp(0, lo, 0x124, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x124, 0xF, 0xF, false); break;
        case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x122, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x122, 0xF, 0xF, false); break;
        default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x121, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x121, 0xF, 0xF, false); break;
    }
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double cq_wave_sum(double v)
{
    v += cq_dpp_ror(v, 8); v += cq_dpp_ror(v, 4); v += cq_dpp_ror(v, 2); v += cq_dpp_ror(v, 1);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double s = 0.;
#pragma unroll
    for (int r = 0; r < 4; ++r) s += __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * r), __builtin_amdgcn_readlane(lo, 16 * r));
    return s;
}

// Row s of a node.  level 1: node = chunk c, rows j0 + c CH + s.  level 2: the stacked top rows, s = c NB + i -> j0 + c CH + i.
struct CaqrGeom { int m, n, ld, j0, nb, CH, nch; };
template <int NB> __device__ __forceinline__ int cq_row(const CaqrGeom& g, int level, int node, int s)
{
    return level == 1 ? g.j0 + node * g.CH + s : g.j0 + (s / NB) * g.CH + (s % NB);
}
template <int NB> __device__ __forceinline__ int cq_rows(const CaqrGeom& g, int level, int node)
{
    if (level == 1) { const int lo = g.j0 + node * g.CH; const int r = g.m - lo; return r < g.CH ? r : g.CH; }
    return g.nch * NB;                                                  // rows beyond the matrix read as zero
}

#define CQ_FTHREADS 1024
#define CQ_FWAVES (CQ_FTHREADS / 64)
// Factor one node's panel.  V (rows x NB, row-major, stride NB) and T (NB x NB) go to the node's workspace slots; R goes back into
// the panel's own columns (row s < nb: R[s][s..nb); rows below: zero when they are top rows of a chunk, untouched otherwise).
// The panel lives in REGISTERS: wavefront w owns NB / 16 columns, lane l their rows l, l + 64, ... (CH / 64 = 8 or 16 values per
// column).  A Householder step is: everybody reads the current reflector v_k from a double-buffered LDS vector (independent loads,
// one round trip), every wavefront updates its own columns in registers (DPP wave reduction for the dot product), the owner of
// column k+1 derives the next reflector and publishes it - ONE barrier per step and no LDS traffic for the matrix itself.  The
// first version kept the panel in LDS and paid ~3.4 us per step in dependent LDS round trips (110 us per 512 x 32 panel).
template <int NB>
__global__ void __launch_bounds__(CQ_FTHREADS) k_caqr_factor(double* __restrict__ A, CaqrGeom g, int level, double* __restrict__ Vws, double* __restrict__ Tws)
{
    extern __shared__ double sm[];
    constexpr int CH = 16384 / NB, RPL = CH / 64, CPW = NB / CQ_FWAVES, PLD = NB + 1;
    const int node = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int R = cq_rows<NB>(g, level, node), nb = g.nb;
    double* sP = sm;                                  // CH x PLD staging: panel in (coalesced rows), then V out
    double* vbuf = sP + (size_t)CH * PLD;             // 2 x CH : reflector k in vbuf[k & 1]
    double* diag = vbuf + 2 * CH;                     // NB
    double* beta = diag + NB;                         // NB
    double* Z = beta + NB;                            // NB x NB : Z[j][k] = v_j . v_k (j < k)
    double* T = Z + NB * NB;                          // NB x NB
    {   // panel in: CH x NB elements, 16 per thread, every load of a thread issued before its first LDS store (the panel was written by
        // other workgroups in the previous launch: a dependent load is a microsecond)
        constexpr int PER = CH * NB / CQ_FTHREADS;
        double v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = t + CQ_FTHREADS * u, s_ = e / NB, c = e - s_ * NB;
            const int row = s_ < R ? cq_row<NB>(g, level, node, s_) : g.m;
            v[u] = (row < g.m && c < nb) ? A[(size_t)row * g.ld + g.j0 + c] : 0.;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) { const int e = t + CQ_FTHREADS * u, s_ = e / NB, c = e - s_ * NB; sP[(size_t)s_ * PLD + c] = v[u]; }
    }
    for (int e = t; e < NB * NB; e += CQ_FTHREADS) { Z[e] = 0.; T[e] = 0.; }
    if (t < NB) { beta[t] = 0.; diag[t] = 0.; }
    __syncthreads();
    double a[CPW][RPL];
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int q = 0; q < RPL; ++q) a[c][q] = sP[(size_t)(lane + 64 * q) * PLD + wave * CPW + c];
    const int steps = nb < R ? nb : R;
    // reflector of own column c (global column index col) below row k, published into vbuf[k & 1]
    auto prep = [&](int c, int k) {
        double part = 0.;
#pragma unroll
        for (int q = 0; q < RPL; ++q) { const int i = lane + 64 * q; if (i >= k) part += a[c][q] * a[c][q]; }
        const double s = cq_wave_sum(part);
        const int lk = k & 63, qk = k >> 6;           // row k sits in lane lk, slot qk
        double akk = 0.;
#pragma unroll
        for (int q = 0; q < RPL; ++q) if (q == qk) akk = a[c][q];
        akk = __shfl(akk, lk);                        // the diagonal element to every lane (once per step)
        const double nrm = sqrt(s);
        const double tail = s - akk * akk;            // nothing below the diagonal: leave the column alone (no sign flip of an existing R)
        const double alpha = akk >= 0. ? -nrm : nrm;
        const double vn2 = 2. * (s - alpha * akk);
        const double bk = (s <= 1e-200 /* a negligible column is a zero column: QR_NEGLIGIBLE, be_qr.hip */ || vn2 == 0. || tail <= 0.) ? 0. : 2. / vn2;
        if (lane == lk && bk != 0.) {
#pragma unroll
            for (int q = 0; q < RPL; ++q) if (q == qk) a[c][q] = akk - alpha;
        }
        if (lane == 0) { diag[k] = bk != 0. ? alpha : akk; beta[k] = bk; }
        double* vb = vbuf + (size_t)(k & 1) * CH;
#pragma unroll
        for (int q = 0; q < RPL; ++q) { const int i = lane + 64 * q; vb[i] = (i >= k && bk != 0.) ? a[c][q] : 0.; }
    };
    if (steps > 0 && wave == 0) prep(0, 0);           // column 0 belongs to wavefront 0, slot 0
    __syncthreads();
    for (int k = 0; k < steps; ++k) {
        const double bk = beta[k];
        const double* vb = vbuf + (size_t)(k & 1) * CH;
        double vr[RPL];
#pragma unroll
        for (int q = 0; q < RPL; ++q) vr[q] = vb[lane + 64 * q];
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int j = wave * CPW + c;
            if (j > k) {
                if (bk != 0.) {
                    double s = 0.;
#pragma unroll
                    for (int q = 0; q < RPL; ++q) s += vr[q] * a[c][q];
                    s = cq_wave_sum(s) * bk;
#pragma unroll
                    for (int q = 0; q < RPL; ++q) a[c][q] -= s * vr[q];
                }
                if (j == k + 1 && k + 1 < steps) prep(c, k + 1);
            } else if (j < k && bk != 0.) {               // z = v_j . v_k for the compact-WY factor (v_k is zero above row k)
                double s = 0.;
#pragma unroll
                for (int q = 0; q < RPL; ++q) s += vr[q] * a[c][q];
                s = cq_wave_sum(s);
                if (lane == 0 && beta[j] != 0.) Z[j * NB + k] = s;
            }
        }
        __syncthreads();
    }
    // T (upper triangular): T[k][k] = beta_k ; T[0:k, k] = -beta_k T[0:k, 0:k] Z[0:k, k].  One wavefront, lane i keeps row i of T in
    // registers, Z[j][k] is an LDS broadcast: NB (NB - 1) / 2 independent reads instead of a dependent read per term (the loop form
    // of this took longer than the whole factorisation)
    if (wave == 0) {
        double trow[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) trow[j] = 0.;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            double sk = 0.;
#pragma unroll
            for (int j = 0; j < NB; ++j) if (j < k) sk += trow[j] * Z[j * NB + k];
            const double bk = beta[k];
            trow[k] = lane == k ? bk : (lane < k ? -bk * sk : 0.);
        }
        if (lane < NB) {
#pragma unroll
            for (int j = 0; j < NB; ++j) T[lane * NB + j] = trow[j];
        }
    }
    // registers -> staging (the panel after the factorisation: R above the diagonal, the reflectors on and below it)
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int q = 0; q < RPL; ++q) sP[(size_t)(lane + 64 * q) * PLD + wave * CPW + c] = a[c][q];
    __syncthreads();
    double* Vn = Vws + (size_t)node * CH * NB;         // level 2 has one node; its stack is at most CH rows long (nch NB <= CH)
    for (int e = t; e < R * NB; e += CQ_FTHREADS) {
        const int s = e / NB, c = e - s * NB;
        Vn[e] = (c < steps && s >= c && beta[c] != 0.) ? sP[(size_t)s * PLD + c] : 0.;
    }
    double* Tn = Tws + (size_t)node * NB * NB;
    for (int e = t; e < NB * NB; e += CQ_FTHREADS) Tn[e] = T[e];
    // R back into the panel: rows s < NB of the node (level 1: the chunk's top rows, read again by level 2; level 2: top rows of chunk 0 =
    // the panel's final R, the other chunks' top rows become zero)
    const int top = level == 1 ? (NB < R ? NB : R) : R;
    for (int s = wave; s < top; s += CQ_FWAVES) {
        const int row = cq_row<NB>(g, level, node, s);
        if (row >= g.m) continue;
        double* dst = A + (size_t)row * g.ld + g.j0;
        for (int c = lane; c < nb; c += 64) {
            double val = 0.;
            if (s < nb && c >= s) val = (c == s) ? (s < steps ? diag[s] : sP[(size_t)s * PLD + s]) : sP[(size_t)s * PLD + c];
            dst[c] = val;
        }
    }
}

#define CQ_ATHREADS 256
// Apply one node's block reflector to the columns right of the panel (+ the residual as the last column):
//   W = V^T C ; W <- T^T W ; C -= V W          C = rows of the node x [j0 + nb, n]  (column n = r)
// grid (nodes, tile groups); a workgroup stages V once and walks its 16-column tiles.  Every global load of a phase is issued before
// the first MFMA that needs one (C was written by other workgroups in the previous launch: a dependent load costs a microsecond).
template <int NB>
__global__ void __launch_bounds__(CQ_ATHREADS) k_caqr_apply(double* __restrict__ A, double* __restrict__ rv, CaqrGeom g, int level, const double* __restrict__ Vws,
                                                            const double* __restrict__ Tws)
{
    extern __shared__ double sm[];
    constexpr int CH = 16384 / NB, NBp = NB + 1, MT = NB / 16, KIT = CH / 16, GIT = CH / 64;   // k-steps / 16-row groups per wavefront
    const int node = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, i16 = lane & 15, kk = lane >> 4;
    const int R = cq_rows<NB>(g, level, node), Rr = (R + 15) & ~15;
    double* sV = sm;                                  // CH x NBp
    double* sT = sV + (size_t)CH * NBp;               // NB x NB
    double* sW = sT + NB * NB;                        // 4 partial W (one per wavefront): 4 x NB x 17, then W in slot 0
    int* srow = (int*)(sW + 4 * NB * 17);             // CH: matrix row of node row s (or -1)
    const double* Vn = Vws + (size_t)node * CH * NB;
    for (int e0 = t; e0 < CH * NB; e0 += CQ_ATHREADS * 16) {          // V in, 16 loads in flight per thread
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int e = e0 + CQ_ATHREADS * u; v[u] = (e < CH * NB && e / NB < R) ? Vn[e] : 0.; }
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int e = e0 + CQ_ATHREADS * u; if (e < CH * NB) { const int s = e / NB, c = e - s * NB; sV[(size_t)s * NBp + c] = v[u]; } }
    }
    for (int e = t; e < NB * NB; e += CQ_ATHREADS) sT[e] = Tws[(size_t)node * NB * NB + e];
    for (int s = t; s < CH; s += CQ_ATHREADS) { const int row = s < R ? cq_row<NB>(g, level, node, s) : -1; srow[s] = (row >= 0 && row < g.m) ? row : -1; }
    __syncthreads();
    const int c0 = g.j0 + g.nb;                       // first trailing column; columns c0 .. n-1 of A, then the residual as column n
    const int ncolsC = g.n - c0 + 1, ntiles = (ncolsC + 15) / 16;
    for (int tile = blockIdx.y; tile < ntiles; tile += gridDim.y) {
        const int col = c0 + tile * 16 + i16;         // this lane's C column (as B operand / D column)
        const bool col_ok = col <= g.n;
        // ---- phase A: W = V^T C, rows split over the four wavefronts (k-steps wave, wave + 4, ...)
        double cv[KIT];
#pragma unroll
        for (int u = 0; u < KIT; ++u) {
            const int s = (wave + 4 * u) * 4 + kk;
            const int row = s < Rr ? srow[s] : -1;
            cv[u] = (row >= 0 && col_ok) ? (col < g.n ? A[(size_t)row * g.ld + col] : rv[row]) : 0.;
        }
        d4 acc[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) acc[mi] = d4{0., 0., 0., 0.};
#pragma unroll
        for (int u = 0; u < KIT; ++u) {
            const int s = (wave + 4 * u) * 4 + kk;
            if ((wave + 4 * u) * 4 < Rr) {
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(sV[(size_t)s * NBp + mi * 16 + i16], cv[u], acc[mi], 0, 0, 0);
            }
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) sW[(size_t)wave * NB * 17 + (mi * 16 + kk + 4 * r) * 17 + i16] = acc[mi][r];
        __syncthreads();
        // ---- phase B: W <- T^T (sum of the four partials); NB * 16 / 256 = 1 or 2 entries per thread
        double wv[2] = {0., 0.};
        {
            int cnt = 0;
            for (int e = t; e < NB * 16; e += CQ_ATHREADS, ++cnt) {
                const int i = e / 16, j = e - i * 16;
                double w = 0.;
                for (int q = 0; q <= i; ++q) {        // (T^T W)[i][j] = sum_q T[q][i] W[q][j], T upper triangular
                    const double wq = sW[q * 17 + j] + sW[NB * 17 + q * 17 + j] + sW[2 * NB * 17 + q * 17 + j] + sW[3 * NB * 17 + q * 17 + j];
                    w += sT[q * NB + i] * wq;
                }
                wv[cnt] = w;
            }
        }
        __syncthreads();                              // every thread has read the partials before slot 0 is overwritten
        {
            int cnt = 0;
            for (int e = t; e < NB * 16; e += CQ_ATHREADS, ++cnt) { const int i = e / 16, j = e - i * 16; sW[i * 17 + j] = wv[cnt]; }
        }
        __syncthreads();
        // ---- phase C: C -= V W, 16-row groups dealt to the wavefronts (group = wave + 4 gi); old values fetched up front
        double old[GIT][4];
#pragma unroll
        for (int gi = 0; gi < GIT; ++gi) {
            const int gq = wave + 4 * gi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = gq * 16 < Rr ? srow[gq * 16 + kk + 4 * r] : -1;
                old[gi][r] = (row >= 0 && col_ok) ? (col < g.n ? A[(size_t)row * g.ld + col] : rv[row]) : 0.;
            }
        }
#pragma unroll
        for (int gi = 0; gi < GIT; ++gi) {
            const int gq = wave + 4 * gi;
            if (gq * 16 >= Rr) continue;
            d4 u = {0., 0., 0., 0.};
#pragma unroll
            for (int k0 = 0; k0 < NB; k0 += 4) u = __builtin_amdgcn_mfma_f64_16x16x4f64(sV[(size_t)(gq * 16 + i16) * NBp + k0 + kk], sW[(k0 + kk) * 17 + i16], u, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = srow[gq * 16 + kk + 4 * r];
                if (row >= 0 && col_ok) { if (col < g.n) A[(size_t)row * g.ld + col] = old[gi][r] - u[r]; else rv[row] = old[gi][r] - u[r]; }
            }
        }
        __syncthreads();
    }
}

// zero what is structurally zero in the result: below the diagonal of the top `keep` rows
__global__ void k_caqr_clean(double* __restrict__ A, int ld, int n, int keep)
{
    const int i = blockIdx.x;
    if (i >= keep) return;
    for (int c = threadIdx.x; c < i && c < n; c += blockDim.x) A[(size_t)i * ld + c] = 0.;
}

template <int NB>
static lvk_status caqr_run(lvk_context* ctx, double* d_H, int ld, int m, int n, double* d_r)
{
    const int CH = 16384 / NB;
    const int nch_max = (m + CH - 1) / CH;
    double* Vws = (double*)lvk_ctx_scratch(ctx, 7, sizeof(double) * ((size_t)nch_max * CH * NB + (size_t)CH * NB));
    double* Tws = (double*)lvk_ctx_scratch(ctx, 8, sizeof(double) * ((size_t)(nch_max + 1) * NB * NB));
    if (!Vws || !Tws) return lvk_set_error(ctx, LVK_ERR_DEVICE, "scratch allocation failed");
    double* V2 = Vws + (size_t)nch_max * CH * NB; double* T2 = Tws + (size_t)nch_max * NB * NB;
    const size_t lds_f = sizeof(double) * ((size_t)CH * (NB + 1) + 2 * (size_t)CH + 2 * NB + 2 * NB * NB + 2);
    const size_t lds_a = sizeof(double) * ((size_t)CH * (NB + 1) + NB * NB + 4 * NB * 17 + 2) + sizeof(int) * (size_t)CH;
    if (lds_f > 160 * 1024 || lds_a > 160 * 1024) return lvk_set_error(ctx, LVK_ERR_CAPACITY, "CAQR: LDS budget exceeded (%zu / %zu bytes)", lds_f, lds_a);
    LVK_LDS_OPTIN(ctx, NB == 32 ? 5 : 6, k_caqr_factor<NB>, lds_f);
    LVK_LDS_OPTIN(ctx, NB == 32 ? 7 : 4, k_caqr_apply<NB>, lds_a);
    hipStream_t s = ctx->stream;
    const int panels = (n + NB - 1) / NB;
    for (int p = 0; p < panels; ++p) {
        CaqrGeom g; g.m = m; g.n = n; g.ld = ld; g.j0 = p * NB; g.nb = (n - g.j0) < NB ? (n - g.j0) : NB; g.CH = CH;
        if (g.j0 >= m) break;
        g.nch = (m - g.j0 + CH - 1) / CH;
        const int ntiles = (n - (g.j0 + g.nb) + 1 + 15) / 16;
        const int gy1 = ntiles < 8 ? ntiles : 8, gy2 = ntiles < 32 ? ntiles : 32;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_caqr_factor<NB>), dim3(g.nch), dim3(CQ_FTHREADS), lds_f, s, d_H, g, 1, Vws, Tws);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_caqr_apply<NB>), dim3(g.nch, gy1), dim3(CQ_ATHREADS), lds_a, s, d_H, d_r, g, 1, (const double*)Vws, (const double*)Tws);
        if (g.nch > 1) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_caqr_factor<NB>), dim3(1), dim3(CQ_FTHREADS), lds_f, s, d_H, g, 2, V2, T2);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_caqr_apply<NB>), dim3(1, gy2), dim3(CQ_ATHREADS), lds_a, s, d_H, d_r, g, 2, (const double*)V2, (const double*)T2);
        }
    }
    const int keep = m < n ? m : n;
    hipLaunchKernelGGL(k_caqr_clean, dim3(keep), dim3(128), 0, s, d_H, ld, n, keep);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

lvk_status lvk_qr_compress_dev(lvk_context* ctx, double* d_H, int ldh, int rows, int cols, double* d_r, int* rows_out)
{
    if (rows > 65536) return lvk_set_error(ctx, LVK_ERR_CAPACITY, "QR compression supports up to 65536 rows (got %d)", rows);
    lvk_status st = rows <= 8192 ? caqr_run<32>(ctx, d_H, ldh, rows, cols, d_r) : caqr_run<16>(ctx, d_H, ldh, rows, cols, d_r);
    if (st != LVK_OK) return st;
    *rows_out = rows < cols ? rows : cols;
    return LVK_OK;
}
