// be_qr.hip — Householder compression of the tall measurement matrix: [H | r] (rows x cols) -> at most `cols` rows that carry the
// same H^T H and H^T r.  Replaces the SuiteSparse SPQR calls of /root/reference/src/larvio.cpp:1430-1445,2209-2229: any orthogonal
// Q that zeroes rows preserves H^T H and H^T r, which is all the update uses (isotropic noise).  Two paths:
//   structure-aware (this file): the stacked rows are block-sparse and the host knows every row group's column set -> a TSQR tree of
//     LDS-resident Householder nodes over the few columns each node touches (what SPQR's sparsity buys the reference);
//   dense (be_qr_dense.hip): communication-avoiding blocked Householder with MFMA trailing updates, for blocks without usable
//     structure and for the stage-level entry point.
#include "lvk_internal.h"
#include "be_qr.h"
#include <unordered_map>
#include "lvk_wave.h"
#include "chi2_table.inc"
#include <algorithm>
#include <iterator>
#include <utility>
#include <type_traits>

double lvk_chi2_005(int dof) { return (dof >= 1 && dof <= 99) ? k_chi2_005[dof] : 0.0; }

// ------------------------------------------------------------------------- structure-aware compression (the product path)
// The stacked MSCKF rows are block-sparse: a feature's rows touch the extrinsics/td columns and the 6-column blocks of the few clones
// that observed it (7 + 6 M columns of N), and features tracked over the same stretch of the window share those columns - which is
// what lets SPQR (larvio.cpp:1430-1445) beat a dense factorisation in the reference.  The host knows every row group's column set
// (it built the stacking map), so it plans a TSQR tree over CONSECUTIVE row groups whose column union fits one workgroup's LDS:
// each node gathers its rows restricted to the union columns into LDS (column-major, one wavefront per row on the way in), runs a
// Householder QR there (one wavefront per column on the apply step, wave reductions for the dot products, ONE barrier per step: the
// wavefront that updates column k+1 also prepares its reflector) and writes min(rows, columns) rows of R back, expanded to the dense
// column layout.  Work drops from 2 r N^2 to ~2 r c^2 (c ~ 50 of N ~ 220..450) and every level is a single launch.
// Blocks that would not shrink (rows <= columns, e.g. the two rows of each in-state feature with their scattered anchor columns)
// are passed through by a copy.
#define QS_THREADS 1024
// A column whose sum of squares (at and below its diagonal) is at most this is left alone, like a zero column (beta = 0).  When the gate has
// rejected all rows of a node but one or two, the matrix has rank 1 or 2: each further step works on the rounding residue of an exactly
// cancelled column - 1e-17, 1e-34, ... of the entries - and around the 18th column the sum of squares is a denormal number, where 2 / |v|^2
// overflows and the fast reciprocal square root below is not defined: NaN in every row handed to the update (found by the whole-program
// fuzz: "pivot 0 of 19 rows" - 19 = the columns of a pruning node).  Entries below 1e-100 carry no information next to a pixel noise of 1e-2.
#define QR_NEGLIGIBLE 1e-200
#define QS_WAVES (QS_THREADS / 64)

__device__ __forceinline__ void qs_prep_column(double* __restrict__ colk, int R, int k, int lane, double* __restrict__ diag, double* __restrict__ scal)
{   // one wavefront: reflector of column k below row k.  v overwrites the column (v_k = a_kk - alpha), R's diagonal goes to diag[k]
    double part = 0.;
    for (int i = k + lane; i < R; i += 64) { const double a = colk[i]; part += a * a; }
    const double s = wave_sum_f64(part);
    if (lane == 0) {
        const double akk = colk[k];
        const double nrm = sqrt(s);
        const double alpha = akk >= 0. ? -nrm : nrm;
        const double vn2 = 2. * (s - alpha * akk);                 // |x - alpha e_k|^2
        const double beta = (s <= QR_NEGLIGIBLE || vn2 == 0.) ? 0. : 2. / vn2;
        diag[k] = beta != 0. ? alpha : akk;
        if (beta != 0.) colk[k] = akk - alpha;
        scal[k & 1] = beta;
    }
}

__global__ void __launch_bounds__(QS_THREADS) k_qr_sparse(const double* __restrict__ Hin, int ldin, const double* __restrict__ rin,
                                                         double* __restrict__ Hout, int ldout, double* __restrict__ rout,
                                                         const QrBlock* __restrict__ blocks, const int* __restrict__ col_lists, int N)
{
    extern __shared__ double sm[];
    const QrBlock b = blocks[blockIdx.x];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (b.copy) {
        for (int i = wave; i < b.in_rows; i += QS_WAVES) {
            const double* src = Hin + (size_t)(b.in_start + i) * ldin; double* dst = Hout + (size_t)(b.out_start + i) * ldout;
            for (int j = lane; j < N; j += 64) dst[j] = src[j];
            if (lane == 0) rout[b.out_start + i] = rin[b.in_start + i];
        }
        return;
    }
    const int nc = b.ncols, R = b.in_rows, Rp = R | 1;               // odd column stride: the row-wise fill does not pile onto one bank
    double* A = sm;                                                    // (nc + 1) columns of Rp doubles; column nc is the residual
    double* diag = A + (size_t)(nc + 1) * Rp;                          // nc
    double* scal = diag + nc;                                          // 2 (beta of the current / the next step)
    int* inv = (int*)(scal + 2);                                       // N: dense column -> position in the union (or -1)
    const int* cols = col_lists + b.col_off;
    for (int j = t; j < N; j += QS_THREADS) inv[j] = -1;
    for (int i = wave; i < R; i += QS_WAVES) {
        const double* src = Hin + (size_t)(b.in_start + i) * ldin;
        for (int c = lane; c <= nc; c += 64) A[(size_t)c * Rp + i] = c < nc ? src[cols[c]] : rin[b.in_start + i];
    }
    __syncthreads();
    for (int c = t; c < nc; c += QS_THREADS) inv[cols[c]] = c;
    const int steps = nc < R - 1 ? nc : R - 1;
    if (wave == 0 && steps > 0) qs_prep_column(A, R, 0, lane, diag, scal);
    __syncthreads();
    for (int k = 0; k < steps; ++k) {
        const double beta = scal[k & 1];
        const double* v = A + (size_t)k * Rp;
        for (int j = k + 1 + wave; j <= nc; j += QS_WAVES) {
            double* col = A + (size_t)j * Rp;
            if (beta != 0.) {
                double s = 0.;
                for (int i = k + lane; i < R; i += 64) s += v[i] * col[i];
                s = wave_sum_f64(s) * beta;
                if (s != 0.) for (int i = k + lane; i < R; i += 64) col[i] -= s * v[i];
            }
            if (j == k + 1 && k + 1 < steps) {                        // always wave 0: the next reflector, in the shadow of the other columns' updates
                __builtin_amdgcn_wave_barrier();
                qs_prep_column(col, R, k + 1, lane, diag, scal);
            }
        }
        __syncthreads();
    }
    // R (upper trapezoid in the union's column order) expanded to the dense layout: one wavefront per output row, coalesced
    for (int i = wave; i < b.out_rows; i += QS_WAVES) {
        double* dst = Hout + (size_t)(b.out_start + i) * ldout;
        for (int j = lane; j < N; j += 64) {
            const int c = inv[j];
            double val = 0.;
            if (c >= i && i < R) val = (c == i) ? (i < steps ? diag[i] : A[(size_t)i * Rp + i]) : A[(size_t)c * Rp + i];
            dst[j] = val;
        }
        if (lane == 0) rout[b.out_start + i] = i < R ? A[(size_t)nc * Rp + i] : 0.;
    }
}

// The same node with its columns in REGISTERS.  What paced k_qr_sparse was not flops but its per-step chain: three LDS-latency-bound loops
// over the column (dot product, update, next norm) and a 64-lane reduction per column (~1.3-2.2 us per reflector, ~55 reflectors per
// node: 77-150 us per level).  Here a column is spread over SIXTEEN lanes (lane l holds rows l, l + 16, ...), four columns share one
// wavefront instruction, and a workgroup of sixteen wavefronts covers the 64 columns of a node: a dot product is RPL FMAs on registers
// and a four-step DPP rotation inside the 16-lane row - for four columns at once, the sum arriving in every lane that needs it - and
// the update is RPL FMAs.  Only the reflector itself travels through LDS (double-buffered, one barrier per step: the wavefront that
// owns column k+1 updates it first and publishes its reflector in the shadow of the other columns' updates).
// Round 5 - the step is ONE dependent chain (LDS read of v, dot, row sum, update, next norm, row sum, 1/sqrt, 1/x, LDS write, barrier)
// on the wavefront that owns the next column, and what it cost was instructions on that chain, not flops:
//   * the file is built without FP contraction, so `s += v * a` was a multiply AND an add - 32 dependent operations per dot product
//     and as many per norm; now explicit FMAs into four partial sums (6 dependent levels);
//   * the finished row chunks (rows < 16 (k >> 4)) were skipped through sixteen exec-masked LDS reads with six scalar instructions
//     each; the step loop is now unrolled over the chunk index, so every register index, the set of chunks still alive and the one
//     chunk that needs a per-lane row mask are compile-time facts (the reflector buffer is padded to 16 RPL rows: no `i < R` tests);
//   * the pivot element comes from one v_readlane pair instead of a masked row sum, 2 / |v|^2 = (1/|x|) / (|x| + |x_k|) from the
//     1/sqrt already at hand and ONE refined reciprocal instead of a second 1/sqrt chain;
//   * the rows come straight from global memory into the registers, sixteen loads in flight per lane (the gather through the LDS
//     image issued a column-list load and a row load per row, one after the other: 2 x 16 dependent trips to L2 before the first
//     reflector); the LDS image is only used on the way out, for the expansion to the dense column layout, and only its first
//     `ncols` rows are written.
template <int... I, class F> __device__ __forceinline__ void lvk_static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
__device__ __forceinline__ double row16_sum_f64(double v)
{   // every lane ends up with the sum over its 16-lane DPP row
    v += dpp_ror_f64(v, 8); v += dpp_ror_f64(v, 4); v += dpp_ror_f64(v, 2); v += dpp_ror_f64(v, 1);
    return v;
}
__device__ __forceinline__ double readlane_dyn_f64(double v, int lane)
{   // lane: wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rcp_refined(double x)
{   // v_rcp_f64 seed + two Newton steps, all FMAs
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0); r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0); r = __builtin_fma(r, e, r);
    return r;
}
template <int RPL, int QUADS>
__global__ void __launch_bounds__(QS_THREADS) k_qr_sparse_reg(const double* __restrict__ Hin, int ldin, const double* __restrict__ rin,
                                                             double* __restrict__ Hout, int ldout, double* __restrict__ rout,
                                                             const QrBlock* __restrict__ blocks, const int* __restrict__ col_lists, int N)
{
    static_assert(QUADS == 1, "one column quad per wavefront: nodes of at most 63 columns + the residual");
    extern __shared__ double sm[];
    const QrBlock b = blocks[blockIdx.x];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l16 = lane & 15, cq = lane >> 4;
    if (b.copy) {
        for (int i = wave; i < b.in_rows; i += QS_WAVES) {
            const double* src = Hin + (size_t)(b.in_start + i) * ldin; double* dst = Hout + (size_t)(b.out_start + i) * ldout;
            for (int j = lane; j < N; j += 64) dst[j] = src[j];
            if (lane == 0) rout[b.out_start + i] = rin[b.in_start + i];
        }
        return;
    }
    constexpr int VLD = 16 * RPL;                                      // rows of a reflector buffer (rows >= R hold zeros, as the registers do)
    const int nc = b.ncols, R = b.in_rows, Rp = R | 1;
    double* A = sm;                                                    // (nc + 1) columns of Rp doubles; column nc is the residual (used on the way out only)
    double* diag = A + (size_t)(nc + 1) * Rp;                          // nc
    double* scal = diag + nc;                                          // 2 (beta of the current / the next step)
    double* vbuf = scal + 2;                                           // 2 x VLD: the reflector of the current / the next step
    int* inv = (int*)(vbuf + 2 * (size_t)VLD);                         // N: dense column -> position in the union (or -1)
    const int* cols = col_lists + b.col_off;
    // ---- columns into registers: this wavefront holds columns 4 wave .. 4 wave + 3, this lane column j = 4 wave + cq, rows l16 + 16 rr
    const int j = 4 * wave + cq;
    const int my_col = j < nc ? cols[j] : 0, inv_col = t < nc ? cols[t] : 0;
    for (int q = t; q < N; q += QS_THREADS) inv[q] = -1;
    double a[RPL];
    {
        const double* base = j < nc ? Hin + (size_t)b.in_start * ldin + my_col : rin + b.in_start;
        const size_t stride = j < nc ? (size_t)ldin : 1;
        double x[RPL];
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) { const int i = l16 + 16 * rr; x[rr] = base[(size_t)(i < R ? i : R - 1) * stride]; }     // all in flight
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) { const int i = l16 + 16 * rr; a[rr] = (j <= nc && i < R) ? x[rr] : 0.; }
    }
    const int steps = nc < R - 1 ? nc : R - 1;
    // reflector of column k (pivot row k in the row chunk PC = k >> 4), by the 16 lanes that hold the column (the caller made sure this
    // wavefront owns it): v -> vbuf[k & 1] (chunks >= PC: all a later step reads), beta -> scal[k & 1], R's diagonal -> diag[k]
    auto prep = [&](auto pc_, int k) __attribute__((always_inline)) {
        constexpr int PC = decltype(pc_)::value;
        if constexpr (PC < RPL) {
            if (cq != (k & 3)) return;
            const int k15 = k & 15;
            double p[4] = {0., 0., 0., 0.};
            { const double x = l16 >= k15 ? a[PC] : 0.; p[0] = x * x; }
#pragma unroll
            for (int rr = PC + 1; rr < RPL; ++rr) p[(rr - PC) & 3] = __builtin_fma(a[rr], a[rr], p[(rr - PC) & 3]);
            const double s = row16_sum_f64((p[0] + p[1]) + (p[2] + p[3]));
            const double akk = readlane_dyn_f64(a[PC], 16 * (k & 3) + k15);
            // (1/sqrt and 1/x from FMA-refined hardware seeds: the IEEE sqrt and divide expand to ~80 instructions on this step's chain)
            const double y = s > QR_NEGLIGIBLE ? rsqrt_goldschmidt(s) : 0.;       // (a negligible column is a zero column: see QR_NEGLIGIBLE)
            const double nrm = s * y;
            const double alpha = akk >= 0. ? -nrm : nrm;
            // beta = 2 / |x - alpha e_k|^2 = 2 / (2 (s - alpha a_kk)) = 1 / (|x| (|x| + |a_kk|))
            const double beta = nrm != 0. ? y * rcp_refined(nrm + fabs(akk)) : 0.;
            double* vb = vbuf + (size_t)(k & 1) * VLD;
            { const double x = a[PC]; vb[l16 + 16 * PC] = l16 < k15 ? 0. : (l16 == k15 && beta != 0.) ? x - alpha : x; }
#pragma unroll
            for (int rr = PC + 1; rr < RPL; ++rr) vb[l16 + 16 * rr] = a[rr];
            if (l16 == 0) { diag[k] = beta != 0. ? alpha : akk; scal[k & 1] = beta; }
        }
    };
    // step k inside row chunk R0 = k >> 4 (rows below 16 R0 are finished); PC: the chunk of the NEXT pivot row
    auto step = [&](auto r0_, auto pc_, int k) __attribute__((always_inline)) {
        constexpr int R0 = decltype(r0_)::value;
        if (4 * wave + 3 > k) {                                        // (wave-uniform, known without a memory access) a column right of k lives here
            // beta == 0 (a column that was zero below its diagonal): s = 0, nothing moves.  Finished columns of the quad keep their values
            // (a - 0 v); the column test is a FACTOR of beta, not a select around it: with a select the compiler turns it into a branch and
            // sinks beta's LDS read into it, behind the row sum - one more LDS latency on the chain.
            const double bm = scal[k & 1] * ((j > k && j <= nc) ? 1.0 : 0.0);
            const double* vb = vbuf + (size_t)(k & 1) * VLD;
            double v[RPL];
#pragma unroll
            for (int rr = R0; rr < RPL; ++rr) v[rr] = vb[l16 + 16 * rr];
            double p[4] = {0., 0., 0., 0.};
#pragma unroll
            for (int rr = R0; rr < RPL; ++rr) p[(rr - R0) & 3] = __builtin_fma(v[rr], a[rr], p[(rr - R0) & 3]);
            const double sj = row16_sum_f64((p[0] + p[1]) + (p[2] + p[3])) * bm;
#pragma unroll
            for (int rr = R0; rr < RPL; ++rr) a[rr] = __builtin_fma(-sj, v[rr], a[rr]);
        }
        if (k + 1 < steps && ((k + 1) >> 2) == wave) prep(pc_, k + 1);  // in the shadow of the other wavefronts' updates
        __syncthreads();
    };
    if (wave == 0 && steps > 0) prep(std::integral_constant<int, 0>{}, 0);
    __syncthreads();
    if (t < nc) inv[inv_col] = t;
    lvk_static_for(std::make_integer_sequence<int, (RPL < 4 ? RPL : 4)>{}, [&](auto r0_) __attribute__((always_inline)) {      // steps <= 63: four chunks at most
        constexpr int R0 = decltype(r0_)::value;
        if (16 * R0 >= steps) return;
        const int kend = 16 * R0 + 15 < steps ? 16 * R0 + 15 : steps;
        for (int k = 16 * R0; k < kend; ++k) step(r0_, r0_, k);
        if (16 * R0 + 15 < steps) step(r0_, std::integral_constant<int, R0 + 1>{}, 16 * R0 + 15);
    });
    // ---- the first nc rows back to the LDS image (rows above the diagonal are R; the diagonal comes from diag[]), then expanded to
    // the dense layout
#pragma unroll
    for (int rr = 0; rr < RPL; ++rr) {
        const int i = l16 + 16 * rr;
        if (16 * rr < nc && j <= nc && i < R) A[(size_t)j * Rp + i] = a[rr];
    }
    __syncthreads();
    for (int i = wave; i < b.out_rows; i += QS_WAVES) {
        double* dst = Hout + (size_t)(b.out_start + i) * ldout;
        for (int jj = lane; jj < N; jj += 64) {
            const int c = inv[jj];
            double val = 0.;
            if (c >= i && i < R) val = (c == i) ? (i < steps ? diag[i] : A[(size_t)i * Rp + i]) : A[(size_t)c * Rp + i];
            dst[jj] = val;
        }
        if (lane == 0) rout[b.out_start + i] = i < R ? A[(size_t)nc * Rp + i] : 0.;
    }
}

// LDS a node needs (bytes): the planner's fit test and the launch use the same formula
size_t lvk_qr_sparse_lds_bytes(int rows, int ncols, int N)
{
    const size_t Rp = (size_t)(rows | 1);
    return sizeof(double) * ((size_t)(ncols + 1) * Rp + (size_t)ncols + 2 + 2 * (Rp > 256 ? Rp : 256)) + sizeof(int) * (size_t)N + 16;    // (2 x 256: the register kernel's padded reflector buffers)
}
template <int RPL, int QUADS>
static lvk_status launch_qr_reg(lvk_context* ctx, int slot, const double* d_Hin, int ldin, const double* d_rin, double* d_Hout, int ldout, double* d_rout,
                                const QrBlock* d_blocks, int n_blocks, const int* d_cols, int N, size_t max_lds)
{
    if (max_lds > 64 * 1024) LVK_LDS_OPTIN(ctx, slot, (k_qr_sparse_reg<RPL, QUADS>), max_lds);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_qr_sparse_reg<RPL, QUADS>), dim3(n_blocks), dim3(QS_THREADS), max_lds, ctx->stream, d_Hin, ldin, d_rin, d_Hout, ldout, d_rout, d_blocks, d_cols, N);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}
lvk_status lvk_qr_sparse_level(lvk_context* ctx, const double* d_Hin, int ldin, const double* d_rin, double* d_Hout, int ldout, double* d_rout,
                               const QrBlock* d_blocks, int n_blocks, const int* d_cols, int N, size_t max_lds, int max_rows, int max_cols)
{
    if (n_blocks <= 0) return LVK_OK;
    if (max_lds > 160 * 1024) return lvk_set_error(ctx, LVK_ERR_CAPACITY, "QR node needs %zu bytes of LDS", max_lds);
    // register-resident nodes where every node of the level fits one of the compiled shapes (rows <= 16 RPL, columns + 1 <= 64 QUADS)
    const int rpl = (max_rows + 15) / 16, quads = (max_cols + 1 + 63) / 64;
#define QR_REG(R_, Q_, slot_) return launch_qr_reg<R_, Q_>(ctx, slot_, d_Hin, ldin, d_rin, d_Hout, ldout, d_rout, d_blocks, n_blocks, d_cols, N, max_lds)
    if (max_rows > 0 && quads == 1) { if (rpl <= 8) QR_REG(8, 1, 12); if (rpl <= 16) QR_REG(16, 1, 13); }        // (every node the planner makes; anything else: the LDS kernel below)
#undef QR_REG
    if (max_lds > 64 * 1024) LVK_LDS_OPTIN(ctx, 11, k_qr_sparse, max_lds);
    hipLaunchKernelGGL(k_qr_sparse, dim3(n_blocks), dim3(QS_THREADS), max_lds, ctx->stream, d_Hin, ldin, d_rin, d_Hout, ldout, d_rout, d_blocks, d_cols, N);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

// Greedy, level by level: consecutive groups are merged into one node while the node still fits the workgroup's LDS; a node that
// already shrinks (rows > columns) does not take in a group that brings more new columns than rows (the two rows of an in-state
// feature with its own anchor block).  A node with rows <= columns is passed through.  A level is kept only if it removes at
// least a fifth of the rows.  Everything here is known on the host before any kernel runs: no counts come back from the device.
// Column sets are BIT SETS over the state's columns while the plan is drawn up (union = OR, size = popcount; a group's set is built
// once per distinct column list - features that share their clone set share the list): the sorted-list merges of rounds 2-3 cost the
// filter's thread 250-400 us per update at configs[4] depth (1900 groups), this ~40 (same nodes, same order).
void lvk_qr_sparse_plan(const std::vector<RowGroup>& groups, int N, std::vector<QrPlanLevel>& levels, int* final_rows, std::vector<RowGroup>* final_groups)
{
    const size_t LDS_CAP = (size_t)152 * 1024;
    // the shapes k_qr_sparse_reg holds in registers: 16 rows per lane x 16 lanes, one column "quad" (63 columns + the residual).  Nodes of
    // up to 127 columns (two quads) were planned until the middle of round 4: their reflector chain is twice as long, and at configs[4]
    // a level with such a node took 186 us against 68 - narrower nodes and, now and then, one more level are faster (3653-3670 ->
    // 3744-3763 frames/s for caps of 48 / 64 / 80, profiles/r4_an_qr_cols_cap_ab.txt)
    const int ROWS_CAP = 256, COLS_CAP = 63;
    levels.clear();
    const size_t W = ((size_t)N + 63) / 64;               // words per set
    struct G { int start, rows; const std::vector<int>* cols; size_t bits; int n; ColList keep; };     // bits: offset of the set in `pool`; keep: owner of a list made here
    std::vector<uint64_t> pool; pool.reserve(W * 64);
    const std::vector<int>* last_list = nullptr; size_t last_off = 0;                                // consecutive groups often carry the same list
    std::unordered_map<const std::vector<int>*, size_t> seen; seen.reserve(64);                      // distinct lists of the input (a handful in the filter)
    auto set_of = [&](const std::vector<int>* c) -> size_t {
        if (c == last_list) return last_off;
        auto it = seen.find(c);
        if (it == seen.end()) {
            const size_t off = pool.size(); pool.resize(off + W, 0);
            for (int x : *c) pool[off + ((size_t)x >> 6)] |= 1ull << (x & 63);
            it = seen.emplace(c, off).first;
        }
        last_list = c; last_off = it->second;
        return last_off;
    };
    std::vector<G> cur; cur.reserve(groups.size());
    int total = 0;
    for (const RowGroup& g : groups) { G x; x.start = total; x.rows = g.rows; x.cols = g.cols.get(); x.bits = set_of(x.cols); x.n = (int)x.cols->size(); total += g.rows; cur.push_back(std::move(x)); }
    std::vector<uint64_t> uni(W), merged(W);
    std::vector<int> list;
    for (int lvl = 0; lvl < 8 && cur.size() > 0; ++lvl) {
        QrPlanLevel L; std::vector<G> next; next.reserve(cur.size() / 4 + 4);
        size_t i = 0; int out_row = 0; bool any = false;
        while (i < cur.size()) {
            for (size_t w = 0; w < W; ++w) uni[w] = pool[cur[i].bits + w];
            int n_uni = cur[i].n, rows = cur[i].rows; size_t j = i + 1;
            const std::vector<int>* same = cur[i].cols;        // while every merged group carries this very list the union is unchanged
            while (j < cur.size()) {
                const int r2 = rows + cur[j].rows;
                if (cur[j].cols == same) {
                    if (lvk_qr_sparse_lds_bytes(r2, n_uni, N) > LDS_CAP || r2 > ROWS_CAP) break;
                    rows = r2; ++j; continue;
                }
                int n_m = 0;
                for (size_t w = 0; w < W; ++w) { merged[w] = uni[w] | pool[cur[j].bits + w]; n_m += __builtin_popcountll(merged[w]); }
                if (lvk_qr_sparse_lds_bytes(r2, n_m, N) > LDS_CAP || r2 > ROWS_CAP || n_m > COLS_CAP) break;
                if (rows > n_uni && n_m - n_uni > cur[j].rows) break;
                if (n_m != n_uni) same = nullptr;
                uni.swap(merged); n_uni = n_m; rows = r2; ++j;
            }
            QrBlock b; memset(&b, 0, sizeof b);
            b.in_start = cur[i].start; b.in_rows = rows; b.out_start = out_row;
            if (rows > n_uni && n_uni > 0 && lvk_qr_sparse_lds_bytes(rows, n_uni, N) <= LDS_CAP) {
                b.copy = 0; b.ncols = n_uni; b.out_rows = b.ncols; b.col_off = (int)L.cols.size();
                list.clear();
                for (size_t w = 0; w < W; ++w) for (uint64_t m = uni[w]; m; m &= m - 1) list.push_back((int)(64 * w) + __builtin_ctzll(m));
                L.cols.insert(L.cols.end(), list.begin(), list.end());
                L.lds = std::max(L.lds, lvk_qr_sparse_lds_bytes(rows, b.ncols, N));
                L.max_rows = std::max(L.max_rows, rows); L.max_cols = std::max(L.max_cols, b.ncols);
                G x; x.start = out_row; x.rows = b.out_rows; x.keep = std::make_shared<const std::vector<int>>(list); x.cols = x.keep.get(); x.n = n_uni;
                x.bits = pool.size(); pool.insert(pool.end(), uni.begin(), uni.end());
                next.push_back(std::move(x));
                any = true;
            } else {
                b.copy = 1; b.ncols = 0; b.out_rows = rows; b.col_off = 0;
                for (size_t g = i; g < j; ++g) { next.push_back(cur[g]); next.back().start = out_row + (cur[g].start - cur[i].start); }
            }
            L.blocks.push_back(b);
            out_row += b.out_rows; i = j;
        }
        L.in_rows = total; L.out_rows = out_row;
        if (!any || out_row * 5 > total * 4) break;
        levels.push_back(std::move(L));
        cur.swap(next); total = out_row;
    }
    *final_rows = total;
    if (final_groups) {
        // (lists that came in with the caller's groups are shared again by pointer identity: the caller's ColLists stay the owners)
        final_groups->clear(); final_groups->reserve(cur.size());
        size_t gi = 0;
        for (const G& x : cur) {
            RowGroup r; r.start = x.start; r.rows = x.rows;
            if (x.keep) r.cols = x.keep;
            else { while (gi < groups.size() && groups[gi].cols.get() != x.cols) ++gi; r.cols = gi < groups.size() ? groups[gi].cols : std::make_shared<const std::vector<int>>(*x.cols); }
            final_groups->push_back(std::move(r));
        }
    }
}

lvk_status lvk_qr_compress_dev(lvk_context* ctx, double* d_H, int ldh, int rows, int cols, double* d_r, int* rows_out);   // be_qr_dense.hip

// stage-level C ABI: compress a device matrix in place (rows x cols, leading dimension ld), d_r likewise
extern "C" lvk_status lvk_ekf_compress_qr(lvk_context* ctx, double* d_H, int ld, int rows, int cols, double* d_r, int* rows_out)
{
    if (!ctx || !d_H || !d_r || !rows_out || rows < 0 || cols <= 0 || ld < cols) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_ekf_compress_qr: bad argument");
    if (rows <= cols) { *rows_out = rows; return LVK_OK; }
    return lvk_qr_compress_dev(ctx, d_H, ld, rows, cols, d_r, rows_out);
}

// ---- stage-level C ABI of the structure-aware compression
static void groups_from_arrays(int n_groups, const int* rows, const int* col_off, const int* cols, std::vector<RowGroup>& g)
{
    g.resize((size_t)n_groups);
    for (int i = 0; i < n_groups; ++i) { g[i].start = 0; g[i].rows = rows[i]; g[i].cols = std::make_shared<const std::vector<int>>(cols + col_off[i], cols + col_off[i + 1]); }
}
// host-only: the TSQR tree for `n_groups` consecutive row groups (group i: h_rows[i] rows, columns h_cols[h_col_off[i] .. h_col_off[i+1]),
// ascending).  Blocks come back as 8 ints each (QrBlock), level after level; h_level_blocks[l] / h_level_cols[l] = blocks / column-list
// entries of level l.  Returns the number of levels (0: not worth compressing), or -1 on a bad argument / too small an output buffer.
extern "C" int lvk_ekf_qr_plan(int N, int n_groups, const int* h_rows, const int* h_col_off, const int* h_cols, int* h_blocks, int cap_blocks,
                               int* h_block_cols, int cap_cols, int* h_level_blocks, int* h_level_cols, int cap_levels, int* final_rows)
{
    if (N <= 0 || n_groups < 0 || !h_rows || !h_col_off || !h_cols || !final_rows) return -1;
    std::vector<RowGroup> g; groups_from_arrays(n_groups, h_rows, h_col_off, h_cols, g);
    std::vector<QrPlanLevel> levels;
    lvk_qr_sparse_plan(g, N, levels, final_rows);
    int nb = 0, ncl = 0;
    if ((int)levels.size() > cap_levels) return -1;
    for (size_t l = 0; l < levels.size(); ++l) {
        if (nb + (int)levels[l].blocks.size() > cap_blocks || ncl + (int)levels[l].cols.size() > cap_cols) return -1;
        memcpy(h_blocks + 8 * nb, levels[l].blocks.data(), sizeof(QrBlock) * levels[l].blocks.size());
        if (!levels[l].cols.empty()) memcpy(h_block_cols + ncl, levels[l].cols.data(), sizeof(int) * levels[l].cols.size());
        h_level_blocks[l] = (int)levels[l].blocks.size(); h_level_cols[l] = (int)levels[l].cols.size();
        nb += (int)levels[l].blocks.size(); ncl += (int)levels[l].cols.size();
    }
    return (int)levels.size();
}
// device: [H | r] (rows x cols, leading dimension ld) whose rows come in `n_groups` consecutive groups with known column sets ->
// rows_out rows that carry the same H^T H and H^T r, in place at the top of d_H / d_r (the levels ping-pong through scratch memory).
// rows_out < rows only if the plan found something to remove; follow with lvk_ekf_compress_qr for the dense finish when
// rows_out > cols.
extern "C" lvk_status lvk_ekf_compress_qr_groups(lvk_context* ctx, double* d_H, int ld, int rows, int cols, double* d_r, int n_groups,
                                                 const int* h_rows, const int* h_col_off, const int* h_cols, int* rows_out)
{
    if (!ctx || !d_H || !d_r || !rows_out || rows < 0 || cols <= 0 || ld < cols || n_groups < 0 || !h_rows || !h_col_off || !h_cols)
        return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_ekf_compress_qr_groups: bad argument");
    std::vector<RowGroup> g; groups_from_arrays(n_groups, h_rows, h_col_off, h_cols, g);
    int tot = 0; for (auto& x : g) tot += x.rows;
    if (tot != rows) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_ekf_compress_qr_groups: the groups hold %d rows, the matrix %d", tot, rows);
    for (auto& x : g) for (size_t k = 0; k < x.cols->size(); ++k)
        if ((*x.cols)[k] < 0 || (*x.cols)[k] >= cols || (k > 0 && (*x.cols)[k] <= (*x.cols)[k - 1])) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_ekf_compress_qr_groups: column lists must be ascending and < cols");
    std::vector<QrPlanLevel> levels; int m2 = rows;
    lvk_qr_sparse_plan(g, cols, levels, &m2);
    *rows_out = rows;
    if (levels.empty()) return LVK_OK;
    double* Hb = (double*)lvk_ctx_scratch(ctx, 7, sizeof(double) * (size_t)rows * ld);
    double* rb = (double*)lvk_ctx_scratch(ctx, 8, sizeof(double) * (size_t)rows);
    size_t nb = 0, ncl = 0; for (auto& L : levels) { nb += L.blocks.size(); ncl += L.cols.size() + 1; }
    char* meta = (char*)lvk_ctx_scratch(ctx, 9, sizeof(QrBlock) * nb + sizeof(int) * ncl);
    if (!Hb || !rb || !meta) return lvk_set_error(ctx, LVK_ERR_DEVICE, "scratch allocation failed");
    QrBlock* d_blocks = (QrBlock*)meta; int* d_cols = (int*)(meta + sizeof(QrBlock) * nb);
    std::vector<QrBlock> hb; std::vector<int> hc;
    for (auto& L : levels) { hb.insert(hb.end(), L.blocks.begin(), L.blocks.end()); hc.insert(hc.end(), L.cols.begin(), L.cols.end()); hc.push_back(0); }
    LVK_HIP(ctx, hipMemcpyAsync(d_blocks, hb.data(), sizeof(QrBlock) * nb, hipMemcpyHostToDevice, ctx->stream));
    LVK_HIP(ctx, hipMemcpyAsync(d_cols, hc.data(), sizeof(int) * ncl, hipMemcpyHostToDevice, ctx->stream));
    LVK_HIP(ctx, hipStreamSynchronize(ctx->stream));                  // hb / hc are stack-local
    double* H = d_H; double* r = d_r; size_t ob = 0, oc = 0;
    for (auto& L : levels) {
        double* Ho = (H == d_H) ? Hb : d_H; double* ro = (r == d_r) ? rb : d_r;
        lvk_status st = lvk_qr_sparse_level(ctx, H, ld, r, Ho, ld, ro, d_blocks + ob, (int)L.blocks.size(), d_cols + oc, cols, L.lds, L.max_rows, L.max_cols);
        if (st != LVK_OK) return st;
        ob += L.blocks.size(); oc += L.cols.size() + 1;
        H = Ho; r = ro;
    }
    if (H != d_H) {
        LVK_HIP(ctx, hipMemcpy2DAsync(d_H, sizeof(double) * ld, H, sizeof(double) * ld, sizeof(double) * cols, m2, hipMemcpyDeviceToDevice, ctx->stream));
        LVK_HIP(ctx, hipMemcpyAsync(d_r, r, sizeof(double) * m2, hipMemcpyDeviceToDevice, ctx->stream));
    }
    *rows_out = m2;
    return LVK_OK;
}
