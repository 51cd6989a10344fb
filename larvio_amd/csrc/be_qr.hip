// be_qr.hip — Householder compression of the tall measurement matrix: [H | r] (rows x cols) -> top `cols` rows of
// Q^T [H | r], in place.  Replaces the SuiteSparse SPQR calls of /root/reference/src/larvio.cpp:1430-1445,2209-2229
// (natural ordering, Q^T applied densely): any orthogonal Q that zeroes the rows below `cols` preserves H^T H and
// H^T r, which is all the update uses (isotropic noise).  Structure: TSQR — row blocks of QR_BLOCK_ROWS are reduced
// independently (one workgroup each, reflectors kept in LDS, columns spread over the wavefronts, per-column dot
// products by wave reduction), the surviving triangles are stacked and reduced again until one block remains.
// At the north-star size (a few hundred rows, ~200 columns) that is a single workgroup; at 18,000 rows (config 5)
// the first level runs ~36 workgroups in parallel.
#include "lvk_internal.h"
#include "be_qr.h"
#include "chi2_table.inc"
#include <algorithm>
#include <iterator>

double lvk_chi2_005(int dof) { return (dof >= 1 && dof <= 99) ? k_chi2_005[dof] : 0.0; }

#define QR_THREADS 1024
#define QR_BLOCK_ROWS 1024

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one workgroup reduces rows [row0, row0+nrows) of H (ld) and r; result: its first min(nrows, cols) rows hold R.
__global__ void __launch_bounds__(QR_THREADS) k_qr_block(double* __restrict__ H, int ld, int cols, double* __restrict__ r,
                                                        int total_rows, int block_rows)
{
    __shared__ double v[QR_BLOCK_ROWS];
    __shared__ double red[QR_THREADS / 64];
    __shared__ double scal[2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, n_waves = QR_THREADS / 64;
    const int row0 = blockIdx.x * block_rows;
    const int nrows = min(block_rows, total_rows - row0);
    if (nrows <= 0) return;
    double* A = H + (size_t)row0 * ld;
    double* rb = r + row0;
    const int steps = min(cols, nrows - 1);
    for (int k = 0; k < steps; ++k) {
        // ||A[k:, k]||^2
        double part = 0.;
        for (int i = k + t; i < nrows; i += QR_THREADS) { double a = A[(size_t)i * ld + k]; part += a * a; }
        part = wave_sum_f64(part);
        if (lane == 0) red[wave] = part;
        __syncthreads();
        if (t == 0) {
            double s = 0.; for (int w = 0; w < n_waves; ++w) s += red[w];
            const double akk = A[(size_t)k * ld + k];
            const double nrm = sqrt(s);
            const double alpha = akk >= 0. ? -nrm : nrm;
            // v = x - alpha e_k ; |v|^2 = s - 2 alpha akk + alpha^2 = 2 (s - alpha akk)
            const double vn2 = 2. * (s - alpha * akk);
            scal[0] = alpha; scal[1] = (nrm == 0. || vn2 == 0.) ? 0. : 2. / vn2;
        }
        __syncthreads();
        const double alpha = scal[0], beta = scal[1];
        if (beta != 0.) {
            for (int i = k + t; i < nrows; i += QR_THREADS) v[i] = A[(size_t)i * ld + k] - (i == k ? alpha : 0.);
            __syncthreads();
            // apply (I - beta v v^T) to columns k+1..cols-1 and to r: one wavefront per column
            for (int c = k + 1 + wave; c <= cols; c += n_waves) {
                double s = 0.;
                if (c < cols) { for (int i = k + lane; i < nrows; i += 64) s += v[i] * A[(size_t)i * ld + c]; }
                else { for (int i = k + lane; i < nrows; i += 64) s += v[i] * rb[i]; }
                s = wave_sum_f64(s) * beta;
                if (s != 0.) {
                    if (c < cols) { for (int i = k + lane; i < nrows; i += 64) A[(size_t)i * ld + c] -= s * v[i]; }
                    else { for (int i = k + lane; i < nrows; i += 64) rb[i] -= s * v[i]; }
                }
            }
            // column k itself: (alpha, 0, ..., 0)
            for (int i = k + t; i < nrows; i += QR_THREADS) A[(size_t)i * ld + k] = (i == k) ? alpha : 0.;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------- structure-aware compression (the product path)
// The stacked MSCKF rows are block-sparse: a feature's rows touch the extrinsics/td columns and the 6-column blocks of the few clones
// that observed it (7 + 6 M columns of N), and features tracked over the same stretch of the window share those columns - which is
// what lets SPQR (larvio.cpp:1430-1445) beat a dense factorisation in the reference.  The host knows every row group's column set
// (it built the stacking map), so it merges CONSECUTIVE row groups into nodes whose column union stays small (<= 127 columns, any
// number of rows) and each node is reduced to `union` rows:
//     G = [A | r]^T [A | r]   (A = the node's rows restricted to its columns)   on the FP64 matrix cores, one workgroup per 256-row
//                             chunk, partial Grams summed in a fixed order (no atomics: results are reproducible bit for bit)
//     G = L L^T               in LDS, one workgroup per node; R = L^T, the last row of L is Q^T r
// The update downstream consumes the measurement only through H^T H and H^T r (information form: P+ = (P^-1 + H^T H / s^2)^-1,
// dx = P+ H^T r / s^2), so ANY (R, rho) with R^T R = H^T H and R^T rho = H^T r is equivalent to the Householder factor - and forming
// the Gram perturbs H^T H by O(eps |H|^2), the same order as a backward-stable QR does.  Directions whose pivot falls below the
// rounding noise of G (exactly-zero columns such as td when it is not estimated, or information below eps |G|) are dropped as zero
// rows.  Work: one streaming pass over the rows (2 r c^2 flops on MFMA, c ~ 50 of N ~ 220..450) + c^3 / 3 in LDS; ONE level reduces
// 17,000 rows at configs[4].  (Two LDS-resident Householder versions of the nodes were measured first: 130-250 us per 430 x 44 node -
// one barrier-separated level-2 step per column, bound by LDS round trips - against ~10 + 20 us here.)
// Blocks that would not shrink (rows <= columns, e.g. the two rows of each in-state feature with their scattered anchor columns)
// are passed through by a copy.  Unions wider than 127 columns (tracks longer than ~20 clones) fall back to the dense TSQR above.
#define QG_THREADS 256
#define QG_MAX_NC 127

// leading dimension of a staged chunk: 16 per column tile + 16 so that the four k-groups of an MFMA operand read fall on two
// disjoint bank halves (2 LDS passes per 512-byte read instead of 4); rows per chunk so that the chunk fits 128 KB
__host__ __device__ static inline int qg_lda(int ncols) { return ((ncols + 1 + 15) / 16) * 16 + 16; }
static inline int qg_chunk_rows(int ncols) { int r = 16384 / qg_lda(ncols); r &= ~3; return r > 256 ? 256 : r; }

// one workgroup per chunk: partial Gram of rows [row_start, row_start + rows) of the level's input, restricted to the node's columns
// (+ the residual as column nc), as upper-triangular 16x16 tiles (ti <= tj)
__global__ void __launch_bounds__(QG_THREADS) k_gram_partial(const double* __restrict__ Hin, int ldin, const double* __restrict__ rin,
                                                            const QrBlock* __restrict__ blocks, const QrChunk* __restrict__ chunks,
                                                            const int* __restrict__ col_lists, double* __restrict__ part)
{
    extern __shared__ double sA[];
    const QrChunk ch = chunks[blockIdx.x];
    const QrBlock b = blocks[ch.block];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i16 = lane & 15, kk = lane >> 4;
    const int nc = b.ncols, T = (nc + 1 + 15) / 16, lda = qg_lda(nc);
    const int rp = (ch.rows + 3) & ~3;
    const int* cols = col_lists + b.col_off;
    // stage the chunk (row-major, one wavefront per row, gathered columns: mostly runs of 6-7 consecutive doubles)
    for (int i = wave; i < rp; i += QG_THREADS / 64) {
        const bool ok = i < ch.rows;
        const double* src = Hin + (size_t)(ch.row_start + (ok ? i : 0)) * ldin;
        for (int c = lane; c < 16 * T; c += 64) sA[(size_t)i * lda + c] = !ok ? 0. : c < nc ? src[cols[c]] : c == nc ? rin[ch.row_start + i] : 0.;
    }
    __syncthreads();
    const int npairs = T * (T + 1) / 2;
    double* out = part + ((size_t)ch.part_first) * 256;
    int ti = 0, tj = 0;                                                 // pair index p -> (ti <= tj), row-major over the upper triangle
    for (int p = 0; p < npairs; ++p) {
        if ((p & 3) == wave) {
            typedef double d4 __attribute__((ext_vector_type(4)));
            d4 acc = {0., 0., 0., 0.};
            const double* pa = sA + 16 * ti + i16; const double* pb = sA + 16 * tj + i16;
            for (int k0 = 0; k0 < rp; k0 += 4) {
                const double av = pa[(size_t)(k0 + kk) * lda], bv = pb[(size_t)(k0 + kk) * lda];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)p * 256 + (kk + 4 * r) * 16 + i16] = acc[r];
        }
        if (++tj == T) { ++ti; tj = ti; }
    }
}

// one workgroup per node: sum the chunk partials (fixed order), Cholesky of the (nc + 1) x (nc + 1) Gram in LDS, write nc rows of
// R = L^T and Q^T r = the last row of L expanded to the dense column layout; pass-through blocks are copied
__global__ void __launch_bounds__(QG_THREADS) k_gram_chol(const double* __restrict__ Hin, int ldin, const double* __restrict__ rin,
                                                         double* __restrict__ Hout, int ldout, double* __restrict__ rout,
                                                         const QrBlock* __restrict__ blocks, const int* __restrict__ col_lists,
                                                         const double* __restrict__ part, int N)
{
    extern __shared__ double sG[];
    const QrBlock b = blocks[blockIdx.x];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (b.copy) {
        for (int i = wave; i < b.in_rows; i += QG_THREADS / 64) {
            const double* src = Hin + (size_t)(b.in_start + i) * ldin; double* dst = Hout + (size_t)(b.out_start + i) * ldout;
            for (int j = lane; j < N; j += 64) dst[j] = src[j];
            if (lane == 0) rout[b.out_start + i] = rin[b.in_start + i];
        }
        return;
    }
    const int nc = b.ncols, n = nc + 1, T = (n + 15) / 16, ldg = 16 * T + 1, npairs = T * (T + 1) / 2;
    double* G = sG;                                                     // (16 T) x ldg, lower triangle becomes L
    double* dmax = G + (size_t)16 * T * ldg;                            // [0] tolerance
    int* inv = (int*)(dmax + 2);                                        // N: dense column -> position in the union (or -1)
    const int* cols = col_lists + b.col_off;
    for (int j = t; j < N; j += QG_THREADS) inv[j] = -1;
    {
        int ti = 0, tj = 0;
        for (int p = 0; p < npairs; ++p) {
            double s = 0.;
            const double* src = part + ((size_t)b.part_first + p) * 256 + t;
            for (int c = 0; c < b.n_chunks; ++c) s += src[(size_t)c * npairs * 256];
            const int i = 16 * ti + (t >> 4), j = 16 * tj + (t & 15);
            G[(size_t)j * ldg + i] = s;                                 // G[j][i], j >= i side (lower) ...
            if (ti == tj) G[(size_t)i * ldg + j] = s;                   // ... diagonal tiles hold both halves
            if (++tj == T) { ++ti; tj = ti; }
        }
    }
    __syncthreads();
    for (int c = t; c < nc; c += QG_THREADS) inv[cols[c]] = c;
    if (t == 0) {
        double m = 0.; for (int i = 0; i < nc; ++i) m = fmax(m, G[(size_t)i * ldg + i]);
        dmax[0] = m * (double)n * 1.8e-15;                              // ~ 8 n eps |G|: below this a pivot is rounding noise of the Gram
    }
    __syncthreads();
    const double tol = dmax[0];
    // right-looking Cholesky on the lower triangle, columns 0..nc-1 (row nc = the residual's row rides along)
    for (int j = 0; j < nc; ++j) {
        __syncthreads();                                                // the trailing update of column j-1 is complete
        const double d = G[(size_t)j * ldg + j];
        const bool ok = d > tol;
        const double s = ok ? 1.0 / sqrt(d) : 0.0;
        __syncthreads();                                                // everyone has read the pivot before it is overwritten
        for (int i = j + t; i < n; i += QG_THREADS) G[(size_t)i * ldg + j] = ok ? (i == j ? sqrt(d) : G[(size_t)i * ldg + j] * s) : 0.0;
        __syncthreads();
        if (ok) {
            const int m = n - j - 1;                                    // trailing rows j+1..n-1; element (a, c), c <= a
            for (int e = t; e < m * m; e += QG_THREADS) {
                const int a = j + 1 + e / m, c = j + 1 + e % m;
                if (c <= a) G[(size_t)a * ldg + c] -= G[(size_t)a * ldg + j] * G[(size_t)c * ldg + j];
            }
        }
    }
    __syncthreads();
    for (int i = wave; i < b.out_rows; i += QG_THREADS / 64) {          // row i of R = column i of L, expanded
        double* dst = Hout + (size_t)(b.out_start + i) * ldout;
        for (int j = lane; j < N; j += 64) { const int c = inv[j]; dst[j] = (c >= i && i < nc) ? G[(size_t)c * ldg + i] : 0.; }
        if (lane == 0) rout[b.out_start + i] = i < nc ? G[(size_t)nc * ldg + i] : 0.;
    }
}

lvk_status lvk_qr_sparse_level(lvk_context* ctx, const double* d_Hin, int ldin, const double* d_rin, double* d_Hout, int ldout, double* d_rout,
                               const QrBlock* d_blocks, int n_blocks, const QrChunk* d_chunks, int n_chunks, const int* d_cols, int N, int max_nc,
                               size_t part_doubles)
{
    if (n_blocks <= 0) return LVK_OK;
    double* part = nullptr;
    if (n_chunks > 0) {
        part = (double*)lvk_ctx_scratch(ctx, 11, sizeof(double) * part_doubles);
        if (!part) return lvk_set_error(ctx, LVK_ERR_DEVICE, "scratch allocation failed");
        const size_t lds1 = sizeof(double) * 16384;                   // every chunk is sized to fit 128 KB (qg_chunk_rows)
        LVK_LDS_OPTIN(ctx, 3, k_gram_partial, lds1);
        hipLaunchKernelGGL(k_gram_partial, dim3(n_chunks), dim3(QG_THREADS), lds1, ctx->stream, d_Hin, ldin, d_rin, d_blocks, d_chunks, d_cols, part);
    }
    const int T = (max_nc + 1 + 15) / 16;
    const size_t lds2 = sizeof(double) * ((size_t)16 * T * (16 * T + 1) + 2) + sizeof(int) * (size_t)N + 16;
    if (lds2 > 160 * 1024) return lvk_set_error(ctx, LVK_ERR_CAPACITY, "Gram node needs %zu bytes of LDS", lds2);
    if (lds2 > 64 * 1024) LVK_LDS_OPTIN(ctx, 4, k_gram_chol, lds2);
    hipLaunchKernelGGL(k_gram_chol, dim3(n_blocks), dim3(QG_THREADS), lds2, ctx->stream, d_Hin, ldin, d_rin, d_Hout, ldout, d_rout, d_blocks, d_cols, (const double*)part, N);
    LVK_LAUNCH_CHECK(ctx);
    return LVK_OK;
}

static void merge_cols(const std::vector<int>& a, const std::vector<int>& b, std::vector<int>& out)
{
    out.clear(); out.reserve(a.size() + b.size());
    std::set_union(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(out));
}
// Greedy, level by level: consecutive groups are merged into one node while the column union stays within QG_MAX_NC; a node that
// already shrinks (rows > columns) does not take in a group that brings more new columns than rows (the two rows of an in-state
// feature with its own anchor block).  A node with rows <= columns is passed through.  A level is kept only if it removes at
// least a fifth of the rows.  Everything here is known on the host before any kernel runs: no counts come back from the device.
void lvk_qr_sparse_plan(std::vector<RowGroup> cur, int N, std::vector<QrPlanLevel>& levels, int* final_rows, std::vector<RowGroup>* final_groups)
{
    levels.clear();
    int total = 0; for (auto& g : cur) { g.start = total; total += g.rows; }
    std::vector<int> uni, merged;
    for (int lvl = 0; lvl < 6 && cur.size() > 0; ++lvl) {
        QrPlanLevel L; std::vector<RowGroup> next;
        size_t i = 0; int out_row = 0; bool any = false;
        while (i < cur.size()) {
            uni = cur[i].cols; int rows = cur[i].rows; size_t j = i + 1;
            while (j < cur.size()) {
                merge_cols(uni, cur[j].cols, merged);
                if ((int)merged.size() > QG_MAX_NC) break;
                if (rows > (int)uni.size() && (int)(merged.size() - uni.size()) > cur[j].rows) break;
                uni.swap(merged); rows += cur[j].rows; ++j;
            }
            QrBlock b; memset(&b, 0, sizeof b);
            b.in_start = cur[i].start; b.in_rows = rows; b.out_start = out_row;
            if (rows > (int)uni.size() && (int)uni.size() <= QG_MAX_NC && !uni.empty()) {
                b.copy = 0; b.ncols = (int)uni.size(); b.out_rows = b.ncols; b.col_off = (int)L.cols.size();
                L.cols.insert(L.cols.end(), uni.begin(), uni.end());
                const int cr = qg_chunk_rows(b.ncols), T = (b.ncols + 1 + 15) / 16, npairs = T * (T + 1) / 2;
                b.chunk_first = (int)L.chunks.size(); b.n_chunks = (rows + cr - 1) / cr; b.part_first = (int)L.part_tiles;
                for (int c = 0; c < b.n_chunks; ++c) {
                    QrChunk ch; ch.block = (int)L.blocks.size(); ch.row_start = b.in_start + c * cr; ch.rows = std::min(cr, rows - c * cr);
                    ch.part_first = (int)L.part_tiles + c * npairs;
                    L.chunks.push_back(ch);
                }
                L.part_tiles += (size_t)b.n_chunks * npairs;
                L.max_nc = std::max(L.max_nc, b.ncols);
                next.emplace_back(); next.back().start = out_row; next.back().rows = b.out_rows; next.back().cols = uni;
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
    if (final_groups) final_groups->swap(cur);
}

// move the first `keep` rows of every block to the front (block b -> rows [b*keep, ...))
__global__ void k_qr_pack(const double* __restrict__ H, int ld, int cols, const double* __restrict__ r, int total_rows, int block_rows, int keep,
                          double* __restrict__ Hout, double* __restrict__ rout)
{
    const int b = blockIdx.y, i = blockIdx.x;                  // row i of block b
    const int nrows = min(block_rows, total_rows - b * block_rows);
    const int kept = min(keep, nrows);
    if (i >= kept) return;
    const int prev = b * keep;                                   // all earlier blocks are full, so they kept `keep` rows each
    const double* src = H + (size_t)(b * block_rows + i) * ld;
    double* dst = Hout + (size_t)(prev + i) * ld;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) dst[c] = src[c];
    if (threadIdx.x == 0) rout[prev + i] = r[b * block_rows + i];
}

lvk_status lvk_qr_compress_dev(lvk_context* ctx, double* d_H, int ldh, int rows, int cols, double* d_r, int* rows_out)
{
    if (cols > QR_BLOCK_ROWS / 2) return lvk_set_error(ctx, LVK_ERR_CAPACITY, "QR compression supports up to %d columns", QR_BLOCK_ROWS / 2);
    int cur_rows = rows;
    double* tmpH = nullptr; double* tmpr = nullptr;
    while (cur_rows > cols) {
        const int nblk = (cur_rows + QR_BLOCK_ROWS - 1) / QR_BLOCK_ROWS;
        hipLaunchKernelGGL(k_qr_block, dim3(nblk), dim3(QR_THREADS), 0, ctx->stream, d_H, ldh, cols, d_r, cur_rows, QR_BLOCK_ROWS);
        if (nblk == 1) { cur_rows = cols; break; }
        // pack the triangles: block b keeps min(cols, its rows) rows
        if (!tmpH) {
            tmpH = (double*)lvk_ctx_scratch(ctx, 7, sizeof(double) * (size_t)nblk * cols * ldh);
            tmpr = (double*)lvk_ctx_scratch(ctx, 8, sizeof(double) * (size_t)nblk * cols);
            if (!tmpH || !tmpr) return lvk_set_error(ctx, LVK_ERR_DEVICE, "scratch allocation failed");
        }
        const int last_rows = cur_rows - (nblk - 1) * QR_BLOCK_ROWS;
        const int new_rows = (nblk - 1) * cols + (last_rows < cols ? last_rows : cols);
        hipLaunchKernelGGL(k_qr_pack, dim3(cols, nblk), dim3(128), 0, ctx->stream, (const double*)d_H, ldh, cols, (const double*)d_r, cur_rows, QR_BLOCK_ROWS, cols, tmpH, tmpr);
        LVK_HIP(ctx, hipMemcpy2DAsync(d_H, sizeof(double) * ldh, tmpH, sizeof(double) * ldh, sizeof(double) * cols, new_rows, hipMemcpyDeviceToDevice, ctx->stream));
        LVK_HIP(ctx, hipMemcpyAsync(d_r, tmpr, sizeof(double) * new_rows, hipMemcpyDeviceToDevice, ctx->stream));
        cur_rows = new_rows;
    }
    LVK_LAUNCH_CHECK(ctx);
    *rows_out = cur_rows < cols ? cur_rows : cols;
    return LVK_OK;
}

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
    for (int i = 0; i < n_groups; ++i) { g[i].start = 0; g[i].rows = rows[i]; g[i].cols.assign(cols + col_off[i], cols + col_off[i + 1]); }
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
        for (size_t k = 0; k < levels[l].blocks.size(); ++k) memcpy(h_blocks + 8 * (nb + (int)k), &levels[l].blocks[k], 8 * sizeof(int));   // the first 8 ints of QrBlock
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
    for (auto& x : g) for (size_t k = 0; k < x.cols.size(); ++k)
        if (x.cols[k] < 0 || x.cols[k] >= cols || (k > 0 && x.cols[k] <= x.cols[k - 1])) return lvk_set_error(ctx, LVK_ERR_ARG, "lvk_ekf_compress_qr_groups: column lists must be ascending and < cols");
    std::vector<QrPlanLevel> levels; int m2 = rows;
    lvk_qr_sparse_plan(g, cols, levels, &m2);
    *rows_out = rows;
    if (levels.empty()) return LVK_OK;
    double* Hb = (double*)lvk_ctx_scratch(ctx, 7, sizeof(double) * (size_t)rows * ld);
    double* rb = (double*)lvk_ctx_scratch(ctx, 8, sizeof(double) * (size_t)rows);
    size_t nb = 0, ncl = 0, nch = 0; for (auto& L : levels) { nb += L.blocks.size(); ncl += L.cols.size() + 1; nch += L.chunks.size(); }
    const size_t o_ch = (sizeof(QrBlock) * nb + 63) & ~(size_t)63, o_cl = (o_ch + sizeof(QrChunk) * nch + 63) & ~(size_t)63;
    char* meta = (char*)lvk_ctx_scratch(ctx, 9, o_cl + sizeof(int) * ncl);
    if (!Hb || !rb || !meta) return lvk_set_error(ctx, LVK_ERR_DEVICE, "scratch allocation failed");
    QrBlock* d_blocks = (QrBlock*)meta; QrChunk* d_chunks = (QrChunk*)(meta + o_ch); int* d_cols = (int*)(meta + o_cl);
    std::vector<QrBlock> hb; std::vector<QrChunk> hch; std::vector<int> hc;
    for (auto& L : levels) { hb.insert(hb.end(), L.blocks.begin(), L.blocks.end()); hch.insert(hch.end(), L.chunks.begin(), L.chunks.end());
                             hc.insert(hc.end(), L.cols.begin(), L.cols.end()); hc.push_back(0); }
    LVK_HIP(ctx, hipMemcpyAsync(d_blocks, hb.data(), sizeof(QrBlock) * nb, hipMemcpyHostToDevice, ctx->stream));
    if (nch) LVK_HIP(ctx, hipMemcpyAsync(d_chunks, hch.data(), sizeof(QrChunk) * nch, hipMemcpyHostToDevice, ctx->stream));
    LVK_HIP(ctx, hipMemcpyAsync(d_cols, hc.data(), sizeof(int) * ncl, hipMemcpyHostToDevice, ctx->stream));
    LVK_HIP(ctx, hipStreamSynchronize(ctx->stream));                  // hb / hch / hc are stack-local
    double* H = d_H; double* r = d_r; size_t ob = 0, oc = 0, och = 0;
    for (auto& L : levels) {
        double* Ho = (H == d_H) ? Hb : d_H; double* ro = (r == d_r) ? rb : d_r;
        lvk_status st = lvk_qr_sparse_level(ctx, H, ld, r, Ho, ld, ro, d_blocks + ob, (int)L.blocks.size(), d_chunks + och, (int)L.chunks.size(), d_cols + oc, cols,
                                            L.max_nc, L.part_tiles * 256);
        if (st != LVK_OK) return st;
        ob += L.blocks.size(); oc += L.cols.size() + 1; och += L.chunks.size();
        H = Ho; r = ro;
    }
    if (H != d_H) {
        LVK_HIP(ctx, hipMemcpy2DAsync(d_H, sizeof(double) * ld, H, sizeof(double) * ld, sizeof(double) * cols, m2, hipMemcpyDeviceToDevice, ctx->stream));
        LVK_HIP(ctx, hipMemcpyAsync(d_r, r, sizeof(double) * m2, hipMemcpyDeviceToDevice, ctx->stream));
    }
    *rows_out = m2;
    return LVK_OK;
}
