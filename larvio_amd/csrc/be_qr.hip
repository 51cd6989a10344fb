// be_qr.hip — Householder compression of the tall measurement matrix: [H | r] (rows x cols) -> top `cols` rows of
// Q^T [H | r], in place.  Replaces the SuiteSparse SPQR calls of /root/reference/src/larvio.cpp:1430-1445,2209-2229
// (natural ordering, Q^T applied densely): any orthogonal Q that zeroes the rows below `cols` preserves H^T H and
// H^T r, which is all the update uses (isotropic noise).  Structure: TSQR — row blocks of QR_BLOCK_ROWS are reduced
// independently (one workgroup each, reflectors kept in LDS, columns spread over the wavefronts, per-column dot
// products by wave reduction), the surviving triangles are stacked and reduced again until one block remains.
// At the north-star size (a few hundred rows, ~200 columns) that is a single workgroup; at 18,000 rows (config 5)
// the first level runs ~36 workgroups in parallel.
#include "lvk_internal.h"
#include "chi2_table.inc"

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
