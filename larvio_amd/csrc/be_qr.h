// be_qr.h — the structure-aware compression of the stacked measurement rows: plan (host) + level launcher (be_qr.hip).
#pragma once
#include "lvk_internal.h"
#include <memory>
#include <vector>

// A run of consecutive stacked rows that share one column set (the rows of one feature job): what the TSQR tree is planned from.
// cols = the dense columns the rows can be non-zero in (ascending), as k_feature_rows lays them out: extrinsics + td 15..21, the
// observing clones' 6-blocks, the anchor's block and the feature's own column for in-state features.
// (column lists are shared: at configs[4] two thousand features of one generation carry the same list)
typedef std::shared_ptr<const std::vector<int>> ColList;
struct RowGroup { int start, rows; ColList cols; int owner = 0; };   // owner: the rank that builds these rows in the sharded update
// one node of one level: rows [in_start, in_start + in_rows) of the level's input, restricted to ncols columns (col_lists + col_off),
// reduced to out_rows = min(in_rows, ncols) rows written at out_start of the level's output; copy = pass the rows through unchanged
struct QrBlock { int in_start, in_rows, out_start, out_rows, ncols, col_off, copy, pad; };
struct QrPlanLevel { std::vector<QrBlock> blocks; std::vector<int> cols; int in_rows = 0, out_rows = 0; size_t lds = 0; int max_rows = 0, max_cols = 0; };   // max_*: over the level's factoring nodes

size_t lvk_qr_sparse_lds_bytes(int rows, int ncols, int N);
// final_groups (optional): the row groups of the result (consecutive, with their column unions) - the input of a further stage
void lvk_qr_sparse_plan(const std::vector<RowGroup>& groups, int N, std::vector<QrPlanLevel>& levels, int* final_rows, std::vector<RowGroup>* final_groups = nullptr);
lvk_status lvk_qr_sparse_level(lvk_context* ctx, const double* d_Hin, int ldin, const double* d_rin, double* d_Hout, int ldout, double* d_rout,
                               const QrBlock* d_blocks, int n_blocks, const int* d_cols, int N, size_t max_lds, int max_rows, int max_cols);
