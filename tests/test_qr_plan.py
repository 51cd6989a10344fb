"""The structure-aware compression of the stacked measurement rows (larvio_amd/csrc/be_qr.hip): the tree of nodes is planned on the
host (lvk_ekf_qr_plan - no device needed), so the plan itself is checked here on CPU: an emulation of every node in numpy (any
factor R of the node's rows restricted to its columns with R^T R = A^T A) must preserve H^T H and H^T r exactly as the kernels have
to, and every node's column union must fit the Gram a workgroup holds in LDS.  tests/test_gpu_backend.py runs the same shapes through the kernel."""
import numpy as np
import pytest


def _msckf_like(seed, n_feat, n_clones, leg=22, track=6, n_state_feat=0, burst=False):
    """groups and a matching random H for `n_feat` MSCKF features (2 M - 3 rows, columns 15..21 + 6-blocks of M consecutive clones
    ending near the newest) followed by `n_state_feat` in-state features (2 rows: extrinsics, anchor block, newest clone, own column)"""
    rng = np.random.default_rng(seed)
    N = leg + 6 * n_clones + n_state_feat
    groups, rows_H = [], []
    for f in range(n_feat):
        M = track if burst else int(rng.integers(3, track + 1))
        last = n_clones - 1 if burst else int(rng.integers(max(M - 1, n_clones - 3), n_clones))
        cols = list(range(15, 22)) + [leg + 6 * c + k for c in range(last - M + 1, last + 1) for k in range(6)]
        cols = sorted(set(cols))
        r = 2 * M - 3
        B = np.zeros((r, N)); B[:, cols] = rng.normal(0, 1, (r, len(cols)))
        groups.append((r, cols)); rows_H.append(B)
    for s in range(n_state_feat):
        anchor = int(rng.integers(0, n_clones - 1))
        cols = sorted(set(list(range(15, 22)) + [leg + 6 * anchor + k for k in range(6)] + [leg + 6 * (n_clones - 1) + k for k in range(6)] + [leg + 6 * n_clones + s]))
        B = np.zeros((2, N)); B[:, cols] = rng.normal(0, 1, (2, len(cols)))
        groups.append((2, cols)); rows_H.append(B)
    H = np.vstack(rows_H); r = rng.normal(0, 1, len(H))
    return N, groups, H, r


def _lds_bytes(rows, ncols, N):
    return 8 * ((ncols + 1) * (rows | 1) + ncols + 2) + 4 * N + 16


def emulate(levels, H, r):
    """what k_qr_sparse has to compute, node by node"""
    N = H.shape[1]
    for L in levels:
        out_rows = sum(b["out_rows"] for b in L["blocks"])
        Ho = np.zeros((out_rows, N)); ro = np.zeros(out_rows)
        for b in L["blocks"]:
            A = H[b["in_start"]:b["in_start"] + b["in_rows"]]; a = r[b["in_start"]:b["in_start"] + b["in_rows"]]
            if b["copy"]:
                Ho[b["out_start"]:b["out_start"] + b["out_rows"]] = A; ro[b["out_start"]:b["out_start"] + b["out_rows"]] = a
                continue
            cols = L["cols"][b["col_off"]:b["col_off"] + b["ncols"]]
            rest = np.setdiff1d(np.arange(N), cols)
            assert not np.any(A[:, rest]), "a node's rows are non-zero outside its column union"
            Q, R = np.linalg.qr(np.column_stack([A[:, cols], a]), mode="reduced")
            k = b["out_rows"]
            assert k == min(b["in_rows"], b["ncols"])
            Ho[b["out_start"]:b["out_start"] + k][:, cols] = R[:k, :-1]; ro[b["out_start"]:b["out_start"] + k] = R[:k, -1]
        H, r = Ho, ro
    return H, r


@pytest.mark.parametrize("case", ["steady_A", "burst_5", "steady_5", "long_tracks"])
def test_plan_preserves_the_information_and_fits_the_lds(case):
    from larvio_amd import larvio as lv
    if case == "steady_A":          # configs[1]: ~25 MSCKF features + 30 in-state ones, 30 clones
        N, groups, H, r = _msckf_like(1, 25, 30, n_state_feat=30)
    elif case == "burst_5":         # configs[4]: a generation of 1900 features reaches max_track_len together: 17,100 rows
        N, groups, H, r = _msckf_like(2, 1900, 60, n_state_feat=60, burst=True)
    elif case == "steady_5":
        N, groups, H, r = _msckf_like(3, 330, 60, n_state_feat=60)
    else:                           # 20-observation tracks: unions of up to 127 columns
        N, groups, H, r = _msckf_like(4, 200, 40, track=20)
    levels, final_rows = lv.qr_plan(N, groups)
    rows = len(H)
    G0, g0 = H.T @ H, H.T @ r
    for L in levels:
        start = 0; out = 0
        for b in L["blocks"]:
            assert b["in_start"] == start and b["out_start"] == out         # consecutive, nothing skipped
            start += b["in_rows"]; out += b["out_rows"]
            if not b["copy"]:
                assert b["in_rows"] > b["ncols"] and _lds_bytes(b["in_rows"], b["ncols"], N) <= 152 * 1024
                c = L["cols"][b["col_off"]:b["col_off"] + b["ncols"]]
                assert np.all(np.diff(c) > 0) and c[-1] < N
        assert start == rows and out * 5 <= rows * 4                        # a level removes at least a fifth of the rows
        rows = out
    assert final_rows == rows
    H2, r2 = emulate(levels, H, r)
    assert len(H2) == final_rows
    G1, g1 = H2.T @ H2, H2.T @ r2
    assert np.abs(G1 - G0).max() <= 1e-10 * np.abs(G0).max() and np.abs(g1 - g0).max() <= 1e-10 * np.abs(g0).max()
    if case == "steady_A":
        assert levels and final_rows <= 60 + 60                             # MSCKF part <= its column union (7 + 6 * 8), in-state rows pass through
    if case == "burst_5":
        assert final_rows <= 43 + 120 and len(levels) <= 5
    if case == "steady_5":
        assert final_rows <= 61 + 120
    print(case, "rows", len(H), "->", final_rows, "levels", [(len(L["blocks"]), sum(b["out_rows"] for b in L["blocks"])) for L in levels])


def test_plan_leaves_small_or_wide_problems_alone():
    from larvio_amd import larvio as lv
    N, groups, H, r = _msckf_like(5, 0, 20, n_state_feat=20)                  # only in-state features: nothing shrinks
    levels, final_rows = lv.qr_plan(N, groups)
    assert levels == [] and final_rows == 40
    levels, final_rows = lv.qr_plan(100, [(50, list(range(100)))])             # one dense group, rows < columns
    assert levels == [] and final_rows == 50
    levels, final_rows = lv.qr_plan(100, [])
    assert levels == [] and final_rows == 0
