"""one dense compression of a 3536 x 442 block (the stacked triangles of 8 ranks at configs[4]) for a rocprofv3 kernel table"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import larvio_amd
from larvio_amd import larvio as lv
ctx = larvio_amd.Context(0)
rng = np.random.default_rng(0)
for rows, cols in ((3536, 442), (18000, 442)):
    H = rng.normal(0, 1, (rows, cols)); r = rng.normal(0, 1, rows)
    for _ in range(3):
        R, rc = lv.compress_qr(ctx, H, r)
    print(rows, cols, np.abs(R.T @ R - H.T @ H).max() / np.abs(H.T @ H).max())
