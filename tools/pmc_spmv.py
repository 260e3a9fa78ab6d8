#!/usr/bin/env python3
"""Small driver for the PMC passes: the C3 stiffness matrix, 20 SpMV launches, 20 SpMM launches of 32 columns and, as a byte-count
calibration of the counters for 8-byte-per-lane streams, 20 column-dot launches over two n x 32 blocks
(exactly 2 * 8 * n * 32 bytes read each)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from eigd_amd.device import CSRMatrix, default_context  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

ctx = default_context()
col = BucklingColumn(706, 706, seed=0)
K = col.stiffness()
n = K.shape[0]
dK = CSRMatrix(ctx, K)
rng = np.random.default_rng(0)
x = ctx.from_host(rng.normal(size=n))
y = ctx.empty(n, 1)
X = ctx.from_host(rng.normal(size=(n, 32)))
Y = ctx.from_host(rng.normal(size=(n, 32)))
for _ in range(20):
    dK.apply(x, y)
ctx.sync()
X2 = ctx.empty(n, 32)
for _ in range(20):                     # the tiled SpMM at 32 columns
    dK.apply(X, X2)
ctx.sync()
del X2
for _ in range(20):
    X.coldot(Y)
ctx.sync()
print("n", n, "nnz", K.nnz, "spmv algorithmic bytes", dK.spmv_bytes(1), "spmm(32)", dK.spmv_bytes(32), "coldot bytes", 2 * 8 * n * 32)
