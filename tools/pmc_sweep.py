#!/usr/bin/env python3
"""Driver for the PMC passes over the triangular sweep: the C3 shifted-matrix factor, 10 sweeps of 32 columns and,
as a byte-count calibration of the counters, 20 column-dot launches over two n x 32 blocks (2 * 8 * n * 32 bytes)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigd_amd.device import Factor, default_context  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

ctx = default_context()
col = BucklingColumn(706, 706, seed=0)
K = col.stiffness()
n = K.shape[0]
F = Factor(ctx, K, coords=col.dof_coords())
rng = np.random.default_rng(0)
kcols = int(os.environ.get("SWEEP_K", "32"))       # (development: the narrow sweeps' levels with SWEEP_K=4)
B = ctx.from_host(rng.normal(size=(n, kcols)))
X = ctx.empty(n, kcols)
for _ in range(10):
    F.solve_to(B, X)
ctx.sync()
B = ctx.from_host(rng.normal(size=(n, 32)))
Y = ctx.from_host(rng.normal(size=(n, 32)))
for _ in range(20):
    B.coldot(Y)
ctx.sync()
print("n", n, "nnzL", F.stats()["nnzL"], "sweep algorithmic bytes (k=32)", F.solve_bytes(32), "coldot bytes", 2 * 8 * n * 32)
