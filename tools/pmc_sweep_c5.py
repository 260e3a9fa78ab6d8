#!/usr/bin/env python3
"""Driver for PMC passes over the triangular sweep of config C5 (2 M-dof shell box): the factor of K + 6.75 G, 10 sweeps of
SWEEP_K (32) columns, 20 column-dot launches as calibration (see tools/pmc_sweep.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigd_amd.problems import ShellBox, ShellBoxOnDevice  # noqa: E402

box = ShellBox(832, 160, 40, nseg=2, seed=0)
dev = ShellBoxOnDevice(box)
ctx = dev.ctx
dev.assemble()
assert dev.refactor(6.75) == 0
F = dev.factor.factor
n = box.n
rng = np.random.default_rng(0)
kcols = int(os.environ.get("SWEEP_K", "32"))
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
