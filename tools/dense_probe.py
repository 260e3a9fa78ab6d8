#!/usr/bin/env python3
"""timings of the dense / sparse block kernels at the C3 size (development aid)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigd_amd.device import CSRMatrix, default_context  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402


def timeit(ctx, fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop_ms() / reps * 1e3


ctx = default_context()
col = BucklingColumn(706, 706, seed=0)
K = col.stiffness()
n = K.shape[0]
dK = CSRMatrix(ctx, K)
rng = np.random.default_rng(0)
Phi = ctx.from_host(rng.normal(size=(n, 32)))
BPhi = ctx.from_host(rng.normal(size=(n, 32)))
for k in (4, 8, 16, 32):
    X = ctx.from_host(rng.normal(size=(n, k)))
    Y = ctx.empty(n, k)
    t_spmm = timeit(ctx, lambda: dK.apply(X, Y))
    t_tn = timeit(ctx, lambda: BPhi.tdot(X))
    t_proj = timeit(ctx, lambda: X.project(Phi, BPhi))
    t_cn = timeit(ctx, lambda: X.colnorms())
    print(f"k={k:2d}: spmm {t_spmm:7.1f} us ({(12 * K.nnz + 4 * n + 16 * n * k) / t_spmm / 1e6:6.2f} TB/s)  "
          f"gemm_tn {t_tn:6.1f} us ({8 * n * (32 + k) / t_tn / 1e6:5.2f} TB/s)  project {t_proj:6.1f} us "
          f"({8 * n * (64 + 3 * k) / t_proj / 1e6:5.2f} TB/s)  colnorms {t_cn:6.1f} us", flush=True)
