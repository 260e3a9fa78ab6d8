#!/usr/bin/env python3
"""wall time of the SpMM on the C3 stiffness matrix at several widths, with a digest of the result (development aid)"""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigd_amd.device import CSRMatrix, default_context  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

ctx = default_context()
col = BucklingColumn(706, 706, seed=0)
K = col.stiffness()
n = K.shape[0]
dK = CSRMatrix(ctx, K)
rng = np.random.default_rng(0)
out = []
for k in (2, 4, 8, 16, 20, 32, 64):
    Xh = rng.normal(size=(n, k))
    X = ctx.from_host(Xh)
    Y = ctx.empty(n, k)
    for _ in range(3):
        dK.apply(X, Y)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        dK.apply(X, Y)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 20
    y = Y.get()
    exact = np.array_equal(y, K @ Xh)
    out.append(f"k={k}: {1e6 * dt:.1f} us [{hashlib.sha1(y.tobytes()).hexdigest()[:8]}{'=' if exact else '!'}]")
print(" ".join(out))
