#!/usr/bin/env python3
"""wall time of k-column sweeps on the C3 factor (development aid); DIGEST=1 adds a hash of each result, EIGD_LIB picks the build"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigd_amd.device import Factor, default_context  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

ctx = default_context()
col = BucklingColumn(706, 706, seed=0)
K = col.stiffness()
F = Factor(ctx, K, coords=col.dof_coords(), leaf_size=int(os.environ.get("LEAF", "0")))
rng = np.random.default_rng(0)
out = []
for k in tuple(int(v) for v in os.environ.get("WIDTHS", "1,4,8,16,32").split(",")):
    B = ctx.from_host(rng.normal(size=(K.shape[0], k)))
    X = ctx.empty(K.shape[0], k)
    for _ in range(3):
        F.solve_to(B, X)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        F.solve_to(B, X)
    ctx.sync()
    out.append(f"k={k}: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms")
    if os.environ.get("DIGEST"):  # bitwise comparison of kernel variants between runs
        import hashlib

        out.append("[" + hashlib.sha1(X.get().tobytes()).hexdigest()[:10] + "]")
x = X.get()
r = np.linalg.norm(K @ x - B.get()) / np.linalg.norm(B.get())
print(" ".join(out), f"resid {r:.1e}")
