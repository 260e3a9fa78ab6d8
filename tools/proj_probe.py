#!/usr/bin/env python3
"""development aid: time of the projection X <- X - U (V^T X) at the C3 size for the shapes a step uses"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigd_amd.device import default_context  # noqa: E402

ctx = default_context()
n = 998284
rng = np.random.default_rng(0)
for ku in (32, 63):
    U, V = ctx.from_host(rng.normal(size=(n, ku))), ctx.from_host(rng.normal(size=(n, ku)))
    for kx in (8, 32, 64):
        X = ctx.from_host(rng.normal(size=(n, kx)))
        ref = None
        for reps, label in ((1, "check"), (10, "time")):
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                C = V.tdot(X)
            ctx.sync()
            t_tn = (time.perf_counter() - t0) / reps
            t0 = time.perf_counter()
            for _ in range(reps):
                X.project(U, V)
            ctx.sync()
            t_pr = (time.perf_counter() - t0) / reps
        Ch = V.get().T @ ctx.from_host(rng.normal(size=(1, 1))).get() if False else None
        Xh, Vh = X.get(), V.get()
        err = np.abs(V.tdot(X) - Vh.T @ Xh).max() / np.abs(Vh.T @ Xh).max()
        print(f"ku={ku} kx={kx}: tdot {1e3 * t_tn:.3f} ms, project {1e3 * t_pr:.3f} ms, tdot rel-err {err:.1e}", flush=True)
