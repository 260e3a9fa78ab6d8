#!/usr/bin/env python3
"""development aid: k-column sweep times of FOUR factors of the C3 matrix in one process, alternating: two whose workgroups
gather their right-hand sides themselves and two with every multi-tile level pre-assembled (EIGD_PRE_MIN_WG read at factor
creation) -- the time of a sweep depends on where a process's allocations landed (+-2 % from process to process), which a
comparison of builds or settings across processes cannot tell from their effect"""
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
facs = []
for thr in os.environ.get("THRS", "100000000,1,100000000,1").split(","):
    os.environ["EIGD_PRE_MIN_WG"] = thr
    facs.append((thr, Factor(ctx, K, coords=col.dof_coords())))
rng = np.random.default_rng(0)
for k in tuple(int(v) for v in os.environ.get("WIDTHS", "8,16,32").split(",")):
    B = ctx.from_host(rng.normal(size=(K.shape[0], k)))
    X = ctx.empty(K.shape[0], k)
    res = {i: [] for i in range(len(facs))}
    for rep in range(6):
        for i, (thr, F) in enumerate(facs):
            for _ in range(2):
                F.solve_to(B, X)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(20):
                F.solve_to(B, X)
            ctx.sync()
            res[i].append((time.perf_counter() - t0) / 20 * 1e3)
    for i, (thr, F) in enumerate(facs):
        print(f"k={k:2d} factor {i} (pre-assembly threshold {thr:>9s}): " + " ".join(f"{t:.4f}" for t in res[i]) + f"  median {np.median(res[i]):.4f} ms", flush=True)
