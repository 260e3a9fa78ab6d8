#!/usr/bin/env python3
"""development aid: k-column sweep times for several leaf sizes of the nested dissection (Factor(..., leaf_size=)), two
factors per size in ONE process, interleaved and alternating (sweep times move by +-2 % with where allocations landed):
    LEAFS=64,48,96 WIDTHS=8,32 python tools/leaf_ab_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigd_amd.device import Factor, Symbolic, default_context  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

ctx = default_context()
col = BucklingColumn(706, 706, seed=0)
K = col.stiffness()
leafs = [int(v) for v in os.environ.get("LEAFS", "64,48,96").split(",")]
syms = {lf: Symbolic(K, leaf_size=lf, coords=col.dof_coords()) for lf in leafs}
for lf in leafs:
    s = syms[lf].sizes
    print(f"leaf {lf}: fronts {s['nfronts']} levels {s['nlevels']} nnzL {s['nnzL'] / 1e6:.1f} M sumd {s['sumd'] / 1e6:.2f} M", flush=True)
facs = [(lf, Factor(ctx, K, symbolic=syms[lf])) for _ in range(2) for lf in leafs]
rng = np.random.default_rng(0)
for k in tuple(int(v) for v in os.environ.get("WIDTHS", "8,32").split(",")):
    B = ctx.from_host(rng.normal(size=(K.shape[0], k)))
    X = ctx.empty(K.shape[0], k)
    res = {i: [] for i in range(len(facs))}
    for rep in range(5):
        for i, (lf, F) in enumerate(facs):
            for _ in range(2):
                F.solve_to(B, X)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(20):
                F.solve_to(B, X)
            ctx.sync()
            res[i].append((time.perf_counter() - t0) / 20 * 1e3)
    for lf in leafs:
        meds = [float(np.median(res[i])) for i, (l2, _) in enumerate(facs) if l2 == lf]
        print(f"k={k:2d} leaf {lf:3d}: " + " ".join(f"{m:.4f}" for m in meds) + f"  mean {np.mean(meds):.4f} ms", flush=True)
