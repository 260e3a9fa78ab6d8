#!/usr/bin/env python3
"""sweep time vs leaf size (development aid)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from eigd_amd.device import Factor, Symbolic, default_context  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

ctx = default_context()
col = BucklingColumn(706, 706, seed=0)
K = col.stiffness()
n = K.shape[0]
coords = col.dof_coords()
rng = np.random.default_rng(0)
for leaf in (16, 24, 32, 48, 64, 96, 128):
    t0 = time.time()
    sym = Symbolic(K, leaf_size=leaf, coords=coords)
    ts = time.time() - t0
    F = Factor(ctx, K, symbolic=sym)
    out = [f"leaf={leaf} symbolic {ts:.1f}s nfronts={sym.sizes['nfronts']} levels={sym.sizes['nlevels']} nnzL={sym.sizes['nnzL']/1e6:.0f}M steps={sym.sizes['nlaunch_steps']}"]
    for k in (1, 4, 32):
        B = ctx.from_host(rng.normal(size=(n, k)))
        F.solve_inplace(B)
        ctx.sync()
        ctx.timer_start()
        for _ in range(5):
            F.solve_inplace(B)
        out.append(f"k={k}: {ctx.timer_stop_ms()/5:.2f} ms")
    print("  ".join(out), flush=True)
    del F
