#!/usr/bin/env python3
"""development aid: 32-column sweep times of ONE factor through its own vector workspaces and through three sweep lanes
(other streams' workspaces: carry planes, Y, partial slabs) -- is the spread between factors of one matrix a matter of
where the factor's arrays landed or of where the vector workspaces did?"""
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
k = 32
rng = np.random.default_rng(0)
Bh = rng.normal(size=(K.shape[0], k))
for fi in range(2):
    F = Factor(ctx, K, coords=col.dof_coords())
    ctxs = [ctx] + [ctx.fork(i + 1) for i in range(3)]
    blocks = [(c, c.from_host(Bh), c.empty(K.shape[0], k)) for c in ctxs]
    res = {i: [] for i in range(len(ctxs))}
    for rep in range(5):
        for i, (c, B, X) in enumerate(blocks):
            for _ in range(2):
                F.solve_to(B, X)
            c.sync()
            t0 = time.perf_counter()
            for _ in range(20):
                F.solve_to(B, X)
            c.sync()
            res[i].append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"factor {fi}: own workspace {np.median(res[0]):.4f}  lanes " + " ".join(f"{np.median(res[i]):.4f}" for i in (1, 2, 3)) + " ms", flush=True)
