#!/usr/bin/env python3
"""development aid: cProfile of IRAM.solve on the C3 problem (where does the host see the time of the eigensolve go)"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eigd_amd as eg  # noqa: E402
from eigd_amd.device import default_context  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

ctx = default_context()
nx = int(os.environ.get("NX", "706"))
col = BucklingColumn(nx, nx, seed=0)
K = col.stiffness()
coords = col.dof_coords()
Kfac = eg.SpLuOperator(K, ctx=ctx, check_symmetry=False, coords=coords)
u = col.full_vector(Kfac(col.f[col.reduced]))
G = col.geometric_stiffness(u)
sigma = float(os.environ.get("SIGMA", "1.0971"))
fac = eg.SpLuOperator((K + sigma * G).tocsr(), ctx=ctx, symbolic=Kfac.symbolic, check_symmetry=False, coords=coords)
from eigd_amd.device import CSRMatrix  # noqa: E402

dK, dG = CSRMatrix(ctx, K), CSRMatrix(ctx, G)
for rep in range(2):
    s = eg.IRAM(N=32, m=65, mode="buckling", ctx=ctx)
    ctx.sync()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    lam, Phi = s.solve(dG, dK, fac, sigma)
    ctx.sync()
    pr.disable()
    print(f"rep {rep}: solve {time.perf_counter() - t0:.3f} s, sweeps {s.sweeps}, restarts {s.n_restarts}, block {s.block_size}, "
          f"basis {s.internal_basis}, extras {s.n_extra}, reorth passes {s._dev.reorth_passes}", flush=True)
pstats.Stats(pr, stream=sys.stdout).sort_stats("cumulative").print_stats(40)
