#!/usr/bin/env python3
"""development aid: the C3 eigensolve under the two Gram-Schmidt forms of the block Lanczos step (tuning.lanczos_local_first_pass),
alternating on one box: time, sweeps, restarts, eigenvalue difference, true residuals, B-orthonormality of the returned vectors"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eigd_amd as eg  # noqa: E402
from eigd_amd.device import CSRMatrix, default_context  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

ctx = default_context()
nx = int(os.environ.get("NX", "706"))
col = BucklingColumn(nx, nx, seed=0)
K = col.stiffness()
coords = col.dof_coords()
Kfac = eg.SpLuOperator(K, ctx=ctx, check_symmetry=False, coords=coords)
u = col.full_vector(Kfac(col.f[col.reduced]))
G = col.geometric_stiffness(u)
sigma = 1.0971
fac = eg.SpLuOperator((K + sigma * G).tocsr(), ctx=ctx, symbolic=Kfac.symbolic, check_symmetry=False, coords=coords)
dK, dG = CSRMatrix(ctx, K), CSRMatrix(ctx, G)
ref = None
for rep in range(6):
    eg.tuning.lanczos_local_first_pass = bool(rep % 2)
    lam = Phi = s = None
    s = eg.IRAM(N=32, m=65, mode="buckling", ctx=ctx)
    ctx.sync()
    t0 = time.perf_counter()
    lam, Phi = s.solve(dG, dK, fac, sigma)
    ctx.sync()
    t = time.perf_counter() - t0
    if ref is None:
        ref = lam.copy()
    gram = Phi.T @ (K @ Phi)
    res = np.linalg.norm(G @ Phi + (K @ Phi) / lam[None, :] * 1.0, axis=0) if False else None
    print(f"local_first_pass={eg.tuning.lanczos_local_first_pass}: solve {t:.3f} s, sweeps {s.sweeps}, restarts {s.n_restarts}, "
          f"extras {s.n_extra}, reorth passes {s._dev.reorth_passes}, max |dlam|/|lam| {np.max(np.abs(lam - ref) / np.abs(ref)):.1e}, "
          f"eig_res_true max {np.max(getattr(s, 'eig_res_true', [np.nan])):.1e}, |Phi^T K Phi - I| {np.max(np.abs(gram - np.eye(32))):.1e}", flush=True)
