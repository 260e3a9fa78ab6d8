#!/usr/bin/env python3
"""development aid: tolerance of the extra (deflation-only) eigenpairs against eigensolve time, Krylov steps and psi (C3)"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eigd_amd as eg  # noqa: E402
from eigd_amd.device import CSRMatrix  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

warnings.simplefilter("ignore")
col = BucklingColumn(706, 706, seed=0)
K = col.stiffness()
u = col.full_vector(eg.SpLuOperator(K, check_symmetry=False, coords=col.dof_coords())(col.f[col.reduced]))
G = col.geometric_stiffness(u)
sigma, N = 1.0971, 32
fac = eg.SpLuOperator((K + sigma * G).tocsr(), coords=col.dof_coords(), check_symmetry=False)
ctx = fac.ctx
dK, dG = CSRMatrix(ctx, K), CSRMatrix(ctx, G)
dPhib = ctx.from_host(np.random.default_rng(1).uniform(size=(K.shape[0], N)))
ref = None
for tol in [float(v) for v in (sys.argv[1:] or ["1e-11", "1e-10", "1e-9", "1e-8"])]:
    eg.tuning.iram_extra_tol = tol
    ts = []
    for rep in range(3):
        s = lam = Phi = None
        s = eg.IRAM(N=N, m=65, mode="buckling", ctx=ctx)
        ctx.sync()
        t0 = time.perf_counter()
        lam, Phi = s.solve(dG, dK, fac, sigma)
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    tt = []
    for rep in range(2):
        ctx.sync()
        t0 = time.perf_counter()
        dpsi, data = s.solve_adjoint(dPhib, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
        ctx.sync()
        tt.append(time.perf_counter() - t0)
    psi = dpsi.get()
    if ref is None:
        ref = psi
    res, _ = s.eval_adjoint_residual_norm(dPhib, dpsi, b_ortho=True)
    print(f"extra tol {tol:.0e}: eigensolve {1e3 * min(ts[1:]):.0f} ms (sweeps {s.sweeps}, restarts {s.n_restarts}, extras {s.n_extra}), "
          f"solve_adjoint {1e3 * min(tt):.1f} ms, steps {sum(s.last_info)} (longest {max(s.last_info)}), psi vs first {np.linalg.norm(psi - ref) / np.linalg.norm(ref):.1e}, "
          f"residual max {res.max():.1e}", flush=True)
