#!/usr/bin/env python3
"""development aid: the C3 eigensolve (repeated call) for several block sizes / internal basis sizes of the restarted block
Lanczos (tuning.iram_block, tuning.iram_basis):  python tools/iram_plan_probe.py 8:0 8:160 16:160 16:192 ..."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eigd_amd as eg  # noqa: E402
from eigd_amd.device import CSRMatrix, default_context  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

ctx = default_context()
col = BucklingColumn(706, 706, seed=0)
K = col.stiffness()
coords = col.dof_coords()
Kfac = eg.SpLuOperator(K, ctx=ctx, check_symmetry=False, coords=coords)
u = col.full_vector(Kfac(col.f[col.reduced]))
G = col.geometric_stiffness(u)
sigma = 1.0971
fac = eg.SpLuOperator((K + sigma * G).tocsr(), ctx=ctx, symbolic=Kfac.symbolic, check_symmetry=False, coords=coords)
dK, dG = CSRMatrix(ctx, K), CSRMatrix(ctx, G)
rng = np.random.default_rng(1)
dPhib = ctx.from_host(rng.uniform(size=(K.shape[0], 32)))
for spec in sys.argv[1:] or ["8:0"]:
    parts = [int(v) for v in spec.split(":")]
    p, basis = parts[0], parts[1]
    eg.tuning.iram_block, eg.tuning.iram_basis = p, basis
    eg.tuning.iram_extra = parts[2] if len(parts) > 2 else None
    ts = []
    for rep in range(3):
        s = eg.IRAM(N=32, m=65, mode="buckling", ctx=ctx)
        ctx.sync()
        t0 = time.perf_counter()
        lam, Phi = s.solve(dG, dK, fac, sigma)
        ctx.sync()
        ts.append(time.perf_counter() - t0)
        del lam, Phi
    ta = []
    for rep in range(3):                                   # the adjoint solves that this set of deflated pairs leaves
        ctx.sync()
        t0 = time.perf_counter()
        dpsi, data = s.solve_adjoint(dPhib, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
        ctx.sync()
        ta.append(time.perf_counter() - t0)
    steps = max(s.last_info)
    from eigd_amd import adjoint as _adj
    print("   ", {k: v for k, v in _adj.LAST_ROUND.items() if k.startswith("cg") or k == "recurrence"}, flush=True)
    print(f"block {p} basis {basis or 'auto'} -> internal {s.internal_basis}: {min(ts[1:]):.3f} s (first {ts[0]:.3f}), sweeps {s.sweeps}, "
          f"restarts {s.n_restarts}, extras {s.n_extra}; solve_adjoint {1e3 * min(ta):.1f} ms, longest chain {steps}; max true residual / |theta| "
          f"{np.max(s.eig_res_true / np.abs(s.theta[s.indices[:32]])):.1e}", flush=True)
