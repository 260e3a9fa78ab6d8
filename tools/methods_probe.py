#!/usr/bin/env python3
"""development aid: wall time of solve_adjoint at C3 (1 M dof, 32 modes) for every method of the reference's dispatcher"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eigd_amd as eg  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

warnings.simplefilter("ignore")
col = BucklingColumn(706, 706, seed=0)
K = col.stiffness()
u = col.full_vector(eg.SpLuOperator(K, check_symmetry=False, coords=col.dof_coords())(col.f[col.reduced]))
G = col.geometric_stiffness(u)
sigma, N = 1.0971, 32
fac = eg.SpLuOperator((K + sigma * G).tocsr(), coords=col.dof_coords(), check_symmetry=False)
s = eg.IRAM(N=N, m=65, mode="buckling")
s.solve(G, K, fac, sigma)
dPhib = fac.ctx.from_host(np.random.default_rng(1).uniform(size=(K.shape[0], N)))
ref = None
for method in sys.argv[1:] or ["sibk", "laa", "pcpg", "pgmres"]:
    kw = dict(update_guess=False, bs_target=1) if method == "sibk" else {}
    ts = []
    for _ in range(2):
        fac.ctx.sync()
        fac.count = 0
        t0 = time.perf_counter()
        dpsi, data = s.solve_adjoint(dPhib, method=method, rtol=1e-10, **kw)
        fac.ctx.sync()
        ts.append(time.perf_counter() - t0)
    psi = dpsi.get()
    res, _ = s.eval_adjoint_residual_norm(dPhib, dpsi, b_ortho=True)
    if ref is None:
        ref = psi
    print(f"{method:7s}: {1e3 * min(ts):9.1f} ms, factor applications {fac.count}, residual max {np.max(res):.1e}, "
          f"psi vs {sys.argv[1] if len(sys.argv) > 1 else 'sibk'} {np.linalg.norm(psi - ref) / np.linalg.norm(ref):.1e}", flush=True)
