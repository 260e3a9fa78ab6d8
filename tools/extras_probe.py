#!/usr/bin/env python3
"""development aid: C3 (32 modes) with fewer deflated extra pairs than the default 32 -- the short recurrence against the
Arnoldi form: time of solve_adjoint, which form ran, steps, residual histories of the slowest mode"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eigd_amd as eg  # noqa: E402
from eigd_amd import adjoint as adj  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

warnings.simplefilter("ignore")
col = BucklingColumn(706, 706, seed=0)
K = col.stiffness()
u = col.full_vector(eg.SpLuOperator(K, check_symmetry=False, coords=col.dof_coords())(col.f[col.reduced]))
G = col.geometric_stiffness(u)
sigma, N = 1.0971, 32
fac = eg.SpLuOperator((K + sigma * G).tocsr(), coords=col.dof_coords(), check_symmetry=False)
Phib = np.random.default_rng(1).uniform(size=(K.shape[0], N))
dPhib = fac.ctx.from_host(Phib)
for extra in [int(v) for v in (sys.argv[1:] or ["0", "8", "16", "24", "32"])]:
    eg.tuning.iram_extra = extra
    s = eg.IRAM(N=N, m=65, mode="buckling")
    s.solve(G, K, fac, sigma)
    ref = None
    for form in ("arnoldi", "auto", "auto period 1"):
        eg.tuning.recurrence = form.split()[0]
        eg.tuning.cg_projection_period = 1 if "period 1" in form else 0
        eg.tuning.cg_project_extra_pairs = "all pairs" in form
        eg.tuning.cg_projection_tol = 1e-13 if "1e-13" in form else (1e-300 if "tol 0" in form else 1e-11)
        hist = []
        ts = []
        for rep in range(2):
            hist.clear()
            fac.ctx.sync()
            t0 = time.perf_counter()
            dpsi, data = s.solve_adjoint(dPhib, method="sibk", rtol=1e-10, update_guess=False, bs_target=1, callback=hist.append)
            fac.ctx.sync()
            ts.append(time.perf_counter() - t0)
        psi = dpsi.get()
        if ref is None:
            ref = psi
        print(f"extras {s.n_extra:2d} {form:30s}: {1e3 * min(ts):7.1f} ms, ran '{adj.LAST_ROUND['recurrence']}', steps {sum(s.last_info)} "
              f"(longest {max(s.last_info)}), cg steps {adj.LAST_ROUND.get('cg_steps')}, psi vs arnoldi {np.linalg.norm(psi - ref) / np.linalg.norm(ref):.1e}",
              flush=True)
        if form != "arnoldi" and adj.LAST_ROUND["recurrence"] != "short":
            print("    ", adj.LAST_ROUND.get("cg_exit"), flush=True)
    eg.tuning.recurrence = "auto"
