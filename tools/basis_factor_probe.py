#!/usr/bin/env python3
"""development aid: eigensolve time against the internal basis size rule (tuning.iram_basis_factor) on C2, C4, C3 and a
C3 with 8 modes; third solve of each setting (pools warm)"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eigd_amd as eg  # noqa: E402
from eigd_amd.problems import BucklingColumn, FreePlate, ThermalPlate  # noqa: E402

warnings.simplefilter("ignore")
factors = [float(v) for v in (sys.argv[1:] or ["2.0", "2.5", "3.0"])]


def run(name, A, B, sigma, mode, N, m, coords):
    P = (A - sigma * B) if mode == "normal" else (B + sigma * A)
    fac = eg.SpLuOperator(P.tocsr(), coords=coords, check_symmetry=False)
    out = []
    for f in factors:
        eg.tuning.iram_basis_factor = f
        ts = []
        for rep in range(3):
            s = eg.IRAM(N=N, m=m, mode=mode)
            fac.ctx.sync()
            t0 = time.perf_counter()
            lam, Phi = s.solve(A, B, fac, sigma)
            fac.ctx.sync()
            ts.append(time.perf_counter() - t0)
            del lam, Phi
        out.append(f"factor {f}: basis {s.internal_basis}, {1e3 * min(ts[1:]):.0f} ms, sweeps {s.sweeps}, restarts {s.n_restarts}")
    print(f"{name} (n = {B.shape[0]}, N = {N}): " + "; ".join(out), flush=True)


pl = FreePlate(316, 316, seed=1)
run("C2", pl.stiffness(), pl.mass(), -10.0, "normal", 13, 60, pl.dof_coords())
th = ThermalPlate(706, epsilon=1e-8, rhoE=np.random.default_rng(0).uniform(0.3, 1.0, size=706 * 706))
run("C4", th.stiffness(), th.mass(), -0.1, "normal", 20, 90, th.dof_coords())
col = BucklingColumn(706, 706, seed=0)
K = col.stiffness()
u = col.full_vector(eg.SpLuOperator(K, check_symmetry=False, coords=col.dof_coords())(col.f[col.reduced]))
G = col.geometric_stiffness(u)
run("C3", G, K, 1.0971, "buckling", 32, 65, col.dof_coords())
run("C3/8", G, K, 1.0971, "buckling", 8, 60, col.dof_coords())
