#!/usr/bin/env python3
"""development aid: the C3 eigensolve for several block sizes of the restarted block Lanczos (tuning.iram_block) and basis
factors (tuning.iram_basis_factor), alternating on one box: time, sweeps, restarts, eigenvalue difference, true residuals.
BLOCKS=8,16 FACTORS=2.5 python tools/eig_block_probe.py"""
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
blocks = [int(v) for v in os.environ.get("BLOCKS", "8,16").split(",")]
factors = [float(v) for v in os.environ.get("FACTORS", "2.5").split(",")]
ref = None
for rep in range(3):
    for b in blocks:
        for fct in factors:
            eg.tuning.iram_block, eg.tuning.iram_basis_factor = b, fct
            lam = Phi = s = None
            s = eg.IRAM(N=32, m=65, mode="buckling", ctx=ctx)
            ctx.sync()
            t0 = time.perf_counter()
            try:
                lam, Phi = s.solve(dG, dK, fac, sigma)
            except Exception as exc:
                print(f"block {b} factor {fct}: {type(exc).__name__}: {exc}", flush=True)
                continue
            ctx.sync()
            t = time.perf_counter() - t0
            if ref is None:
                ref = lam.copy()
            if rep:
                print(f"block {b:2d} basis factor {fct}: solve {t:.3f} s, sweeps {s.sweeps}, restarts {s.n_restarts}, basis {s.internal_basis}, "
                      f"extras {s.n_extra}, max |dlam|/|lam| {np.max(np.abs(lam - ref) / np.abs(ref)):.1e}, "
                      f"eig_res_true max {np.max(s.eig_res_true):.1e}", flush=True)
