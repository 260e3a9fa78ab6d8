#!/usr/bin/env python3
"""development aid: how well does a mode's residual history predict that it finishes within the next two Krylov steps
(C3 problem; the lock-step solver uses the prediction to decide what the next cycle has to carry)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eigd_amd as eg  # noqa: E402
from eigd_amd.device import CSRMatrix, default_context  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

ctx = default_context()
nx = int(os.environ.get("NX", "706"))
N = int(os.environ.get("MODES", "32"))
col = BucklingColumn(nx, nx, seed=0)
K = col.stiffness()
coords = col.dof_coords()
Kfac = eg.SpLuOperator(K, ctx=ctx, check_symmetry=False, coords=coords)
u = col.full_vector(Kfac(col.f[col.reduced]))
G = col.geometric_stiffness(u)
sigma = float(os.environ.get("SIGMA", "1.0971"))
fac = eg.SpLuOperator((K + sigma * G).tocsr(), ctx=ctx, symbolic=Kfac.symbolic, check_symmetry=False, coords=coords)
dK, dG = CSRMatrix(ctx, K), CSRMatrix(ctx, G)
s = eg.IRAM(N=N, m=2 * N + 1, mode="buckling", ctx=ctx)
s.solve(dG, dK, fac, sigma)
Phib = np.random.default_rng(1).uniform(size=(K.shape[0], N))
seq = []
psi, data = s.solve_adjoint(ctx.from_host(Phib), method="sibk", rtol=1e-10, callback=seq.append)
its = [int(i) for i in s.last_info]
rn0 = float(np.sqrt(np.max(np.sum(Phib**2, axis=0))))
tol = 1e-10 * rn0
hist, p = [], 0
for i in its:
    hist.append(np.array(seq[p:p + i + 1]))
    p += i + 1
assert p == len(seq), (p, len(seq))
print("iterations", its)
ratios = []
for alpha in (0.1, 0.25, 0.5, 1.0, 4.0):
    hit = miss = lost = 0
    for h in hist:
        for i in range(2, len(h), 2):                       # after cycle i/2: steps 1..i solved, i+1 and i+2 in flight
            pred = h[i] * min(h[i] / h[i - 2], 1.0)
            finishes = len(h) - 1 <= i + 2                 # the mode's last step is i+1 or i+2
            if i + 2 < len(h):
                ratios.append(h[i + 2] / pred)
            if pred < alpha * tol:
                hit += finishes
                miss += not finishes
            else:
                lost += finishes
    print(f"alpha {alpha}: predicted and finished {hit}, predicted but not finished {miss}, finished unpredicted {lost}")
r = np.array(ratios)
print("actual / predicted residual two steps on: percentiles 5 25 50 75 95:", np.percentile(r, [5, 25, 50, 75, 95]).round(3))
