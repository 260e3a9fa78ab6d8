#!/usr/bin/env python3
"""development aid: how large the residual's components along the deflated requested pairs get between the periodic
projections of the short recurrence (C3, 32 modes, EXTRAS deflated extra pairs): per step, the largest
|phi_a^T r_b| |B phi_a| / |r_b| over the active columns b, and which (a, b) it is"""
import os
import sys
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
rows = []


def hook(prob, j, rv, lo, hi, projected):
    C = prob.Phi.tdot(rv)                                # N x (hi - lo)
    un = prob.BPhi.colnorms()
    rn = rv.colnorms()
    rel = np.abs(C) * un[:, None] / rn[None, :]
    a, b = np.unravel_index(np.argmax(rel), rel.shape)
    rows.append((j, lo, hi, projected, rel[a, b], a, lo + b, rn.min(), rn.max()))


adj._CG_TRACE_HOOK = hook
eg.tuning.cg_project_previous = os.environ.get("PREVIOUS", "1") != "0"   # 0: the residual alone (the scheme before the fix)
print("projection of the previous residual:", eg.tuning.cg_project_previous)
for extra in [int(v) for v in (sys.argv[1:] or ["16", "32"])]:
    eg.tuning.iram_extra = extra
    s = eg.IRAM(N=N, m=65, mode="buckling")
    s.solve(G, K, fac, sigma)
    print("lam", np.round(s.lam[[0, 1, 15, 30, 31]], 4))
    rows.clear()
    dpsi, data = s.solve_adjoint(dPhib, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
    print(f"extras {s.n_extra}: ran '{adj.LAST_ROUND['recurrence']}', period {adj.LAST_ROUND.get('cg_projection_period')}")
    for r in rows:
        print("   step %2d cols [%2d, %2d) %s  max rel component %.1e (pair %2d in column %2d)  |r| %.1e .. %.1e" % (
            r[0], r[1], r[2], "P" if r[3] else " ", r[4], r[5], r[6], r[7], r[8]))
