#!/usr/bin/env python3
"""development aid: C3 (32 modes) solve_adjoint + nothing else, alternating two values of one tuning attribute on one box:
ab_tuning.py NAME VALUE_A VALUE_B [repetitions]   (values through eval)"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eigd_amd as eg  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

warnings.simplefilter("ignore")
name, va, vb = sys.argv[1], eval(sys.argv[2]), eval(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
col = BucklingColumn(706, 706, seed=0)
K = col.stiffness()
u = col.full_vector(eg.SpLuOperator(K, check_symmetry=False, coords=col.dof_coords())(col.f[col.reduced]))
G = col.geometric_stiffness(u)
sigma, N = 1.0971, 32
fac = eg.SpLuOperator((K + sigma * G).tocsr(), coords=col.dof_coords(), check_symmetry=False)
s = eg.IRAM(N=N, m=65, mode="buckling")
s.solve(G, K, fac, sigma)
dPhib = fac.ctx.from_host(np.random.default_rng(1).uniform(size=(K.shape[0], N)))
out = {}
for rep in range(reps + 1):
    for v in (va, vb):
        setattr(eg.tuning, name, v)
        ts = []
        for _ in range(3):
            fac.ctx.sync()
            t0 = time.perf_counter()
            dpsi, data = s.solve_adjoint(dPhib, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
            fac.ctx.sync()
            ts.append(1e3 * (time.perf_counter() - t0))
        if rep:
            out.setdefault(repr(v), []).append(min(ts))
        psi = dpsi.get()
        out.setdefault("psi" + repr(v), psi)
for v in (va, vb):
    print(f"{name} = {v!r}: solve_adjoint ms {np.round(out[repr(v)], 2)}  steps {sum(s.last_info)}")
print("psi rel diff", np.linalg.norm(out["psi" + repr(va)] - out["psi" + repr(vb)]) / np.linalg.norm(out["psi" + repr(va)]))
