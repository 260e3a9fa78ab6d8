#!/usr/bin/env python3
"""development aid: the two forms of sibk (short recurrence / Arnoldi) on the other configurations -- C2 natural
frequency 200 978 dof N = 13, C4 thermal 499 849 dof N = 20 (generalized), a 1 M-dof column with 8 modes -- time of
solve_adjoint, total and longest step counts, which form ran"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eigd_amd as eg  # noqa: E402
from eigd_amd import adjoint as adj  # noqa: E402
from eigd_amd.problems import BucklingColumn, FreePlate, ThermalPlate  # noqa: E402

warnings.simplefilter("ignore")
rng = np.random.default_rng(1)


def run(name, A, B, sigma, mode, N, m, coords, zero_first=0):
    P = (A - sigma * B) if mode == "normal" else (B + sigma * A)
    fac = eg.SpLuOperator(P.tocsr(), coords=coords, check_symmetry=False)
    s = eg.IRAM(N=N, m=m, mode=mode)
    t0 = time.perf_counter()
    s.solve(A, B, fac, sigma)
    te = time.perf_counter() - t0
    Phib = rng.uniform(size=(B.shape[0], N))
    Phib[:, :zero_first] = 0.0
    dPhib = fac.ctx.from_host(Phib)
    out = []
    for form in ("auto", "arnoldi"):
        eg.tuning.recurrence = form
        ts = []
        for rep in range(3):
            fac.ctx.sync()
            t0 = time.perf_counter()
            dpsi, data = s.solve_adjoint(dPhib, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
            fac.ctx.sync()
            ts.append(time.perf_counter() - t0)
        out.append((form, adj.LAST_ROUND["recurrence"], min(ts), sum(s.last_info), max(s.last_info), dpsi.get()))
    eg.tuning.recurrence = "auto"
    d = np.linalg.norm(out[0][5] - out[1][5]) / np.linalg.norm(out[1][5])
    print(f"{name}: n = {B.shape[0]}, N = {N}, extras {s.n_extra}, block {s.block_size}, eigensolve {te:.2f} s", flush=True)
    for form, ran, t, tot, mx, _ in out:
        print(f"    {form:8s} ran {ran:50s} {1e3 * t:8.1f} ms  steps {tot} (longest {mx})", flush=True)
    print(f"    psi rel diff {d:.1e}", flush=True)


which = sys.argv[1:] or ["c2", "c4", "col8"]
if "c2" in which:
    pl = FreePlate(316, 316, seed=1)
    run("C2 natural frequency", pl.stiffness(), pl.mass(), -10.0, "normal", 13, 60, pl.dof_coords(), zero_first=3)
if "c4" in which:
    th = ThermalPlate(706, epsilon=1e-8, rhoE=np.random.default_rng(0).uniform(0.3, 1.0, size=706 * 706))
    run("C4 thermal", th.stiffness(), th.mass(), -0.1, "normal", 20, 90, th.dof_coords())
if "col8" in which:
    col = BucklingColumn(706, 706, seed=0)
    K = col.stiffness()
    u = col.full_vector(eg.SpLuOperator(K, check_symmetry=False, coords=col.dof_coords())(col.f[col.reduced]))
    G = col.geometric_stiffness(u)
    run("C3 column, 8 modes", G, K, 1.0971, "buckling", 8, 60, col.dof_coords())
