#!/usr/bin/env python3
"""Development aid: Bunch-Kaufman factor on many interior shifts of several pencils -- worst residuals with and without the refinement step."""
import os, sys, warnings
import numpy as np
from scipy import sparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_symbolic_cpu import grid_matrix
import eigd_amd as eg
from eigd_amd._ffi import NotPositiveDefiniteError
from eigd_amd.device import default_context
ctx = default_context()
rng = np.random.default_rng(0)
worst_f = worst_r = 0.0
nfail = ntot = 0
for (nx, ny, dof, seed) in ((40, 37, 1, 1), (25, 24, 2, 2), (60, 50, 2, 3), (90, 80, 1, 4)):
    K = grid_matrix(nx, ny, dof, seed)
    n = K.shape[0]
    M = sparse.diags(rng.uniform(0.5, 1.5, size=n)).tocsr()
    dmin, dmax = (K.diagonal() / M.diagonal()).min(), (K.diagonal() / M.diagonal()).max()
    B = rng.normal(size=(n, 4))
    for sigma in rng.uniform(0.0, 1.2 * dmax, size=25):
        mat = (K - sigma * M).tocsr()
        ntot += 1
        try:
            op = eg.SpLuOperator(mat.tocsc(), ctx=ctx)
        except NotPositiveDefiniteError:
            nfail += 1
            continue
        Xf = op.factor.solve_inplace(ctx.from_host(B)).get()
        X = op(B)
        rf = np.linalg.norm(mat @ Xf - B) / np.linalg.norm(B)
        rr = np.linalg.norm(mat @ X - B) / np.linalg.norm(B)
        worst_f, worst_r = max(worst_f, rf), max(worst_r, rr)
        if rf > 1e-6:
            print(f"n={n} sigma={sigma:.4f} negative pivots {op.negative_pivots}: factor residual {rf:.1e}, refined {rr:.1e}")
print(f"{ntot} shifts, {nfail} refused; worst residual factor-only {worst_f:.1e}, with refinement {worst_r:.1e}")
