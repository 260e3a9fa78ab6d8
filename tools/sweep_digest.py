#!/usr/bin/env python3
"""
SHA-256 of sweep results over grid matrices, widths 1..32, Cholesky and Bunch-Kaufman factors, before and after a numeric
refactorisation -- bitwise comparison of two builds of the library (EIGD_LIB selects the shared object):
    python tools/sweep_digest.py ; EIGD_LIB=build/r3/libeigd_hip.so python tools/sweep_digest.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_symbolic_cpu import grid_matrix  # noqa: E402
from eigd_amd.device import Factor, default_context  # noqa: E402

ctx = default_context()
rng = np.random.default_rng(0)
h = hashlib.sha256()
for (nx, ny, dof, seed, leaf, shift) in ((150, 140, 2, 4, 0, 0.0), (90, 95, 3, 5, 50, 0.0), (201, 77, 1, 6, 37, 0.0),
                                         (120, 110, 2, 7, 0, 8.0)):
    A = grid_matrix(nx, ny, dof, seed)
    if shift:  # interior shift: the Bunch-Kaufman path
        import scipy.sparse as sp

        A = (A - shift * sp.identity(A.shape[0])).tocsr()
    F = Factor(ctx, A, leaf_size=leaf)
    for rep in range(2):
        for k in (1, 4, 7, 9, 16, 21, 32):
            B = rng.normal(size=(A.shape[0], k))
            X = F.solve_inplace(ctx.from_host(B)).get()
            r = np.linalg.norm(A @ X - B) / np.linalg.norm(B)
            assert r < (1e-7 if shift else 1e-11), (nx, k, r)
            h.update(X.tobytes())
        A = (A + 0.25 * grid_matrix(nx, ny, dof, seed + 10)).tocsr()   # same pattern, new values
        F.refactor(A)
    print(nx, ny, dof, "negative pivots", F.stats()["negative_pivots"], "static", F.stats()["static_pivots"], h.hexdigest()[:16], flush=True)
print("digest", h.hexdigest())
