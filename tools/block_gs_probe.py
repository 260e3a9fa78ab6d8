#!/usr/bin/env python3
"""development aid: device time of the pieces of one block Lanczos step at the C3 size (no factor needed)"""
import os
import sys
import time

import numpy as np
from scipy import sparse

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigd_amd.device import CSRMatrix, DevicePanels, default_context  # noqa: E402

ctx = default_context()
n = 998284
rng = np.random.default_rng(0)


def timed(label, fn, reps=10):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    print(f"{label}: {1e3 * (time.perf_counter() - t0) / reps:.3f} ms", flush=True)


P = DevicePanels(ctx, 192, n)
for q in range(3):
    P.view(64 * q, 64 * q + 64).copy_from(ctx.from_host(rng.normal(size=(n, 64))))
for p in (1, 4, 8):
    X = ctx.from_host(rng.normal(size=(n, p)))
    for c in (64, 136):
        H = P.tdot_block(X, ns=c)
        timed(f"p={p} c={c} panels tdot_block", lambda: P.tdot_block(X, ns=c))
        timed(f"p={p} c={c} panels times_into", lambda: P.times_into(X, 1e-3 * H, ns=c, alpha=-1.0, beta=1.0))
    timed(f"p={p} get_block", lambda: P.get_block(100, p))
    timed(f"p={p} set_block", lambda: P.set_block(100, X))
    timed(f"p={p} X.tdot(X)", lambda: X.tdot(X))
    timed(f"p={p} add_product p x p", lambda: ctx.empty(n, p).add_product(X, np.eye(p), alpha=1.0, beta=0.0))
st = ctx.stack(136, n, 1)
X1 = ctx.from_host(rng.normal(size=(n, 1)))
timed("k=1 stack dot 136", lambda: st.dot(X1, ns=136))
timed("k=1 stack axpy 136", lambda: st.axpy_into(X1, np.full((136, 1), 1e-6), alpha=-1.0))
X8 = ctx.from_host(rng.normal(size=(n, 8)))
timed("k=1 stack tdot_block 136 x 8", lambda: st.tdot_block(X8, ns=136))
timed("k=1 stack times_into 136 x 8", lambda: st.times_into(X8, np.full((136, 8), 1e-6), ns=136, alpha=-1.0, beta=1.0))
# SpMM of a Q4-like matrix (18 nnz / row)
nn = 1000
ii = np.arange(n)
offs = [-nn - 1, -nn, -nn + 1, -1, 0, 1, nn - 1, nn, nn + 1]
A = sparse.diags([np.ones(n - abs(o)) for o in offs], offs, shape=(n, n), format="csr")
dA = CSRMatrix(ctx, A)
for p in (1, 8, 32):
    Xp = ctx.from_host(rng.normal(size=(n, p)))
    Yp = ctx.empty(n, p)
    timed(f"SpMM 9/row p={p}", lambda: dA.apply(Xp, Yp))
