#!/usr/bin/env python3
"""Kernel timing probe at configuration sizes (development aid; writes plain text)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_symbolic_cpu import grid_matrix  # noqa: E402

from eigd_amd.device import CSRMatrix, Factor, Symbolic, default_context  # noqa: E402


def timeit(ctx, fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop_ms() / reps


def main(nx=707):
    ctx = default_context()
    t0 = time.time()
    A = grid_matrix(nx, nx, 2, seed=1)
    n = A.shape[0]
    print(f"n={n} nnz={A.nnz} build {time.time()-t0:.1f}s", flush=True)
    dA = CSRMatrix(ctx, A)
    rng = np.random.default_rng(0)
    x = ctx.from_host(rng.normal(size=n))
    y = ctx.empty(n, 1)
    ms = timeit(ctx, lambda: dA.apply(x, y), reps=50)
    print(f"spmv: {ms*1e3:.1f} us  {dA.spmv_bytes(1)/ms/1e6:.1f} GB/s algorithmic", flush=True)
    for k in (4, 8, 32):
        X = ctx.from_host(rng.normal(size=(n, k)))
        Y = ctx.empty(n, k)
        ms = timeit(ctx, lambda: dA.apply(X, Y), reps=20)
        print(f"spmm k={k}: {ms*1e3:.1f} us  {dA.spmv_bytes(k)/ms/1e6:.1f} GB/s", flush=True)
    X = ctx.from_host(rng.normal(size=(n, 32)))
    Y = ctx.from_host(rng.normal(size=(n, 32)))
    ms = timeit(ctx, lambda: X.coldot(Y))
    print(f"coldot k=32: {ms*1e3:.1f} us {2*8*n*32/ms/1e6:.1f} GB/s", flush=True)
    ms = timeit(ctx, lambda: X.tdot(Y))
    print(f"gemm_tn 32x32: {ms*1e3:.1f} us {2*8*n*32/ms/1e6:.1f} GB/s", flush=True)
    ms = timeit(ctx, lambda: Y.project(X, X))
    print(f"project 32/32: {ms*1e3:.1f} us {(4*8*n*32+8*n*32)/ms/1e6:.1f} GB/s", flush=True)
    st = ctx.stack(20, n, 32)
    for j in range(20):
        st[j].copy_from(X)
    ms = timeit(ctx, lambda: st.dot(Y, ns=20), reps=5)
    print(f"stack_dot ns=20 k=32: {ms*1e3:.1f} us {21*8*n*32/ms/1e6:.1f} GB/s", flush=True)
    H = np.ones((20, 32)) * 1e-3
    ms = timeit(ctx, lambda: st.axpy_into(Y, H), reps=5)
    print(f"stack_axpy ns=20 k=32: {ms*1e3:.1f} us {22*8*n*32/ms/1e6:.1f} GB/s", flush=True)
    del st
    t0 = time.time()
    sym = Symbolic(A)
    print(f"symbolic {time.time()-t0:.2f}s {sym.sizes}", flush=True)
    t0 = time.time()
    F = Factor(ctx, A, symbolic=sym)
    ctx.sync()
    print(f"factor {time.time()-t0:.2f}s {F.stats()}", flush=True)
    t0 = time.time()
    F.refactor(A)
    ctx.sync()
    print(f"refactor {time.time()-t0:.2f}s", flush=True)
    for k in (1, 4, 8, 16, 32):
        B = rng.normal(size=(n, k))
        dB = ctx.from_host(B)
        ms = timeit(ctx, lambda: F.solve_inplace(dB), reps=5, warm=1)
        dB.set(B)
        Xs = F.solve_inplace(dB).get()
        res = np.linalg.norm(A @ Xs - B) / np.linalg.norm(B)
        print(f"solve k={k}: {ms:.3f} ms  {F.solve_bytes(k)/ms/1e6:.1f} GB/s  residual {res:.2e}", flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 707)
