"""
CPU prototype: ONE block Krylov space for all adjoint systems instead of one Krylov space per mode.

The reference's sibk is a block method by design (bs_target > 1: the modes of a block share the space spanned by their
residuals and its images, every mode solved in it with its own shift, eigenvector_derivatives.py:1195-1321).  Here the block
is all N modes and the space is built a block at a time -- what a device wants: one N-column sweep and product per step --
by block Lanczos in the factor inner product (OP = P K F is self-adjoint there), each mode i solved by Galerkin projection
of (I - alpha_i OP) v = r_i on the space (block tridiagonal T: a banded solve per mode and step).  Counted: block steps until
EVERY mode meets the reference's stopping rule (1275) on its true Euclidean residual, against the steps of the slowest
mode in its own Krylov space (the lock-step solver's chain).
Run: python tools/block_krylov_probe.py [nx ny [extras]]
"""
import os
import sys

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from short_recurrence_probe import build  # noqa: E402


def main():
    nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (60, 120)
    NX = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    N = 32
    K, G, F, sigma, lam, PhiD = build(nx, ny, N, max(NX, 1))
    if NX == 0:
        PhiD, lam = PhiD[:, :N], lam[:N]
    n = K.shape[0]
    BPhiD = K @ PhiD
    rng = np.random.default_rng(1)
    Phib = rng.uniform(0.0, 1.0, size=(n, N))
    rnorm0 = np.sqrt(np.max(np.sum(Phib**2, axis=0)))
    tol = 1e-10 * rnorm0
    alpha = -(lam[:N] - sigma)

    def P(x):
        return x - BPhiD @ (PhiD.T @ x)

    def OP(z):
        return P(G @ z)

    Fs = lambda X: np.column_stack([F.solve(np.ascontiguousarray(X[:, c])) for c in range(X.shape[1])])  # noqa: E731
    R = P(-Phib)
    # ---- per-mode CG in the F inner product (the lock-step solver's arithmetic): steps per mode
    steps = []
    for i in range(N):
        r = R[:, i].copy()
        zr = F.solve(r)
        p, zp = r.copy(), zr.copy()
        rho = r @ zr
        for j in range(1, 200):
            Cp = p - alpha[i] * OP(zp)
            a = rho / (zp @ Cp)
            r = P(r - a * Cp)
            if np.linalg.norm(r) < tol:
                steps.append(j)
                break
            zr = F.solve(r)
            rho_new = r @ zr
            p = r + (rho_new / rho) * p
            zp = zr + (rho_new / rho) * zp
            rho = rho_new
    print(f"n = {n}, deflated pairs {PhiD.shape[1]}: per-mode CG steps {steps}\n  total {sum(steps)}, longest chain {max(steps)}")
    # ---- block Lanczos in the F inner product + Galerkin per mode
    Zt = Fs(R)
    Gm = R.T @ Zt
    Lc = np.linalg.cholesky(0.5 * (Gm + Gm.T))
    Q = [sla.solve_triangular(Lc, R.T, lower=True).T]
    Z = [sla.solve_triangular(Lc, Zt.T, lower=True).T]
    B1 = Lc.T                                              # R = Q_1 B_1
    Ad, Bd = [], []
    psi_ref = None
    for j in range(1, 60):
        V = OP(Z[-1])
        A = Z[-1].T @ V
        V = V - Q[-1] @ A
        if j > 1:
            V = V - Q[-2] @ Bd[-1].T
        for q in range(len(Q)):                            # (prototype: full re-orthogonalisation, see the count below)
            V = V - Q[q] @ (Z[q].T @ V)
        V = P(V)
        Zt = Fs(V)
        Gm = V.T @ Zt
        w, U = np.linalg.eigh(0.5 * (Gm + Gm.T))
        keep = w > 1e-24 * w.max()
        Bn = (U[:, keep] * np.sqrt(w[keep])).T            # V = Q_new Bn, Q_new has rank(keep) columns
        Qn = V @ (U[:, keep] / np.sqrt(w[keep]))
        Zn = Zt @ (U[:, keep] / np.sqrt(w[keep]))
        Ad.append(0.5 * (A + A.T))
        Bd.append(Bn)
        Q.append(Qn)
        Z.append(Zn)
        sizes = [q.shape[1] for q in Q[:-1]]
        off = np.concatenate([[0], np.cumsum(sizes)])
        m = off[-1]
        T = np.zeros((m, m))
        for q in range(j):
            T[off[q]:off[q + 1], off[q]:off[q + 1]] = Ad[q]
            if q + 1 < j:
                T[off[q + 1]:off[q + 2], off[q]:off[q + 1]] = Bd[q]
                T[off[q]:off[q + 1], off[q + 1]:off[q + 2]] = Bd[q].T
        rhs = np.zeros((m, N))
        rhs[:N] = B1
        res = np.zeros(N)
        Y = np.zeros((m, N))
        EG = Qn.T @ Qn                                     # Euclidean Gram of the residual block
        for i in range(N):
            Y[:, i] = np.linalg.solve(np.eye(m) - alpha[i] * T, rhs[:, i])
            c = alpha[i] * (Bn @ Y[off[j - 1]:off[j], i])
            res[i] = np.sqrt(max(c @ EG @ c, 0.0))
        done = int(np.count_nonzero(res < tol))
        print(f"block step {j:2d}: basis {m:4d} vectors, new block rank {Qn.shape[1]:2d}, modes below the tolerance {done:2d}, "
              f"worst residual / tol {res.max() / tol:.2e}")
        if done == N:
            psi = np.column_stack(Z[:-1]) @ Y
            # against the per-mode solution: direct dense check on a few modes through the true residual
            tr = [np.linalg.norm(P(R[:, i] - (np.column_stack(Q[:-1]) @ Y[:, i] - alpha[i] * OP(np.column_stack(Z[:-1]) @ Y[:, i]))))
                  for i in (0, N // 2, N - 1)]
            print(f"  converged in {j} block steps (sweeps of {N} columns) against a longest chain of {max(steps)}; "
                  f"true residuals / tol of modes 0, {N // 2}, {N - 1}: {[f'{t / tol:.2f}' for t in tr]}")
            break


if __name__ == "__main__":
    main()
