"""
The host algebra of the two-step Krylov cycle (eigd_amd.adjoint.pair_arnoldi_columns) without a GPU: a cycle's two
Arnoldi columns and the coefficient transform of the solution update are rebuilt from what the device hands back, and
must reproduce a plain one-step Arnoldi process (reference loop eigenvector_derivatives.py:1246-1277) on the same
operator: same Hessenberg matrix, same basis, same  Z y.
"""
import numpy as np

from eigd_amd.adjoint import pair_arnoldi_columns, solve_shifted_lstsq


def _arnoldi(op, fac, w0, m):
    n = w0.shape[0]
    W = np.zeros((n, m + 1)); Z = np.zeros((n, m)); H = np.zeros((m + 1, m))
    W[:, 0] = w0 / np.linalg.norm(w0)
    for j in range(m):
        Z[:, j] = fac(W[:, j])
        t = op(Z[:, j])
        for _ in range(2):
            h = W[:, : j + 1].T @ t
            t = t - W[:, : j + 1] @ h
            H[: j + 1, j] += h
        H[j + 1, j] = np.linalg.norm(t)
        W[:, j + 1] = t / H[j + 1, j]
    return W, Z, H


def test_two_step_cycle_rebuilds_the_arnoldi_relation():
    rng = np.random.default_rng(3)
    n, m = 300, 12
    A = rng.normal(size=(n, n)); A = A @ A.T / n + np.eye(n)             # factor = A^-1 (SPD), K = a second matrix
    Kmat = rng.normal(size=(n, n)) / np.sqrt(n)
    Ainv = np.linalg.inv(A)
    fac = lambda x: Ainv @ x
    op = lambda z: Kmat @ z
    w0 = rng.normal(size=n)
    W1, Z1, H1 = _arnoldi(op, fac, w0, m)
    # two-step cycles with the quantities the device returns
    W = np.zeros((n, m + 1)); Zs = np.zeros((n, m)); H = np.zeros((m + 2, m + 1)); Cz = np.zeros((m + 1, m + 1))
    W[:, 0] = w0 / np.linalg.norm(w0)
    for j in range(0, m, 2):
        Zs[:, j] = fac(W[:, j]); v1 = op(Zs[:, j])
        Zs[:, j + 1] = fac(v1); v2 = op(Zs[:, j + 1])
        Wj = W[:, : j + 1]
        h1, g1 = Wj.T @ v1, Wj.T @ v2
        v1p, v2p = v1 - Wj @ h1, v2 - Wj @ g1
        d1, d2 = Wj.T @ v1p, Wj.T @ v2p                                  # the measured second pass, applied here
        v1p, v2p, h1, g1 = v1p - Wj @ d1, v2p - Wj @ d2, h1 + d1, g1 + d2
        b1 = np.linalg.norm(v1p); gamma = v1p @ v2p
        w1 = v1p / b1
        v2pp = v2p - (gamma / b1**2) * v1p
        b2 = np.linalg.norm(v2pp)
        W[:, j + 1], W[:, j + 2 if j + 2 <= m else m] = w1, (v2pp / b2 if j + 2 <= m else W[:, m])
        pair_arnoldi_columns(H, Cz, j, h1, g1, b1, gamma, b2)
    assert np.abs(np.abs(W1.T @ W[:, : m + 1]) - np.eye(m + 1)).max() < 1e-10        # same basis (up to rounding)
    assert np.abs(H[: m + 1, :m] - H1).max() < 1e-9 * np.abs(H1).max()               # same Hessenberg matrix
    Zb = Zs @ Cz[:m, :m]                                                              # factor(w_i) from the stored slabs
    assert np.abs(Zb - Z1).max() < 1e-9 * np.abs(Z1).max()
    r = np.zeros(m + 1); r[0] = 1.0
    y1, res1 = solve_shifted_lstsq(0.37, H1, r)
    y2, res2 = solve_shifted_lstsq(0.37, H[: m + 1, :m], r)
    assert abs(res1 - res2) < 1e-10 and np.abs(Z1 @ y1 - Zs @ (Cz[:m, :m] @ y2)).max() < 1e-9 * np.abs(Z1 @ y1).max()


def test_shifted_least_squares_matches_lstsq_and_handles_rank_deficiency():
    rng = np.random.default_rng(5)
    for m, n in ((2, 1), (9, 8), (37, 36)):
        H = np.triu(rng.normal(size=(m, n)), -1)
        r = np.zeros(m); r[0] = 2.5
        y, res = solve_shifted_lstsq(-0.4, H, r)
        H0 = np.eye(m, n) + 0.4 * H
        y0 = np.linalg.lstsq(H0, r, rcond=None)[0]
        assert np.abs(y - y0).max() < 1e-12 * np.abs(y0).max() and abs(res - np.linalg.norm(H0 @ y0 - r)) < 1e-12
    H = np.zeros((5, 4)); H[0, 0] = 1.0 / 0.3                                        # I - 0.3 H has a zero column
    r = rng.normal(size=5)
    y, res = solve_shifted_lstsq(0.3, H, r)
    H0 = np.eye(5, 4) - 0.3 * H
    assert abs(res - np.linalg.norm(H0 @ np.linalg.lstsq(H0, r, rcond=None)[0] - r)) < 1e-12
