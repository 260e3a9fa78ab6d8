"""
Host logic of the restarted (block) Lanczos eigensolver -- eigd_amd.lanczos.thick_restart_block_lanczos and
compress_to_single_vector_basis -- on a numpy stand-in for the device backend (same interface as
_BlockLanczosDevice: expand / restart on a B-orthonormal basis).  Checked: eigenvalues against scipy's eigsh, and the
contract the reference's IRAM hands to laa (eigd/arpack.py:58-101): a B-orthonormal basis V of exactly m
vectors and a symmetric T with OP V = V T + f e_m^T.
"""
import numpy as np
import pytest
from scipy import sparse
from scipy.sparse.linalg import eigsh, splu

from eigd_amd.lanczos import (_svqb, arpack_converged, compress_to_single_vector_basis, ritz_bounds,
                              thick_restart_block_lanczos)


class NumpyBackend:
    def __init__(self, K, M, sigma, nvec):
        self.B = M.tocsr()
        self.lu = splu((K - sigma * M).tocsc())
        self.V = np.zeros((K.shape[0], nvec))
        self.applications = 0

    def _orthonormalise(self, X):
        C = np.eye(X.shape[1])
        for _ in range(2):
            G = X.T @ (self.B @ X)
            Tr, Cq, bad = _svqb(0.5 * (G + G.T))
            assert not bad.any()
            X = X @ Tr
            C = Cq @ C
        return X, C

    def start(self, V0):
        X, _ = self._orthonormalise(V0)
        self.V[:, : X.shape[1]] = X

    def expand(self, c, p):
        W = self.lu.solve(self.B @ self.V[:, c - p:c])
        self.applications += 1
        Q = self.V[:, :c]
        H = Q.T @ (self.B @ W)
        W = W - Q @ H
        H2 = Q.T @ (self.B @ W)
        W = W - Q @ H2
        X, C = self._orthonormalise(W)
        self.V[:, c:c + p] = X
        return H + H2, C

    def restart(self, S, c, keep, p):
        Y = self.V[:, :c] @ S
        tail = self.V[:, c:c + p].copy()
        self.V[:, :keep] = Y
        self.V[:, keep:keep + p] = tail


class DeferredNumpyBackend(NumpyBackend):
    """the same backend with the device's two-phase protocol: a cycle's steps are enqueued, their coefficients are
    collected at the end of the cycle"""

    def __init__(self, *a):
        super().__init__(*a)
        self.outstanding = 0
        self.most_outstanding = 0

    def expand_begin(self, c, p):
        self.outstanding += 1
        self.most_outstanding = max(self.most_outstanding, self.outstanding)
        return NumpyBackend.expand(self, c, p)

    def expand_end(self, token):
        self.outstanding -= 1
        return token

    def expand(self, c, p):
        return self.expand_end(self.expand_begin(c, p))


def laplace(nside, seed=0):
    I = sparse.identity(nside)
    T = sparse.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(nside, nside))
    K = (sparse.kron(I, T) + sparse.kron(T, I)).tocsr()
    M = sparse.diags(np.random.default_rng(seed).uniform(0.5, 1.5, size=nside * nside)).tocsr()
    return K, M


@pytest.mark.parametrize("p,k_want,m,m_int", [(1, 10, 30, 30), (1, 12, 25, 25), (2, 12, 30, 40), (4, 14, 33, 48), (8, 12, 30, 64)])
def test_restarted_block_lanczos_contract(p, k_want, m, m_int):
    K, M = laplace(30)
    n, sigma = K.shape[0], -0.1
    be = NumpyBackend(K, M, sigma, m_int + p)
    be.start(np.random.default_rng(12345).uniform(-1, 1, size=(n, p)))
    eps = np.finfo(float).eps
    T, C, c, nconv, nrest = thick_restart_block_lanczos(be, k_want, m_int, p, eps, 500)
    assert nconv >= k_want and c <= m_int
    # the block relation before the compression: OP V = V T + Q_res C E_last^T
    V = be.V[:, :c]
    OPV = be.lu.solve(be.B @ V)
    Rm = OPV - V @ T
    Rm[:, c - p:] -= be.V[:, c:c + p] @ C
    assert np.linalg.norm(Rm) < 1e-11 * np.linalg.norm(OPV)
    if not (p == 1 and c == m):
        T, beta_m = compress_to_single_vector_basis(be, T, C, c, p, m, eps)
    else:
        beta_m = float(C[0, 0])
    V = be.V[:, :m]
    assert T.shape == (m, m) and np.allclose(T, T.T)
    assert np.linalg.norm(V.T @ (be.B @ V) - np.eye(m)) < 1e-11
    OPV = be.lu.solve(be.B @ V)
    Rm = OPV - V @ T
    assert np.linalg.norm(Rm[:, :-1]) < 1e-11 * np.linalg.norm(OPV)          # OP V = V T + f e_m^T
    f = Rm[:, -1]
    assert abs(np.sqrt(f @ (be.B @ f)) - abs(beta_m)) < 1e-10 * max(abs(beta_m), 1e-30) + 1e-13
    assert np.linalg.norm(V.T @ (be.B @ f)) < 1e-10 * max(np.linalg.norm(f), 1e-30) + 1e-13   # f is B-orthogonal to V
    # eigenvalues: the k_want wanted pairs are in T, converged
    theta, Y = np.linalg.eigh(T)
    lam = np.sort(1.0 / theta + sigma)
    ref = eigsh(K, k=k_want, M=M, sigma=sigma, which="LM", return_eigenvectors=False)
    assert np.allclose(lam[:k_want], np.sort(ref), rtol=1e-10, atol=0)
    bounds = np.abs(beta_m * Y[m - 1, :])
    order = np.argsort(-np.abs(theta))[:k_want]
    assert arpack_converged(bounds[order], theta[order], eps, abs(beta_m)).all()


def test_deferred_protocol_gives_the_same_projected_matrix():
    """a backend with expand_begin / expand_end (the device's) is driven cycle by cycle: same T, same restarts"""
    K, M = laplace(30)
    n, sigma, p, k_want, m_int = K.shape[0], -0.1, 4, 14, 48
    out = []
    for cls in (NumpyBackend, DeferredNumpyBackend):
        be = cls(K, M, sigma, m_int + p)
        be.start(np.random.default_rng(12345).uniform(-1, 1, size=(n, p)))
        out.append((be,) + thick_restart_block_lanczos(be, k_want, m_int, p, np.finfo(float).eps, 500))
    (b0, T0, C0, c0, nconv0, nr0), (b1, T1, C1, c1, nconv1, nr1) = out
    assert (c0, nconv0, nr0, b0.applications) == (c1, nconv1, nr1, b1.applications)
    assert np.array_equal(T0, T1) and np.array_equal(C0, C1)
    assert b1.outstanding == 0 and b1.most_outstanding == m_int // p        # the first cycle: all its steps in flight


def test_no_convergence_is_reported_not_hidden():
    K, M = laplace(30)
    be = NumpyBackend(K, M, -0.1, 26)
    be.start(np.random.default_rng(1).uniform(-1, 1, size=(K.shape[0], 1)))
    T, C, c, nconv, nrest = thick_restart_block_lanczos(be, 12, 25, 1, np.finfo(float).eps, 0)
    assert nconv < 12 and nrest == 0 and c == 25


def test_convergence_floor_scales_with_the_value_itself():
    """a small Ritz value far from the shift is not declared converged on the strength of the largest one's size"""
    eps = np.finfo(float).eps
    theta = np.array([1.0e4, 1.0e-2])
    bounds = np.array([1e-13, 1e-13])
    ok = arpack_converged(bounds, theta, eps, beta_scale=1e-3)
    assert ok[0] and not ok[1]            # 1e-13 is noise next to 1e4, but 1e-11 relative for the value 1e-2
    assert ritz_bounds(np.array([[2.0]]), np.array([[0.0, 1.0], [0.5, 0.0]]), 1).tolist() == [1.0, 0.0]
