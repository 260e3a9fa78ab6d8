"""
Host-side ordering / symbolic analysis (C++ in libeigd_hip.so, no GPU needed), checked by
replaying the multifrontal algorithm of csrc/factor.hip in numpy from the symbolic arrays.
The numpy replay is test code only.
"""
import numpy as np
import pytest
from scipy import sparse
from scipy.sparse.linalg import spsolve

from eigd_amd.device import Symbolic


def grid_matrix(nx, ny, dof=2, seed=0):
    """SPD matrix with the sparsity of a Q4 mesh with `dof` unknowns per node."""
    rng = np.random.default_rng(seed)
    nn = nx * ny
    idx = np.arange(nn).reshape(nx, ny)
    rows, cols = [], []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            a = idx[max(0, -dx): nx - max(0, dx), max(0, -dy): ny - max(0, dy)].ravel()
            b = idx[max(0, dx): nx - max(0, -dx), max(0, dy): ny - max(0, -dy)].ravel()
            rows.append(a)
            cols.append(b)
    r, c = np.concatenate(rows), np.concatenate(cols)
    P = sparse.coo_matrix((np.ones(len(r)), (r, c)), shape=(nn, nn)).tocsr()
    P = sparse.kron(P, np.ones((dof, dof))).tocsr()
    P.data = rng.uniform(-1.0, 1.0, size=P.nnz)
    A = P + P.T
    A = A + sparse.diags(np.abs(A).sum(axis=1).A1 + 1.0)
    A = A.tocsr()
    A.sort_indices()
    return A


def replay_factor_and_solve(sym, A, B):
    s = sym.sizes
    nf = s["nfronts"]
    ns, bs, parent = sym.array("f_ns"), sym.array("f_bs"), sym.array("f_parent")
    foff, voff, bptr = sym.array("f_foff"), sym.array("f_voff"), sym.array("f_bptr")
    rel, v_src = sym.array("rel"), sym.array("v_src")
    a_src, a_dst = sym.array("a_src"), sym.array("a_dst")
    F = np.zeros(s["front_doubles"])
    F[a_dst] = A.data[a_src]
    fronts = []
    for f in range(nf):
        d = ns[f] + bs[f]
        fronts.append(F[foff[f]: foff[f] + d * d].reshape(d, d).T)  # column-major view: M[i, j] = F[j*d + i]
    L11s, L21s = [], []
    for f in range(nf):  # postorder
        M = fronts[f]
        n1 = ns[f]
        M[:] = np.tril(M) + np.tril(M, -1).T
        L11 = np.linalg.cholesky(M[:n1, :n1])
        L21 = np.linalg.solve(L11, M[n1:, :n1].T).T
        U = M[n1:, n1:] - L21 @ L21.T
        L11s.append(L11)
        L21s.append(L21)
        p = parent[f]
        if p >= 0:
            r = rel[bptr[f]: bptr[f + 1]]
            Mp = fronts[p]
            Mp[np.ix_(r, r)] += np.tril(U) + np.tril(U, -1).T * 0  # lower triangle only, as the kernel does
    k = B.shape[1]
    V = np.zeros((s["sumd"], k))
    own = v_src >= 0
    V[own] = B[v_src[own]]
    for f in range(nf):  # forward
        n1 = ns[f]
        v = V[voff[f]: voff[f] + n1 + bs[f]]
        v[:n1] = np.linalg.solve(L11s[f], v[:n1])
        v[n1:] -= L21s[f] @ v[:n1]
        p = parent[f]
        if p >= 0:
            r = rel[bptr[f]: bptr[f + 1]]
            V[voff[p] + r] += v[n1:]
    for f in range(nf - 1, -1, -1):  # backward
        n1 = ns[f]
        v = V[voff[f]: voff[f] + n1 + bs[f]]
        p = parent[f]
        if p >= 0:
            r = rel[bptr[f]: bptr[f + 1]]
            v[n1:] = V[voff[p] + r]
        v[:n1] = np.linalg.solve(L11s[f].T, v[:n1] - L21s[f].T @ v[n1:])
    X = np.zeros_like(B)
    X[v_src[own]] = V[own]
    return X


@pytest.mark.parametrize("nx,ny,dof,leaf", [(13, 11, 2, 16), (24, 17, 1, 8), (9, 9, 3, 12), (40, 40, 2, 48)])
def test_symbolic_structures_reproduce_the_solution(nx, ny, dof, leaf):
    A = grid_matrix(nx, ny, dof)
    sym = Symbolic(A, leaf_size=leaf, panel_width=16)
    s = sym.sizes
    n = A.shape[0]
    perm, iperm = sym.array("perm"), sym.array("iperm")
    assert sorted(perm.tolist()) == list(range(n))
    assert np.array_equal(iperm[perm], np.arange(n))
    ns, bs, c0 = sym.array("f_ns"), sym.array("f_bs"), sym.array("f_c0")
    assert ns.sum() == n and np.array_equal(c0, np.concatenate([[0], np.cumsum(ns)[:-1]]))
    parent, level = sym.array("f_parent"), sym.array("f_level")
    assert np.all((parent > np.arange(s["nfronts"])) | (parent < 0))  # postorder numbering
    assert np.all(level[parent[parent >= 0]] > level[parent >= 0])
    assert s["nnzL"] == int(np.sum(ns.astype(np.int64) * (ns + 1) // 2 + ns.astype(np.int64) * bs))
    rng = np.random.default_rng(5)
    B = rng.normal(size=(n, 3))
    X = replay_factor_and_solve(sym, A, B)
    Xref = spsolve(A.tocsc(), B)
    assert np.linalg.norm(X - Xref) / np.linalg.norm(Xref) < 1e-12


def test_disconnected_and_tiny_matrices():
    A = sparse.block_diag([grid_matrix(6, 5, 2, seed=1), grid_matrix(4, 4, 1, seed=2), sparse.identity(3)]).tocsr()
    sym = Symbolic(A, leaf_size=8, panel_width=8)
    B = np.random.default_rng(1).normal(size=(A.shape[0], 2))
    X = replay_factor_and_solve(sym, A, B)
    assert np.linalg.norm(A @ X - B) / np.linalg.norm(B) < 1e-12
    one = sparse.csr_matrix(np.array([[2.0]]))
    sym1 = Symbolic(one)
    assert sym1.sizes["nfronts"] == 1 and sym1.sizes["nnzL"] == 1


def test_symbolic_rejects_bad_input():
    A = grid_matrix(5, 5, 1).tolil()
    A[3, 3] = 0.0
    A = A.tocsr()
    A.eliminate_zeros()
    with pytest.raises(ValueError):
        Symbolic(A)
    with pytest.raises(ValueError):
        Symbolic(grid_matrix(5, 5, 1), panel_width=100)


def test_fill_is_nested_dissection_like():
    """fill of the level-structure dissection on a 120 x 120 two-dof grid stays near n log n"""
    A = grid_matrix(120, 120, 2)
    sym = Symbolic(A)
    n = A.shape[0]
    s = sym.sizes
    assert s["nnzL"] < 12 * n * np.log2(n)
    assert s["nlevels"] < 40


def test_geometric_dissection_with_coordinates():
    """the optional coordinate hint gives straight separators: correct solution, less fill, shorter panel chain"""
    nx, ny = 60, 47
    A = grid_matrix(nx, ny, 2)
    xy = np.stack([np.repeat(np.arange(nx), ny), np.tile(np.arange(ny), nx)], axis=1).astype(float)
    coords = np.repeat(xy, 2, axis=0)
    geo = Symbolic(A, leaf_size=24, panel_width=16, coords=coords)
    alg = Symbolic(A, leaf_size=24, panel_width=16)
    B = np.random.default_rng(2).normal(size=(A.shape[0], 2))
    X = replay_factor_and_solve(geo, A, B)
    assert np.linalg.norm(A @ X - B) / np.linalg.norm(B) < 1e-12
    assert geo.sizes["nnzL"] <= alg.sizes["nnzL"]
    assert geo.sizes["maxns"] <= 2 * min(nx, ny) + 2      # the root separator is one mesh line
    with pytest.raises(ValueError):
        Symbolic(A, coords=np.zeros((A.shape[0], 4)))
