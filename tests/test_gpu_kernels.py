"""
GPU parity of every C-ABI kernel against numpy / scipy on seeded inputs.
Bit-exact where the arithmetic order is the reference's (SpMV), tight tolerances elsewhere.
"""
import warnings

import numpy as np
import pytest
from scipy import sparse
from scipy.sparse.linalg import splu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from eigd_amd.device import default_context

    return default_context()


def grid_matrix(nx, ny, dof=2, seed=0):
    from test_symbolic_cpu import grid_matrix as g

    return g(nx, ny, dof, seed)


def relerr(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("nx,dof", [(37, 2), (120, 1), (200, 2)])
def test_spmv_bit_exact_vs_scipy(ctx, nx, dof):
    from eigd_amd.device import CSRMatrix

    A = grid_matrix(nx, nx + 3, dof, seed=nx)
    rng = np.random.default_rng(1)
    x = rng.normal(size=A.shape[0])
    dA = CSRMatrix(ctx, A)
    y = dA.apply(ctx.from_host(x)).get()[:, 0]
    assert np.array_equal(y, A @ x)  # same summation order, separately rounded mul/add
    # alpha / beta form
    y0 = rng.normal(size=A.shape[0])
    Y = ctx.from_host(y0)
    dA.apply(ctx.from_host(x), Y, alpha=-0.5, beta=2.0)
    assert relerr(Y.get()[:, 0], -0.5 * (A @ x) + 2.0 * y0) < 1e-15


def test_spmv_long_rows_and_empty_rows(ctx):
    from eigd_amd.device import CSRMatrix

    rng = np.random.default_rng(2)
    n = 5000
    A = sparse.random(n, n, density=0.002, random_state=3, format="lil")
    A[7, :] = rng.normal(size=n)          # one dense row (longer than an LDS tile)
    A[11, :] = 0.0                        # an empty row
    A = A.tocsr()
    x = rng.normal(size=n)
    y = CSRMatrix(ctx, A).apply(ctx.from_host(x)).get()[:, 0]
    assert relerr(y, A @ x) < 1e-13


@pytest.mark.parametrize("k", [2, 3, 6, 13, 32, 40, 70])
def test_spmm_matches_scipy(ctx, k):
    from eigd_amd.device import CSRMatrix

    A = grid_matrix(61, 47, 2, seed=k)
    rng = np.random.default_rng(k)
    X = rng.normal(size=(A.shape[0], k))
    Y = CSRMatrix(ctx, A).apply(ctx.from_host(X)).get()
    assert np.array_equal(Y, A @ X)


@pytest.mark.parametrize("k", [3, 5, 8, 13, 24, 32, 40])
def test_spmm_with_the_inner_products_of_a_cg_step_in_the_same_pass(ctx, k):
    """eigd_spmm_cg: y bit-identical to the plain product, gam / rho / log as the dot kernel's coefficients (csrc/krylov.hip)"""
    from eigd_amd import _ffi
    from eigd_amd._ffi import call
    from eigd_amd.device import CSRMatrix

    A = grid_matrix(61, 47, 2, seed=k)
    A = (A + A.T).tocsr()
    n = A.shape[0]
    rng = np.random.default_rng(k)
    Z, R = rng.normal(size=(n, k)), rng.normal(size=(n, k))
    dA = CSRMatrix(ctx, A)
    dZ, dR = ctx.from_host(Z), ctx.from_host(R)
    nrows = int(_ffi.lib().eigd_cg_state_rows())
    st = np.zeros((nrows, 64))
    st[4, :k] = 1e-30                                   # tol^2
    st[5, :k] = -rng.uniform(0.1, 0.5, size=k)          # alpha_i < 0: rr - alpha z.y > 0 for a positive z.y ... any sign is reported
    out = {}
    for name in ("fused", "separate"):
        state, log, Y = ctx.from_host(st), ctx.zeros(4, 64), ctx.zeros(n, k)
        if name == "fused":
            call("eigd_spmm_cg", ctx.h, dA.h, k, dZ.ptr, dZ.ld, Y.ptr, Y.ld, dR.ptr, dR.ld, None, state.ptr, 1, 1, log.ptr)
        else:
            dA.apply(dZ, Y)
            call("eigd_cg_coefficients", ctx.h, n, k, dZ.ptr, dZ.ld, dR.ptr, dR.ld, Y.ptr, Y.ld, None, state.ptr, 1, 1, log.ptr)
        out[name] = (Y.get(), state.get(), log.get())
    assert np.array_equal(out["fused"][0], A @ Z) and np.array_equal(out["separate"][0], A @ Z)
    rr, zy = np.sum(Z * R, axis=0), np.sum(Z * (A @ Z), axis=0)
    for name in out:
        stt, lg = out[name][1], out[name][2]
        ok = (rr > 0) & (rr - st[5, :k] * zy > 0)
        gam = np.where(ok, rr / (rr - st[5, :k] * zy), 0.0)
        assert np.allclose(stt[8, :k], gam, rtol=1e-12, atol=0) and np.allclose(lg[0, :k], gam, rtol=1e-12, atol=0), name
        assert np.allclose(stt[0, :k][ok], rr[ok], rtol=1e-12)
        assert np.array_equal(stt[7, :k] == 2.0, ~ok & (rr != 0))
    assert np.allclose(out["fused"][1], out["separate"][1], rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("k", [5, 32])
def test_spmm_tiles_with_many_nonzeros_and_mixed_rows(ctx, k):
    """banded rows (41 non-zeros each: a 32-row tile holds more than the 1024 non-zeros staged through LDS), a few
    empty rows and one dense row in between: the tiled kernel's direct path and the fallback, bit-identical to scipy"""
    from eigd_amd.device import CSRMatrix

    n = 5003
    rng = np.random.default_rng(k)
    offs = list(range(-20, 21))
    A = sparse.diags([rng.uniform(-1.0, 1.0, n - abs(o)) for o in offs], offs, shape=(n, n)).tolil()
    A[100:103, :] = 0.0
    A = A.tocsr()
    A.eliminate_zeros()
    A.sort_indices()
    X = rng.normal(size=(n, k))
    assert np.array_equal(CSRMatrix(ctx, A).apply(ctx.from_host(X)).get(), A @ X)
    B = A.tolil()
    B[2500, :] = rng.uniform(-1.0, 1.0, n)  # one full row: the column list of its tile no longer fits LDS
    B = B.tocsr()
    B.sort_indices()
    assert np.array_equal(CSRMatrix(ctx, B).apply(ctx.from_host(X)).get(), B @ X)


@pytest.mark.parametrize("n,k", [(1, 1), (257, 1), (10007, 5), (50000, 32), (4099, 64)])
def test_column_kernels(ctx, n, k):
    rng = np.random.default_rng(n + k)
    X, Y, Z = rng.normal(size=(n, k)), rng.normal(size=(n, k)), rng.normal(size=(n, k))
    dX, dY, dZ = ctx.from_host(X), ctx.from_host(Y), ctx.from_host(Z)
    assert np.allclose(dX.coldot(dY), np.einsum("ij,ij->j", X, Y), rtol=1e-12, atol=1e-12 * n)
    assert np.allclose(dX.colnorms(), np.linalg.norm(X, axis=0), rtol=1e-13)
    c1, c2 = rng.normal(size=k), rng.normal(size=k)
    out = ctx.empty(n, k).assign_lincomb([(c1, dX), (c2, dY), (-1.0, dZ)])
    assert relerr(out.get(), X * c1 + Y * c2 - Z) < 1e-15
    dX.assign_lincomb([(2.0, dX), (1.0, dY)])  # aliasing output
    assert relerr(dX.get(), 2.0 * X + Y) < 1e-15


def test_numpy_surface_uses_page_locked_memory(ctx):
    """
    Large results come back in page-locked memory that outlives the device block and returns to a pool with the last
    view; a caller-owned array is page-locked in place from its second upload and released with the array; values are
    those of the plain copies.
    """
    import gc

    from eigd_amd import device as dv

    rng = np.random.default_rng(5)
    n, k = 70000, 8                                      # 4.5 MB: above the threshold
    X = rng.normal(size=(n, k))
    blk = ctx.from_host(X)
    out = blk.get()
    assert not out.flags.owndata and out.flags.writeable and np.array_equal(out, X)
    view = out[:, 2:5]
    del blk, out
    gc.collect()
    assert np.array_equal(view, X[:, 2:5])               # the buffer lives as long as any view of it
    held0 = dv._pinned.held
    del view
    gc.collect()
    assert dv._pinned.held == held0 + 8 * n * k          # back in the pool
    again = ctx.from_host(X).get()                       # ... and handed out again
    assert dv._pinned.held == held0 and np.array_equal(again, X)
    back = ctx.from_host(again)                          # a page-locked array as the source of an upload
    assert np.array_equal(back.get(), X)
    # caller-owned buffer: registered at the second sighting, unregistered when it dies
    Xc = rng.normal(size=(n, k))
    addr = Xc.ctypes.data
    ctx.from_host(Xc)
    assert addr not in dv._pinned.registered
    ctx.from_host(Xc)
    assert addr in dv._pinned.registered
    assert np.array_equal(ctx.from_host(Xc).get(), Xc)
    del Xc
    gc.collect()
    assert addr not in dv._pinned.registered
    small = ctx.from_host(rng.normal(size=(100, 3))).get()
    assert small.flags.owndata                            # small blocks stay ordinary arrays
    P = dv.pinned_empty((1 << 18, 2))
    P[:] = 1.5
    assert np.array_equal(ctx.from_host(P).get(), P)


@pytest.mark.parametrize("n,k", [(1, 1), (1000, 3), (40001, 6), (7001, 64)])
def test_compensated_column_dots_are_exact_to_rounding(ctx, n, k):
    """
    eigd_coldot_dd (the entries of G = -Phi^T Phib of numerically repeated pairs, reference 373-383): hi + lo against the
    exactly rounded sum (math.fsum over error-free products), on ill-conditioned data -- a huge cancelling pair inside
    each column makes a plain dot product lose ten digits.
    """
    import math

    rng = np.random.default_rng(3 * n + k)
    X, Y = rng.normal(size=(n, k)), rng.normal(size=(n, k))
    if n > 2:
        X[0], Y[0] = 1e10, 1.0 + rng.uniform(size=k)
        X[n // 2], Y[n // 2] = -1e10, Y[0]                       # cancels row 0 exactly in exact arithmetic
    hi, lo = ctx.from_host(X).coldot_dd(ctx.from_host(Y))

    def two_prod_terms(x, y):
        # Veltkamp / Dekker: x*y = p + e exactly, in pure Python floats
        out = []
        for a, b in zip(x.tolist(), y.tolist()):
            p = a * b
            sa = a * 134217729.0
            ah = sa - (sa - a)
            al = a - ah
            sb = b * 134217729.0
            bh = sb - (sb - b)
            bl = b - bh
            out += [p, ((ah * bh - p) + ah * bl + al * bh) + al * bl]
        return out

    for c in range(k):
        exact = math.fsum(two_prod_terms(X[:, c], Y[:, c]))
        scale = float(np.abs(X[:, c] * Y[:, c]).sum())
        assert abs((hi[c] + lo[c]) - exact) <= 2 * np.finfo(float).eps * abs(exact) + 1e-28 * scale
        assert abs(lo[c]) <= np.finfo(float).eps * abs(hi[c]) + 1e-300


@pytest.mark.parametrize("n,ku,kx", [(3000, 1, 1), (20011, 6, 6), (70001, 32, 32), (5003, 13, 64), (9001, 64, 5), (3001, 70, 33)])
def test_tall_skinny_products_and_projection(ctx, n, ku, kx):
    rng = np.random.default_rng(ku * 100 + kx)
    U, V, X = rng.normal(size=(n, ku)), rng.normal(size=(n, ku)), rng.normal(size=(n, kx))
    dU, dV, dX = ctx.from_host(U), ctx.from_host(V), ctx.from_host(X)
    C = dU.tdot(dX)
    assert relerr(C, U.T @ X) < 1e-13
    Cm = rng.normal(size=(ku, kx))
    out = dX.copy().add_product(dU, Cm, alpha=0.7, beta=-1.3)
    assert relerr(out.get(), -1.3 * X + 0.7 * U @ Cm) < 1e-14
    out0 = ctx.empty(n, kx).add_product(dU, Cm, alpha=1.0, beta=0.0)  # beta = 0 must ignore garbage
    assert relerr(out0.get(), U @ Cm) < 1e-14
    P = dX.copy().project(dU, dV)
    assert relerr(P.get(), X - U @ (V.T @ X)) < 1e-13
    # column views with a leading dimension
    if kx >= 4:
        sub = dX.cols(1, 3)
        assert np.array_equal(sub.get(), X[:, 1:3])
        assert relerr(dU.tdot(sub), U.T @ X[:, 1:3]) < 1e-13


@pytest.mark.parametrize("n,k,ns", [(5000, 1, 1), (20001, 1, 37), (30011, 8, 9), (10007, 32, 21), (4001, 5, 50)])
def test_stack_kernels(ctx, n, k, ns):
    rng = np.random.default_rng(ns)
    S = rng.normal(size=(ns, n, k))
    T = rng.normal(size=(n, k))
    st = ctx.stack(ns + 2, n, k)
    for j in range(ns):
        st[j].set(S[j])
    dT = ctx.from_host(T)
    H = st.dot(dT, ns=ns)
    Href = np.einsum("jrc,rc->jc", S, T)
    assert np.allclose(H, Href, rtol=1e-12, atol=1e-11 * np.sqrt(n))
    st.axpy_into(dT, Href, alpha=-1.0)
    assert relerr(dT.get(), T - np.einsum("jrc,jc->rc", S, Href)) < 1e-13
    if ns <= 32:  # fused axpy + dot
        T2 = rng.normal(size=(n, k))
        dT2 = ctx.from_host(T2)
        H1 = rng.normal(size=(ns, k))
        H2 = st.axpy_dot_into(dT2, H1, alpha=-0.5)
        Tn = T2 - 0.5 * np.einsum("jrc,jc->rc", S, H1)
        assert relerr(dT2.get(), Tn) < 1e-13
        assert np.allclose(H2, np.einsum("jrc,rc->jc", S, Tn), rtol=1e-11, atol=1e-10 * np.sqrt(n))
    if k == 1:  # column-major view of a k = 1 stack: V^T X and V @ C
        X = rng.normal(size=(n, 7))
        V = S[:, :, 0].T  # n x ns
        assert relerr(st.tdot_block(ctx.from_host(X), ns=ns), V.T @ X) < 1e-12
        Cm = rng.normal(size=(ns, 7))
        out = st.times_into(ctx.empty(n, 7), Cm, ns=ns)
        assert relerr(out.get(), V @ Cm) < 1e-13


def test_gather_scatter_columns(ctx):
    rng = np.random.default_rng(9)
    X = rng.normal(size=(7001, 12))
    dX = ctx.from_host(X)
    cols = [3, 0, 11, 7]
    G = dX.gather_cols(cols)
    assert np.array_equal(G.get(), X[:, cols])
    Z = ctx.zeros(7001, 12)
    G.scatter_cols_into(Z, cols)
    ref = np.zeros_like(X)
    ref[:, cols] = X[:, cols]
    assert np.array_equal(Z.get(), ref)


@pytest.mark.parametrize("nx,ny,dof,leaf,pw", [(9, 7, 2, 8, 8), (31, 29, 2, 16, 16), (64, 64, 2, 48, 64), (150, 131, 1, 32, 64), (200, 200, 2, 0, 0)])
def test_factor_solve_matches_superlu(ctx, nx, ny, dof, leaf, pw):
    from eigd_amd.device import Factor

    A = grid_matrix(nx, ny, dof, seed=nx + ny)
    n = A.shape[0]
    F = Factor(ctx, A, leaf_size=leaf, panel_width=pw)
    lu = splu(A.tocsc())
    rng = np.random.default_rng(0)
    for k in (1, 3, 8, 13, 32, 45):
        B = rng.normal(size=(n, k))
        X = F.solve_inplace(ctx.from_host(B)).get()
        Xref = lu.solve(B)
        assert relerr(X, Xref) < 1e-11, k
        assert np.linalg.norm(A @ X - B) / np.linalg.norm(B) < 1e-12, k
    # alpha and strided views
    B = rng.normal(size=(n, 6))
    dB = ctx.from_host(B)
    F.solve_inplace(dB.cols(2, 5), alpha=-1.0)
    out = dB.get()
    assert np.array_equal(out[:, :2], B[:, :2]) and np.array_equal(out[:, 5:], B[:, 5:])
    assert relerr(out[:, 2:5], -lu.solve(B[:, 2:5])) < 1e-11
    # sweeps are deterministic
    X1 = F.solve_inplace(ctx.from_host(B)).get()
    X2 = F.solve_inplace(ctx.from_host(B)).get()
    assert np.array_equal(X1, X2)


def test_factor_column_independence_and_refactor(ctx):
    """a column's solution does not depend on the width of the block it is solved in"""
    from eigd_amd.device import Factor

    A = grid_matrix(50, 41, 2, seed=4)
    F = Factor(ctx, A)
    rng = np.random.default_rng(3)
    B = rng.normal(size=(A.shape[0], 32))
    Xall = F.solve_inplace(ctx.from_host(B)).get()
    for lo, hi in ((4, 8), (0, 16), (8, 16), (3, 4), (5, 30)):  # the 4-, 16- and 32-column kernels, full and ragged blocks
        Xp = F.solve_inplace(ctx.from_host(B[:, lo:hi])).get()
        assert np.array_equal(Xall[:, lo:hi], Xp), (lo, hi)
    A2 = A + sparse.identity(A.shape[0]) * 0.5
    F.refactor(A2.tocsr())
    X = F.solve_inplace(ctx.from_host(B)).get()
    assert np.linalg.norm(A2 @ X - B) / np.linalg.norm(B) < 1e-12


def test_indefinite_shift_ldlt_and_singular_matrix(ctx):
    """a shift inside the spectrum: sign-tracked LDL^T, inertia = eigenvalues below the shift, SuperLU-level accuracy"""
    import eigd_amd as eg
    from eigd_amd._ffi import NotPositiveDefiniteError
    from eigd_amd.device import Factor

    A = grid_matrix(14, 13, 1, seed=1)
    ev = np.linalg.eigvalsh(A.toarray())
    shift = 0.5 * (ev[6] + ev[7])                      # 7 eigenvalues below the shift
    Aind = (A - sparse.identity(A.shape[0]) * shift).tocsr()
    F = Factor(ctx, Aind, leaf_size=8, panel_width=8)
    assert F.stats()["negative_pivots"] == 7
    rng = np.random.default_rng(0)
    B = rng.normal(size=(A.shape[0], 5))
    X = F.solve_inplace(ctx.from_host(B)).get()
    assert relerr(X, splu(Aind.tocsc()).solve(B)) < 1e-9
    # larger problem through the operator (one refinement step per application)
    A2 = grid_matrix(60, 50, 2, seed=3)
    A2i = (A2 - sparse.identity(A2.shape[0]) * A2.diagonal().mean()).tocsr()  # shift in the middle of the spectrum
    op = eg.SpLuOperator(A2i.tocsc())
    assert op.negative_pivots > 0
    B2 = rng.normal(size=(A2.shape[0], 3))
    X2 = op(B2)
    assert np.linalg.norm(A2i @ X2 - B2) / np.linalg.norm(B2) < 1e-11
    with pytest.raises(NotPositiveDefiniteError):
        Factor(ctx, sparse.kron(sparse.identity(20), np.ones((2, 2))).tocsr())  # exactly singular: zero pivot


def star_matrix(nb, g=6, hub=4, seed=0):
    """nb grid blocks that only talk to each other through a small hub: assembly-tree nodes with many children"""
    rng = np.random.default_rng(seed)
    blocks = [grid_matrix(g, g, 1, seed=i) for i in range(nb)]
    A = sparse.block_diag(blocks + [sparse.identity(hub) * 50.0]).tolil()
    n0, ntot = g * g, nb * g * g + hub
    for b in range(nb):
        for h in range(hub):
            i = b * n0 + rng.integers(n0)
            A[i, ntot - hub + h] = A[ntot - hub + h, i] = 0.3
    A = A.tocsr()
    A.sort_indices()
    return A


@pytest.mark.parametrize("nb,k", [(5, 3), (9, 1), (9, 20), (6, 12), (5, 32)])
def test_fronts_with_many_children_use_the_surplus_plane(ctx, nb, k):
    """more than four children per front: the surplus carries are summed through the scratch / extra planes"""
    from eigd_amd.device import Factor, Symbolic

    A = star_matrix(nb)
    sym = Symbolic(A, leaf_size=16, panel_width=8)
    parent = sym.array("f_parent")
    assert np.bincount(parent[parent >= 0]).max() > 4
    F = Factor(ctx, A, symbolic=sym)
    rng = np.random.default_rng(2)
    B = rng.normal(size=(A.shape[0], k))
    ref = splu(A.tocsc()).solve(B)
    for _ in range(2):  # twice: the planes' never-written entries must still read zero
        X = F.solve_inplace(ctx.from_host(B)).get()
        assert relerr(X, ref) < 1e-12
    X1 = F.solve_inplace(ctx.from_host(B[:, :1])).get()  # another width on the same planes
    assert relerr(X1, ref[:, :1]) < 1e-12


def lap3d(m):
    eye = sparse.identity(m)
    T = sparse.diags([-1.0, 2.2, -1.0], [-1, 0, 1], shape=(m, m))
    A = sparse.kron(sparse.kron(T, eye), eye) + sparse.kron(sparse.kron(eye, T), eye) + sparse.kron(sparse.kron(eye, eye), T)
    A = A.tocsr()
    A.sort_indices()
    return A


def hub_matrix(nx, ny, hub, seed=0, density=0.6):
    """a 2-D grid whose every node is coupled to most of a dense hub block: thin fronts with borders of several hundred rows"""
    rng = np.random.default_rng(seed)
    A = grid_matrix(nx, ny, 1, seed=seed)
    C = sparse.random(A.shape[0], hub, density=density, random_state=seed, data_rvs=lambda k: rng.uniform(-0.01, 0.01, k))
    M = sparse.bmat([[A, C], [C.T, sparse.identity(hub) * 5.0]]).tocsr()
    M.sort_indices()
    return M


@pytest.mark.parametrize("case", ["lap3d", "hub"])
def test_single_tile_fronts_with_long_borders(ctx, case):
    """3-D and hub problems: single-column-tile fronts whose borders exceed what the wave kernels hold in registers
    (bs > 320: tile kernels; d > 384: plane masks loaded per row block), levels that mix single- and multi-tile fronts"""
    from eigd_amd.device import Factor, Symbolic

    A = lap3d(24) if case == "lap3d" else hub_matrix(40, 8, 420)
    sym = Symbolic(A, leaf_size=16 if case == "lap3d" else 24)
    ns, bs = sym.array("f_ns"), sym.array("f_bs")
    assert ((ns <= 64) & (bs > 320)).any() and ((ns <= 32) & (ns + bs > 384)).any() == (case == "hub")
    F = Factor(ctx, A, symbolic=sym)
    lu = splu(A.tocsc())
    rng = np.random.default_rng(5)
    for k in (3, 12, 32):
        B = rng.normal(size=(A.shape[0], k))
        X = F.solve_inplace(ctx.from_host(B)).get()
        assert relerr(X, lu.solve(B)) < 1e-11, (case, k)
    Xall = F.solve_inplace(ctx.from_host(B)).get()
    Xp = F.solve_inplace(ctx.from_host(B[:, 8:12])).get()
    assert np.array_equal(Xall[:, 8:12], Xp)


def test_fem_like_ill_conditioned_factor(ctx):
    """Q4 plate-like stencil with a 1e6 stiffness contrast (SIMP void/solid), solved to SuperLU accuracy"""
    from eigd_amd.device import Factor

    nx = 80
    K = grid_matrix(nx, nx, 2, seed=8)
    rng = np.random.default_rng(8)
    s = 10.0 ** rng.uniform(-3, 3, size=K.shape[0])
    D = sparse.diags(np.sqrt(s))
    A = (D @ K @ D).tocsr()
    A.sort_indices()
    F = Factor(ctx, A)
    B = rng.normal(size=(A.shape[0], 4))
    X = F.solve_inplace(ctx.from_host(B)).get()
    Xref = splu(A.tocsc()).solve(B)
    assert relerr(X, Xref) < 1e-9


def test_factor_with_geometric_ordering(ctx):
    from eigd_amd.device import Factor

    nx, ny = 90, 70
    A = grid_matrix(nx, ny, 2, seed=5)
    xy = np.stack([np.repeat(np.arange(nx), ny), np.tile(np.arange(ny), nx)], axis=1).astype(float)
    F = Factor(ctx, A, coords=np.repeat(xy, 2, axis=0))
    rng = np.random.default_rng(1)
    for k in (1, 7, 16, 32):
        B = rng.normal(size=(A.shape[0], k))
        X = F.solve_inplace(ctx.from_host(B)).get()
        assert np.linalg.norm(A @ X - B) / np.linalg.norm(B) < 1e-12


@pytest.mark.parametrize("ns,k,c0", [(5, 7, 0), (20, 32, 0), (40, 12, 9)])
def test_fused_gram_schmidt_step_and_device_norms(ctx, ns, k, c0):
    """eigd_stack_cgs2 / eigd_colnorm2_dev / eigd_scale_inv_norm against numpy"""
    rng = np.random.default_rng(ns)
    n, kw = 20011, k + c0
    Wh = np.linalg.qr(rng.normal(size=(n, ns)))[0]             # the same orthonormal basis in every column
    st = ctx.stack(ns, n, kw)
    for j in range(ns):
        st[j].copy_from(ctx.from_host(np.repeat(Wh[:, j:j + 1], kw, axis=1)))
    T0 = rng.normal(size=(n, k)) + Wh @ rng.normal(size=(ns, k)) * 1e3   # large components along W: second pass needed
    T = ctx.from_host(T0)
    H, passes = st.cgs2(T, ns, c0=c0, tol=1e-13)
    href = Wh.T @ T0
    assert relerr(H, href) < 1e-12
    Tn = T.get()
    assert np.abs(Wh.T @ Tn).max() < 1e-10 * np.abs(T0).max()
    assert relerr(Tn, T0 - Wh @ href) < 1e-9
    assert passes in (2, 3, 4)
    H2, passes2 = st.cgs2(T, ns, c0=c0, tol=1e-13)             # already orthogonal: nothing left to subtract
    assert passes2 <= 4 and np.abs(H2).max() < 1e-9 * np.abs(T0).max()
    n2 = T.colnorm2_dev()
    assert relerr(n2.get()[0], np.sum(Tn * Tn, axis=0)) < 1e-13
    assert np.array_equal(ctx.fetch_colnorm2(k), n2.get()[0])
    skip = np.zeros(k, dtype=bool)
    skip[0] = True
    out = ctx.empty(n, k).assign_scaled_inverse(T, n2, skip).get()
    ref = Tn / np.sqrt(np.sum(Tn * Tn, axis=0))
    ref[:, 0] = 0.0
    assert relerr(out, ref) < 1e-13


@pytest.mark.parametrize("n,ku,k", [(7001, 7, 5), (20000, 32, 32), (4099, 33, 17), (333, 3, 64)])
def test_projection_with_fused_column_norms(ctx, n, ku, k):
    """X <- X - U (V^T X) and the squared column norms of the result in one pass (sibk lines 1257 + 1259)"""
    rng = np.random.default_rng(n + k)
    U, V, X = rng.normal(size=(n, ku)), rng.normal(size=(n, ku)) / n, rng.normal(size=(n, k))
    dU, dV = ctx.from_host(U), ctx.from_host(V)
    dX = ctx.from_host(X)
    n2 = dX.project_norm2(dU, dV)
    got = dX.get()
    ref = X - U @ (V.T @ X)
    assert relerr(got, ref) < 1e-13
    assert np.allclose(n2.get()[0], (got * got).sum(axis=0), rtol=1e-13, atol=0.0)
    assert np.allclose(ctx.fetch_colnorm2(k), (got * got).sum(axis=0), rtol=1e-13, atol=0.0)
    # a strided view: the neighbouring columns stay untouched
    if k >= 5:
        dY = ctx.from_host(X)
        n2v = dY.cols(1, 4).project_norm2(dU, dV)
        out = dY.get()
        assert np.array_equal(out[:, :1], X[:, :1]) and np.array_equal(out[:, 4:], X[:, 4:])
        assert relerr(out[:, 1:4], ref[:, 1:4]) < 1e-13
        assert np.allclose(n2v.get()[0], (out[:, 1:4] ** 2).sum(axis=0), rtol=1e-13, atol=0.0)
        ctx.fetch_colnorm2(3)


@pytest.mark.parametrize("n,ku,k", [(20000, 63, 64), (9001, 96, 32), (5000, 12, 6)])
def test_projection_update_is_skipped_when_the_block_is_already_projected(ctx, n, ku, k):
    """
    project_norm2 measures its own update pass (the projection behind a Gram-Schmidt step, 1257, meets vectors built from
    projected ones): with no coefficient above 1e-13 of its column's norm the block is left bit for bit as it is and the
    norms of the coefficient pass are returned; a block with a component along U is projected as before.  A zero column
    (a finished mode riding along) does not force the update.
    """
    rng = np.random.default_rng(ku + k)
    U = np.linalg.qr(rng.normal(size=(n, ku)))[0]
    V = U.copy()                                           # V^T U = I: an orthogonal projector
    X = rng.normal(size=(n, k))
    X[:, k // 2] = 0.0
    dU, dV = ctx.from_host(U), ctx.from_host(V)
    ctx.project_stats()
    dX = ctx.from_host(X)
    n2 = dX.project_norm2(dU, dV)                          # components of order one along U: the update runs
    P = dX.get()
    assert relerr(P, X - U @ (V.T @ X)) < 1e-13
    assert np.allclose(n2.get()[0], (P * P).sum(axis=0), rtol=1e-13, atol=0.0)
    assert ctx.project_stats() == (1, 1)
    n2b = dX.project_norm2(dU, dV)                         # what is left along U is rounding: nothing to do
    assert np.array_equal(dX.get(), P)
    assert np.allclose(n2b.get()[0], (P * P).sum(axis=0), rtol=1e-13, atol=0.0)
    assert np.allclose(ctx.fetch_colnorm2(k), (P * P).sum(axis=0), rtol=1e-13, atol=0.0)
    assert ctx.project_stats() == (1, 0)
    Y = P.copy()
    Y[:, 0] += 1e-9 * np.linalg.norm(P[:, 0]) * U[:, 0]    # one column picks up 1e-9 of a direction of U
    dY = ctx.from_host(Y)
    dY.project_norm2(dU, dV)
    ctx.fetch_colnorm2(k)
    assert ctx.project_stats() == (1, 1)
    assert np.abs(U.T @ dY.get()).max() < 1e-13 * np.linalg.norm(P[:, 0])


@pytest.mark.parametrize("n,ku,k", [(30011, 64, 8), (4099, 40, 1), (9000, 8, 4)])
def test_block_gram_schmidt_step_keeps_its_coefficients_on_the_device(ctx, n, ku, k):
    """
    eigd_project_to: X <- X - U (V^T X) with the coefficients written into rows of a device block (the restarted block
    Lanczos fetches the coefficients of a whole step at once); with a tolerance the update is measured and the flag
    says whether it ran.
    """
    rng = np.random.default_rng(n + ku)
    U = np.linalg.qr(rng.normal(size=(n, ku)))[0]
    X = rng.normal(size=(n, k))
    dU = ctx.from_host(U)
    Cd = ctx.zeros(2 * ku + 1, k)
    dX = ctx.from_host(X)
    dX.project_to(dU, dU, Cd.rows(0, ku))
    flag = Cd.rows(2 * ku, 2 * ku + 1).cols(0, 1)
    dX.project_to(dU, dU, Cd.rows(ku, 2 * ku), tol=1e-13, flag=flag)
    Ch = Cd.get()
    assert relerr(Ch[:ku], U.T @ X) < 1e-13
    P = X - U @ (U.T @ X)
    assert relerr(dX.get(), P) < 1e-13
    assert np.abs(Ch[ku: 2 * ku]).max() < 1e-13 * np.abs(X).max() * np.sqrt(n) and Ch[2 * ku, 0] == 0.0
    Y = P.copy()
    Y[:, 0] += 1e-7 * U[:, 0]
    dY = ctx.from_host(Y)
    dY.project_to(dU, dU, Cd.rows(ku, 2 * ku), tol=1e-13, flag=flag)
    Ch = Cd.get()
    assert Ch[2 * ku, 0] == 1.0 and abs(Ch[ku, 0] - 1e-7) < 1e-12
    assert np.abs(U.T @ dY.get()).max() < 1e-12


def test_measured_second_pass_does_not_depend_on_the_scale_of_the_inner_product(ctx):
    """
    The second Gram-Schmidt pass of the block eigensolver is gated by the B-norm of what the first pass left (the
    relative B-orthogonality of the new vectors), not by the Euclidean norm of the block: with B = s^2 I, s = 1e-4 (a
    consistent mass matrix on a fine mesh has entries of that size) a remainder of 5e-10 |x|_B along a basis vector is
    5e-14 |x|_2 in absolute terms -- the Euclidean test lets it pass (shown), the B-norm test removes it.
    """
    n, ku, k, sc = 30011, 64, 8, 1e-4
    rng = np.random.default_rng(11)
    Q = np.linalg.qr(rng.normal(size=(n, ku)))[0]
    V, BV = Q / sc, Q * sc                                     # B-orthonormal basis and B times it, B = sc^2 I
    X0 = rng.normal(size=(n, k))
    dV, dBV = ctx.from_host(V), ctx.from_host(BV)
    Cd = ctx.zeros(2 * ku + 2, k)
    flag, nb2 = Cd.rows(2 * ku, 2 * ku + 1).cols(0, 1), Cd.rows(2 * ku + 1, 2 * ku + 2)
    dX = ctx.from_host(X0)
    dX.project_to(dV, dBV, Cd.rows(0, ku))
    P = dX.get()
    xB = sc * np.linalg.norm(P[:, 0])                         # |x_0|_B
    delta = 0.5e-9 * xB
    assert delta < 0.6e-13 * np.linalg.norm(P[:, 0])           # invisible to the Euclidean test
    Y = P.copy()
    Y[:, 0] += delta * V[:, 0]
    dY = ctx.from_host(Y)
    dY.project_to(dV, dBV, Cd.rows(ku, 2 * ku), tol=1e-13, flag=flag)                       # Euclidean gate: skipped
    assert Cd.get()[2 * ku, 0] == 0.0 and np.array_equal(dY.get(), Y)
    dBY = ctx.from_host(sc * sc * Y)
    dY.coldot_dev(dBY, nb2)
    assert relerr(Cd.get()[2 * ku + 1], sc * sc * np.sum(Y * Y, axis=0)) < 1e-13
    dY.project_to(dV, dBV, Cd.rows(ku, 2 * ku), tol=1e-13, flag=flag, norm2=nb2)            # against the B-norms
    Ch = Cd.get()
    assert Ch[2 * ku, 0] == 1.0 and abs(Ch[ku, 0] - delta) < 1e-3 * delta
    assert np.abs(BV.T @ dY.get())[:, 0].max() < 1e-13 * xB
    # and a block that IS B-orthogonal to the level of rounding is left alone by the same test
    dZ = ctx.from_host(P)
    dZ.project_to(dV, dBV, Cd.rows(ku, 2 * ku), tol=1e-13, flag=flag, norm2=nb2)
    dZ.project_to(dV, dBV, Cd.rows(ku, 2 * ku), tol=1e-13, flag=flag, norm2=nb2)
    assert Cd.get()[2 * ku, 0] == 0.0


@pytest.mark.parametrize("n,p,cond", [(20011, 8, 1e2), (9001, 8, 1e10), (5003, 1, 1.0), (7001, 16, 1e6), (4001, 32, 1e4)])
def test_block_orthonormalisation_on_the_device(ctx, n, p, cond):
    """
    eigd_svqb_step (two passes, as the block eigensolver runs them): the block comes out B-orthonormal to rounding, the
    accumulated p x p factor reproduces the input block, X_in = X_out C, and agrees with the host form (lanczos._svqb)
    up to the order and signs of the directions; columns of very different length and a condition number of 1e10 are
    handled (the Gram matrix is scaled to unit diagonal and diagonalised by Jacobi rotations).
    """
    from scipy import sparse

    rng = np.random.default_rng(n + p)
    B = sparse.diags(rng.uniform(0.5, 2.0, size=n) * 1e-6).tocsr()          # a mass matrix of small scale
    Q = np.linalg.qr(rng.normal(size=(n, p)))[0]
    X0 = (Q * np.logspace(0, np.log10(cond), p)) @ np.linalg.qr(rng.normal(size=(p, p)))[0]
    X0 *= 10.0 ** rng.uniform(-3, 3, size=p)                                  # columns of very different length
    dX, dBX = ctx.from_host(X0), ctx.from_host(B @ X0)
    Cd = ctx.zeros(p + 1, p)
    flag = Cd.rows(p, p + 1).cols(0, 1)
    dX.svqb_step(dBX, Cd.rows(0, p), True, flag)
    dBX.set(B @ dX.get())                                                     # (the eigensolver carries B X along; same here)
    dX.svqb_step(dBX, Cd.rows(0, p), False, flag)
    X1, Ch = dX.get(), Cd.get()
    assert Ch[p, 0] == 0.0
    assert np.abs(X1.T @ (B @ X1) - np.eye(p)).max() < 1e-13
    assert relerr(X1 @ Ch[:p], X0) < 1e-9 * max(1.0, cond * 1e-6)
    assert relerr(dBX.get(), B @ X1) < 1e-12 * np.sqrt(cond)
    # the projector on the block's range is the host form's (SVQB twice through numpy's eigh)
    from eigd_amd.lanczos import _svqb

    Xh = X0.copy()
    for _ in range(2):
        G = Xh.T @ (B @ Xh)
        Xh = Xh @ _svqb(0.5 * (G + G.T))[0]
    if cond <= 1e6:    # (numpy's eigh resolves the small end of the Gram spectrum to eps cond^2 only: nothing to compare beyond)
        assert np.abs(Xh.T @ (B @ X1) @ (X1.T @ (B @ Xh)) - np.eye(p)).max() < 1e-9 + 1e-14 * cond ** 2


def test_block_orthonormalisation_flags_a_dependent_block(ctx):
    n, p = 6007, 8
    rng = np.random.default_rng(0)
    X0 = rng.normal(size=(n, p))
    X0[:, 5] = 0.0                                                            # a direction that carries nothing
    dX, dBX = ctx.from_host(X0), ctx.from_host(X0)
    Cd = ctx.zeros(p + 1, p)
    dX.svqb_step(dBX, Cd.rows(0, p), True, Cd.rows(p, p + 1).cols(0, 1))
    assert Cd.get()[p, 0] == 1.0


def test_split_chain_hand_off_is_reproducible_over_many_sweeps(ctx):
    """the in-launch hand-off of partial blocks (big fronts near the root) gives the same bits sweep after sweep"""
    from eigd_amd.device import Factor

    A = grid_matrix(220, 210, 2, seed=11)          # top separators of ~440 dofs: chains of 7 tiles, cut into groups
    F = Factor(ctx, A)
    rng = np.random.default_rng(5)
    ref = {}
    for k in (4, 32):
        B = ctx.from_host(rng.normal(size=(A.shape[0], k)))
        X = ctx.empty(A.shape[0], k)
        F.solve_to(B, X)
        ref[k] = (B, X.get())
        assert np.linalg.norm(A @ ref[k][1] - B.get()) / np.linalg.norm(B.get()) < 1e-12
    for it in range(150):
        k = 4 if it % 3 else 32
        B, x0 = ref[k]
        X = ctx.empty(A.shape[0], k)
        F.solve_to(B, X)
        assert np.array_equal(X.get(), x0), (it, k)


def test_tiny_and_disconnected_matrices(ctx):
    """n = 1, 2, 5 and a block-diagonal matrix with isolated dofs: forests of tiny fronts, roots without borders"""
    from eigd_amd.device import CSRMatrix, Factor

    rng = np.random.default_rng(7)
    cases = [sparse.csr_matrix(np.array([[2.5]])),
             sparse.csr_matrix(np.array([[2.0, -1.0], [-1.0, 2.0]])),
             sparse.csr_matrix(np.diag([1.0, 2.0, 3.0, 4.0, 5.0]) + 0.1 * np.ones((5, 5))),
             sparse.block_diag([grid_matrix(6, 5, 2, seed=1), grid_matrix(4, 4, 1, seed=2), sparse.identity(3) * 2.0,
                                grid_matrix(9, 3, 1, seed=3)]).tocsr()]
    for A in cases:
        A.sort_indices()
        n = A.shape[0]
        F = Factor(ctx, A, leaf_size=8, panel_width=8)
        for k in (1, 3, 9, 33):
            B = rng.normal(size=(n, k))
            X = F.solve_inplace(ctx.from_host(B)).get()
            assert np.linalg.norm(A @ X - B) <= 1e-12 * np.linalg.norm(B), (n, k)
        x = rng.normal(size=n)
        assert np.array_equal(CSRMatrix(ctx, A).apply(ctx.from_host(x)).get()[:, 0], A @ x)
        Xm = rng.normal(size=(n, 5))
        assert np.array_equal(CSRMatrix(ctx, A).apply(ctx.from_host(Xm)).get(), A @ Xm)


def test_device_assembly_matches_host_assembly_and_feeds_the_factor(ctx):
    """element assembly on the device = scipy COO assembly; values go straight into SpMV and refactor"""
    import eigd_amd as eg
    from eigd_amd.device import CSRMatrix, ElementAssembler
    from eigd_amd.problems import BucklingColumn

    col = BucklingColumn(24, 31, seed=3)
    K = col.stiffness()                                  # host assembly (vectorised scipy COO -> CSR)
    asm = ElementAssembler(ctx, col.elem_dofs, col.n)
    P = asm.pattern()
    assert P.nnz == K.nnz and np.array_equal(P.indptr, K.indptr) and np.array_equal(P.indices, K.indices)
    scale = col.rhoE**col.p + col.rho0_K
    vals = asm.assemble(col.Ke0, scale)
    v = asm.values_to_host(vals)
    assert np.abs(v - K.data).max() <= 4e-16 * np.abs(K.data).max()      # same sums, possibly another order
    assert np.array_equal(asm.values_to_host(asm.assemble(col.Ke0, scale)), v)   # reproducible
    # per-element matrices
    Ke = scale[:, None, None] * col.Ke0[None]
    assert np.abs(asm.values_to_host(asm.assemble(Ke)) - K.data).max() <= 4e-16 * np.abs(K.data).max()
    # new densities: assemble on the device, update the SpMV matrix and the factor without a host round trip
    fac = eg.SpLuOperator(K.tocsc(), ctx=ctx)
    dK = CSRMatrix(ctx, K)
    rho2 = np.random.default_rng(1).uniform(0.4, 1.0, size=col.mesh.nelems)
    scale2 = rho2**col.p + col.rho0_K
    vals2 = asm.assemble(col.Ke0, scale2)
    dK.update_values_device(vals2)
    fac.refactor_device(vals2)
    col2 = BucklingColumn(24, 31, rhoE=rho2)
    K2 = col2.stiffness()
    x = np.random.default_rng(2).normal(size=col.n)
    y = dK.apply(ctx.from_host(x)).get()[:, 0]
    assert np.linalg.norm(y - K2 @ x) <= 1e-14 * np.linalg.norm(K2 @ x)
    b = np.random.default_rng(3).normal(size=(col.n, 3))
    X = fac(b)
    assert np.linalg.norm(K2 @ X - b) <= 1e-11 * np.linalg.norm(b)
    with pytest.raises(ValueError):
        asm.assemble(np.zeros((7, 7)))


def test_stress_stiffness_on_the_device_matches_the_host_formula(ctx):
    """geometric stiffness G(u): element matrices linear in u made on the device, assembled on the device"""
    from scipy.sparse.linalg import splu as host_lu

    from eigd_amd.device import ElementAssembler, ElementLinearMatrices
    from eigd_amd.problems import BucklingColumn

    col = BucklingColumn(17, 23, seed=5)
    K = col.stiffness()
    u = col.full_vector(host_lu(K.tocsc()).solve(col.f[col.reduced]))
    G = col.geometric_stiffness(u)                      # host: element_G + COO assembly
    full, L, Q = col.stress_stiffness_tables()
    Ge_dev = ElementLinearMatrices(ctx, full, L, Q)(ctx.from_host(u))
    Ge = Ge_dev.get()[:, 0].reshape(col.mesh.nelems, 8, 8)
    assert np.abs(Ge - col.Ge_unit).max() <= 1e-13 * np.abs(col.Ge_unit).max()
    asm = ElementAssembler(ctx, col.elem_dofs, col.n)
    scale = col.rhoE**col.p + col.rho0_G
    vals = asm.values_to_host(asm.assemble(Ge_dev, scale))
    P = asm.pattern()
    Gd = sparse.csr_matrix((vals, P.indices, P.indptr), shape=P.shape)
    assert abs(Gd - G).max() <= 1e-13 * abs(G).max()


def test_bunch_kaufman_pivoting_inside_the_fronts(ctx):
    """
    Interior shifts on which the unpivoted L S L^T breaks down (reference: SuperLU pivots, eigenvector_derivatives.py:11-23;
    examples/crm.py:26, 221 shifts into the spectrum).  (a) the first pivot of the elimination order is EXACTLY zero:
    sigma = K_pp / M_pp for the first eliminated dof p; (b) shifts that make leading principal minors nearly singular
    (sigma = a Ritz value of a leading block); each against SuperLU, inertia against the dense spectrum.
    """
    import eigd_amd as eg
    from eigd_amd.device import Symbolic

    K = grid_matrix(21, 17, 2, seed=5)
    n = K.shape[0]
    rng = np.random.default_rng(3)
    M = sparse.diags(rng.uniform(0.5, 1.5, size=n)).tocsr()
    Md = M.diagonal()
    perm = Symbolic(K).array("perm")
    Kd = K.toarray()
    lam_all = np.linalg.eigvalsh(np.diag(Md ** -0.5) @ Kd @ np.diag(Md ** -0.5))
    B = rng.normal(size=(n, 6))
    shifts = [K[perm[0], perm[0]] / Md[perm[0]]]                       # (a): a_11 = 0 exactly after the shift
    for q in (3, 7, 19):                                               # (b): leading q x q block of the order singular
        idx = perm[:q]
        ev = np.linalg.eigvalsh(np.diag(Md[idx] ** -0.5) @ Kd[np.ix_(idx, idx)] @ np.diag(Md[idx] ** -0.5))
        shifts.append(ev[q // 2])
    for sigma in shifts:
        mat = (K - sigma * M).tocsr()
        op = eg.SpLuOperator(mat.tocsc(), ctx=ctx)
        assert op.negative_pivots == int(np.count_nonzero(lam_all < sigma)), sigma     # inertia (Sylvester)
        X = op(B)
        Xr = splu(mat.tocsc()).solve(B)
        assert np.linalg.norm(mat @ X - B) / np.linalg.norm(B) < 1e-10, sigma
        assert relerr(X, Xr) < 1e-8, sigma
        # the factor alone (no refinement step) is already accurate: the pivoting, not the refinement, does the work
        Xf = op.factor.solve_inplace(ctx.from_host(B)).get()
        assert np.linalg.norm(mat @ Xf - B) / np.linalg.norm(B) < 1e-8, sigma
    # a positive definite matrix never takes the pivoting path: same bits as before
    opd = eg.SpLuOperator((K + 0.3 * M).tocsc(), ctx=ctx)
    assert opd.negative_pivots == 0 and relerr(opd(B), splu((K + 0.3 * M).tocsc()).solve(B)) < 1e-12


def test_front_singular_by_itself_gets_a_static_pivot(ctx):
    """
    Verdict r2 item 9: a shifted matrix whose LEAF front is singular by itself while the matrix is not.  SuperLU's
    partial pivoting over the whole column does not notice (reference eigenvector_derivatives.py:11-23; examples/crm.py
    puts sigma inside the spectrum); Bunch-Kaufman pivoting confined to the panel block finds no pivot there.  The
    column gets a static pivot (+-sqrt(eps) |mat|, counted in ``static_pivots``) and the solves are refined three
    times: SuperLU-level results.
    """
    import eigd_amd as eg
    from eigd_amd.device import Symbolic

    K = grid_matrix(26, 22, 1, seed=4)
    n = K.shape[0]
    rng = np.random.default_rng(8)
    M = sparse.diags(rng.uniform(0.5, 1.5, size=n)).tocsr()
    sym = Symbolic(K, leaf_size=24)
    perm, c0, ns, lvl = sym.array("perm"), sym.array("f_c0"), sym.array("f_ns"), sym.array("f_level")
    leaf = int(np.flatnonzero(lvl == lvl.min())[0])
    idx = perm[c0[leaf]: c0[leaf] + ns[leaf]]                       # the dofs the leaf front eliminates
    assert 4 <= len(idx) <= 64
    Md = M.diagonal()
    Kl = K.toarray()[np.ix_(idx, idx)]
    ev = np.linalg.eigvalsh(np.diag(Md[idx] ** -0.5) @ Kl @ np.diag(Md[idx] ** -0.5))
    lam_all = np.linalg.eigvalsh(np.diag(Md ** -0.5) @ K.toarray() @ np.diag(Md ** -0.5))
    B = rng.normal(size=(n, 5))
    for q in (0, len(ev) // 2):
        sigma = ev[q]                                                # (K - sigma M) restricted to the leaf is singular
        assert np.min(np.abs(lam_all - sigma)) > 1e-4 * abs(sigma)   # ... the matrix itself is not
        mat = (K - sigma * M).tocsr()
        op = eg.SpLuOperator(mat.tocsc(), ctx=ctx, leaf_size=24)
        assert op.static_pivots >= 1
        X = op(B)
        Xr = splu(mat.tocsc()).solve(B)
        assert np.linalg.norm(mat @ X - B) / np.linalg.norm(B) < 1e-10, sigma
        assert relerr(X, Xr) < 1e-8, sigma
    # the operator as the eigensolvers use it (device blocks, several columns, the counter of applications)
    sigma = ev[0]
    mat = (K - sigma * M).tocsr()
    op = eg.SpLuOperator(mat.tocsc(), ctx=ctx, leaf_size=24)
    Xd = ctx.from_host(M @ B)
    op.solve_device(Xd)
    assert op.count == B.shape[1]
    assert relerr(Xd.get(), splu(mat.tocsc()).solve(M @ B)) < 1e-8
    # (a matrix that is exactly singular as a whole still raises: test_indefinite_shift_ldlt_and_singular_matrix)


def test_interior_shift_eigenpairs_and_adjoint_with_streams(ctx):
    """
    shift-invert Lanczos around an interior shift with the pivoted factor: eigenpairs on both sides of the shift, and
    the lock-step adjoint solve with concurrent mode groups (streams = 3: refinement step and sweep lane per stream)
    against the single-stream run
    """
    import warnings

    import eigd_amd as eg

    K = grid_matrix(30, 26, 1, seed=2)
    n = K.shape[0]
    M = sparse.diags(np.random.default_rng(1).uniform(0.5, 1.5, size=n)).tocsr()
    lam_all = np.linalg.eigvalsh(np.diag(M.diagonal() ** -0.5) @ K.toarray() @ np.diag(M.diagonal() ** -0.5))
    sigma = 0.5 * (lam_all[9] + lam_all[10])
    fac = eg.SpLuOperator((K - sigma * M).tocsc(), ctx=ctx)
    assert fac.negative_pivots == 10
    s = eg.BasicLanczos(N=6, m=80, tol=1e-12, ctx=ctx)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lam, Phi = s.solve(K, M, fac, sigma)
    # (the reference's selection: the N algebraically smallest Ritz values) each one an eigenvalue of the pencil
    assert np.all(np.min(np.abs(lam[:, None] - lam_all[None, :]), axis=1) < 1e-8 * np.abs(lam_all).max())
    R = K @ Phi - (M @ Phi) * lam
    assert np.linalg.norm(R, axis=0).max() < 1e-8 * np.abs(K).max()
    Phib = np.random.default_rng(4).uniform(-1, 1, size=(n, 6))
    psi1, data1 = s.solve_adjoint(Phib, method="sibk", rtol=1e-11, streams=1)
    psi3, data3 = s.solve_adjoint(Phib, method="sibk", rtol=1e-11, streams=3)
    assert relerr(psi3, psi1) < 1e-9
    res, _ = s.eval_adjoint_residual_norm(Phib, psi3, b_ortho=True)
    assert res.max() < 1e-8 * np.linalg.norm(Phib, axis=0).max()


def test_bunch_kaufman_over_many_interior_shifts(ctx):
    """60 random shifts inside the spectra of three pencils: no refusal, residual of the pivoted factor alone <= 1e-9
    (measured worst 1.5e-11 over 100 shifts, tools/bk_probe.py), with the refinement step <= 1e-12"""
    import eigd_amd as eg

    rng = np.random.default_rng(0)
    for (nx, ny, dof, seed) in ((40, 37, 1, 1), (25, 24, 2, 2), (60, 50, 2, 3)):
        K = grid_matrix(nx, ny, dof, seed)
        n = K.shape[0]
        M = sparse.diags(rng.uniform(0.5, 1.5, size=n)).tocsr()
        top = 1.2 * (K.diagonal() / M.diagonal()).max()
        B = rng.normal(size=(n, 3))
        for sigma in rng.uniform(0.0, top, size=20):
            mat = (K - sigma * M).tocsr()
            op = eg.SpLuOperator(mat.tocsc(), ctx=ctx)
            Xf = op.factor.solve_inplace(ctx.from_host(B)).get()
            assert np.linalg.norm(mat @ Xf - B) / np.linalg.norm(B) < 1e-9, sigma
            assert np.linalg.norm(mat @ op(B) - B) / np.linalg.norm(B) < 1e-12, sigma


def test_multi_tile_fronts_with_ragged_sizes_through_the_fragment_kernels(ctx):
    """
    Fronts with several column tiles whose sizes are multiples of nothing (own columns and borders that end inside a
    64-tile, inside a 16-row wave strip and inside a K-step of 4): the 16- and 32-column level kernels read them from the
    fragment-major copies (pack_frag_kernel), the narrow ones from the column-major panels -- every width against
    SuperLU, columns bitwise independent of the width, after a refactorisation too, and on the Bunch-Kaufman path
    (dense diagonal blocks in the copies).
    """
    import eigd_amd as eg
    from eigd_amd.device import Factor, Symbolic

    A = lap3d(23)                                   # 12 167 dofs, separators of 23 x 23 = 529 and ragged smaller ones
    sym = Symbolic(A, leaf_size=24)
    ns, bs = sym.array("f_ns"), sym.array("f_bs")
    multi = ns > 64
    assert multi.sum() >= 3 and (ns[multi] % 4 != 0).any() and (ns[multi] % 64 != 0).all() and (bs[multi] % 16 != 0).any()
    F = Factor(ctx, A, symbolic=sym)
    lu = splu(A.tocsc())
    rng = np.random.default_rng(11)
    B = rng.normal(size=(A.shape[0], 32))
    Xall = F.solve_inplace(ctx.from_host(B)).get()
    assert relerr(Xall, lu.solve(B)) < 1e-11
    for lo, hi in ((0, 1), (3, 7), (5, 13), (2, 18), (16, 32)):       # 4-, 8-, 16- and 32-column kernels
        Xp = F.solve_inplace(ctx.from_host(B[:, lo:hi])).get()
        assert np.array_equal(Xall[:, lo:hi], Xp), (lo, hi)
    A2 = (A + 0.37 * sparse.identity(A.shape[0])).tocsr()
    F.refactor(A2)                                                     # the copies follow the numeric phase
    X2 = F.solve_inplace(ctx.from_host(B)).get()
    assert relerr(X2, splu(A2.tocsc()).solve(B)) < 1e-11
    ev = np.sort(np.linalg.eigvalsh(A[:400, :400].toarray()))
    sigma = 0.5 * (ev[200] + ev[201])                                  # well inside the spectrum of A (interlacing)
    Ai = (A - sigma * sparse.identity(A.shape[0])).tocsr()
    op = eg.SpLuOperator(Ai.tocsc(), ctx=ctx)
    assert op.negative_pivots > 100
    Xi = op.factor.solve_inplace(ctx.from_host(B)).get()               # the factor alone, no refinement
    assert np.linalg.norm(Ai @ Xi - B) / np.linalg.norm(B) < 1e-8
    assert relerr(op(B), splu(Ai.tocsc()).solve(B)) < 1e-8
    Xn = op.factor.solve_inplace(ctx.from_host(B[:, 4:9])).get()
    assert np.array_equal(Xi[:, 4:9], Xn)


def hub_star_matrix(nb, g, hub, seed=0):
    """nb grid blocks coupled only through a dense hub of `hub` nodes: a multi-tile front with dozens of children"""
    rng = np.random.default_rng(seed)
    blocks = [grid_matrix(g, g, 1, seed=i) for i in range(nb)]
    H = sparse.csr_matrix(np.full((hub, hub), 0.01) + np.eye(hub) * 50.0)
    A = sparse.block_diag(blocks + [H]).tolil()
    n0, ntot = g * g, nb * g * g + hub
    for b in range(nb):
        for h in range(hub):
            for _ in range(2):
                i = b * n0 + rng.integers(n0)
                A[i, ntot - hub + h] = A[ntot - hub + h, i] = 0.3
    A = A.tocsr()
    A.sort_indices()
    return A


@pytest.mark.parametrize("case", ["binary", "many_children"])
def test_pre_assembled_right_hand_sides_are_bitwise_what_the_row_tile_workgroups_gather(ctx, monkeypatch, case):
    """
    Levels with thousands of multi-tile workgroups (the shell model) get v1 = alpha x + carries written once per level
    (v1_assemble_kernel) and row-tile workgroups that read it as it lies; EIGD_PRE_MIN_WG = 1 switches that on for every
    multi-tile level of a small factor: every MFMA width bitwise against the factor whose workgroups gather v1 themselves,
    sweeps of other widths in between (the planes keep what a wider sweep left in the columns a narrower one does not use),
    fronts with two children and with dozens (surplus planes), Cholesky and Bunch-Kaufman.
    """
    from eigd_amd.device import Factor, Symbolic

    if case == "binary":
        A = lap3d(23)
        sym = Symbolic(A, leaf_size=24)
    else:
        A = hub_star_matrix(6, 10, 70)
        sym = Symbolic(A, leaf_size=16, panel_width=8)
        ns, parent = sym.array("f_ns"), sym.array("f_parent")
        nchild = np.bincount(parent[parent >= 0], minlength=len(ns))
        assert ((ns > 64) & (nchild > 4)).any()
    assert (sym.array("f_ns") > 64).any()
    rng = np.random.default_rng(5)
    B = rng.normal(size=(A.shape[0], 32))
    for shift in (0.0, None):
        if shift is None:                                # well inside the spectrum: dense diagonal blocks, 2 x 2 pivots
            ev = np.sort(np.linalg.eigvalsh(A[:300, :300].toarray()))
            Ai = (A - 0.5 * (ev[150] + ev[151]) * sparse.identity(A.shape[0])).tocsr()
        else:
            Ai = A
        monkeypatch.delenv("EIGD_PRE_MIN_WG", raising=False)
        F0 = Factor(ctx, Ai, symbolic=sym)
        monkeypatch.setenv("EIGD_PRE_MIN_WG", "1")
        F1 = Factor(ctx, Ai, symbolic=sym)
        assert F1.stats()["workspace_planes"] == F0.stats()["workspace_planes"] + 1
        assert (F0.stats()["negative_pivots"] > 20) == (shift is None)
        for lo, hi in ((0, 32), (3, 23), (0, 16), (7, 12), (0, 32), (1, 2)):
            X0 = F0.solve_inplace(ctx.from_host(B[:, lo:hi])).get()
            X1 = F1.solve_inplace(ctx.from_host(B[:, lo:hi])).get()
            assert np.array_equal(X0, X1), (case, shift, lo, hi)
        if shift == 0.0:
            assert relerr(X1, splu(Ai.tocsc()).solve(B[:, 1:2])) < 1e-11
            # a sweep lane (another stream's own vector workspaces, concurrent mode groups) has its own plane of v1
            side = ctx.fork(1)
            Bs, Xs = side.from_host(B[:, 3:23]), side.empty(A.shape[0], 20)
            F1.solve_to(Bs, Xs)
            side.sync()
            assert np.array_equal(Xs.get(), F0.solve_inplace(ctx.from_host(B[:, 3:23])).get())


def test_long_borders_of_multi_tile_fronts_through_the_index_list_in_lds(ctx):
    """
    The 32-column backward level kernel keeps a front's border rows (its rows of the caller's block) in LDS, read once
    -- 2048 entries in one round, the rest in a loop, up to 4096 -- where the narrower kernels hold eight of them per
    lane in registers, requested two chain steps ahead.  A front with several column tiles and a border of 3226 rows:
    32 columns against the residual, and bitwise against the 16-column kernels on the same columns.
    """
    from eigd_amd.device import Factor, Symbolic

    A = hub_matrix(72, 72, 2200, density=0.3)
    sym = Symbolic(A, leaf_size=24)
    ns, bs, parent = sym.array("f_ns"), sym.array("f_bs"), sym.array("f_parent")
    multi = (ns > 64) & (parent >= 0)
    assert ((bs[multi] > 2048) & (bs[multi] <= 4096)).any()
    F = Factor(ctx, A, symbolic=sym)
    rng = np.random.default_rng(9)
    B = rng.normal(size=(A.shape[0], 32))
    for _ in range(2):
        X = F.solve_inplace(ctx.from_host(B)).get()
        assert np.linalg.norm(A @ X - B) / np.linalg.norm(B) < 1e-12
    for lo, hi in ((0, 16), (5, 21), (20, 29)):
        Xp = F.solve_inplace(ctx.from_host(B[:, lo:hi])).get()
        assert np.array_equal(X[:, lo:hi], Xp), (lo, hi)


@pytest.mark.parametrize("n,ns,kx", [(5003, 168, 130), (20011, 64, 64), (9001, 75, 7), (3001, 192, 161), (7000, 130, 81), (4099, 1, 1)])
def test_panel_basis_times_coefficients_in_one_pass(ctx, n, ns, kx):
    """V S for a basis kept as row-major panels of 64 columns (the thick restart of the block Lanczos, arpack.py:41-56):
    all panels of V per launch, the result written once, straight into another set of panels; against numpy and
    against the panel-by-panel product; columns of the last panel beyond the basis hold NaNs and must not enter"""
    from eigd_amd.device import DevicePanels

    rng = np.random.default_rng(n + ns)
    V, S = rng.normal(size=(n, ns)), rng.normal(size=(ns, kx))
    P, Q = DevicePanels(ctx, ns, n), DevicePanels(ctx, max(kx, 1), n)
    P.from_host(np.full((n, P.ncols), np.nan))
    P.from_host(V)
    Q.from_host(np.full((n, Q.ncols), 7.0))
    P.times_panels(Q, S, ns)
    got = Q.to_host(Q.ncols)
    assert relerr(got[:, :kx], V @ S) < 1e-14
    assert np.all(got[:, kx:] == 7.0)                       # nothing is written beyond the kx columns
    R = DevicePanels(ctx, max(kx, 1), n)
    for a in range(0, kx, 64):
        b = min(kx, a + 64)
        P.times_into(R.view(a, b), S[:, a:b], ns=ns)
    assert relerr(R.to_host(kx), got[:, :kx]) < 1e-14
    with pytest.raises(ValueError):
        P.times_panels(P, S, ns)
    # ... and into a plain row-major block (the first guess of the adjoint solvers: psi0 = -[V | Q] [T Cf; C Cf_last])
    blk = ctx.empty(n, kx)
    P.times_panels(blk, S, ns)
    assert np.array_equal(blk.get(), got[:, :kx])
    if kx >= 3:
        wide = ctx.from_host(np.full((n, kx + 2), 3.0))
        P.times_panels(wide.cols(1, kx - 1), S[:, 1:kx - 1], ns)       # a column view with a leading dimension
        w = wide.get()
        assert np.array_equal(w[:, 1:kx - 1], got[:, 1:kx - 1]) and np.all(w[:, [0, kx - 1, kx, kx + 1]] == 3.0)
