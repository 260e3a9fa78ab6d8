"""
Operator layer: the drop-in ``SpLuOperator`` (reference eigd/eigenvector_derivatives.py:11-23)
and the adapters that let the device drivers apply A, B and ``factor`` to device blocks.
"""

import threading

import numpy as np
from scipy import sparse
from scipy.sparse.linalg import LinearOperator

from .device import CSRMatrix, Factor, default_context


def _with_structural_diagonal(csr):
    """
    The symbolic analysis needs every diagonal entry stored.  A shift that annihilates one exactly (sigma = K_pp / M_pp:
    scipy's subtraction then drops the entry) is a legitimate interior shift: the entry is put back as an explicit zero.
    """
    n = csr.shape[0]
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(csr.indptr))
    have = np.zeros(n, dtype=bool)
    have[rows[rows == csr.indices]] = True
    if have.all():
        return csr
    miss = np.flatnonzero(~have)
    coo = csr.tocoo()
    out = sparse.csr_matrix((np.concatenate([coo.data, np.zeros(len(miss))]),
                             (np.concatenate([coo.row, miss]), np.concatenate([coo.col, miss]))), shape=csr.shape)
    out.sort_indices()
    return out


class SpLuOperator(LinearOperator):
    """
    Shift-invert operator ``x -> mat^{-1} x`` factored and applied on the MI355X.

    Same surface as the reference class (``shape``, ``dtype``, ``count``, callable on
    ``(n,)`` and ``(n, k)`` numpy arrays).  ``mat`` must be symmetric.  The factorisation is
    ``P mat P^T = L S L^T`` with ``S = diag(+-1)``: plain Cholesky for the positive definite
    shifts of the reference's examples (K - sigma M below the spectrum, K + sigma G below the
    first buckling load); for a shift inside the spectrum (the reference's CRM example, sigma = omega_0^2) the numeric
    phase is repeated with Bunch-Kaufman pivoting (1 x 1 and 2 x 2 pivots, interchanges inside the 64-column panel of
    a front, so the symbolic structure is kept; ``negative_pivots`` = number of eigenvalues of the pencil below the
    shift) and every application is followed by one step of iterative refinement.  Pivots are not delayed to the parent
    front; a column that is singular inside its panel block -- a front whose own block is singular by itself, which
    SuperLU's partial pivoting over whole columns survives -- gets a static pivot of +-sqrt(eps) |mat| instead
    (``static_pivots`` counts them; SuperLU_DIST and PARDISO do the same) and every application is then refined three
    times against the true matrix.  A matrix that is singular to working precision as a whole is told apart by one
    refined solve of a random system at factorisation time and raises ``NotPositiveDefiniteError`` (a
    ``numpy.linalg.LinAlgError``): the shift sits on an eigenvalue.
    """

    def __init__(self, mat, ctx=None, symbolic=None, leaf_size=0, panel_width=0, check_symmetry=True, coords=None):
        if not sparse.issparse(mat):
            mat = sparse.csr_matrix(mat)
        if mat.shape[0] != mat.shape[1]:
            raise ValueError("expected a square matrix")
        self.ctx = ctx if ctx is not None else default_context()
        self.shape = mat.shape
        self.dtype = np.dtype(np.float64)
        self.count = 0
        self._count_lock = threading.Lock()  # mode groups on different streams share the counter
        # Complex-step matrices (reference 11-23 with a complex mat): mat = M + i dM with dM ~ 1e-20 M is a
        # dual number -- products of two imaginary parts vanish below rounding -- so mat^{-1}(b + i db) =
        # x + i M^{-1}(db - dM x) with x = M^{-1} b: the real factor, applied twice.
        self._imag_dev = None
        if np.issubdtype(mat.dtype, np.complexfloating):
            cm = mat.tocsr()
            cm.sort_indices()
            self.dtype = np.dtype(np.complex128)
            self._imag_dev = CSRMatrix(self.ctx, sparse.csr_matrix((cm.data.imag.copy(), cm.indices, cm.indptr), shape=cm.shape))
            mat = sparse.csr_matrix((cm.data.real.copy(), cm.indices, cm.indptr), shape=cm.shape)
        csr = mat.tocsr().astype(np.float64)  # for a symmetric matrix CSC and CSR coincide
        csr.sort_indices()
        csr = _with_structural_diagonal(csr)
        if check_symmetry:
            x = np.random.default_rng(0).uniform(-1.0, 1.0, size=csr.shape[0])
            d = csr @ x - csr.T @ x
            if np.linalg.norm(d) > 1e-10 * max(np.linalg.norm(csr @ x), 1e-300):
                raise ValueError("SpLuOperator (MI355X) needs a symmetric matrix")
        # coords (optional, one row per dof): geometric nested dissection; without it the ordering is algebraic
        self.factor = Factor(self.ctx, csr, symbolic=symbolic, leaf_size=leaf_size, panel_width=panel_width,
                             coords=coords)
        self.symbolic = self.factor.symbolic
        self._read_inertia()
        # indefinite: pivoting is confined to the panel blocks -> one refinement step per application (three when static
        # pivots were needed)
        self._mat_dev = CSRMatrix(self.ctx, csr) if self._pivoted() else None

    def _pivoted(self):
        return self.negative_pivots > 0 or self.static_pivots > 0

    def _read_inertia(self):
        st = self.factor.stats()
        self.negative_pivots = st["negative_pivots"]
        self.static_pivots = st["static_pivots"]   # pivots singular inside their panel block, replaced by +-sqrt(eps)|A|
        # a static pivot takes its sign from a diagonal entry at rounding level: the inertia is then known only up to
        # their number -- negative_pivots_bounds brackets the count of eigenvalues below the shift
        self.negative_pivots_bounds = (max(0, self.negative_pivots - self.static_pivots),
                                       self.negative_pivots + self.static_pivots)
        self._refine_steps = self.factor.STATIC_PIVOT_REFINEMENTS if self.static_pivots > 0 else 1

    def _refine(self, B, X, alpha):
        """X <- X + mat^{-1} (alpha B - mat X): iterative refinement on device blocks (one step; three with static pivots)"""
        # everything on X's context: with concurrent mode groups (streams > 1) X lives on a forked context whose
        # stream and sweep lane must carry the whole refinement step
        return self.factor.refine(self._mat_dev, B, X, alpha, steps=self._refine_steps)

    # -- device path (used by the drivers) ------------------------------------
    def solve_device(self, X, alpha=1.0, count=None):
        """
        X <- alpha * mat^{-1} X in place on a device block.  ``count`` is the number of columns
        that carry a live right-hand side (finished modes of a lock-step block are zero columns);
        the counter then means what the reference's does: applications per mode (ref 19-22).
        """
        with self._count_lock:
            self.count += X.k if count is None else int(count)
        if self._mat_dev is None:
            return self.factor.solve_inplace(X, alpha)
        B = X.copy()
        self.factor.solve_inplace(X, alpha)
        return self._refine(B, X, alpha)

    def solve_device_to(self, Xin, Xout, alpha=1.0, count=None):
        """Xout <- alpha * mat^{-1} Xin on device blocks, Xin untouched"""
        with self._count_lock:
            self.count += Xin.k if count is None else int(count)
        self.factor.solve_to(Xin, Xout, alpha)
        if self._mat_dev is not None:
            self._refine(Xin, Xout, alpha)
        return Xout

    def refactor_device(self, vals, indefinite_matrix=None):
        """
        numeric refactorisation from CSR values on the device (``ElementAssembler.assemble``; the pattern must be the
        one this operator was built on).  The positive definite case needs nothing else; for an indefinite result pass
        the device CSRMatrix holding the same values (``indefinite_matrix``) for the refinement step.
        """
        self.factor.refactor_device(vals)
        self._read_inertia()
        if self._pivoted() and indefinite_matrix is None:
            raise ValueError("indefinite refactorisation: pass the device matrix for the refinement step")
        self._mat_dev = indefinite_matrix if self._pivoted() else None
        if self.static_pivots > 0:
            self.factor.verify_static_pivots(self._mat_dev)

    def refactor(self, mat):
        """numeric refactorisation with new values on the same sparsity pattern"""
        csr = mat.tocsr().astype(np.float64)
        csr.sort_indices()
        self.factor.refactor(csr)
        self._read_inertia()
        self._mat_dev = CSRMatrix(self.ctx, csr) if self._pivoted() else None

    def solve_device_dual(self, Xr, Xi, count=None):
        """complex-step operand (Xr + i Xi) <- mat^{-1} (Xr + i Xi) in place on two device blocks (see __init__)"""
        if self._imag_dev is None:
            raise TypeError("solve_device_dual needs an operator built on a complex (complex-step) matrix")
        self.solve_device(Xr, count=count)
        T = self._imag_dev.apply(Xr)
        Xi.assign_lincomb([(1.0, Xi), (-1.0, T)])
        self.solve_device(Xi, count=0)
        return Xr, Xi

    # -- host path (reference call surface) ------------------------------------
    def _matvec(self, x):
        x = np.asarray(x)
        if self._imag_dev is not None:
            xc = x.astype(np.complex128).reshape(self.shape[0], -1)
            Xr, Xi = self.ctx.from_host(np.ascontiguousarray(xc.real)), self.ctx.from_host(np.ascontiguousarray(xc.imag))
            self.solve_device_dual(Xr, Xi)
            out = Xr.get() + 1j * Xi.get()
            return out[:, 0] if x.ndim == 1 else out
        X = self.ctx.from_host(x.astype(np.float64).reshape(self.shape[0], -1))
        self.solve_device(X)
        out = X.get()
        return out[:, 0] if x.ndim == 1 else out

    def _matmat(self, X):
        return self._matvec(X)


# ---------------------------------------------------------------------------
class DeviceOperator:
    """y = A x on device blocks for a scipy sparse matrix (device CSR) or a host LinearOperator."""

    def __init__(self, ctx, A):
        self.ctx = ctx
        self.shape = A.shape
        self.host = None
        self.csr = None
        if isinstance(A, CSRMatrix):
            self.csr = A
        elif sparse.issparse(A):
            self.csr = CSRMatrix(ctx, A)
        elif hasattr(A, "A") and sparse.issparse(getattr(A, "A")):  # scipy MatrixLinearOperator
            self.csr = CSRMatrix(ctx, A.A)
        elif isinstance(A, np.ndarray):
            self.csr = CSRMatrix(ctx, sparse.csr_matrix(A))
        else:
            self.host = A  # foreign operator: applied on the host, block copied both ways

    def apply(self, X, Y=None):
        if self.csr is not None:
            return self.csr.apply(X, Y)
        out = np.asarray(self.host @ X.get())
        if Y is None:
            return self.ctx.from_host(out)
        Y.set(out)
        return Y


class FactorApply:
    """X <- alpha factor(X) on device blocks for our SpLuOperator or any foreign callable."""

    def __init__(self, ctx, factor):
        self.ctx = ctx
        self.factor = factor
        self.native = isinstance(factor, SpLuOperator)

    def __call__(self, X, alpha=1.0, count=None):
        if self.native:
            return self.factor.solve_device(X, alpha, count)
        out = np.asarray(self.factor(X.get()))  # honours a user-supplied operator
        X.set(alpha * out.reshape(X.n, X.k))
        return X

    def apply_to(self, Xin, Xout, alpha=1.0, count=None):
        """Xout <- alpha factor(Xin)"""
        if self.native:
            return self.factor.solve_device_to(Xin, Xout, alpha, count)
        out = np.asarray(self.factor(Xin.get()))
        Xout.set(alpha * out.reshape(Xin.n, Xin.k))
        return Xout

