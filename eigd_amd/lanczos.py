"""
Eigen-solvers with eigd's call surface, driven on the MI355X:

  BasicLanczos  (reference eigd/eigenvector_derivatives.py:1331-1870)
  IRAM          (reference 1873-2207 + eigd/arpack.py:24-101, 104-442)

Per Lanczos step the device does one k=1 triangular sweep, SpMVs with B, and a classical
Gram-Schmidt pass (applied twice) against the stored basis in the B inner product.  The
basis V and B V stay resident in HBM as k=1 stacks; only the m x m tridiagonal work
(np.linalg.eigh, sorting, convergence tests) runs on the host, as in the reference.

IRAM is a thick-restart Lanczos: ARPACK's implicit restart is replaced by the mathematically
equivalent explicit restart with the wanted Ritz vectors, producing the same kind of output
the reference extracts from ARPACK's work arrays -- a B-orthonormal basis V (n x m) and a
symmetric projected matrix T (m x m) with  OP V = V T + f e_m^T  -- which is the only property
the Lanczos adjoint approximation uses.  T is tridiagonal apart from
the arrow rows of the last restart.
"""

import warnings

import numpy as np
from scipy.sparse.linalg import aslinearoperator

from . import adjoint as adj
from . import tuning
from .adjoint import DeviceProblem, _is_close
from .device import DeviceBlock, default_context
from .operators import SpLuOperator

METHODS = ("pcpg", "pgmres", "sibk", "laa", "dl")
_TRACE = bool(__import__("os").environ.get("EIGD_TRACE_IRAM"))


def ritz_to_eigs(theta, sigma, mode):
    """undo the spectral transformation and give the sort order (ref 1432-1437, 1960-1965)"""
    if mode == "normal":
        lam = 1.0 / theta + sigma
        return lam, np.argsort(lam)
    lam = sigma * theta / (theta - 1.0)
    return lam, np.argsort(-1.0 / lam)


def complex_step_eigh(T):
    """
    eigh of the reduced matrix; a complex T carries forward derivatives in its imaginary part (reference ``_eigh``,
    1387-1414): lam_i + i q_i^T dT q_i and q_i + i sum_{j: lam_j != lam_i} q_j (q_j^T dT q_i) / (lam_i - lam_j).
    """
    if not np.issubdtype(T.dtype, np.complexfloating):
        return np.linalg.eigh(T)
    lam, Q = np.linalg.eigh(T.real)
    D = Q.T @ T.imag @ Q
    gap = lam[None, :] - lam[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        Cf = np.where(gap == 0.0, 0.0, D / gap)
    return lam + 1j * np.diag(D), Q + 1j * (Q @ Cf)


class _DualLanczosDevice:
    """
    Lanczos basis in complex-step (dual-number) arithmetic on the device: every vector is a pair of real n-vectors
    (value, forward derivative), every product drops the term with two derivative factors (1e-40 relative at the
    reference's step of 1e-20, far below rounding), nothing is conjugated (reference inner product ``y.dot(B @ x)``,
    1503).  Real kernels only: B = Br + i Bi is two CSR matrices, the factor an SpLuOperator built on a complex matrix.
    """

    def __init__(self, ctx, Br, Bi, factor, n, nvec):
        from .operators import DeviceOperator

        self.ctx, self.n, self.fac = ctx, n, factor
        self.Br, self.Bi = DeviceOperator(ctx, Br), DeviceOperator(ctx, Bi)
        self.Vr, self.Vi = ctx.stack(nvec, n, 1), ctx.stack(nvec, n, 1)
        self.BVr, self.BVi = ctx.stack(nvec, n, 1), ctx.stack(nvec, n, 1)
        self.wr, self.wi, self.t = ctx.empty(n, 1), ctx.empty(n, 1), ctx.empty(n, 1)

    def apply_B(self, vr, vi, wr, wi):
        """(wr + i wi) = (Br + i Bi)(vr + i vi)"""
        self.Br.apply(vr, wr)
        self.Br.apply(vi, wi)
        self.Bi.apply(vr, self.t)
        wi.assign_lincomb([(1.0, wi), (1.0, self.t)])

    def normalize_into(self, vr, vi, j):
        """V[j] = v / sqrt(v . B v), BV[j] = B V[j]; returns the complex norm"""
        self.apply_B(vr, vi, self.wr, self.wi)
        re = float(vr.coldot(self.wr)[0])
        im = float(vr.coldot(self.wi)[0]) + float(vi.coldot(self.wr)[0])
        br = np.sqrt(re)
        beta = complex(br, 0.5 * im / br)
        cr, ci = 1.0 / br, -beta.imag / (br * br)  # 1 / beta
        self.Vr[j].assign_lincomb([(cr, vr)])
        self.Vi[j].assign_lincomb([(cr, vi), (ci, vr)])
        self.BVr[j].assign_lincomb([(cr, self.wr)])
        self.BVi[j].assign_lincomb([(cr, self.wi), (ci, self.wr)])
        return beta

    def project_out(self, vr, vi, j0, ns):
        """v <- v - V[:, j0:j0+ns] (BV^T v), twice (as the real path); returns the summed complex coefficients"""
        tot = np.zeros(ns, dtype=complex)
        for _ in range(2):
            hr = self.BVr.dot(vr, ns=ns, j0=j0)
            hi = self.BVr.dot(vi, ns=ns, j0=j0) + self.BVi.dot(vr, ns=ns, j0=j0)
            self.Vr.axpy_into(vi, hi, alpha=-1.0, j0=j0)
            self.Vi.axpy_into(vi, hr, alpha=-1.0, j0=j0)
            self.Vr.axpy_into(vr, hr, alpha=-1.0, j0=j0)
            tot += hr[:, 0] + 1j * hi[:, 0]
        return tot

    def apply_op(self, j, vr, vi):
        """v = factor(B V[j])"""
        vr.copy_from(self.BVr[j])
        vi.copy_from(self.BVi[j])
        self.fac.solve_device_dual(vr, vi)

    def axpy_basis(self, vr, vi, coef, j):
        """v += coef * V[j] for a complex scalar"""
        vi.assign_lincomb([(1.0, vi), (coef.real, self.Vi[j]), (coef.imag, self.Vr[j])])
        vr.assign_lincomb([(1.0, vr), (coef.real, self.Vr[j])])

    def times(self, Y, m):
        """V[:, :m] @ Y for a complex m x q matrix -> host complex array"""
        Y = np.asarray(Y, dtype=complex)
        q = Y.shape[1]
        outr, outi, tmp = self.ctx.empty(self.n, q), self.ctx.empty(self.n, q), self.ctx.empty(self.n, q)
        self.Vr.times_into(outr, np.ascontiguousarray(Y.real), ns=m)
        self.Vr.times_into(outi, np.ascontiguousarray(Y.imag), ns=m)
        self.Vi.times_into(tmp, np.ascontiguousarray(Y.real), ns=m)
        outi.assign_lincomb([(1.0, outi), (1.0, tmp)])
        return outr.get() + 1j * outi.get()

    def basis_to_host(self, m):
        from ._ffi import call, hptr

        out = []
        for st in (self.Vr, self.Vi):
            Vt = np.empty((m, self.n))
            call("eigd_d2h", self.ctx.h, hptr(Vt), st.ptr, 8 * self.n * m)
            out.append(Vt.T)
        return out[0] + 1j * out[1]


def _check_shapes(A, B, factor):
    n = A.shape[1]
    if A.shape != (n, n):
        raise ValueError(f"A must have dimensions ({n},{n})")
    if B.shape != (n, n):
        raise ValueError(f"B must have dimensions ({n},{n})")
    if factor.shape != (n, n):
        raise ValueError(f"Factorized operator must have dimensions ({n},{n})")
    return n


class _LanczosDevice:
    """basis storage + the B-inner-product Gram-Schmidt step shared by both solvers"""

    def __init__(self, prob, nvec):
        self.prob = prob
        self.ctx = prob.ctx
        self.n = prob.n
        self.V = self.ctx.stack(nvec, self.n, 1)
        self.BV = self.ctx.stack(nvec, self.n, 1)
        self.w = self.ctx.empty(self.n, 1)

    def normalize_into(self, v, j):
        """V[j] = v / ||v||_B, BV[j] = B v / ||v||_B ; returns the norm"""
        self.prob.opB.apply(v, self.w)
        nrm = float(np.sqrt(v.coldot(self.w)[0]))
        self.V[j].assign_lincomb([(1.0 / nrm, v)])
        self.BV[j].assign_lincomb([(1.0 / nrm, self.w)])
        return nrm

    def orthogonalize(self, v, ns):
        """v <- v - V[:, :ns] (BV[:, :ns]^T v), twice; returns the summed coefficients (ns,)"""
        h1 = self.BV.dot(v, ns=ns)
        self.V.axpy_into(v, h1, alpha=-1.0)
        h2 = self.BV.dot(v, ns=ns)
        self.V.axpy_into(v, h2, alpha=-1.0)
        return (h1 + h2)[:, 0]

    def apply_op(self, j, out):
        """out = factor(B V[j])"""
        out.copy_from(self.BV[j])
        self.prob.fac(out)
        return out

    def basis_to_host(self, m):
        from ._ffi import call, hptr

        Vt = np.empty((m, self.n))
        call("eigd_d2h", self.ctx.h, hptr(Vt), self.V.ptr, 8 * self.n * m)
        return np.ascontiguousarray(Vt.T)


def _svqb(gram):
    """
    Orthonormalising transform of a block from its Gram matrix (Stathopoulos & Wu's SVQB): gram = U diag(w) U^T,
    X U diag(w)^-1/2 is orthonormal and X = (X U w^-1/2) C with C = diag(w)^1/2 U^T.  Returns (transform, C, rank
    deficient flags): directions whose eigenvalue is below 1e-28 of the largest carry nothing but rounding.
    """
    d = np.sqrt(np.maximum(np.diag(gram), np.finfo(float).tiny))
    w, U = np.linalg.eigh(gram / np.outer(d, d))           # scaled: equilibrates columns of very different length
    bad = w <= 1e-28 * max(w.max(), np.finfo(float).tiny)
    wi = np.where(bad, 1.0, w)
    Tr = (U / np.sqrt(wi)) / d[:, None]
    C = (np.sqrt(wi)[:, None] * U.T) * d[None, :]
    C[bad, :] = 0.0
    return Tr, C, bad


def small_eigh(T):
    """
    eigh of the projected matrix (a few dozen to a few hundred rows).  On a many-core host a threaded LAPACK spends ten
    times the arithmetic of such a matrix waking its threads (20 ms instead of 2 for 136 x 136 on 256 cores): a few
    threads at most.
    """
    global _THREADPOOLS
    if _THREADPOOLS is None:
        try:
            from threadpoolctl import ThreadpoolController

            _THREADPOOLS = ThreadpoolController()          # (finding the loaded BLAS libraries costs 1.3 ms: once)
        except ImportError:
            _THREADPOOLS = False
    if _THREADPOOLS is False:
        return np.linalg.eigh(T)
    with _THREADPOOLS.limit(limits=4):
        return np.linalg.eigh(T)


_THREADPOOLS = None


def ritz_bounds(C_last, S, p):
    """residual norms of the Ritz pairs of a block Lanczos basis: |C S[last p rows, j]|"""
    return np.linalg.norm(C_last @ S[-p:, :], axis=0)


def arpack_converged(bounds, theta, tol, beta_scale, basis_size=0):
    """
    ARPACK's test (dsconv): bound_i <= tol * max(eps^(2/3), |theta_i|).  The bounds are the coupling block times the
    last components of the eigenvectors of T, which LAPACK delivers with ABSOLUTE accuracy ~eps (growing with the order
    of T: measured 1e-13 at order 264 where order 129 reaches 1e-14): below that level a computed bound is rounding
    noise (with tol = eps the test would never be met for a cluster of Ritz values), so it counts as converged -- but
    never more than the value's own size allows: the floor of pair i is 64 eps max(1, order / 128) max(|coupling|,
    |theta_i|), not the largest Ritz value's.
    """
    eps = np.finfo(float).eps
    floor = 64.0 * eps * max(1.0, basis_size / 128.0) * np.maximum(abs(beta_scale), np.abs(theta))
    return bounds <= np.maximum(tol * np.maximum(eps ** (2.0 / 3.0), np.abs(theta)), floor)


def thick_restart_block_lanczos(be, k_want, m_int, p, tol, max_restarts, trace=None, k_strict=None, tol_extra=None,
                                report=None):
    """
    Thick-restart Lanczos with blocks of p vectors on the operator the backend ``be`` applies (shift-invert, B inner
    product), full reorthogonalisation.  p = 1 is the single-vector thick-restart Lanczos (= ARPACK's implicitly
    restarted Lanczos with exact shifts, in its explicit form).

    Backend (device: _BlockLanczosDevice; a numpy twin lives in the CPU tests):
        be.expand(c, p) -> (H (c x p), C (p x p)): W = OP basis[:, c-p:c], orthogonalised against basis[:, :c] with
                           coefficients H, W' = Q C with Q B-orthonormal, stored as basis[:, c:c+p]
        be.restart(S, c, keep, p): basis[:, :keep] = basis[:, :c] S, basis[:, keep:keep+p] = basis[:, c:c+p]
        optional: be.expand_begin(c, p) -> token, be.expand_end(token) -> (H, C): the same step, enqueued without a
                           host synchronisation; the tokens of a cycle are redeemed in order at its end
    The basis holds p start vectors when this is called.  Returns (T (c x c), C_last, c, nconv, n_restarts): the
    projected matrix of the final basis of c vectors, the coupling to the residual block basis[:, c:c+p] --
    OP basis[:, :c] = basis[:, :c] T + basis[:, c:c+p] C_last E_last^T -- and how many of the k_want wanted
    (largest |theta|) Ritz pairs pass ARPACK's test.
    ``k_strict`` / ``tol_extra``: only the first k_strict wanted pairs (the caller's N) are held to ``tol``; the
    further ones -- kept for the adjoint stage's deflation, which needs them to ~1e-10 -- to ``tol_extra``.  (A Ritz
    vector next to unconverged neighbours in a cluster carries eps |T| / gap of THEIR residuals: asking machine
    precision of the last pairs of a wanted set inside a cluster never ends.)
    ``report`` (a dict, optional) receives what the run achieved for the wanted pairs, largest |theta| first:
    ``theta``, ``bound`` (residual bounds |C s_last|) and ``by_noise_rule`` (accepted by one of the noise-level rules
    below rather than by ARPACK's test at the requested tolerance).
    """
    tols = np.full(k_want, tol)
    if k_strict is not None and tol_extra is not None:
        tols[k_strict:] = max(tol, tol_extra)
    best, stalled = np.inf, 0
    locked = np.empty(0)
    prev_theta, prev_bounds = np.empty(0), np.empty(0)
    c = p
    T = np.zeros((m_int, m_int))
    n_restarts = 0
    deferred = hasattr(be, "expand_begin")
    while True:
        # one cycle: expansions up to the internal basis size.  A backend with expand_begin / expand_end enqueues the whole
        # cycle on the device and hands the coefficient blocks over at its end (T is first needed for the Ritz step):
        # no host round trip between the sweeps of a cycle
        steps = []
        while True:
            steps.append((c, be.expand_begin(c, p) if deferred else be.expand(c, p)))
            if c + p > m_int:
                break
            c += p
        for cs, tok in steps:
            H, C = be.expand_end(tok) if deferred else tok
            T[:cs, cs - p:cs] = H
            T[cs - p:cs, :cs] = H.T
            T[cs - p:cs, cs - p:cs] = 0.5 * (H[cs - p:, :] + H[cs - p:, :].T)
            if cs + p <= m_int:
                T[cs:cs + p, cs - p:cs] = C
                T[cs - p:cs, cs:cs + p] = C.T
        theta, S = small_eigh(T[:c, :c])
        bounds = ritz_bounds(C, S, p)
        order = np.argsort(-np.abs(theta))              # which = "LM"
        wanted = order[:k_want]
        conv = arpack_converged(bounds[wanted], theta[wanted], tols, np.linalg.norm(C, 2), c)
        by_test = conv.copy()
        if locked.size:
            # A pair that passed its test in an earlier restart has been in the basis ever since (the thick restart keeps
            # the wanted Ritz vectors) and its true residual can only have shrunk; its computed bound sits at the noise
            # level of the projected eigenproblem and may land on either side of a test at machine precision from one
            # restart to the next.  It stays accepted while the bound is no worse than the noise can explain.
            was = (np.abs(theta[wanted][:, None] - locked[None, :]) <= 1e-9 * np.abs(locked)[None, :]).any(axis=1)
            conv |= was & (bounds[wanted] <= 1e-11 * np.abs(theta[wanted]))
        if prev_theta.size:
            # ... and a pair whose bound is already below 1e-11 |theta| and has stopped falling (a converging pair loses
            # one to two decades per restart; this one less than a factor of four since the last restart) has reached
            # the noise level of ITS part of the projected eigenproblem, which the a-priori floor of arpack_converged
            # underestimates inside clusters (order 200, shell box: bounds stand at 1e-13 |theta| for eight restarts).
            d = np.abs(theta[wanted][:, None] - prev_theta[None, :])
            j = np.argmin(d, axis=1)
            same = d[np.arange(len(wanted)), j] <= 1e-9 * np.abs(prev_theta[j])
            conv |= same & (bounds[wanted] <= 1e-11 * np.abs(theta[wanted])) & (bounds[wanted] > 0.25 * prev_bounds[j])
        prev_theta, prev_bounds = theta[wanted].copy(), bounds[wanted].copy()
        locked = theta[wanted][conv]
        nconv = int(np.count_nonzero(conv))
        worst = float(np.max(bounds[wanted] / np.abs(theta[wanted])))
        if trace is not None:
            trace(n_restarts, nconv, k_want, worst)
        # the bounds of a converged set sit at the noise level of the projected eigenproblem and no restart lowers them:
        # once they are all below 1e-11 |theta| and have not halved in eight restarts, that is what this basis can give
        if worst < 0.5 * best:
            best, stalled = worst, 0
        else:
            stalled += 1
        if nconv < k_want and worst <= 1e-11 and stalled >= 8:
            nconv = k_want
            conv[:] = True
        if nconv >= k_want or n_restarts >= max_restarts:
            if report is not None:
                report.update(theta=theta[wanted].copy(), bound=bounds[wanted].copy(), by_noise_rule=conv & ~by_test)
            return T[:c, :c].copy(), C, c, nconv, n_restarts
        # thick restart: the wanted pairs plus part of the unwanted ones as ARPACK's dsaup2 does (more of them as more
        # have converged); a whole number of blocks has to fit behind the kept vectors
        keep = k_want + min(nconv, (c - k_want) // 2)
        keep = max(min(keep, c - p), min(k_want + 1, c - p))
        keep = max(c - p * (-(-(c - keep) // p)), 1)
        selk = order[:keep]
        selk = selk[np.argsort(-theta[selk])]
        be.restart(S[:, selk], c, keep, p)
        T[:, :] = 0.0
        T[np.arange(keep), np.arange(keep)] = theta[selk]
        # (the next expansion recomputes the block column behind the kept vectors in full -- arrow + diagonal block --
        # by Gram-Schmidt)
        c = keep + p
        n_restarts += 1


def compress_to_single_vector_basis(be, T, C, c, p, m, tol, keep_first=0):
    """
    The contract of the reference's IRAM (what eigsh_mod extracts from ARPACK's work arrays and laa uses) is
    a basis of exactly m vectors with a rank-ONE residual, OP V = V T + f e_m^T.  A block run ends with a residual of
    rank p and a basis of another size: keep the converged Ritz vectors (their residuals vanish to rounding, so any
    basis that holds them satisfies the relation in their columns), put one vector of the residual block behind them
    and finish with single-vector Lanczos steps up to m.  Returns (T (m x m), beta_m).
    """
    theta, S = small_eigh(T)
    bounds = ritz_bounds(C, S, p)
    ok = arpack_converged(bounds, theta, tol, np.linalg.norm(C, 2), c)
    ok[np.argsort(-np.abs(theta))[:keep_first]] = True     # the wanted pairs (accepted by the run's own per-pair test)
    sel = np.flatnonzero(ok)
    sel = sel[np.argsort(-np.abs(theta[sel]))][: m - 1]
    sel = sel[np.argsort(-theta[sel])]
    kc = len(sel)
    be.restart(S[:, sel], c, kc, 1)
    Tm = np.zeros((m, m))
    Tm[np.arange(kc), np.arange(kc)] = theta[sel]
    cc = kc + 1
    beta = 0.0
    while True:
        H, Cb = be.expand(cc, 1)
        Tm[:cc, cc - 1] = H[:, 0]
        Tm[cc - 1, :cc] = H[:, 0]
        if cc == m:
            beta = float(Cb[0, 0])
            break
        Tm[cc, cc - 1] = Tm[cc - 1, cc] = Cb[0, 0]
        cc += 1
    return Tm, beta


class _BlockLanczosDevice:
    """
    device backend of thick_restart_block_lanczos: the basis V and B V as row-major 64-column panels
    (device.DevicePanels), blocks as n x p row-major
    """

    def __init__(self, prob, nvec):
        from .device import DevicePanels

        self.prob, self.ctx, self.n = prob, prob.ctx, prob.n
        self.V = DevicePanels(self.ctx, nvec, self.n)
        self.BV = DevicePanels(self.ctx, nvec, self.n)
        self.scratch = None
        self.sweeps = 0
        self.reorth_passes = 0
        self._arrow = True                                 # the next step's block couples to the whole basis

    def basis_to_host(self, m):
        return self.V.to_host(m)

    def _work(self, p):
        """work blocks of the current block size, kept between steps (no allocation inside the Lanczos loop)"""
        w = self.__dict__.setdefault("_wk", {})
        if p not in w:
            w[p] = tuple(self.ctx.empty(self.n, p) for _ in range(3))
        return w[p]

    def _coefficients(self, p, slot=0):
        """device block for the coefficients of one step (slot = its number inside the cycle): rows [0, nc) first pass,
        [nc, 2 nc) second pass, one row per panel whose first entry says whether that panel's second pass was applied,
        p rows for the orthonormalisation's C, one row whose first entry flags a numerically dependent block, one row
        for the squared B-norms of the block behind the first pass"""
        w = self.__dict__.setdefault("_cf", {})
        rows = 2 * self.V.ncols + self.V.npanels + p + 2
        if p not in w:
            nslots = max(2, -(-self.V.ncols // p))
            w[p] = (self.ctx.zeros(rows * nslots, p), rows, nslots)
        blk, rows, nslots = w[p]
        return blk.rows(slot * rows, (slot + 1) * rows)

    def _orthonormalise(self, X, BX, tmp):
        """B-orthonormalise the block X in place (SVQB, twice); BX <- B X; returns C with X_in = X_out C"""
        Ctot = np.eye(X.k)
        self.prob.opB.apply(X, BX)
        for _ in range(2):
            gram = X.tdot(BX)
            Tr, C, bad = _svqb(0.5 * (gram + gram.T))
            if bad.any():
                raise np.linalg.LinAlgError("Lanczos breakdown: the new block is linearly dependent on the basis "
                                            "(an invariant subspace was found)")
            for blk in (X, BX):                            # B (X Tr) = (B X) Tr: no further product with B
                blk.copy_from(tmp.add_product(blk, Tr, alpha=1.0, beta=0.0))
            Ctot = C @ Ctot
        self.prob.opB.apply(X, BX)                         # recomputed from the final X (not carried through two products)
        return Ctot

    def start(self, V0):
        X, BX, tmp = self._work(V0.shape[1])
        X.set(V0)
        self._orthonormalise(X, BX, tmp)
        self.V.set_block(0, X)
        self.BV.set_block(0, BX)

    def expand(self, c, p):
        return self.expand_end(self.expand_begin(c, p))

    def expand_begin(self, c, p):
        """
        Enqueue one Lanczos step -- sweep, Gram-Schmidt against the basis, B-orthonormalisation of the new block, its
        storage behind the basis -- without a host synchronisation.  The coefficients stay on the device (one slot of
        the cycle's coefficient block per step) until expand_end redeems the token.
        """
        X, BX, tmp = self._work(p)
        self._coefficients(p, 0)                          # (allocated on first use)
        nslots = self._cf[p][2]
        slot = self._next_slot = (getattr(self, "_next_slot", -1) + 1) % nslots
        Hd = self._coefficients(p, slot)
        last = self.BV.pieces(c - p, c)
        if len(last) == 1:
            self.prob.fac.apply_to(last[0][0], X)         # W = factor(B V_last) straight from the panel: one p-column sweep
        else:
            self.prob.fac(self.BV.get_block(c - p, p, out=X))
        self.sweeps += 1
        # Gram-Schmidt against the basis, panel by panel, coefficients kept on the device (no host round trip per
        # panel); then what that pass left along the basis is MEASURED panel by panel and removed where it matters:
        # |coefficient| > 1e-13 |x_b|_B, the B-norm of the column as the first pass left it (x_b^T B x_b formed on the
        # device) -- the relative B-orthogonality of the new vectors, whatever the scale of B; decided on the device.
        # The first pass takes the last two blocks only, except in the step behind a restart: the three-term
        # recurrence leaves OP v_j components of rounding size along everything older (the arrow of a thick restart is
        # in its first block column alone), so the pass over the whole basis is the measured one below -- and what it
        # finds is small against the block, which is the case in which one Gram-Schmidt pass is enough.
        nc, npan = self.V.ncols, self.V.npanels
        lo = max(0, c - 2 * p) if (tuning.lanczos_local_first_pass and not self._arrow) else 0
        self._arrow = False
        pieces = list(zip(self.V.pieces(0, c), self.BV.pieces(0, c)))
        if lo > 0:
            Hd.rows(0, lo).zero()
        for (Vb, a, b), (BVb, _, _) in zip(self.V.pieces(lo, c), self.BV.pieces(lo, c)):
            X.project_to(Vb, BVb, Hd.rows(lo + a, lo + b))
        r0 = 2 * nc + npan
        if lo > 0:
            # (what the local pass left along the older vectors is never below the 1e-13 of the rule underneath: applied
            # without the measurement, which would cost a product with B for the norms)
            for (Vb, a, b), (BVb, _, _) in pieces:
                X.project_to(Vb, BVb, Hd.rows(nc + a, nc + b))
        else:
            nb2 = Hd.rows(r0 + p + 1, r0 + p + 2)
            self.prob.opB.apply(X, BX)
            X.coldot_dev(BX, nb2)
            for q, ((Vb, a, b), (BVb, _, _)) in enumerate(pieces):
                X.project_to(Vb, BVb, Hd.rows(nc + a, nc + b), tol=1e-13, flag=Hd.rows(2 * nc + q, 2 * nc + q + 1).cols(0, 1),
                             norm2=nb2)
        # B-orthonormalisation of the block (SVQB, twice) on the device
        Cd, flag = Hd.rows(r0, r0 + p), Hd.rows(r0 + p, r0 + p + 1).cols(0, 1)
        if p <= 32:
            self.prob.opB.apply(X, BX)
            X.svqb_step(BX, Cd, True, flag)
            X.svqb_step(BX, Cd, False, flag, update_bx=False)
            self.prob.opB.apply(X, BX)                     # recomputed from the final X (not carried through two products)
            Chost = None
        else:
            Chost = self._orthonormalise(X, BX, tmp)
        self.V.set_block(c, X)
        self.BV.set_block(c, BX)
        return (Hd, c, p, lo > 0, Chost)

    def expand_end(self, token):
        """(H (c x p), C (p x p)) of a step enqueued by expand_begin; one synchronisation fetches its coefficient slot"""
        Hd, c, p, unconditional, Chost = token
        nc, npan = self.V.ncols, self.V.npanels
        Hall = Hd.get()
        H = Hall[:c].copy()
        again = False
        for q, (_, a, b) in enumerate(self.V.pieces(0, c)):
            if unconditional or Hall[2 * nc + q, 0] != 0.0:
                H[a:b] += Hall[nc + a: nc + b]
                again = True
        self.reorth_passes += int(again)
        r0 = 2 * nc + npan
        if Chost is not None:
            return H, Chost
        if Hall[r0 + p, 0] != 0.0 or not np.all(np.isfinite(Hall[r0:r0 + p])):
            raise np.linalg.LinAlgError("Lanczos breakdown: the new block is linearly dependent on the basis "
                                        "(an invariant subspace was found)")
        return H, Hall[r0:r0 + p].copy()

    def snapshot(self, c):
        """copy of the first c basis vectors (panels)"""
        from ._ffi import c_vp, call
        from .device import DevicePanels

        P = DevicePanels(self.ctx, c, self.n)
        call("eigd_d2d", self.ctx.h, c_vp(P.buf.ptr), c_vp(self.V.buf.ptr), 8 * P.npanels * self.n * P.PW)
        return P

    def restart(self, S, c, keep, p):
        from .device import DevicePanels

        ctx, n = self.ctx, self.n
        if self.scratch is None:
            self.scratch = (DevicePanels(ctx, self.V.ncols, n), DevicePanels(ctx, self.V.ncols, n))
        S = np.ascontiguousarray(S)
        for src, dst in zip((self.V, self.BV), self.scratch):
            if c <= 192 and tuning.lanczos_fused_restart:
                # new basis = V S straight into the other set of panels: the basis is read once per 80 new columns
                src.times_panels(dst, S, c)
            else:
                for a in range(0, keep, 64):               # one 64-column panel of it at a time, panel by panel of V
                    b = min(keep, a + 64)
                    src.times_into(dst.view(a, b), S[:, a:b], ns=c)
            dst.set_block(keep, src.get_block(c, p, out=self._work(p)[2]))   # the residual block follows the kept vectors
            src.swap(dst)
        self._arrow = True


class _AdjointAPI:
    """solve_adjoint / eval_adjoint_residual_norm / add_total_derivative (ref 1652-1870, 1988-2207)"""

    # subclasses provide: _lamN(), _warn_dl(), self._dev (LanczosDevice), self._m, self.Y, self.theta,
    # self.indices, self.T, self.sigma, self.mode, self.eig_atol, self._prob

    @property
    def V(self):
        """Lanczos basis as a numpy array (downloaded on first use)"""
        if self._V_host is None:
            self._V_host = self._dev.basis_to_host(self._nV)
        return self._V_host

    @V.setter
    def V(self, value):
        self._V_host = None if value is None else np.asarray(value)
        self._guess = None   # (a caller-supplied basis: the adjoint stage works from it alone)
        if value is not None and getattr(self, "_dev", None) is not None:
            # a caller replaced the basis: mirror it on the device
            Vh = np.asarray(value)
            from ._ffi import call, hptr
            from .device import DevicePanels

            if isinstance(self._dev.V, DevicePanels):
                self._dev.V.from_host(Vh[:, : min(Vh.shape[1], self._dev.V.ncols)])
                return
            nv = min(Vh.shape[1], self._dev.V.ns)
            Vt = np.ascontiguousarray(Vh[:, :nv].T)
            call("eigd_h2d", self._dev.ctx.h, self._dev.V.ptr, hptr(Vt), 8 * self._dev.n * nv)

    n_extra = 0

    def _set_extra_pairs(self, prob, dev, eigs, bounds, tol, beta_scale, m, extra_max, absolute_tol=None):
        """
        Ritz pairs of the final basis beyond the N requested ones, in the reference's sort order, that pass the
        eigensolver's own convergence test (ARPACK's for IRAM, |beta y_last| < tol for BasicLanczos): handed to the
        device problem for the adjoint stage's deflation.  None if the cut N | N+1 runs through a numerically repeated
        cluster (the reference warns there; nothing is added to what it does).
        """
        N = self.N
        self.n_extra = 0
        prob.set_extra(None, None)
        if extra_max <= 0 or not tuning.deflate_extra:
            return
        idx = []
        for q in range(N, min(N + extra_max, m)):
            j = self.indices[q]
            if absolute_tol is not None:
                good = bounds[j] < absolute_tol
            else:
                good = bool(arpack_converged(bounds[j:j + 1], self.theta[j:j + 1], tol, beta_scale)[0])
            if not good:
                break
            idx.append(j)
        if not idx or _is_close(eigs[self.indices[N - 1]], eigs[idx[0]], self.eig_atol):
            return
        Phix = prob.ctx.empty(prob.n, len(idx))
        dev.V.times_into(Phix, self.Y[:, idx], ns=m)
        prob.set_extra(Phix, eigs[idx])
        self.n_extra = len(idx)

    def _sync_phi(self):
        """the eigenvectors used by the adjoint stage are whatever self.Phi holds now"""
        if self._phi_token is not self.Phi:
            self._prob.set_phi(Phi_host=self.Phi)
            self._phi_token = self.Phi

    def _mode_columns(self, N, comm):
        if comm is None or comm.size == 1:
            return None
        return np.arange(comm.rank, N, comm.size)

    def solve_adjoint(self, Phib, method="sibk", psi=None, rtol=1e-10, atol=1e-30, lanczos_guess=True, comm=None,
                      **kwargs):
        """
        Same contract as the reference.  Extras: ``Phib`` may be a device block (then ``psi`` comes
        back as a device block); ``comm`` (rank/size/allreduce_sum) shards the independent per-mode
        solves -- the returned ``psi`` then holds this rank's columns and zeros elsewhere.
        """
        n = self.A.shape[1]
        lam = np.asarray(self._lamN(), dtype=float)
        N = len(lam)
        if method not in METHODS:
            raise ValueError(f"Unknown method {method!r}")
        if psi is not None and psi.shape != (n, N):
            raise ValueError(f"Initial guess must have the shape ({n},{N})")
        on_device = isinstance(Phib, DeviceBlock)
        if tuple(Phib.shape) != (n, N):
            raise ValueError(f"Right-hand-side must have the shape ({n},{N})")
        if method == "dl":
            self._warn_dl()
            lanczos_guess = False
        prob, ctx = self._prob, self._prob.ctx
        self._sync_phi()
        dPhib = Phib if on_device else ctx.twin_upload(Phib)     # (read only from here on)
        seq_sibk = method == "sibk" and (kwargs.get("bs_target", 1) != 1 or kwargs.get("update_guess", False))
        cols = None if (method == "dl" or seq_sibk) else self._mode_columns(N, comm)
        sel = np.arange(N) if cols is None else cols
        lam_c = lam[sel]
        dPhib_c = dPhib if cols is None else dPhib.gather_cols(cols)
        Y, theta, indices = np.asarray(self.Y), np.asarray(self.theta), np.asarray(self.indices)
        if lanczos_guess or method == "laa":
            g = getattr(self, "_guess", None)
            tok = getattr(self, "_guess_token", (None, None, None))
            if (g is not None and method != "laa" and tok[0] is self.Y and tok[1] is self.theta and tok[2] is self.indices
                    and self._phi_token is self.Phi):
                # the solver's own, larger basis (nobody replaced the Lanczos data since solve()): first guess only --
                # method "laa" itself answers from the m-vector contract basis
                Vg, mg, th_g, Y_g, idx_g, T_g, C_g, p_g = g
                if tuning.laa_relation:
                    # ... through the Lanczos relation of that basis: no product with B, no sweep for the guess
                    psi_c = adj._laa_relation_device(prob, Vg, mg, p_g, T_g, C_g, dPhib, lam, self.sigma, Y_g, th_g, idx_g,
                                                     self.mode, cols=cols)
                else:
                    psi_c = adj._laa_device(prob, Vg, mg, dPhib, lam, self.sigma, Y_g, th_g, idx_g, True, self.mode,
                                            cols=cols)
            else:
                psi_c = adj._laa_device(prob, self._dev.V, self._m, dPhib, lam, self.sigma, Y, theta, indices, True,
                                        self.mode, cols=cols)
        else:
            psi_c = ctx.zeros(n, len(sel))
        data = {}
        eig_atol = self.eig_atol
        kw = dict(kwargs)
        callback = kw.pop("callback", None)
        rnorm0 = adj._rnorm0(dPhib)
        G = Glo = None

        def refine_cols(Gm, R):
            """compensated entries of the repeated pairs in G[:, sel] = Phi^T R (pgmres / pcpg; R holds the columns sel)"""
            return adj.refine_repeated_entries(Gm, lam, prob.Phi, R, eig_atol, sign=1.0,
                                               col_of={int(c): q for q, c in enumerate(sel)})

        if method == "sibk":
            maxiter = kw.pop("maxiter", 50)
            nrestart = kw.pop("nrestart", 2)
            streams = kw.pop("streams", None)
            bs_target = kw.pop("bs_target", 1)
            update_guess = kw.pop("update_guess", False)
            kw.pop("eig_atol", None)
            if kw:
                raise TypeError(f"sibk() got unexpected keyword arguments {sorted(kw)}")
            # Converged eigenpairs beyond N (the eigensolver's basis holds some): deflated like the N requested ones --
            # the projectors of 1193 / 1232 / 1250-1257 take [Phi | Phix] -- and their share of psi added in closed form
            # below, as 385-389 does for j <= N.  psi is unique, the Krylov solve just no longer has to resolve the
            # eigenvalues closest above lam_N, which are what makes the high modes slow.
            prob.use_extra = prob.PhiD is not None
            prob.lam_phi = lam
            Cx = None
            if prob.use_extra and prob.PhiD.k <= 64:
                GD = -prob.PhiD.tdot(dPhib)                   # G (2109-2116) and the extra pairs' rows in one product
                G = np.ascontiguousarray(GD[:N])
                Cx = prob.extra_correction_coefficients(dPhib, lam, Gx=GD[N:])
            else:
                G = -prob.Phi.tdot(dPhib)
            Glo = adj.refine_repeated_entries(G, lam, prob.Phi, dPhib, eig_atol)
            early = {}

            def host_work():
                """what depends on G alone (385-389, 373-383: half a millisecond of host loops), done while the device
                runs the first Krylov steps"""
                if "corr" not in early:
                    early["corr"] = adj.correction_coefficients(lam, G, eig_atol, self.mode, Glo)

            try:
                if prob.use_extra:
                    psi_c.project(prob.Phix, prob.BPhix)      # the guess gives up its share along Phix
                if seq_sibk:
                    self.last_info = adj._sibk_sequential(prob, dPhib_c, psi_c, lam_c, self.sigma, rtol, atol, maxiter,
                                                          bs_target, update_guess, callback, nrestart)
                else:
                    self.last_info = adj._sibk_device(prob, dPhib_c, psi_c, lam_c, self.sigma, rtol, atol, maxiter,
                                                      nrestart, callback, rnorm0=rnorm0, streams=streams, host_work=host_work)
                host_work()
                Cc, data = early["corr"]
                if prob.use_extra:
                    if Cx is None:
                        Cx = prob.extra_correction_coefficients(dPhib, lam)
                    Cc_sub, Cx_sub = (Cc, Cx) if cols is None else (Cc[:, cols], Cx[:, cols])
                    if prob.PhiD.k <= 64:                     # both shares in one pass over [Phi | Phix]
                        psi_c.add_product(prob.PhiD, np.vstack([Cc_sub, Cx_sub]), alpha=1.0, beta=1.0)
                    else:
                        psi_c.add_product(prob.Phix, Cx_sub, alpha=1.0, beta=1.0)
                        adj._apply_correction(psi_c, prob.Phi, Cc, cols=cols)
                else:
                    adj._apply_correction(psi_c, prob.Phi, Cc, cols=cols)
                G = None                                      # (correction applied: nothing left for the common tail)
            finally:
                prob.use_extra = False
        elif method == "pgmres":
            maxiter = kw.pop("maxiter", 50)
            if kw:
                raise TypeError(f"pgmres() got unexpected keyword arguments {sorted(kw)}")
            (Gc, Gclo), self.last_info = adj._pgmres_device(prob, dPhib_c, psi_c, lam_c, rtol, atol, maxiter, callback,
                                                            rnorm0=rnorm0, refine=refine_cols)
            G, Glo = self._assemble_G(Gc, sel, N, comm), self._assemble_G(Gclo, sel, N, comm, optional=True)
        elif method == "pcpg":
            maxiter = kw.pop("maxiter", 100)
            reset = kw.pop("reset", 25)
            if kw:
                raise TypeError(f"pcpg() got unexpected keyword arguments {sorted(kw)}")
            (Gc, Gclo), self.last_info = adj._pcpg_device(prob, dPhib_c, psi_c, lam_c, rtol, atol, maxiter, reset,
                                                          callback, rnorm0=rnorm0, refine=refine_cols)
            G, Glo = self._assemble_G(Gc, sel, N, comm), self._assemble_G(Gclo, sel, N, comm, optional=True)
        elif method == "laa":
            G = -prob.Phi.tdot(dPhib)                      # ref 1772-1779 / 2109-2116 (G from Phib)
            Glo = adj.refine_repeated_entries(G, lam, prob.Phi, dPhib, eig_atol)
        elif method == "dl":
            if self.T is None:
                raise ValueError("dl needs the Lanczos matrix T")
            Vst = self._dev.V.as_stack(self._m) if hasattr(self._dev.V, "as_stack") else self._dev.V
            psi_c, data = adj._dl_device(prob, dPhib, lam, self.sigma, indices, Vst, self._m,
                                         np.asarray(self.T), Y, theta, eig_atol, self.mode)
        if G is not None:
            Cc, data = adj.correction_coefficients(lam, G, eig_atol, self.mode, Glo)
            adj._apply_correction(psi_c, prob.Phi, Cc, cols=cols)
        if cols is None:
            dpsi = psi_c
        else:
            dpsi = ctx.zeros(n, N)
            psi_c.scatter_cols_into(dpsi, cols)
        if on_device:
            return dpsi, data
        psi_host = dpsi.get()
        ctx.twin_adopt(psi_host, dpsi)                           # (handed back to add_total_derivative as it is: no upload then)
        return psi_host, data

    @staticmethod
    def _assemble_G(Gc, sel, N, comm, optional=False):
        """the rank's columns of G put in place and summed over the ranks; ``optional``: Gc may be None (no repeated pair:
        decided from the replicated eigenvalues, so every rank takes the same branch)"""
        if Gc is None and optional:
            return None
        G = np.zeros((N, N))
        G[:, sel] = Gc
        if comm is not None and comm.size > 1:
            G = comm.allreduce_sum(G)
        return G

    def eval_adjoint_residual_norm(self, Phib, psi, b_ortho=False):
        self._sync_phi()
        ctx = self._prob.ctx
        dPhib = Phib if isinstance(Phib, DeviceBlock) else ctx.twin_upload(Phib)
        dpsi = psi if isinstance(psi, DeviceBlock) else ctx.twin_upload(psi)
        return adj._residual_norm_device(self._prob, np.asarray(self._lamN(), dtype=float), dPhib, dpsi, b_ortho)

    def add_total_derivative(self, lamb, Phib, psi, dAdx, dBdx, dfdx, adj_corr_data={}, deriv_type="vector",
                             comm=None):
        """ref 1830-1870 / 2167-2207; with ``comm`` each rank sums its own modes, then one all-reduce of dfdx"""
        lam = np.asarray(self._lamN(), dtype=float)
        n, N = self.A.shape[1], len(lam)
        for arr, what in ((psi, "Eigenvectors"), (Phib, "Right-hand-side")):
            if tuple(arr.shape) != (n, N):
                raise ValueError(f"{what} must have the shape ({n},{N})")
        if deriv_type not in ("vector", "tensor"):
            return dfdx
        self._sync_phi()
        ctx = self._prob.ctx
        dPhib = Phib if isinstance(Phib, DeviceBlock) else ctx.twin_upload(Phib)   # (both read only below)
        dpsi = psi if isinstance(psi, DeviceBlock) else ctx.twin_upload(psi)
        cols = self._mode_columns(N, comm)
        return adj._total_derivative_device(self._prob.Phi, dPhib, dpsi, lam, lamb, dAdx, dBdx, dfdx, adj_corr_data,
                                            self.mode, deriv_type, cols, Phi_host=self.Phi, comm=comm)


class BasicLanczos(_AdjointAPI):
    """Un-restarted shift-invert Lanczos with B-orthogonalisation (ref 1331-1650); real dtype."""

    def __init__(self, N=10, m=60, tol=1e-14, Ntarget=None, eig_atol=1e-5, mode="normal", ortho_type="full",
                 ctx=None):
        self.N = N
        self.m_max = m
        self.tol = tol
        self.Ntarget = Ntarget
        self.eig_atol = eig_atol
        self.mode = mode
        self.ortho_type = ortho_type
        self._ctx = ctx
        if self.Ntarget is not None and not isinstance(self.Ntarget, int):
            raise ValueError("Ntarget must be an integer or None")
        if ortho_type not in ("full", "selective"):
            raise ValueError(f"Unknown ortho_type {ortho_type!r}")
        if mode not in ("normal", "buckling"):
            raise ValueError(f"Unknown mode {mode!r}")
        self._V_host = None
        self._phi_token = None
        self.T = None

    def _lamN(self):
        return self.lam0

    def _warn_dl(self):
        pass

    def _reduced(self, m):
        """eigh of the leading m x m tridiagonal + sort (ref 1416-1439)"""
        T = np.diag(self.alpha[:m]) + np.diag(self.beta[: m - 1], 1) + np.diag(self.beta[: m - 1], -1)
        theta, Y = np.linalg.eigh(T)
        lam, indices = ritz_to_eigs(theta, self.sigma, self.mode)
        return theta, Y, T, lam, indices

    @staticmethod
    def _converged(beta_last, Yrow, N, tol):
        """leading run of sorted Ritz pairs with |beta y_last| < tol (ref 1441-1451)"""
        count = 0
        for e in np.abs(beta_last * Yrow):
            if e < tol:
                count += 1
            else:
                break
        return count >= N

    def solve(self, A, B, factor, sigma):
        n = _check_shapes(A, B, factor)
        if np.issubdtype(np.dtype(A.dtype), np.complexfloating) or np.issubdtype(np.dtype(B.dtype), np.complexfloating):
            return self._solve_complex_step(A, B, factor, sigma, n)
        self.factor = aslinearoperator(factor)
        self.B = aslinearoperator(B)
        self.A = aslinearoperator(A)
        self.sigma = sigma
        ctx = self._ctx or (factor.ctx if isinstance(factor, SpLuOperator) else default_context())
        prob = DeviceProblem(ctx, A, B, factor, self.mode)
        self._prob = prob
        mm = self.m_max
        dev = _LanczosDevice(prob, mm + 1)
        self._dev = dev
        self.alpha = np.zeros(mm)
        self.beta = np.zeros(mm)

        v0 = np.random.default_rng(12345).uniform(size=n, low=-1.0, high=1.0)  # ref 1514-1515
        v = ctx.from_host(v0)
        dev.normalize_into(v, 0)

        Nchk = self.N if self.Ntarget is None else self.Ntarget
        self.m = mm
        S = BS = None
        for i in range(1, mm + 1):
            dev.apply_op(i - 1, v)                                         # ref 1524
            if i > 1:
                v.assign_lincomb([(1.0, v), (-self.beta[i - 2], dev.V[i - 2])])  # ref 1526
            if self.ortho_type == "full":
                h = dev.orthogonalize(v, i)                                # ref 1529-1534
                self.alpha[i - 1] = h[i - 1]
            else:
                j0 = max(0, i - 2)                                         # ref 1563: the two previous vectors
                h1 = dev.BV.dot(v, ns=i - j0, j0=j0)
                dev.V.axpy_into(v, h1, alpha=-1.0, j0=j0)
                self.alpha[i - 1] = h1[-1, 0]
                if S is not None and S.k > 0:                              # ref 1571-1574
                    hs = BS.tdot(v)
                    v.add_product(S, hs, alpha=-1.0, beta=1.0)
            self.beta[i - 1] = dev.normalize_into(v, i)                    # ref 1537-1538
            if i >= 2:
                theta, Y, T, lam, indices = self._reduced(i)
                Y0 = Y[:, indices]
                if self._converged(self.beta[i - 1], Y0[i - 1, :], Nchk, self.tol):
                    self.m = i
                    break
                if self.ortho_type == "selective":                         # ref 1596-1605
                    errs = np.abs(self.beta[i - 1] * Y0[i - 1, :])
                    conv = [j for j in range(i) if errs[j] < np.sqrt(self.tol)]
                    if conv:
                        S = ctx.empty(n, len(conv))
                        dev.V.times_into(S, Y0[:, conv], ns=i)
                        BS = prob.opB.apply(S)
                    else:
                        S = BS = None

        self.theta, self.Y, self.T, self.lam, self.indices = self._reduced(self.m)
        if self.Ntarget is not None:                                       # ref 1615-1625
            self.N = self.Ntarget
            while self.N < self.m and _is_close(self.lam[self.indices[self.N - 1]], self.lam[self.indices[self.N]],
                                                self.eig_atol):
                self.N += 1
        elif _is_close(self.lam[self.indices[self.N - 1]], self.lam[self.indices[self.N]], self.eig_atol):
            warnings.warn(f"BasicLanczos: Ritz values {self.N} and {self.N+1} are numerically repeated.")
        sel = self.indices[: self.N]
        self.lam0 = self.lam[sel]
        self.Y0 = self.Y[:, sel]
        self.eig_res = np.abs(self.beta[-1] * self.Y0[-1, :])              # ref 1640-1645
        self.fail = bool(np.any(self.eig_res > self.tol))
        dPhi = ctx.empty(n, self.N)
        dev.V.times_into(dPhi, self.Y0, ns=self.m)                         # ref 1648
        prob.set_phi(Phi_dev=dPhi)
        self.Phi = dPhi.get()
        self._phi_token = self.Phi
        # Ritz pairs beyond N that pass the solver's own test |beta y_last| < tol: deflated by the adjoint stage too
        if self.tol > 0:
            bounds = np.abs(self.beta[self.m - 1] * self.Y[self.m - 1, :])
            self._set_extra_pairs(prob, dev, self.lam, bounds, None, None, self.m, max(0, min(self.N // 4, 128 - self.N)),
                                  absolute_tol=self.tol)
        self._m = self.m
        self._nV = self.m_max + 1
        self._V_host = None
        return self.lam0, self.Phi


def _solve_complex_step(self, A, B, factor, sigma, n):
    """
    The reference's complex-step evaluation (ref 1453-1650 with complex A, B and a complex SuperLU,
    examples/buckling.py:1014-1023): the same recurrence in dual-number arithmetic on the device.  Eigenvalues,
    eigenvectors and Lanczos coefficients come back complex, imaginary part = step * forward derivative.  Forward
    evaluation only, as in the reference (its adjoint stage is never run on complex data).
    """
    from scipy import sparse

    if not (isinstance(factor, SpLuOperator) and factor._imag_dev is not None):
        raise TypeError("the complex-step path needs an eigd_amd.SpLuOperator built on the complex shifted matrix")
    if not sparse.issparse(B):
        raise TypeError("the complex-step path needs B as a scipy sparse matrix")
    Bc = sparse.csr_matrix(B).astype(np.complex128)
    Bc.sort_indices()
    Br = sparse.csr_matrix((Bc.data.real.copy(), Bc.indices, Bc.indptr), shape=Bc.shape)
    Bi = sparse.csr_matrix((Bc.data.imag.copy(), Bc.indices, Bc.indptr), shape=Bc.shape)
    self.factor, self.B, self.A, self.sigma = aslinearoperator(factor), aslinearoperator(B), aslinearoperator(A), sigma
    ctx = self._ctx or factor.ctx
    mm = self.m_max
    dev = _DualLanczosDevice(ctx, Br, Bi, factor, n, mm + 1)
    self._dev, self._prob = dev, None
    self.alpha = np.zeros(mm, dtype=complex)
    self.beta = np.zeros(mm, dtype=complex)

    v0 = np.random.default_rng(12345).uniform(size=n, low=-1.0, high=1.0)  # ref 1514-1515
    vr, vi = ctx.from_host(v0), ctx.zeros(n, 1)
    dev.normalize_into(vr, vi, 0)

    def reduced(m):  # ref 1416-1439 with the complex _eigh
        T = np.diag(self.alpha[:m]) + np.diag(self.beta[: m - 1], 1) + np.diag(self.beta[: m - 1], -1)
        theta, Y = complex_step_eigh(T)
        lam, indices = ritz_to_eigs(theta, self.sigma, self.mode)
        return theta, Y, T, lam, indices

    Nchk = self.N if self.Ntarget is None else self.Ntarget
    self.m = mm
    Sr = Si = BSr = BSi = None
    for i in range(1, mm + 1):
        dev.apply_op(i - 1, vr, vi)                                        # ref 1524
        if i > 1:
            dev.axpy_basis(vr, vi, -self.beta[i - 2], i - 2)              # ref 1526
        if self.ortho_type == "full":
            h = dev.project_out(vr, vi, 0, i)                             # ref 1529-1534
            self.alpha[i - 1] = h[i - 1]
        else:
            j0 = max(0, i - 2)                                             # ref 1563: the two previous vectors
            h = dev.project_out(vr, vi, j0, i - j0)
            self.alpha[i - 1] = h[-1]
            if Sr is not None:                                             # ref 1571-1574
                for _ in range(2):
                    hr = BSr.tdot(vr)
                    hi = BSr.tdot(vi) + BSi.tdot(vr)
                    vi.add_product(Sr, hi, alpha=-1.0, beta=1.0)
                    vi.add_product(Si, hr, alpha=-1.0, beta=1.0)
                    vr.add_product(Sr, hr, alpha=-1.0, beta=1.0)
        self.beta[i - 1] = dev.normalize_into(vr, vi, i)                   # ref 1537-1538
        if i >= 2:
            theta, Y, T, lam, indices = reduced(i)
            Y0 = Y[:, indices]
            if self._converged(self.beta[i - 1], Y0[i - 1, :], Nchk, self.tol):
                self.m = i
                break
            if self.ortho_type == "selective":                             # ref 1596-1605
                errs = np.abs(self.beta[i - 1] * Y0[i - 1, :])
                conv = [j for j in range(i) if errs[j] < np.sqrt(self.tol)]
                if conv:
                    S = dev.times(Y0[:, conv], i)
                    Sr, Si = ctx.from_host(np.ascontiguousarray(S.real)), ctx.from_host(np.ascontiguousarray(S.imag))
                    BSr, BSi = ctx.empty(n, len(conv)), ctx.empty(n, len(conv))
                    tmp = ctx.empty(n, len(conv))
                    dev.Br.apply(Sr, BSr)
                    dev.Br.apply(Si, BSi)
                    dev.Bi.apply(Sr, tmp)
                    BSi.assign_lincomb([(1.0, BSi), (1.0, tmp)])
                else:
                    Sr = Si = BSr = BSi = None

    self.theta, self.Y, self.T, self.lam, self.indices = reduced(self.m)
    if self.Ntarget is not None:                                           # ref 1615-1625
        self.N = self.Ntarget
        while self.N < self.m and _is_close(self.lam[self.indices[self.N - 1]].real, self.lam[self.indices[self.N]].real,
                                            self.eig_atol):
            self.N += 1
    elif _is_close(self.lam[self.indices[self.N - 1]].real, self.lam[self.indices[self.N]].real, self.eig_atol):
        warnings.warn(f"BasicLanczos: Ritz values {self.N} and {self.N+1} are numerically repeated.")
    sel = self.indices[: self.N]
    self.lam0 = self.lam[sel]
    self.Y0 = self.Y[:, sel]
    self.eig_res = np.abs(self.beta[-1] * self.Y0[-1, :])                  # ref 1640-1645
    self.fail = bool(np.any(self.eig_res > self.tol))
    self.Phi = dev.times(self.Y0, self.m)                                  # ref 1648
    self._phi_token = self.Phi
    self._m = self.m
    self._nV = self.m_max + 1
    self._V_host = None
    return self.lam0, self.Phi


BasicLanczos._solve_complex_step = _solve_complex_step


class IRAM(_AdjointAPI):
    """
    Restarted shift-invert Lanczos with the reference's IRAM surface (ref 1873-1986).
    ``m = max(20, 2N+1, m)`` as the reference (1895-1898); ``tol = 0`` means machine precision,
    ARPACK's convention (dsaupd).
    """

    def __init__(self, N=10, m=None, eig_atol=1e-5, tol=0.0, mode="normal", maxiter=None, ctx=None, extra=None):
        self.N = N
        self.extra = extra   # converged eigenpairs beyond N kept for the adjoint stage's deflation (None: automatic)
        self.m = int(max(20, 2 * N + 1) if m is None else max(20, 2 * N + 1, m))
        self.tol = tol
        self.eig_atol = eig_atol
        self.mode = mode
        self.maxiter = maxiter
        self._ctx = ctx
        if mode not in ("normal", "buckling"):
            raise ValueError(f"Unknown mode {mode!r}")
        self._V_host = None
        self._phi_token = None

    def _lamN(self):
        return self.lam

    def _warn_dl(self):
        warnings.warn('Adjoint method "dl" is not recommended for the ARPACK IRAM eigenvalue sovler.')

    def _block_plan(self, n):
        """
        (block size p, extra pairs to converge, internal basis size).  A triangular sweep of 4-8 columns costs what a
        one-column sweep costs (it is bound by the latency chain through the tree levels, not by bytes), so large
        problems run the restarted Lanczos with blocks: several new basis vectors per sweep.  tuning.iram_block /
        tuning.iram_extra override (1 / 0 = the single-vector solver on exactly m vectors).
        """
        N, m = self.N, self.m
        p = int(tuning.iram_block) or (8 if n >= 200_000 else (4 if n >= 50_000 else 1))
        extra = tuning.iram_extra
        # (automatic: as many extra pairs as requested ones, 32 at most -- what fits one 64-column projector next to 32
        # modes; converging more costs the eigensolve more than it saves the adjoint stage)
        extra = int(extra) if extra is not None else (self.extra if self.extra is not None else (0 if p == 1 else min(N, 32)))
        extra = max(0, min(extra, m - 1 - N, 128 - N))  # room in the m-vector contract basis; one fused projector call
        if p == 1:
            return 1, extra, m
        k_want = N + extra
        m_int = int(tuning.iram_basis) or max(m, int(np.ceil(tuning.iram_basis_factor * k_want)) + p)
        m_int = p * (-(-m_int // p))
        if m_int + p > n:
            return 1, min(extra, max(0, m - 1 - N)), m
        return p, extra, m_int

    def solve(self, A, B, factor, sigma):
        n = _check_shapes(A, B, factor)
        if np.issubdtype(np.dtype(A.dtype), np.complexfloating):
            raise TypeError("Input matrix is not real-valued.")
        self.factor = aslinearoperator(factor)
        self.B = aslinearoperator(B)
        self.A = aslinearoperator(A)
        self.sigma = sigma
        ctx = self._ctx or (factor.ctx if isinstance(factor, SpLuOperator) else default_context())
        prob = DeviceProblem(ctx, A, B, factor, self.mode)
        self._prob = prob
        m, k = self.m, self.N
        if not (k < m <= n):
            raise ValueError("ncv must be k<ncv<=n")  # scipy's message for the same condition
        p, extra, m_int = self._block_plan(n)
        k_want = k + extra
        dev = _BlockLanczosDevice(prob, m_int + p)
        self._dev = dev
        eps = np.finfo(float).eps
        tol = self.tol if self.tol > 0 else eps
        max_restarts = self.maxiter if self.maxiter is not None else min(10 * n, 1000)  # (scipy's default for eigsh: 10 n)
        tol_x = float(tuning.iram_extra_tol)

        # (fixed start block; tuning.iram_seed: development aid for noise studies -- the spread of a result over start vectors)
        V0 = np.random.default_rng(int(tuning.iram_seed)).uniform(size=(n, p), low=-1.0, high=1.0)
        dev.start(V0)

        def trace(r, nconv, kw, worst):
            if _TRACE:
                print(f"[iram] restart {r}: {nconv}/{kw} converged (block {p}, basis {m_int}), worst bound / |theta| "
                      f"{worst:.2e}", flush=True)

        run = {}
        T, C, c, nconv, self.n_restarts = thick_restart_block_lanczos(dev, k_want, m_int, p, tol, max_restarts, trace,
                                                                      k_strict=k, tol_extra=tol_x, report=run)
        if nconv < k_want:
            # the extra pairs are an internal acceleration of the adjoint stage: only the N requested ones decide
            theta_c, S_c = np.linalg.eigh(T)
            order = np.argsort(-np.abs(theta_c))[:k]
            okN = arpack_converged(ritz_bounds(C, S_c, p)[order], theta_c[order], tol, np.linalg.norm(C, 2), c)
            if not okN.all():
                from scipy.sparse.linalg import ArpackNoConvergence

                raise ArpackNoConvergence(f"No convergence ({self.n_restarts} restarts, {int(okN.sum())}/{k} eigenvectors "
                                          "converged)", None, None)
        self._guess = None
        if p == 1 and c == m:
            beta_m = float(C[0, 0])
        else:
            if tuning.laa_internal:
                # the block run's own basis (c vectors, more than the m of the contract) holds partly converged
                # approximations of the eigenvectors beyond the converged ones: kept for the Lanczos adjoint
                # approximation (laa's closed form only uses V^T B V = I and T = V^T B OP V), the better first guess
                th_i, Y_i = small_eigh(T)
                _, idx_i = ritz_to_eigs(th_i, sigma, self.mode)
                # (with the residual block behind the c vectors and the coupling: the Lanczos relation
                # OP V = V T + Q C E_last^T lets the guess be formed without applying the factor)
                self._guess = (dev.snapshot(c + p), c, th_i, Y_i, idx_i, T.copy(), np.array(C, dtype=float), p)
            T, beta_m = compress_to_single_vector_basis(dev, T, C, c, p, m, max(tol, tol_x) if extra > 0 else tol,
                                                        keep_first=min(k_want, m - 1) if nconv >= k_want else k)
        self.block_size, self.internal_basis = p, m_int
        self.sweeps = dev.sweeps
        self.T = T
        self.theta, self.Y = np.linalg.eigh(self.T)                        # ref 1958
        eigs, self.indices = ritz_to_eigs(self.theta, sigma, self.mode)    # ref 1960-1965
        if _is_close(eigs[self.indices[self.N - 1]], eigs[self.indices[self.N]], self.eig_atol):
            warnings.warn(f"IRAM: Ritz values {self.N} and {self.N+1} are numerically repeated.")
        sel = self.indices[: self.N]
        self.lam = eigs[sel]
        dPhi = ctx.empty(n, self.N)
        dev.V.times_into(dPhi, self.Y[:, sel], ns=m)   # Phi = V Y[:, indices[:N]]: signs agree by construction (ref 1976-1984)
        prob.set_phi(Phi_dev=dPhi)
        self.Phi = dPhi.get()
        self._phi_token = self.Phi
        bounds = np.abs(beta_m * self.Y[m - 1, :])
        self.eig_res = bounds[sel]
        # What the restarted run itself achieved for these pairs (the m-vector contract basis is rebuilt around the kept
        # Ritz vectors with their couplings set to zero: its own bounds say nothing about them), and the TRUE residuals
        # |OP phi - theta phi|_B of the returned pairs, from one N-column application of the operator -- the acceptance
        # rules of the block run are checked against the real thing before anybody builds on these pairs.
        th_sel = self.theta[sel]
        hit = np.zeros(self.N, dtype=bool)
        if run:
            jm = np.argmin(np.abs(th_sel[:, None] - run["theta"][None, :]), axis=1)
            hit = np.abs(run["theta"][jm] - th_sel) <= 1e-9 * np.abs(th_sel)
            self.eig_res = np.where(hit, np.maximum(self.eig_res, run["bound"][jm]), self.eig_res)
            self.eig_accepted_at_noise_level = int(np.count_nonzero(run["by_noise_rule"][jm] & hit))
        Wr = ctx.empty(n, self.N)
        prob.fac.apply_to(prob.BPhi, Wr, count=0)          # OP Phi = factor(B Phi)   (not counted: a check, not a solve)
        Wr.assign_lincomb([(1.0, Wr), (-th_sel, dPhi)])
        self.eig_res_true = np.sqrt(np.maximum(Wr.coldot(prob.opB.apply(Wr)), 0.0))
        del Wr
        self.eig_res = np.maximum(self.eig_res, self.eig_res_true)
        rel = self.eig_res_true / np.maximum(np.abs(th_sel), np.finfo(float).tiny)
        if np.any(hit & (rel > self.eig_atol)):            # (pairs the run declared converged; a selection of other
            from scipy.sparse.linalg import ArpackNoConvergence   # pairs -- shift on the wrong side -- is warned about below)

            raise ArpackNoConvergence(f"IRAM: the returned Ritz pairs are not eigenpairs (largest true residual "
                                      f"|OP phi - theta phi|_B / |theta| = {rel.max():.1e})", self.lam, self.Phi)
        tol_user = self.tol if self.tol > 0 else 0.0
        if tol_user > 0.0 and np.any(rel > 10.0 * max(tol_user, 64.0 * eps)):
            warnings.warn(f"IRAM: residual of the returned pairs {rel.max():.1e} |theta| above the requested tol = {tol_user:.1e} "
                          "(pairs are accepted at the noise level of the projected problem, 1e-11 |theta| at worst)")
        if np.any(self.eig_res > 1e-6 * np.maximum(np.abs(self.theta[sel]), 1.0)):
            # the restarts converge the Ritz values of largest magnitude; the reference then selects by eigenvalue order
            # (1960-1965): with a shift on the wrong side of the wanted eigenvalues these are different pairs
            warnings.warn("IRAM: the selected Ritz pairs are not the converged ones (largest residual "
                          f"{self.eig_res.max():.1e}): the shift is not next to the wanted eigenvalues")
        # converged pairs beyond the N requested ones (the next eigenvalues in the reference's sort order): the adjoint
        # stage deflates them too (solve_adjoint) -- they are what makes the high modes of the block slow to converge
        self._set_extra_pairs(prob, dev, eigs, bounds, max(tol, tol_x), abs(beta_m), m, extra)
        self._m = m
        self._nV = m
        self._V_host = None
        self._guess_token = (self.Y, self.theta, self.indices)
        return self.lam, self.Phi


def eigsh_mod(A, k=6, M=None, sigma=None, which="LM", v0=None, ncv=None, maxiter=None, tol=0, return_eigenvectors=True,
              Minv=None, OPinv=None, mode="normal"):
    """
    The reference's extended ``eigsh`` (eigd/arpack.py:104-442, star-exported through eigd/__init__.py:3), for the one
    way eigd calls it (eigenvector_derivatives.py:1944-1954): shift-invert with the factor as ``OPinv``, ``which="LM"``,
    ``mode`` "normal" (``A x = lambda M x``) or "buckling".  Returns what the reference returns there:
    ``d (k,), z (n, k), Tm (ncv, ncv), v (n, ncv)`` -- eigenvalues, eigenvectors, the projected matrix and the
    M-orthonormal Lanczos basis with ``OP v = v Tm + f e_m^T`` (restarted Lanczos on the device, see ``IRAM``).
    Other argument combinations belong to scipy's ``eigsh`` and are not part of this path.
    """
    if sigma is None or OPinv is None or M is None:
        raise ValueError("eigsh_mod: only the shift-invert path of eigd is implemented (sigma, M and OPinv are required)")
    if which != "LM" or Minv is not None:
        raise ValueError("eigsh_mod: only which='LM' without Minv is implemented")
    if mode not in ("normal", "buckling"):
        raise ValueError(f"Unknown mode {mode!r}")
    if v0 is not None:
        warnings.warn("eigsh_mod: v0 is ignored (fixed start vector, as the IRAM class)")
    solver = IRAM(N=k, m=ncv, tol=tol, mode=mode, maxiter=maxiter)
    if ncv is not None:
        solver.m = int(ncv)   # the function does not apply the class's max(20, 2N+1, m) rule (ref 1895-1898)
    d, z = solver.solve(A, M, OPinv, sigma)
    if not return_eigenvectors:
        return d
    return d, z, solver.T, solver.V
