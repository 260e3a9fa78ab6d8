"""
Adjoint solvers of the eigenvector-derivative path, driven from Python over the HIP kernels.

Public functions keep the reference signatures (eigd/eigenvector_derivatives.py):
``laa`` 394, ``dl`` 526, ``pcpg`` 699, ``pgmres`` 872, ``sibk`` 1052,
``generate_adjoint_correction`` 303, ``add_eig_total_derivative`` 33,
``eval_adjoint_residual_norm`` 185, ``are_eigenvalues_repeated`` 284.

MI355X design: the N adjoint systems are independent (for ``sibk`` with bs_target=1 and
update_guess=False, for ``pgmres`` and ``pcpg``), so all modes advance in lock step and every
operator application is ONE multi-RHS call: a k-column triangular sweep, a k-column SpMM, one
fused projection, one batched Gram-Schmidt pass.  The factor is streamed once per Krylov
iteration instead of once per mode and iteration -- SuperLU gains nothing from multiple
right-hand sides, the GPU sweep does.  Columns never mix: a mode's
result depends on the other modes in its block only through the shape of the reduction trees
(rounding level), so mode-sharded runs agree with the single-GPU run to ~1e-13.

Small dense work (Hessenberg least squares, index sets, xi/eta) stays on the host in numpy,
exactly as in the reference.
"""

import numpy as np

from . import tuning
from .device import writable_result
from .operators import DeviceOperator, FactorApply, SpLuOperator

MODES = ("normal", "buckling")


def _check_mode(mode):
    if mode not in MODES:
        raise ValueError(f"Unknown mode {mode!r}")


def _is_close(a, b, atol=1e-5):
    return bool(np.fabs(a - b) < atol)


def are_eigenvalues_repeated(lam, atol=1e-5):
    """ref 284-300"""
    return any(_is_close(lam[i], lam[i + 1], atol) for i in range(len(lam) - 1))


# ---------------------------------------------------------------------------
# device problem: A, B, factor and the eigenvector blocks resident in HBM
# ---------------------------------------------------------------------------
class DeviceProblem:
    def __init__(self, ctx, A, B, factor, mode):
        self.ctx = ctx
        self.opA = A if isinstance(A, DeviceOperator) else DeviceOperator(ctx, A)
        self.opB = B if isinstance(B, DeviceOperator) else DeviceOperator(ctx, B)
        self.fac = factor if isinstance(factor, FactorApply) else (FactorApply(ctx, factor) if factor is not None else None)
        self.mode = mode
        self.n = self.opB.shape[0]
        self.Phi = None
        self.BPhi = None
        # converged eigenpairs beyond the N requested ones (from the eigensolver's basis): the projected Krylov solvers
        # may deflate them as well -- [Phi | Phix] in the projectors -- and add their share of psi in closed form
        self.Phix = self.BPhix = self.lam_x = None
        self.PhiD = self.BPhiD = None
        self.use_extra = False
        self.lam_phi = None   # eigenvalues of the columns of Phi when the caller knows them (all N, also on a rank that solves a share)
        self._uscale = {}     # largest Euclidean column norm of B Phi ("N") / B [Phi | Phix] ("D"), formed on first use

    def on(self, ctx):
        """the same problem (shared device data) with work enqueued on another context / stream of the device"""
        import copy

        other = copy.copy(self)
        other.ctx = ctx
        return other

    def set_phi(self, Phi_host=None, Phi_dev=None):
        self.Phi = Phi_dev if Phi_dev is not None else self.ctx.from_host(Phi_host)
        self.BPhi = self.opB.apply(self.Phi)
        self._uscale = {}
        self._rebuild_deflation()

    def set_extra(self, Phix_dev, lam_x):
        """eigenvectors (B-orthonormal, B-orthogonal to Phi) and eigenvalues of further converged pairs; None clears"""
        self.Phix, self.lam_x = Phix_dev, (None if lam_x is None else np.asarray(lam_x, dtype=float))
        self.BPhix = None if Phix_dev is None else self.opB.apply(Phix_dev)
        self._rebuild_deflation()

    def _rebuild_deflation(self):
        self.PhiD = self.BPhiD = None
        self._uscale.pop("D", None)
        if self.Phix is None or self.Phi is None or self.Phix.k == 0:
            return
        # the caller may have replaced Phi (signs, rotations inside a cluster, another set altogether): the extra vectors
        # stay in use only while they are B-orthogonal to it
        if np.max(np.abs(self.BPhix.tdot(self.Phi))) > 1e-9:
            return
        N, nx = self.Phi.k, self.Phix.k
        self.PhiD, self.BPhiD = self.ctx.empty(self.n, N + nx), self.ctx.empty(self.n, N + nx)
        for dst, a, b in ((self.PhiD, self.Phi, self.Phix), (self.BPhiD, self.BPhi, self.BPhix)):
            dst.cols(0, N).copy_from(a)
            dst.cols(N, N + nx).copy_from(b)

    def _projector(self):
        if self.use_extra and self.PhiD is not None:
            return self.BPhiD, self.PhiD
        return self.BPhi, self.Phi

    def project_r(self, X):
        """X <- X - BPhi (Phi^T X)   (residual-side projector P)"""
        U, V = self._projector()
        return X.project(U, V)

    def project_r_norm2(self, X, tol=0.0, requested_only=False):
        """project_r followed by the squared column norms of the result (device block; see DeviceBlock.project_norm2);
        ``requested_only``: against the N requested pairs even while the extra ones are deflated"""
        U, V = (self.BPhi, self.Phi) if requested_only else self._projector()
        # the measured update compares |u_a| |c_ab| with |x_b|: the largest column norm of U, once per projector
        which = "D" if U is self.BPhiD else "N"
        if self._uscale.get(which) is None:
            self._uscale[which] = float(np.max(U.colnorms()))
        return X.project_norm2(U, V, uscale=self._uscale[which], tol=tol)

    def project_s(self, X):
        """X <- X - Phi (BPhi^T X)   (solution-side projector P^T)"""
        U, V = self._projector()
        return X.project(V, U)

    def extra_correction_coefficients(self, dPhib, lam, Gx=None):
        """
        Share of psi along the extra eigenvectors, in closed form as for the pairs j <= N of reference 385-389:
        psi_i += phi_j G0[j, i] / (lam_j - lam_i) with G = -Phix^T Phib (buckling: G0 = diag(lam_x) G).  nx x N.
        ``Gx``: -Phix^T Phib if the caller has it already.
        """
        if Gx is None:
            Gx = -self.Phix.tdot(dPhib)
        if self.mode != "normal":
            Gx = self.lam_x[:, None] * Gx
        return Gx / (self.lam_x[:, None] - np.asarray(lam, dtype=float)[None, :])

    def adjoint_operator(self, X, lam_cols, out=None):
        """(A - lam B) X or (B + lam A) X, column-wise lam"""
        tA = self.opA.apply(X)
        tB = self.opB.apply(X)
        out = self.ctx.empty(X.n, X.k) if out is None else out
        if self.mode == "normal":
            return out.assign_lincomb([(1.0, tA), (-np.asarray(lam_cols), tB)])
        return out.assign_lincomb([(1.0, tB), (np.asarray(lam_cols), tA)])

    def residual(self, Phib, psi, lam_cols):
        """-Phib - op(psi)   (ref 807-809, 986-988, 1189-1192)"""
        tA = self.opA.apply(psi)
        tB = self.opB.apply(psi)
        R = self.ctx.empty(psi.n, psi.k)
        lam_cols = np.asarray(lam_cols)
        if self.mode == "normal":
            return R.assign_lincomb([(-1.0, Phib), (-1.0, tA), (lam_cols, tB)])
        return R.assign_lincomb([(-1.0, Phib), (-1.0, tB), (-lam_cols, tA)])


def _default_factor(A, B, lam, sigma, mode, ctx):
    """ref 783-790 / 954-961 / 1160-1167"""
    if sigma is None:
        sigma = 0.9 * lam[0]
    P = A - sigma * B if mode == "normal" else B + sigma * A
    return SpLuOperator(P.tocsc(), ctx=ctx), sigma


def _ctx_of(factor, ctx):
    if ctx is not None:
        return ctx
    if isinstance(factor, SpLuOperator):
        return factor.ctx
    from .device import default_context

    return default_context()


def _check_iter_args(Phib, A, B, lam, Phi, psi, mode, check_lam=True):
    n = A.shape[1]
    N = Phib.shape[1]
    _check_mode(mode)
    if check_lam and len(lam) != N:
        raise ValueError(f"Eigenvalues must be of length {N}")
    if A.shape != (n, n):
        raise ValueError(f"A must have dimensions ({n},{n})")
    if B.shape != (n, n):
        raise ValueError(f"B must have dimensions ({n},{n})")
    if psi is not None and psi.shape != (n, N):
        raise ValueError(f"Initial guess must have the shape ({n},{N})")
    if Phi.shape != (n, N):
        raise ValueError(f"Eigenvectors must have the shape ({n},{N})")
    if Phib.shape != (n, N):
        raise ValueError(f"Right-hand-side must have the shape ({n},{N})")
    return n, N


def _rnorm0(Phib_dev):
    """sqrt of the largest column sum of squares of Phib (ref 798, 973, 1170)"""
    return float(np.sqrt(np.max(Phib_dev.coldot(Phib_dev))))


def _emit(callback, histories, order):
    """replay the residual histories mode by mode, the order the reference's sequential loop produces"""
    if callback is None:
        return
    for c in order:
        for r in histories[c]:
            callback(r)


# ---------------------------------------------------------------------------
# correction along the eigenvectors (ref 303-391)
# ---------------------------------------------------------------------------
def repeated_pairs(lam, eig_atol=1e-5):
    """the pairs (i, j), j < i, that the reference treats as numerically repeated (ref 370-372), in its loop order"""
    lam = np.asarray(lam, dtype=float)
    close = np.tril(np.fabs(lam[:, None] - lam[None, :]) < eig_atol, -1)   # (_is_close for every pair j < i at once)
    return [(int(i), int(j)) for i, j in zip(*np.nonzero(close))]            # row-major: i ascending, then j


def compensated_entries(dU, dX, rows, cols, sign=1.0):
    """
    sign * U[:, rows[q]] . X[:, cols[q]] accumulated in twice the working precision (eigd_coldot_dd): (hi, lo) arrays
    with hi + lo = the dot products to ~1e-30 of sum |u x|.
    """
    hi, lo = dU.gather_cols(np.asarray(rows)).coldot_dd(dX.gather_cols(np.asarray(cols)))
    return sign * hi, sign * lo


def refine_repeated_entries(G, lam, dU, dX, eig_atol=1e-5, sign=-1.0, col_of=None):
    """
    The entries G[j, i], G[i, j] of numerically repeated pairs recomputed with compensated dot products, in place; the
    low-order parts come back in a second matrix (None when no pair is repeated).

    xi and eta (ref 373-383) divide the DIFFERENCE of two n-term dot products by the gap of the pair -- 1e-7 for the
    thermal plate at epsilon = 1e-8 -- so the rounding of a plain dot product, a few 1e-16 of |phi| |Phib|, arrives
    in df/dx seven digits larger.  The reference's own value carries that noise (1.4e-9 of df/dx under a change of
    summation order); ours carries the noise of whatever reduction tree the GEMM kernel uses.  Two exact-to-rounding
    entries per pair remove our share, which keeps the repeated branch inside the 1e-8 of north_star.
    ``col_of``: global mode -> column of dX / G (mode-sharded solvers hold a subset of the columns); pairs with a
    missing column are skipped.
    """
    pairs = repeated_pairs(lam, eig_atol)
    if not pairs:
        return None
    rows, cols = [], []
    for i, j in pairs:
        rows += [j, i]
        cols += [i, j]
    Glo = np.zeros_like(G)
    if col_of is not None:
        keep = [q for q in range(len(rows)) if cols[q] in col_of]
        rows, cols = [rows[q] for q in keep], [col_of[cols[q]] for q in keep]
        if not rows:
            return Glo         # (not None: whether a low-order matrix exists is the same decision on every rank)
    hi, lo = compensated_entries(dU, dX, rows, cols, sign)
    G[rows, cols] = hi
    Glo[rows, cols] = lo
    return Glo


def correction_coefficients(lam, G, eig_atol=1e-5, mode="normal", Glo=None):
    """
    Host part of generate_adjoint_correction: the N x N matrix Cc with psi += Phi @ Cc for the
    distinct pairs (ref 385-389) and the dict of (j, xi, eta) tuples of the repeated pairs
    (ref 373-383).  Loop order and formulas are the reference's, so index sets are identical.
    ``Glo``: low-order parts of the entries of repeated pairs (refine_repeated_entries); xi and eta are then formed in
    extended precision from hi + lo, so the division by the gap magnifies no rounding of ours.
    """
    lam = np.asarray(lam, dtype=float)
    N = len(lam)
    G0 = G if mode == "normal" else np.diag(lam) @ G
    # distinct pairs (385-389): Cc[a, b] = G0[a, b] / (lam_a - lam_b), all of them at once (every entry is written once in
    # the reference's loop, so the order does not enter); repeated pairs (373-383) in the reference's loop order
    diff = lam[:, None] - lam[None, :]
    distinct = ~(np.fabs(diff) < eig_atol)
    np.fill_diagonal(distinct, False)
    Cc = np.zeros((N, N))
    np.divide(G0, diff, out=Cc, where=distinct)
    data = {}
    ld = np.longdouble
    for i, j in repeated_pairs(lam, eig_atol):
        gap = lam[j] - lam[i]
        if Glo is None:
            xi = 0.5 * (G0[j, i] - G0[i, j]) / gap
            eta = 0.5 * (lam[i] * G0[j, i] - lam[j] * G0[i, j]) / gap
        else:
            gji, gij = ld(G[j, i]) + ld(Glo[j, i]), ld(G[i, j]) + ld(Glo[i, j])
            if mode != "normal":                      # G0 = diag(lam) G
                gji, gij = ld(lam[j]) * gji, ld(lam[i]) * gij
            xi = float(ld(0.5) * (gji - gij) / ld(gap))
            eta = float(ld(0.5) * (ld(lam[i]) * gji - ld(lam[j]) * gij) / ld(gap))
        data.setdefault(i, [])
        data.setdefault(j, [])
        data[i].append((j, xi, eta))
        data[j].append((i, xi, eta))
    return Cc, data


def _apply_correction(psi_dev, Phi_dev, Cc, cols=None):
    """psi[:, cols] += Phi @ Cc[:, cols]"""
    sub = Cc if cols is None else Cc[:, cols]
    if np.any(sub != 0.0):
        psi_dev.add_product(Phi_dev, sub, alpha=1.0, beta=1.0)
    return psi_dev


def generate_adjoint_correction(lam, Phi, psi, G=None, Phib=None, eig_atol=1e-5, mode="normal", ctx=None):
    N = len(lam)
    n = Phi.shape[0]
    _check_mode(mode)
    if G is None:
        if Phi.shape != (n, N):
            raise ValueError(f"Eigenvectors must have the shape ({n},{N})")
        if Phib.shape != (n, N):
            raise ValueError(f"Right-hand-side must have the shape ({n},{N})")
        if psi.shape != (n, N):
            raise ValueError(f"Eigenvector adjoint must have the shape ({n},{N})")
    else:
        if G.shape != (N, N):
            raise ValueError(f"G must have dimensions ({N},{N})")
        if Phi.shape != (n, N):
            raise ValueError(f"Phi must have dimensions ({n},{N})")
    ctx = _ctx_of(None, ctx)
    dPhi = ctx.from_host(Phi)
    Glo = None
    if G is None:
        dPhib = ctx.from_host(Phib)
        G = -dPhi.tdot(dPhib)
        Glo = refine_repeated_entries(G, lam, dPhi, dPhib, eig_atol)
    Cc, data = correction_coefficients(lam, G, eig_atol, mode, Glo)
    dpsi = ctx.from_host(psi)
    _apply_correction(dpsi, dPhi, Cc)
    writable_result(psi)[:] = dpsi.get()  # in place, as ref 386-389
    return data


# ---------------------------------------------------------------------------
# total derivative (ref 33-182)
# ---------------------------------------------------------------------------
def derivative_weight_coefficients(lam, lamb, beta, adj_corr_data, mode, N):
    """
    Coefficients of WA = Phi @ CA + psi * sa and WB = Phi @ CB + psi * sb (columns = modes):
    normal   (ref 96-111):  WA_i = lamb_i phi_i + psi_i + sum xi phi_j
                            WB_i = (beta_i + lam_i lamb_i) phi_i + lam_i psi_i + sum eta phi_j
    buckling (ref 118-132): WA_i = lam_i (lamb_i phi_i + psi_i) + sum eta phi_j
                            WB_i = (lamb_i - beta_i) phi_i + psi_i + sum xi phi_j
    """
    CA, CB = np.zeros((N, N)), np.zeros((N, N))
    idx = np.arange(N)
    if mode == "normal":
        CA[idx, idx] = lamb
        CB[idx, idx] = beta + lam * lamb
        sa, sb = np.ones(N), np.array(lam, dtype=float)
        ia, ib = 1, 2
    else:
        CA[idx, idx] = lam * lamb
        CB[idx, idx] = lamb - beta
        sa, sb = np.array(lam, dtype=float), np.ones(N)
        ia, ib = 2, 1
    for i, lst in adj_corr_data.items():
        if i >= N:
            continue
        for tup in lst:
            CA[tup[0], i] += tup[ia]
            CB[tup[0], i] += tup[ib]
    return CA, CB, sa, sb


def device_derivative_weights(dPhi, dPhib, dpsi, lam, lamb, adj_corr_data, mode, cols=None):
    """WA, WB device blocks (all modes, or the listed columns when mode-sharded)"""
    N = dPhi.k
    lam = np.asarray(lam, dtype=float)
    lamb = np.asarray(lamb, dtype=float)
    beta = 0.5 * dPhi.coldot(dPhib)
    CA, CB, sa, sb = derivative_weight_coefficients(lam, lamb, beta, adj_corr_data, mode, N)
    ctx = dPhi.ctx
    if cols is None:
        cols = np.arange(N)
        psi_c = dpsi
    else:
        cols = np.asarray(cols)
        psi_c = dpsi.gather_cols(cols) if dpsi.k == N else dpsi
    k = len(cols)
    WA = ctx.empty(dPhi.n, k).assign_lincomb([(sa[cols], psi_c)])
    WA.add_product(dPhi, CA[:, cols])
    WB = ctx.empty(dPhi.n, k).assign_lincomb([(sb[cols], psi_c)])
    WB.add_product(dPhi, CB[:, cols])
    return WA, WB


def add_eig_total_derivative(lam, Phi, lamb, Phib, psi, dAdx, dBdx, dfdx, adj_corr_data={}, mode="normal",
                             deriv_type="vector", ctx=None, cols=None):
    """
    ref 33-182.  The weight vectors are assembled on the device in one batched pass; the
    user callbacks dAdx / dBdx run on the calling thread with numpy arrays as in the reference
    (or with device blocks if the callback has a true ``device`` attribute).  ``cols``
    restricts the sum to a subset of modes (mode sharding: the caller all-reduces dfdx).
    """
    n, N = Phi.shape
    _check_mode(mode)
    if len(lam) != N:
        raise ValueError(f"Eigenvalues must be of length {N}")
    for arr, what in ((psi, "Eigenvectors"), (Phi, "Eigenvectors"), (Phib, "Right-hand-side")):
        if arr.shape != (n, N):
            raise ValueError(f"{what} must have the shape ({n},{N})")
    if deriv_type not in ("vector", "tensor"):
        return dfdx  # the reference silently does nothing for an unknown deriv_type (ref 91, 135)
    ctx = _ctx_of(None, ctx)
    dPhi, dPhib, dpsi = ctx.from_host(Phi), ctx.from_host(Phib), ctx.from_host(psi)
    return _total_derivative_device(dPhi, dPhib, dpsi, lam, lamb, dAdx, dBdx, dfdx, adj_corr_data, mode, deriv_type,
                                    cols, Phi_host=Phi)


def _total_derivative_device(dPhi, dPhib, dpsi, lam, lamb, dAdx, dBdx, dfdx, adj_corr_data, mode, deriv_type, cols,
                             Phi_host=None, comm=None):
    """
    ``cols`` restricts the sum to a rank's modes; with ``comm`` the partial sums of all ranks are then added by ONE
    all-reduce -- on the device vector they were accumulated in when the callbacks are device callbacks (RCCL, no host
    round trip), on a host vector for numpy callbacks.
    """
    N = dPhi.k
    sel = np.arange(N) if cols is None else np.asarray(cols)
    sharded = comm is not None and comm.size > 1
    sB = -1.0 if mode == "normal" else 1.0
    cbs = (dAdx, dBdx)
    dev_cb = [bool(getattr(cb, "device", False)) for cb in cbs]
    # device callbacks that can accumulate (ElementBilinear) share one device vector: one D2H, one host addition
    fused = [on_dev and hasattr(cb, "accumulate") and getattr(cb, "nelem", -1) == len(dfdx)
             for cb, on_dev in zip(cbs, dev_cb)]
    any_fused = any(f and cb is not None for f, cb in zip(fused, cbs))
    any_host = any(cb is not None and not f for f, cb in zip(fused, cbs))
    ctx = dPhi.ctx
    acc = ctx.zeros(len(dfdx), 1) if any_fused else None
    host = np.zeros_like(dfdx) if (sharded and any_host) else dfdx
    if len(sel) > 0:
        WA, WB = device_derivative_weights(dPhi, dPhib, dpsi, lam, lamb, adj_corr_data, mode, cols)
        Phi_sel_dev = dPhi if cols is None else dPhi.gather_cols(sel)
        Phi_sel = None
        if not all(d or cb is None for d, cb in zip(dev_cb, cbs)):
            Phi_sel = (Phi_host if Phi_host is not None else dPhi.get())[:, sel]
        for cb, W, sign, on_dev, fuse in ((dAdx, WA, 1.0, dev_cb[0], fused[0]), (dBdx, WB, sB, dev_cb[1], fused[1])):
            if cb is None:
                continue
            if fuse:
                cb.accumulate(W, Phi_sel_dev, acc, alpha=sign)
                continue
            if on_dev:
                host += sign * cb(W, Phi_sel_dev)
                continue
            Wh = W.get()
            if deriv_type == "vector":
                for q in range(len(sel)):
                    host += sign * cb(Wh[:, q].copy(), Phi_sel[:, q])
            else:
                host += sign * cb(Wh, Phi_sel)
    if acc is not None:
        if sharded:
            if hasattr(comm, "allreduce_sum_device"):
                comm.allreduce_sum_device(acc)               # RCCL, in place on the device
                dfdx += acc.get()[:, 0]
            else:
                dfdx += comm.allreduce_sum(acc.get()[:, 0])
        else:
            dfdx += acc.get()[:, 0]
    if sharded and any_host:
        dfdx += comm.allreduce_sum(host)
    return dfdx


# ---------------------------------------------------------------------------
# residual check (ref 185-275)
# ---------------------------------------------------------------------------
def eval_adjoint_residual_norm(A, B, lam, Phi, Phib, psi, mode="normal", b_ortho=False, ctx=None):
    n = A.shape[1]
    N = Phi.shape[1]
    if len(lam) != N:
        raise ValueError(f"Eigenvalues must be of length {N}")
    if A.shape != (n, n):
        raise ValueError(f"A must have dimensions ({n},{n})")
    if B.shape != (n, n):
        raise ValueError(f"B must have dimensions ({n},{n})")
    for arr, what in ((psi, "Eigenvectors"), (Phi, "Eigenvectors"), (Phib, "Right-hand-side")):
        if arr.shape != (n, N):
            raise ValueError(f"{what} must have the shape ({n},{N})")
    _check_mode(mode)
    ctx = _ctx_of(None, ctx)
    prob = DeviceProblem(ctx, A, B, None, mode)
    prob.set_phi(Phi_host=Phi)
    return _residual_norm_device(prob, np.asarray(lam, dtype=float), ctx.from_host(Phib), ctx.from_host(psi), b_ortho)


def _residual_norm_device(prob, lam, dPhib, dpsi, b_ortho):
    ctx = prob.ctx
    # b = -(Phib - BPhi * (phi_i . Phib_i))  ->  r = op(psi) - b = op(psi) + Phib - BPhi * diag(d)
    d = prob.Phi.coldot(dPhib)
    R = prob.adjoint_operator(dpsi, lam)
    R.assign_lincomb([(1.0, R), (1.0, dPhib), (-d, prob.BPhi)])
    if b_ortho:
        prob.project_r(R)
        ortho = np.max(np.abs(prob.BPhi.tdot(dpsi)), axis=0)
    else:
        ortho = np.abs(prob.BPhi.coldot(dpsi))
    return R.colnorms(), ortho


# ---------------------------------------------------------------------------
# Lanczos adjoint approximation (ref 394-523)
# ---------------------------------------------------------------------------
def laa_coefficients(Yb, lam, sigma, Y, theta, indices, b_ortho, mode):
    """host part of laa: the m x N coefficient matrix Cf with psi = -factor(B V Cf) (ref 501-521)"""
    m = len(theta)
    N = Yb.shape[1]
    C = Y.T @ Yb
    th_sel = theta[indices[:N]]
    D = np.zeros((m, N))
    if b_ortho:
        rows = indices[N:]
        D[rows, :] = C[rows, :] / (th_sel[None, :] - theta[rows, None])
    else:
        with np.errstate(divide="ignore", invalid="ignore"):
            D = C / (th_sel[None, :] - theta[:, None])
        D[indices[:N], np.arange(N)] = 0.0
    scale = 1.0 if mode == "normal" else sigma
    return Y @ (scale * (D / (np.asarray(lam) - sigma)))


def _laa_device(prob, Vstack, m, dPhib, lam, sigma, Y, theta, indices, b_ortho, mode, cols=None):
    """psi (device, n x len(cols)) from the Lanczos data; V is a k=1 stack of m vectors"""
    Yb = Vstack.tdot_block(dPhib, ns=m)  # V^T Phib, all modes (D couples only through theta)
    Cf = laa_coefficients(Yb, lam, sigma, Y, theta, indices, b_ortho, mode)
    if cols is not None:
        Cf = Cf[:, cols]
    X = prob.ctx.empty(prob.n, Cf.shape[1])
    Vstack.times_into(X, Cf, ns=m, alpha=1.0, beta=0.0)
    psi = prob.opB.apply(X)
    prob.fac(psi, alpha=-1.0)
    return psi


def _laa_relation_device(prob, Vpan, c, p, T, C_last, dPhib, lam, sigma, Y, theta, indices, mode, cols=None):
    """
    The Lanczos adjoint approximation from the block eigensolver's own basis WITHOUT the factor application of 521:
    psi = -factor(B V Cf), and the basis satisfies the Lanczos relation factor(B V) = V T + Q C_last E_last^T (Q: the
    residual block, stored behind the c basis vectors; it holds to the rounding of the eigensolve, ~1e-13), so
    psi = -(V (T Cf) + Q (C_last Cf[c-p:c])): two tall-skinny products instead of a product with B and a sweep.
    Used as the first guess of the Krylov solvers only (the method "laa" itself answers as the reference does).
    """
    Yb = Vpan.tdot_block(dPhib, ns=c)
    Cf = laa_coefficients(Yb, lam, sigma, Y, theta, indices, True, mode)
    if cols is not None:
        Cf = Cf[:, cols]
    psi = prob.ctx.empty(prob.n, Cf.shape[1])
    if c + p <= 192 and hasattr(Vpan, "times_panels"):
        # one pass over the panels for both terms: [V | Q] [T Cf; C_last Cf_last]
        Vpan.times_panels(psi, -np.vstack([T @ Cf, C_last @ Cf[c - p:c]]), c + p)
    else:
        Vpan.times_into(psi, T @ Cf, ns=c, alpha=-1.0, beta=0.0)
        Vpan.times_into(psi, C_last @ Cf[c - p:c], ns=p, alpha=-1.0, beta=1.0, j0=c)
    return psi


def _vstack_from_host(ctx, V):
    n, m = V.shape
    st = ctx.stack(m, n, 1)
    Vt = np.ascontiguousarray(V.T)
    from ._ffi import call, hptr

    call("eigd_h2d", ctx.h, st.ptr, hptr(Vt), 8 * n * m)
    return st


def laa(Phib, B, factor, sigma, lam, V, Y, theta, indices, D0=None, b_ortho=False, mode="normal", ctx=None):
    n = B.shape[1]
    m = len(theta)
    N = Phib.shape[1]
    _check_mode(mode)
    if len(lam) != N:
        raise ValueError(f"Eigenvalues must be of length {N}")
    if Phib.shape != (n, N):
        raise ValueError(f"Right-hand-side must have the shape ({n},{N})")
    if B.shape != (n, n):
        raise ValueError(f"B must have dimensions ({n},{n})")
    if factor.shape != (n, n):
        raise ValueError(f"Factorized operator must have dimensions ({n},{n})")
    if len(indices) != m:
        raise ValueError(f"Length of indices array must be (m = {m})")
    if V.shape != (n, m):
        raise ValueError(f"Dimension of the Lanczos subspace must be ({n},{m})")
    if D0 is not None:
        raise NameError("name 'D' is not defined")  # ref 492-500 reads D before assignment
    ctx = _ctx_of(factor, ctx)
    prob = DeviceProblem(ctx, B, B, factor, mode)
    psi = _laa_device(prob, _vstack_from_host(ctx, V), m, ctx.from_host(Phib), lam, sigma, Y, theta,
                      np.asarray(indices), b_ortho, mode)
    return psi.get()


# ---------------------------------------------------------------------------
# shift-invert block Krylov, lock-step batched form (ref 1052-1328)
# ---------------------------------------------------------------------------
def solve_shifted_lstsq(alpha, H, r):
    """
    min || (I - alpha H) y - r ||  (ref 1043-1049: numpy's lstsq there).  Householder QR instead of lstsq's SVD: the
    same minimiser to rounding for the full-rank Hessenberg systems of the Krylov loops at a fifth of the host time
    (the lock-step solver does two of these per mode and cycle while the next cycle's sweeps are in flight; with the SVD
    the host was the slower side of that overlap in a quarter of the cycles).  Rank-deficient systems (an exhausted
    Krylov space) go to lstsq as in the reference.
    """
    m, n = H.shape
    H0 = -alpha * H
    H0[np.arange(min(m, n)), np.arange(min(m, n))] += 1.0
    if n > 0 and m >= n:
        qr, x, info = _dgels(H0, r)                       # one LAPACK call: QR, Q^T r, back substitution
        d = np.abs(qr.diagonal())
        if info == 0 and d.min() > 1e-13 * d.max():
            return x[:n], float(np.linalg.norm(x[n:]))    # (the tail of Q^T r is the residual of the minimiser)
    y = np.linalg.lstsq(H0, r, rcond=None)[0]
    return y, np.linalg.norm(H0 @ y - r)


from scipy.linalg.lapack import dgels as _dgels  # noqa: E402


def _cgs2(Wst, T, ns, c0=0):
    """
    T <- (I - W W^T) T over the first ns slabs (columns c0.. of the stack); returns the coefficients.

    Classical Gram-Schmidt with a measured second pass: the first projection h1 = W^T T is subtracted and what
    is left along W, h2 = W^T (T - W h1), is measured in the same pass over the stack.  h2 / h1 is the loss of
    orthogonality that one-pass Gram-Schmidt would leave behind; it is subtracted (third pass) only when it
    exceeds 1e-13 for some column -- the reference's modified Gram-Schmidt (eigenvector_derivatives.py:1254-1256)
    leaves O(eps * cond) there, orders of magnitude more.
    """
    if T.k * ns * 8 > 60 * 1024:  # deeper than one coefficient block: pass by pass
        h1 = Wst.dot(T, ns=ns, c0=c0)
        Wst.axpy_into(T, h1, alpha=-1.0, c0=c0)
        h2 = Wst.dot(T, ns=ns, c0=c0)
        n1 = np.sqrt(np.sum(h1 * h1, axis=0))
        n2 = np.sqrt(np.sum(h2 * h2, axis=0))
        if not np.any(n2 > tuning.reorth_tol * n1):
            return h1
        Wst.axpy_into(T, h2, alpha=-1.0, c0=c0)
        return h1 + h2
    return Wst.cgs2(T, ns, c0=c0, tol=tuning.reorth_tol)[0]  # one call, one host synchronisation


def _active_range(done):
    """smallest column range [lo, hi) holding every unfinished mode (low modes finish first)"""
    live = np.flatnonzero(~done)
    return int(live[0]), int(live[-1]) + 1


def _sibk_round(prob, R0, lam_c, sigma, rnorm0, rtol, atol, maxiter, hist):
    """
    One attempt of the bs_target=1 solver for all columns of R0 at once.
    Returns (update block dpsi, converged flags, info list).

    The modes advance in lock step; every operator application acts on the contiguous column
    range that still holds unfinished modes (the range shrinks as the low modes converge), so
    late iterations do not pay for finished columns.
    """
    ctx, mode = prob.ctx, prob.mode
    k = R0.k
    Kop = prob.opB if mode == "normal" else prob.opA  # Krylov operator P K factor (ref 1249-1252)
    sgn = 1.0 if mode == "normal" else -1.0            # ref 1265-1268
    info = [None] * k
    done = np.zeros(k, dtype=bool)
    converged = np.zeros(k, dtype=bool)
    beta0 = R0.colnorms()
    for c in range(k):
        hist[c].append(beta0[c])
        if beta0[c] < rtol * rnorm0 or beta0[c] < atol:  # ref 1223-1225
            info[c] = 0
            done[c] = converged[c] = True
    dpsi = ctx.zeros(prob.n, k)
    if done.all():
        return dpsi, converged, info
    W = ctx.workspace_stack("krylov_W", maxiter + 1, prob.n, k)
    Z = ctx.workspace_stack("krylov_Z", maxiter, prob.n, k)  # columns of finished modes are never written: their
    # coefficients in the final psi += Z y are exact zeros, which eigd_stack_axpy skips without touching the entry
    W0 = W[0]
    W0.copy_from(R0)
    prob.project_r(W0)                                   # ref 1232
    r00 = W0.colnorms()                                  # ref 1233
    scale = np.where(done | (r00 == 0.0), 0.0, 1.0 / np.where(r00 == 0.0, 1.0, r00))
    W0.assign_lincomb([(scale, W0)])                     # ref 1234 (finished columns are zeroed)
    H = np.zeros((k, maxiter + 1, maxiter))
    Ycoef = np.zeros((maxiter, k))
    T = ctx.empty(prob.n, k)
    cols = np.arange(k)          # original column of every column of the current stacks
    jlast = 0

    def enqueue_operator(kp, lo, hi, nlive):
        """device work of one Krylov step up to the first host-visible result (no synchronisation)"""
        Zk, Ta = Z[kp].cols(lo, hi), T.cols(lo, hi)
        prob.fac.apply_to(W[kp].cols(lo, hi), Zk, count=nlive)   # ref 1248: one multi-column sweep
        Kop.apply(Zk, Ta)                                # ref 1250 / 1252
        prob.project_r(Ta)
        return Ta

    def small_solves(j, lo, hi, h, hn):
        """host side of step j: Hessenberg least squares and convergence tests of the live modes (ref 1262-1321)"""
        kp = j - 1
        for cl in range(lo, hi):
            c = cols[cl]
            if done[c]:
                continue
            H[c, :j, kp] = h[:, cl - lo]
            H[c, j, kp] = hn[cl - lo]
            rvec = np.zeros(j + 1)
            rvec[0] = r00[c]
            y, res = solve_shifted_lstsq(sgn * (lam_c[c] - sigma), H[c, : j + 1, :j], rvec)  # ref 1262-1270
            hist[c].append(res)
            if res < rtol * rnorm0 or res < atol:        # ref 1275
                info[c] = j
                Ycoef[:j, c] = y
                done[c] = converged[c] = True
            elif j == maxiter:                           # ref 1312-1313: keep the best iterate
                Ycoef[:j, c] = y
                done[c] = True

    def flush_finished(nslabs):
        """psi += Z y for the finished columns of the current stacks (ref 1277 / 1313), scattered to their places"""
        fin = np.flatnonzero(done[cols])
        if len(fin) == 0 or nslabs == 0:
            return
        upd = ctx.zeros(prob.n, len(cols))
        Z.axpy_into(upd, Ycoef[:nslabs][:, cols], alpha=1.0)
        if len(cols) == k and len(fin) == k:
            dpsi.copy_from(upd)
        else:
            upd.gather_cols(fin).scatter_cols_into(dpsi, cols[fin])

    # Software pipeline: the device work of step j+1 (sweep, SpMM, projection) is enqueued BEFORE the host does
    # the least-squares problems of step j, so the small dense solves overlap the triangular sweep.  A mode that
    # turns out to have converged at step j rides along for one extra step (its coefficients there are zero).
    lo, hi = _active_range(done[cols])
    Ta = enqueue_operator(0, lo, hi, int(np.count_nonzero(~done)))
    for j in range(1, maxiter + 1):
        h = _cgs2(W, Ta, j, c0=lo)                       # ref 1254-1256 (Gram-Schmidt vs all previous W)
        hn2 = prob.project_r_norm2(Ta)                   # ref 1257 + 1259: the norms stay on the device ...
        jlast = j
        cur = (lo, hi)
        nxt = None
        if j < maxiter:
            W[j].cols(lo, hi).assign_scaled_inverse(Ta, hn2, done[cols[lo:hi]])  # ref 1260 ... where the next basis vector
            nxt = enqueue_operator(j, lo, hi, 0)         # is formed; the host reads them behind the sweep in flight
        hn = np.sqrt(ctx.fetch_colnorm2(cur[1] - cur[0]))
        small_solves(j, cur[0], cur[1], h, hn)
        if done.all():
            break
        if nxt is not None and prob.fac.native:          # count the sweep in flight for the modes that go on
            with prob.fac.factor._count_lock:
                prob.fac.factor.count += int(np.count_nonzero(~done))
        if nxt is not None:
            # the step in flight keeps the range it was launched with; narrower ranges apply from the step after.
            # Columns that finished meanwhile are zeroed when the next basis vector is formed (scale above).
            nlo, nhi = _active_range(done[cols])
            if (nlo, nhi) != (lo, hi):
                Ta = nxt.cols(nlo - lo, nhi - lo)
                lo, hi = nlo, nhi
            else:
                Ta = nxt
    flush_finished(jlast)                                # ref 1277 / 1313: psi += Z y
    return dpsi, converged, info


LAST_ROUND = {"steps_per_pass": None, "inner_projections": None}   # what the last lock-step round of the two-step solver ran with (tests)


def pair_arnoldi_columns(Hc, Czc, j, h1, g1, b1, gamma, b2):
    """
    Host algebra of one two-step cycle for one mode (see _sibk_round_pair).  Given the Gram-Schmidt coefficients of the
    raw pair against W_0..W_j (v1 = OP w_j: h1, v2 = OP v1: g1) and what the orthonormalisation inside the pair measured
    (b1 = |v1'|, gamma = v1'.v2', b2 = |v2' - (w_{j+1}.v2') w_{j+1}|), fills
      * columns j and j+1 of the Arnoldi matrix Hc (OP W = W H):  H[:, j] = [h1; b1],
        H[:, j+1] = ([g1; gamma/b1; b2] - H[:, :j+1] h1) / b1;
      * columns j and j+1 of Czc, the coordinates of factor(w_i) in the STORED slabs (slab j = factor(w_j),
        slab j+1 = factor(v1)):  factor(w_{j+1}) = (slab_{j+1} - sum_i h1[i] factor(w_i)) / b1.
    b1 == 0 (the Krylov space is exhausted): only column j is filled.
    """
    ns = j + 1
    Hc[:ns, j] = h1
    Hc[ns, j] = b1
    Czc[j, j] = 1.0
    if not b1 > 0.0:
        return
    colv = np.zeros(ns + 2)
    colv[:ns] = g1
    colv[ns] = gamma / b1
    colv[ns + 1] = b2
    colv[: ns + 1] -= Hc[: ns + 1, :ns] @ h1
    Hc[: ns + 2, j + 1] = colv / b1
    Czc[:, j + 1] = 0.0
    Czc[j + 1, j + 1] = 1.0 / b1
    Czc[:ns, j + 1] -= (Czc[:ns, :ns] @ h1) / b1


def _sibk_round_pair(prob, R0, lam_c, sigma, rnorm0, rtol, atol, maxiter, hist):
    """
    The lock-step solver with TWO Krylov steps per Gram-Schmidt pass.  Same Krylov spaces, same Hessenberg matrices
    and residual history (to rounding) and therefore the same iteration counts and solutions as _sibk_round; half the
    passes over the Krylov history, which are a third of a step at 1 M dof.

    A cycle starts from an orthonormal w_j and applies the operator twice without orthogonalising in between:
        z1 = factor(w_j),  v1 = P K z1            (reference 1248-1252)
        z2 = factor(v1),   v2 = P K z2
    Then the pair [v1 | v2] is orthogonalised against W_0..W_j in ONE classical Gram-Schmidt step with measured second
    pass (two passes over the stack for two new vectors, ref 1254-1256), projected (1257), and orthonormalised inside
    the pair on the device: w_{j+1} = v1'/b1, w_{j+2} = (v2' - (w_{j+1}.v2') w_{j+1})/b2.  With v1 = W h1 + b1 w_{j+1}
    and v2 = W g1 + gw w_{j+1} + b2 w_{j+2} the Arnoldi columns are
        H[:, j]   = [h1; b1]
        H[:, j+1] = ([g1; gw; b2] - H[:, :j+1] h1) / b1        (OP w_{j+1} = (v2 - OP W_{<=j} h1) / b1)
    and factor(w_{j+1}) = (z2 - sum_i h1[i] factor(w_i)) / b1 is never formed: the solution update psi += Z y (1277)
    uses the stored slabs with coefficients transformed on the host (Cz).  The residual of every step is evaluated from
    H as before, so a mode stops at the same step as in the one-step form.  The operator does not amplify along the
    Krylov vectors here (|OP w| / b is 2-3 on the 1 M-dof benchmark; measured against the one-step form on the CPU:
    identical iteration counts, psi equal to 1e-14): two unorthogonalised steps lose no accuracy.  What one
    Gram-Schmidt pass inside the pair leaves (w_{j+1}.w_{j+2}) is measured on the device; if it ever exceeds 1e-10 the
    caller repeats the solve with the one-step form.
    """
    ctx, mode = prob.ctx, prob.mode
    n, k = prob.n, R0.k
    Kop = prob.opB if mode == "normal" else prob.opA
    sgn = 1.0 if mode == "normal" else -1.0
    info = [None] * k
    done = np.zeros(k, dtype=bool)
    converged = np.zeros(k, dtype=bool)
    beta0 = R0.colnorms()
    for c in range(k):
        hist[c].append(beta0[c])
        if beta0[c] < rtol * rnorm0 or beta0[c] < atol:  # ref 1223-1225
            info[c] = 0
            done[c] = converged[c] = True
    dpsi = ctx.zeros(n, k)
    if done.all():
        return dpsi, converged, info, True
    W = ctx.workspace_stack("krylov_W2", maxiter + 2, n, k)
    Z = ctx.workspace_stack("krylov_Z2", maxiter + 1, n, k)
    W0 = W[0]
    W0.copy_from(R0)
    # ref 1232-1233: the start residual was projected by the caller already (1193): this projection is measured -- its
    # coefficient pass delivers the column norms as well, the update runs only if some coefficient matters
    if k <= 64:
        r00 = np.sqrt(np.maximum(prob.project_r_norm2(W0).get().ravel(), 0.0))
        ctx.project_stats()                              # (not one of the projections behind a Gram-Schmidt step: not counted)
    else:
        prob.project_r(W0)
        r00 = W0.colnorms()
    scale = np.where(done | (r00 == 0.0), 0.0, 1.0 / np.where(r00 == 0.0, 1.0, r00))
    W0.assign_lincomb([(scale, W0)])                     # ref 1234
    H = np.zeros((k, maxiter + 3, maxiter + 2))
    Cz = np.zeros((k, maxiter + 2, maxiter + 2))         # factor(w_i) = sum_s Cz[c][s, i] * (stored slab s)
    Ycoef = np.zeros((maxiter + 1, k))                   # coefficients of the STORED slabs in psi += Z y
    TP = ctx.empty(n, 2 * k)
    ok = True

    inner_proj = [True]

    def enqueue_cycle(j, lo, hi):
        """
        device work of one cycle up to the pair of raw vectors (no synchronisation after the first cycle).

        The projections behind the two operator applications (1250-1252) can be taken together: with F B phi = theta phi
        the range of P is invariant under K F (Phi^T K F w = Theta Phi^T w = 0), so P K F P = P K F and
        v2 = P K F (P v1') = P K F v1' for the unprojected v1' = K F w.  After the first cycle the two are therefore
        replaced by ONE projection of the raw pair [v1' | v2'] right before its Gram-Schmidt step (Phi and B Phi are
        streamed once instead of twice; the projection after the Gram-Schmidt step, 1257, stays as it is).  What the
        unprojected v1' carries along B Phi -- rounding, and the non-invariance of range(P) for inexact eigenvectors,
        which is MEASURED in the first cycle as max |Phi^T K F w_0| for unit w_0 -- enters v2 only through P K F (B Phi c)
        = P K Phi Theta c, i.e. squared; the Gram-Schmidt step itself sees projected vectors as before (projecting only
        AFTER it was tried: the part removed there is not orthogonal to W, nine of eleven cycles then needed a correcting
        Gram-Schmidt pass).  Above a measured non-invariance of 1e-9 every projection stays where the reference has it.
        """
        kk = hi - lo
        T1, T2 = TP.cols(0, kk), TP.cols(kk, 2 * kk)
        Zj, Zj1 = Z[j].cols(lo, hi), Z[j + 1].cols(lo, hi)
        prob.fac.apply_to(W[j].cols(lo, hi), Zj, count=0)
        Kop.apply(Zj, T1)
        if j == 0:
            if not tuning.inner_projections:
                _, Vp = prob._projector()
                inner_proj[0] = not (np.max(np.abs(Vp.tdot(T1))) <= 1e-9)
            LAST_ROUND["inner_projections"] = inner_proj[0]
        if inner_proj[0] or j == 0:
            prob.project_r(T1)
        prob.fac.apply_to(T1, Zj1, count=0)
        Kop.apply(Zj1, T2)
        if inner_proj[0] or j == 0:
            prob.project_r(T2)

    def small_solves(j, lo, hi, h, vals):
        """host side of a cycle: two Arnoldi columns, the coefficient transform and the residuals of steps j+1, j+2"""
        nonlocal ok
        kk = hi - lo
        ns = j + 1
        n1sq, gam, n2sq, defect = vals[:kk], vals[kk: 2 * kk], vals[2 * kk: 3 * kk], vals[3 * kk:]
        for cl in range(lo, hi):
            c, q = cl, cl - lo
            if done[c]:
                continue
            Hc, Czc = H[c], Cz[c]
            b1, b2 = np.sqrt(n1sq[q]), np.sqrt(n2sq[q])
            steps = (j + 1, j + 2)
            if b1 > 0.0:
                if abs(defect[q]) > tuning.pair_defect_tol * max(b2, np.finfo(float).tiny):
                    ok = False                            # (never seen: the pair was not orthogonalised well enough)
            else:
                steps = (j + 1,)                          # breakdown: the Krylov space is exhausted at step j+1
            pair_arnoldi_columns(Hc, Czc, j, h[:, q], h[:, kk + q], b1, gam[q], b2)
            for jj in steps:
                if jj > maxiter:
                    break
                rvec = np.zeros(jj + 1)
                rvec[0] = r00[c]
                y, res = solve_shifted_lstsq(sgn * (lam_c[c] - sigma), Hc[: jj + 1, :jj], rvec)  # ref 1262-1270
                hist[c].append(res)
                if res < rtol * rnorm0 or res < atol:    # ref 1275
                    info[c] = jj
                    Ycoef[:jj, c] = Czc[:jj, :jj] @ y
                    done[c] = converged[c] = True
                    break
                if jj == maxiter or (jj == steps[-1] and len(steps) == 1):   # ref 1312-1313: keep the best iterate
                    Ycoef[:jj, c] = Czc[:jj, :jj] @ y
                    done[c] = True
                    break

    # The device part of the next cycle is put in flight while the host solves the current one.  What it has to carry is
    # not known then, so it is PREDICTED from each mode's own residual history (_expected_to_finish: the reduction per
    # step of its last cycle applied to the two steps being solved; at C3 the residual two steps on is 0.3 ... 1.15 of that):
    #  * every live mode expected to finish (generous margin): nothing is enqueued -- the cycle behind the last one
    #    would be two sweeps and two products for nothing; if a mode has not finished after all, the next cycle starts
    #    when the host knows, at the price of the host solve's time;
    #  * otherwise the next cycle runs on the column range of the modes NOT expected to finish (strict margin), instead
    #    of the range that was unfinished a cycle ago: the sweeps drop to the 16- and 8-column kernels one cycle
    #    earlier.  A mode outside that range that did not finish after all gets the cycle of its own afterwards
    #    (columns never mix: the same cycle index may be worked through range by range).
    ranges = [_active_range(done)]                        # column ranges to work through at cycle index j; [0] is in flight
    j = 0
    enqueue_cycle(0, *ranges[0])
    while True:
        lo, hi = ranges.pop(0)                            # a cycle keeps the column range it was launched with
        kk = hi - lo
        TPv = TP.cols(0, 2 * kk)
        if not inner_proj[0] and j > 0:
            prob.project_r(TPv)                                       # ref 1250-1252 for both vectors of the cycle at once
        h, npass = W.cgs2_pair(TPv, j + 1, c0=lo, tol=tuning.reorth_tol)  # ref 1254-1256 for both vectors; one host sync
        LAST_ROUND["gs_cycles"] = LAST_ROUND.get("gs_cycles", 0) + 1
        LAST_ROUND["gs_correcting_passes"] = LAST_ROUND.get("gs_correcting_passes", 0) + (npass - (2 if j + 1 <= 32 else 3))
        n2 = prob.project_r_norm2(TPv)                                # ref 1257 + 1259
        TPv.pair_orthonormalise(n2, W[j + 1].cols(lo, hi), W[j + 2].cols(lo, hi), done[lo:hi])   # ref 1260
        more = j + 2 < maxiter
        nxt = None                                        # (cycle index, lo, hi) put in flight behind this one
        if ranges:
            nxt = (j,) + tuple(ranges[0])                 # another range waits at this cycle index
        elif more:
            tol_c = max(rtol * rnorm0, atol)
            in_cycle = np.zeros(k, dtype=bool)
            in_cycle[lo:hi] = True                        # (their steps j+1, j+2 are the ones being solved)
            if tuning.predict_finish and j > 0:
                stay_lax = ~done & ~_expected_to_finish(hist, done, in_cycle, tol_c, 4.0)
                stay = ~done & ~_expected_to_finish(hist, done, in_cycle, tol_c, 0.5)
            else:
                stay_lax = stay = ~done
            if stay_lax.any():
                cols_n = np.flatnonzero(stay)
                nxt = (j + 2, int(cols_n[0]), int(cols_n[-1]) + 1)
        if nxt is not None:
            enqueue_cycle(*nxt)
        vals = ctx.fetch_colnorm2(4 * kk)
        small_solves(j, lo, hi, h, vals)
        jlast = min(j + 2, maxiter)
        if not ok:
            break
        if ranges:
            continue                                      # the next range of this cycle index is in flight
        if done.all() or not more:
            LAST_ROUND["cycles_enqueued_for_nothing"] = LAST_ROUND.get("cycles_enqueued_for_nothing", 0) + int(nxt is not None)
            break
        live = np.flatnonzero(~done)
        if nxt is None:
            LAST_ROUND["cycles_waited_for"] = LAST_ROUND.get("cycles_waited_for", 0) + 1
            ranges = [(int(live[0]), int(live[-1]) + 1)]  # (with this cycle's results: the narrowest range)
            enqueue_cycle(j + 2, *ranges[0])
        else:
            ranges = [nxt[1:]]
            for missed in (live[live < nxt[1]], live[live >= nxt[2]]):   # expected to finish, did not: cycles of their own
                if len(missed):                                          # (below and above the range in flight, never across it)
                    LAST_ROUND["cycles_repeated_for_missed_modes"] = LAST_ROUND.get("cycles_repeated_for_missed_modes", 0) + 1
                    ranges.append((int(missed[0]), int(missed[-1]) + 1))
        j += 2
    if prob.fac.native and ok:                            # one factor application per Krylov step and mode (ref 1248);
        with prob.fac.factor._count_lock:                 # an abandoned attempt is counted by the one-step form that redoes it
            prob.fac.factor.count += int(sum((i if i is not None else maxiter) for i in info))
    fin = np.flatnonzero(done)
    if len(fin) and jlast > 0:
        upd = ctx.zeros(n, k)
        Z.axpy_into(upd, Ycoef[:jlast], alpha=1.0)       # ref 1277 / 1313: psi += Z y
        dpsi.copy_from(upd)
    if ok:
        LAST_ROUND["steps_per_pass"] = 2
    measured, applied = ctx.project_stats()               # ref 1257: how often the update behind the Gram-Schmidt
    LAST_ROUND["post_gs_projections"] = LAST_ROUND.get("post_gs_projections", 0) + measured      # step was needed
    LAST_ROUND["post_gs_updates_applied"] = LAST_ROUND.get("post_gs_updates_applied", 0) + applied
    return dpsi, converged, info, ok


# ---------------------------------------------------------------------------
# sibk in short-recurrence form: conjugate gradients in the inner product of the factor (csrc/krylov.hip)
# ---------------------------------------------------------------------------
def _short_recurrence_applies(prob):
    """a native, positive definite, real factor: no negative or static pivots (the shift lies below the spectrum)"""
    fac = prob.fac
    if tuning.recurrence == "arnoldi" or fac is None or not fac.native:
        return False
    op = fac.factor
    return not op._pivoted() and op._imag_dev is None and prob.opA.csr is not None and prob.opB.csr is not None


_CG_ROWS = {"rr": 0, "gam": 1, "rho": 2, "done": 3, "tol2": 4, "alpha": 5, "steps": 6, "flag": 7}
_CG_CHUNK = 16             # slabs per allocation of the z history
_CG_TRACE_HOOK = None      # probes: function(prob, step, residual block, lo, hi, projected) called before a step's projection


def _cg_solution_coefficients(log, k):
    """
    ``log`` (2 steps x 64): rows 2 (j - 1), 2 (j - 1) + 1 = gam_j, rho_j of step j per column (gam = 0: the column did not
    move).  The three-term recurrence psi_j = rho_j (psi_{j-1} + gam_j z_j) + (1 - rho_j) psi_{j-2} is the two-term form
    psi_j = psi_{j-1} + alpha_j p_j, p_j = z_j + beta_{j-1} p_{j-1} with alpha_j = rho_j gam_j and
    beta_{j-1} = (rho_j - 1) alpha_{j-1} / alpha_j, so psi_m = sum_j s_j z_j with s_m = alpha_m, s_j = alpha_j + beta_j s_{j+1}
    -- all positive for a positive definite operator: no cancellation in the sum.  Returns S (steps x k).
    """
    nst = log.shape[0] // 2
    gam, rho = log[0::2, :k], log[1::2, :k]
    S = np.zeros((nst, k))
    for c in range(k):
        mv = np.flatnonzero(gam[:, c] != 0.0)              # the steps in which the column moved, in order
        if mv.size == 0:
            continue
        alpha = rho[mv, c] * gam[mv, c]
        s = alpha[-1]
        S[mv[-1], c] = s
        for i in range(mv.size - 2, -1, -1):
            beta = (rho[mv[i + 1], c] - 1.0) * alpha[i] / alpha[i + 1]
            s = alpha[i] + beta * s
            S[mv[i], c] = s
    return S


def _sibk_cg_round(prob, R0, lam_c, sigma, rnorm0, rtol, atol, maxsteps, hist, host_work=None):
    """
    All columns of R0 (at most 64) by conjugate gradients in the factor inner product, in lock step.  Same Krylov spaces
    as the Arnoldi form (reference 1246-1277), same stopping rule on the true Euclidean residual (1275), no Krylov
    history: per step ONE multi-column sweep, one SpMM, one measured projection and two streaming kernels over the
    work blocks (three-term recurrence for the residual, csrc/krylov.hip); every per-mode scalar stays on the device, the
    host reads the residual norms of a step behind the next step already in flight.  The solution is formed once, at the
    end: psi = sum_k s_k z_k over the z = factor(r) of the steps, which the sweeps leave in the slabs of a history stack
    (tuning.cg_solution_from_history; the coefficients s_k > 0 from the (gam, rho) log of the device).  Returns (update block,
    converged flags, info list, ok); ok False = a breakdown was flagged (the operator was not positive definite in the
    deflated space): the caller redoes the solve in the Arnoldi form.
    """
    from . import _ffi
    from ._ffi import call

    ctx, mode = prob.ctx, prob.mode
    n, k = prob.n, R0.k
    assert k <= 64
    Kop = prob.opB if mode == "normal" else prob.opA      # Krylov operator P K factor (ref 1249-1252)
    sgn = 1.0 if mode == "normal" else -1.0                # ref 1265-1268
    nrows = int(_ffi.lib().eigd_cg_state_rows())
    tol = max(rtol * rnorm0, atol)
    tol2 = tol * tol
    info = [None] * k
    done = np.zeros(k, dtype=bool)
    converged = np.zeros(k, dtype=bool)
    beta0 = R0.coldot(R0)
    for c in range(k):
        hist[c].append(float(np.sqrt(beta0[c])))
        if beta0[c] < tol2:                                # ref 1223-1225
            info[c] = 0
            done[c] = converged[c] = True
    psi = ctx.zeros(n, k)
    if done.all():
        return psi, converged, info, True
    st_h = np.zeros((nrows, 64))
    st_h[_CG_ROWS["done"], :k] = done
    st_h[_CG_ROWS["tol2"], :k] = tol2
    st_h[_CG_ROWS["alpha"], :k] = sgn * (np.asarray(lam_c, dtype=float) - sigma)   # ref 1264-1269
    state = ctx.from_host(st_h)
    r = R0                                                 # (the caller's block is the work block: it is consumed)
    r_old, y = ctx.empty(n, k), ctx.empty(n, k)
    deferred = bool(tuning.cg_solution_from_history)
    if deferred:
        psi_old = z = None
        zchunks = []                                       # stacks of _CG_CHUNK slabs each, kept between calls
        log = ctx.zeros(2 * (maxsteps + 2), 64)            # (gam, rho) of every step, written by eigd_cg_coefficients
    else:
        psi_old, z = ctx.zeros(n, k), ctx.empty(n, k)
        log = None
    failed = []                                            # (no memory for another chunk of the history)
    gave_up = []                                           # columns seen frozen or stalled while the loop ran
    # what the measured projection lets pass: a component along B Phi_D of relative size 1e-11 grows by |1 - alpha theta_j|
    # per step until it is taken out again; psi carries it at that relative size at most, and is projected once at the end
    proj_tol = tuning.cg_projection_tol
    # How often the residual is projected at all (1257).  What the sweep's rounding (1e-13 of the residual) leaves along
    # a deflated direction B phi_j with lam_j below lam_i -- where C_i is NEGATIVE, 1 - (lam_i - sigma) / (lam_j - sigma) --
    # is multiplied by rho (1 - gam mu) per step while the residual shrinks: measured on the 1 M-dof column
    # (tools/contamination_probe.py) a factor of 30 per step relative to the residual, 4 g with g the largest |mu| is the
    # model.  Left alone it takes the recurrence apart (<r, C r>_F turns negative: step 24 on that column).  Every p-th
    # step therefore projects the residual AND the previous one, which the three-term recurrence brings back in the next
    # step (projecting r alone only divided the component by |1 - rho|: it kept growing from period to period);
    # p = the largest period with 1e-13 (4 g)^p < 1e-7 (measured at the ends of such periods: 3e-6 at most; the
    # recurrence's scalars feel the square of it), four at most.  The steps in between take their residual norms
    # from the update kernel.
    lam_all = np.asarray(lam_c, dtype=float)
    lam_defl = lam_all if prob.lam_phi is None else np.asarray(prob.lam_phi, dtype=float)
    with np.errstate(divide="ignore", invalid="ignore"):
        g = np.nanmax(np.abs(1.0 - (lam_all[:, None] - sigma) / (lam_defl[None, :] - sigma)))
    proj_every = 1
    if np.isfinite(g):
        proj_every = int(max(1, min(4, np.floor(np.log(1e6) / np.log(max(4.0 * g, 1.0 + 1e-12))))))
    if tuning.cg_projection_period:
        proj_every = int(tuning.cg_projection_period)
    LAST_ROUND["cg_projection_period"] = proj_every
    LAST_ROUND["cg_solution"] = "from the z history" if deferred else "recurrence"

    def sptr(lo):
        return state.cols(lo, 64).ptr

    def zslab(j):
        """the block step j's sweep writes: a slab of the history (deferred solution) or the one work block"""
        if not deferred:
            return z
        q, i = divmod(j - 1, _CG_CHUNK)
        while len(zchunks) <= q:
            zchunks.append(ctx.workspace_stack(("cg_z", len(zchunks), n, k), _CG_CHUNK, n, k))
        return zchunks[q][i]

    if deferred:
        # the history stacks are kept between calls per block shape; a caller that walks through many widths keeps the two
        # last ones only (16 slabs of n x k each per allocation)
        shapes = ctx.__dict__.setdefault("_cg_z_shapes", [])
        if (n, k) in shapes:
            shapes.remove((n, k))
        shapes.append((n, k))
        for old in shapes[:-2]:
            for tag in [t for t in ctx.__dict__.get("_ws", {}) if isinstance(t, tuple) and t[0] == "cg_z" and t[2:] == old]:
                del ctx.__dict__["_ws"][tag]
        del shapes[:-2]

    norms_of = {}                                          # step -> device block of its residual norms (unprojected steps)

    def project_in(j):
        return j % proj_every == 0 or j >= maxsteps

    def first_part(j, lo, hi, n2, n2_lo):
        """sweep of the residual of step j - 1 (into the history slab of step j), product with the step's inner products,
        coefficients, recurrence of the residual (no synchronisation)"""
        nonlocal r, r_old, psi, psi_old
        kk = hi - lo
        try:
            zj = zslab(j)
        except _ffi.EigdHipError:
            failed.append(j)
            return
        rv, zv, yv = r.cols(lo, hi), zj.cols(lo, hi), y.cols(lo, hi)
        prob.fac.apply_to(rv, zv, count=0)                # ref 1248
        n2p = None if n2 is None else n2.cols(lo - n2_lo, hi - n2_lo).ptr
        logp = log.cols(lo, 64).ptr if deferred else None
        if tuning.cg_dots_in_spmm:
            # ref 1250 / 1252 with the two inner products of the step formed in the product's own pass
            call("eigd_spmm_cg", ctx.h, Kop.csr.h, kk, zv.ptr, zv.ld, yv.ptr, yv.ld, rv.ptr, rv.ld, n2p, sptr(lo), int(j),
                 1 if j == 1 else 0, logp)
        else:
            Kop.apply(zv, yv)                             # ref 1250 / 1252
            call("eigd_cg_coefficients", ctx.h, n, kk, zv.ptr, zv.ld, rv.ptr, rv.ld, yv.ptr, yv.ld, n2p, sptr(lo), int(j),
                 1 if j == 1 else 0, logp)
        rov = r_old.cols(lo, hi)
        own = None if project_in(j) else ctx.empty(1, kk)  # (a step that is not projected forms its own residual norms)
        if deferred:
            call("eigd_cg_update", ctx.h, n, kk, rv.ptr, rv.ld, rov.ptr, rov.ld, None, 0, None, 0, None, 0,
                 yv.ptr, yv.ld, sptr(lo), 1 if j == 1 else 0, own.ptr if own is not None else None)
        else:
            psv, pov = psi.cols(lo, hi), psi_old.cols(lo, hi)
            call("eigd_cg_update", ctx.h, n, kk, rv.ptr, rv.ld, rov.ptr, rov.ld, psv.ptr, psv.ld, pov.ptr, pov.ld, zv.ptr,
                 zv.ld, yv.ptr, yv.ld, sptr(lo), 1 if j == 1 else 0, own.ptr if own is not None else None)
            psi, psi_old = psi_old, psi                    # (all columns of a block share the parity: the ranges lag one
        norms_of[j] = own                                  # step behind the flags, see below)
        r, r_old = r_old, r

    lo, hi = _active_range(done)
    # (ref 1232 projects the start residual once more: the caller has just done that, 1193 -- nothing to take out)
    # Pipeline.  Step j = first_part(j) [sweep .. recurrences] + projection of the new residual, whose norms the host
    # needs to know who has finished.  first_part(j + 1) is put in flight BEFORE the host waits for the norms of step j
    # (unless every live mode is expected to finish with step j -- its last reduction applied once more, within a factor
    # of four of the tolerance: a wrong 'finishes' costs a pipeline bubble of ~0.1 ms, a wrong 'goes on' a narrow step of ~1.3 ms --),
    # on the column range of the modes that were unfinished after step j - 1 (speculative or not): a mode that finishes in
    # step j is frozen on the device in step j + 1 (its residual copied into both buffers of the recurrence, its later
    # coefficients of psi zero) and leaves the range in step j + 2, so every column of a block has seen the same number of
    # buffer swaps when it stops moving.
    first_part(1, lo, hi, None, lo)
    if host_work is not None:
        host_work()                                        # (the caller's host-side work, under the first step's kernels)
    rng_j = (lo, hi)                                       # range of the step whose projection is next
    nsteps = 0
    j = 1
    while True:
        lo, hi = rng_j
        if _CG_TRACE_HOOK is not None:                     # (probes: the residual of step j before its projection)
            _CG_TRACE_HOOK(prob, j, r.cols(lo, hi), lo, hi, project_in(j))
        if project_in(j):
            # ref 1257 + the residual norm of 1275; measured update.  Against the N requested pairs only: along an extra
            # pair (lam_j above every lam_i) the eigenvalue of C_i lies in (0, 1) -- rounding there shrinks from step
            # to step like any other part of the residual, it is the pairs BELOW a mode that amplify
            if j > 1 and proj_every > 1 and tuning.cg_project_previous:
                # the previous residual takes part in the next step's recurrence: what it carries along the deflated
                # pairs would come back with it
                prob.project_r_norm2(r_old.cols(lo, hi), tol=proj_tol, requested_only=not tuning.cg_project_extra_pairs)
            n2 = prob.project_r_norm2(r.cols(lo, hi), tol=proj_tol, requested_only=not tuning.cg_project_extra_pairs)
        else:
            n2 = norms_of[j]                               # (formed by the update kernel of this step)
            call("eigd_colnorm2_publish", ctx.h, n2.ptr, hi - lo)
        norms_of.pop(j - 1, None)
        nsteps = j
        live = np.flatnonzero(~done)
        expect_all = tuning.predict_finish and j > 1 and all(
            len(hist[c]) >= 2 and hist[c][-2] > 0.0 and hist[c][-1] * min(hist[c][-1] / hist[c][-2], 1.0) < 4.0 * tol
            for c in live)
        nxt = _active_range(done)                          # (flags after step j - 1)
        in_flight = False
        if j < maxsteps and not expect_all:
            # (the pinned copy of the norms was issued behind the projection: nothing enqueued here overwrites it)
            first_part(j + 1, nxt[0], nxt[1], n2, lo)
            in_flight = True
        norms2 = ctx.fetch_colnorm2(hi - lo)
        for c in range(lo, hi):
            if done[c]:
                continue
            hist[c].append(float(np.sqrt(max(norms2[c - lo], 0.0))))
            if norms2[c - lo] < tol2:                     # ref 1275 (the comparison the device makes in eigd_cg_coefficients)
                info[c] = j
                done[c] = converged[c] = True
        if failed:
            break
        # A column that broke down stops moving (gam = 0: flag 2 on the device) and never meets the tolerance; a
        # recurrence that has lost its footing stalls.  Both are visible in the norms the host reads anyway: give up at
        # once -- not after maxsteps sweeps and as many slabs of z history -- and let the Arnoldi form decide.
        for c in np.flatnonzero(~done):
            h = hist[c]
            frozen = len(h) >= 3 and h[-1] > 0.0 and abs(h[-1] - h[-2]) <= 1e-14 * h[-1] and abs(h[-2] - h[-3]) <= 1e-14 * h[-2]
            if frozen or (len(h) > 12 and not h[-1] < 0.5 * h[-11]):
                gave_up.append(int(c))
        if gave_up:
            break
        if done.all() or j == maxsteps:
            # (a step in flight behind the last one only copies: every column is frozen by then, both buffers hold psi)
            LAST_ROUND["cg_sweeps_for_nothing"] = LAST_ROUND.get("cg_sweeps_for_nothing", 0) + int(in_flight)
            break
        if not in_flight:
            LAST_ROUND["cg_waited_for"] = LAST_ROUND.get("cg_waited_for", 0) + 1
            first_part(j + 1, nxt[0], nxt[1], n2, lo)
            if failed:                                     # (no memory for the step's slab: nothing was enqueued)
                break
        rng_j = nxt
        j += 1
    if deferred and not failed and not gave_up:
        # psi = sum_j s_j z_j, coefficients and sum on the device, enqueued before the host looks at the flags (a solve
        # that ends in the Arnoldi form throws the block away)
        nlog = nsteps + 1
        Sd = ctx.empty(nlog, k)
        call("eigd_cg_solution_coefficients", ctx.h, k, log.ptr, nlog, Sd.ptr)
        for q, chunk in enumerate(zchunks):
            a, b = q * _CG_CHUNK, min(nlog, (q + 1) * _CG_CHUNK)
            if a < b:
                call("eigd_stack_axpy_dev", ctx.h, n, k, b - a, chunk.ptr, chunk.slab, chunk.k, Sd.rows(a, b).ptr, psi.ptr, psi.ld, 1.0)
    st = state.get()
    # flag 2: r^T F r or <r, C r>_F not positive (the column stopped moving); flag 1: a step taken with rho = 1 because the
    # recurrence's denominator was not positive in finite precision -- a restart from the current iterate, counted only
    ok = not failed and not np.any(st[_CG_ROWS["flag"], :k] == 2.0) and bool(np.all(np.isfinite(st[_CG_ROWS["rr"], :k])))
    LAST_ROUND["cg_restarted_modes"] = LAST_ROUND.get("cg_restarted_modes", 0) + int(np.count_nonzero(st[_CG_ROWS["flag"], :k] == 1.0))
    # an unfinished mode whose residual has not halved over its last ten steps: the recurrence is not converging (the
    # caller's Phi is not invariant enough for the deflated operator to stay positive definite in finite precision, or
    # lam is not the eigenvalue of its column): the Arnoldi form, which minimises the true residual step by step, decides
    stalled = sorted(set(gave_up) | {c for c in range(k) if not converged[c] and len(hist[c]) > 12 and not hist[c][-1] < 0.5 * hist[c][-11]})
    if stalled:
        ok = False
    if not ok and deferred:
        # the z history of a failed attempt goes before the Arnoldi form allocates its own stacks (16 slabs of n x k per chunk)
        zchunks.clear()
        for tag in [t for t in ctx.__dict__.get("_ws", {}) if isinstance(t, tuple) and t[0] == "cg_z" and t[2:] == (n, k)]:
            del ctx.__dict__["_ws"][tag]
    if not ok:                                             # (why the Arnoldi form takes over: for the caller's log)
        LAST_ROUND["cg_exit"] = {"steps": nsteps, "broke_down": [int(c) for c in np.flatnonzero(st[_CG_ROWS["flag"], :k] == 2.0)],
                                 "stalled": stalled, "no_memory_at_step": failed[:1],
                                 "breakdown_saw": {int(c): (float(st[10, c]), float(st[11, c]), int(st[12, c]))
                                                   for c in np.flatnonzero(st[_CG_ROWS["flag"], :k] == 2.0)} if nrows > 12 else {},
                                 "residual_over_tolerance": {int(c): float(hist[c][-1] / tol) for c in range(k) if not converged[c]}}
    prob.project_s(psi)                                    # what the measured projections let pass (see proj_tol)
    if prob.fac.native and ok:                            # one factor application per step and mode (ref 1248)
        with prob.fac.factor._count_lock:
            prob.fac.factor.count += int(sum((i if i is not None else nsteps) for i in info))
    measured, applied = ctx.project_stats()
    LAST_ROUND["cg_steps"] = LAST_ROUND.get("cg_steps", 0) + nsteps
    LAST_ROUND["cg_projections"] = LAST_ROUND.get("cg_projections", 0) + measured
    LAST_ROUND["cg_projection_updates"] = LAST_ROUND.get("cg_projection_updates", 0) + applied
    return psi, converged, info, ok


def _sibk_cg(prob, R, lam_c, sigma, rnorm0, rtol, atol, maxsteps, hist, host_work=None):
    """the short-recurrence solver over all columns of R, 64 at a time; (update, converged, info, ok)"""
    k = R.k
    upd = prob.ctx.zeros(prob.n, k) if k > 64 else None
    conv, info, ok = np.zeros(k, dtype=bool), [None] * k, True
    for a in range(0, k, 64):
        b = min(k, a + 64)
        ua, ca, ia, oka = _sibk_cg_round(prob, R.cols(a, b) if k > 64 else R, lam_c[a:b], sigma, rnorm0, rtol, atol, maxsteps,
                                         hist[a:b], host_work=host_work if a == 0 else None)
        if k > 64:
            upd.cols(a, b).copy_from(ua)
        else:
            upd = ua
        conv[a:b] = ca
        info[a:b] = ia
        ok = ok and oka
        if not ok:
            break
    return upd, conv, info, ok


def _expected_to_finish(hist, done, in_cycle, tol, margin):
    """
    Per mode: True if its residual, extrapolated over the two steps of the cycle being solved with the reduction per step
    of its last cycle, falls below ``margin`` times its tolerance.  ``hist[c]`` holds the residual norms of mode c so far
    (entry 0: the start residual); only modes of the cycle in flight (``in_cycle``) with three entries can be judged.
    ``_PREDICT_HOOK`` (tests): a function (c, verdict) -> verdict.
    """
    out = np.zeros(len(done), dtype=bool)
    for c in np.flatnonzero(~np.asarray(done) & in_cycle):
        hc = hist[c]
        v = False
        if len(hc) >= 3 and hc[-3] > 0.0:
            v = bool(hc[-1] * min(hc[-1] / hc[-3], 1.0) < margin * tol)
        out[c] = _PREDICT_HOOK(c, v) if _PREDICT_HOOK is not None else v
    return out


_PREDICT_HOOK = None


def _default_streams():
    import os

    return max(1, int(os.environ.get("EIGD_STREAMS", "1")))


def _run_groups(prob, Rc, lam_p, sigma, rnorm0, rtol, atol, maxiter, sub_hist, streams):
    """
    One attempt for the columns of Rc, optionally split into `streams` interleaved mode groups that run
    concurrently: each group has its own context (HIP stream), Krylov workspaces and sweep lane, and is driven
    by its own host thread (ctypes releases the GIL inside the library), so the dependent launch chain of one
    group's triangular sweep overlaps the other groups' work.  Groups never exchange data.
    """
    k = Rc.k
    groups = max(1, min(int(streams), k))
    if groups == 1:
        # the pair form orthogonalises 2 x min(k, 32) columns against up to maxiter slabs in one coefficient block of the
        # Gram-Schmidt kernels (60 KB of LDS): deeper histories (maxiter > 120 at 32 columns) take the one-step form,
        # whose Gram-Schmidt goes pass by pass beyond that depth
        fits = maxiter * 2 * min(k, 32) * 8 <= 60 * 1024
        if tuning.steps_per_pass == 2 and fits:
            keep = [list(hh) for hh in sub_hist]
            if 2 * k <= 64:
                upd, conv, inf, ok = _sibk_round_pair(prob, Rc, lam_p, sigma, rnorm0, rtol, atol, maxiter, sub_hist)
            else:
                # a pair of blocks holds 2 x 32 columns: wider problems go chunk by chunk (the sweeps take 32 columns at
                # a time anyway, and the low modes, which finish first, share a chunk)
                upd = prob.ctx.zeros(Rc.n, k)
                conv, inf, ok = np.zeros(k, dtype=bool), [None] * k, True
                for a in range(0, k, 32):
                    b = min(k, a + 32)
                    ua, ca, ia, oka = _sibk_round_pair(prob, Rc.cols(a, b), lam_p[a:b], sigma, rnorm0, rtol, atol, maxiter,
                                                       sub_hist[a:b])
                    upd.cols(a, b).copy_from(ua)
                    conv[a:b] = ca
                    inf[a:b] = ia
                    ok = ok and oka
                    if not ok:
                        break
            if ok:
                return upd, conv, inf
            for hh, h0 in zip(sub_hist, keep):            # (a pair lost orthogonality: the one-step form decides)
                hh[:] = h0
            for tag in ("krylov_W2", "krylov_Z2"):        # its stacks go before the one-step stacks are allocated
                prob.ctx.__dict__.get("_ws", {}).pop(tag, None)
        LAST_ROUND["steps_per_pass"] = 1
        return _sibk_round(prob, Rc, lam_p, sigma, rnorm0, rtol, atol, maxiter, sub_hist)
    import threading

    parts = [np.arange(g, k, groups) for g in range(groups)]
    Rg = [Rc.gather_cols(part) for part in parts]
    prob.ctx.sync()
    out = [None] * groups
    errors = []

    def work(g):
        try:
            ctxg = prob.ctx.fork(g)
            ctxg.make_current()   # a fresh host thread: HIP's current device is per thread
            pg = prob.on(ctxg)
            R0 = ctxg.empty(Rc.n, len(parts[g])).copy_from(Rg[g])
            hg = [sub_hist[c] for c in parts[g]]
            kg = len(parts[g])
            res = None
            if tuning.steps_per_pass == 2 and 2 * kg <= 64 and maxiter * 2 * min(kg, 32) * 8 <= 60 * 1024:
                keep = [list(hh) for hh in hg]
                ug, cg, ig, okg = _sibk_round_pair(pg, R0, lam_p[parts[g]], sigma, rnorm0, rtol, atol, maxiter, hg)
                if okg:
                    res = (ug, cg, ig)
                else:
                    for hh, h0 in zip(hg, keep):
                        hh[:] = h0
            out[g] = res if res is not None else _sibk_round(pg, R0, lam_p[parts[g]], sigma, rnorm0, rtol, atol, maxiter, hg)
            ctxg.sync()
        except BaseException as exc:  # re-raised on the calling thread
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(g,)) for g in range(groups)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    upd = prob.ctx.zeros(Rc.n, k)
    conv = np.zeros(k, dtype=bool)
    info = [None] * k
    for g, part in enumerate(parts):
        ug, cg, ig = out[g]
        prob.ctx.empty(Rc.n, len(part)).copy_from(ug).scatter_cols_into(upd, part)
        conv[part] = cg
        for q, c in enumerate(part):
            info[c] = ig[q]
    return upd, conv, info


def _sibk_device(prob, dPhib, dpsi, lam_c, sigma, rtol, atol, maxiter, nrestart, callback, rnorm0=None, streams=None,
                 host_work=None):
    """
    Lock-step sibk (bs_target = 1, update_guess = False) on the columns of dPhib / dpsi.
    dpsi is updated in place; returns the info list.
    """
    k = dPhib.k
    lam_c = np.asarray(lam_c, dtype=float)
    if rnorm0 is None:
        rnorm0 = _rnorm0(dPhib)
    if streams is None:
        streams = _default_streams()
    R = prob.residual(dPhib, dpsi, lam_c)                # ref 1189-1192
    prob.project_r(R)                                    # ref 1193
    hist = [[] for _ in range(k)]
    info = [None] * k
    LAST_ROUND["recurrence"] = "arnoldi"
    for key in [key for key in LAST_ROUND if key.startswith("cg_")]:
        del LAST_ROUND[key]                                # (the cg_* entries describe this call)
    if streams == 1 and _short_recurrence_applies(prob):
        # positive definite shift: conjugate gradients in the factor inner product (same spaces, no history); as many
        # steps as the reference's restarted loop may take in all (1312-1321)
        upd, conv, inf, ok = _sibk_cg(prob, R, lam_c, sigma, rnorm0, rtol, atol, maxiter * (nrestart + 1), hist,
                                      host_work=host_work)
        if ok:
            LAST_ROUND["recurrence"] = "short"
            dpsi.assign_lincomb([(1.0, dpsi), (1.0, upd)])
            _emit(callback, hist, range(k))
            return [i for i in inf if i is not None]
        LAST_ROUND["recurrence"] = "arnoldi (the short recurrence broke down or stalled)"
        hist = [[] for _ in range(k)]
        R = prob.residual(dPhib, dpsi, lam_c)            # (the short recurrence worked in the residual block)
        prob.project_r(R)
    pending = np.arange(k)
    for attempt in range(nrestart + 1):                  # ref 1312-1321: restarts reuse the same residual
        Rc = R if len(pending) == k else R.gather_cols(pending)
        sub_hist = [hist[c] for c in pending]
        upd, conv, inf = _run_groups(prob, Rc, lam_c[pending], sigma, rnorm0, rtol, atol, maxiter, sub_hist, streams)
        if len(pending) == k:
            dpsi.assign_lincomb([(1.0, dpsi), (1.0, upd)])
        else:
            full = prob.ctx.zeros(prob.n, k)
            upd.scatter_cols_into(full, pending)
            dpsi.assign_lincomb([(1.0, dpsi), (1.0, full)])
        for q, c in enumerate(pending):
            if inf[q] is not None:
                info[c] = inf[q]
        pending = pending[~conv]
        if len(pending) == 0:
            break
    _emit(callback, hist, range(k))
    return [i for i in info if i is not None]


def _sibk_sequential(prob, dPhib, dpsi, lam, sigma, rtol, atol, maxiter, bs_target, update_guess, callback, nrestart):
    """
    General form (block size > 1 and / or update_guess): the reference's sequential algorithm
    (ref 1195-1321) with every n-vector operation on the device.  W is a k=1 stack.
    """
    ctx, mode, n = prob.ctx, prob.mode, prob.n
    N = dPhib.k
    Kop = prob.opB if mode == "normal" else prob.opA
    sgn = 1.0 if mode == "normal" else -1.0
    rnorm0 = _rnorm0(dPhib)
    W = ctx.stack(maxiter + bs_target, n, 1)
    Z = ctx.stack(maxiter, n, 1)
    R = prob.residual(dPhib, dpsi, lam)
    prob.project_r(R)
    info = []
    i = 0
    restart = 0
    while i < N:
        r = np.zeros((maxiter + bs_target, bs_target))
        bs = 0
        while i + bs < N and bs < bs_target:
            kk = i + bs
            Wb = W[bs]
            if update_guess:
                pk = dpsi.cols(kk, kk + 1)
                prob.project_s(pk)
                Wb.copy_from(prob.residual(dPhib.cols(kk, kk + 1), pk, lam[kk: kk + 1]))
                prob.project_r(Wb)
            else:
                Wb.copy_from(R.cols(kk, kk + 1))
            beta0 = float(Wb.colnorms()[0])
            if callback is not None:
                callback(beta0)
            if beta0 < rtol * rnorm0 or beta0 < atol:
                info.append(0)
                break
            if bs > 0:
                hh = W.dot(Wb, ns=bs)
                W.axpy_into(Wb, hh, alpha=-1.0)
                r[:bs, bs] = hh[:, 0]
            prob.project_r(Wb)
            r[bs, bs] = float(Wb.colnorms()[0])
            Wb.assign_lincomb([(1.0 / r[bs, bs], Wb)])
            bs += 1
        if bs == 0:
            i += 1
            continue
        H = np.zeros((maxiter + bs, maxiter))
        y = np.zeros((maxiter, bs))
        for j in range(bs, maxiter + bs):
            kp = j - bs
            Zk = Z[kp]
            Zk.copy_from(W[kp])
            prob.fac(Zk)
            Wj = W[j]
            Kop.apply(Zk, Wj)
            prob.project_r(Wj)
            H[:j, kp] = _cgs2(W, Wj, j)[:, 0]
            prob.project_r(Wj)
            H[j, kp] = float(Wj.colnorms()[0])
            Wj.assign_lincomb([(1.0 / H[j, kp], Wj)])
            res = 0.0
            H0 = H[: j + 1, : j + 1 - bs]
            for q in range(bs):
                alpha = sgn * (lam[i + q] - sigma)
                y[: kp + 1, q], res0 = solve_shifted_lstsq(alpha, H0, r[: j + 1, q])
                res = max(res, res0)
            if callback is not None:
                callback(res)
            finished = res < rtol * rnorm0 or res < atol
            if finished or j == maxiter + bs - 1:
                # psi[:, i:i+bs] += Z[:, :j] @ y[:j]   (ref 1277 / 1313; rows beyond kp of y are zero)
                blk = dpsi.cols(i, i + bs)
                Z.times_into(blk, y[: kp + 1, :], ns=kp + 1, alpha=1.0, beta=1.0)
            if finished:
                info.append(j)
                if update_guess and i + bs < N:
                    rest = R.cols(i + bs, N)
                    r0 = W.tdot_block(rest, ns=j + 1)
                    nrest = N - (i + bs)
                    y0 = np.zeros((j + 1 - bs, nrest))
                    t0 = np.zeros((j + 1, nrest))
                    for q in range(nrest):
                        alpha = sgn * (lam[i + bs + q] - sigma)
                        yk, _ = solve_shifted_lstsq(alpha, H0, r0[:, q])
                        y0[:, q] = yk
                        t0[:, q] = -alpha * H0 @ yk
                        t0[:-bs, q] += yk
                    Z.times_into(dpsi.cols(i + bs, N), y0, ns=j + 1 - bs, alpha=1.0, beta=1.0)
                    W.times_into(rest, t0, ns=j + 1, alpha=-1.0, beta=1.0)
                i += bs
                restart = 0
                break
            elif j == maxiter + bs - 1:
                if restart >= nrestart:
                    restart = 0
                    i += bs
                    break
                restart += 1
    return info


def sibk(Phib, A, B, lam, Phi, mode="normal", psi=None, sigma=None, factor=None, rtol=1e-10, atol=1e-30,
         eig_atol=1e-5, maxiter=50, bs_target=1, update_guess=False, callback=None, nrestart=2, ctx=None):
    n, N = _check_iter_args(Phib, A, B, lam, Phi, psi, mode)
    ctx = _ctx_of(factor, ctx)
    if factor is None:
        factor, sigma = _default_factor(A, B, lam, sigma, mode, ctx)
    prob = DeviceProblem(ctx, A, B, factor, mode)
    prob.set_phi(Phi_host=Phi)
    _psi = psi if psi is not None else np.zeros((n, N), dtype=Phib.dtype)
    dpsi = ctx.from_host(_psi)
    dPhib = ctx.from_host(Phib)
    lam = np.asarray(lam, dtype=float)
    prob.lam_phi = lam
    G = -prob.Phi.tdot(dPhib)                            # ref 1180
    Glo = refine_repeated_entries(G, lam, prob.Phi, dPhib, eig_atol)
    if bs_target == 1 and not update_guess:
        info = _sibk_device(prob, dPhib, dpsi, lam, sigma, rtol, atol, maxiter, nrestart, callback)
    else:
        info = _sibk_sequential(prob, dPhib, dpsi, lam, sigma, rtol, atol, maxiter, bs_target, update_guess,
                                callback, nrestart)
    Cc, data = correction_coefficients(lam, G, eig_atol, mode, Glo)  # ref 1324-1326
    _apply_correction(dpsi, prob.Phi, Cc)
    writable_result(_psi)[:] = dpsi.get()
    return _psi, data, info


# ---------------------------------------------------------------------------
# PGMRES, lock-step batched (ref 872-1040)
# ---------------------------------------------------------------------------
def _pgmres_device(prob, dPhib, dpsi, lam_c, rtol, atol, maxiter, callback, rnorm0=None, refine=None):
    ctx = prob.ctx
    k = dPhib.k
    lam_c = np.asarray(lam_c, dtype=float)
    if rnorm0 is None:
        rnorm0 = _rnorm0(dPhib)
    R = prob.residual(dPhib, dpsi, lam_c)
    Gc = prob.Phi.tdot(R)                                # ref 989: G[:, i] = Phi^T R
    Gclo = refine(Gc, R) if refine is not None else None
    R.add_product(prob.BPhi, Gc, alpha=-1.0, beta=1.0)   # ref 990
    beta = R.colnorms()
    hist = [[float(beta[c])] for c in range(k)]
    info = [None] * k
    done = np.zeros(k, dtype=bool)
    for c in range(k):
        if beta[c] < rtol * rnorm0 or beta[c] < atol:
            info[c] = 0
            done[c] = True
    if not done.all():
        W = ctx.workspace_stack("krylov_W", maxiter + 1, prob.n, k)
        Z = ctx.workspace_stack("krylov_Z", maxiter, prob.n, k).zero()
        scale = np.where(done | (beta == 0.0), 0.0, 1.0 / np.where(beta == 0.0, 1.0, beta))
        W[0].assign_lincomb([(scale, R)])
        H = np.zeros((k, maxiter + 1, maxiter))
        Ycoef = np.zeros((maxiter, k))
        T = ctx.empty(prob.n, k)
        jlast = 0
        for j in range(maxiter):
            lo, hi = _active_range(done)
            Zj, Ta = Z[j].cols(lo, hi), T.cols(lo, hi)
            Zj.copy_from(W[j].cols(lo, hi))
            prob.project_r(Zj)                           # ref 963: oper = factor(project(x))
            prob.fac(Zj, count=int(np.count_nonzero(~done)))
            prob.adjoint_operator(Zj, lam_c[lo:hi], out=Ta)  # ref 1004-1010
            prob.project_r(Ta)
            h = _cgs2(W, Ta, j + 1, c0=lo)               # ref 1012-1014
            hn = Ta.colnorms()                           # ref 1016
            jlast = j + 1
            for c in range(lo, hi):
                if done[c]:
                    continue
                H[c, : j + 1, j] = h[:, c - lo]
                H[c, j + 1, j] = hn[c - lo]
                rhs = np.zeros(j + 2)
                rhs[0] = beta[c]
                Hj = H[c, : j + 2, : j + 1]
                y = np.linalg.lstsq(Hj, rhs, rcond=None)[0]
                res = np.linalg.norm(Hj.dot(y) - rhs)
                hist[c].append(res)
                if res < rtol * rnorm0 or res < atol:
                    Ycoef[: j + 1, c] = y
                    info[c] = j
                    done[c] = True
                elif j == maxiter - 1:
                    Ycoef[: j + 1, c] = y
                    info[c] = -1
                    done[c] = True
            if done.all():
                break
            dn = done[lo:hi]
            scale = np.where(dn | (hn == 0.0), 0.0, 1.0 / np.where(hn == 0.0, 1.0, hn))
            W[j + 1].cols(lo, hi).assign_lincomb([(scale, Ta)])
        Z.axpy_into(dpsi, Ycoef[:jlast], alpha=1.0)      # ref 1028 / 1032
    _emit(callback, hist, range(k))
    return (Gc, Gclo), info


def pgmres(Phib, A, B, lam, Phi, mode="normal", psi=None, sigma=None, factor=None, rtol=1e-10, atol=1e-30,
           eig_atol=1e-5, maxiter=50, callback=None, ctx=None):
    n, N = _check_iter_args(Phib, A, B, lam, Phi, psi, mode)
    ctx = _ctx_of(factor, ctx)
    if factor is None:
        factor, sigma = _default_factor(A, B, lam, sigma, mode, ctx)
    prob = DeviceProblem(ctx, A, B, factor, mode)
    prob.set_phi(Phi_host=Phi)
    _psi = psi if psi is not None else np.zeros((n, N), dtype=Phib.dtype)
    dpsi = ctx.from_host(_psi)
    lam = np.asarray(lam, dtype=float)
    (G, Glo), info = _pgmres_device(prob, ctx.from_host(Phib), dpsi, lam, rtol, atol, maxiter, callback,
                                    refine=lambda Gm, R: refine_repeated_entries(Gm, lam, prob.Phi, R, eig_atol, sign=1.0))
    Cc, data = correction_coefficients(lam, G, eig_atol, mode, Glo)
    _apply_correction(dpsi, prob.Phi, Cc)
    writable_result(_psi)[:] = dpsi.get()
    return _psi, data, info


# ---------------------------------------------------------------------------
# PCPG, lock-step batched (ref 699-869)
# ---------------------------------------------------------------------------
def _pcpg_device(prob, dPhib, dpsi, lam_c, rtol, atol, maxiter, reset, callback, rnorm0=None, refine=None):
    ctx = prob.ctx
    k = dPhib.k
    lam_c = np.asarray(lam_c, dtype=float)
    if rnorm0 is None:
        rnorm0 = _rnorm0(dPhib)
    R = prob.residual(dPhib, dpsi, lam_c)
    Gc = prob.Phi.tdot(R)                                # ref 810
    Gclo = refine(Gc, R) if refine is not None else None
    R.add_product(prob.BPhi, Gc, alpha=-1.0, beta=1.0)   # ref 811
    hist = [[] for _ in range(k)]
    conv = [False] * k
    done = np.zeros(k, dtype=bool)
    P0 = ctx.zeros(prob.n, k)
    P = ctx.empty(prob.n, k)
    Zb = ctx.empty(prob.n, k)
    Q = ctx.empty(prob.n, k)
    zTr_prev = np.ones(k)
    for it in range(maxiter):
        res = R.colnorms()
        for c in range(k):
            if done[c]:
                continue
            hist[c].append(float(res[c]))
            if res[c] < rtol * rnorm0 or res[c] < atol:  # ref 823-825
                conv[c] = True
                done[c] = True
        if done.all():
            break
        Zb.copy_from(R)
        prob.project_r(Zb)
        prob.fac(Zb, count=int(np.count_nonzero(~done)))
        prob.project_s(Zb)                               # ref 829-830
        zTr = Zb.coldot(R)
        if it % reset == 0:                              # ref 832-840
            P.copy_from(Zb)
        else:
            bcoef = np.where(done, 0.0, zTr / np.where(zTr_prev == 0.0, 1.0, zTr_prev))
            P.assign_lincomb([(1.0, Zb), (bcoef, P0)])
        zTr_prev = zTr
        prob.adjoint_operator(P, lam_c, out=Q)           # ref 843-848
        den = Q.coldot(P)
        alpha = np.where(done | (den == 0.0), 0.0, zTr / np.where(den == 0.0, 1.0, den))
        dpsi.assign_lincomb([(1.0, dpsi), (alpha, P)])   # ref 851
        R.assign_lincomb([(1.0, R), (-alpha, Q)])        # ref 855 / 857
        P0.copy_from(P)
    _emit(callback, hist, range(k))
    return (Gc, Gclo), conv


def pcpg(Phib, A, B, lam, Phi, mode="normal", psi=None, sigma=None, factor=None, rtol=1e-10, atol=1e-30,
         eig_atol=1e-5, maxiter=100, reset=25, callback=None, ctx=None):
    n, N = _check_iter_args(Phib, A, B, lam, Phi, psi, mode, check_lam=False)
    ctx = _ctx_of(factor, ctx)
    if factor is None:
        factor, sigma = _default_factor(A, B, lam, sigma, mode, ctx)
    prob = DeviceProblem(ctx, A, B, factor, mode)
    prob.set_phi(Phi_host=Phi)
    _psi = psi if psi is not None else np.zeros((n, N), dtype=Phib.dtype)
    dpsi = ctx.from_host(_psi)
    lam = np.asarray(lam, dtype=float)
    (G, Glo), info = _pcpg_device(prob, ctx.from_host(Phib), dpsi, lam, rtol, atol, maxiter, reset, callback,
                                  refine=lambda Gm, R: refine_repeated_entries(Gm, lam, prob.Phi, R, eig_atol, sign=1.0))
    Cc, data = correction_coefficients(lam, G, eig_atol, mode, Glo)
    _apply_correction(dpsi, prob.Phi, Cc)
    writable_result(_psi)[:] = dpsi.get()
    return _psi, data, info


# ---------------------------------------------------------------------------
# reverse-mode sweep through the Lanczos recurrence (ref 526-696)
# ---------------------------------------------------------------------------
def _dl_device(prob, dPhib, lam, sigma, indices, Vst, m, T, Y, theta, eig_atol, mode):
    """
    Vb is kept as a row-major n x m device block (columns = Lanczos indices); V and the work
    vectors are k=1 stacks / blocks.  m + 1 factor sweeps in total, independent of N.
    """
    ctx, n = prob.ctx, prob.n
    N = dPhib.k
    lam = np.asarray(lam, dtype=float)
    repeated = are_eigenvalues_repeated(lam, atol=eig_atol)
    sel = np.asarray(indices[:N])
    G = Glo = None
    if repeated:                                          # ref 607-617
        G = -prob.Phi.tdot(dPhib)
        Glo = refine_repeated_entries(G, lam, prob.Phi, dPhib, eig_atol)
        Rb = dPhib.copy().add_product(prob.BPhi, G, alpha=1.0, beta=1.0)
    else:
        Rb = dPhib
    Vb = ctx.empty(n, m).add_product(Rb, np.ascontiguousarray(Y[:, sel].T), alpha=1.0, beta=0.0)
    Yb = Vst.tdot_block(Rb, ns=m)
    D = np.zeros((m, m))
    for i in range(m):                                    # ref 622-631
        for j in range(N):
            ii, jj = indices[i], indices[j]
            if ii == jj:
                continue
            if i < N and j < N and _is_close(lam[i], lam[j], atol=eig_atol):
                continue
            D[ii, jj] = Y[:, ii].dot(Yb[:, j]) / (theta[jj] - theta[ii])
    Tb = Y @ (D @ Y.T)

    def col(blk, j):
        return blk.cols(j, j + 1)

    t = ctx.empty(n, 1)
    tmp = ctx.empty(n, 1)
    sb = ctx.empty(n, 1)
    u = ctx.empty(n, 1)
    # t = B factor(B V[:, m-1])                            ref 637
    prob.opB.apply(Vst[m - 1], tmp)
    prob.fac(tmp)
    prob.opB.apply(tmp, t)
    Vb.add_product(t, Tb[:m, m - 1].reshape(1, m))        # ref 639-640
    Vst.times_into(tmp, Tb[:m, m - 1].reshape(m, 1), ns=m)
    prob.opB.apply(tmp, sb)                               # ref 641
    u.copy_from(sb)
    prob.fac(u)                                           # ref 643
    prob.opB.apply(u, tmp)
    col(Vb, m - 1).assign_lincomb([(1.0, col(Vb, m - 1)), (1.0, tmp)])  # ref 644
    for i in range(m - 2, -1, -1):                        # ref 647-671
        lo = max(i - 1, 0)
        Vst_lo = _StackView(Vst, lo)
        Vst_lo.times_into(tmp, T[lo: i + 2, i].reshape(-1, 1), ns=i + 2 - lo)
        prob.opB.apply(tmp, t)                            # ref 649-652
        vi1 = Vst[i + 1]
        vbi1 = col(Vb, i + 1)
        c0 = float(vi1.coldot(vbi1)[0]) - T[i + 1, i] * Tb[i + 1, i]  # ref 654
        prob.opB.apply(vi1, tmp)
        sb.assign_lincomb([(1.0 / T[i + 1, i], vbi1), (-c0 / T[i + 1, i], tmp)])  # ref 655
        im1 = (i - 1) % m                                 # ref 657: column -1 when i == 0, as numpy indexes it
        col(Vb, im1).assign_lincomb([(1.0, col(Vb, im1)), (-T[i - 1, i], sb)])
        col(Vb, i).assign_lincomb([(1.0, col(Vb, i)), (-T[i, i], sb)])  # ref 658
        hb = Vst.tdot_block(sb, ns=i + 1)[:, 0] - Tb[: i + 1, i]  # ref 660
        crow = np.zeros((1, m))
        crow[0, : i + 1] = -hb
        Vb.add_product(t, crow)                           # ref 662-663
        Vst.times_into(tmp, hb.reshape(-1, 1), ns=i + 1)
        prob.opB.apply(tmp, t)                            # reuse t as scratch: B (V hb)
        sb.assign_lincomb([(1.0, sb), (-1.0, t)])         # ref 664
        vbi1.copy_from(u)                                 # ref 667
        u.copy_from(sb)
        prob.fac(u)                                       # ref 670
        prob.opB.apply(u, tmp)
        col(Vb, i).assign_lincomb([(1.0, col(Vb, i)), (1.0, tmp)])  # ref 671
    col(Vb, 0).copy_from(u)                               # ref 674
    scale = 1.0 if mode == "normal" else sigma
    Cf = -(scale * Y[:, sel] / (lam - sigma))
    psi = ctx.empty(n, N).add_product(Vb, Cf, alpha=1.0, beta=0.0)  # ref 677-680
    data = {}
    if repeated:                                          # ref 682-694
        prob.project_s(psi)
        Cc, data = correction_coefficients(lam, G, eig_atol, mode, Glo)
        _apply_correction(psi, prob.Phi, Cc)
    return psi, data


class _StackView:
    """a k=1 stack seen from slab `lo` on (for V[:, lo:hi] @ c products)"""

    def __init__(self, st, lo):
        self.st, self.lo = st, lo

    def times_into(self, X, Cmat, ns):
        from ._ffi import call, hptr

        st = self.st
        Cmat = np.ascontiguousarray(Cmat, dtype=np.float64).reshape(ns, X.k)
        call("eigd_gemm_nn", st.ctx.h, st.n, ns, X.k, st.slab_ptr(self.lo), 1, st.slab, hptr(Cmat), X.ptr, X.ld, 1.0,
             0.0)
        return X


def dl(Phib, B, factor, sigma, lam, Phi, indices, V, T, Y, theta, eig_atol=1e-5, mode="normal", ctx=None):
    n = B.shape[1]
    m = len(theta)
    N = Phib.shape[1]
    _check_mode(mode)
    if len(lam) != N:
        raise ValueError(f"Eigenvalues must be of length {N}")
    if Phib.shape != (n, N):
        raise ValueError(f"Right-hand-side must have the shape ({n},{N})")
    if B.shape != (n, n):
        raise ValueError(f"B must have dimensions ({n},{n})")
    if factor.shape != (n, n):
        raise ValueError(f"Factorized operator must have dimensions ({n},{n})")
    if len(indices) != m:
        raise ValueError(f"Length of indices array must be (m = {m})")
    if V.shape != (n, m):
        raise ValueError(f"Dimension of the Lanczos subspace must be ({n},{m})")
    ctx = _ctx_of(factor, ctx)
    prob = DeviceProblem(ctx, B, B, factor, mode)
    prob.set_phi(Phi_host=Phi)
    psi, data = _dl_device(prob, ctx.from_host(Phib), lam, sigma, np.asarray(indices), _vstack_from_host(ctx, V), m,
                           np.asarray(T), np.asarray(Y), np.asarray(theta), eig_atol, mode)
    return psi.get(), data
