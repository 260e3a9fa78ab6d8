"""
CPU oracle for the eigd adjoint eigenvector-derivative hot path.

THIS FILE IS TEST INFRASTRUCTURE.  It is a numpy/scipy restatement of the
algorithms of smdogroup/eigd (``eigd/eigenvector_derivatives.py`` and
``eigd/arpack.py``) and is imported only by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.
The product (``eigd_amd``) never imports it and has no CPU fallback.

Parity pin: every function here is checked against fixtures captured from the
reference itself (``tools/make_golden.py`` -> ``tests/golden/*.npz``) by
``tests/test_oracle_golden.py``.  Third-party arithmetic below the reference
(SuperLU / ARPACK / LAPACK through scipy 1.15.3 + numpy 2.2.6) is reached the
same way the reference reaches it.

Reference citations are ``eigenvector_derivatives.py:<line>`` unless stated.
"""

import warnings

import numpy as np
from scipy.sparse.linalg import LinearOperator, aslinearoperator, splu

MODES = ("normal", "buckling")
METHODS = ("pcpg", "pgmres", "sibk", "laa", "dl")


def _check_mode(mode):
    if mode not in MODES:
        raise ValueError(f"Unknown mode {mode!r}")


# --------------------------------------------------------------------------
# operator wrapper (ref 11-23)
# --------------------------------------------------------------------------
class SpLuOperator(LinearOperator):
    """SuperLU shift-invert operator with an application counter (ref 11-23)."""

    def __init__(self, mat):
        self.lu = splu(mat)
        self.shape = mat.shape
        self.dtype = mat.dtype
        self.count = 0

    def _matvec(self, x):
        self.count += x.shape[1] if x.ndim == 2 else 1
        return self.lu.solve(x.astype(self.dtype))

    # scipy routes 2-D input of ``factor(X)`` through _matmat -> column loop by
    # default; the reference relies on LinearOperator.__call__ -> dot -> matmat,
    # whose default implementation stacks matvec results.  SuperLU handles 2-D
    # RHS natively, and the count bookkeeping of ref 19-22 expects that path.
    def _matmat(self, X):
        return self._matvec(X)


def project(U, V, X):
    """X <- X - U (V^T X), in place (ref 26-30)."""
    X[:] -= U @ (V.T @ X)
    return X


def is_close(a, b, atol=1e-5):
    """ref 278-281"""
    return bool(np.fabs(a - b) < atol)


def are_eigenvalues_repeated(lam, atol=1e-5):
    """Any consecutive pair closer than atol (ref 284-300)."""
    return any(is_close(lam[i], lam[i + 1], atol) for i in range(len(lam) - 1))


# --------------------------------------------------------------------------
# total derivative (ref 33-182)
# --------------------------------------------------------------------------
def derivative_weights(lam, Phi, lamb, Phib, psi, adj_corr_data, mode):
    """
    The two n x N weight matrices (WA, WB) such that
      normal  : dfdx += dAdx(WA_i, phi_i) - dBdx(WB_i, phi_i)   (ref 96-113)
      buckling: dfdx += dAdx(WA_i, phi_i) + dBdx(WB_i, phi_i)   (ref 118-134)
    """
    N = Phi.shape[1]
    beta = 0.5 * np.einsum("ij,ij->j", Phi, Phib)
    if mode == "normal":
        WA = Phi * lamb + psi
        WB = Phi * (beta + lam * lamb) + psi * lam
        ca, cb = 1, 2  # xi feeds A, eta feeds B
    else:
        WA = (Phi * lamb + psi) * lam
        WB = Phi * (lamb - beta) + psi
        ca, cb = 2, 1  # eta feeds A, xi feeds B
    for i in range(N):
        for tup in adj_corr_data.get(i, ()):
            j = tup[0]
            WA[:, i] += tup[ca] * Phi[:, j]
            WB[:, i] += tup[cb] * Phi[:, j]
    return WA, WB


def add_eig_total_derivative(
    lam, Phi, lamb, Phib, psi, dAdx, dBdx, dfdx, adj_corr_data={}, mode="normal", deriv_type="vector"
):
    n, N = Phi.shape
    _check_mode(mode)
    if len(lam) != N:
        raise ValueError(f"Eigenvalues must be of length {N}")
    for arr, what in ((psi, "Eigenvectors"), (Phi, "Eigenvectors"), (Phib, "Right-hand-side")):
        if arr.shape != (n, N):
            raise ValueError(f"{what} must have the shape ({n},{N})")

    WA, WB = derivative_weights(lam, Phi, lamb, Phib, psi, adj_corr_data, mode)
    sB = -1.0 if mode == "normal" else 1.0
    if deriv_type == "vector":
        for i in range(N):
            if dAdx is not None:
                dfdx += dAdx(WA[:, i].copy(), Phi[:, i])
            if dBdx is not None:
                dfdx += sB * dBdx(WB[:, i].copy(), Phi[:, i])
    elif deriv_type == "tensor":
        if dAdx is not None:
            dfdx += dAdx(WA, Phi)
        if dBdx is not None:
            dfdx += sB * dBdx(WB, Phi)
    return dfdx


# --------------------------------------------------------------------------
# residual check (ref 185-275)
# --------------------------------------------------------------------------
def eval_adjoint_residual_norm(A, B, lam, Phi, Phib, psi, mode="normal", b_ortho=False):
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

    BPhi = B @ Phi
    Apsi, Bpsi = A @ psi, B @ psi
    rhs = -(Phib - BPhi * np.einsum("ij,ij->j", Phi, Phib))
    if mode == "normal":
        R = (Apsi - Bpsi * lam) - rhs
    else:
        R = (Bpsi + Apsi * lam) - rhs
    if b_ortho:
        R = project(BPhi, Phi, R)
        ortho = np.max(np.abs(BPhi.T @ psi), axis=0)
    else:
        ortho = np.abs(np.einsum("ij,ij->j", BPhi, psi))
    return np.linalg.norm(R, axis=0), ortho


# --------------------------------------------------------------------------
# correction along the eigenvectors (ref 303-391)
# --------------------------------------------------------------------------
def generate_adjoint_correction(lam, Phi, psi, G=None, Phib=None, eig_atol=1e-5, mode="normal"):
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
        G = -Phi.T @ Phib
    else:
        if G.shape != (N, N):
            raise ValueError(f"G must have dimensions ({N},{N})")
        if Phi.shape != (n, N):
            raise ValueError(f"Phi must have dimensions ({n},{N})")

    G0 = G if mode == "normal" else np.diag(lam) @ G
    data = {}
    for i in range(N):
        for j in range(i):
            gap = lam[j] - lam[i]
            if is_close(lam[i], lam[j], atol=eig_atol):
                xi = 0.5 * (G0[j, i] - G0[i, j]) / gap
                eta = 0.5 * (lam[i] * G0[j, i] - lam[j] * G0[i, j]) / gap
                data.setdefault(i, [])
                data.setdefault(j, [])
                data[i].append((j, xi, eta))
                data[j].append((i, xi, eta))
            else:
                psi[:, i] += (G0[j, i] / gap) * Phi[:, j]
                psi[:, j] += (G0[i, j] / (lam[i] - lam[j])) * Phi[:, i]
    return data


# --------------------------------------------------------------------------
# Lanczos adjoint approximation (ref 394-523)
# --------------------------------------------------------------------------
def laa(Phib, B, factor, sigma, lam, V, Y, theta, indices, D0=None, b_ortho=False, mode="normal", cols=None):
    """ref 394-523.  ``cols`` (not in the reference): evaluate only those modes' columns -- the coefficient matrix D is
    built for all N modes as the reference does, the N-column product and factor solve of ref 519-521 are restricted
    to the listed columns (bench.py times a bounded sample of the modes on the CPU); the result keeps shape (n, N)."""
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
        # ref 492-500 dereferences an unassigned array; keep that observable.
        raise NameError("name 'D' is not defined")

    Yb = V.T @ Phib
    C = Y.T @ Yb  # C[i, j] = Y[:, i] . Yb[:, j]
    th_sel = theta[indices[:N]]
    D = np.zeros((m, N))
    if b_ortho:  # ref 501-508: only the unconverged Ritz directions
        rows = indices[N:]
        D[rows, :] = C[rows, :] / (th_sel[None, :] - theta[rows, None])
    else:  # ref 509-516: everything but the mode's own Ritz vector
        with np.errstate(divide="ignore", invalid="ignore"):
            D = C / (th_sel[None, :] - theta[:, None])
        D[indices[:N], np.arange(N)] = 0.0
    scale = 1.0 if mode == "normal" else sigma
    Cf = Y @ (scale * (D / (lam - sigma)))
    if cols is None:
        return -factor(B @ V @ Cf)
    out = np.zeros((n, N))
    out[:, cols] = -factor(B @ (V @ Cf[:, cols]))
    return out


# --------------------------------------------------------------------------
# reverse-mode sweep through the Lanczos recurrence (ref 526-696)
# --------------------------------------------------------------------------
def dl(Phib, B, factor, sigma, lam, Phi, indices, V, T, Y, theta, eig_atol=1e-5, mode="normal"):
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

    repeated = are_eigenvalues_repeated(lam, atol=eig_atol)
    data, G, BPhi = {}, None, None
    sel = indices[:N]
    if repeated:  # ref 607-617
        BPhi = B @ Phi
        G = -Phi.T @ Phib
        R = Phib + BPhi @ G
    else:
        R = Phib
    Vb = R @ Y[:, sel].T
    Yb = V.T @ R

    D = np.zeros((m, m))
    for i in range(m):  # ref 622-631
        for j in range(N):
            ii, jj = indices[i], indices[j]
            if ii == jj:
                continue
            if i < N and j < N and is_close(lam[i], lam[j], atol=eig_atol):
                continue
            D[ii, jj] = Y[:, ii].dot(Yb[:, j]) / (theta[jj] - theta[ii])
    Tb = Y @ (D @ Y.T)

    # ref 636-674
    t = B @ factor(B @ V[:, m - 1])
    Vb += np.outer(t, Tb[:m, m - 1])
    sb = B @ (V[:, :m] @ Tb[:, m - 1])
    u = factor(sb)
    Vb[:, m - 1] += B @ u
    for i in range(m - 2, -1, -1):
        lo = max(i - 1, 0)
        t = B @ (V[:, lo : i + 2] @ T[lo : i + 2, i])
        c0 = V[:, i + 1].dot(Vb[:, i + 1]) - T[i + 1, i] * Tb[i + 1, i]
        sb = (Vb[:, i + 1] - c0 * (B @ V[:, i + 1])) / T[i + 1, i]
        Vb[:, i - 1] -= T[i - 1, i] * sb  # i == 0 wraps to column -1 exactly as ref 657
        Vb[:, i] -= T[i, i] * sb
        hb = V[:, : i + 1].T @ sb - Tb[: i + 1, i]
        Vb[:, : i + 1] -= np.outer(t, hb)
        sb -= B @ (V[:, : i + 1] @ hb)
        Vb[:, i + 1] = u
        u = factor(sb)
        Vb[:, i] += B @ u
    Vb[:, 0] = u

    scale = 1.0 if mode == "normal" else sigma
    psi = -Vb @ (scale * Y[:, sel] / (lam - sigma))
    if repeated:  # ref 682-694
        psi = project(Phi, BPhi, psi)
        data = generate_adjoint_correction(lam, Phi, psi, G=G, eig_atol=eig_atol, mode=mode)
    return psi, data


# --------------------------------------------------------------------------
# shared pieces of the three iterative solvers
# --------------------------------------------------------------------------
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


def _default_factor(A, B, lam, sigma, mode):
    """ref 783-790 / 954-961 / 1160-1167"""
    if sigma is None:
        sigma = 0.9 * lam[0]
    P = A - sigma * B if mode == "normal" else B + sigma * A
    return SpLuOperator(P.tocsc()), sigma


def _adjoint_operator(A, B, x, lam_i, mode):
    """(A - lam B) x  or  (B + lam A) x"""
    if mode == "normal":
        return A @ x - lam_i * (B @ x)
    return B @ x + lam_i * (A @ x)


# --------------------------------------------------------------------------
# PCPG (ref 699-869)
# --------------------------------------------------------------------------
def pcpg(
    Phib, A, B, lam, Phi, mode="normal", psi=None, sigma=None, factor=None,
    rtol=1e-10, atol=1e-30, eig_atol=1e-5, maxiter=100, reset=25, callback=None,
):
    n, N = _check_iter_args(Phib, A, B, lam, Phi, psi, mode, check_lam=False)
    if factor is None:
        factor, sigma = _default_factor(A, B, lam, sigma, mode)
    _psi = psi if psi is not None else np.zeros((n, N), dtype=Phib.dtype)
    rnorm0 = np.sqrt(np.max(np.sum(Phib**2, axis=0)))
    BPhi = B @ Phi
    G = np.zeros((N, N))
    info = []
    for i in range(N):
        R = -Phib[:, i] - _adjoint_operator(A, B, _psi[:, i], lam[i], mode)
        G[:, i] = Phi.T @ R
        R -= BPhi @ G[:, i]
        P0 = np.zeros(n)
        zTr_prev = 1.0
        converged = False
        for k in range(maxiter):
            res = np.linalg.norm(R)
            if callback is not None:
                callback(res)
            if res < rtol * rnorm0 or res < atol:
                converged = True
                break
            Z = project(Phi, BPhi, factor(project(BPhi, Phi, R.copy())))
            zTr = Z.dot(R)
            if k % reset == 0:
                P = Z.copy()
            else:
                P = Z + (zTr / zTr_prev) * P0
            zTr_prev = zTr
            tA, tB = A @ P, B @ P
            q = tA - lam[i] * tB if mode == "normal" else tB + lam[i] * tA
            alpha = zTr / q.dot(P)
            _psi[:, i] += alpha * P
            R = R - alpha * q
            P0 = P
        info.append(converged)
    data = generate_adjoint_correction(lam, Phi, _psi, G=G, eig_atol=eig_atol, mode=mode)
    return _psi, data, info


# --------------------------------------------------------------------------
# PGMRES (ref 872-1040)
# --------------------------------------------------------------------------
def pgmres(
    Phib, A, B, lam, Phi, mode="normal", psi=None, sigma=None, factor=None,
    rtol=1e-10, atol=1e-30, eig_atol=1e-5, maxiter=50, callback=None,
):
    n, N = _check_iter_args(Phib, A, B, lam, Phi, psi, mode)
    if factor is None:
        factor, sigma = _default_factor(A, B, lam, sigma, mode)
    _psi = psi if psi is not None else np.zeros((n, N), dtype=Phib.dtype)
    rnorm0 = np.sqrt(np.max(np.sum(Phib**2, axis=0)))
    BPhi = B @ Phi
    G = np.zeros((N, N))
    W = np.zeros((n, maxiter + 1))
    Z = np.zeros((n, maxiter))
    H = np.zeros((maxiter + 1, maxiter))
    info = []
    for i in range(N):
        R = -Phib[:, i] - _adjoint_operator(A, B, _psi[:, i], lam[i], mode)
        G[:, i] = Phi.T @ R
        R -= BPhi @ G[:, i]
        beta = np.sqrt(R.dot(R))
        if callback is not None:
            callback(beta)
        if beta < rtol * rnorm0 or beta < atol:
            info.append(0)
            continue
        W[:, 0] = R / beta
        for j in range(maxiter):
            Z[:, j] = factor(project(BPhi, Phi, W[:, j].copy()))
            W[:, j + 1] = project(BPhi, Phi, _adjoint_operator(A, B, Z[:, j], lam[i], mode))
            for k in range(j + 1):
                H[k, j] = W[:, j + 1].dot(W[:, k])
                W[:, j + 1] -= H[k, j] * W[:, k]
            H[j + 1, j] = np.sqrt(W[:, j + 1].dot(W[:, j + 1]))
            W[:, j + 1] /= H[j + 1, j]
            rhs = np.zeros(j + 2)
            rhs[0] = beta
            Hj = H[: j + 2, : j + 1]
            y = np.linalg.lstsq(Hj, rhs, rcond=None)[0]
            res = np.linalg.norm(Hj.dot(y) - rhs)
            if callback is not None:
                callback(res)
            if res < rtol * rnorm0 or res < atol:
                _psi[:, i] += Z[:, : j + 1] @ y
                info.append(j)
                break
            elif j == maxiter - 1:
                _psi[:, i] += Z[:, : j + 1] @ y
                info.append(-1)
    data = generate_adjoint_correction(lam, Phi, _psi, G=G, eig_atol=eig_atol, mode=mode)
    return _psi, data, info


# --------------------------------------------------------------------------
# shift-invert block Krylov (ref 1043-1328)
# --------------------------------------------------------------------------
def solve_shifted_lstsq(alpha, H, r):
    """min || (I - alpha H) y - r ||  (ref 1043-1049)"""
    H0 = np.eye(H.shape[0], H.shape[1]) - alpha * H
    y = np.linalg.lstsq(H0, r, rcond=None)[0]
    return y, np.linalg.norm(H0 @ y - r)


def sibk(
    Phib, A, B, lam, Phi, mode="normal", psi=None, sigma=None, factor=None,
    rtol=1e-10, atol=1e-30, eig_atol=1e-5, maxiter=50, bs_target=1,
    update_guess=False, callback=None, nrestart=2, modes=None,
):
    """ref 1052-1328.  ``modes`` (not in the reference; bs_target=1, update_guess=False only): solve only the listed
    modes -- projector, G and the correction still use all N eigenvectors, the other columns of psi are left as given
    (bench.py times a bounded sample of the modes on the CPU)."""
    n, N = _check_iter_args(Phib, A, B, lam, Phi, psi, mode)
    if modes is not None and (bs_target != 1 or update_guess):
        raise ValueError("modes= needs bs_target=1 and update_guess=False")
    if factor is None:
        factor, sigma = _default_factor(A, B, lam, sigma, mode)
    rnorm0 = np.sqrt(np.max(np.sum(Phib**2, axis=0)))
    BPhi = B @ Phi
    W = np.zeros((n, maxiter + bs_target))
    Z = np.zeros((n, maxiter))
    G = -Phi.T @ Phib
    _psi = psi if psi is not None else np.zeros((n, N), dtype=Phib.dtype)
    K = B if mode == "normal" else A  # Krylov operator is P K factor (ref 1249-1252)
    sgn = 1.0 if mode == "normal" else -1.0  # ref 1265-1268

    if modes is None:
        if mode == "normal":
            R = -Phib - (A @ _psi - (B @ _psi) * lam)
        else:
            R = -Phib - (B @ _psi + (A @ _psi) * lam)
        R = project(BPhi, Phi, R)
    else:
        sel = np.asarray(modes)
        R = np.zeros((n, N))
        if mode == "normal":
            Rs = -Phib[:, sel] - (A @ _psi[:, sel] - (B @ _psi[:, sel]) * lam[sel])
        else:
            Rs = -Phib[:, sel] - (B @ _psi[:, sel] + (A @ _psi[:, sel]) * lam[sel])
        R[:, sel] = project(BPhi, Phi, Rs)

    info = []
    i = 0
    restart = 0
    while i < N:
        if modes is not None and i not in modes:
            i += 1
            continue
        r = np.zeros((maxiter + bs_target, bs_target))
        bs = 0
        while i + bs < N and bs < bs_target:
            k = i + bs
            if update_guess:  # ref 1205-1215
                _psi[:, k] = project(Phi, BPhi, _psi[:, k])
                W[:, bs] = -Phib[:, k] - _adjoint_operator(A, B, _psi[:, k], lam[k], mode)
                W[:, bs] = project(BPhi, Phi, W[:, bs])
            else:
                W[:, bs] = R[:, k]
            beta0 = np.sqrt(W[:, bs].dot(W[:, bs]))
            if callback is not None:
                callback(beta0)
            if beta0 < rtol * rnorm0 or beta0 < atol:
                info.append(0)
                break
            for j in range(bs):
                r[j, bs] = W[:, bs].dot(W[:, j])
                W[:, bs] -= r[j, bs] * W[:, j]
            W[:, bs] = project(BPhi, Phi, W[:, bs])
            r[bs, bs] = np.sqrt(W[:, bs].dot(W[:, bs]))
            W[:, bs] /= r[bs, bs]
            bs += 1
        if bs == 0:
            i += 1
            continue

        H = np.zeros((maxiter + bs, maxiter))
        y = np.zeros((maxiter, bs))
        for j in range(bs, maxiter + bs):
            kp = j - bs
            Z[:, kp] = factor(W[:, kp])
            W[:, j] = project(BPhi, Phi, K @ Z[:, kp])
            for k in range(j - 1, -1, -1):
                H[k, kp] = W[:, j].dot(W[:, k])
                W[:, j] -= H[k, kp] * W[:, k]
            W[:, j] = project(BPhi, Phi, W[:, j])
            H[j, kp] = np.sqrt(W[:, j].dot(W[:, j]))
            W[:, j] /= H[j, kp]

            res = 0.0
            H0 = H[: j + 1, : j + 1 - bs]
            for k in range(bs):
                alpha = sgn * (lam[i + k] - sigma)
                y[: kp + 1, k], res0 = solve_shifted_lstsq(alpha, H0, r[: j + 1, k])
                res = max(res, res0)
            if callback is not None:
                callback(res)

            if res < rtol * rnorm0 or res < atol:
                info.append(j)
                _psi[:, i : i + bs] += Z[:, :j] @ y[:j, :]
                if update_guess and i + bs < N:  # ref 1279-1305
                    rest = slice(i + bs, N)
                    r0 = W[:, : j + 1].T @ R[:, rest]
                    y0 = np.zeros((j + 1 - bs, N - (i + bs)))
                    t0 = np.zeros((j + 1, N - (i + bs)))
                    for k in range(i + bs, N):
                        alpha = sgn * (lam[k] - sigma)
                        yk, _ = solve_shifted_lstsq(alpha, H0, r0[:, k - (i + bs)])
                        y0[:, k - (i + bs)] = yk
                        t0[:, k - (i + bs)] = -alpha * H0 @ yk
                        t0[:-bs, k - (i + bs)] += yk
                    _psi[:, rest] += Z[:, : j + 1 - bs] @ y0
                    R[:, rest] -= W[:, : j + 1] @ t0
                i += bs
                restart = 0
                break
            elif j == maxiter + bs - 1:
                _psi[:, i : i + bs] += Z[:, :j] @ y[:j, :]
                if restart >= nrestart:
                    restart = 0
                    i += bs
                    break
                restart += 1
    data = generate_adjoint_correction(lam, Phi, _psi, G=G, eig_atol=eig_atol, mode=mode)
    return _psi, data, info


# --------------------------------------------------------------------------
# eigen-solvers
# --------------------------------------------------------------------------
def complex_step_eigh(T):
    """eigh; for a complex T the imaginary part is a forward derivative of the real symmetric problem (ref 1387-1414):
    d(lam_i) = q_i^T dT q_i, d(q_i) = sum_{j != i, lam_j != lam_i} q_j (q_j^T dT q_i) / (lam_i - lam_j)."""
    if not np.issubdtype(T.dtype, np.complexfloating):
        return np.linalg.eigh(T)
    lam, Q = np.linalg.eigh(T.real)
    D = Q.T @ T.imag @ Q
    w = lam + 1j * np.diag(D)
    gap = lam[None, :] - lam[:, None]  # gap[i, j] = lam[j] - lam[i]
    with np.errstate(divide="ignore", invalid="ignore"):
        C = np.where(gap == 0.0, 0.0, D / gap)
    return w, Q + 1j * (Q @ C)


def ritz_to_eigs(theta, sigma, mode):
    """Undo the spectral transformation and give the sort order (ref 1432-1437, 1960-1965)."""
    if mode == "normal":
        lam = 1.0 / theta + sigma
        return lam, np.argsort(lam)
    lam = sigma * theta / (theta - 1.0)
    return lam, np.argsort(-1.0 / lam)


class _AdjointMixin:
    """solve_adjoint / residual / total-derivative plumbing (ref 1652-1870, 1988-2207)."""

    def _lam_N(self):
        raise NotImplementedError

    def solve_adjoint(self, Phib, method="sibk", psi=None, rtol=1e-10, atol=1e-30, lanczos_guess=True, **kwargs):
        n = self.A.shape[1]
        lam, V = self._lam_N(), self._basis()
        N = len(lam)
        if method not in METHODS:
            raise ValueError(f"Unknown method {method!r}")
        if psi is not None and psi.shape != (n, N):
            raise ValueError(f"Initial guess must have the shape ({n},{N})")
        if method == "dl":
            self._warn_dl()
            lanczos_guess = False
        data = {}
        if lanczos_guess or method == "laa":
            psi = laa(Phib, self.B, self.factor, self.sigma, lam, V, self.Y, self.theta,
                      self.indices, b_ortho=True, mode=self.mode)
        else:
            psi = np.zeros((n, N))
        common = dict(mode=self.mode, psi=psi, factor=self.factor, rtol=rtol, atol=atol, eig_atol=self.eig_atol)
        if method == "pcpg":
            psi, data, self.last_info = pcpg(Phib, self.A, self.B, lam, self.Phi, **common, **kwargs)
        elif method == "pgmres":
            psi, data, self.last_info = pgmres(Phib, self.A, self.B, lam, self.Phi, **common, **kwargs)
        elif method == "sibk":
            psi, data, self.last_info = sibk(Phib, self.A, self.B, lam, self.Phi, sigma=self.sigma, **common, **kwargs)
        elif method == "laa":
            data = generate_adjoint_correction(lam, self.Phi, psi, Phib=Phib, eig_atol=self.eig_atol, mode=self.mode)
        elif method == "dl":
            psi, data = dl(Phib, self.B, self.factor, self.sigma, lam, self.Phi, self.indices, V,
                           self.T, self.Y, self.theta, self.eig_atol, mode=self.mode)
        return psi, data

    def eval_adjoint_residual_norm(self, Phib, psi, b_ortho=False):
        return eval_adjoint_residual_norm(self.A, self.B, self._lam_N(), self.Phi, Phib, psi,
                                          mode=self.mode, b_ortho=b_ortho)

    def add_total_derivative(self, lamb, Phib, psi, dAdx, dBdx, dfdx, adj_corr_data={}, deriv_type="vector"):
        return add_eig_total_derivative(self._lam_N(), self.Phi, lamb, Phib, psi, dAdx, dBdx, dfdx,
                                        adj_corr_data=adj_corr_data, mode=self.mode, deriv_type=deriv_type)


class BasicLanczos(_AdjointMixin):
    """Un-restarted shift-invert Lanczos with B-orthogonalisation (ref 1331-1650).  Complex matrices follow the
    reference's complex-step semantics: products are not conjugated and the reduced problem treats imaginary parts
    as forward derivatives (``_eigh``, ref 1387-1414)."""

    def __init__(self, N=10, m=60, tol=1e-14, Ntarget=None, eig_atol=1e-5, mode="normal", ortho_type="full"):
        self.N, self.m_max, self.tol, self.Ntarget = N, m, tol, Ntarget
        self.eig_atol, self.mode, self.ortho_type = eig_atol, mode, ortho_type
        if Ntarget is not None and not isinstance(Ntarget, int):
            raise ValueError("Ntarget must be an integer or None")
        if ortho_type not in ("full", "selective"):
            raise ValueError(f"Unknown ortho_type {ortho_type!r}")
        _check_mode(mode)

    def _lam_N(self):
        return self.lam0

    def _basis(self):
        return self.V[:, : self.m]

    def _warn_dl(self):
        pass

    def _reduced(self, m):
        """eigh of the leading m x m tridiagonal + sort (ref 1416-1439)."""
        T = np.diag(self.alpha[:m]) + np.diag(self.beta[: m - 1], 1) + np.diag(self.beta[: m - 1], -1)
        theta, Y = complex_step_eigh(T)
        lam, indices = ritz_to_eigs(theta, self.sigma, self.mode)
        return theta, Y, T, lam, indices

    @staticmethod
    def _converged(beta_last, Yrow, N, tol):
        """Leading run of Ritz pairs with |beta y_last| < tol (ref 1441-1451)."""
        count = 0
        for e in np.abs(beta_last * Yrow):
            if e < tol:
                count += 1
            else:
                break
        return count >= N

    def solve(self, A, B, factor, sigma):
        n = A.shape[1]
        if A.shape != (n, n):
            raise ValueError(f"A must have dimensions ({n},{n})")
        if B.shape != (n, n):
            raise ValueError(f"B must have dimensions ({n},{n})")
        if factor.shape != (n, n):
            raise ValueError(f"Factorized operator must have dimensions ({n},{n})")
        self.factor = aslinearoperator(factor)
        self.B = aslinearoperator(B)
        self.A = aslinearoperator(A)
        self.sigma = sigma
        Bm, fac = self.B, self.factor
        inner = lambda x, y: y.dot(Bm @ x)  # ref 1503: one SpMV per inner product

        mm = self.m_max
        dtype = A.dtype  # ref 1483: complex matrices switch every array to complex
        self.alpha = np.zeros(mm, dtype=dtype)
        self.beta = np.zeros(mm, dtype=dtype)
        self.V = np.zeros((n, mm + 1), dtype=dtype)
        V = self.V
        V[:, 0] = np.random.default_rng(12345).uniform(size=n, low=-1.0, high=1.0)
        V[:, 0] /= np.sqrt(inner(V[:, 0], V[:, 0]))

        Nchk = self.N if self.Ntarget is None else self.Ntarget
        self.m = mm
        S = None
        for i in range(1, mm + 1):
            V[:, i] = fac(Bm @ V[:, i - 1])
            if i > 1:
                V[:, i] -= self.beta[i - 2] * V[:, i - 2]
            jlo = -1 if self.ortho_type == "full" else max(-1, i - 3)
            for j in range(i - 1, jlo, -1):
                h = inner(V[:, j], V[:, i])
                V[:, i] -= h * V[:, j]
                if j == i - 1:
                    self.alpha[i - 1] = h
            if self.ortho_type == "selective" and S is not None:  # ref 1571-1574
                for j in range(S.shape[1]):
                    h = inner(S[:, j], V[:, i])
                    V[:, i] -= h * S[:, j]
            self.beta[i - 1] = np.sqrt(inner(V[:, i], V[:, i]))
            V[:, i] /= self.beta[i - 1]
            if i >= 2:
                theta, Y, T, lam, indices = self._reduced(i)
                Y0 = Y[:, indices]
                if self._converged(self.beta[i - 1], Y0[i - 1, :], Nchk, self.tol):
                    self.m = i
                    break
                if self.ortho_type == "selective":  # ref 1596-1605
                    errs = np.abs(self.beta[i - 1] * Y0[i - 1, :])
                    conv = [j for j in range(i) if errs[j] < np.sqrt(self.tol)]
                    S = V[:, :i] @ Y0[:, conv]

        self.theta, self.Y, self.T, self.lam, self.indices = self._reduced(self.m)
        if self.Ntarget is not None:  # ref 1615-1625
            self.N = self.Ntarget
            while self.N < self.m and is_close(
                self.lam[self.indices[self.N - 1]].real, self.lam[self.indices[self.N]].real, self.eig_atol
            ):
                self.N += 1
        elif is_close(self.lam[self.indices[self.N - 1]].real, self.lam[self.indices[self.N]].real, self.eig_atol):
            warnings.warn(f"BasicLanczos: Ritz values {self.N} and {self.N+1} are numerically repeated.")
        sel = self.indices[: self.N]
        self.lam0 = self.lam[sel]
        self.Y0 = self.Y[:, sel]
        # ref 1640-1645: note beta[-1] is the LAST slot of the m_max-long array
        self.eig_res = np.abs(self.beta[-1] * self.Y0[-1, :])
        self.fail = bool(np.any(self.eig_res > self.tol))
        self.Phi = V[:, : self.m].dot(self.Y0)
        return self.lam0, self.Phi


def arpack_shift_invert(B_op, factor_op, n, k, ncv, sigma, mode, tol=0.0, A0_op=None):
    """
    ARPACK dsaupd/dseupd in shift-invert (mode 3) or buckling (mode 4) form with
    OP = factor o B and the B inner product, also returning the final m-step
    Lanczos tridiagonal and basis (arpack.py:24-101 reads them out of ARPACK's
    ``workl[0:2 ncv]`` and ``v`` work arrays; arpack.py:381-403 for the modes).
    """
    from scipy.sparse.linalg._eigen.arpack.arpack import (
        ArpackError, _SymmetricArpackParams,
    )

    if mode == "normal":
        p = _SymmetricArpackParams(n, k, "d", None, 3, B_op.matvec, factor_op.matvec, sigma, ncv, None, None, "LM", tol)
    else:
        p = _SymmetricArpackParams(n, k, "d", A0_op.matvec, 4, None, factor_op.matvec, sigma, ncv, None, None, "LM", tol)
    while not p.converged:
        p.iterate()
    h = p.workl[: 2 * ncv].copy()
    V = p.v.copy().reshape((-1, ncv))
    d, z, ierr = p._arpack_extract(
        True, "A", np.zeros(ncv, "int"), p.sigma, p.bmat, p.which, p.k, p.tol, p.resid, p.v,
        p.iparam[0:7], p.ipntr, p.workd[0 : 2 * n], p.workl, 0,
    )
    if ierr != 0:
        raise ArpackError(ierr, infodict=p.extract_infodict)
    k_ok = p.iparam[4]
    T = np.diag(h[ncv:]) + np.diag(h[1:ncv], 1) + np.diag(h[1:ncv], -1)
    return d[:k_ok], z[:, :k_ok], T, V


class IRAM(_AdjointMixin):
    """ARPACK-backed solver exposing (V, T) for the Lanczos adjoint approximation (ref 1873-1986)."""

    def __init__(self, N=10, m=None, eig_atol=1e-5, tol=0.0, mode="normal"):
        self.N = N
        self.m = max(20, 2 * N + 1) if m is None else max(20, 2 * N + 1, m)
        self.tol, self.eig_atol, self.mode = tol, eig_atol, mode
        _check_mode(mode)

    def _lam_N(self):
        return self.lam

    def _basis(self):
        return self.V

    def _warn_dl(self):
        warnings.warn('Adjoint method "dl" is not recommended for the ARPACK IRAM eigenvalue sovler.')

    def solve(self, A, B, factor, sigma):
        n = A.shape[1]
        if A.shape != (n, n):
            raise ValueError(f"A must have dimensions ({n},{n})")
        if B.shape != (n, n):
            raise ValueError(f"B must have dimensions ({n},{n})")
        if factor.shape != (n, n):
            raise ValueError(f"Factorized operator must have dimensions ({n},{n})")
        self.factor = aslinearoperator(factor)
        self.B = aslinearoperator(B)
        self.A = aslinearoperator(A)
        self.sigma = sigma
        # buckling: ARPACK mode 4 is driven with "A" = B (ref 1941-1942)
        self.lam, self.Phi, self.T, self.V = arpack_shift_invert(
            self.B, self.factor, n, self.N, self.m, sigma, self.mode, tol=self.tol, A0_op=self.B
        )
        self.theta, self.Y = np.linalg.eigh(self.T)
        eigs, self.indices = ritz_to_eigs(self.theta, sigma, self.mode)
        if is_close(eigs[self.indices[self.N - 1]], eigs[self.indices[self.N]], self.eig_atol):
            warnings.warn(f"IRAM: Ritz values {self.N} and {self.N+1} are numerically repeated.")
        for i in range(self.N):  # MAC sign alignment (ref 1976-1984)
            q = self.V @ self.Y[:, self.indices[i]]
            if self.Phi[:, i].dot(q) < 0.0:
                self.Y[:, self.indices[i]] *= -1.0
        return self.lam, self.Phi
