"""
CPU prototype (verdict r3 item 2): short recurrences for the adjoint systems of an SPD shift.

With F = factor (SPD for a shift below the spectrum) the Krylov operator of sibk, OP = P K F (reference
eigenvector_derivatives.py:1246-1252), is self-adjoint in the F inner product <u, v>_F = u^T F v, and
C_i = I - alpha_i OP is positive definite there once every pair with lam_j <= lam_i is deflated.  This script compares,
mode by mode on a buckling column with 32 wanted + 32 extra deflated pairs,

  (a) the reference's Arnoldi / least-squares form (Euclidean Gram-Schmidt against the whole history, 1254-1270),
  (b) conjugate gradients in the F inner product  (three vectors r, p, F p; psi += a F p; no history),
  (c) the Lanczos / minimal-residual form in the F inner product (residual minimised in the F norm),

counting operator applications until the TRUE Euclidean residual meets the reference's stopping rule (1275) and the
distance of psi from a direct solve.  Run: python tools/short_recurrence_probe.py [nx ny]
"""
import sys
import os

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from eigd_amd.problems import BucklingColumn  # noqa: E402  (host-side generator only)


def build(nx, ny, N, NX):
    col = BucklingColumn(nx, ny, Lx=1.0, Ly=2.0)
    K = col.stiffness().tocsc()
    f = col.f[col.reduced]
    u = spla.splu(K).solve(f)
    G = col.geometric_stiffness(col.full_vector(u)).tocsc()
    # BLF_1 by a few shift-invert steps at sigma = 0:  (K + lam G) phi = 0
    lu0 = spla.splu(K)
    op = spla.LinearOperator(K.shape, matvec=lambda x: -lu0.solve(G @ x))
    mu = spla.eigsh(op, k=1, which="LA", tol=1e-6)[0][0]       # largest mu = 1 / lam_1 (lam_1 > 0)
    lam1 = 1.0 / mu
    sigma = 0.7 * lam1
    F = spla.splu((K + sigma * G).tocsc())
    # (K + lam G) phi = 0  <=>  -G phi = mu K phi, mu = 1 / lam; shift-invert at mu = 1 / sigma with the same factor:
    # (-G - K / sigma)^-1 K = -sigma F K
    n = K.shape[0]
    opinv = spla.LinearOperator((n, n), matvec=lambda x: -sigma * F.solve(x))
    mu, Phi = spla.eigsh((-G).tocsc(), k=N + NX, M=K, sigma=1.0 / sigma, which="LM", OPinv=opinv, tol=0.0,
                         ncv=2 * (N + NX) + 8)
    order = np.argsort(-mu)
    lam, Phi = 1.0 / mu[order], Phi[:, order]        # K-orthonormal (eigsh normalises in the M inner product)
    return K, G, F, sigma, lam, Phi


def main():
    nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (60, 120)
    N, NX = 32, 32
    K, G, F, sigma, lam, PhiD = build(nx, ny, N, NX)
    n = K.shape[0]
    print(f"n = {n}, sigma = {sigma:.4f}, lam[0..3] = {lam[:4]}, lam[31] = {lam[31]:.4f}, lam[63] = {lam[63]:.4f}")
    res_eig = np.linalg.norm(K @ PhiD + (G @ PhiD) * lam, axis=0) / np.linalg.norm(K @ PhiD, axis=0)
    print("eigen-residuals: max %.2e" % res_eig.max())
    BPhiD = K @ PhiD                            # buckling mode: B = K, A = G
    rng = np.random.default_rng(1)
    Phib = rng.uniform(0.0, 1.0, size=(n, N))
    rnorm0 = np.sqrt(np.max(np.sum(Phib**2, axis=0)))
    tol = 1e-10 * rnorm0

    def P(x):                                   # residual-side projector (reference 26-30 with (B Phi, Phi))
        return x - BPhiD @ (PhiD.T @ x)

    def OP(z):                                  # P K F acting on what F already produced: P A z  (buckling: A = G)
        return P(G @ z)

    tot = {"arnoldi": 0, "cg": 0, "minres": 0}
    worst = {"arnoldi": 0.0, "cg": 0.0, "minres": 0.0}
    rows = []
    three = []
    for i in range(N):
        alpha = -(lam[i] - sigma)               # reference 1265-1268 (buckling)
        b = P(-Phib[:, i])                      # psi_0 = 0: R = -Phib, projected (1189-1193)
        # direct reference solution in the deflated space: psi = F w,  (I - alpha OP) w = b
        # ---- (a) Arnoldi + least squares, Euclidean ------------------------------------------------------------
        W = [b / np.linalg.norm(b)]
        Z = []
        H = np.zeros((80, 79))
        beta0 = np.linalg.norm(b)
        ka = None
        for j in range(1, 79):
            Z.append(F.solve(W[j - 1]))
            v = OP(Z[-1])
            for q in range(j - 1, -1, -1):
                H[q, j - 1] = v @ W[q]
                v -= H[q, j - 1] * W[q]
            v = P(v)
            H[j, j - 1] = np.linalg.norm(v)
            W.append(v / H[j, j - 1])
            H0 = np.eye(j + 1, j) - alpha * H[: j + 1, :j]
            rhs = np.zeros(j + 1)
            rhs[0] = beta0
            y = np.linalg.lstsq(H0, rhs, rcond=None)[0]
            if np.linalg.norm(H0 @ y - rhs) < tol:
                ka = j
                psi_a = np.array(Z).T @ y
                break
        # ---- (b) CG in the F inner product ----------------------------------------------------------------------
        r = b.copy()
        zr = F.solve(r)
        p, zp = r.copy(), zr.copy()
        rho = r @ zr
        psi_c = np.zeros(n)
        kc = None
        for j in range(1, 79):
            Cp = p - alpha * OP(zp)
            a = rho / (zp @ Cp)
            psi_c += a * zp
            r = P(r - a * Cp)
            if np.linalg.norm(r) < tol:
                kc = j
                break
            zr = F.solve(r)
            rho_new = r @ zr
            bt = rho_new / rho
            rho = rho_new
            p = r + bt * p
            zp = zr + bt * zp
        # CG applies F once per step plus once at the start: j steps = j + 1 sweeps, of which the last is not needed
        # ---- (d) three-term form of the same CG iterates (Rutishauser): r and psi by three-term recurrences, no
        # direction vectors -- 11 instead of 18 streaming passes over the blocks per step on the device
        r3, r3o = b.copy(), np.zeros(n)
        ps3, ps3o = np.zeros(n), np.zeros(n)
        rho_o = gam_o = rr_o = None
        k3 = None
        for j in range(1, 79):
            z3 = F.solve(r3)
            y3 = OP(z3)
            rr = r3 @ z3
            gam = rr / (rr - alpha * (z3 @ y3))
            rho3 = 1.0 if j == 1 else 1.0 / (1.0 - (gam / gam_o) * (rr / rr_o) / rho_o)
            rn = P(rho3 * (r3 - gam * (r3 - alpha * y3)) + (1.0 - rho3) * r3o)
            pn = rho3 * (ps3 + gam * z3) + (1.0 - rho3) * ps3o
            r3o, r3, ps3o, ps3 = r3, rn, ps3, pn
            rho_o, gam_o, rr_o = rho3, gam, rr
            if np.linalg.norm(r3) < tol:
                k3 = j
                break
        psi_3 = ps3
        # ---- (c) Lanczos in the F inner product + minimal residual (F norm) via the tridiagonal -----------------
        z0 = F.solve(b)
        bF = np.sqrt(b @ z0)
        Wl, Zl = [b / bF], [z0 / bF]
        al, be = [], []
        km = None
        for j in range(1, 79):
            v = OP(Zl[-1])
            a_ = Zl[-1] @ v
            v = v - a_ * Wl[-1] - (be[-1] * Wl[-2] if be else 0.0)
            v = P(v)
            zv = F.solve(v)
            b_ = np.sqrt(max(v @ zv, 0.0))
            al.append(a_)
            be.append(b_)
            Wl.append(v / b_)
            Zl.append(zv / b_)
            T = np.zeros((j + 1, j))
            T[np.arange(j), np.arange(j)] = al
            T[np.arange(1, j + 1), np.arange(j)] = be
            T[np.arange(j - 1), np.arange(1, j)] = be[:-1]
            H0 = np.eye(j + 1, j) - alpha * T
            rhs = np.zeros(j + 1)
            rhs[0] = bF
            y = np.linalg.lstsq(H0, rhs, rcond=None)[0]
            rvec = np.array(Wl).T @ (rhs - H0 @ y)          # true Euclidean residual (prototype: from the basis)
            if np.linalg.norm(rvec) < tol:
                km = j
                psi_m = np.array(Zl[:j]).T @ y
                break
        # reference: direct sparse solve of the bordered system is overkill here; compare (b), (c) against (a)
        ea = np.linalg.norm(psi_a)
        dc = np.linalg.norm(psi_c - psi_a) / ea
        dm = np.linalg.norm(psi_m - psi_a) / ea
        # true residuals of the ORIGINAL system (K + lam G) psi = b in the deflated space
        tr = lambda ps: np.linalg.norm(P(b - (K @ ps + lam[i] * (G @ ps)))) / rnorm0
        d3 = np.linalg.norm(psi_3 - psi_a) / ea
        rows.append((i, ka, kc, km, dc, dm, tr(psi_a), tr(psi_c), tr(psi_m)))
        three.append((k3, d3, tr(psi_3)))
        tot["arnoldi"] += ka
        tot["cg"] += kc
        tot["minres"] += km
        print("mode %2d  steps arnoldi %2d  cg %2d  lanczos-mr %2d   |psi_cg - psi_a|/|psi_a| %.1e  mr %.1e   true res a %.1e cg %.1e mr %.1e"
              % rows[-1])
    print("total operator applications (32 modes): arnoldi %d, cg %d (+%.1f %%), lanczos-mr %d (+%.1f %%)"
          % (tot["arnoldi"], tot["cg"], 100.0 * (tot["cg"] / tot["arnoldi"] - 1), tot["minres"],
             100.0 * (tot["minres"] / tot["arnoldi"] - 1)))
    print("longest chain: arnoldi %d, cg %d, lanczos-mr %d"
          % (max(r[1] for r in rows), max(r[2] for r in rows), max(r[3] for r in rows)))
    print("three-term form: total steps %d, longest chain %d, max distance of psi from the Arnoldi result %.1e, max true residual %.1e"
          % (sum(t[0] for t in three), max(t[0] for t in three), max(t[1] for t in three), max(t[2] for t in three)))
    print("max distance of psi from the Arnoldi result: cg %.1e, lanczos-mr %.1e" % (max(r[4] for r in rows), max(r[5] for r in rows)))


if __name__ == "__main__":
    main()
