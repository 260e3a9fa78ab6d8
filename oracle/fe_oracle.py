"""
CPU oracle for the harness pieces either side of the eigd calls: the Q4 element derivative
callbacks, the fundamental-path adjoint of the buckling harness, the node filter and the aggregate functionals.

THIS FILE IS TEST INFRASTRUCTURE (see oracle/eigd_oracle.py): a numpy/scipy restatement of the reference's
``examples/fe_utils.py``, ``examples/node_filter.py`` and of the derivative routines of ``examples/buckling.py``,
``examples/natural_frequency.py`` and ``examples/thermal.py``, in the Gauss-point form the reference uses (the product
works with assembled element matrices on the device instead).  Imported only by ``tests/``.

Parity pin: ``tests/test_oracle_golden.py`` checks every function against the fixtures captured from the reference's
own harness runs (``rhoEb``, ``dfdu0``, ``rhob``, ``xb``, ``ans``, ``ks``, ``ks_grad``, filter outputs; tools/make_golden.py).
"""

import numpy as np
from scipy import sparse, spatial
from scipy.sparse import linalg

GAUSS = (-1.0 / np.sqrt(3.0), 1.0 / np.sqrt(3.0))


def shape_functions(xi, eta):
    """fe_utils.py:4-16"""
    N = 0.25 * np.array([(1 - xi) * (1 - eta), (1 + xi) * (1 - eta), (1 + xi) * (1 + eta), (1 - xi) * (1 + eta)])
    Nxi = 0.25 * np.array([-(1 - eta), (1 - eta), (1 + eta), -(1 + eta)])
    Neta = 0.25 * np.array([-(1 - xi), -(1 + xi), (1 + xi), (1 - xi)])
    return N, Nxi, Neta


def _physical_derivs(xi, eta, xe, ye):
    """N, dN/dx, dN/dy per element and detJ at one quadrature point (fe_utils.py:19-49)"""
    N, Nxi, Neta = shape_functions(xi, eta)
    J00, J10, J01, J11 = xe @ Nxi, ye @ Nxi, xe @ Neta, ye @ Neta
    detJ = J00 * J11 - J01 * J10
    i00, i01, i10, i11 = J11 / detJ, -J01 / detJ, -J10 / detJ, J00 / detJ
    Nx = np.outer(i00, Nxi) + np.outer(i10, Neta)
    Ny = np.outer(i01, Nxi) + np.outer(i11, Neta)
    return N, Nx, Ny, detJ


class Q4Tables:
    """
    Gauss-point tables of a Q4 mesh: Be (nelem, 3, 8, 4) strain-displacement, He (nelem, 2, 8, 4) interpolation,
    Te (nelem, 3, 4, 4, 4) stress-stiffening, thermal Bt (nelem, 2, 4, 4) / Ht (nelem, 4, 4), detJ (nelem, 4)
    (fe_utils.py:19-55, 58-97, 123-156; the order of the four points differs between the harnesses and never matters:
    every use sums over them).
    """

    def __init__(self, conn, X):
        self.conn = np.asarray(conn)
        self.X = np.asarray(X)
        ne = self.conn.shape[0]
        self.nelems = ne
        self.nnodes = int(self.conn.max()) + 1
        xe, ye = self.X[self.conn, 0], self.X[self.conn, 1]
        self.Be = np.zeros((ne, 3, 8, 4))
        self.He = np.zeros((ne, 2, 8, 4))
        self.Te = np.zeros((ne, 3, 4, 4, 4))
        self.Bt = np.zeros((ne, 2, 4, 4))
        self.Ht = np.zeros((ne, 4, 4))
        self.detJ = np.zeros((ne, 4))
        q = 0
        for eta in GAUSS:
            for xi in GAUSS:
                N, Nx, Ny, detJ = _physical_derivs(xi, eta, xe, ye)
                self.Be[:, 0, ::2, q] = Nx
                self.Be[:, 1, 1::2, q] = Ny
                self.Be[:, 2, ::2, q] = Ny
                self.Be[:, 2, 1::2, q] = Nx
                self.He[:, 0, ::2, q] = N
                self.He[:, 1, 1::2, q] = N
                self.Te[:, 0, :, :, q] = np.einsum("ni,nj->nij", Nx, Nx)
                self.Te[:, 1, :, :, q] = np.einsum("ni,nj->nij", Ny, Ny)
                self.Te[:, 2, :, :, q] = np.einsum("ni,nj->nij", Nx, Ny) + np.einsum("ni,nj->nij", Ny, Nx)
                self.Bt[:, 0, :, q] = Nx
                self.Bt[:, 1, :, q] = Ny
                self.Ht[:, :, q] = N
                self.detJ[:, q] = detJ
                q += 1

    def element_vectors2(self, u):
        """(nelem, 8[, k]) element dof values of a 2-dof/node vector or block"""
        u = np.asarray(u)
        ue = np.zeros((self.nelems, 8) + u.shape[1:], dtype=u.dtype)
        ue[:, ::2, ...] = u[2 * self.conn, ...]
        ue[:, 1::2, ...] = u[2 * self.conn + 1, ...]
        return ue


def plane_stress_C0(E=1.0, nu=0.3):
    """buckling.py:88-92"""
    return (E / (1.0 - nu**2)) * np.array([[1.0, nu, 0.0], [nu, 1.0, 0.0], [0.0, 0.0, 0.5 * (1.0 - nu)]])


# ----------------------------------------------------------------------------------------------------------------
# element derivative callbacks (per element, BEFORE the nodal averaging / filter of the harness)
# ----------------------------------------------------------------------------------------------------------------
def stiffness_deriv(tab, C0, rhoE, p, psi, u):
    """d(psi^T K u)/d rhoE, SIMP (buckling.py:178-205; natural_frequency.py:162-203); vectors or n x k blocks"""
    psie, ue = tab.element_vectors2(psi), tab.element_vectors2(u)
    out = np.zeros(tab.nelems)
    for q in range(4):
        Be, detJ = tab.Be[:, :, :, q], tab.detJ[:, q]
        if psie.ndim == 2:
            se = np.einsum("nij,nj->ni", Be, psie)
            te = np.einsum("nij,nj->ni", Be, ue)
            out += detJ * np.einsum("ij,nj,ni->n", C0, se, te)
        else:
            se, te = Be @ psie, Be @ ue
            out += detJ * np.einsum("ij,njk,nik->n", C0, se, te)
    return out * p * rhoE ** (p - 1.0)


def mass_deriv(tab, rhoE, density, u, v):
    """d(u^T M v)/d rhoE, linear density interpolation (natural_frequency.py:238-284)"""
    ue, ve = tab.element_vectors2(u), tab.element_vectors2(v)
    out = np.zeros(tab.nelems)
    for q in range(4):
        He, detJ = tab.He[:, :, :, q], tab.detJ[:, q]
        if ue.ndim == 2:
            out += detJ * np.einsum("ni,ni->n", np.einsum("nij,nj->ni", He, ue), np.einsum("nij,nj->ni", He, ve))
        else:
            out += detJ * np.einsum("nik,nik->n", He @ ue, He @ ve)
    return out * density


def thermal_stiffness_deriv(tab, rhoE, p, kappa, beta, psi, u):
    """thermal.py:150-190"""
    ue, psie = np.asarray(u)[tab.conn, ...], np.asarray(psi)[tab.conn, ...]
    dfdk = np.zeros(tab.nelems)
    for q in range(4):
        Be, detJ = tab.Bt[:, :, :, q], tab.detJ[:, q]
        if ue.ndim == 2:
            dfdk += detJ * np.einsum("ni,ni->n", np.einsum("nij,nj->ni", Be, psie), np.einsum("nij,nj->ni", Be, ue))
        else:
            dfdk += detJ * np.einsum("nik,nik->n", Be @ psie, Be @ ue)
    return (1.0 - beta) * kappa * dfdk * p * rhoE ** (p - 1.0)


def thermal_mass_deriv(tab, heat_capacity, density, beta, u, v):
    """thermal.py:216-246"""
    ue, ve = np.asarray(u)[tab.conn, ...], np.asarray(v)[tab.conn, ...]
    out = np.zeros(tab.nelems)
    for q in range(4):
        He, detJ = tab.Ht[:, :, q], tab.detJ[:, q]
        if ue.ndim == 2:
            out += np.einsum("n,ni,nj,ni,nj->n", detJ, He, He, ue, ve)
        else:
            out += np.einsum("n,ni,nj,nik,njk->n", detJ, He, He, ue, ve)
    return out * (1.0 - beta) * heat_capacity * density


def stress_dfds(tab, psi, phi):
    """d(sum_c psi_c^T G phi_c)/d(stress) at the Gauss points, (nelem, 3, 4) (buckling.py:283-307)"""
    psie, phie = tab.element_vectors2(psi), tab.element_vectors2(phi)
    if psie.ndim == 2:
        psie, phie = psie[:, :, None], phie[:, :, None]
    pp = psie[:, ::2] @ phie[:, ::2].transpose(0, 2, 1) + psie[:, 1::2] @ phie[:, 1::2].transpose(0, 2, 1)
    se = np.einsum("nijlm,njl->nim", tab.Te, pp)
    return tab.detJ[:, None, :] * se


def stress_uderiv(tab, C, dfds):
    """d/du of the same, full dof vector (buckling.py:309-319); C: (nelem, 3, 3) penalised constitutive matrices"""
    dfdue = np.einsum("nijm,nim->nj", tab.Be, C @ dfds)
    dfdu = np.zeros(2 * tab.nnodes)
    np.add.at(dfdu, 2 * tab.conn, dfdue[:, 0::2])
    np.add.at(dfdu, 2 * tab.conn + 1, dfdue[:, 1::2])
    return dfdu


def stress_xderiv(tab, C0, rhoE, p, u, dfds):
    """d/d rhoE of the same at fixed u (buckling.py:321-340, before the nodal averaging)"""
    ue = tab.element_vectors2(u)
    d = np.einsum("nim,ij->njm", dfds, C0)
    return np.einsum("njm,njkm,nk->n", d, tab.Be, ue) * p * rhoE ** (p - 1.0)


def element_to_node(conn, dfdrhoE, nnodes):
    """transpose of the element average rhoE = mean of the 4 nodal densities (buckling.py:209-213, 866-870)"""
    out = np.zeros(nnodes)
    for i in range(4):
        np.add.at(out, conn[:, i], dfdrhoE)
    return 0.25 * out


def node_to_element(conn, rho):
    """buckling.py:843-849"""
    return 0.25 * (rho[conn[:, 0]] + rho[conn[:, 1]] + rho[conn[:, 2]] + rho[conn[:, 3]])


# ----------------------------------------------------------------------------------------------------------------
# node filter (node_filter.py:10-217)
# ----------------------------------------------------------------------------------------------------------------
class NodeFilter:
    def __init__(self, conn, X, r0=1.0, ftype="spatial", dvmap=None, num_design_vars=None, beta=10.0, eta=0.5,
                 projection=False):
        self.conn, self.X = np.asarray(conn), np.asarray(X)
        self.nelems, self.nnodes = self.conn.shape[0], int(self.conn.max()) + 1
        self.ftype, self.r0, self.beta, self.eta, self.projection = ftype, r0, beta, eta, projection
        self.dvmap = None if dvmap is None or num_design_vars is None else np.asarray(dvmap)
        self.num_design_vars = self.nnodes if self.dvmap is None else int(num_design_vars)
        self.F = self.B = None
        if ftype == "spatial":                                       # node_filter.py:61-88
            tree = spatial.cKDTree(self.X)
            rows, cols, vals = [], [], []
            for i, idx in enumerate(tree.query_ball_point(self.X, r0)):
                idx = np.asarray(idx)
                w = r0 - np.sqrt(np.sum((self.X[i] - self.X[idx]) ** 2, axis=1))
                rows.append(np.full(len(idx), i))
                cols.append(idx)
                vals.append(w / np.sum(w))
            self.F = sparse.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                                       shape=(self.nnodes, self.nnodes))
        else:                                                        # node_filter.py:90-162
            xe, ye = self.X[self.conn, 0], self.X[self.conn, 1]
            Ae = np.zeros((self.nelems, 4, 4))
            Ce = np.zeros((self.nelems, 4, 4))
            for eta_ in GAUSS:
                for xi in GAUSS:
                    N, Nx, Ny, detJ = _physical_derivs(xi, eta_, xe, ye)
                    Ce += detJ[:, None, None] * np.outer(N, N)[None]
                    Ae += (detJ * r0**2)[:, None, None] * (np.einsum("ni,nj->nij", Nx, Nx) + np.einsum("ni,nj->nij", Ny, Ny))
            Ae += Ce
            ii = np.repeat(self.conn, 4, axis=1).ravel()
            jj = np.tile(self.conn, (1, 4)).ravel()
            self.factor = linalg.factorized(sparse.coo_matrix((Ae.ravel(), (ii, jj))).tocsc())
            self.B = sparse.coo_matrix((Ce.ravel(), (ii, jj))).tocsr()

    def _expand(self, x):
        if self.dvmap is None:
            return x
        x = np.asarray(x)[self.dvmap]
        x[self.dvmap <= -1] = 1.0
        return x

    def _linear(self, x):
        return self.F @ x if self.F is not None else self.factor(self.B @ x)

    def apply(self, x):
        rho = self._linear(self._expand(x))
        if self.projection:
            denom = np.tanh(self.beta * self.eta) + np.tanh(self.beta * (1.0 - self.eta))
            rho = (np.tanh(self.beta * self.eta) + np.tanh(self.beta * (rho - self.eta))) / denom
        return rho

    def apply_gradient(self, g, x=None):
        grad = g
        if self.projection:
            rho = self._linear(self._expand(x))
            denom = np.tanh(self.beta * self.eta) + np.tanh(self.beta * (1.0 - self.eta))
            grad = g * ((self.beta / denom) / np.cosh(self.beta * (rho - self.eta)) ** 2)
        g0 = self.F.T @ grad if self.F is not None else self.B.T @ self.factor(grad)
        if self.dvmap is None:
            return g0
        out = np.zeros(self.num_design_vars)
        np.add.at(out, self.dvmap[self.dvmap >= 0], g0[self.dvmap >= 0])
        return out


# ----------------------------------------------------------------------------------------------------------------
# aggregate functionals and their adjoint seeds
# ----------------------------------------------------------------------------------------------------------------
def eigenvector_aggregate(lam, Q, node, rho, mode="tanh"):
    """h = sum_i eta_i Q[node, i]^2 with tanh / exp weights (buckling.py:702-722); returns (h, eta, a, b)"""
    a = b = None
    if mode == "exp":
        eta = np.exp(-rho * (lam - np.min(lam)))
    else:
        a, b = np.tanh(rho * (lam - 0.0)), np.tanh(rho * (lam - 50.0))
        eta = a - b
    eta = eta / np.sum(eta)
    return float(np.sum(eta * Q[node, :] ** 2)), eta, a, b


def eigenvector_aggregate_seeds(lam, Q, node, rho, hb=1.0, mode="tanh"):
    """(Qb, lamb) of buckling.py:724-760"""
    h, eta, a, b = eigenvector_aggregate(lam, Q, node, rho, mode)
    Qb = np.zeros(Q.shape)
    Qb[node, :] = 2.0 * hb * eta * Q[node, :]
    if mode == "exp":
        lamb = -hb * rho * eta * (Q[node, :] ** 2 - h)
    else:
        lamb = -hb * rho * eta * (a + b) * (Q[node, :] ** 2 - h)
    return Qb, lamb


def ks_buckling(BLF, ks_rho):
    """KS maximum of mu = 1 / BLF and its weights (buckling.py:641-647, 650-654)"""
    mu = 1.0 / BLF
    c = np.max(mu)
    e = np.exp(ks_rho * (mu - c))
    return float(c + np.log(np.sum(e)) / ks_rho), e / np.sum(e), mu


def thermal_compliance(lam, Q, vec):
    """thermal.py:428-434"""
    val = Q[:, 1:].T @ vec
    return float(np.sum(val * val / lam[1:]))


def thermal_compliance_seeds(lam, Q, vec, compb=1.0):
    """(Qb, lamb) of thermal.py:436-442"""
    Qb, lamb = np.zeros(Q.shape), np.zeros(len(lam))
    val = Q[:, 1:].T @ vec
    Qb[:, 1:] = 2.0 * compb * np.outer(vec, val / lam[1:])
    lamb[1:] = -compb * val * val / lam[1:] ** 2
    return Qb, lamb


def min_frequency_ks(lam, Q, node_sets, ks_param=1.0, fixed_mass=1.0):
    """
    KS minimum of the natural frequencies with a point mass at each node set in turn (MinFreqOpt._eval_min_frequency,
    natural_frequency.py:742-807, with get_point_coefficients 530-550): per set the reduced problem
    diag(omega^2) q = w0^2 (I + m c^T c) q.  Returns (ks, omegab, {set: xcoefb}).
    """
    from scipy.linalg import eigh

    lam = np.asarray(lam, dtype=float)
    omega = np.sqrt(lam)
    N = len(omega)
    xcoef = []
    for nodes in node_sets:
        c0 = np.zeros((3, N))
        c0[0] = np.sum(Q[2 * nodes, :], axis=0) / len(nodes)
        c0[1] = np.sum(Q[2 * nodes + 1, :], axis=0) / len(nodes)
        xcoef.append(c0)
    eigs, ks0 = [], []
    min_val = np.min(omega)
    for c0 in xcoef:
        lam0, Q0 = eigh(np.diag(omega**2), np.eye(N) + fixed_mass * (c0.T @ c0))
        om0 = np.sqrt(lam0)
        eigs.append((om0, Q0))
        ks0.append(np.min(om0) - np.log(np.sum(np.exp(-ks_param * (om0 - np.min(om0))))) / ks_param)
        min_val = min(min_val, ks0[-1])
    eta0 = np.exp(-ks_param * (np.array(ks0) - min_val))
    ks = min_val - np.log(np.sum(eta0)) / ks_param
    eta0 = eta0 / np.sum(eta0)
    omegab = np.zeros(N)
    xcoefb = []
    for c0, (om0, Q0), e0 in zip(xcoef, eigs, eta0):
        w = np.exp(-ks_param * (om0 - np.min(om0)))
        om0b = 0.5 * (w / np.sum(w)) * e0 / om0
        omegab += 2.0 * omega * np.diag(Q0 @ (np.diag(om0b) @ Q0.T))
        cb = np.zeros(c0.shape)
        for i in range(N):
            cb -= 2.0 * om0b[i] * fixed_mass * om0[i] ** 2 * np.outer(c0 @ Q0[:, i], Q0[:, i])
        xcoefb.append(cb)
    return float(ks), omegab, xcoefb


def min_frequency_seeds(lam, Q, node_sets, ks_param=1.0, fixed_mass=1.0):
    """(ks, Qb, lamb): add_frequency_derivatives 527-532 and add_point_derivative 552-562 applied to the above"""
    ks, omegab, xcoefb = min_frequency_ks(lam, Q, node_sets, ks_param, fixed_mass)
    lamb = 0.5 * omegab / np.sqrt(np.asarray(lam, dtype=float))
    Qb = np.zeros(Q.shape)
    for nodes, cb in zip(node_sets, xcoefb):
        Qb[2 * nodes, :] += cb[0] / len(nodes)
        Qb[2 * nodes + 1, :] += cb[1] / len(nodes)
    return ks, Qb, lamb


class BucklingHarness:
    """
    The derivative side of examples/buckling.py at one design point, from stored matrices and fields: callbacks of
    finalize_adjoint (dAdu, dAdx, dBdx: 925-972), the fundamental-path adjoint (974-979), the KS gradient (650-700)
    and the compliance gradient (636-639).  ``Kr`` is the reduced stiffness, ``reduced`` the free dofs.
    """

    def __init__(self, conn, X, rhoE, u, reduced, Kr, p=3.0, rho0_G=1e-9, E=1.0, nu=0.3):
        self.tab = Q4Tables(conn, X)
        self.C0 = plane_stress_C0(E, nu)
        self.rhoE, self.u, self.p = np.asarray(rhoE), np.asarray(u), p
        self.reduced = np.asarray(reduced)
        self.nvars = 2 * self.tab.nnodes
        self.C = np.outer(self.rhoE**p + rho0_G, self.C0).reshape(-1, 3, 3)
        self.Kfact = linalg.factorized(sparse.csc_matrix(Kr))

    def full(self, xr):
        out = np.zeros((self.nvars,) + np.shape(xr)[1:])
        out[self.reduced, ...] = xr
        return out

    def dAdu(self, wr, vr):
        return stress_uderiv(self.tab, self.C, stress_dfds(self.tab, self.full(wr), self.full(vr)))

    def dAdx(self, wr, vr):
        dfds = stress_dfds(self.tab, self.full(wr), self.full(vr))
        return element_to_node(self.tab.conn, stress_xderiv(self.tab, self.C0, self.rhoE, self.p, self.u, dfds),
                               self.tab.nnodes)

    def dBdx(self, wr, vr):
        return element_to_node(self.tab.conn, stiffness_deriv(self.tab, self.C0, self.rhoE, self.p, self.full(wr),
                                                              self.full(vr)), self.tab.nnodes)

    def path_adjoint(self, dfdu0):
        """rhob contribution of u = K^-1 f: K adj = -dfdu0, then d(adj^T K u)/drho (buckling.py:974-979)"""
        adj = self.full(-self.Kfact(dfdu0[self.reduced]))
        return element_to_node(self.tab.conn, stiffness_deriv(self.tab, self.C0, self.rhoE, self.p, adj, self.u),
                               self.tab.nnodes)

    def ks_gradient(self, BLF, Qr, ks_rho):
        """d KS / d rho (nodal), tensor form (buckling.py:678-697)"""
        _, eta, mu = ks_buckling(BLF, ks_rho)
        Q = self.full(Qr)
        dKdx = self.dBdx(Qr * (eta * mu), Qr)
        dfds = stress_dfds(self.tab, Q * eta, Q)
        dGdx = element_to_node(self.tab.conn, stress_xderiv(self.tab, self.C0, self.rhoE, self.p, self.u, dfds),
                               self.tab.nnodes)
        dGdx += self.path_adjoint(stress_uderiv(self.tab, self.C, dfds))
        return -(dGdx + dKdx)

    def compliance_gradient(self):
        """buckling.py:636-639"""
        return -element_to_node(self.tab.conn, stiffness_deriv(self.tab, self.C0, self.rhoE, self.p, self.u, self.u),
                                self.tab.nnodes)
