"""
The harness around the eigd calls, on the device: what the reference's example drivers do
between the design variables and the matrices, and between the eigenvector adjoint and the design gradient.

  NodeFilter            examples/node_filter.py:10-217    spatial filter as CSR SpMV, Helmholtz filter as one more
                                                          factor (SpLuOperator) + SpMV, tanh projection, dv map
  ElementAverage        examples/buckling.py:843-849, 209-213   nodal -> element densities and the transposed map
  functionals           examples/buckling.py:641-760, thermal.py:428-442   KS / tanh / compliance aggregates and the
                                                          adjoint seeds (lamb, Qb) they hand to solve_adjoint
  BucklingAnalysis      examples/buckling.py:548-632, 822-986   assembly of K(x), the fundamental path u = K^-1 f, G(u, x),
                                                          shift-invert eigensolve, adjoint, total derivative INCLUDING
                                                          the path adjoint through u, chain rule back to x
  ModalAnalysis         examples/natural_frequency.py:317-392, 442-519; thermal.py:268-342, 560-623
                                                          the K - lam M harnesses (plate with rigid-body modes, heat
                                                          conduction) from x to df/dx

Vectors of the mesh (n, nnodes, nelems long) never visit the host between x and df/dx: matrices are assembled by
``eigd_assemble``, the gather / scatter / averaging maps are rectangular CSR products (fixed summation order,
reproducible), the callbacks are the element kernels of fem.hip.  N-sized bookkeeping (weights of the aggregates)
is numpy, as in the reference.
"""

import ctypes as C

import numpy as np
from scipy import sparse

from ._ffi import c_vp, call
from .device import (CSRMatrix, DeviceBlock, ElementAssembler, ElementBilinear, ElementLinearMatrices, _Buffer,
                     default_context)
from .fem import Q4Elements, plane_stress_C0
from .operators import SpLuOperator


def design_map(ctx, kind, x, p=0.0, c0=0.0, c1=0.0, g=None, out=None):
    """elementwise map of a device vector (eigd_design_map; kinds listed in include/eigd_hip.h)"""
    out = ctx.empty(x.n, 1) if out is None else out
    call("eigd_design_map", ctx.h, x.n, int(kind), float(p), float(c0), float(c1), x.ptr,
         g.ptr if g is not None else c_vp(None), out.ptr)
    return out


SIMP, SIMP_DERIV, PROJECT, PROJECT_DERIV, AFFINE = 0, 1, 2, 3, 4


class SparseMap:
    """a (rectangular) sparse matrix and its transpose on the device: y = M x, g = M^T y"""

    def __init__(self, ctx, M):
        M = sparse.csr_matrix(M)
        M.sort_indices()
        self.ctx, self.shape = ctx, M.shape
        self.fwd = CSRMatrix(ctx, M)
        self._host = M
        self._t = None

    @property
    def T(self):
        if self._t is None:
            Mt = self._host.T.tocsr()
            Mt.sort_indices()
            self._t = CSRMatrix(self.ctx, Mt)
        return self._t

    def apply(self, x, out=None):
        return self.fwd.apply(x, out)

    def apply_t(self, y, out=None):
        return self.T.apply(y, out)


def element_average(ctx, conn, nnodes):
    """rhoE = mean of the element's nodal densities (examples/buckling.py:843-849) as an nelem x nnodes map"""
    conn = np.asarray(conn)
    ne, npe = conn.shape
    E = sparse.csr_matrix((np.full(ne * npe, 1.0 / npe), (np.repeat(np.arange(ne), npe), conn.ravel())),
                          shape=(ne, nnodes))
    return SparseMap(ctx, E)


class NodeFilter:
    """
    Density filter of the reference (examples/node_filter.py) with its own call surface -- ``apply(x)``,
    ``apply_gradient(g, x)`` on numpy arrays -- and ``*_device`` forms on device vectors.
    spatial: rho = F x (cone weights inside r0, rows normalised; 61-88) -> one SpMV;
    helmholtz: (r0^2 K + M) rho = M x (90-162) -> SpMV + a triangular sweep of one more sparse factor.
    """

    def __init__(self, conn, X, r0=1.0, ftype="spatial", dvmap=None, num_design_vars=None, beta=10.0, eta=0.5,
                 projection=False, ctx=None):
        self.ctx = ctx if ctx is not None else default_context()
        self.conn, self.X = np.asarray(conn), np.asarray(X, dtype=float)
        self.nelems, self.nnodes = self.conn.shape[0], int(self.conn.max()) + 1
        self.ftype, self.r0 = ftype, r0
        self.beta, self.eta, self.projection = beta, eta, bool(projection)
        if dvmap is not None and num_design_vars is not None:
            self.dvmap, self.num_design_vars = np.asarray(dvmap), int(num_design_vars)
            inside = np.flatnonzero(self.dvmap >= 0)
            D = sparse.csr_matrix((np.ones(len(inside)), (inside, self.dvmap[inside])),
                                  shape=(self.nnodes, self.num_design_vars))
            self._D = SparseMap(self.ctx, D)
            self._fixed = self.ctx.from_host((self.dvmap <= -1).astype(float))  # nodes outside the design: x = 1
        else:
            self.dvmap, self.num_design_vars, self._D, self._fixed = None, self.nnodes, None, None
        self.F = self.B = self.factor = None
        if ftype == "spatial":
            from scipy.spatial import cKDTree

            pairs = cKDTree(self.X).sparse_distance_matrix(cKDTree(self.X), r0, output_type="coo_matrix")
            off = pairs.row != pairs.col
            w = sparse.csr_matrix((r0 - pairs.data[off], (pairs.row[off], pairs.col[off])),
                                  shape=(self.nnodes, self.nnodes)) + r0 * sparse.identity(self.nnodes, format="csr")
            inv = sparse.diags(1.0 / np.asarray(w.sum(axis=1)).ravel())
            self.F = SparseMap(self.ctx, inv @ w)
        else:
            el = Q4Elements(self.conn, self.X)
            Ce = el.capacity()
            Ae = r0**2 * el.conduction() + Ce
            dofs = self.conn.astype(np.int32)
            asm = ElementAssembler(self.ctx, dofs, self.nnodes)
            pat = asm.pattern()
            A = sparse.csr_matrix((asm.values_to_host(asm.assemble(Ae)), pat.indices, pat.indptr), shape=pat.shape)
            Bm = sparse.csr_matrix((asm.values_to_host(asm.assemble(Ce)), pat.indices, pat.indptr), shape=pat.shape)
            self.factor = SpLuOperator(A, ctx=self.ctx, check_symmetry=False, coords=self.X)
            self.B = SparseMap(self.ctx, Bm)

    # -- device forms -------------------------------------------------------------------------------------------
    def _expand(self, x_dev):
        if self._D is None:
            return x_dev
        xn = self._D.apply(x_dev)
        return xn.assign_lincomb([(1.0, xn), (1.0, self._fixed)])

    def _linear(self, xn):
        if self.F is not None:
            return self.F.apply(xn)
        rho = self.B.apply(xn)
        self.factor.solve_device(rho)
        return rho

    def apply_device(self, x_dev):
        rho = self._linear(self._expand(x_dev))
        if self.projection:
            rho = design_map(self.ctx, PROJECT, rho, p=self.beta, c0=self.eta)
        return rho

    def apply_gradient_device(self, g_dev, x_dev=None):
        grad = g_dev
        if self.projection:
            if x_dev is None:
                raise ValueError("the projection's derivative needs the design variables x")
            rho = self._linear(self._expand(x_dev))
            grad = design_map(self.ctx, PROJECT_DERIV, rho, p=self.beta, c0=self.eta, g=g_dev)
        if self.F is not None:
            g0 = self.F.apply_t(grad)
        else:
            t = self.ctx.empty(grad.n, 1).copy_from(grad)
            self.factor.solve_device(t)
            g0 = self.B.apply_t(t)
        return g0 if self._D is None else self._D.apply_t(g0)

    # -- reference call surface (numpy in, numpy out) -----------------------------------------------------------
    def apply(self, x):
        return self.apply_device(self.ctx.from_host(np.asarray(x, dtype=float))).get()[:, 0]

    def apply_gradient(self, g, x=None, rho=None):
        xd = None if x is None else self.ctx.from_host(np.asarray(x, dtype=float))
        return self.apply_gradient_device(self.ctx.from_host(np.asarray(g, dtype=float)), xd).get()[:, 0]


# --------------------------------------------------------------------------------------------------------------
# aggregate functionals: values and adjoint seeds (N-sized host arithmetic, as in the reference)
# --------------------------------------------------------------------------------------------------------------
def aggregate_weights(lam, rho, mode="tanh", lam_a=0.0, lam_b=50.0):
    """normalised weights eta and the tanh factors (a, b) (examples/buckling.py:703-713)"""
    lam = np.asarray(lam, dtype=float)
    if mode == "exp":
        eta, a, b = np.exp(-rho * (lam - np.min(lam))), None, None
    else:
        a, b = np.tanh(rho * (lam - lam_a)), np.tanh(rho * (lam - lam_b))
        eta = a - b
    return eta / np.sum(eta), a, b


def eigenvector_aggregate(lam, Qrow, rho, mode="tanh"):
    """h = sum_i eta_i q_i^2 for the eigenvector entries ``Qrow`` of one dof (examples/buckling.py:702-722)"""
    eta, _, _ = aggregate_weights(lam, rho, mode)
    return float(np.sum(eta * np.asarray(Qrow) ** 2))


def eigenvector_aggregate_seeds(lam, Qrow, rho, hb=1.0, mode="tanh"):
    """(row of Qb at that dof, lamb) (examples/buckling.py:724-760)"""
    eta, a, b = aggregate_weights(lam, rho, mode)
    Qrow = np.asarray(Qrow, dtype=float)
    h = float(np.sum(eta * Qrow**2))
    lamb = -hb * rho * eta * (Qrow**2 - h) * (1.0 if mode == "exp" else (a + b))
    return 2.0 * hb * eta * Qrow, lamb


def ks_buckling(BLF, ks_rho=160.0):
    """KS maximum of mu = 1 / BLF; returns (value, weights eta, mu) (examples/buckling.py:641-654)"""
    mu = 1.0 / np.asarray(BLF, dtype=float)
    c = np.max(mu)
    e = np.exp(ks_rho * (mu - c))
    return float(c + np.log(np.sum(e)) / ks_rho), e / np.sum(e), mu


def min_frequency_ks(lam, Q, node_sets, ks_param=1.0, fixed_mass=1.0):
    """
    The reference's minimum-frequency functional (examples/natural_frequency.py:742-807, MinFreqOpt): KS minimum over
    the natural frequencies of the structure carrying a point mass ``fixed_mass`` at each node set in turn -- per set a
    reduced N x N problem diag(omega^2) q = w^2 (I + m c^T c) q with c = mean eigenvector displacement over the set
    (530-550).  N-sized host arithmetic, as in the reference.  Returns (ks, Qb, lamb): the value and the adjoint seeds
    that solve_adjoint / add_total_derivative take (527-532, 552-562).  ``Q``: (n, N) eigenvectors of the non-rigid modes.
    """
    from scipy.linalg import eigh

    lam = np.asarray(lam, dtype=float)
    Q = np.asarray(Q)
    omega, N = np.sqrt(lam), len(lam)
    coefs = []
    for nodes in node_sets:
        nodes = np.asarray(nodes)
        coefs.append(np.stack([Q[2 * nodes].mean(axis=0), Q[2 * nodes + 1].mean(axis=0)]))
    sols, ks_set = [], []
    for c in coefs:
        w2, Y = eigh(np.diag(lam), np.eye(N) + fixed_mass * (c.T @ c))
        w0 = np.sqrt(w2)
        sols.append((w0, Y))
        ks_set.append(w0.min() - np.log(np.sum(np.exp(-ks_param * (w0 - w0.min())))) / ks_param)
    floor = min(omega.min(), min(ks_set))
    wt = np.exp(-ks_param * (np.array(ks_set) - floor))
    ks = floor - np.log(wt.sum()) / ks_param
    wt /= wt.sum()
    omegab = np.zeros(N)
    Qb = np.zeros(Q.shape)
    for nodes, c, (w0, Y), e0 in zip(node_sets, coefs, sols, wt):
        nodes = np.asarray(nodes)
        inner = np.exp(-ks_param * (w0 - w0.min()))
        w0b = 0.5 * (inner / inner.sum()) * e0 / w0                     # d ks / d (w0^2) chain through sqrt
        omegab += 2.0 * omega * np.einsum("ij,j,ij->i", Y, w0b, Y)
        cb = -2.0 * fixed_mass * np.einsum("j,aj,ij->ai", w0b * w0**2, c @ Y, Y)
        Qb[2 * nodes] += cb[0] / len(nodes)
        Qb[2 * nodes + 1] += cb[1] / len(nodes)
    return float(ks), Qb, 0.5 * omegab / omega


def thermal_compliance(lam, Q, vec):
    """sum_{i >= 1} (q_i . vec)^2 / lam_i (examples/thermal.py:428-434); Q, vec numpy"""
    val = np.asarray(Q)[:, 1:].T @ vec
    return float(np.sum(val * val / np.asarray(lam)[1:]))


def thermal_compliance_seeds(lam, Q, vec, compb=1.0):
    """(Qb, lamb) (examples/thermal.py:436-442)"""
    lam = np.asarray(lam, dtype=float)
    Qb, lamb = np.zeros(np.shape(Q)), np.zeros(len(lam))
    val = np.asarray(Q)[:, 1:].T @ vec
    Qb[:, 1:] = 2.0 * compb * np.outer(vec, val / lam[1:])
    lamb[1:] = -compb * val * val / lam[1:] ** 2
    return Qb, lamb


# --------------------------------------------------------------------------------------------------------------
class ElementLinearAdjoint:
    """
    Device callback ``cb(W, V)`` = d/du of sum_c w_c^T G(u) v_c for element matrices linear in u (the reference's
    ``dAdu``, examples/buckling.py:925-939 -> 283-319), returned on the REDUCED dofs (what the K factor acts on).
    ``accumulate`` adds into a device vector: add_total_derivative then never leaves HBM.
    """

    device = True

    def __init__(self, ctx, elin, elem_dofs, full_dofs, free_map, n_reduced, scale_dev):
        self.ctx, self.elin, self.scale = ctx, elin, scale_dev
        ed = np.ascontiguousarray(elem_dofs, dtype=np.int32)
        self._dofs = _Buffer(ctx, ed.nbytes)
        call("eigd_h2d", ctx.h, c_vp(self._dofs.ptr), ed.ctypes.data_as(c_vp), ed.nbytes)
        # incidence of the element entries on the reduced dofs: row = reduced dof of full_dofs[e, a], column = e*nd + a
        red = np.asarray(free_map)[np.asarray(full_dofs)].ravel()
        keep = np.flatnonzero(red >= 0)
        S = sparse.csr_matrix((np.ones(len(keep)), (red[keep], keep)), shape=(int(n_reduced), red.size))
        self.S = SparseMap(ctx, S)
        self.nout = self.nelem = int(n_reduced)

    def accumulate(self, W, V, out, alpha=1.0):
        ge = self.elin.adjoint(self._dofs, W, V, scale=self.scale, alpha=alpha)
        self.S.fwd.apply(ge, out, alpha=1.0, beta=1.0)
        return out

    def __call__(self, W, V):
        if not isinstance(W, DeviceBlock):
            W = self.ctx.from_host(np.asarray(W, dtype=float).reshape(np.shape(W)[0], -1))
            V = self.ctx.from_host(np.asarray(V, dtype=float).reshape(np.shape(V)[0], -1))
        out = self.ctx.zeros(self.nout, 1)
        return self.accumulate(W, V, out).get()[:, 0]


class BucklingAnalysis:
    """
    Device pipeline of the reference's buckling harness (examples/buckling.py TopologyAnalysis): design variables x
    -> filter -> element densities -> K(x), u = K^-1 f, G(u, x) -> (K + lam G) phi = 0 by shift-invert Lanczos ->
    adjoint + total derivative -> path adjoint through u -> filter transpose -> df/dx.

    ``conn``/``X``: Q4 mesh of congruent elements, ``fixed_dofs``: clamped dofs, ``f``: load (full dofs),
    ``fltr``: an eigd_amd.design.NodeFilter (or None: x are the nodal densities).
    """

    def __init__(self, conn, X, fixed_dofs, f, fltr=None, N=10, m=None, sigma=3.0, solver_type="IRAM", tol=0.0,
                 rtol=1e-10, eig_atol=1e-5, E=1.0, nu=0.3, p=3.0, rho0_K=1e-6, rho0_G=1e-9, adjoint_method="sibk",
                 adjoint_options=None, ctx=None):
        self.ctx = ctx = ctx if ctx is not None else default_context()
        self.el = el = Q4Elements(conn, X)
        self.fltr = fltr
        self.N, self.m, self.sigma, self.solver_type, self.tol = N, m, sigma, solver_type, tol
        self.rtol, self.eig_atol = rtol, eig_atol
        self.p, self.rho0_K, self.rho0_G = p, rho0_K, rho0_G
        self.adjoint_method = adjoint_method
        self.adjoint_options = dict(adjoint_options or {"lanczos_guess": True, "update_guess": False, "bs_target": 1})
        self.nnodes, self.nelems = el.nnodes, el.nelems
        self.nvars = 2 * el.nnodes
        fixed = np.zeros(self.nvars, dtype=bool)
        fixed[np.asarray(fixed_dofs, dtype=np.int64)] = True
        self.free_map = np.where(fixed, -1, np.cumsum(~fixed) - 1)
        self.reduced = np.flatnonzero(~fixed)
        self.n = len(self.reduced)
        self.C0 = plane_stress_C0(E, nu)
        self.Ke0 = el.stiffness(self.C0)
        self.full_dofs = el.dofs2()
        self.elem_dofs = el.dofs2(self.free_map)
        L, Q = el.stress_tables(self.C0)
        self.f = np.asarray(f, dtype=float)
        self.f_r = ctx.from_host(self.f[self.reduced])
        # device structures, analysed once per mesh
        self.avg = element_average(ctx, el.conn, el.nnodes)
        self.asm = ElementAssembler(ctx, self.elem_dofs, self.n)
        self.elin = ElementLinearMatrices(ctx, self.full_dofs, L, Q)
        self.expand = SparseMap(ctx, sparse.csr_matrix((np.ones(self.n), (self.reduced, np.arange(self.n))),
                                                       shape=(self.nvars, self.n)))
        pat = self.asm.pattern()
        self._pattern = pat
        coords = el.X[self.reduced // 2]
        ones = sparse.csr_matrix((np.ones(pat.nnz), pat.indices, pat.indptr), shape=pat.shape)
        ones = ones + sparse.identity(self.n, format="csr") * (10.0 * pat.nnz)   # placeholder values: analysis only
        ones.sort_indices()
        if ones.nnz != pat.nnz or not np.array_equal(ones.indices, pat.indices):
            raise ValueError("every dof needs a diagonal entry in the assembled pattern")
        self.Kfac = SpLuOperator(ones, ctx=ctx, check_symmetry=False, coords=coords)
        self.factor = SpLuOperator(ones, ctx=ctx, symbolic=self.Kfac.symbolic, check_symmetry=False)
        self.dK = CSRMatrix(ctx, ones)
        self.dG = CSRMatrix(ctx, ones)
        self._shifted = CSRMatrix(ctx, ones)
        self.x = None

    # ---------------------------------------------------------------------------------------------------------
    def initialize(self, x):
        """assemble, solve the fundamental path, factor the shifted matrix, solve the eigenproblem (548-632, 822-857)"""
        ctx = self.ctx
        self.x = np.asarray(x, dtype=float)
        self.x_dev = ctx.from_host(self.x)
        self.rho = self.fltr.apply_device(self.x_dev) if self.fltr is not None else self.x_dev
        self.rhoE = self.avg.apply(self.rho)
        sK = design_map(ctx, SIMP, self.rhoE, p=self.p, c0=self.rho0_K)
        self.sG = design_map(ctx, SIMP, self.rhoE, p=self.p, c0=self.rho0_G)
        self.dscale = design_map(ctx, SIMP_DERIV, self.rhoE, p=self.p)            # p rho^(p-1): both K and G (207-208, 336)
        vK = self.asm.assemble(self.Ke0, sK)
        self.dK.update_values_device(vK)
        self.Kfac.refactor_device(vK, indefinite_matrix=self.dK)
        self.u_r = ctx.empty(self.n, 1).copy_from(self.f_r)
        self.Kfac.solve_device(self.u_r)                                           # u = K^-1 f (560-562)
        self.u_full = self.expand.apply(self.u_r)
        self.Ge = self.elin(self.u_full)                                           # unit stress-stiffness matrices
        vG = self.asm.assemble(self.Ge, self.sG)
        self.dG.update_values_device(vG)
        vS = ctx.empty(vK.n, 1).assign_lincomb([(1.0, vK), (float(self.sigma), vG)])
        self._shifted.update_values_device(vS)
        self.factor.refactor_device(vS, indefinite_matrix=self._shifted)           # K + sigma G (582-584)
        self.factor.count = 0
        from .lanczos import IRAM, BasicLanczos

        if self.solver_type == "IRAM":
            m = self.m if self.m is not None else max(2 * self.N + 1, 60)
            self.eig_solver = IRAM(N=self.N, m=m, eig_atol=self.eig_atol, mode="buckling", ctx=ctx)
        else:
            m = self.m if self.m is not None else max(3 * self.N + 1, 60)
            self.eig_solver = BasicLanczos(N=self.N, m=m, eig_atol=self.eig_atol, tol=self.tol, mode="buckling", ctx=ctx)
        self.lam, self.Qr = self.eig_solver.solve(self.dG, self.dK, self.factor, self.sigma)
        self.BLF = self.lam[: self.N]
        return self.lam, self.Qr

    # ---- functionals -----------------------------------------------------------------------------------------
    def compliance(self):
        return float(self.f_r.coldot(self.u_r)[0])

    def reduced_index(self, node, comp=0):
        r = int(self.free_map[2 * node + comp])
        if r < 0:
            raise ValueError("that dof is clamped")
        return r

    def get_eigenvector_aggregate(self, rho, node, mode="tanh"):
        """as the reference writes it (np.dot(Q[node, i], Q[node, i]) with an integer node: the full dof `node`)"""
        return eigenvector_aggregate(self.lam, self._qrow(node), rho, mode)

    def _qrow(self, dof):
        r = int(self.free_map[dof])
        return self.Qr[r, :] if r >= 0 else np.zeros(self.Qr.shape[1])

    def eigenvector_aggregate_seeds(self, rho, node, hb=1.0, mode="tanh"):
        """(Qrb, lamb) for solve_adjoint / add_total_derivative (724-760); `node` indexes the full dof vector"""
        row, lamb = eigenvector_aggregate_seeds(self.lam, self._qrow(node), rho, hb, mode)
        Qrb = np.zeros(self.Qr.shape)
        r = int(self.free_map[node])
        if r >= 0:
            Qrb[r, :] = row
        return Qrb, lamb

    # ---- derivative side -------------------------------------------------------------------------------------
    def _callbacks(self):
        ctx = self.ctx
        dAdx = ElementBilinear.from_device(ctx, self.elem_dofs, self.Ge, self.dscale)       # d(w^T G v)/d rhoE at fixed u
        dBdx = ElementBilinear.from_device(ctx, self.elem_dofs, self.Ke0, self.dscale)      # d(w^T K v)/d rhoE
        dAdu = ElementLinearAdjoint(ctx, self.elin, self.elem_dofs, self.full_dofs, self.free_map, self.n, self.sG)
        return dAdx, dBdx, dAdu

    def path_adjoint_into(self, dfdu_r, rhoEb_dev, dBdx):
        """K adj = -dfdu, then rhoEb += d(adj^T K u)/d rhoE (974-979)"""
        adj = self.ctx.empty(self.n, 1).copy_from(dfdu_r)
        self.Kfac.solve_device(adj, alpha=-1.0)
        dBdx.accumulate(adj, self.u_r, rhoEb_dev, alpha=1.0)
        return rhoEb_dev

    def chain_to_design(self, rhoEb_dev):
        """element -> node (transpose of the average) -> filter transpose -> design variables (977-981)"""
        rhob = self.avg.apply_t(rhoEb_dev)
        if self.fltr is None:
            return rhob, rhob
        return rhob, self.fltr.apply_gradient_device(rhob, self.x_dev)

    def finalize_adjoint(self, Qrb, lamb, comm=None):
        """
        solve_adjoint + the two total derivatives + path adjoint + chain rule (874-986).  Returns a dict with the
        stages the reference's harness produces: psir, corr_data, dfdu0 (reduced dofs), rhoEb, rhob, xb.
        """
        ctx = self.ctx
        s = self.eig_solver
        dQrb = Qrb if isinstance(Qrb, DeviceBlock) else ctx.from_host(Qrb)
        dpsi, data = s.solve_adjoint(dQrb, rtol=self.rtol, method=self.adjoint_method, comm=comm, **self.adjoint_options)
        dAdx, dBdx, dAdu = self._callbacks()
        dfdu0 = s.add_total_derivative(lamb, dQrb, dpsi, dAdu, None, np.zeros(self.n), adj_corr_data=data,
                                       deriv_type="tensor", comm=comm)
        rhoEb = s.add_total_derivative(lamb, dQrb, dpsi, dAdx, dBdx, np.zeros(self.nelems), adj_corr_data=data,
                                       deriv_type="tensor", comm=comm)
        rhoEb_dev = ctx.from_host(rhoEb)
        rhob_eig = self.avg.apply_t(rhoEb_dev).get()[:, 0]
        self.path_adjoint_into(ctx.from_host(dfdu0), rhoEb_dev, dBdx)
        rhob, xb = self.chain_to_design(rhoEb_dev)
        return {"psir": dpsi, "corr_data": data, "dfdu0": dfdu0, "rhoEb": rhoEb_dev.get()[:, 0], "rhob_eig": rhob_eig,
                "rhob": rhob.get()[:, 0], "xb": xb.get()[:, 0]}

    def ks_buckling_gradient(self, ks_rho=160.0):
        """d KS(1 / BLF) / dx, tensor form with the path adjoint (650-700)"""
        ctx = self.ctx
        _, eta, mu = ks_buckling(self.BLF, ks_rho)
        dAdx, dBdx, dAdu = self._callbacks()
        Q = self.eig_solver._prob.Phi
        Qeta = ctx.empty(Q.n, Q.k).assign_lincomb([(eta, Q)])
        Qem = ctx.empty(Q.n, Q.k).assign_lincomb([(eta * mu, Q)])
        acc = ctx.zeros(self.nelems, 1)
        dBdx.accumulate(Qem, Q, acc)                                  # dKdx
        dAdx.accumulate(Qeta, Q, acc)                                 # dGdx at fixed u
        dGdu = ctx.zeros(self.n, 1)
        dAdu.accumulate(Qeta, Q, dGdu)
        self.path_adjoint_into(dGdu, acc, dBdx)
        acc.assign_lincomb([(-1.0, acc)])
        return self.chain_to_design(acc)[1].get()[:, 0]

    def compliance_gradient(self):
        """d(f . u)/dx = -d(u^T K u)/dx (636-639)"""
        _, dBdx, _ = self._callbacks()
        acc = self.ctx.zeros(self.nelems, 1)
        dBdx.accumulate(self.u_r, self.u_r, acc, alpha=-1.0)
        return self.chain_to_design(acc)[1].get()[:, 0]


class ModalAnalysis:
    """
    Device pipeline of the reference's K - lam M harnesses: ``kind="natural_frequency"`` (examples/natural_frequency.py:
    Q4 plane-stress plate, free-free, 2 dof / node; the three rigid-body modes are solved for and dropped, 348, 383-384)
    and ``kind="thermal"`` (examples/thermal.py: heat conduction, 1 dof / node).  Design variables x -> filter -> element
    densities -> K(x), M(x) assembled on the device -> K phi = lam M phi by shift-invert Lanczos -> adjoint + total
    derivative -> filter transpose -> df/dx; no vector of the mesh visits the host in between.

      natural_frequency:  K = sum (rho^p + rho0_K) Ke0,               M = sum density * rho * Me0        (134-160, 205-236)
      thermal:            K = sum kappa ((1 - beta) rho^p + beta) Kt0,  M = sum c rho_d ((1 - beta) rho + beta) Mt0  (126-148, 192-214)
    """

    def __init__(self, conn, X, kind="natural_frequency", fltr=None, N=10, m=None, sigma=None, solver_type="IRAM",
                 tol=None, rtol=1e-10, eig_atol=1e-5, Ntarget=None, E=1.0, nu=0.3, p=3.0, rho0_K=1e-6, density=1.0,
                 kappa=1.0, heat_capacity=1.0, beta=1e-6, adjoint_method="sibk", adjoint_options=None, ctx=None,
                 unit_mass=False):
        if kind not in ("natural_frequency", "thermal"):
            raise ValueError(f"Unknown kind {kind!r}")
        # unit_mass: the symmetric STANDARD eigenproblem K phi = lam phi (B = I, independent of the design) on the same
        # mesh and sparsity pattern -- BASELINE configs[3] as it is worded; the reference's thermal example itself is
        # generalized (consistent capacity matrix, thermal.py:192-214)
        self.unit_mass = bool(unit_mass)
        self.ctx = ctx = ctx if ctx is not None else default_context()
        self.kind, self.fltr = kind, fltr
        self.el = el = Q4Elements(conn, X)
        self.nrigid = 3 if kind == "natural_frequency" else 0
        self.N, self.m, self.solver_type, self.rtol, self.eig_atol, self.Ntarget = N, m, solver_type, rtol, eig_atol, Ntarget
        self.sigma = sigma if sigma is not None else (-10.0 if kind == "natural_frequency" else -0.1)
        self.tol = tol if tol is not None else (1e-14 if kind == "natural_frequency" else 0.0)
        self.p, self.rho0_K, self.density = p, rho0_K, density
        self.kappa, self.heat_capacity, self.beta = kappa, heat_capacity, beta
        self.adjoint_method = adjoint_method
        self.adjoint_options = dict(adjoint_options or {"lanczos_guess": True, "update_guess": False, "bs_target": 1})
        self.nnodes, self.nelems = el.nnodes, el.nelems
        if kind == "natural_frequency":
            self.Ke0, self.Me0 = el.stiffness(plane_stress_C0(E, nu)), el.mass()
            self.elem_dofs = el.dofs2()
            self.n = 2 * el.nnodes
            coords = np.repeat(el.X, 2, axis=0)
        else:
            self.Ke0, self.Me0 = el.conduction(), el.capacity()
            self.elem_dofs = el.conn.astype(np.int32)
            self.n = el.nnodes
            coords = el.X
        self.avg = element_average(ctx, el.conn, el.nnodes)
        self.asm = ElementAssembler(ctx, self.elem_dofs, self.n)
        pat = self.asm.pattern()
        ones = sparse.csr_matrix((np.ones(pat.nnz), pat.indices, pat.indptr), shape=pat.shape) \
            + sparse.identity(self.n, format="csr") * (10.0 * pat.nnz)   # placeholder values: analysis only
        ones.sort_indices()
        if ones.nnz != pat.nnz or not np.array_equal(ones.indices, pat.indices):
            raise ValueError("every dof needs a diagonal entry in the assembled pattern")
        self.factor = SpLuOperator(ones, ctx=ctx, check_symmetry=False, coords=coords)
        self.dK, self.dM, self._shifted = CSRMatrix(ctx, ones), CSRMatrix(ctx, ones), CSRMatrix(ctx, ones)

    def _scales(self, rhoE):
        """(K scale, M scale, dK/drho factor, dM/drho factor) per element, on the device"""
        ctx, one = self.ctx, None
        if self.kind == "natural_frequency":
            sK = design_map(ctx, SIMP, rhoE, p=self.p, c0=self.rho0_K)
            sM = design_map(ctx, AFFINE, rhoE, c0=self.density, c1=0.0)
            dK = design_map(ctx, SIMP_DERIV, rhoE, p=self.p)
            dM = design_map(ctx, AFFINE, rhoE, c0=0.0, c1=self.density)
        else:
            a = self.kappa * (1.0 - self.beta)
            sK = design_map(ctx, SIMP, rhoE, p=self.p, c0=0.0)
            sK = design_map(ctx, AFFINE, sK, c0=a, c1=self.kappa * self.beta)
            c = self.heat_capacity * self.density
            sM = design_map(ctx, AFFINE, rhoE, c0=c * (1.0 - self.beta), c1=c * self.beta)
            dK = design_map(ctx, SIMP_DERIV, rhoE, p=self.p)
            dK = design_map(ctx, AFFINE, dK, c0=a, c1=0.0)
            dM = design_map(ctx, AFFINE, rhoE, c0=0.0, c1=c * (1.0 - self.beta))
        return sK, sM, dK, dM

    def initialize(self, x):
        """assemble K(x), M(x), factor K - sigma M, solve the eigenproblem; returns (lam, Q) without the rigid-body modes"""
        ctx = self.ctx
        self.x = np.asarray(x, dtype=float)
        self.x_dev = ctx.from_host(self.x)
        self.rho = self.fltr.apply_device(self.x_dev) if self.fltr is not None else self.x_dev
        self.rhoE = self.avg.apply(self.rho)
        sK, sM, self._dKs, self._dMs = self._scales(self.rhoE)
        vK = self.asm.assemble(self.Ke0, sK)
        if self.unit_mass:
            pat = self.asm.pattern()
            rows = np.repeat(np.arange(self.n), np.diff(pat.indptr))
            vM = ctx.from_host((pat.indices == rows).astype(np.float64).reshape(-1, 1))   # identity in the assembled pattern
            self._dMs = ctx.zeros(self.nelems, 1)                                       # d I / d rho = 0
        else:
            vM = self.asm.assemble(self.Me0, sM)
        self.dK.update_values_device(vK)
        self.dM.update_values_device(vM)
        vS = ctx.empty(vK.n, 1).assign_lincomb([(1.0, vK), (-float(self.sigma), vM)])
        self._shifted.update_values_device(vS)
        self.factor.refactor_device(vS, indefinite_matrix=self._shifted)
        self.factor.count = 0
        from .lanczos import IRAM, BasicLanczos

        Nall = self.N + self.nrigid
        if self.solver_type == "IRAM":
            m = self.m if self.m is not None else max(2 * Nall + 1, 60)
            self.eig_solver = IRAM(N=Nall, m=m, eig_atol=self.eig_atol, ctx=ctx)
        else:
            m = self.m if self.m is not None else max(3 * Nall + 1, 60)
            self.eig_solver = BasicLanczos(N=Nall, m=m, eig_atol=self.eig_atol, tol=self.tol, Ntarget=self.Ntarget, ctx=ctx)
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")   # (the rigid-body cluster at lam ~ 1e-15 is numerically repeated by construction)
            lam, Q = self.eig_solver.solve(self.dK, self.dM, self.factor, self.sigma)
        self.lam_all, self.Q_all = lam, Q
        self.lam, self.Q = lam[self.nrigid:], Q[:, self.nrigid:]
        return self.lam, self.Q

    def finalize_adjoint(self, Qb, lamb):
        """
        solve_adjoint + total derivative + chain rule for seeds on the non-rigid modes (natural_frequency.py:442-519,
        thermal.py:560-623); returns psi, corr_data, rhoEb, rhob, xb.
        """
        ctx, r = self.ctx, self.nrigid
        Q0b = np.zeros((self.n, len(self.lam_all)))
        Q0b[:, r:] = Qb
        lamb0 = np.zeros(len(self.lam_all))
        lamb0[r:] = lamb
        dQ0b = ctx.from_host(Q0b)
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            dpsi, data = self.eig_solver.solve_adjoint(dQ0b, rtol=self.rtol, method=self.adjoint_method,
                                                       **self.adjoint_options)
        data0 = {}
        for i in data:                       # pairs that involve a rigid-body mode are dropped (486-497)
            if i >= r:
                items = [(j, xi, eta) for j, xi, eta in data[i] if j >= r]
                if items:
                    data0[i] = items
        dAdx = ElementBilinear.from_device(ctx, self.elem_dofs, self.Ke0, self._dKs)
        dBdx = ElementBilinear.from_device(ctx, self.elem_dofs, self.Me0, self._dMs)
        rhoEb = self.eig_solver.add_total_derivative(lamb0, dQ0b, dpsi, dAdx, dBdx, np.zeros(self.nelems),
                                                     adj_corr_data=data0, deriv_type="tensor")
        rhob = self.avg.apply_t(ctx.from_host(rhoEb))
        xb = rhob if self.fltr is None else self.fltr.apply_gradient_device(rhob, self.x_dev)
        return {"psi": dpsi, "corr_data": data, "rhoEb": rhoEb, "rhob": rhob.get()[:, 0], "xb": xb.get()[:, 0]}
