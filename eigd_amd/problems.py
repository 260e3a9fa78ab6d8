"""
Synthetic structured-mesh problems for the benchmark and the tests: vectorised re-statement of
the matrices the reference's example harnesses assemble (out of the hot path; the reference's
Python triple loops take ~35 s at 1 M dof).

  q4_plane_stress_mesh      node / element numbering of examples/buckling.py:1300-1325
  plane_stress_K            examples/buckling.py:152-176   (SIMP penalised Q4 stiffness)
  stress_stiffness_G        examples/buckling.py:220-255   (geometric stiffness G(u))
  consistent_mass_M         examples/natural_frequency.py:205-236
  thermal_K / thermal_M     examples/thermal.py:126-148, 192-214 (1 dof / node)

All element integrals use 2 x 2 Gauss quadrature as examples/fe_utils.py.  The meshes are uniform
rectangles, so the unit element matrices are computed once and scaled per element.
"""

import numpy as np
from scipy import sparse

GAUSS = (-1.0 / np.sqrt(3.0), 1.0 / np.sqrt(3.0))


class Q4Mesh:
    def __init__(self, nx, ny, Lx=1.0, Ly=1.0):
        self.nx, self.ny, self.Lx, self.Ly = nx, ny, Lx, Ly
        self.nnodes = (nx + 1) * (ny + 1)
        self.nelems = nx * ny
        nodes = np.arange(self.nnodes).reshape(nx + 1, ny + 1)  # node(i, j) = i*(ny+1) + j
        self.nodes = nodes
        i, j = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
        conn = np.stack([nodes[i, j], nodes[i + 1, j], nodes[i + 1, j + 1], nodes[i, j + 1]], axis=-1)
        # element index = i + nx * j (buckling.py:1320-1325)
        self.conn = conn.transpose(1, 0, 2).reshape(-1, 4)
        self.hx, self.hy = Lx / nx, Ly / ny
        x = np.linspace(0, Lx, nx + 1)
        y = np.linspace(0, Ly, ny + 1)
        self.X = np.stack([np.repeat(x, ny + 1), np.tile(y, nx + 1)], axis=1)

    def shape_derivs(self, xi, eta):
        """N, dN/dx, dN/dy at a quadrature point of the (uniform) element; detJ"""
        N = 0.25 * np.array([(1 - xi) * (1 - eta), (1 + xi) * (1 - eta), (1 + xi) * (1 + eta), (1 - xi) * (1 + eta)])
        Nxi = 0.25 * np.array([-(1 - eta), (1 - eta), (1 + eta), -(1 + eta)])
        Neta = 0.25 * np.array([-(1 - xi), -(1 + xi), (1 + xi), (1 - xi)])
        detJ = 0.25 * self.hx * self.hy
        return N, Nxi * (2.0 / self.hx), Neta * (2.0 / self.hy), detJ


def _assemble(mesh, Ke, dofs_per_node, free_map=None):
    """COO -> CSR assembly of per-element matrices Ke (nelems, nd, nd) or a single (nd, nd) scaled by `scale`"""
    conn = mesh.conn
    nd = 4 * dofs_per_node
    var = np.empty((mesh.nelems, nd), dtype=np.int64)
    for a in range(dofs_per_node):
        var[:, a::dofs_per_node] = dofs_per_node * conn + a
    ntot = dofs_per_node * mesh.nnodes
    if free_map is not None:
        var = free_map[var]
        ntot = int(free_map.max()) + 1
    rows = np.repeat(var, nd, axis=1).ravel()
    cols = np.tile(var, (1, nd)).ravel()
    vals = Ke.reshape(mesh.nelems, -1).ravel()
    if free_map is not None:
        keep = (rows >= 0) & (cols >= 0)
        rows, cols, vals = rows[keep], cols[keep], vals[keep]
    A = sparse.coo_matrix((vals, (rows, cols)), shape=(ntot, ntot)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A, var


def plane_stress_C0(E=1.0, nu=0.3):
    return (E / (1.0 - nu**2)) * np.array([[1.0, nu, 0.0], [nu, 1.0, 0.0], [0.0, 0.0, 0.5 * (1.0 - nu)]])


def _Bmat(Nx, Ny):
    Be = np.zeros((3, 8))
    Be[0, ::2] = Nx
    Be[1, 1::2] = Ny
    Be[2, ::2] = Ny
    Be[2, 1::2] = Nx
    return Be


def unit_stiffness(mesh, C0):
    Ke0 = np.zeros((8, 8))
    for eta in GAUSS:
        for xi in GAUSS:
            _, Nx, Ny, detJ = mesh.shape_derivs(xi, eta)
            Be = _Bmat(Nx, Ny)
            Ke0 += detJ * Be.T @ C0 @ Be
    return Ke0


def unit_mass(mesh):
    Me0 = np.zeros((8, 8))
    for eta in GAUSS:
        for xi in GAUSS:
            N, _, _, detJ = mesh.shape_derivs(xi, eta)
            He = np.zeros((2, 8))
            He[0, ::2] = N
            He[1, 1::2] = N
            Me0 += detJ * He.T @ He
    return Me0


class BucklingColumn:
    """
    Compressed column of examples/buckling.py (domain_compressed_column 1300-1369): bottom edge
    clamped, vertical tip load 1e-3 spread over the top-middle nodes; (K + lam G(u)) phi = 0.
    Matrices are assembled directly in the reduced (free-dof) numbering.
    """

    def __init__(self, nx, ny, Lx=1.0, Ly=1.0, E=1.0, nu=0.3, p=3.0, rho0_K=1e-6, rho0_G=1e-9, rhoE=None, seed=0):
        self.mesh = mesh = Q4Mesh(nx, ny, Lx, Ly)
        self.p, self.rho0_K, self.rho0_G = p, rho0_K, rho0_G
        self.C0 = plane_stress_C0(E, nu)
        if rhoE is None:
            rhoE = np.random.default_rng(seed).uniform(0.3, 1.0, size=mesh.nelems)
        self.rhoE = np.asarray(rhoE, dtype=float)
        nvars = 2 * mesh.nnodes
        fixed = np.zeros(nvars, dtype=bool)
        bottom = mesh.nodes[:, 0]
        fixed[2 * bottom] = True
        fixed[2 * bottom + 1] = True
        self.free_map = np.where(fixed, -1, np.cumsum(~fixed) - 1)
        self.reduced = np.flatnonzero(~fixed)
        self.n = len(self.reduced)
        P = 1e-3
        f = np.zeros(nvars)
        offset = int(np.ceil(nx / 30))
        top = mesh.nodes[:, ny]
        for i in range(offset):
            f[2 * top[nx // 2 - i - 1] + 1] += -P / (2 * offset + 1)
            f[2 * top[nx // 2 + i + 1] + 1] += -P / (2 * offset + 1)
        f[2 * top[nx // 2] + 1] += -P / (2 * offset + 1)
        self.f = f
        self.Ke0 = unit_stiffness(mesh, self.C0)
        self.elem_dofs = None

    def stiffness(self):
        scale = self.rhoE**self.p + self.rho0_K
        Ke = scale[:, None, None] * self.Ke0[None]
        K, var = _assemble(self.mesh, Ke, 2, self.free_map)
        self.elem_dofs = var.astype(np.int32)  # reduced dof of each element dof, -1 if clamped
        return K

    def element_G(self, u_full):
        """unit (unpenalised) element geometric stiffness matrices, 8 x 8 each (buckling.py:220-255)"""
        mesh = self.mesh
        ue = np.empty((mesh.nelems, 8))
        ue[:, ::2] = u_full[2 * mesh.conn]
        ue[:, 1::2] = u_full[2 * mesh.conn + 1]
        G4 = np.zeros((mesh.nelems, 4, 4))
        for eta in GAUSS:
            for xi in GAUSS:
                _, Nx, Ny, detJ = mesh.shape_derivs(xi, eta)
                Be = _Bmat(Nx, Ny)
                s = ue @ (self.C0 @ Be).T  # (nelems, 3) stresses of the unit-stiffness material
                Te = np.stack([np.outer(Nx, Nx), np.outer(Ny, Ny), np.outer(Nx, Ny) + np.outer(Ny, Nx)])
                G4 += detJ * np.einsum("ni,ijl->njl", s, Te)
        Ge = np.zeros((mesh.nelems, 8, 8))
        Ge[:, 0::2, 0::2] = G4
        Ge[:, 1::2, 1::2] = G4
        return Ge

    def stress_stiffness_tables(self):
        """
        (full_dofs, L, Q) with Ge_unit[e] = sum_m (L[m] . u_e) Q[m]: the Gauss-point form of element_G for the device
        (eigd_amd.device.ElementLinearMatrices).  m runs over 4 Gauss points x 3 stress components.
        """
        mesh = self.mesh
        full = np.empty((mesh.nelems, 8), dtype=np.int32)
        full[:, ::2] = 2 * mesh.conn
        full[:, 1::2] = 2 * mesh.conn + 1
        L, Q = [], []
        for eta in GAUSS:
            for xi in GAUSS:
                _, Nx, Ny, detJ = mesh.shape_derivs(xi, eta)
                CB = self.C0 @ _Bmat(Nx, Ny)                      # stresses of the unit-stiffness material, 3 x 8
                Te = [np.outer(Nx, Nx), np.outer(Ny, Ny), np.outer(Nx, Ny) + np.outer(Ny, Nx)]
                for i in range(3):
                    G8 = np.zeros((8, 8))
                    G8[0::2, 0::2] = detJ * Te[i]
                    G8[1::2, 1::2] = detJ * Te[i]
                    L.append(CB[i])
                    Q.append(G8)
        return full, np.array(L), np.array(Q)

    def geometric_stiffness(self, u_full):
        self.Ge_unit = self.element_G(u_full)
        scale = self.rhoE**self.p + self.rho0_G
        G, _ = _assemble(self.mesh, scale[:, None, None] * self.Ge_unit, 2, self.free_map)
        return G

    def full_vector(self, ur):
        u = np.zeros(2 * self.mesh.nnodes)
        u[self.reduced] = ur
        return u

    def dof_coords(self):
        """(n, 2) coordinates of the free dofs: optional ordering hint for SpLuOperator(coords=...)"""
        return self.mesh.X[self.reduced // 2]

    # d/d rhoE of w^T K v and w^T G v at fixed u: element-wise bilinear forms (buckling.py:178-218, 283-340)
    def dK_scale(self):
        return self.p * self.rhoE ** (self.p - 1.0)

    def dG_scale(self):
        return self.p * self.rhoE ** (self.p - 1.0)


class FreePlate:
    """free-free Q4 plate of examples/natural_frequency.py: K phi = lam M phi, 3 rigid-body modes"""

    def __init__(self, nx, ny, Lx=1.0, Ly=1.0, E=1.0, nu=0.3, p=3.0, rho0_K=1e-6, density=1.0, rhoE=None, seed=0):
        self.mesh = mesh = Q4Mesh(nx, ny, Lx, Ly)
        self.p, self.rho0_K, self.density = p, rho0_K, density
        self.C0 = plane_stress_C0(E, nu)
        if rhoE is None:
            rhoE = np.random.default_rng(seed).uniform(0.3, 1.0, size=mesh.nelems)
        self.rhoE = np.asarray(rhoE, dtype=float)
        self.n = 2 * mesh.nnodes
        self.Ke0 = unit_stiffness(mesh, self.C0)
        self.Me0 = unit_mass(mesh)

    def stiffness(self):
        K, var = _assemble(self.mesh, (self.rhoE**self.p + self.rho0_K)[:, None, None] * self.Ke0[None], 2)
        self.elem_dofs = var.astype(np.int32)
        return K

    def mass(self):
        M, _ = _assemble(self.mesh, (self.density * self.rhoE)[:, None, None] * self.Me0[None], 2)
        return M

    def dof_coords(self):
        return np.repeat(self.mesh.X, 2, axis=0)


class ThermalPlate:
    """square heat-conduction plate of examples/thermal.py (1 dof / node): repeated eigenvalues when Lx == Ly"""

    def __init__(self, nx, Lx=1.0, epsilon=0.0, kappa=1.0, heat_capacity=1.0, p=3.0, rho0=1e-6, rhoE=None):
        self.mesh = mesh = Q4Mesh(nx, nx, Lx, Lx + epsilon)
        self.rhoE = np.full(mesh.nelems, 0.5) if rhoE is None else np.asarray(rhoE, dtype=float)
        self.p, self.rho0, self.kappa, self.c = p, rho0, kappa, heat_capacity
        self.n = mesh.nnodes
        Ke0 = np.zeros((4, 4))
        Me0 = np.zeros((4, 4))
        for eta in GAUSS:
            for xi in GAUSS:
                N, Nx, Ny, detJ = mesh.shape_derivs(xi, eta)
                Ke0 += detJ * kappa * (np.outer(Nx, Nx) + np.outer(Ny, Ny))
                Me0 += detJ * heat_capacity * np.outer(N, N)
        self.Ke0, self.Me0 = Ke0, Me0

    def stiffness(self):
        K, var = _assemble(self.mesh, (self.rhoE**self.p + self.rho0)[:, None, None] * self.Ke0[None], 1)
        self.elem_dofs = var.astype(np.int32)
        return K

    def mass(self):
        M, _ = _assemble(self.mesh, (self.rhoE + self.rho0)[:, None, None] * self.Me0[None], 1)
        return M

    def dof_coords(self):
        return self.mesh.X


class ShellBox:
    """
    Stand-in for the reference's CRM wingbox configuration (BASELINE configs[4]; examples/crm.py needs TACS and a mesh
    file that are not available, SURVEY.md 8d "C5"): a thin-walled box beam of flat-shell Q4 facets on a closed
    3-D surface grid, 6 dof / node (3 translations, 3 rotations), clamped at the root.

      K(t) = sum_e t_e (K_membrane + K_shear + K_drill)_e + t_e^3 (K_bending)_e        (Mindlin facet in its panel frame)
      G(t) = sum_e t_e sigma_e (G_xx)_e       geometric stiffness of a prescribed bending pre-stress: the upper skin in
                                              compression, growing towards the root (an up-bending wing)
      (K + lam G) phi = 0,   design variables: one wall thickness per (panel, spanwise segment) group.

    Four element types (upper skin, right spar, lower skin, left spar) differ by the rotation of the panel frame; the
    device kernels take the type matrices and ``etype`` (no per-element matrices).  Everything here is vectorised numpy
    set-up; assembly, factorisation, eigensolve, adjoint and derivative run on the device.
    """

    def __init__(self, nx, nw, nh, h=0.0125, t0=0.02, nseg=4, E=1.0, nu=0.3, sigma0=1e-4, seed=0, drill=0.1,
                 shear_full=0.05):
        self.nx, self.nw, self.nh, self.h, self.nseg = nx, nw, nh, h, nseg
        nc = 2 * (nw + nh)
        self.nc = nc
        self.nnodes = (nx + 1) * nc
        self.nelems = nx * nc
        self.L, self.Wd, self.Hd = nx * h, nw * h, nh * h
        # perimeter: upper skin (y: 0 -> W at z = H), right spar (z: H -> 0 at y = W), lower skin, left spar
        s = np.arange(nc)
        y = np.where(s < nw, s * h, np.where(s < nw + nh, self.Wd, np.where(s < 2 * nw + nh, (2 * nw + nh - s) * h, 0.0)))
        z = np.where(s < nw, self.Hd, np.where(s < nw + nh, (nw + nh - s) * h,
                                                np.where(s < 2 * nw + nh, 0.0, (s - 2 * nw - nh) * h)))
        ii, jj = np.meshgrid(np.arange(nx + 1), s, indexing="ij")
        self.X = np.stack([ii.ravel() * h, y[jj.ravel()], z[jj.ravel()]], axis=1)
        ei, ej = np.meshgrid(np.arange(nx), s, indexing="ij")
        ei, ej = ei.ravel(), ej.ravel()
        node = lambda i, j: i * nc + (j % nc)  # noqa: E731
        self.conn = np.stack([node(ei, ej), node(ei + 1, ej), node(ei + 1, ej + 1), node(ei, ej + 1)], axis=1)
        self.etype = np.where(ej < nw, 0, np.where(ej < nw + nh, 1, np.where(ej < 2 * nw + nh, 2, 3))).astype(np.int32)
        self.group = (self.etype * nseg + np.minimum(ei * nseg // nx, nseg - 1)).astype(np.int64)
        self.ngroups = 4 * nseg
        rng = np.random.default_rng(seed)
        self.t = t0 * rng.uniform(0.8, 1.2, size=self.ngroups)
        # pre-stress of the reference load at the element centres: -sigma0 on the upper skin at the root
        zc = 0.25 * self.X[self.conn, 2].sum(axis=1)
        xc = 0.25 * self.X[self.conn, 0].sum(axis=1)
        self.sigma_e = -sigma0 * (1.0 - xc / self.L) * (zc - 0.5 * self.Hd) / (0.5 * self.Hd)
        # dofs: 6 per node, root section clamped
        nv = 6 * self.nnodes
        fixed = np.zeros(nv, dtype=bool)
        fixed[: 6 * nc] = True
        self.free_map = np.where(fixed, -1, np.cumsum(~fixed) - 1)
        self.reduced = np.flatnonzero(~fixed)
        self.n = len(self.reduced)
        full = (6 * self.conn[:, :, None] + np.arange(6)[None, None, :]).reshape(self.nelems, 24)
        self.elem_dofs = self.free_map[full].astype(np.int32)
        self._element_types(E, nu, drill, shear_full)

    def _element_types(self, E, nu, drill, shear_full):
        hx = hs = self.h
        C0 = plane_stress_C0(E, nu)
        Gs = (5.0 / 6.0) * E / (2.0 * (1.0 + nu))
        Kl = np.zeros((24, 24))   # thickness-linear part: membrane + transverse shear + drilling
        Kc = np.zeros((24, 24))   # thickness-cubic part: bending
        Gx = np.zeros((24, 24))   # unit axial membrane stress acting on the three translations
        iu, iv, iw, irx, iry, irz = (6 * np.arange(4) + c for c in range(6))

        def shape(xi, eta):
            N = 0.25 * np.array([(1 - xi) * (1 - eta), (1 + xi) * (1 - eta), (1 + xi) * (1 + eta), (1 - xi) * (1 + eta)])
            Nx = 0.25 * np.array([-(1 - eta), (1 - eta), (1 + eta), -(1 + eta)]) * (2.0 / hx)
            Ny = 0.25 * np.array([-(1 - xi), -(1 + xi), (1 + xi), (1 - xi)]) * (2.0 / hs)
            return N, Nx, Ny

        def shear_B(N, Nx, Ny):  # gamma = [w,x + theta_y ; w,y - theta_x]
            Bs = np.zeros((2, 24))
            Bs[0, iw], Bs[0, iry] = Nx, N
            Bs[1, iw], Bs[1, irx] = Ny, -N
            return Bs

        detJ = 0.25 * hx * hs
        for eta in GAUSS:
            for xi in GAUSS:
                N, Nx, Ny = shape(xi, eta)
                Bm = np.zeros((3, 24))
                Bm[0, iu], Bm[1, iv], Bm[2, iu], Bm[2, iv] = Nx, Ny, Ny, Nx
                Kl += detJ * Bm.T @ C0 @ Bm
                Bb = np.zeros((3, 24))  # curvatures of beta_x = theta_y, beta_y = -theta_x
                Bb[0, iry], Bb[1, irx], Bb[2, iry], Bb[2, irx] = Nx, -Ny, Ny, -Nx
                Kc += detJ * Bb.T @ (C0 / 12.0) @ Bb
                Bs = shear_B(N, Nx, Ny)
                Kl += shear_full * detJ * Gs * Bs.T @ Bs
                for idx in (iu, iv, iw):
                    Gx[np.ix_(idx, idx)] += detJ * np.outer(Nx, Nx)
        N, Nx, Ny = shape(0.0, 0.0)                      # transverse shear: one-point rule (no locking)
        Bs = shear_B(N, Nx, Ny)
        Kl += (1.0 - shear_full) * (4.0 * detJ) * Gs * Bs.T @ Bs
        Kl[irz, irz] += drill * E * hx * hs / 4.0        # drilling rotations: a small penalty keeps K definite
        frames = [np.array([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]]), np.array([[1.0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]]),
                  np.array([[1.0, 0, 0], [0, -1.0, 0], [0, 0, -1.0]]), np.array([[1.0, 0, 0], [0, 0, 1.0], [0, -1.0, 0]])]
        self.K_lin, self.K_cub, self.G_xx = [], [], []
        for T in frames:                                  # local = T global, for translations and rotations alike
            R = np.kron(np.eye(8), T)
            self.K_lin.append(R.T @ Kl @ R)
            self.K_cub.append(R.T @ Kc @ R)
            self.G_xx.append(R.T @ Gx @ R)
        self.K_lin, self.K_cub, self.G_xx = (np.array(a) for a in (self.K_lin, self.K_cub, self.G_xx))

    # ---- per-element factors of a thickness vector ---------------------------------------------------------
    def scales(self, t=None):
        te = (self.t if t is None else np.asarray(t))[self.group]
        return te, te**3, te * self.sigma_e

    def dof_coords(self):
        return np.repeat(self.X, 6, axis=0)[self.reduced]

    def assemble_host(self, t=None):
        """K, G as scipy CSR (COO assembly on the host: for the small test sizes)"""
        s1, s3, sg = self.scales(t)
        Ke = s1[:, None, None] * self.K_lin[self.etype] + s3[:, None, None] * self.K_cub[self.etype]
        Ge = sg[:, None, None] * self.G_xx[self.etype]
        ed = self.elem_dofs.astype(np.int64)
        rows = np.repeat(ed, 24, axis=1).ravel()
        cols = np.tile(ed, (1, 24)).ravel()
        keep = (rows >= 0) & (cols >= 0)
        out = []
        for Me in (Ke, Ge):
            A = sparse.coo_matrix((Me.ravel()[keep], (rows[keep], cols[keep])), shape=(self.n, self.n)).tocsr()
            A.sum_duplicates()
            A.sort_indices()
            out.append(A)
        return out

    def group_map(self):
        """ngroups x nelems incidence matrix: sums element quantities into the design groups"""
        return sparse.csr_matrix((np.ones(self.nelems), (self.group, np.arange(self.nelems))),
                                 shape=(self.ngroups, self.nelems))


class ShellBoxOnDevice:
    """
    The ShellBox design point on the device: typed element assembly of K(t), G(t) (eigd_assemble with one matrix per
    element type), the shifted factor K + sigma G, and the derivative callbacks d(w^T K v)/dt_g, d(w^T G v)/dt_g as
    element kernels summed into the thickness groups.
    """

    def __init__(self, box, ctx=None):
        from .device import CSRMatrix, ElementAssembler, default_context
        from .operators import SpLuOperator

        self.box, self.ctx = box, (ctx if ctx is not None else default_context())
        ctx = self.ctx
        self.asm = ElementAssembler(ctx, box.elem_dofs, box.n)
        pat = self.asm.pattern()
        ones = sparse.csr_matrix((np.ones(pat.nnz), pat.indices, pat.indptr), shape=pat.shape) \
            + sparse.identity(box.n, format="csr") * (10.0 * 24 * 9)
        ones.sort_indices()
        if ones.nnz != pat.nnz:
            raise ValueError("every dof needs a diagonal entry in the assembled pattern")
        self.factor = SpLuOperator(ones, ctx=ctx, check_symmetry=False, coords=box.dof_coords())
        self.dK, self.dG, self._shifted = CSRMatrix(ctx, ones), CSRMatrix(ctx, ones), CSRMatrix(ctx, ones)
        self.sigma = None

    def assemble(self, t=None):
        """K(t), G(t) values on the device (and in self.dK / self.dG)"""
        box, ctx = self.box, self.ctx
        s1, s3, sg = box.scales(t)
        vK = self.asm.assemble(box.K_lin, s1, etype=box.etype)
        vK3 = self.asm.assemble(box.K_cub, s3, etype=box.etype)
        vK.assign_lincomb([(1.0, vK), (1.0, vK3)])
        vG = self.asm.assemble(box.G_xx, sg, etype=box.etype)
        self.dK.update_values_device(vK)
        self.dG.update_values_device(vG)
        self.vK, self.vG = vK, vG
        return vK, vG

    def refactor(self, sigma):
        """numeric factorisation of K + sigma G from the device values; returns the number of negative pivots"""
        vS = self.ctx.empty(self.vK.n, 1).assign_lincomb([(1.0, self.vK), (float(sigma), self.vG)])
        self._shifted.update_values_device(vS)
        self.factor.refactor_device(vS, indefinite_matrix=self._shifted)
        self.sigma = float(sigma)
        if self.factor.static_pivots > 0:
            import warnings

            lo, hi = self.factor.negative_pivots_bounds
            warnings.warn(f"refactor: {self.factor.static_pivots} static pivots -- the number of eigenvalues below the shift is "
                          f"only known to lie in [{lo}, {hi}]")
        return self.factor.negative_pivots

    def callbacks(self, t=None):
        """(dAdx, dBdx) = d(w^T G v)/dt_g, d(w^T K v)/dt_g as device callbacks (A = G, B = K: buckling mode)"""
        from .device import ElementBilinear, GroupedElementDerivative

        box, ctx = self.box, self.ctx
        te = (box.t if t is None else np.asarray(t))[box.group]
        dK = [ElementBilinear(ctx, box.elem_dofs, box.K_lin, etype=box.etype),
              ElementBilinear(ctx, box.elem_dofs, box.K_cub, scale=3.0 * te**2, etype=box.etype)]
        dG = [ElementBilinear(ctx, box.elem_dofs, box.G_xx, scale=box.sigma_e, etype=box.etype)]
        return (GroupedElementDerivative(ctx, dG, box.group, box.ngroups),
                GroupedElementDerivative(ctx, dK, box.group, box.ngroups))
