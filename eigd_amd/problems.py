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
