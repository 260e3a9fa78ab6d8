"""
Q4 element matrices from a mesh (conn, X) -- host-side set-up for the device kernels of the derivative path.

The reference's harnesses evaluate every derivative callback with einsums over Gauss-point tables (Be, He, Te, detJ:
examples/fe_utils.py:19-97, 123-156).  On the device the same bilinear forms are contractions with assembled element
matrices (``eigd_elem_bilinear``), so this module turns the mesh into those matrices once:

  stiffness     Ke0[e] = sum_q detJ Be^T C0 Be          (examples/buckling.py:152-176, natural_frequency.py:134-160)
  mass          Me0[e] = sum_q detJ He^T He             (natural_frequency.py:205-236)
  conduction    Kt0[e] = sum_q detJ Bt^T Bt             (thermal.py:126-148)
  capacity      Mt0[e] = sum_q detJ N N^T               (thermal.py:192-214)
  stress tables L (12 x 8), Q (12 x 8 x 8) with Ge[e] = sum_m (L[m] . u_e) Q[m]   (buckling.py:220-255)

A mesh of identical (translated) elements is detected and its matrices are stored once.
"""

import numpy as np

GAUSS = (-1.0 / np.sqrt(3.0), 1.0 / np.sqrt(3.0))


def plane_stress_C0(E=1.0, nu=0.3):
    return (E / (1.0 - nu**2)) * np.array([[1.0, nu, 0.0], [nu, 1.0, 0.0], [0.0, 0.0, 0.5 * (1.0 - nu)]])


def _gauss_point(xi, eta, xe, ye):
    """shape functions, their physical derivatives (nelem x 4) and detJ (nelem) at one point of the reference square"""
    N = 0.25 * np.array([(1 - xi) * (1 - eta), (1 + xi) * (1 - eta), (1 + xi) * (1 + eta), (1 - xi) * (1 + eta)])
    dxi = 0.25 * np.array([-(1 - eta), (1 - eta), (1 + eta), -(1 + eta)])
    deta = 0.25 * np.array([-(1 - xi), -(1 + xi), (1 + xi), (1 - xi)])
    xxi, yxi, xeta, yeta = xe @ dxi, ye @ dxi, xe @ deta, ye @ deta
    detJ = xxi * yeta - xeta * yxi
    Nx = (yeta[:, None] * dxi[None, :] - yxi[:, None] * deta[None, :]) / detJ[:, None]
    Ny = (-xeta[:, None] * dxi[None, :] + xxi[:, None] * deta[None, :]) / detJ[:, None]
    return N, Nx, Ny, detJ


class Q4Elements:
    """element matrices of a Q4 mesh; ``uniform`` meshes (all elements congruent by translation) keep one copy"""

    def __init__(self, conn, X):
        self.conn = np.ascontiguousarray(conn, dtype=np.int64)
        self.X = np.asarray(X, dtype=float)
        self.nelems = self.conn.shape[0]
        self.nnodes = int(self.conn.max()) + 1
        xe, ye = self.X[self.conn, 0], self.X[self.conn, 1]
        rel = np.stack([xe - xe[:, :1], ye - ye[:, :1]], axis=-1)
        scale = max(np.abs(rel).max(), 1e-300)
        self.uniform = bool(np.abs(rel - rel[:1]).max() <= 1e-12 * scale)
        if self.uniform:
            xe, ye = xe[:1], ye[:1]
        self._pts = [_gauss_point(xi, eta, xe, ye) for eta in GAUSS for xi in GAUSS]

    def _shared(self, M):
        return M[0] if self.uniform else M

    @staticmethod
    def _strain(Nx, Ny):
        Be = np.zeros((Nx.shape[0], 3, 8))
        Be[:, 0, ::2] = Nx
        Be[:, 1, 1::2] = Ny
        Be[:, 2, ::2] = Ny
        Be[:, 2, 1::2] = Nx
        return Be

    def stiffness(self, C0):
        """(8, 8) or (nelem, 8, 8)"""
        Ke = 0.0
        for _, Nx, Ny, detJ in self._pts:
            Be = self._strain(Nx, Ny)
            Ke = Ke + detJ[:, None, None] * (Be.transpose(0, 2, 1) @ C0 @ Be)
        return self._shared(Ke)

    def mass(self):
        Me = 0.0
        for N, _, _, detJ in self._pts:
            He = np.zeros((2, 8))
            He[0, ::2] = N
            He[1, 1::2] = N
            Me = Me + detJ[:, None, None] * (He.T @ He)[None]
        return self._shared(Me)

    def conduction(self):
        Kt = 0.0
        for _, Nx, Ny, detJ in self._pts:
            Kt = Kt + detJ[:, None, None] * (np.einsum("ni,nj->nij", Nx, Nx) + np.einsum("ni,nj->nij", Ny, Ny))
        return self._shared(Kt)

    def capacity(self):
        Mt = 0.0
        for N, _, _, detJ in self._pts:
            Mt = Mt + detJ[:, None, None] * np.outer(N, N)[None]
        return self._shared(Mt)

    def stress_tables(self, C0):
        """L (12, 8), Q (12, 8, 8): one term per Gauss point and stress component; uniform meshes only"""
        if not self.uniform:
            raise ValueError("the shared stress-stiffness tables need a mesh of congruent elements")
        L, Q = [], []
        for _, Nx, Ny, detJ in self._pts:
            nx, ny, dj = Nx[0], Ny[0], float(detJ[0])
            CB = C0 @ self._strain(Nx, Ny)[0]
            for i, T in enumerate((np.outer(nx, nx), np.outer(ny, ny), np.outer(nx, ny) + np.outer(ny, nx))):
                G8 = np.zeros((8, 8))
                G8[0::2, 0::2] = dj * T
                G8[1::2, 1::2] = dj * T
                L.append(CB[i])
                Q.append(G8)
        return np.array(L), np.array(Q)

    def dofs2(self, free_map=None):
        """(nelem, 8) dof lists of the 2-dof/node problems; with ``free_map`` (full -> reduced, -1 = fixed) the reduced ones"""
        full = np.empty((self.nelems, 8), dtype=np.int64)
        full[:, ::2] = 2 * self.conn
        full[:, 1::2] = 2 * self.conn + 1
        return (full if free_map is None else np.asarray(free_map)[full]).astype(np.int32)
