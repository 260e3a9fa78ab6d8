#!/usr/bin/env python3
"""Development aid: accuracy of factor / eigen-solvers on the small shell box against SuperLU and scipy."""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import eigd_amd as eg
from eigd_amd.problems import ShellBox, ShellBoxOnDevice
from scipy.sparse.linalg import splu
warnings.simplefilter("ignore")
box = ShellBox(40, 12, 4, nseg=3, seed=1)
dev = ShellBoxOnDevice(box)
ctx = dev.ctx
dev.assemble()
K, G = box.assemble_host()
sigma = 389.0
print("negative pivots", dev.refactor(sigma))
mat = (K + sigma * G).tocsc()
lu = splu(mat)
rng = np.random.default_rng(0)
b = rng.normal(size=(box.n, 3))
x = dev.factor(b)
print("factor residual", np.linalg.norm(mat @ x - b) / np.linalg.norm(b), "vs superlu", np.linalg.norm(lu.solve(b) - x) / np.linalg.norm(x))
y = dev.dK.matvec(b[:, 0]); print("K spmv err", np.linalg.norm(y - K @ b[:, 0]) / np.linalg.norm(y))
y = dev.dG.matvec(b[:, 0]); print("G spmv err", np.linalg.norm(y - G @ b[:, 0]) / np.linalg.norm(y))
Y = dev.dK.matmat(b); print("K spmm err", np.linalg.norm(Y - K @ b) / np.linalg.norm(Y))
for name, s in (("basic", eg.BasicLanczos(N=8, m=120, tol=1e-13, mode="buckling", ctx=ctx)),
                ("iram40", eg.IRAM(N=8, m=40, mode="buckling", ctx=ctx)), ("iram80", eg.IRAM(N=8, m=80, mode="buckling", ctx=ctx))):
    lam, Phi = s.solve(dev.dG, dev.dK, dev.factor, sigma)
    R = K @ Phi + (G @ Phi) * lam
    print(name, getattr(s, "m", None), getattr(s, "n_restarts", None), lam, "res", np.linalg.norm(R, axis=0) / np.linalg.norm(K @ Phi, axis=0))
# the same with host scipy matrices handed in (re-uploaded CSR)
fac2 = eg.SpLuOperator(mat, ctx=ctx)
s = eg.IRAM(N=8, m=40, mode="buckling", ctx=ctx)
lam, Phi = s.solve(G, K, fac2, sigma)
print("iram40 host-matrices", lam)
x2 = fac2(b)
print("fac2 residual", np.linalg.norm(mat @ x2 - b) / np.linalg.norm(b))
