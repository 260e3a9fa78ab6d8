#!/usr/bin/env python3
"""Development aid: print the actual GPU-vs-reference errors behind the psi gates of tests/test_gpu_path.py."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import align_signs, csr_from, load_golden, relerr  # noqa: E402

import eigd_amd as eg  # noqa: E402

warnings.simplefilter("ignore")


def colerr(a, b):
    return np.linalg.norm(a - b, axis=0) / np.linalg.norm(b, axis=0)


g = load_golden("g1_buckling50_basiclanczos")
K, G = csr_from(g, "K"), csr_from(g, "G")
sigma = float(g["sigma"])
fac = eg.SpLuOperator((K + sigma * G).tocsc())
s = eg.BasicLanczos(N=6, m=60, tol=0.0, mode="buckling")
lam, Phi = s.solve(G, K, fac, sigma)
Phi_a, sg = align_signs(Phi, g["Phi"])
print("g1 basic: lam", relerr(lam, g["lam"]), "Phi cols", colerr(Phi_a, g["Phi"]))
for rtol in (1e-10, 1e-12, 1e-14):
    psi, data = s.solve_adjoint(g["Qrb"] * sg, method="sibk", rtol=rtol, update_guess=False, bs_target=1)
    print(f"g1 basic own-eig psi rtol={rtol:g}:", relerr(psi * sg, g["psir"]), colerr(psi * sg, g["psir"]))
# how much of that is the eigenvector difference?  the reference's psi depends on its Phi (converged to its tol)
res = np.linalg.norm(K @ g["Phi"] + (G @ g["Phi"]) * g["lam"], axis=0) / np.linalg.norm(K @ g["Phi"], axis=0)
res_me = np.linalg.norm(K @ Phi + (G @ Phi) * lam, axis=0) / np.linalg.norm(K @ Phi, axis=0)
print("eigen residuals ref", res, "mine", res_me)

s2 = eg.IRAM(N=6, m=60, mode="buckling")
g2 = load_golden("g1_buckling50_iram")
lam2, Phi2 = s2.solve(G, K, fac, sigma)
P2, sg2 = align_signs(Phi2, g2["Phi"])
print("g1 iram: lam", relerr(lam2, g2["lam"]), "Phi cols", colerr(P2, g2["Phi"]))
psi, data = s2.solve_adjoint(g2["Qrb"] * sg2, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
print("g1 iram own-eig psi:", relerr(psi * sg2, g2["psir"]), colerr(psi * sg2, g2["psir"]))
resr = np.linalg.norm(K @ g2["Phi"] + (G @ g2["Phi"]) * g2["lam"], axis=0) / np.linalg.norm(K @ g2["Phi"], axis=0)
print("iram eigen residuals ref", resr)

g = load_golden("g2_natfreq32x16_basiclanczos")
K, M = csr_from(g, "K"), csr_from(g, "M")
sigma = float(g["sigma"])
fac = eg.SpLuOperator((K - sigma * M).tocsc())
s = eg.BasicLanczos(N=13, m=60, tol=1e-14)
s.solve(K, M, fac, sigma)
s.lam0 = g["lam"].copy(); s.Phi = g["Phi"].copy(); s.m = s._m = int(g["m"]); s.V = g["V"]
s.Y, s.theta, s.indices, s.T = g["Y"].copy(), g["theta"].copy(), g["indices"].copy(), g["T"].copy()
for rtol in (1e-10, 1e-12):
    psi0, data = s.solve_adjoint(g["Q0b"], method="sibk", rtol=rtol, update_guess=False, bs_target=1)
    print(f"g2 ref-eig psi rtol={rtol:g}:", relerr(psi0[:, 3:], g["psi"]), colerr(psi0[:, 3:], g["psi"]))
print("g2 reference adjoint residuals (b_ortho):", g["res_bortho"])
r_me, _ = s.eval_adjoint_residual_norm(g["Q0b"], psi0, b_ortho=True)
print("g2 my residuals:", r_me)

g = load_golden("g4_laplace900_basiclanczos")
K, M = csr_from(g, "K"), csr_from(g, "M")
for mode in ("normal", "buckling"):
    A, B = (K, M) if mode == "normal" else ((-0.005 * M).tocsr(), K)
    p = mode + "_"
    sigma = float(g[p + "sigma"])
    fac = eg.SpLuOperator(((A - sigma * B) if mode == "normal" else (B + sigma * A)).tocsc())
    s = eg.BasicLanczos(N=6, m=60, mode=mode)
    s.solve(A, B, fac, sigma)
    s.lam0 = g[p + "lam"].copy(); s.Phi = g[p + "Phi"].copy(); s.m = s._m = int(g[p + "m"]); s.V = g[p + "V"]
    s.Y, s.theta, s.indices, s.T = g[p + "Y"].copy(), g[p + "theta"].copy(), g[p + "indices"].copy(), g[p + "T"].copy()
    for method in ("pcpg", "sibk", "pgmres"):
        psi, data = s.solve_adjoint(g["Phib"].copy(), method=method, rtol=1e-12)
        print(mode, method, "psi err", relerr(psi, g[p + method + "_psi"]), "ref residual", g[p + method + "_res"].max())
    gi = load_golden("g4_laplace900_iram")
    si = eg.IRAM(N=6, m=40, mode=mode)
    lam, Phi = si.solve(A, B, fac, sigma)
    Pa, sgi = align_signs(Phi, gi[p + "Phi"])
    psi, data = si.solve_adjoint(gi["Phib"] * sgi, method="sibk", rtol=1e-12, update_guess=False, bs_target=1)
    print(mode, "iram sibk own-eig psi err", relerr(psi * sgi, gi[p + "sibk_psi"]), "Phi", colerr(Pa, gi[p + "Phi"]))
