import sys, warnings
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from conftest import load_golden, csr_from, relerr
import eigd_amd as eg
from oracle import eigd_oracle as orc
g = load_golden("g1_buckling50_basiclanczos")
K, G = csr_from(g, "K"), csr_from(g, "G")
sigma = float(g["sigma"])
fac = eg.SpLuOperator((K + sigma * G).tocsc())
fac_o = orc.SpLuOperator((K + sigma * G).tocsc())
for m in (12, 16, 20, 24, 30, 40, 60):
    s = eg.BasicLanczos(N=6, m=m, mode="buckling", tol=0.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lam, Phi = s.solve(G, K, fac, sigma)
        Qrb = g["Qrb"] * np.sign(np.einsum("ij,ij->j", Phi, g["Phi"]))
        psi_dl, _ = s.solve_adjoint(Qrb, method="dl")
        psi_s, _ = s.solve_adjoint(Qrb, method="sibk", rtol=1e-12)
        res_dl, _ = s.eval_adjoint_residual_norm(Qrb, psi_dl)
        psi_o, _ = orc.dl(Qrb, K, fac_o, sigma, lam, Phi, np.asarray(s.indices), np.asarray(s.V)[:, :len(s.theta)], np.asarray(s.T), np.asarray(s.Y), np.asarray(s.theta), mode="buckling")
    print(m, s.m, "eig_res", float(np.max(s.eig_res)), "res_dl", res_dl.max(), "dl vs sibk", relerr(psi_dl, psi_s), "gpu dl vs oracle dl", relerr(psi_dl, psi_o), "beta min", float(np.min(np.abs(np.diag(np.asarray(s.T), 1)))))
