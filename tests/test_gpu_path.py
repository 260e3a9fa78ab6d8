"""
GPU parity of the hot path (eigensolve -> adjoint solves -> correction -> total derivative)
against the golden vectors captured from the reference and against the CPU oracle.
Tolerances: eigenvalues / derivatives 1e-8 relative (north_star), index sets bit-exact.
"""
import warnings

import numpy as np
import pytest

from conftest import align_signs, corr_from, csr_from, index_sets, load_golden, relerr

pytestmark = pytest.mark.gpu

RTOL = 1e-8


def _mock_cb(C):
    def cb(w, v):
        if w.ndim == 1:
            return C.T @ (w * v)
        return C.T @ np.sum(w * v, axis=1)
    return cb


def _shift(A, B, sigma, mode):
    return (A - sigma * B) if mode == "normal" else (B + sigma * A)


def _solve_basic(g, A, B, sigma, mode, prefix="", **kw):
    import eigd_amd as eg

    factor = eg.SpLuOperator(_shift(A, B, sigma, mode).tocsc())
    s = eg.BasicLanczos(mode=mode, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lam, Phi = s.solve(A, B, factor, sigma)
    assert s.N == int(g[prefix + "N"])
    # mode ordering: identical index sequence, except inside clusters of numerically equal Ritz values
    # (e.g. the three rigid-body modes at lam ~ 1e-15), whose internal order is decided by round-off
    mine, ref = s.indices[: s.N], g[prefix + "indices"][: s.N]
    lam_ref = g[prefix + "lam"]
    start = 0
    for e in range(1, s.N + 1):
        if e == s.N or abs(lam_ref[e] - lam_ref[e - 1]) > 1e-9 * max(1.0, abs(lam_ref[e])):
            assert sorted(mine[start:e]) == sorted(ref[start:e])
            if e - start == 1:
                assert mine[start] == ref[start]
            start = e
    assert relerr(lam, g[prefix + "lam"]) < RTOL
    assert relerr(s.alpha[:5], g[prefix + "alpha"][:5]) < 1e-9
    assert relerr(s.beta[:5], g[prefix + "beta"][:5]) < 1e-9
    Phi_a, sg = align_signs(Phi, g[prefix + "Phi"])
    return s, factor, Phi_a, sg


def _adopt_reference_lanczos(s, g, prefix=""):
    """run the adjoint stage from the reference's own (Phi, V, Y, theta, indices, T)"""
    m = int(g[prefix + "m"])
    if hasattr(s, "lam0"):
        s.lam0 = g[prefix + "lam"].copy()
    else:
        s.lam = g[prefix + "lam"].copy()
    s.Phi = g[prefix + "Phi"].copy()
    s.m = s._m = m
    s.V = g[prefix + "V"]
    s.Y, s.theta = g[prefix + "Y"].copy(), g[prefix + "theta"].copy()
    s.indices, s.T = g[prefix + "indices"].copy(), g[prefix + "T"].copy()


def test_g1_buckling_basiclanczos_full_chain():
    g = load_golden("g1_buckling50_basiclanczos")
    K, G = csr_from(g, "K"), csr_from(g, "G")
    s, factor, Phi_a, sg = _solve_basic(g, G, K, float(g["sigma"]), "buckling", N=6, m=60, tol=0.0)
    assert relerr(Phi_a, g["Phi"]) < 1e-6
    # B-orthonormality of the computed eigenvectors
    assert np.linalg.norm(s.Phi.T @ (K @ s.Phi) - np.eye(6)) < 1e-10
    # own Lanczos data: converged psi agrees with the reference up to the eigenvector signs
    Qrb = g["Qrb"] * sg
    factor.count = 0
    res_hist = []
    psi, data = s.solve_adjoint(Qrb, method="sibk", rtol=1e-10, update_guess=False, bs_target=1,
                                callback=res_hist.append)
    assert index_sets(data) == index_sets(corr_from(g, "corr"))
    assert relerr(psi * sg, g["psir"]) < RTOL          # (measured 6e-14 .. 5e-11: tools/tol_probe.py)
    assert abs(factor.count - int(g["count_adjoint"])) <= 12  # applications per mode; reported, loosely gated
    assert len(res_hist) > 0
    res, ortho = s.eval_adjoint_residual_norm(Qrb, psi, b_ortho=False)
    assert res.max() < 1e-8 * np.linalg.norm(Qrb)
    # reference's Lanczos data: tight parity on psi
    _adopt_reference_lanczos(s, g)
    psi2, data2 = s.solve_adjoint(g["Qrb"], method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
    assert index_sets(data2) == index_sets(corr_from(g, "corr"))
    assert relerr(psi2, g["psir"]) < RTOL


@pytest.mark.parametrize("name", ["g3_thermal32_eps1e-1_basiclanczos", "g3_thermal32_eps1e-8_basiclanczos"])
def test_g3_repeated_eigenvalue_index_sets(name):
    g = load_golden(name)
    K, M = csr_from(g, "K"), csr_from(g, "M")
    s, factor, Phi_a, sg = _solve_basic(g, K, M, float(g["sigma"]), "normal", N=8, m=60, tol=0.0)
    _adopt_reference_lanczos(s, g)
    psi, data = s.solve_adjoint(g["Qb"], method="sibk", rtol=1e-12, update_guess=False, bs_target=1)
    ref = corr_from(g, "corr")
    assert index_sets(data) == index_sets(ref)                     # bit-exact index sets
    if "eps1e-8" in name:
        assert index_sets(data) == {1: [2], 2: [1], 4: [5], 5: [4], 6: [7], 7: [6]}
    # xi, eta divide the difference of two n-term dot products by the gap of a numerically repeated pair (1e-7 here).
    # The device forms those entries with compensated dot products and xi, eta in extended precision: held against the
    # EXACT rational value computed from the same inputs (a plain double dot product misses this gate by 10-100x), and
    # against the reference's own floating-point value within the rounding of ITS dot products.
    from fractions import Fraction

    def exact_dot(x, y):
        return sum(Fraction(a) * Fraction(b) for a, b in zip(x.tolist(), y.tolist()))

    lam, Phi_r, Qb = g["lam"], g["Phi"], g["Qb"]
    eps = np.finfo(float).eps
    for i in ref:
        for (j, xi, eta), (jr, xir, etar) in zip(data[i], ref[i]):
            gap = lam[j] - lam[i]
            gji, gij = -exact_dot(Phi_r[:, j], Qb[:, i]), -exact_dot(Phi_r[:, i], Qb[:, j])
            xi_x = float(Fraction(1, 2) * (gji - gij) / Fraction(gap))
            eta_x = float(Fraction(1, 2) * (Fraction(lam[i]) * gji - Fraction(lam[j]) * gij) / Fraction(gap))
            norms = np.linalg.norm(Phi_r[:, j]) * np.linalg.norm(Qb[:, i]) + np.linalg.norm(Phi_r[:, i]) * np.linalg.norm(Qb[:, j])
            floor = 1e-2 * eps * norms / abs(gap)            # 100x below what one rounding of a plain dot product costs
            assert abs(xi - xi_x) <= 1e-10 * abs(xi_x) + floor
            assert abs(eta - eta_x) <= 1e-10 * abs(eta_x) + floor * max(1.0, abs(lam[i]))
            assert abs(xi - xir) * abs(gap) <= 64 * eps * norms     # the reference's value: inside its own dot-product rounding
    assert relerr(psi, g["psi"]) < RTOL
    res, ortho = s.eval_adjoint_residual_norm(g["Qb"], psi, b_ortho=True)
    assert np.allclose(res, g["res_bortho"], atol=1e-9)
    assert np.allclose(ortho, g["ortho_bortho"], atol=1e-9)


def test_g2_normal_mode_with_rigid_body_modes():
    g = load_golden("g2_natfreq32x16_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    s, factor, Phi_a, sg = _solve_basic(g, K, M, float(g["sigma"]), "normal", N=13, m=60, tol=1e-14)
    _adopt_reference_lanczos(s, g)
    psi0, data = s.solve_adjoint(g["Q0b"], method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
    assert index_sets(data) == index_sets(corr_from(g, "corr"))
    assert relerr(psi0[:, 3:], g["psi"]) < RTOL


@pytest.mark.parametrize("mode", ["normal", "buckling"])
def test_g4_method_matrix_basiclanczos(mode):
    g = load_golden("g4_laplace900_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    A, B = (K, M) if mode == "normal" else ((-0.005 * M).tocsr(), K)
    p = mode + "_"
    sigma = float(g[p + "sigma"])
    s, factor, Phi_a, sg = _solve_basic(g, A, B, sigma, mode, prefix=p, N=6, m=60)
    assert s.m == int(g[p + "m"])
    _adopt_reference_lanczos(s, g, p)
    for method, tol in (("laa", 1e-9), ("sibk", RTOL), ("pcpg", RTOL), ("pgmres", RTOL), ("dl", RTOL)):
        if method != "laa" and g[p + method + "_res"].max() > 1e-6:
            continue  # not converged in the reference itself: nothing to pin
        kw = {"update_guess": False, "bs_target": 1} if method == "sibk" else {}
        factor.count = 0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            psi, data = s.solve_adjoint(g["Phib"].copy(), method=method, rtol=1e-12, **kw)
        assert index_sets(data) == index_sets(corr_from(g, p + method + "_corr")), method
        assert relerr(psi, g[p + method + "_psi"]) < tol, method
        if method in ("laa", "dl"):
            assert factor.count == int(g[p + method + "_count"]), method
    psi, _ = s.solve_adjoint(g["Phib"].copy(), method="sibk", rtol=1e-12, bs_target=2)
    assert relerr(psi, g[p + "sibk_bs2_psi"]) < RTOL
    psi, _ = s.solve_adjoint(g["Phib"].copy(), method="sibk", rtol=1e-12, update_guess=True)
    assert relerr(psi, g[p + "sibk_ug_psi"]) < RTOL


@pytest.mark.parametrize("mode", ["normal", "buckling"])
def test_iram_eigenpairs_lanczos_relation_and_adjoint(mode):
    import eigd_amd as eg

    g = load_golden("g4_laplace900_iram")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    A, B = (K, M) if mode == "normal" else ((-0.005 * M).tocsr(), K)
    p = mode + "_"
    sigma = float(g[p + "sigma"])
    factor = eg.SpLuOperator(_shift(A, B, sigma, mode).tocsc())
    s = eg.IRAM(N=6, m=40, mode=mode)
    lam, Phi = s.solve(A, B, factor, sigma)
    assert relerr(lam, g[p + "lam"]) < RTOL
    Phi_a, sg = align_signs(Phi, g[p + "Phi"])
    assert relerr(Phi_a, g[p + "Phi"]) < 1e-6
    V, T = s.V, s.T
    assert V.shape == (K.shape[0], s.m) and T.shape == (s.m, s.m)
    lu = __import__("scipy.sparse.linalg", fromlist=["splu"]).splu(_shift(A, B, sigma, mode).tocsc())
    OPV = lu.solve(B @ V)
    Rm = OPV - V @ T
    assert np.linalg.norm(Rm[:, :-1]) < 1e-9 * np.linalg.norm(OPV)   # OP V = V T + f e_m^T
    assert np.linalg.norm(V.T @ (B @ V) - np.eye(s.m)) < 1e-10
    assert np.allclose(T, T.T)
    psi, data = s.solve_adjoint(g["Phib"] * sg, method="sibk", rtol=1e-12, update_guess=False, bs_target=1)
    assert relerr(psi * sg, g[p + "sibk_psi"]) < RTOL
    psi_l, _ = s.solve_adjoint(g["Phib"] * sg, method="laa")
    res, _ = s.eval_adjoint_residual_norm(g["Phib"] * sg, psi_l)
    assert np.all(np.isfinite(res))


def test_iram_g1_buckling_eigenvalues():
    import eigd_amd as eg

    g = load_golden("g1_buckling50_iram")
    K, G = csr_from(g, "K"), csr_from(g, "G")
    sigma = float(g["sigma"])
    factor = eg.SpLuOperator((K + sigma * G).tocsc())
    s = eg.IRAM(N=6, m=60, mode="buckling")
    lam, Phi = s.solve(G, K, factor, sigma)
    assert relerr(lam, g["lam"]) < RTOL
    assert s.m == 60
    Phi_a, sg = align_signs(Phi, g["Phi"])
    psi, data = s.solve_adjoint(g["Qrb"] * sg, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
    assert index_sets(data) == index_sets(corr_from(g, "corr"))
    assert relerr(psi * sg, g["psir"]) < RTOL


@pytest.mark.parametrize("tag", ["distinct", "repeated"])
@pytest.mark.parametrize("mode", ["normal", "buckling"])
def test_g5_correction_and_total_derivative_units(tag, mode):
    import eigd_amd as eg

    g = load_golden("g5_units")
    lam = g[tag + "_lam"]
    assert eg.are_eigenvalues_repeated(lam) == bool(g[tag + "_repeated"])
    p = f"{tag}_{mode}_"
    psi = g[p + "psi_in"].copy()
    data = eg.generate_adjoint_correction(lam, g["Phi"], psi, Phib=g["Phib"], mode=mode)
    ref = corr_from(g, p + "corr")
    assert index_sets(data) == index_sets(ref)
    for i in ref:
        for (j, xi, eta), (jr, xir, etar) in zip(data[i], ref[i]):
            assert abs(xi - xir) <= 1e-12 * max(1.0, abs(xir)) and abs(eta - etar) <= 1e-12 * max(1.0, abs(etar))
    assert relerr(psi, g[p + "psi_out"]) < 1e-13
    for dt in ("vector", "tensor"):
        dfdx = eg.add_eig_total_derivative(
            lam, g["Phi"], g["lamb"], g["Phib"], psi, _mock_cb(g["Ca"]), _mock_cb(g["Cb"]),
            np.zeros(g["Ca"].shape[1]), adj_corr_data=data, mode=mode, deriv_type=dt)
        assert relerr(dfdx, g[p + "dfdx_" + dt]) < 1e-12


def test_module_level_solvers_match_oracle():
    """functional API (sibk / pgmres / pcpg / laa / dl / residual) on a fresh small problem vs the CPU oracle"""
    import eigd_amd as eg
    from oracle import eigd_oracle as orc

    g = load_golden("g4_laplace900_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    sigma = -0.1
    p = "normal_"
    lam, Phi = g[p + "lam"], g[p + "Phi"]
    Phib = g["Phib"]
    fac_d = eg.SpLuOperator((K - sigma * M).tocsc())
    fac_o = orc.SpLuOperator((K - sigma * M).tocsc())
    for name in ("sibk", "pgmres", "pcpg"):
        kw = dict(sigma=sigma) if name == "sibk" else {}
        psi_d, data_d, info_d = getattr(eg, name)(Phib, K, M, lam, Phi, factor=fac_d, rtol=1e-12, **kw)
        psi_o, data_o, info_o = getattr(orc, name)(Phib, K, M, lam, Phi, factor=fac_o, rtol=1e-12, **kw)
        assert relerr(psi_d, psi_o) < RTOL, name
        assert index_sets(data_d) == index_sets(data_o)
    m = int(g[p + "m"])
    args = (Phib, M, None, sigma, lam, g[p + "V"], g[p + "Y"], g[p + "theta"], g[p + "indices"])
    for b_ortho in (False, True):
        a_d = eg.laa(args[0], args[1], fac_d, *args[3:], b_ortho=b_ortho)
        a_o = orc.laa(args[0], args[1], fac_o, *args[3:], b_ortho=b_ortho)
        assert relerr(a_d, a_o) < 1e-9
    psi_d, data_d = eg.dl(Phib, M, fac_d, sigma, lam, Phi, g[p + "indices"], g[p + "V"], g[p + "T"], g[p + "Y"], g[p + "theta"])
    psi_o, data_o = orc.dl(Phib, M, fac_o, sigma, lam, Phi, g[p + "indices"], g[p + "V"], g[p + "T"], g[p + "Y"], g[p + "theta"])
    assert relerr(psi_d, psi_o) < RTOL
    for b_ortho in (False, True):
        r_d, o_d = eg.eval_adjoint_residual_norm(K, M, lam, Phi, Phib, psi_o, b_ortho=b_ortho)
        r_o, o_o = orc.eval_adjoint_residual_norm(K, M, lam, Phi, Phib, psi_o, b_ortho=b_ortho)
        assert np.allclose(r_d, r_o, atol=1e-9) and np.allclose(o_d, o_o, atol=1e-9)
    # default factor path (factor=None builds 0.9 lam_0 shift, ref 1160-1167)
    psi_d, _, _ = eg.sibk(Phib, K, M, lam, Phi, rtol=1e-12)
    psi_o, _, _ = orc.sibk(Phib, K, M, lam, Phi, rtol=1e-12)
    assert relerr(psi_d, psi_o) < RTOL


def test_error_behaviour_matches_reference():
    import eigd_amd as eg

    g = load_golden("g4_laplace900_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    with pytest.raises(ValueError):
        eg.BasicLanczos(mode="nope")
    with pytest.raises(ValueError):
        eg.BasicLanczos(ortho_type="nope")
    with pytest.raises(ValueError):
        eg.BasicLanczos(Ntarget=2.5)
    with pytest.raises(ValueError):
        eg.IRAM(mode="nope")
    assert eg.IRAM(N=4).m == 20 and eg.IRAM(N=30, m=40).m == 61
    fac = eg.SpLuOperator((K + 0.1 * M).tocsc())
    s = eg.BasicLanczos(N=4, m=30)
    s.solve(K, M, fac, -0.1)
    n = K.shape[0]
    with pytest.raises(ValueError):
        s.solve_adjoint(np.zeros((n, 4)), method="shift-invert")
    with pytest.raises(ValueError):
        s.solve_adjoint(np.zeros((n, 3)))
    with pytest.raises(ValueError):
        eg.sibk(np.zeros((n, 4)), K, M, np.zeros(3), np.zeros((n, 4)), factor=fac, sigma=-0.1)
    with pytest.raises(ValueError):
        eg.add_eig_total_derivative(np.zeros(4), np.zeros((n, 4)), np.zeros(4), np.zeros((n, 4)), np.zeros((n, 3)),
                                    None, None, np.zeros(2))
    # count bookkeeping and host call surface of SpLuOperator (ref 18-23)
    fac.count = 0
    x = np.random.default_rng(0).normal(size=n)
    y = fac(x)
    assert y.shape == (n,) and fac.count == 1
    Y = fac(np.random.default_rng(0).normal(size=(n, 5)))
    assert Y.shape == (n, 5) and fac.count == 6
    assert np.linalg.norm((K + 0.1 * M) @ y - x) < 1e-11 * np.linalg.norm(x)
    with pytest.raises(ValueError):
        eg.SpLuOperator((K + sparsify_asym(K)).tocsc())


def sparsify_asym(K):
    from scipy import sparse

    n = K.shape[0]
    return sparse.coo_matrix(([1.0], ([0], [n - 1])), shape=(n, n)).tocsr()


def test_selective_orthogonalisation_and_ntarget():
    import eigd_amd as eg
    from oracle import eigd_oracle as orc

    g = load_golden("g3_thermal32_eps1e-8_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    sigma = float(g["sigma"])
    fd = eg.SpLuOperator((K - sigma * M).tocsc())
    fo = orc.SpLuOperator((K - sigma * M).tocsc())
    sd = eg.BasicLanczos(N=5, m=60, tol=1e-12, ortho_type="selective")
    so = orc.BasicLanczos(N=5, m=60, tol=1e-12, ortho_type="selective")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ld, _ = sd.solve(K, M, fd, sigma)
        lo, _ = so.solve(K, M, fo, sigma)
    assert relerr(ld, lo) < 1e-7
    # Ntarget widens N over the repeated pair {1,2} (ref 1615-1625)
    sd = eg.BasicLanczos(Ntarget=2, m=60, tol=1e-12)
    so = orc.BasicLanczos(Ntarget=2, m=60, tol=1e-12)
    sd.solve(K, M, fd, sigma)
    so.solve(K, M, fo, sigma)
    assert sd.N == so.N == 3


def test_mode_sharding_single_process_equivalence():
    """rank-by-rank replay of the sharded path reproduces the unsharded psi and df/dx to rounding"""
    import eigd_amd as eg

    g = load_golden("g4_laplace900_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    fac = eg.SpLuOperator((K + 0.1 * M).tocsc())
    s = eg.BasicLanczos(N=6, m=60)
    s.solve(K, M, fac, -0.1)
    Phib, lamb = g["Phib"], g["lamb"]
    rng = np.random.default_rng(2)
    Ca, Cb = rng.normal(size=(K.shape[0], 9)), rng.normal(size=(K.shape[0], 9))
    psi, data = s.solve_adjoint(Phib, method="sibk", rtol=1e-12)
    dfdx = s.add_total_derivative(lamb, Phib, psi, _mock_cb(Ca), _mock_cb(Cb), np.zeros(9), adj_corr_data=data,
                                  deriv_type="tensor")

    class FakeComm:
        def __init__(self, rank, size, acc):
            self.rank, self.size, self.acc = rank, size, acc

        def allreduce_sum(self, a):
            self.acc.append(np.array(a))
            return a

    P = 3
    psi_sh = np.zeros_like(psi)
    parts = []
    for r in range(P):
        comm = FakeComm(r, P, [])
        psi_r, data_r = s.solve_adjoint(Phib, method="sibk", rtol=1e-12, comm=comm)
        assert index_sets(data_r) == index_sets(data)
        cols = np.arange(r, 6, P)
        assert np.all(psi_r[:, [c for c in range(6) if c not in cols]] == 0.0)
        psi_sh[:, cols] = psi_r[:, cols]
        s.add_total_derivative(lamb, Phib, psi_r, _mock_cb(Ca), _mock_cb(Cb), np.zeros(9), adj_corr_data=data_r,
                               deriv_type="tensor", comm=comm)
        parts.append(comm.acc[-1])
    assert relerr(psi_sh, psi) < 1e-12   # only the reduction trees depend on the block width
    assert relerr(np.sum(parts, axis=0), dfdx) < 1e-13


_RANK_WORKER = '''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["EIGD_ROOT"]); sys.path.insert(0, os.path.join(os.environ["EIGD_ROOT"], "tests"))
import torch, torch.distributed as dist      # torch first: one HIP runtime for torch and libeigd_hip.so
backend = os.environ["EIGD_TEST_BACKEND"]
dist.init_process_group("gloo")
os.environ["EIGD_DEVICE"] = "0"
import eigd_amd as eg
from eigd_amd.comm import TorchDistComm
from conftest import csr_from, load_golden
comm = TorchDistComm(device="cpu")
g = load_golden("g4_laplace900_basiclanczos")
K, M = csr_from(g, "K"), csr_from(g, "M")
fac = eg.SpLuOperator((K + 0.1 * M).tocsc())
s = eg.BasicLanczos(N=6, m=60)
s.solve(K, M, fac, -0.1)
rng = np.random.default_rng(2)
Ca, Cb = rng.normal(size=(K.shape[0], 9)), rng.normal(size=(K.shape[0], 9))
cb = lambda C: (lambda w, v: C.T @ np.sum(w * v, axis=1))
Phib, lamb = g["Phib"], g["lamb"]
psi, data = s.solve_adjoint(Phib, method="sibk", rtol=1e-12)
ref = s.add_total_derivative(lamb, Phib, psi, cb(Ca), cb(Cb), np.zeros(9), adj_corr_data=data, deriv_type="tensor")
for method in ("sibk", "pgmres", "pcpg"):
    psi_r, data_r = s.solve_adjoint(Phib, method=method, rtol=1e-12, comm=comm)
    out = s.add_total_derivative(lamb, Phib, psi_r, cb(Ca), cb(Cb), np.zeros(9), adj_corr_data=data_r,
                                 deriv_type="tensor", comm=comm)
    err = np.linalg.norm(out - ref) / np.linalg.norm(ref)
    assert err < 1e-8, (method, err)
    mine = np.arange(comm.rank, 6, comm.size)
    others = [c for c in range(6) if c not in mine]
    assert np.all(psi_r[:, others] == 0.0)
dist.destroy_process_group()
print("rank", comm.rank, "ok")
'''


def _run_ranks(tmp_path, nproc, backend, port):
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank_worker.py"
    script.write_text(_RANK_WORKER)
    env = dict(os.environ, EIGD_ROOT=root, EIGD_TEST_BACKEND=backend, MASTER_ADDR="127.0.0.1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("ok") == nproc


def test_two_rank_gpu_sharding_gloo(tmp_path):
    """two processes share the GPU, modes sharded 3 + 3, df/dx all-reduced over gloo"""
    _run_ranks(tmp_path, 2, "gloo", 29531)


def test_rccl_allreduce_through_the_c_abi(tmp_path):
    """
    eigd_comm_unique_id / eigd_comm_init / eigd_allreduce_sum / eigd_allreduce_max (include/eigd_hip.h) on the ranks this
    box offers: librccl is bound at run time, a communicator is created from the 128-byte id exactly as the rank
    processes of bench.py do (file rendezvous), the df/dx reduction runs in place on a device vector.  With one GPU
    this is a world of one rank (RCCL refuses two ranks on the same device); the sum over more ranks is the same call.
    """
    import ctypes as C
    import os

    import eigd_amd as eg
    from eigd_amd import _ffi
    from eigd_amd.comm import RcclComm, exchange_unique_id
    from eigd_amd.device import default_context

    ctx = default_context()
    buf = C.create_string_buffer(128)
    _ffi.call("eigd_comm_unique_id", buf)                       # loads librccl, asks it for an id
    assert any(b != 0 for b in buf.raw)
    os.environ["EIGD_COMM_DIR"] = str(tmp_path)
    try:
        uid = exchange_unique_id(0, 1, lambda: buf.raw, tag="t")
        assert uid == buf.raw and (tmp_path / "t.bin").read_bytes() == buf.raw
        h = _ffi.c_vp()
        _ffi.call("eigd_comm_init", ctx.h, 1, 0, uid, C.byref(h))   # a real ncclCommInitRank when nranks > 1
        nr, rk = C.c_int(), C.c_int()
        _ffi.call("eigd_comm_info", h, C.byref(nr), C.byref(rk))
        assert (nr.value, rk.value) == (1, 0)
        _ffi.lib().eigd_comm_destroy(h)
        with pytest.raises(ValueError):
            _ffi.call("eigd_comm_init", ctx.h, 2, 5, uid, C.byref(h))
        comm = RcclComm(ctx, rank=0, size=1)
    finally:
        del os.environ["EIGD_COMM_DIR"]
    x = np.random.default_rng(0).normal(size=1000)
    d = ctx.from_host(x)
    assert comm.allreduce_sum(d) is d and np.array_equal(d.get()[:, 0], x)
    assert np.array_equal(comm.allreduce_sum(x), x) and comm.allreduce_max(3.5) == 3.5
    comm.barrier()
    # the sharded derivative with this communicator equals the unsharded one
    g = load_golden("g4_laplace900_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    fac = eg.SpLuOperator((K + 0.1 * M).tocsc())
    s = eg.BasicLanczos(N=6, m=60)
    s.solve(K, M, fac, -0.1)
    rng = np.random.default_rng(2)
    Ca, Cb = rng.normal(size=(K.shape[0], 9)), rng.normal(size=(K.shape[0], 9))
    psi, data = s.solve_adjoint(g["Phib"], method="sibk", rtol=1e-12)
    ref = s.add_total_derivative(g["lamb"], g["Phib"], psi, _mock_cb(Ca), _mock_cb(Cb), np.zeros(9), adj_corr_data=data,
                                 deriv_type="tensor")
    psi_r, data_r = s.solve_adjoint(g["Phib"], method="sibk", rtol=1e-12, comm=comm)
    out = s.add_total_derivative(g["lamb"], g["Phib"], psi_r, _mock_cb(Ca), _mock_cb(Cb), np.zeros(9), adj_corr_data=data_r,
                                 deriv_type="tensor", comm=comm)
    assert relerr(out, ref) < 1e-13
    comm.close()


def test_bench_launcher_starts_rank_processes(tmp_path):
    """`python bench.py --gpus N` starts its own rank processes before touching the GPU and relays rank 0's line"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-launch", "--nx", "60", "--ny", "60",
           "--modes", "4", "--m", "20", "--steps", "1", "--warmup", "0", "--cpu-sample", "none", "--no-fd-check",
           "--numpy-steps", "0", "--spmv-reps", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["modes"] == 4


_SMALL_BENCH = ["--nx", "60", "--ny", "60", "--modes", "4", "--m", "20", "--steps", "1", "--warmup", "0", "--cpu-sample",
                "none", "--no-fd-check", "--numpy-steps", "0", "--spmv-reps", "4"]


def test_bench_launcher_ends_when_a_rank_dies_before_the_communicator():
    """
    Two rank processes, rank 1 leaves before eigd_comm_init: rank 0 would wait in ncclCommInitRank for ever.  The
    launcher has to notice the dead rank, stop the other one and exit non-zero -- within seconds of the failure, not at
    some time limit -- and say which rank failed.
    """
    import os
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EIGD_TEST_FAIL_RANK="1", EIGD_LAUNCH_TIMEOUT="900")
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + _SMALL_BENCH, env=env,
                         capture_output=True, text=True, timeout=600)
    took = time.monotonic() - t0
    assert out.returncode != 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "rank 1 of 2 exited with code" in out.stderr, out.stderr[-3000:]
    assert "EIGD_TEST_FAIL_RANK" in out.stderr          # the dead rank's own message is relayed
    assert out.stdout.strip() == ""                     # no half-finished JSON line
    assert took < 240, f"launcher needed {took:.0f} s to give up"   # (interpreter + library start-up of the ranks, not a timeout)


def test_bench_two_ranks_over_rccl_match_one_rank(tmp_path):
    """
    bench.py --gpus 2 on a small problem: unique-id exchange between the rank processes, ncclCommInitRank, the in-place
    fp64 ncclAllReduce of the partial df/dx vectors and the sharded total derivative, against the one-rank df/dx.
    Needs two devices (RCCL refuses two ranks on one device): skipped on a one-GPU box.
    """
    import ctypes
    import json
    import os
    import subprocess
    import sys

    from eigd_amd._ffi import call

    cnt = ctypes.c_int()
    call("eigd_device_count", ctypes.byref(cnt))
    if cnt.value < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for g in (1, 2):
        dump = str(tmp_path / f"dfdx{g}.npy")
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(g), "--dump-dfdx", dump]
                             + _SMALL_BENCH, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        line = json.loads(out.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == g
        res[g] = np.load(dump)
    assert np.linalg.norm(res[2] - res[1]) <= 1e-10 * np.linalg.norm(res[1])


def _fd_functional(lam, Phi, lamb, Phib, Phi_ref, mode):
    """
    f = lamb . g(lam) + sum_i Phib_i . phi_i with the eigenvector signs aligned to the base point.
    g(lam) = lam in normal mode.  In buckling mode the reference's total-derivative formula
    (eigenvector_derivatives.py:118-134) weights the eigenvalue term as lamb_i (lam_i phi^T dA phi + phi^T dB phi)
    = lamb_i d(ln lam_i): its `lamb` is the sensitivity w.r.t. ln(lam) (a reference convention that its own
    tests never exercise -- the tanh aggregate of examples/buckling.py has lamb = 0).  We keep the reference
    formula (parity) and therefore check it against g(lam) = ln(lam).
    """
    sg = np.sign(np.einsum("ij,ij->j", Phi, Phi_ref))
    g = lam if mode == "normal" else np.log(lam)
    return float(lamb @ g + np.einsum("ij,ij->", Phib, Phi * sg))


@pytest.mark.parametrize("kind", ["buckling", "normal"])
def test_total_derivative_matches_central_difference(kind):
    """
    df/d(rho_E) from the adjoint path (device element callbacks) against a central finite difference of
    f(rho) = lamb . lam(rho) + sum_i Phib_i . phi_i(rho) along a random direction (examples/buckling.py:1025-1035
    does the same check with dh = 1e-4).  The fundamental path u is frozen for the buckling case, as in the
    callbacks.
    """
    import eigd_amd as eg
    from eigd_amd.device import ElementBilinear, default_context
    from eigd_amd.problems import BucklingColumn, FreePlate

    ctx = default_context()
    rng = np.random.default_rng(4)
    N = 5
    if kind == "buckling":
        prob = BucklingColumn(28, 28, seed=2)
        K0 = prob.stiffness()
        u = prob.full_vector(eg.SpLuOperator(K0)(prob.f[prob.reduced]))
        Ge_unit = prob.element_G(u)
        sigma, mode = None, "buckling"

        def matrices(rho):
            prob.rhoE = rho
            K = prob.stiffness()
            prob.Ge_unit = Ge_unit
            from eigd_amd.problems import _assemble
            G, _ = _assemble(prob.mesh, (rho**prob.p + prob.rho0_G)[:, None, None] * Ge_unit, 2, prob.free_map)
            return G, K   # (A, B)
    else:
        prob = FreePlate(24, 20, Lx=1.2, Ly=1.0, seed=3)
        sigma, mode = -10.0, "normal"

        def matrices(rho):
            prob.rhoE = rho
            return prob.stiffness(), prob.mass()   # (A, B)

    rho0 = prob.rhoE.copy()

    def solve(rho, sig):
        A, B = matrices(rho)
        mat = (A - sig * B) if mode == "normal" else (B + sig * A)
        fac = eg.SpLuOperator(mat.tocsc())
        nmodes = N + 3 if mode == "normal" else N
        s = eg.BasicLanczos(N=nmodes, m=80, tol=1e-13, mode=mode)
        s.solve(A, B, fac, sig)
        return s, A, B

    if sigma is None:  # shift below the first buckling load
        A, B = matrices(rho0)
        from scipy.sparse.linalg import eigsh
        mu = eigsh(-A, k=1, M=B, which="LA", return_eigenvectors=False)[0]
        sigma = 0.6 / mu
    s0, A0, B0 = solve(rho0, sigma)
    lam0 = np.asarray(s0._lamN())
    Nn = len(lam0)
    keep = np.arange(3, Nn) if mode == "normal" else np.arange(Nn)  # rigid-body modes carry no sensitivity weight
    Phib = np.zeros((A0.shape[0], Nn))
    Phib[:, keep] = rng.uniform(-1, 1, size=(A0.shape[0], len(keep)))
    lamb = np.zeros(Nn)
    lamb[keep] = rng.uniform(0.5, 1.5, size=len(keep))
    psi, data = s0.solve_adjoint(Phib, method="sibk", rtol=1e-13)
    ed = prob.elem_dofs
    if mode == "buckling":
        dAdx = ElementBilinear(ctx, ed, Ge_unit, scale=prob.p * rho0 ** (prob.p - 1.0))
        dBdx = ElementBilinear(ctx, ed, prob.Ke0, scale=prob.p * rho0 ** (prob.p - 1.0))
    else:
        dAdx = ElementBilinear(ctx, ed, prob.Ke0, scale=prob.p * rho0 ** (prob.p - 1.0))
        dBdx = ElementBilinear(ctx, ed, prob.Me0, scale=np.full(len(rho0), prob.density))
    dfdx = s0.add_total_derivative(lamb, Phib, psi, dAdx, dBdx, np.zeros(len(rho0)), adj_corr_data=data,
                                   deriv_type="tensor")
    pert = rng.uniform(size=len(rho0))
    h = 1e-5
    sp, _, _ = solve(rho0 + h * pert, sigma)
    sm, _, _ = solve(rho0 - h * pert, sigma)
    fp = _fd_functional(np.asarray(sp._lamN()), sp.Phi, lamb, Phib, s0.Phi, mode)
    fm = _fd_functional(np.asarray(sm._lamN()), sm.Phi, lamb, Phib, s0.Phi, mode)
    fd = (fp - fm) / (2 * h)
    ans = float(pert @ dfdx)
    assert abs(ans - fd) <= 2e-6 * max(abs(fd), 1e-12), (ans, fd)


def test_c2_natural_frequency_properties_at_full_size():
    """
    Natural-frequency configuration at its stated size (BASELINE configs[1]: 316 x 316 elements = 200 978 dof, 10 + 3
    rigid modes, m = 60): size-independent properties of the whole path -- eigen-residuals, B-orthonormality, the
    three rigid-body modes, adjoint residuals and orthogonality, and IRAM / BasicLanczos agreement.
    """
    import eigd_amd as eg
    from eigd_amd.problems import FreePlate

    prob = FreePlate(316, 316, seed=1)
    assert prob.n == 200978
    K, M = prob.stiffness(), prob.mass()
    sigma, N = -10.0, 13
    fac = eg.SpLuOperator((K - sigma * M).tocsc(), coords=prob.dof_coords())
    s = eg.IRAM(N=N, m=60)
    lam, Phi = s.solve(K, M, fac, sigma)
    assert np.all(np.abs(lam[:3]) < 1e-7) and lam[3] > 1e-3               # free-free plate: 3 rigid-body modes
    R = K @ Phi - (M @ Phi) * lam
    assert np.linalg.norm(R, axis=0).max() < 1e-8 * np.abs(K).max()
    assert np.linalg.norm(Phi.T @ (M @ Phi) - np.eye(N)) < 1e-9
    s2 = eg.BasicLanczos(N=N, m=80, tol=1e-13)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lam2, _ = s2.solve(K, M, fac, sigma)
    assert relerr(lam2[3:], lam[3:]) < 1e-9
    rng = np.random.default_rng(1)
    Phib = np.zeros((K.shape[0], N))
    Phib[:, 3:] = rng.uniform(size=(K.shape[0], N - 3))                  # harness convention: no weight on rigid modes
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        psi, data = s.solve_adjoint(Phib, method="sibk", rtol=1e-10)
    res, _ = s.eval_adjoint_residual_norm(Phib, psi, b_ortho=True)       # residual in the complement of span(B Phi)
    _, ortho = s.eval_adjoint_residual_norm(Phib, psi, b_ortho=False)    # |phi_i^T B psi_i| (own mode)
    rn0 = np.sqrt(np.max(np.sum(Phib**2, axis=0)))
    assert res[3:].max() < 1e-7 * rn0
    assert ortho[3:].max() < 1e-7 * np.abs(psi).max()
    assert set(data.keys()) <= {0, 1, 2}                                   # only the rigid-body cluster is repeated
    # the oracle (SuperLU + the reference's Arnoldi loop from a zero guess) at full size, for the first elastic mode and
    # the last one: psi to the 1e-8 of north_star
    from oracle import eigd_oracle as orc

    fac_o = orc.SpLuOperator((K - sigma * M).tocsc())
    for i in (3, N - 1):
        psi_o, _, info_o = orc.sibk(Phib, K, M, lam, Phi, mode="normal", sigma=sigma, factor=fac_o, rtol=1e-10, modes=[i])
        assert relerr(psi[:, i], psi_o[:, i]) < RTOL, (i, info_o, relerr(psi[:, i], psi_o[:, i]))


def test_c4_thermal_repeated_eigenvalues_at_full_size():
    """BASELINE configs[3] at its stated size: square thermal plate, 706 x 706 elements = 499 849 dof (exactly repeated
    pairs by symmetry; the 8 lowest of the configuration's 20 modes, m = 90: exactly repeated eigenvalues are only split
    by rounding in a single-vector Lanczos, in the reference as here): index sets, adjoint residuals, orthogonality"""
    import eigd_amd as eg
    from eigd_amd.problems import ThermalPlate

    prob = ThermalPlate(706, epsilon=0.0)
    assert prob.n == 499849
    K, M = prob.stiffness(), prob.mass()
    sigma, N = -0.1, 8
    fac = eg.SpLuOperator((K - sigma * M).tocsc(), coords=prob.dof_coords())
    s = eg.BasicLanczos(N=N, m=90, tol=1e-13)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lam, Phi = s.solve(K, M, fac, sigma)
    assert abs(lam[1] - lam[2]) < 1e-8 * lam[2] and abs(lam[4] - lam[5]) < 1e-8 * lam[5]
    assert np.linalg.norm(Phi.T @ (M @ Phi) - np.eye(N)) < 1e-9
    vec = np.random.default_rng(0).uniform(size=K.shape[0])
    Phib = np.zeros_like(Phi)
    for i in range(1, N):                                                   # thermal compliance (thermal.py:436-442)
        Phib[:, i] = 2.0 * (Phi[:, i] @ vec) * vec / lam[i]
    psi, data = s.solve_adjoint(Phib, method="sibk", rtol=1e-12)
    sets = index_sets(data)
    assert sets.get(1) == [2] and sets.get(2) == [1] and sets.get(4) == [5] and sets.get(5) == [4]
    res, _ = s.eval_adjoint_residual_norm(Phib, psi, b_ortho=True)
    _, ortho = s.eval_adjoint_residual_norm(Phib, psi, b_ortho=False)
    assert res.max() < 1e-7 * max(np.linalg.norm(Phib, axis=0).max(), 1.0)
    assert ortho.max() < 1e-7 * max(np.abs(psi).max(), 1.0)
    # the oracle at full size on the same eigenpairs and right-hand sides: a single mode (3) and one member of an exactly
    # repeated pair (4; its correction along the partner divides rounding by rounding in the reference's formulas
    # 373-383: the part of psi orthogonal to the pair is what both must agree on)
    from oracle import eigd_oracle as orc

    fac_o = orc.SpLuOperator((K - sigma * M).tocsc())
    BPhi = M @ Phi
    for i in (3, 4):
        psi_o, _, info_o = orc.sibk(Phib, K, M, lam, Phi, mode="normal", sigma=sigma, factor=fac_o, rtol=1e-12, modes=[i])
        a, b = psi[:, i] - Phi @ (BPhi.T @ psi[:, i]), psi_o[:, i] - Phi @ (BPhi.T @ psi_o[:, i])
        assert relerr(a, b) < RTOL, (i, info_o, relerr(a, b))


@pytest.mark.parametrize("variant", ["generalized", "standard"])
def test_c4_full_size_twenty_modes_and_the_derivative_through_the_repeated_branch(variant):
    """
    BASELINE configs[3] as SURVEY 8 sizes it: thermal plate, 706 x 706 elements = 499 849 dof, N = 20 modes, m = 90
    (examples/thermal.py:1662-1663), epsilon = 1e-8 -- the smallest of the reference's own test values (1657): pairs
    (i, j), (j, i) split by ~1e-7, "numerically repeated" by the reference's rule.  (epsilon = 0 makes xi of 373-383 a
    0 / 0 in the reference itself; its index sets at epsilon = 0 are covered by the test above.)  With the generalized
    problem of the reference (consistent capacity matrix, "generalized") and with B = I as BASELINE words it
    ("standard", K scaled so that the eigenvalues are O(1) as in the generalized case).  The block eigensolver finds both
    members of a pair together.  Checked: B-orthonormality and eigen-residuals of all 20 pairs, the numerically repeated
    pairs in the index sets, adjoint residuals, and the derivative of the thermal compliance (thermal.py:428-442,
    560-623) from the nodal design variables through the device callbacks against a central difference along a random
    direction.
    """
    from eigd_amd import design
    from eigd_amd.device import default_context
    from eigd_amd.problems import Q4Mesh

    ctx = default_context()
    mesh = Q4Mesh(706, 706, 1.0, 1.0 + 1e-8)
    N = 20
    std = variant == "standard"
    an = design.ModalAnalysis(mesh.conn, mesh.X, kind="thermal", fltr=None, N=N, m=90, sigma=-0.1, solver_type="IRAM",
                              rtol=1e-12, ctx=ctx, unit_mass=std, kappa=(706.0 ** 2 if std else 1.0))
    assert an.n == 499849
    x0 = np.full(an.n, 0.5)
    vec = np.random.default_rng(0).uniform(size=an.n)
    lam, Q = an.initialize(x0)
    s = an.eig_solver
    assert len(lam) == N and np.all(np.diff(lam) >= 0)
    KQ, MQ = an.dK.apply(s._prob.Phi).get(), an.dM.apply(s._prob.Phi).get()
    assert np.linalg.norm(KQ - MQ * lam, axis=0).max() < 1e-8 * np.linalg.norm(KQ, axis=0).max()
    assert np.abs(Q.T @ MQ - np.eye(N)).max() < 1e-9
    pairs = [(i, i + 1) for i in range(N - 1) if abs(lam[i + 1] - lam[i]) < 1e-5]      # the reference's rule (278-300)
    print(f"C4 {variant}: lam = {np.array2string(lam, precision=8)}", flush=True)
    # generalized: separable, every (i, j) / (j, i) pair coincides (8 among the 20 lowest).  B = I: not separable (the 1-D
    # stiffness and mass factors inside K do not commute at the boundary rows), only the pairs the symmetry of the square
    # protects (i + j odd, a two-dimensional irreducible representation) stay together: 5 among the 20 lowest; the
    # "accidental" ones ((2,0)/(0,2), (3,1)/(1,3), (4,0)/(0,4)) split by 4e-5 ... 2e-4 and are distinct by the reference's rule
    assert len(pairs) >= (5 if std else 8), lam
    assert all(abs(lam[b] - lam[a]) > 1e-12 * lam[b] for a, b in pairs)      # split by epsilon, not exactly repeated
    f0 = design.thermal_compliance(lam, Q, vec)
    Qb, lamb = design.thermal_compliance_seeds(lam, Q, vec)
    out = an.finalize_adjoint(Qb, lamb)
    sets = index_sets(out["corr_data"])
    for a, b in pairs:
        assert b in sets.get(a, []) and a in sets.get(b, [])
    res, _ = s.eval_adjoint_residual_norm(Qb, out["psi"], b_ortho=True)
    # 1e-7 of the right-hand sides, plus what the rounding of the eigenvectors themselves contributes.  psi_i contains
    # c_ji phi_j for the other modes, and op(phi_j) = (lam_j - lam_i) B phi_j + r_j with r_j the eigen-residual of pair j.
    # Phi = V Y (sums of m = 90 terms) carries 100 - 300 eps of relative rounding in its high-frequency content, which K
    # amplifies by |K| (1.3e6 per row in the B = I scaling, against eigenvalues of 1 - 20): |r_j| ~ 1e-7 whatever the
    # solver does, and the pair 4.91384 / 4.91388 -- 4e-5 apart, distinct by the reference's rule -- puts c ~ 300 on it.
    # Measured there over four summation orders of the eigensolver: 3e-5 ... 1.2e-4, the same to six digits for every
    # setting of the Krylov solver (one or two steps per pass, with and without extra pairs, measured projections).
    psi_h = out["psi"].get() if hasattr(out["psi"], "get") else np.asarray(out["psi"])
    sgn_v = ctx.from_host(np.random.default_rng(3).choice([-1.0, 1.0], size=(an.n, 1)))
    normK = float(np.linalg.norm(an.dK.apply(sgn_v).get())) / np.sqrt(an.n)
    gate = 1e-7 * max(np.linalg.norm(Qb, axis=0).max(), 1.0) + 2e3 * np.finfo(float).eps * normK * np.linalg.norm(psi_h, axis=0)
    assert np.all(res < gate), (res, gate)
    # directional derivative: central difference of the compliance along a smooth, unsymmetric direction (along a random
    # one the derivative is ~1e-2 of |f| / |x| -- the sum of 5e5 independent terms -- and the rounding of f, ~1e-10
    # relative, would be 1e-4 of the difference quotient)
    xx, yy = mesh.X[:, 0], mesh.X[:, 1]
    pert = 0.3 * np.sin(2.1 * xx + 0.4) * np.cos(1.3 * yy) + 0.2 * xx - 0.1 * yy * yy
    h = 1e-4
    fpm = []
    for sgn in (1.0, -1.0):
        lp, Qp = an.initialize(x0 + sgn * h * pert)
        fpm.append(design.thermal_compliance(lp, Qp, vec))
    fd = (fpm[0] - fpm[1]) / (2 * h)
    ans = float(out["xb"] @ pert)
    print(f"C4 {variant}: compliance {f0:.6e}, {len(pairs)} numerically repeated pairs among {N} modes (gaps "
          f"{min(lam[b] - lam[a] for a, b in pairs):.1e} ... {max(lam[b] - lam[a] for a, b in pairs):.1e}), directional "
          f"derivative adjoint {ans:.10e} vs central difference {fd:.10e}: rel-err {abs(ans - fd) / abs(fd):.2e}", flush=True)
    assert abs(ans - fd) < 5e-6 * abs(fd)


def test_foreign_factor_and_operators_are_honoured():
    """
    `factor`, A, B may be any LinearOperator (reference 1492-1497, 1933-1936): a foreign factor is applied by the
    caller's own code (block copied to the host and back), everything else stays on the device.
    """
    import eigd_amd as eg
    from scipy.sparse.linalg import LinearOperator, aslinearoperator, splu

    g = load_golden("g4_laplace900_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    sigma = -0.1
    lu = splu((K - sigma * M).tocsc())
    calls = {"n": 0}

    def mv(x):
        calls["n"] += 1
        return lu.solve(x)

    foreign = LinearOperator(K.shape, matvec=mv, matmat=mv, dtype=float)
    s = eg.BasicLanczos(N=6, m=60)
    lam, Phi = s.solve(aslinearoperator(K), M, foreign, sigma)   # A as a scipy MatrixLinearOperator, B sparse
    assert relerr(lam, g["normal_lam"]) < RTOL
    assert calls["n"] > 10
    psi, data = s.solve_adjoint(g["Phib"].copy(), method="sibk", rtol=1e-12)
    s.Phi, s.lam0 = g["normal_Phi"].copy(), g["normal_lam"].copy()
    _adopt_reference_lanczos(s, g, "normal_")
    psi, data = s.solve_adjoint(g["Phib"].copy(), method="sibk", rtol=1e-12)
    assert relerr(psi, g["normal_sibk_psi"]) < RTOL


def test_iram_wanted_ritz_values_spread_over_three_decades():
    """
    ADVICE r2: the noise floor of IRAM's convergence test scales with each Ritz value itself, not with the largest one.
    A pencil whose wanted eigenvalues spread over three decades around the shift (theta = 1 / (lam - sigma) from 2e2 down
    to 0.2): every returned pair against scipy's dense solution at 1e-8, residuals and eig_res small for the pairs far
    from the shift too.
    """
    import eigd_amd as eg
    from scipy import sparse
    from scipy.linalg import eigh as dense_eigh

    rng = np.random.default_rng(4)
    n, N = 400, 10
    # tridiagonal stiffness of a chain whose springs grow geometrically: eigenvalues spread over many decades
    kspr = np.geomspace(1e-3, 1e3, n + 1)
    K = sparse.diags([-kspr[1:-1], kspr[:-1] + kspr[1:], -kspr[1:-1]], [-1, 0, 1], format="csr")
    M = sparse.diags(rng.uniform(0.5, 1.5, size=n)).tocsr()
    lam_ref, Phi_ref = dense_eigh(K.toarray(), M.toarray())
    sigma = lam_ref[0] - 0.005 * (lam_ref[1] - lam_ref[0]) - 1e-9
    theta = 1.0 / (lam_ref[:N] - sigma)
    assert theta[0] / theta[-1] > 1e2
    fac = eg.SpLuOperator((K - sigma * M).tocsc())
    s = eg.IRAM(N=N, m=40)
    lam, Phi = s.solve(K, M, fac, sigma)
    assert relerr(lam, lam_ref[:N]) < 1e-8
    Phi_a, _ = align_signs(Phi, Phi_ref[:, :N])
    assert np.max(np.linalg.norm(Phi_a - Phi_ref[:, :N], axis=0) / np.linalg.norm(Phi_ref[:, :N], axis=0)) < 1e-6
    R = K @ Phi - (M @ Phi) * lam                       # backward error: |K| ~ 2e3 against eigenvalues of 4e-6 ... 3e-4
    assert np.linalg.norm(R, axis=0).max() < 1e-12 * abs(K).max() * np.linalg.norm(Phi, axis=0).max()
    assert np.all(s.eig_res <= 1e-10 * np.abs(s.theta[s.indices[:N]]))


def test_iram_reports_non_convergence():
    import eigd_amd as eg
    from scipy.sparse.linalg import ArpackNoConvergence

    g = load_golden("g4_laplace900_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    fac = eg.SpLuOperator((K + 0.1 * M).tocsc())
    with pytest.raises(ArpackNoConvergence):
        eg.IRAM(N=12, m=25, maxiter=0).solve(K, M, fac, -0.1)
    with pytest.raises(ValueError):
        eg.IRAM(N=6, m=2000).solve(K, M, fac, -0.1)   # ncv must not exceed n


def test_concurrent_mode_groups_on_streams_match_single_stream(monkeypatch):
    """streams=3: the modes are split into three groups on three HIP streams / host threads; same psi, data, counts
    (the mode groups run the Arnoldi form: the single-stream run is held to it here)"""
    import eigd_amd as eg

    monkeypatch.setattr(eg.tuning, "recurrence", "arnoldi")

    g = load_golden("g4_laplace900_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    fac = eg.SpLuOperator((K + 0.1 * M).tocsc())
    s = eg.BasicLanczos(N=6, m=60)
    s.solve(K, M, fac, -0.1)
    fac.count = 0
    hist1, hist3 = [], []
    psi1, data1 = s.solve_adjoint(g["Phib"], method="sibk", rtol=1e-12, callback=hist1.append, streams=1)
    c1 = fac.count
    fac.count = 0
    psi3, data3 = s.solve_adjoint(g["Phib"], method="sibk", rtol=1e-12, callback=hist3.append, streams=3)
    assert relerr(psi3, psi1) < 1e-11
    assert index_sets(data3) == index_sets(data1)
    assert fac.count == c1 and len(hist3) == len(hist1)
    assert np.allclose(hist3, hist1, rtol=1e-6, atol=1e-14)


def test_interior_shift_finds_eigenvalues_on_both_sides():
    """shift-invert around an interior shift (indefinite factor): the Ritz values nearest to sigma, residual-checked"""
    import eigd_amd as eg

    g = load_golden("g4_laplace900_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    lam_ref = g["normal_lam"]
    sigma = 0.5 * (lam_ref[2] + lam_ref[3])
    fac = eg.SpLuOperator((K - sigma * M).tocsc())
    assert fac.negative_pivots == 3
    s = eg.BasicLanczos(N=4, m=60, tol=1e-12)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lam, Phi = s.solve(K, M, fac, sigma)
    R = K @ Phi - (M @ Phi) * lam
    assert np.linalg.norm(R, axis=0).max() < 1e-8
    assert np.all(np.isin(np.round(lam, 8), np.round(lam_ref, 8)))


def test_eigsh_mod_function_returns_the_lanczos_relation():
    """the reference's eigsh_mod (arpack.py:104) as eigd calls it: d, z, Tm, v with OP v = v Tm + f e_m^T"""
    import eigd_amd as eg

    g = load_golden("g4_laplace900_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    sigma = -0.1
    fac = eg.SpLuOperator((K - sigma * M).tocsc())
    d, z, Tm, v = eg.eigsh_mod(K, k=6, M=M, sigma=sigma, which="LM", OPinv=fac, ncv=40, mode="normal")
    assert d.shape == (6,) and z.shape == (K.shape[0], 6) and Tm.shape == (40, 40) and v.shape == (K.shape[0], 40)
    assert np.allclose(d, g["normal_lam"][:6], rtol=1e-9)
    R = K @ z - (M @ z) * d
    assert np.linalg.norm(R, axis=0).max() < 1e-8
    assert np.abs(v.T @ (M @ v) - np.eye(40)).max() < 1e-10            # M-orthonormal basis
    OPv = np.column_stack([fac(M @ v[:, j]) for j in range(40)])
    E = OPv - v @ Tm
    assert np.abs(E[:, :-1]).max() < 1e-9 * np.abs(Tm).max()           # only the last column carries the residual f
    assert eg.eigsh_mod(K, k=6, M=M, sigma=sigma, OPinv=fac, ncv=40, return_eigenvectors=False).shape == (6,)
    with pytest.raises(ValueError):
        eg.eigsh_mod(K, k=6, M=M)                                       # not the shift-invert path


def test_g6_complex_step_point_on_the_device():
    """the reference's complex-step evaluation of C1 (complex K/G, complex SuperLU, BasicLanczos with its
    complex _eigh) in dual-number arithmetic on the device, against the reference's own numbers and the oracle"""
    import eigd_amd as eg
    from scipy.sparse.linalg import splu
    from test_oracle_golden import _complex_csr, check_complex_step_solution

    g = load_golden("g6_buckling50_complexstep")
    K, G = _complex_csr(g, "K"), _complex_csr(g, "G")
    sigma = float(g["sigma"])
    mat = (K + sigma * G).tocsc()
    fac = eg.SpLuOperator(mat)
    assert fac.dtype == np.complex128
    # the operator itself against complex SuperLU (reference 11-23): vector, block and real right-hand sides
    lu = splu(mat)
    rng = np.random.default_rng(0)
    b = rng.normal(size=mat.shape[0]) + 1e-20j * rng.normal(size=mat.shape[0])
    x, xr = fac(b), lu.solve(b)
    assert relerr(x.real, xr.real) < 1e-11 and relerr(x.imag, xr.imag) < 1e-9
    Bk = rng.normal(size=(mat.shape[0], 3))
    X, Xr = fac(Bk), lu.solve(Bk.astype(complex))
    assert relerr(X.real, Xr.real) < 1e-11 and relerr(X.imag, Xr.imag) < 1e-9
    from oracle import eigd_oracle as orc

    for ortho in ("full", "selective"):
        fac.count = 0
        tol = 0.0 if ortho == "full" else 1e-12  # (selective reorthogonalisation needs a tolerance to select by)
        s = eg.BasicLanczos(N=6, m=60, tol=tol, mode="buckling", ortho_type=ortho)
        lam, Phi = s.solve(G, K, fac, sigma)
        assert lam.dtype == np.complex128 and Phi.dtype == np.complex128 and fac.count == s.m
        if ortho == "selective":
            so = orc.BasicLanczos(N=6, m=60, tol=tol, mode="buckling", ortho_type=ortho)
            so.solve(G, K, orc.SpLuOperator(mat), sigma)
            assert abs(s.m - so.m) <= 1
        if ortho == "full":
            check_complex_step_solution(g, s)
            assert np.abs(s.alpha - g["alpha"]).max() < 1e-8 * np.abs(g["alpha"]).max()
            assert np.abs(s.V.T @ (K @ s.V[:, : s.m]) - np.eye(s.m + 1, s.m)).max() < 1e-9  # non-conjugated B-orthonormality
        else:
            assert np.abs(lam.real - g["lam"].real).max() < 1e-8 * np.abs(g["lam"].real).max()
            assert np.abs(lam.imag - g["lam"].imag).max() < 1e-6 * np.abs(g["lam"].imag).max()
    with pytest.raises(TypeError):
        eg.IRAM(N=6, mode="buckling").solve(G, K, fac, sigma)          # ARPACK is real only (scipy raises the same)


def test_c3_full_size_properties():
    """
    The benchmark configuration itself (BASELINE configs[2]: 706 x 706 Q4 column, 998 284 dof, 32 modes, IRAM m = 65),
    through size-independent properties: eigen-residuals and K-orthonormality, adjoint residuals, linearity of the adjoint
    solve in the right-hand side, mode sharding (a rank's columns are the columns of the full solve) and the
    total derivative against a central difference of the eigenvalue part.
    """
    import eigd_amd as eg
    from eigd_amd.device import ElementBilinear, default_context
    from eigd_amd.problems import BucklingColumn

    ctx = default_context()
    col = BucklingColumn(706, 706, Lx=1.0, Ly=1.0, seed=0)
    K = col.stiffness()
    n = K.shape[0]
    assert n == 998284
    Kfac = eg.SpLuOperator(K, ctx=ctx, check_symmetry=False, coords=col.dof_coords())
    u = col.full_vector(Kfac(col.f[col.reduced]))
    G = col.geometric_stiffness(u)
    sigma = 1.0                                                            # BLF_1 = 1.83 for this column
    fac = eg.SpLuOperator((K + sigma * G).tocsr(), ctx=ctx, symbolic=Kfac.symbolic, check_symmetry=False)
    assert fac.negative_pivots == 0
    del Kfac
    N = 32
    s = eg.IRAM(N=N, m=65, mode="buckling", ctx=ctx)
    lam, Phi = s.solve(G, K, fac, sigma)
    assert 1.5 < lam[0] < 2.2 and np.all(np.diff(lam) > 0)
    R = K @ Phi + (G @ Phi) * lam                                          # (K + lam G) phi = 0
    assert np.linalg.norm(R, axis=0).max() < 1e-9 * np.linalg.norm(K @ Phi, axis=0).max()
    assert np.abs(Phi.T @ (K @ Phi) - np.eye(N)).max() < 1e-10
    rng = np.random.default_rng(1)
    P1, P2 = rng.uniform(size=(n, N)), rng.uniform(size=(n, N))
    psi1, data1 = s.solve_adjoint(P1, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
    res, _ = s.eval_adjoint_residual_norm(P1, psi1, b_ortho=False)
    assert res.max() < 1e-8 * np.linalg.norm(P1, axis=0).max()
    psi2, _ = s.solve_adjoint(P2, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
    psi12, _ = s.solve_adjoint(2.0 * P1 - 0.5 * P2, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
    assert relerr(psi12, 2.0 * psi1 - 0.5 * psi2) < 1e-8

    class OneOfTwo:  # rank 1 of 2 without a second process
        rank, size = 1, 2

        def allreduce_sum(self, a):
            return a

        def allreduce_max(self, x):
            return x

        def barrier(self):
            pass

    psi_r, _ = s.solve_adjoint(P1, method="sibk", rtol=1e-10, update_guess=False, bs_target=1, comm=OneOfTwo())
    assert relerr(psi_r[:, 1::2], psi1[:, 1::2]) < 1e-9                    # (reduction-tree shapes differ with the width)
    # d(sum_i w_i ln lam_i)/d rhoE along a random direction against a central difference of the eigenvalues
    w = rng.uniform(size=N)
    dBdx = ElementBilinear(ctx, col.elem_dofs, col.Ke0, scale=col.dK_scale())
    dAdx = ElementBilinear(ctx, col.elem_dofs, col.Ge_unit, scale=col.dG_scale())
    zero = np.zeros((n, N))
    dfdx = s.add_total_derivative(w, zero, zero, dAdx, dBdx, np.zeros(col.mesh.nelems), adj_corr_data={},
                                  deriv_type="tensor")
    p = rng.uniform(-1.0, 1.0, size=col.mesh.nelems)
    h = 1e-5
    f = []
    from eigd_amd.problems import _assemble

    rho0 = col.rhoE.copy()
    for sgn in (+1.0, -1.0):
        col.rhoE = rho0 + sgn * h * p
        K2 = col.stiffness()                                               # frozen pre-buckling stresses, as dAdx
        G2, _ = _assemble(col.mesh, (col.rhoE**col.p + col.rho0_G)[:, None, None] * col.Ge_unit, 2, col.free_map)
        fac.refactor((K2 + sigma * G2).tocsr())
        s2 = eg.IRAM(N=N, m=65, mode="buckling", ctx=ctx)
        lam2, _ = s2.solve(G2, K2, fac, sigma)
        f.append(float(np.dot(w, np.log(lam2))))
    col.rhoE = rho0
    fd = (f[0] - f[1]) / (2.0 * h)
    assert abs(float(dfdx @ p) - fd) < 1e-5 * abs(fd)


@pytest.mark.parametrize("mode", ["buckling", "normal"])
def test_deflating_extra_converged_pairs_leaves_psi_unchanged(monkeypatch, mode):
    """
    IRAM converges pairs beyond the N requested ones and solve_adjoint(method="sibk") deflates them too (projectors on
    [Phi | Phi_x], their share of psi in closed form, reference 385-389 extended to j > N): psi is unique, so it must
    equal the solve with the reference's deflation set -- all three sibk forms, mode-sharded columns, fewer steps.  The
    extra vectors go out of use when the caller replaces Phi by something they are not B-orthogonal to.
    """
    import eigd_amd as eg
    from eigd_amd.device import default_context
    from eigd_amd.problems import BucklingColumn, ThermalPlate

    ctx = default_context()
    rng = np.random.default_rng(11)
    if mode == "buckling":
        col = BucklingColumn(80, 80, seed=3)
        K = col.stiffness()
        u = col.full_vector(eg.SpLuOperator(K, ctx=ctx, check_symmetry=False)(col.f[col.reduced]))
        A, B, sigma, N = col.geometric_stiffness(u), K, 1.0, 12
    else:
        pl = ThermalPlate(110, epsilon=0.13, rhoE=rng.uniform(0.3, 1.0, size=110 * 110))   # (distinct eigenvalues)
        A, B, sigma, N = pl.stiffness(), pl.mass(), -0.1, 12
    n = B.shape[0]
    P = (B + sigma * A) if mode == "buckling" else (A - sigma * B)
    fac = eg.SpLuOperator(P.tocsr(), ctx=ctx, check_symmetry=False)
    monkeypatch.setattr(eg.tuning, "iram_block", 4)       # (small problem: the block solver is not the automatic choice)
    Phib = rng.uniform(-1, 1, size=(n, N))
    out = {}
    for extra in (0, 9):
        s = eg.IRAM(N=N, m=2 * N + 13, mode=mode, ctx=ctx, extra=extra)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            lam, Phi = s.solve(A, B, fac, sigma)
        assert s.n_extra == extra and s.block_size == 4
        V, T = s.V, s.T
        assert V.shape == (n, s.m) and np.linalg.norm(V.T @ (B @ V) - np.eye(s.m)) < 1e-10
        psi, data = s.solve_adjoint(Phib, method="sibk", rtol=1e-12, update_guess=False, bs_target=1)
        res, _ = s.eval_adjoint_residual_norm(Phib, psi, b_ortho=True)
        assert res.max() < 1e-9 * np.linalg.norm(Phib)
        out[extra] = (s, lam, Phi, psi, list(s.last_info))
    s0, lam0, Phi0, psi0, it0 = out[0]
    s9, lam9, Phi9, psi9, it9 = out[9]
    assert relerr(lam9, lam0) < 1e-12
    sg = np.sign(np.einsum("ij,ij->j", Phi9, Phi0))
    # psi_i is odd in phi_i only through the right-hand side b_i = -(Phib_i - B phi_i (phi_i . Phib_i)): sign free
    assert relerr(psi9, psi0) < 1e-9
    assert sum(it9) < sum(it0) and max(it9) <= max(it0)
    # the other forms of sibk and a mode-sharded solve with the extra pairs
    for kw in ({"bs_target": 2, "update_guess": False}, {"bs_target": 1, "update_guess": True}):
        pk, _ = s9.solve_adjoint(Phib, method="sibk", rtol=1e-12, **kw)
        assert relerr(pk, psi0) < 1e-9, kw

    class OneOfThree:
        rank, size = 1, 3

        def allreduce_sum(self, a):
            return a

    pr, _ = s9.solve_adjoint(Phib, method="sibk", rtol=1e-12, comm=OneOfThree())
    assert relerr(pr[:, 1::3], psi0[:, 1::3]) < 1e-9 and np.all(pr[:, 0::3] == 0.0)
    # a replaced Phi the extra vectors are not orthogonal to: they go out of use, the solve is the plain one
    Q = Phi9.copy()
    Q[:, -1] = s9._prob.Phix.get()[:, 0]                      # (no longer an eigenvector set of the first N pairs)
    s9.Phi = Q
    s9._sync_phi()
    assert s9._prob.PhiD is None
    s9.Phi = Phi9 * sg                                          # signs flipped: still orthogonal, back in use
    s9._sync_phi()
    assert s9._prob.PhiD is not None and s9._prob.PhiD.k == N + 9


def test_first_guess_through_the_lanczos_relation_needs_no_factor_application(monkeypatch):
    """
    The block eigensolver's basis satisfies factor(B V) = V T + Q C E_last^T, so the Lanczos adjoint approximation that
    starts the Krylov solvers, psi0 = -factor(B V Cf) (reference 501-521), is -(V T Cf + Q C Cf_last): no product with B, no
    sweep.  Against the form that applies the factor: the same guess to the accuracy of the Lanczos relation, the same
    iteration counts and psi, N factor applications fewer per call.
    """
    import eigd_amd as eg
    from eigd_amd import adjoint as _adj
    from eigd_amd.device import default_context
    from eigd_amd.problems import BucklingColumn

    ctx = default_context()
    col = BucklingColumn(165, 165, seed=3)                     # 54 k dof: the block form of the restarted solver
    K = col.stiffness()
    u = col.full_vector(eg.SpLuOperator(K, ctx=ctx, check_symmetry=False)(col.f[col.reduced]))
    A, B, sigma, N = col.geometric_stiffness(u), K, 1.0, 12
    fac = eg.SpLuOperator((B + sigma * A).tocsr(), ctx=ctx, check_symmetry=False)
    s = eg.IRAM(N=N, m=2 * N + 1, mode="buckling", ctx=ctx)
    lam, Phi = s.solve(A, B, fac, sigma)
    assert s.block_size > 1 and s._guess is not None
    Vg, c, th, Y, idx, T, C, p = s._guess
    Phib = np.random.default_rng(5).uniform(-1, 1, size=(B.shape[0], N))
    dPhib = ctx.from_host(Phib)
    g_rel = _adj._laa_relation_device(s._prob, Vg, c, p, T, C, dPhib, lam, sigma, Y, th, idx, "buckling").get()
    g_fac = _adj._laa_device(s._prob, Vg, c, dPhib, lam, sigma, Y, th, idx, True, "buckling").get()
    assert relerr(g_rel, g_fac) < 1e-10
    runs = {}
    for flag in ("1", "0"):
        monkeypatch.setattr(eg.tuning, "laa_relation", flag == "1")
        fac.count = 0
        psi, data = s.solve_adjoint(Phib, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
        runs[flag] = (psi, list(s.last_info), fac.count)
    assert runs["1"][1] == runs["0"][1] and relerr(runs["1"][0], runs["0"][0]) < 1e-10
    assert runs["0"][2] - runs["1"][2] == N


def test_lock_step_cycles_follow_predicted_column_ranges_and_survive_wrong_predictions(monkeypatch):
    """
    The lock-step solver decides what the next cycle carries BEFORE the host has solved the current one, from each mode's
    residual history (no cycle behind the last one, the range of the modes not expected to finish).  The predictions only
    steer the pipeline: with every kind of wrong prediction forced -- "all finish" in every cycle (each cycle then waits
    for the host), "the first two and the last mode finish" in every cycle (they get cycles of their own, below and
    above the range in flight), "nobody ever finishes" (the speculative pipeline of before) -- iteration counts and
    residual histories are the same and psi agrees to rounding.
    """
    import eigd_amd as eg
    from eigd_amd import adjoint as _adj
    from eigd_amd.device import default_context
    from eigd_amd.problems import BucklingColumn

    ctx = default_context()
    col = BucklingColumn(90, 90, seed=2)
    K = col.stiffness()
    u = col.full_vector(eg.SpLuOperator(K, ctx=ctx, check_symmetry=False)(col.f[col.reduced]))
    A, B, sigma, N = col.geometric_stiffness(u), K, 1.0, 24
    fac = eg.SpLuOperator((B + sigma * A).tocsr(), ctx=ctx, check_symmetry=False)
    s = eg.IRAM(N=N, m=2 * N + 1, mode="buckling", ctx=ctx)
    lam, Phi = s.solve(A, B, fac, sigma)
    Phib = np.random.default_rng(7).uniform(-1, 1, size=(B.shape[0], N))
    hooks = {"default": None, "all finish": lambda c, v: True, "edges finish": lambda c, v: c in (0, 1, N - 1),
             "nobody finishes": lambda c, v: False}
    monkeypatch.setattr(eg.tuning, "recurrence", "arnoldi")    # (the cycle pipeline of the Arnoldi form)
    runs = {}
    for name, hook in hooks.items():
        monkeypatch.setattr(_adj, "_PREDICT_HOOK", hook)
        for key in ("cycles_enqueued_for_nothing", "cycles_waited_for", "cycles_repeated_for_missed_modes"):
            _adj.LAST_ROUND[key] = 0
        hist = []
        fac.count = 0
        psi, data = s.solve_adjoint(Phib, method="sibk", rtol=1e-10, update_guess=False, bs_target=1, callback=hist.append)
        runs[name] = (psi, list(s.last_info), hist, fac.count, dict(_adj.LAST_ROUND))
        assert _adj.LAST_ROUND["steps_per_pass"] == 2
    psi0, info0, hist0, count0, round0 = runs["default"]
    assert max(info0) > 8 and round0["cycles_enqueued_for_nothing"] == 0       # the last cycle was seen coming
    for name in ("all finish", "edges finish", "nobody finishes"):
        psi, info, hist, count, rnd = runs[name]
        assert info == info0 and count == count0, name
        assert len(hist) == len(hist0) and np.allclose(hist, hist0, rtol=1e-6, atol=1e-300), name
        assert relerr(psi, psi0) < 1e-11, name
    assert runs["all finish"][4]["cycles_waited_for"] >= (max(info0) + 1) // 2 - 2
    assert runs["edges finish"][4]["cycles_repeated_for_missed_modes"] >= 4
    assert runs["nobody finishes"][4]["cycles_enqueued_for_nothing"] == 1


@pytest.mark.parametrize("mode", ["buckling", "normal"])
def test_two_krylov_steps_per_gram_schmidt_pass_match_the_one_step_solver(monkeypatch, mode):
    """
    The lock-step sibk with two operator applications per Gram-Schmidt pass (tuning.steps_per_pass = 2, the default of the Arnoldi form) against the
    one-step form (the reference's loop order, eigenvector_derivatives.py:1246-1260): same iteration count for every
    mode, same residual histories, psi equal far below the 1e-8 of north_star -- odd and even stopping steps, modes that
    are converged before the first step, a maxiter that cuts a cycle in two, more than 32 modes (chunks), and the
    fall-back to the one-step form when the orthogonality measured inside a pair is not accepted.
    """
    import eigd_amd as eg
    from eigd_amd.device import default_context
    from eigd_amd.problems import BucklingColumn, ThermalPlate

    ctx = default_context()
    rng = np.random.default_rng(7)
    if mode == "buckling":
        col = BucklingColumn(90, 90, seed=2)
        K = col.stiffness()
        u = col.full_vector(eg.SpLuOperator(K, ctx=ctx, check_symmetry=False)(col.f[col.reduced]))
        A, B, sigma, N = col.geometric_stiffness(u), K, 1.0, 40
    else:
        pl = ThermalPlate(90, epsilon=1e-7)                    # nearly repeated pairs: index sets with partners
        A, B, sigma, N = pl.stiffness(), pl.mass(), -0.1, 12
    n = B.shape[0]
    P = (B + sigma * A) if mode == "buckling" else (A - sigma * B)
    fac = eg.SpLuOperator(P.tocsr(), ctx=ctx, check_symmetry=False)
    s = eg.IRAM(N=N, m=2 * N + 1, mode=mode, ctx=ctx)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                        # (the pair N, N+1 of the thermal plate is repeated)
        lam, Phi = s.solve(A, B, fac, sigma)
    Phib = rng.uniform(-1, 1, size=(n, N))
    Phib[:, 3] = 0.0                                           # a right-hand side that needs no Krylov step at all
    runs = {}
    monkeypatch.setattr(eg.tuning, "recurrence", "arnoldi")   # (the forms compared here are the two Arnoldi forms)
    for name, env in (("one", {"steps_per_pass": 1}), ("two", {"steps_per_pass": 2}),
                      ("fallback", {"steps_per_pass": 2, "pair_defect_tol": -1.0})):
        for kname, v in env.items():
            monkeypatch.setattr(eg.tuning, kname, v)
        hist = []
        fac.count = 0
        psi, data = s.solve_adjoint(Phib, method="sibk", rtol=1e-10, update_guess=False, bs_target=1, callback=hist.append)
        runs[name] = (psi, data, list(s.last_info), hist, fac.count)
        from eigd_amd import adjoint as _adj

        assert _adj.LAST_ROUND["steps_per_pass"] == (2 if name == "two" else 1), name   # the form asked for (or the fall-back) ran
        monkeypatch.setattr(eg.tuning, "pair_defect_tol", 1e-10)
    psi1, data1, info1, hist1, count1 = runs["one"]
    psi2, data2, info2, hist2, count2 = runs["two"]
    assert info2 == info1 and len(set(i % 2 for i in info1 if i)) == 2          # odd and even stopping steps
    assert len(hist2) == len(hist1) and np.allclose(hist2, hist1, rtol=1e-6, atol=1e-300)
    assert relerr(psi2, psi1) < 1e-11
    assert index_sets(data2) == index_sets(data1)
    psi3, _, info3, _, count3 = runs["fallback"]
    assert info3 == info1 and np.array_equal(psi3, psi1)                       # the one-step form took over
    assert count1 == count2 == count3, (count1, count2, count3)               # applications per mode (ref 19-22): an
    # abandoned two-step attempt is not counted on top of the one-step solve that replaces it
    # a maxiter inside a cycle: both forms keep the same best iterates
    monkeypatch.setattr(eg.tuning, "steps_per_pass", 1)
    pa, _ = s.solve_adjoint(Phib, method="sibk", rtol=1e-10, update_guess=False, bs_target=1, maxiter=7, nrestart=0)
    monkeypatch.setattr(eg.tuning, "steps_per_pass", 2)
    pb, _ = s.solve_adjoint(Phib, method="sibk", rtol=1e-10, update_guess=False, bs_target=1, maxiter=7, nrestart=0)
    assert relerr(pb, pa) < 1e-10
    # ONE projection of the raw pair in place of the two behind the operator applications (1250-1252) when the measured
    # invariance of range(P) allows (converged eigenvectors; the default): same counts, same psi as the reference's placement
    assert _adj.LAST_ROUND["inner_projections"] is False
    monkeypatch.setattr(eg.tuning, "inner_projections", True)
    pd, _ = s.solve_adjoint(Phib, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
    assert _adj.LAST_ROUND["inner_projections"] is True and list(s.last_info) == info2
    assert relerr(psi2, pd) < 1e-11
    monkeypatch.setattr(eg.tuning, "inner_projections", False)
    # a Krylov history deeper than one coefficient block of the pair kernels (maxiter > 120 at 32 columns): the one-step
    # form is chosen up front, nothing raises in mid-solve
    pc, _ = s.solve_adjoint(Phib, method="sibk", rtol=1e-10, update_guess=False, bs_target=1, maxiter=130)
    assert _adj.LAST_ROUND["steps_per_pass"] == (1 if N >= 32 else 2)
    assert relerr(pc, psi1) < 1e-10


def test_block_lanczos_basis_stays_b_orthonormal_with_the_local_first_pass(monkeypatch):
    """
    ``tuning.lanczos_local_first_pass`` (default) orthogonalises a new block against the last two blocks first and then
    makes ONE unmeasured pass over the whole basis, instead of two passes over everything with the second one measured
    at 1e-13.  Both forms must leave the run's own basis B-orthonormal to 1e-12 and return the same eigenpairs
    (reference: ARPACK's full reorthogonalisation behind eigsh_mod, arpack.py:438-440).
    """
    import eigd_amd as eg
    from eigd_amd.problems import FreePlate

    pl = FreePlate(90, 90, seed=2)                       # 16 562 dof; blocks of 4 forced below
    K, M = pl.stiffness(), pl.mass()
    sigma, N = -10.0, 10
    fac = eg.SpLuOperator((K - sigma * M).tocsr(), check_symmetry=False)
    monkeypatch.setattr(eg.tuning, "iram_block", 4)
    out = {}
    for flag in (True, False):
        monkeypatch.setattr(eg.tuning, "lanczos_local_first_pass", flag)
        s = eg.IRAM(N=N, m=40)
        lam, Phi = s.solve(K, M, fac, sigma)
        assert s.block_size == 4 and s._guess is not None
        Vg, c = s._guess[0], s._guess[1]                 # the run's own basis (c vectors + the residual block)
        V = np.column_stack([Vg.get_block(j, 1).get()[:, 0] for j in range(c)])
        G = V.T @ (M @ V)
        out[flag] = (lam, Phi, np.abs(G - np.eye(c)).max(), s.eig_res_true.max())
    for flag in (True, False):
        assert out[flag][2] <= 1e-12, (flag, out[flag][2])
        assert out[flag][3] <= 1e-9 * abs(1.0 / (out[flag][0][0] - sigma))
    assert relerr(out[True][0], out[False][0]) < 1e-12
    # (the plate's rigid-body modes share an eigenvalue: compare the spectral projector of the N pairs, not the vectors)
    Z = np.random.default_rng(0).normal(size=(K.shape[0], 5))
    PB = lambda P: P @ (P.T @ (M @ Z))                   # noqa: E731
    assert relerr(PB(out[True][1]), PB(out[False][1])) < 1e-7


def test_buckling_mode_dl_follows_the_restated_algorithm_where_its_recursion_is_still_stable():
    """
    Buckling-mode ``dl`` (reference eigenvector_derivatives.py:526-696) has no parity fixture: the reference's own run
    diverges there (G1: residuals 1e14, psi 2e15 away from its own ``sibk``; G4: 1.3e20 -- INTEGRATION.md), so
    ``test_g4_method_matrix_basiclanczos`` skips it.  The reverse recursion through the Lanczos process amplifies
    rounding by ~30 per four steps on this pencil whatever computes it (tools/dl_probe.py,
    profiles/r05_buckling_dl_probe.txt: the device and the oracle's numpy restatement agree to 2e-10 at m = 12, 2e-8 at
    16, 2e-6 at 20, not at all at 60, where both are 1e15 away from ``sibk``), so no residual gate can be met at the
    reference's m = 60.  What CAN be held: with a short Lanczos run, where the amplification is still small, the device
    ``dl`` in buckling mode equals the oracle's ``dl`` (pinned against the reference's normal-mode output) -- the
    buckling branch of the implementation is the restated algorithm's -- and approaches ``sibk`` as far as the
    unconverged Ritz pairs allow.
    """
    import eigd_amd as eg
    from oracle import eigd_oracle as orc

    g = load_golden("g1_buckling50_basiclanczos")
    K, G = csr_from(g, "K"), csr_from(g, "G")
    sigma = float(g["sigma"])
    fac = eg.SpLuOperator((K + sigma * G).tocsc())
    fac_o = orc.SpLuOperator((K + sigma * G).tocsc())
    for m, tol in ((12, 1e-8), (16, 1e-6)):
        s = eg.BasicLanczos(N=6, m=m, mode="buckling", tol=0.0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            lam, Phi = s.solve(G, K, fac, sigma)
            assert s.m == m
            Qrb = g["Qrb"] * np.sign(np.einsum("ij,ij->j", Phi, g["Phi"]))
            psi_dl, data = s.solve_adjoint(Qrb, method="dl")
            psi_o, data_o = orc.dl(Qrb, K, fac_o, sigma, lam, Phi, np.asarray(s.indices), np.asarray(s.V)[:, :m],
                                   np.asarray(s.T), np.asarray(s.Y), np.asarray(s.theta), mode="buckling")
        assert index_sets(data) == index_sets(data_o)
        assert np.all(np.isfinite(psi_dl)) and relerr(psi_dl, psi_o) < tol, (m, relerr(psi_dl, psi_o))
    psi_s, _ = s.solve_adjoint(Qrb, method="sibk", rtol=1e-12)
    assert relerr(psi_dl, psi_s) < 1e-2                   # (m = 16: Ritz pairs converged to 2e-2, the two differ by that)
