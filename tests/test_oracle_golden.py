"""
Pin the CPU oracle (oracle/eigd_oracle.py) to vectors captured from the reference
(tools/make_golden.py).  CPU only.
"""
import warnings

import numpy as np
import pytest

from conftest import align_signs, corr_from, csr_from, index_sets, load_golden, relerr
from oracle import eigd_oracle as orc


def _mock_cb(C):
    def cb(w, v):
        if w.ndim == 1:
            return C.T @ (w * v)
        return C.T @ np.sum(w * v, axis=1)
    return cb


# ----------------------------------------------------------------- G5 units
def test_project_units():
    g = load_golden("g5_units")
    out = orc.project(g["proj_U"], g["proj_V"], g["proj_X"].copy())
    assert np.array_equal(out, g["proj_out"])
    out1 = orc.project(g["proj_U"], g["proj_V"], g["proj_x1"].copy())
    assert np.array_equal(out1, g["proj_out1"])


@pytest.mark.parametrize("tag", ["distinct", "repeated"])
@pytest.mark.parametrize("mode", ["normal", "buckling"])
def test_correction_and_total_derivative_units(tag, mode):
    g = load_golden("g5_units")
    lam = g[tag + "_lam"]
    assert orc.are_eigenvalues_repeated(lam) == bool(g[tag + "_repeated"])
    p = f"{tag}_{mode}_"
    psi = g[p + "psi_in"].copy()
    data = orc.generate_adjoint_correction(lam, g["Phi"], psi, Phib=g["Phib"], mode=mode)
    ref = corr_from(g, p + "corr")
    assert index_sets(data) == index_sets(ref)  # bit-exact index sets
    for i in ref:
        for (j, xi, eta), (jr, xir, etar) in zip(data[i], ref[i]):
            assert xi == xir and eta == etar
    assert np.array_equal(psi, g[p + "psi_out"])
    for dt in ("vector", "tensor"):
        dfdx = orc.add_eig_total_derivative(
            lam, g["Phi"], g["lamb"], g["Phib"], psi, _mock_cb(g["Ca"]), _mock_cb(g["Cb"]),
            np.zeros(g["Ca"].shape[1]), adj_corr_data=data, mode=mode, deriv_type=dt)
        assert relerr(dfdx, g[p + "dfdx_" + dt]) < 1e-13


# ----------------------------------------------------------- BasicLanczos
def _basic_case(g, A, B, sigma, mode, prefix="", **kw):
    mat = (A - sigma * B) if mode == "normal" else (B + sigma * A)
    factor = orc.SpLuOperator(mat.tocsc())
    s = orc.BasicLanczos(mode=mode, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s.solve(A, B, factor, sigma)
    assert s.m == int(g[prefix + "m"])
    assert s.N == int(g[prefix + "N"])
    assert np.array_equal(s.indices[: s.N], g[prefix + "indices"][: s.N])
    assert relerr(s.lam0, g[prefix + "lam"]) < 1e-11
    # the first Lanczos steps are a deterministic recurrence: tight agreement
    assert relerr(s.alpha[:5], g[prefix + "alpha"][:5]) < 1e-10
    assert relerr(s.beta[:5], g[prefix + "beta"][:5]) < 1e-10
    return s, factor


def test_basiclanczos_g1_buckling_full_chain():
    g = load_golden("g1_buckling50_basiclanczos")
    K, G = csr_from(g, "K"), csr_from(g, "G")
    s, factor = _basic_case(g, G, K, float(g["sigma"]), "buckling", N=6, m=60, tol=0.0)
    assert relerr(s.lam0, g["BLF"]) < 1e-11
    Phi, sg = align_signs(s.Phi, g["Phi"])
    assert relerr(Phi, g["Phi"]) < 1e-7
    # adjoint stage on the reference's own (Phi, Phib): compare psi and index sets
    s.Phi = g["Phi"].copy()
    s.Y[:, s.indices[: s.N]] *= sg
    factor.count = 0
    psi, data = s.solve_adjoint(g["Qrb"], method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
    assert index_sets(data) == index_sets(corr_from(g, "corr"))
    assert relerr(psi, g["psir"]) < 1e-8
    assert abs(factor.count - int(g["count_adjoint"])) <= 2


@pytest.mark.parametrize("name", ["g3_thermal32_eps1e-1_basiclanczos", "g3_thermal32_eps1e-8_basiclanczos"])
def test_basiclanczos_g3_repeated_index_sets(name):
    g = load_golden(name)
    K, M = csr_from(g, "K"), csr_from(g, "M")
    s, factor = _basic_case(g, K, M, float(g["sigma"]), "normal", N=8, m=60, tol=0.0)
    # eigen-stage: eigenvalues; adjoint stage is run on the stored (Phi, Qb)
    s.Phi = g["Phi"].copy()
    s.V[:, : s.m] = g["V"]
    s.Y, s.theta, s.indices = g["Y"].copy(), g["theta"].copy(), g["indices"].copy()
    psi, data = s.solve_adjoint(g["Qb"], method="sibk", rtol=1e-12, update_guess=False, bs_target=1)
    ref = corr_from(g, "corr")
    assert index_sets(data) == index_sets(ref)
    if "eps1e-8" in name:
        assert index_sets(data) == {1: [2], 2: [1], 4: [5], 5: [4], 6: [7], 7: [6]}
    else:
        assert data == {}
    for i in ref:
        for (j, xi, eta), (jr, xir, etar) in zip(data[i], ref[i]):
            assert abs(xi - xir) <= 1e-8 * max(1.0, abs(xir))
            assert abs(eta - etar) <= 1e-8 * max(1.0, abs(etar))
    assert relerr(psi, g["psi"]) < 1e-8
    res, ortho = s.eval_adjoint_residual_norm(g["Qb"], psi, b_ortho=True)
    assert np.allclose(res, g["res_bortho"], atol=1e-9)


def test_basiclanczos_g2_normal():
    g = load_golden("g2_natfreq32x16_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    s, factor = _basic_case(g, K, M, float(g["sigma"]), "normal", N=13, m=60, tol=1e-14)
    s.Phi = g["Phi"].copy()
    s.V[:, : s.m] = g["V"]
    s.Y, s.theta, s.indices = g["Y"].copy(), g["theta"].copy(), g["indices"].copy()
    psi0, data = s.solve_adjoint(g["Q0b"], method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
    assert index_sets(data) == index_sets(corr_from(g, "corr"))
    # the rigid-body modes (first three, lam ~ 0 and mutually repeated) are discarded by the harness
    assert relerr(psi0[:, 3:], g["psi"]) < 1e-7


# ------------------------------------------------------------ method matrix
@pytest.mark.parametrize("mode", ["normal", "buckling"])
def test_method_matrix_basiclanczos(mode):
    g = load_golden("g4_laplace900_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    A, B = (K, M) if mode == "normal" else ((-0.005 * M).tocsr(), K)
    p = mode + "_"
    sigma = float(g[p + "sigma"])
    s, factor = _basic_case(g, A, B, sigma, mode, prefix=p, N=6, m=60)
    # run every method from the reference's own Lanczos data so psi is comparable
    s.Phi = g[p + "Phi"].copy()
    s.V[:, : s.m] = g[p + "V"]
    s.Y, s.theta, s.indices, s.T = g[p + "Y"].copy(), g[p + "theta"].copy(), g[p + "indices"].copy(), g[p + "T"].copy()
    for method, tol in (("laa", 1e-10), ("sibk", 1e-9), ("pcpg", 1e-8), ("pgmres", 1e-9), ("dl", 1e-9)):
        kw = {"update_guess": False, "bs_target": 1} if method == "sibk" else {}
        if method != "laa" and g[p + method + "_res"].max() > 1e-6:
            continue  # the reference itself did not converge here (dl, buckling shift): nothing to pin
        factor.count = 0
        psi, data = s.solve_adjoint(g["Phib"].copy(), method=method, rtol=1e-12, **kw)
        assert index_sets(data) == index_sets(corr_from(g, p + method + "_corr")), method
        assert relerr(psi, g[p + method + "_psi"]) < tol, method
        assert factor.count == int(g[p + method + "_count"]), method
    psi, _ = s.solve_adjoint(g["Phib"].copy(), method="sibk", rtol=1e-12, bs_target=2)
    assert relerr(psi, g[p + "sibk_bs2_psi"]) < 1e-9
    psi, _ = s.solve_adjoint(g["Phib"].copy(), method="sibk", rtol=1e-12, update_guess=True)
    assert relerr(psi, g[p + "sibk_ug_psi"]) < 1e-9


# -------------------------------------------------------------------- IRAM
@pytest.mark.parametrize("mode", ["normal", "buckling"])
def test_iram_oracle_eigenpairs_and_lanczos_relation(mode):
    g = load_golden("g4_laplace900_iram")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    A, B = (K, M) if mode == "normal" else ((-0.005 * M).tocsr(), K)
    p = mode + "_"
    sigma = float(g[p + "sigma"])
    mat = (A - sigma * B) if mode == "normal" else (B + sigma * A)
    factor = orc.SpLuOperator(mat.tocsc())
    s = orc.IRAM(N=6, m=40, mode=mode)
    lam, Phi = s.solve(A, B, factor, sigma)
    assert relerr(lam, g[p + "lam"]) < 1e-11
    Phi, _ = align_signs(Phi, g[p + "Phi"])
    assert relerr(Phi, g[p + "Phi"]) < 1e-7
    # (V, T) is a valid B-orthonormal Lanczos factorisation: OP V = V T + f e_m^T
    V, T = s.V, s.T
    OPV = np.column_stack([factor(B @ V[:, j]) for j in range(V.shape[1])])
    Rm = OPV - V @ T
    assert np.linalg.norm(Rm[:, :-1]) < 1e-10 * np.linalg.norm(OPV)
    assert np.linalg.norm(V.T @ (B @ V) - np.eye(V.shape[1])) < 1e-10
    # adjoint on the reference's Lanczos data
    s.Phi, s.V, s.T = g[p + "Phi"].copy(), g[p + "V"].copy(), g[p + "T"].copy()
    s.Y, s.theta, s.indices = g[p + "Y"].copy(), g[p + "theta"].copy(), g[p + "indices"].copy()
    for method, tol in (("laa", 1e-10), ("sibk", 1e-9), ("pgmres", 1e-9)):
        kw = {"update_guess": False, "bs_target": 1} if method == "sibk" else {}
        psi, data = s.solve_adjoint(g["Phib"].copy(), method=method, rtol=1e-12, **kw)
        assert relerr(psi, g[p + method + "_psi"]) < tol, method


def test_iram_g1_buckling_eigs_and_adjoint():
    g = load_golden("g1_buckling50_iram")
    K, G = csr_from(g, "K"), csr_from(g, "G")
    sigma = float(g["sigma"])
    factor = orc.SpLuOperator((K + sigma * G).tocsc())
    s = orc.IRAM(N=6, m=60, mode="buckling")
    lam, Phi = s.solve(G, K, factor, sigma)
    assert relerr(lam, g["lam"]) < 1e-10
    assert np.allclose(lam, g["BLF"], rtol=1e-10)
    s.Phi, s.V, s.T = g["Phi"].copy(), g["V"].copy(), g["T"].copy()
    s.Y, s.theta, s.indices = g["Y"].copy(), g["theta"].copy(), g["indices"].copy()
    psi, data = s.solve_adjoint(g["Qrb"], method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
    assert index_sets(data) == index_sets(corr_from(g, "corr"))
    assert relerr(psi, g["psir"]) < 1e-8


def _complex_csr(g, name):
    from scipy import sparse

    return sparse.csr_matrix((g[name + "_re"] + 1j * g[name + "_im"], g[name + "_indices"], g[name + "_indptr"]),
                             shape=tuple(g[name + "_shape"]))


def check_complex_step_solution(g, s):
    """lam, Phi, alpha, beta of a complex-step BasicLanczos run against the reference's (G6); the imaginary parts are
    step * derivative: compared relative to their own size"""
    dh = float(g["dh"])
    assert s.m == int(g["m"]) and s.N == int(g["N"])
    assert np.abs(s.lam0.real - g["lam"].real).max() < 1e-10 * np.abs(g["lam"].real).max()
    assert np.abs(s.lam0.imag - g["lam"].imag).max() < 1e-8 * np.abs(g["lam"].imag).max()
    for q in range(s.N):
        sg = np.sign(np.dot(s.Phi[:, q].real, g["Phi"][:, q].real))
        assert relerr(sg * s.Phi[:, q].real, g["Phi"][:, q].real) < 1e-8, q
        assert relerr(sg * s.Phi[:, q].imag, g["Phi"][:, q].imag) < 1e-6, q
    # the reference's own use of the result: the tanh aggregate of examples/buckling.py:702-722 and its CS derivative
    lam, Q = s.lam0, np.zeros((int(g["reduced"].max()) + 1 + 1, s.N), dtype=complex)
    Q[g["reduced"]] = s.Phi
    eta = np.tanh(100.0 * (lam - 0.0)) - np.tanh(100.0 * (lam - 50.0))
    eta = eta / np.sum(eta)
    node = int(g["node"])
    h = sum(eta[i] * Q[node, i] * Q[node, i] for i in range(s.N))
    assert abs(h.real - g["h"].real) < 1e-10 * abs(g["h"].real)
    assert abs(h.imag / dh - float(g["cs"])) < 1e-7 * abs(float(g["cs"]))


def test_basiclanczos_g6_complex_step():
    """the oracle's complex-step semantics against the reference's own CS evaluation of C1"""
    g = load_golden("g6_buckling50_complexstep")
    K, G = _complex_csr(g, "K"), _complex_csr(g, "G")
    sigma = float(g["sigma"])
    factor = orc.SpLuOperator((K + sigma * G).tocsc())
    s = orc.BasicLanczos(N=6, m=60, tol=0.0, mode="buckling")
    s.solve(G, K, factor, sigma)
    check_complex_step_solution(g, s)
    assert np.abs(s.alpha - g["alpha"]).max() < 1e-9 * np.abs(g["alpha"]).max()


def test_error_behaviour():
    with pytest.raises(ValueError):
        orc.BasicLanczos(mode="nope")
    with pytest.raises(ValueError):
        orc.BasicLanczos(ortho_type="nope")
    with pytest.raises(ValueError):
        orc.BasicLanczos(Ntarget=1.5)
    with pytest.raises(ValueError):
        orc.IRAM(mode="nope")
    assert orc.IRAM(N=4).m == 20 and orc.IRAM(N=30, m=40).m == 61


def test_oracle_sample_hooks_reproduce_the_full_run():
    """bench.py times a few modes on the CPU through sibk(modes=...) / laa(cols=...): same columns as the full run"""
    g = load_golden("g4_laplace900_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    p = "buckling_"
    A, B = (-0.005 * M).tocsr(), K
    sigma = float(g[p + "sigma"])
    fac = orc.SpLuOperator((B + sigma * A).tocsc())
    lam, Phi, Phib = g[p + "lam"], g[p + "Phi"], g["Phib"]
    args = (Phib, B, fac, sigma, lam, g[p + "V"], g[p + "Y"], g[p + "theta"], g[p + "indices"])
    full0 = orc.laa(*args, b_ortho=True, mode="buckling")
    part0 = orc.laa(*args, b_ortho=True, mode="buckling", cols=[1, 4])
    assert relerr(part0[:, [1, 4]], full0[:, [1, 4]]) < 1e-12 and not part0[:, [0, 2, 3, 5]].any()
    full, data, _ = orc.sibk(Phib, A, B, lam, Phi, mode="buckling", psi=full0.copy(), sigma=sigma, factor=fac, rtol=1e-12)
    for i in (1, 4):
        one, data1, info = orc.sibk(Phib, A, B, lam, Phi, mode="buckling", psi=part0.copy(), sigma=sigma, factor=fac,
                                    rtol=1e-12, modes=[i])
        assert len(info) == 1 and index_sets(data1) == index_sets(data)
        assert relerr(one[:, i], full[:, i]) < 1e-12
    with pytest.raises(ValueError):
        orc.sibk(Phib, A, B, lam, Phi, mode="buckling", sigma=sigma, factor=fac, modes=[0], bs_target=2)
