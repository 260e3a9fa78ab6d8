"""CPU-side checks of the boundary: the shared library loads without a GPU, exports every symbol
include/eigd_hip.h declares, reports errors through the documented channel, and the host-only
entry points work."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "eigd_hip.h")).read()
    return sorted(set(re.findall(r"\b(eigd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from eigd_amd import _ffi

    L = _ffi.lib()
    names = declared_symbols()
    assert len(names) >= 35
    for name in names:
        assert hasattr(L, name), name
    # the ctypes layer binds exactly the declared functions
    assert sorted(_ffi.EXPORTED) == names


def test_version_and_error_channel():
    from eigd_amd import _ffi

    L = _ffi.lib()
    assert L.eigd_version() == 100
    rc = L.eigd_symbolic_create(3, None, None, 0, 0, None)
    assert rc == _ffi.EIGD_E_INVALID
    assert "null" in _ffi.last_error()
    with pytest.raises(ValueError):
        _ffi.check(rc)


def test_product_fails_loudly_without_library(monkeypatch, tmp_path):
    from eigd_amd import _ffi

    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(_ffi.EigdHipError):
        _ffi.lib()


def test_no_gpu_means_error_not_fallback():
    """on a machine without a GPU every device entry point raises; nothing is computed on the host"""
    from eigd_amd import _ffi

    cnt = ctypes.c_int(-1)
    rc = _ffi.lib().eigd_device_count(ctypes.byref(cnt))
    if rc == 0 and cnt.value > 0:
        pytest.skip("a GPU is visible")
    from eigd_amd.device import Context

    with pytest.raises((_ffi.EigdHipError, ValueError)):
        Context(0)
    import eigd_amd as eg
    from scipy import sparse

    with pytest.raises((_ffi.EigdHipError, ValueError)):
        eg.SpLuOperator(sparse.identity(4, format="csr"))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "eigd_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("the CPU oracle", ""), fn


def test_host_helpers_match_oracle():
    """host-side pieces of the product (index sets, weights, Ritz sorting) against the oracle"""
    from eigd_amd import adjoint as adj
    from eigd_amd.lanczos import ritz_to_eigs
    from oracle import eigd_oracle as orc

    rng = np.random.default_rng(0)
    N = 7
    lam = np.array([1.0, 1.0 + 3e-6, 2.0, 2.5, 2.5 + 9e-6, 2.5 + 1.2e-5, 4.0])
    G = rng.normal(size=(N, N))
    n = 30
    Phi = rng.normal(size=(n, N))
    for mode in ("normal", "buckling"):
        Cc, data = adj.correction_coefficients(lam, G, 1e-5, mode)
        psi = rng.normal(size=(n, N))
        psi_o = psi.copy()
        data_o = orc.generate_adjoint_correction(lam, Phi, psi_o, G=G, eig_atol=1e-5, mode=mode)
        assert {i: [t[0] for t in v] for i, v in data.items()} == {i: [t[0] for t in v] for i, v in data_o.items()}
        for i in data:
            for a, b in zip(data[i], data_o[i]):
                assert a == b  # same formulas, same order: identical floats
        assert np.allclose(psi + Phi @ Cc, psi_o, rtol=1e-13, atol=1e-13)
        Phib, lamb = rng.normal(size=(n, N)), rng.normal(size=N)
        beta = 0.5 * np.einsum("ij,ij->j", Phi, Phib)
        CA, CB, sa, sb = adj.derivative_weight_coefficients(lam, lamb, beta, data, mode, N)
        WA_o, WB_o = orc.derivative_weights(lam, Phi, lamb, Phib, psi, data, mode)
        assert np.allclose(Phi @ CA + psi * sa, WA_o, rtol=1e-13, atol=1e-13)
        assert np.allclose(Phi @ CB + psi * sb, WB_o, rtol=1e-13, atol=1e-13)
    theta = rng.uniform(0.2, 3.0, size=12)
    for mode in ("normal", "buckling"):
        l1, i1 = ritz_to_eigs(theta, 0.7, mode)
        l2, i2 = orc.ritz_to_eigs(theta, 0.7, mode)
        assert np.array_equal(l1, l2) and np.array_equal(i1, i2)
    assert adj.are_eigenvalues_repeated(lam) and not adj.are_eigenvalues_repeated(lam[[0, 2, 3, 6]])
    Yb = rng.normal(size=(12, 4))
    Y = np.linalg.qr(rng.normal(size=(12, 12)))[0]
    idx = np.argsort(theta)[::-1]
    for b_ortho in (True, False):
        Cf = adj.laa_coefficients(Yb, np.array([1.0, 2.0, 3.0, 4.0]), 0.3, Y, theta, idx, b_ortho, "buckling")
        assert Cf.shape == (12, 4) and np.all(np.isfinite(Cf))


def test_problem_generators_reproduce_reference_matrices():
    from conftest import csr_from, load_golden
    from scipy.sparse.linalg import splu

    from eigd_amd.problems import BucklingColumn

    g = load_golden("g1_buckling50_basiclanczos")
    Kg, Gg = csr_from(g, "K"), csr_from(g, "G")
    col = BucklingColumn(50, 50, 1.0, 1.0, rhoE=np.full(2500, 0.5))
    K = col.stiffness()
    assert K.shape == Kg.shape and abs(K - Kg).max() < 1e-14
    u = col.full_vector(splu(K.tocsc()).solve(col.f[col.reduced]))
    G = col.geometric_stiffness(u)
    assert abs(G - Gg).max() < 1e-15
