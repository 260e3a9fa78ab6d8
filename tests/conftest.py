import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


_FAULT_LOG = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A crash below Python (an abort in a native library, a fault in a finaliser thread) must leave its stacks behind --
    # of every thread, in a file that survives the process and whatever the output was piped through (docs/LOG.md,
    # round 4: one abort of the GPU suite was lost behind a `tail`).  EIGD_FAULT_LOG names the file; by default it is
    # gpurun_out/faulthandler_<pid>.log where that directory exists (the GPU box), else the system's temporary directory.
    import faulthandler
    import tempfile

    global _FAULT_LOG
    d = os.path.join(ROOT, "gpurun_out")
    path = os.environ.get("EIGD_FAULT_LOG") or os.path.join(d if os.path.isdir(d) else tempfile.gettempdir(),
                                                            f"faulthandler_{os.getpid()}.log")
    try:
        _FAULT_LOG = open(path, "w")
        faulthandler.enable(file=_FAULT_LOG, all_threads=True)
    except OSError:
        faulthandler.enable(all_threads=True)                 # (no writable place: at least stderr)


def pytest_unconfigure(config):
    import faulthandler

    global _FAULT_LOG
    if _FAULT_LOG is not None:
        faulthandler.disable()
        path, size = _FAULT_LOG.name, _FAULT_LOG.tell()
        _FAULT_LOG.close()
        _FAULT_LOG = None
        if size == 0:                                         # a clean run leaves nothing behind
            try:
                os.unlink(path)
            except OSError:
                pass


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def csr_from(g, prefix):
    from scipy import sparse

    shape = tuple(int(v) for v in g[prefix + "_shape"])
    return sparse.csr_matrix((g[prefix + "_data"], g[prefix + "_indices"], g[prefix + "_indptr"]), shape=shape)


def corr_from(g, prefix):
    """flat arrays -> {i: [(j, xi, eta), ...]} preserving order"""
    data = {}
    for i, j, xi, eta in zip(g[prefix + "_i"], g[prefix + "_j"], g[prefix + "_xi"], g[prefix + "_eta"]):
        data.setdefault(int(i), []).append((int(j), float(xi), float(eta)))
    return data


def exact_pair_coefficients(lam, Phi, Phib, data, mode="normal"):
    """
    xi, eta of the repeated pairs (reference eigenvector_derivatives.py:373-383) in EXACT rational arithmetic from the
    given floating-point inputs: the same dict layout as ``data`` (whose index sets are kept).  What a floating-point
    evaluation of those formulas deviates from this by is its own rounding: xi, eta divide the difference of two n-term
    dot products by the eigenvalue gap of the pair.
    """
    from fractions import Fraction

    def dot(x, y):
        return sum(Fraction(a) * Fraction(b) for a, b in zip(x.tolist(), y.tolist()))

    cache = {}

    def g0(j, i):
        if (j, i) not in cache:
            v = -dot(Phi[:, j], Phib[:, i])
            cache[(j, i)] = v if mode == "normal" else Fraction(lam[j]) * v
        return cache[(j, i)]

    out = {}
    for i, lst in data.items():
        for tup in lst:
            j = tup[0]
            hi, lo = max(i, j), min(i, j)             # the reference's loop forms the pair as (i, j) with j < i
            gap = Fraction(lam[lo]) - Fraction(lam[hi])
            xi = Fraction(1, 2) * (g0(lo, hi) - g0(hi, lo)) / gap
            eta = Fraction(1, 2) * (Fraction(lam[hi]) * g0(lo, hi) - Fraction(lam[lo]) * g0(hi, lo)) / gap
            out.setdefault(i, []).append((j, float(xi), float(eta)))
    return out


def pair_rounding_in_dfdx(data_a, data_b, Phi, dAdx, dBdx, mode="normal"):
    """
    first-order change of df/dx when the (xi, eta) of ``data_a`` are replaced by those of ``data_b`` (same index sets):
    the weight vectors are linear in them (reference 96-111 / 118-132), dfdx = dAdx(WA, Phi) -+ dBdx(WB, Phi)
    """
    delta = 0.0
    sB = -1.0 if mode == "normal" else 1.0
    ia, ib = (1, 2) if mode == "normal" else (2, 1)
    for i in data_a:
        for ta, tb in zip(data_a[i], data_b[i]):
            assert ta[0] == tb[0]
            j = ta[0]
            delta = delta + (tb[ia] - ta[ia]) * dAdx(Phi[:, [j]], Phi[:, [i]]) + sB * (tb[ib] - ta[ib]) * dBdx(Phi[:, [j]], Phi[:, [i]])
    return delta


def index_sets(data):
    return {i: [t[0] for t in lst] for i, lst in data.items()}


def align_signs(Phi, Phi_ref):
    """flip columns of Phi to match the sign of Phi_ref (eigenvectors are sign-ambiguous)"""
    s = np.sign(np.einsum("ij,ij->j", Phi, Phi_ref))
    s[s == 0] = 1.0
    return Phi * s, s


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


@pytest.fixture(scope="session")
def golden():
    return load_golden
