import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
