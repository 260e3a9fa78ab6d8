"""
eigd_amd: MI355X-native implementation of smdogroup/eigd's adjoint eigenvector-derivative
hot path.  Drop-in names (reference eigd/__init__.py:1-3 star-exports its module):

    SpLuOperator, IRAM, BasicLanczos,
    add_eig_total_derivative, eval_adjoint_residual_norm, are_eigenvalues_repeated,
    generate_adjoint_correction, laa, dl, pcpg, pgmres, sibk, eigsh_mod

All arithmetic on n-vectors runs in hand-written HIP kernels (libeigd_hip.so, C ABI in
include/eigd_hip.h).  There is no CPU fallback: without the library or a GPU the calls raise.
"""
__version__ = "1.0.0"   # the reference's (eigd/__init__.py:1)

from . import tuning  # noqa: F401
from .adjoint import (  # noqa: F401
    add_eig_total_derivative,
    are_eigenvalues_repeated,
    dl,
    eval_adjoint_residual_norm,
    generate_adjoint_correction,
    laa,
    pcpg,
    pgmres,
    sibk,
)
from .lanczos import IRAM, BasicLanczos, eigsh_mod  # noqa: F401
from .operators import SpLuOperator  # noqa: F401
