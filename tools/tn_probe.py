#!/usr/bin/env python3
"""timings of U^T X (eigd_gemm_tn: device result + host copy) for the shapes of the eigensolver and the projections
at the C3 size, direct against LDS-staged form (development aid): EIGD_TN_STAGED_MIN=64 forces the direct form"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigd_amd.device import default_context  # noqa: E402

ctx = default_context()
n = 998284
rng = np.random.default_rng(0)
U = ctx.from_host(rng.normal(size=(n, 64)))
out = []
for ku, kx in ((64, 8), (64, 4), (63, 8), (40, 8), (64, 16), (64, 32), (64, 64), (32, 32), (16, 8)):
    X = ctx.from_host(rng.normal(size=(n, kx)))
    Uv = U.cols(0, ku)
    for _ in range(3):
        Uv.tdot(X)
    ctx.sync()
    ctx.timer_start()
    for _ in range(20):
        Uv.tdot(X)
    us = ctx.timer_stop_ms() / 20 * 1e3
    out.append(f"({ku},{kx}): {us:6.1f} us {8 * n * (64 + kx) / us / 1e6:5.2f} TB/s")
print("  ".join(out))
