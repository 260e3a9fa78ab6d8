#!/usr/bin/env python3
"""a few sweeps of the widths given on the command line (default 4 32) on the C3 factor, for kernel traces"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigd_amd.device import Factor, default_context  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

ctx = default_context()
col = BucklingColumn(706, 706, seed=0)
K = col.stiffness()
F = Factor(ctx, K, coords=col.dof_coords())
rng = np.random.default_rng(0)
for k in ([int(a) for a in sys.argv[1:]] or [4, 32]):
    B = ctx.from_host(rng.normal(size=(K.shape[0], k)))
    for _ in range(3):
        F.solve_inplace(B)
    ctx.sync()
