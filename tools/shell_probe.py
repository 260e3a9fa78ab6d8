#!/usr/bin/env python3
"""Development aid: convergence of the restarted Lanczos on shell boxes of growing size (EIGD_TRACE_IRAM=1)."""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import eigd_amd as eg
from eigd_amd.problems import ShellBox, ShellBoxOnDevice
from test_gpu_shell import _find_shift
warnings.simplefilter("ignore")
nx, nw, nh, N, m = (int(v) for v in sys.argv[1:6])
box = ShellBox(nx, nw, nh, nseg=2, seed=0)
t0 = time.perf_counter()
dev = ShellBoxOnDevice(box)
dev.assemble()
sigma = _find_shift(dev, start=0.25)
print("n", box.n, "sigma", sigma, "setup", time.perf_counter() - t0, flush=True)
s = eg.IRAM(N=N, m=m, mode="buckling", ctx=dev.ctx, maxiter=int(sys.argv[6]) if len(sys.argv) > 6 else 60)
t0 = time.perf_counter()
try:
    lam, Phi = s.solve(dev.dG, dev.dK, dev.factor, sigma)
    print("eigensolve", time.perf_counter() - t0, "restarts", s.n_restarts, "lam", lam[:6], lam[-3:], flush=True)
    print("gaps", np.diff(lam).min(), flush=True)
except Exception as e:
    print("failed", time.perf_counter() - t0, repr(e)[:300], flush=True)
